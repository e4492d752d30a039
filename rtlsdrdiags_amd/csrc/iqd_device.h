// Shared host/device definitions of the IQ demodulation engine (gfx950).
#pragma once
#include <stdint.h>

namespace iqd {

// ---- geometry -------------------------------------------------------------------------
// A channel's demodulator sees one continuous stream: the concatenation of its squelch-open
// blocks.  Positions are counted in complex samples.  The engine keeps the last samples
// each demodulator family consumed ("tail": as many as the family reaches back for, at most TAIL) instead of per-filter ring buffers: every FIR /
// decimator history in the chains is then a pure function of raw bytes, and only the float
// recurrences carry explicit state.
constexpr int TAIL = 2048;          // raw samples a channel's slot per demodulator family holds (the family keeps tail_keep() of them, below)
constexpr int TAIL_BYTES = 2 * TAIL;
constexpr int SEG = 128;            // de-emphasis IIR segment (one lane's run)
constexpr int FORCED_BACK = 768;    // an exact IIR restart point lies this far before a tile
constexpr int COLD_HALO = 2048;     // zero-state warm-up distance of interior tiles
#ifndef IQD_WBFM_NSEG
#define IQD_WBFM_NSEG 55
#endif
constexpr int WBFM_NSEG = IQD_WBFM_NSEG;  // segments per chunk (55 -> 7040 samples = 440 groups of 16:
                                          // seven wave passes of 63 groups; 3 workgroups per CU)
constexpr int WBFM_CHUNK = WBFM_NSEG * SEG;
constexpr int TSTRIDE = 132;        // dwords between segments of the IIR input buffer
constexpr int WSTRIDE = 68;         // dwords between segments of the int16 output buffer (16-B aligned)

constexpr int FM_LUT_R = 141;        // the FM tuner decimator fed with int8 cannot leave [-141, 141]
constexpr int FM_LUT_W = 2 * FM_LUT_R + 1;

enum Family { FAM_AM = 0, FAM_FM = 1, FAM_WBFM = 2, FAM_SSB = 3, FAM_COUNT = 4 };
// Raw history a tile / a cold streaming segment of the FIR chains rebuilds its filter states from (the chains need 260 / 684 / 1220
// samples: AM 8 + 4 * 12 + 16 * 16 taps deep, FM 32 + 4 * 12 + 16 * 40, SSB the AM front + 32 * 31), and what the closing launch
// keeps of a channel's stream for the next call: the lead-in, the piece in front of it the P waves' first window lies in, and
// slack to 128 (round 6: it was the whole TAIL for every family - 4 KiB read and 4 KiB written per channel and call, a quarter
// of the input at one 64 ms block per call; now 1 / 1.75 / 2.75 KiB).  WBFM keeps the whole TAIL (restart points, cold warm-ups).
constexpr int fir_halo(int family) { return family == FAM_AM ? 384 : family == FAM_FM ? 768 : 1280; }
constexpr int tail_keep(int family) { return family == FAM_WBFM ? TAIL : fir_halo(family) + 128; }
static_assert(tail_keep(FAM_SSB) <= TAIL && tail_keep(FAM_AM) % 8 == 0 && tail_keep(FAM_FM) % 8 == 0 && tail_keep(FAM_SSB) % 8 == 0, "kept tails");

// ---- per-channel parameters (host mirror uploaded when dirty) ---------------------------
struct ChanParams {
    int32_t mode;           // iqd_mode
    int32_t threshold;      // squelch threshold, dBFS
    uint32_t rx_gain_db;
    int32_t rotation;       // +1 / 0 / -1
    int32_t ssb_lsb;        // SsbDemodulator::lsbDemodulationMode
    float gain[4];          // demodulatorGain per family
    float wbfm_k;           // (gain / 75000) * 32767, evaluated in binary32 on the host
    float fm_k;             // (gain / 15000) * 32767
    // A demodulator gain changes between two accept calls, i.e. at the start of a call's data.  The histories a tile
    // rebuilds from the raw tail belong to the time before it: they are computed with the gain that was in force
    // then (k_prev), `since` samples back from the stream end (GainEpoch: a list of such changes, maintained on the device).
    float wbfm_k_prev, fm_k_prev;   // (host -> device hand-over: the K the device last ran with)
    uint32_t k_changed;     // one-shot: bit 0 WBFM, bit 1 FM - the gain differs from the last accept's; bit 2: the rotation does
    int32_t rotation_prev;  // the rotation the device last ran with (for bit 2)
    uint32_t pad;
};
// Gain changes whose samples are still inside the kept tail, most recent first.  A gain can only change between
// two accepts and an accept consumes at least 32 samples (one 64-byte unit; a call the squelch rejects consumes nothing
// and its change replaces the one before it), so TAIL / 32 entries cover every possible history.
#ifndef IQD_EPOCH_UNIT
#define IQD_EPOCH_UNIT 32
#endif
constexpr int EPOCHS = TAIL / IQD_EPOCH_UNIT;    // (a call consumes at least 32 samples - one 64-byte unit - of the 2048 the tail keeps)
struct GainEpochList {
    uint32_t since[EPOCHS];    // samples the family has consumed since change i (saturates at TAIL = out of reach)
    float k_before[EPOCHS];    // the K in force before change i
};
struct GainEpoch { GainEpochList wbfm, fm; };   // per channel, on the device

// ---- per-channel carried state ------------------------------------------------------------
struct WbfmCarry {          // exact de-emphasis state `back` samples before the stream end
    float y;                // IirFilter output y[end - back - 1]
    float u;                // b1 * x[end - back - 1] (numerator history)
    int32_t back;           // restart distance (FORCED_BACK once the stream is long enough)
    float y_end, u_end;     // the same at the stream end (what resetDemodulator() keeps)
    uint32_t pad[3];
};
struct DcCarry { float x_prev, y_prev; };   // AM / SSB DC-removal IIR

// AutomaticGainControl (hdr_diags/AutomaticGainControl.h:57-89), one per channel.  The configuration is written
// by the host; the state lives on the device and moves once per block, in block order.
struct AgcConfig {
    uint32_t enabled, type;          // AGC_TYPE_LOWPASS 0 / AGC_TYPE_HARRIS 1
    int32_t operating_point, deadband;
    float alpha;
    uint32_t blanking_limit;
    uint32_t reset_blanking;         // one-shot: resetBlankingSystem() before the next block
    uint32_t set_gain;               // one-shot: Radio::setReceiveIfGainInDb (0xffffffff = none)
};
struct AgcState {
    uint32_t rx_gain;                // the receiver's IF gain = radio_adjustableReceiveGainInDb (the squelch reads it)
    uint32_t if_gain;                // AutomaticGainControl::ifGainInDb
    float filtered;                  // filteredIfGainInDb
    uint32_t blank_ctr, adjusted;    // blankingCounter, receiveGainWasAdjusted
    int32_t normalized;              // normalizedSignalLevelInDbFs
    uint32_t signal_magnitude;
    uint32_t pad;
};
// FrequencyScanner (hdr_diags/FrequencyScanner.h:33-75), one per channel: configuration from the host,
// the current frequency on the device (it moves on every block the squelch rejects).
struct ScanConfig {
    unsigned long long start_hz, end_hz, increment_hz;
    unsigned long long set_current;   // one-shot value for `current` (start() after new parameters)
    uint32_t scanning;
    uint32_t set_current_flag;        // one-shot: also counts as one tuning command
};
struct ScanState { unsigned long long current_hz; unsigned long long tune_count; };
constexpr uint32_t AGC_MAX_GAIN = 46;   // MAX_ADJUSTIBLE_GAIN, AutomaticGainControl.cc:23

struct VerifyRec { float y_in, y_out, u_out; uint32_t flags; };

// ---- constants (taps, packed the way the kernels consume them) ----------------------------
struct Consts {
    // WBFM pre-demod 16-tap FIR for v_dot4_i32_i8, accumulator DOUBLED: pre_lo holds 2*lo with
    // h = 128*hi + lo; dword q covers window bytes 4q..4q+3 (oldest sample first).
    int32_t pre_lo[4], pre_hi[4];
    // Q15 taps of the decimators, as int16 (index = k, newest sample first)
    int16_t wbfm_d1[8], post12[12], audio40[40];
    int16_t fm_tuner[32];
    int16_t am_s1[8], am_s2[12], am_s3[16];
    int16_t ssb_delay[16], ssb_hilbert[31], pad0;
    // dot4 packings (low / high tap bytes, dword q = window bytes 4q..4q+3, oldest first)
    int32_t fm_tuner_lo[8], fm_tuner_hi[8];
    int32_t am_s1_lo[2], am_s1_hi[2];
    int32_t db_table[257];
    float deemph_b0, deemph_a1;     // 0.0253863, -0.9492274
    float deemph_c, deemph_c16, deemph_c127, deemph_c128, deemph_cinv;  // powers of -a1 (state guess only)
    float dc_a1;                    // -0.95
    float dc_cseg;                  // 0.95^32: decay over one DC segment (state guess only)
};

}  // namespace iqd
