// WBFM chain as a streaming pipeline (gfx950): launch descriptor and host helpers.
//
// One persistent workgroup of 15 waves per CU.  The exact atan2 half table (129 rows, the reference's
// table is odd in y bit for bit: WbFmDemodulator.cc:159-170) lives in LDS for the lifetime of the
// workgroup.  A *segment* is what the tile kernels call a tile: a run of consecutive samples of one
// channel with its lead-in, cold-started and verified against its predecessor (wbfm_verify_kernel).
// The 192 segments of a round advance in lock step, 16 samples (a *window*) at a time:
//
//   12 P waves, 16 segments each: raw bytes -> signed, rotation signs (SDWA byte negation keeps
//       -(-128) = -128) -> both 16-tap Q15 rails as v_mfma_i32_16x16x64_i8 (taps split in a low and a
//       high byte plane, rotation's rail selection folded into the tap matrices) -> table index ->
//       ds_read_b32 gather -> delta theta, branch cut, K, b0 -> u[n] into an LDS ring slot
//   3 IIR waves, 64 segments each (one per lane): u[n] from the ring -> de-emphasis recurrence,
//       (int16), /4 /4 /2 Q15 decimators with their histories in registers -> PCM
//
// Reference: WbFmDemodulator.cc:383-562 behind IqDataProcessor.cc:735-749.
#pragma once
#include <stdint.h>

#include "iqd_device.h"

namespace iqd {

constexpr int ST_WAVES = 15;
constexpr int ST_THREADS = 64 * ST_WAVES;
constexpr int ST_RINGS = 3;                 // = IIR waves
constexpr int ST_P_PER_RING = 4;            // P waves feeding one ring (16 segments each)
constexpr int ST_SEGS = 64 * ST_RINGS;      // segments per workgroup and round when all its rings run (StreamArgs::rings)
constexpr int ST_HALO = FORCED_BACK;         // lead-in of every segment, 768 samples.  A warm segment (a call's first) runs
                                            // them from the carried exact state, which also rebuilds its decimator
                                            // histories.  A cold one uses them for the de-emphasis state to become exact
                                            // (checked against its predecessor's end state) and starts its decimators
                                            // with unknown histories: the 21 PCM samples those reach into are recomputed
                                            // from the boundary records (StHist) by wbfm_stream_fixup_kernel.
#ifndef IQD_ST_AHEAD
#define IQD_ST_AHEAD 2
#endif
constexpr int ST_AHEAD = IQD_ST_AHEAD;      // pieces of input a P wave keeps in flight (2 or 4: the piece loop is unrolled by it).
                                            // Round 4: 2 - the same kernel time as 4 (0.3288 against 0.3297 ms, five interleaved
                                            // runs on one box) with eight vector registers fewer; with 4 the launch that holds all
                                            // four families' pipelines (iqd_stream_mixed.hip, 5 registers of spilled scalars) spilled
                                            // vector registers to scratch, and that build ABORTED on the device (gpurun_out/dbg3.log):
                                            // tests/test_isa_lint.py now refuses any streaming kernel with a private segment
constexpr int ST_MIN_TILE = 768;            // a segment's own end histories must not reach back before its start
constexpr int ST_FIX_PCM = 21;              // PCM samples of a cold segment that depend on its predecessor's histories
constexpr int ST_ROW_FLOATS = 260;          // half-table row stride (1040 B: bank = x + 4 r)
#ifndef IQD_ST_FAKE_SHIFT
#define IQD_ST_FAKE_SHIFT 0                 // TIMING PROBES ONLY (wrong PCM): the LDS copy of the table with its columns 2^shift apart dropped
#endif
constexpr int ST_LDS_ROW_FLOATS = IQD_ST_FAKE_SHIFT ? (256 >> IQD_ST_FAKE_SHIFT) + 4 : ST_ROW_FLOATS;
constexpr int ST_TABLE_BYTES = 129 * ST_LDS_ROW_FLOATS * 4;
constexpr int ST_SLOT_BYTES = 64 * 16 * 4;  // one window of one ring: 64 segments x 16 samples, f32
#ifndef IQD_ST_DEPTH
#define IQD_ST_DEPTH 1
#endif
constexpr int ST_DEPTH = IQD_ST_DEPTH;      // pieces a ring holds (a power of two)
constexpr int ST_RING_SLOTS = 2 * ST_DEPTH;
constexpr int ST_SYNC_WORDS = 24;            // per ring: [0..3] `full` of its slots, [4] `consumed`
constexpr int ST_LDS_BYTES = ST_TABLE_BYTES + ST_RINGS * ST_RING_SLOTS * ST_SLOT_BYTES + ST_SYNC_WORDS * 4;
static_assert(ST_LDS_BYTES <= 160 * 1024, "table + rings must fit the CU's LDS");
static_assert(ST_HALO % 128 == 0 && ST_HALO + 32 <= TAIL, "the lead-in is whole 128-sample units inside the kept tail");

// What a segment leaves for the boundary with its neighbours (int16 values in pairs, older sample in the low half).
struct StHist {               // (every member 16-byte aligned: the IIR lanes write it with a few wide stores)
    uint32_t w_first[2];      // (int16)y of its samples 0..3
    uint32_t y1_first[6];     // stage-1 outputs 0..11  (output 0 reaches into the predecessor: recomputed)
    uint32_t y2_first[24];    // stage-2 outputs 0..47  (outputs 0..2 likewise; 0..41 are used)
    uint32_t y1_last[4];      // its last 8 stage-1 outputs
    uint32_t w_last[2];       // (int16)y of its last 4 samples
    uint32_t pad[6];
    uint32_t y2_last[20];     // its last 40 stage-2 outputs
};
static_assert(sizeof(StHist) == 256, "one boundary record per segment, 256 bytes");

struct StreamArgs {
    StHist *hist;             // [n_segments]
    const uint32_t *amat;     // [8][64][4]: tap matrices as MFMA A operands, see build_stream_amat()
    const float *half_lut;    // [129][ST_ROW_FLOATS]: |atan2(-r, x - 128)|
    uint32_t n_segments;      // n_list * tiles_per_ch
    uint32_t rounds;          // rounds per workgroup
    uint32_t rings;           // rings of 64 segments a workgroup runs, 1 .. ST_RINGS (round 5: a small launch spreads its segments over
                              // all CUs as workgroups of one or two rings - a ring's pieces go by faster with fewer rings beside it -
                              // instead of filling a third of the CUs with three; the waves of the other rings leave at once)
    // Channels of several rotation selectors in one launch (round 4; the kernel instantiation with ROT = 2): the channel
    // list is sorted by selector (+Fs/4, none, -Fs/4) and each group's segment ids are padded to a multiple of 16, so that
    // a P wave's 16 segments share their tap matrices - as the FM / AM / SSB pipelines do (D4Args).  grouped = 0: ids are
    // li * tiles_per_ch + tile and `amat` is the launch's one selector.
    uint32_t grouped;
    uint32_t group_start[4];  // first id of each group; [3] = end of the ids
    uint32_t group_li0[3];    // first channel-list index of each group
    uint32_t group_nseg[3];   // real segments of each group
    const uint32_t *amat3[3]; // tap matrices of the selectors +1, 0, -1
    uint32_t d1p[4];          // decimator taps as v_dot2 pairs (see build_stream_taps)
    uint32_t d1p2[4];         // stage 1 again with doubled taps (the IIR lanes: result in the accumulator's high half)
    uint32_t p12p[6];
    uint32_t a40p[20];
    float b0, a1;
};

// FM / AM / SSB as streaming pipelines (iqd_stream2.hip): the chain's first /4 decimator on the matrix cores in
// P waves, everything behind it in consumer lanes.
enum { D4_AM = 0, D4_SSB = 1, D4_FM = 2 };
constexpr int D4_HALO_AM = fir_halo(FAM_AM), D4_HALO_SSB = fir_halo(FAM_SSB), D4_HALO_FM = fir_halo(FAM_FM);   // FULL lead-in samples (the chains need 260 / 1220 / 684): a channel's first segment, from the kept tail
// Round 6: FM / AM / SSB segments without their long lead-ins.  Every stage of these chains is a FIR, so a segment that starts
// cold is exact once its filters' windows lie inside what it has run itself - rounds 2-5 gave EVERY segment the lead-in that
// takes (AM 384, FM 768, SSB 1280 samples: 7 / 14 / 23 % of a 5.5 k-sample segment, a third to a half of the pieces at the
// reference's own operating point, one 64 ms block per call: DataConsumer.cc:333-346).  The reference carries that state in its
// filters' ring buffers from call to call (Decimator_int16.cc:310-351, FirFilter_int16.cc:151-213); here it travels from segment to
// segment of one call: a channel's segments have consecutive segment ids, i.e. they sit in neighbouring lanes of a consumer
// wave, so at the end of its run every lane takes the end state of the lane below (DPP) and replays its own first outputs -
// 4 (AM), 36 (SSB), 20 (FM) - from inputs it kept in LDS (iqd_stream2.hip: d4_am_wave, d4_fm_wave).  Such a WARM segment runs
// D4_HALO_SHORT = 128 samples of lead-in (what the first two stages need).  A COLD one - a channel's first segment, and every
// segment that is lane 0 of a consumer wave (segment id a multiple of 64: its predecessor sits in another wave) - has nobody
// to take a state from and runs the family's full lead-in.  So that all segments of a launch still run the same number of pieces
// (they advance in lock step), a cold segment's lead-in comes out of its own length: it starts D = full lead-in - 128 samples
// earlier than its outputs (D4Args::lead_shift), the segments behind it move up by D, and what it computes before its first
// output is not stored.  Segment t of a channel whose segment 0 has id sid0:
//     cold(t)  = t == 0 || (sid0 + t) % 64 == 0
//     c(t)     = cold segments among 0 .. t = 1 + floor((sid0 + t) / 64) - floor(sid0 / 64)
//     v0(t)    = t * tile_len - D * c(t)          nominal start: the run is [v0 - 128, v0 + tile_len)
//     skip(t)  = cold(t) ? D : 0                  outputs from v0 + skip on
// (tile 0 starts before the call's first sample: that part is the kept tail.)  Host + device.
constexpr int D4_HALO_SHORT = 128;
constexpr int d4_full_halo(int family) { return family == FAM_FM ? D4_HALO_FM : family == FAM_AM ? D4_HALO_AM : D4_HALO_SSB; }
constexpr int d4_lead_shift(int family) { return d4_full_halo(family) - D4_HALO_SHORT; }
// The boundary replay of a warm segment (iqd_stream2.hip: d4_am_wave, d4_fm_wave; modelled stage by stage in
// tests/test_emu_d4_handoff_model.py).  Pieces count from the start of the run: the 128-sample lead-in is pieces 0..3, the segment's
// first output belongs to piece 4.  A consumer lane starts its run with empty histories; the P waves hand it exact values from
// the run's first sample on (their first piece has the true piece before it for its window; FM: except the run's first two
// discriminator outputs, which would need the phase angles in front of the run).  Stage 2 (/4, 12 taps: a piece's pair needs the
// 8 values of the piece before) is therefore exact from piece 1 on for AM / SSB, from piece 2 on for FM.
//   AM   stage 3 (/2, 16 taps = the pairs of 8 pieces) is exact in the loop from piece 8 on: the outputs of pieces 4..7 are
//        replayed from their kept pairs behind the predecessor's last 7;
//   SSB  the detector (i delayed by 15, Hilbert over 31 of q at 8 kS/s) reaches back 30 outputs: outputs 0..33 need the
//        predecessor's rails - the first 4 come out of the replayed stage 3, the rails of pieces 8..39 are kept as well, and
//        36 outputs (quads) are replayed;
//   FM   the last stage (/2, 40 taps = the pairs of 20 pieces) is exact in the loop from piece 22 on: the 20 outputs of pieces
//        4..23 are replayed from their kept pairs behind the predecessor's last 20.
constexpr int D4_REPLAY_AM = 4, D4_REPLAY_SSB = 36, D4_REPLAY_FM = 20;   // outputs replayed (multiples of 4: a quad is one store)
constexpr int D4_REPLAY_PAIRS_AMSSB = 4;                                  // AM / SSB: pieces 4..7 keep their stage-2 pairs of both rails
constexpr int D4_RAILS_FROM_PIECE = 8;                                    // SSB: pieces 8..39 keep their rails
constexpr int d4_replay_outputs(int family) { return family == FAM_FM ? D4_REPLAY_FM : family == FAM_AM ? D4_REPLAY_AM : D4_REPLAY_SSB; }
struct D4Geom { int64_t v0; uint32_t skip, cold; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline D4Geom d4_geom(uint32_t sid, uint32_t tile, uint32_t tile_len, uint32_t shift)
{
    D4Geom g;
    if (!shift) { g.v0 = (int64_t)tile * tile_len; g.skip = 0; g.cold = 1; return g; }   // (every segment with its full lead-in: rounds 2-5)
    g.cold = tile == 0 || (sid & 63u) == 0 ? 1u : 0u;
    const uint32_t c = 1u + (sid >> 6) - ((sid - tile) >> 6);
    g.v0 = (int64_t)tile * tile_len - (int64_t)shift * c;
    g.skip = g.cold ? shift : 0u;
    return g;
}
// cold segments a channel of n_tiles segments can hold at most (whatever its first id): its first one + the lane-0 ones
constexpr uint32_t d4_max_cold(uint32_t n_tiles) { return 1u + (n_tiles + 62u) / 64u; }
struct D4Args {
    const uint32_t *amat;        // [3 rotation selectors -1, 0, +1][4][64][4]: build_decim4_amat()
    const float *fm_lut;         // 283 x 283 phase angles
    uint32_t group_start[4];     // segment ids of the rotation groups (+Fs/4, none, -Fs/4), each padded to 16; [3] = end
    uint32_t group_li0[3];       // first channel-list index of each group (the list is sorted by group)
    uint32_t group_nseg[3];      // real segments in each group
    uint32_t rounds;
    uint32_t rings;              // rings a workgroup runs (StreamArgs::rings)
    int32_t halo;                // lead-in samples every segment of the launch runs (the family's full lead-in, or D4_HALO_SHORT)
    uint32_t lead_shift;         // round 6 (d4_geom above): full lead-in - halo; 0 = every segment with its full lead-in
    uint32_t s2p[6], s3p[8];     // AM/SSB stage 2 (12 taps, DOUBLED: the result is the accumulator's high half) and stage 3 (16 taps) as v_dot2 pairs, newest pair first
    uint32_t hilb[16];           // SSB: the nonzero Hilbert taps h[0], h[2], ..., h[30] (int16 in the low half)
    uint32_t p12p[6], a40p[20];  // FM post-discriminator decimators
};
// [0] this piece, low tap byte; [1] this piece, high byte; [2] previous piece, low; [3] previous piece, high
void build_decim4_amat(int rotation, const int16_t *taps_q15, int ntaps, uint32_t *out /* [4*64*4] */);
void build_d4_taps(const Consts &c, D4Args &da);

// A operands of v_mfma_i32_16x16x64_i8 for the pre-demodulation FIR of one rotation selector.
// Matrix index = 4 * window_type + 2 * rail + plane; window_type 0 = "N" (the 16 outputs of the second
// half of a 32-sample piece, operand = the piece), 1 = "S" (first half: lanes 0-31 hold the piece's first
// 32 bytes, lanes 32-63 the previous piece's last 32 bytes); rail 0 = I', 1 = Q'; plane 0 = low byte,
// 1 = high byte of the DOUBLED Q15 tap (2 h = lo + 256 hi).
void build_stream_amat(int rotation, const int16_t *pre_q15, uint32_t *out /* [8*64*4] */);
bool build_half_lut(float *out /* [129*ST_ROW_FLOATS] */);   // false: libm's atan2 is not odd in y bit for bit
void build_stream_taps(const int16_t *d1, const int16_t *post12, const int16_t *audio40, StreamArgs &sa);

}  // namespace iqd
