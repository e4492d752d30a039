/* include/iqdemod.h — C-ABI of libiqdemod.so, the MI355X-native IQ demodulation engine.
 *
 * This is the drop-in boundary for the RtlSdrDiags per-sample DSP hot path.  Each entry
 * point names the reference interface it replaces (paths relative to the reference's
 * radioDiags/ directory).  Signatures are plain C: pointers, sizes and scalars only.
 *
 * One engine owns N independent channels.  Each channel behaves like one reference
 * IqDataProcessor with its four demodulators attached (hdr_diags/IqDataProcessor.h:16-93):
 * it keeps its own mode, gains, squelch threshold/tracker and per-demodulator filter state.
 * A channel's input is the same offset-binary uint8 interleaved I/Q the reference receives
 * (256 kS/s) and its output the same 8 kS/s S16 PCM (bytes/64 samples per accepted block).
 *
 * Threading (IqDataProcessor.h:35-37, DataConsumer.cc:342): one thread at a time may be
 * inside iqd_accept_*() for a given engine; setters may be called from another thread
 * between or during accepts and take effect at the next accept (they are mutex-guarded).
 *
 * All functions returning int return IQD_OK (0) or a negative IQD_E* code.
 */
#ifndef IQDEMOD_H
#define IQDEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IQD_ABI_VERSION 1

/* IqDataProcessor::demodulatorType, hdr_diags/IqDataProcessor.h:20 */
enum iqd_mode {
    IQD_MODE_NONE = 0, IQD_MODE_AM = 1, IQD_MODE_FM = 2,
    IQD_MODE_WBFM = 3, IQD_MODE_LSB = 4, IQD_MODE_USB = 5
};

/* Which demodulator a gain applies to: {Am,Fm,WbFm,Ssb}Demodulator::setDemodulatorGain */
enum iqd_demod { IQD_DEMOD_AM = 1, IQD_DEMOD_FM = 2, IQD_DEMOD_WBFM = 3, IQD_DEMOD_SSB = 4 };

enum iqd_status {
    IQD_OK = 0,
    IQD_EINVAL = -1,      /* bad argument (range, alignment, NULL) */
    IQD_ENODEV = -2,      /* no usable HIP device / HIP runtime error at create */
    IQD_ENOMEM = -3,      /* host or device allocation failed */
    IQD_EHIP = -4,        /* a HIP call failed during accept (see iqd_last_error) */
    IQD_ESTATE = -5,      /* exact-state verification could not be repaired (never expected) */
    IQD_EALREADY = -6     /* iqd_agc_enable: every channel of the range already was in that state */
};

typedef struct iqd_engine iqd_t;

typedef struct iqd_config {
    uint32_t abi_version;   /* IQD_ABI_VERSION */
    uint32_t n_channels;    /* >= 1 */
    uint32_t block_bytes;   /* squelch / acceptIqData granularity per channel.  0 -> 32768
                               (Radio.cc:16, 1895).  Multiple of 256, <= 32768 (the cap of
                               SignalDetector.h:49).  A call SHORTER than this is one short block
                               (a short USB read, Radio.cc:1895-1906): any multiple of 64 bytes - 32
                               samples, one PCM sample, the period of the chains' /32 commutators
                               (the reference itself strides 8 bytes, IqDataProcessor.cc:567-611) -
                               in every mode. */
    int32_t device;         /* HIP device ordinal, -1 -> current device */
    uint32_t flags;         /* IQD_F_* */
    uint32_t reserved[3];
} iqd_config;

#define IQD_F_NO_MAGNITUDE 0x1u /* do not produce per-block magnitudes when no channel's squelch can close
                                   (the reference only reports them through an optional callback) */
/* Every chain (WBFM, FM, AM, SSB) exists as two kernels with identical results: tiles (one workgroup per run of
 * samples) and a streaming pipeline (one persistent workgroup per CU, used for launches big enough to fill the chip,
 * squelch-gated ones included).  These two flags pin the choice for all chains (tests, A/B measurements; the names
 * are from the round in which only WBFM had both); the environment variable IQD_WBFM_PATH=tiles|stream does the same.
 * A call with several families whose streaming pipelines all apply runs them as ranges of ONE launch's workgroups
 * (stats.mixed_launches), each family on a share of the CUs chosen so that all of them end together (per-family segment
 * lengths and lead-ins taken into account; IQD_SHARES=cost: in plain proportion to the estimated cost, the earlier rule).
 * More environment variables exist for measurements only, read once by iqd_create: IQD_MIXED=forked runs such a call's
 * families as kernels of their own on side streams instead (the round-2 arrangement), IQD_FULL_GRID=1 then gives every
 * family all CUs in turn instead of a share of them side by side, IQD_FAMILY_WEIGHTS=am,fm,wbfm,ssb replaces the
 * relative cost estimates, IQD_STREAM_WGS=<n> fixes the streaming kernels' workgroup count, IQD_STREAM_GRAN / IQD_D4_GRAN=128|256|512
 * the granule of the WBFM / the other pipelines' segment lengths (defaults 512 / 128), IQD_STREAM_MIN_SEG=<samples> replaces the
 * measured per-family thresholds (samples a launch must bring per segment of the persistent workgroups before it streams),
 * IQD_AM_STREAM_MIN=<PCM samples> the shortest AM / SSB row that may stream (128), IQD_FAMILY_NS=am,fm,wbfm,ssb the ns per
 * segment sample the share planner works with. */
#define IQD_F_WBFM_TILES  0x2u
#define IQD_F_WBFM_STREAM 0x4u
/* Squelch-gated iqd_accept_iq_device calls (some channel's threshold can reject a block) read their input twice: a pre-pass
 * takes every block's magnitude and makes the decisions (Squelch.cc:227-273), then the pipelines walk the open blocks.  With
 * this flag the pre-pass of a call runs on a stream of the engine's own, ordered only behind the PREVIOUS call's pre-pass, so
 * that it overlaps the previous call's pipelines (a continuous stream of calls: magnitudes one call ahead).  The caller
 * promises in return that `iq_dev` is complete when the call is MADE - ordering its producer on iqd_stream() is not enough,
 * the pre-pass does not wait there - and that it does not reuse the small output buffers (pcm_count / magnitude /
 * signal_present) of the call before.  Host-pointer calls ignore the flag.  Results are identical. */
#define IQD_F_PREPASS_OVERLAP 0x8u

/* Replaces: new IqDataProcessor(...) + new {Am,Fm,WbFm,Ssb}Demodulator(pcmCallback) +
 * set*Demodulator() wiring, Radio.cc:150-181.  Every channel starts like the reference:
 * mode None (IqDataProcessor.cc:38), squelch threshold -200 dBFS (:41), receive gain 24 dB
 * (Radio.cc:325-328), default demodulator gains, zero filter state. */
int iqd_create(const iqd_config *cfg, iqd_t **out);
void iqd_destroy(iqd_t *e);

/* Replaces IqDataProcessor::setDemodulatorMode, IqDataProcessor.cc:236-262
 * (Lsb/Usb also select the SsbDemodulator sideband, :244-256). */
int iqd_set_mode(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int mode);

/* Replaces {Am,Fm,WbFm,Ssb}Demodulator::setDemodulatorGain (e.g. WbFmDemodulator.cc:341-348).  Takes effect with the
 * first sample of the next accept; what the filters already hold keeps the gain it was made with, exactly as in the
 * reference, for ANY sequence of changes: the engine keeps every change whose samples can still reach a filter
 * history (up to 64 per channel and demodulator - a change needs an accept of >= 32 samples in between, and the
 * histories reach back 2048 samples). */
int iqd_set_gain(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod, float gain);

/* Replaces IqDataProcessor::setSignalDetectThreshold, IqDataProcessor.cc:284-295. */
int iqd_set_squelch(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int32_t threshold_dbfs);

/* Replaces the global radio_adjustableReceiveGainInDb read at IqDataProcessor.cc:765
 * (per channel here, because each channel stands for its own receiver). */
int iqd_set_rx_gain_db(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t gain_db);

/* Reads the IF gain in force for a channel (it differs from the last value set once the channel's AGC has run). */
int iqd_get_rx_gain_db(iqd_t *e, uint32_t ch, uint32_t *gain_db);

/* Replaces AutomaticGainControl (hdr_diags/AutomaticGainControl.h:22-47, src_diags/AutomaticGainControl.cc), one
 * per channel, constructed as Radio.cc:184 does (operating point -12 dBFS, Harris type, alpha 0.8, deadband 1,
 * blanking limit 1, disabled).  An enabled AGC receives the channel's block magnitudes in block order - the
 * reference's signalMagnitudeCallback, IqDataProcessor.cc:781-790 -> AutomaticGainControl.cc:47-64 - and moves
 * the channel's IF gain (Radio::setReceiveIfGainInDb, Radio.cc:817-868), which the squelch of the NEXT block
 * compares with (IqDataProcessor.cc:765).  The recorded samples themselves do not change with the gain: there
 * is no tuner behind a channel here.  Each setter mirrors the reference method of the same name and returns
 * IQD_EINVAL where that method returns false (setType :287-320, setDeadband :351-369, setBlankingLimit :399-420,
 * setOperatingPoint :440-448, setAgcFilterCoefficient :475-493, enable/disable :516-595). */
enum iqd_agc_type { IQD_AGC_LOWPASS = 0, IQD_AGC_HARRIS = 1 };
int iqd_agc_set_type(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t type);
int iqd_agc_set_deadband(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t deadband_db);     /* 0..10 */
int iqd_agc_set_blanking_limit(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t limit);      /* 0..10 */
int iqd_agc_set_operating_point(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int32_t dbfs);
int iqd_agc_set_filter_coefficient(iqd_t *e, uint32_t first_ch, uint32_t n_ch, float coefficient); /* [0.001, 0.999) */
int iqd_agc_enable(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int enabled);   /* IQD_EALREADY where enable()/disable() return false */

/* displayInternalInformation() (AutomaticGainControl.cc:1082-1149) as values. */
typedef struct iqd_agc_state {
    uint32_t enabled, type;
    int32_t operating_point_dbfs;
    uint32_t deadband_db, blanking_limit;
    float alpha;
    uint32_t rx_gain_db;            /* the receiver's IF gain in force */
    uint32_t if_gain_db;            /* the AGC's own copy (ifGainInDb) */
    float filtered_if_gain_db;
    uint32_t blanking_counter, gain_was_adjusted;
    int32_t normalized_level_dbfs;
    uint32_t signal_magnitude;      /* last magnitude the AGC acted on */
} iqd_agc_state;
int iqd_agc_get_state(iqd_t *e, uint32_t ch, iqd_agc_state *out);

/* Replaces FrequencyScanner (hdr_diags/FrequencyScanner.h:33-75, src_diags/FrequencyScanner.cc), one per channel,
 * idle at 162.55 MHz like the reference's constructor (:96-131).  A scanning channel advances its frequency by the
 * increment on every block its squelch rejects - the reference's signalStateCallback -> run(), :47-62, :378-404 -
 * and wraps from the end to the start frequency; every such step is one Radio::setReceiveFrequency command, which
 * the host forwards to its tuner (nothing retunes recorded bytes).  iqd_scanner_set_parameters mirrors
 * setScanParameters (:190-218, refused while scanning), iqd_scanner_start(…, 1/0) mirrors start()/stop()
 * (:240-310; the first start after new parameters jumps to the END frequency); both return IQD_EALREADY where the
 * reference returns false for every channel of the range. */
int iqd_scanner_set_parameters(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint64_t start_hz, uint64_t end_hz,
                               uint64_t increment_hz);
int iqd_scanner_start(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int start);
/* The frequency the channel's radio was last told to tune to, the number of tuning commands so far, isScanning(). */
int iqd_scanner_get(iqd_t *e, uint32_t ch, uint64_t *current_hz, uint64_t *tune_count, int *scanning);

/* Optional record of the IF gain each block's squelch compared with, for the channels and blocks of the last
 * accept call: [n_ch][n_blocks], host memory. */
int iqd_set_gain_trace(iqd_t *e, int enabled);
int iqd_get_gain_trace(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t *gain_db, size_t n_blocks);
/* With the same switch: the frequency each channel was tuned to after each block, [n_ch][n_blocks]. */
int iqd_get_frequency_trace(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint64_t *hz, size_t n_blocks);

/* Front-end rotation selector: +1 = upconvertByFsOver4 (what acceptIqData applies,
 * IqDataProcessor.cc:749, the default), -1 = downconvertByFsOver4 (:496-540), 0 = none. */
int iqd_set_rotation(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int rotation);
/* (The reference has no such switch.  A selector changed in mid-stream takes effect with the first sample of the next
 * accept; what the filters already hold keeps the old rotation, as in-place rotation at acceptance time would.) */

/* Replaces {Am,Fm,WbFm,Ssb}Demodulator::resetDemodulator() for all four demodulators of the
 * channels (note WbFmDemodulator.cc:304-320 leaves the de-emphasis filter state alone). */
int iqd_reset(iqd_t *e, uint32_t first_ch, uint32_t n_ch);
/* The same for one demodulator only (demod = IQD_DEMOD_*), as the reference's per-object call. */
int iqd_reset_demod(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod);

/* Replaces IqDataProcessor::acceptIqData(timeStamp, bufferPtr, byteCount),
 * IqDataProcessor.cc:722-840, for n_ch channels at once, each receiving
 * bytes_per_ch / block_bytes consecutive calls' worth of data.
 *
 *   iq             [n_ch][bytes_per_ch] uint8, channel-major; NOT modified (the reference
 *                  mutates its buffer in place; callers that relied on that must not).
 *   bytes_per_ch   multiple of block_bytes, or one short block (see iqd_config::block_bytes).
 *   pcm            [n_ch][bytes_per_ch/64] int16; channel c's samples are packed at the front
 *                  of its row (squelched blocks contribute nothing, like the reference's
 *                  callback, IqDataProcessor.cc:793).
 *   pcm_count      [n_ch] number of valid PCM samples per channel (may be NULL).
 *   magnitude      [n_ch][bytes_per_ch/block_bytes] SignalDetector::signalMagnitude per block
 *                  = what registerSignalMagnitudeCallback would deliver (may be NULL).
 *   signal_present [n_ch][bytes_per_ch/block_bytes] Squelch::run() result per block = what
 *                  registerSignalStateCallback would deliver (may be NULL).
 *
 * Host-pointer form: copies in, runs, copies out, returns when done (large calls are sliced and
 * double-buffered so that the copies overlap the kernels; the result is the same). */
int iqd_accept_iq(iqd_t *e, uint32_t first_ch, uint32_t n_ch,
                  const uint8_t *iq, size_t bytes_per_ch,
                  int16_t *pcm, uint32_t *pcm_count,
                  uint32_t *magnitude, uint8_t *signal_present);

/* Device-pointer form: every pointer is HIP device memory (same layouts); the work is
 * enqueued on the engine's stream and the call returns as soon as it is queued (the WBFM
 * hand-off verification and, if ever needed, its repair run on the device).  Keep the
 * buffers alive and use iqd_synchronize() - or synchronise iqd_stream() - before reading
 * results.  Until then the PCM rows are the call's own: the AM / SSB pipelines keep intermediate
 * (detector) values there before the DC-removal pass turns them into PCM in place. */
int iqd_accept_iq_device(iqd_t *e, uint32_t first_ch, uint32_t n_ch,
                         const void *iq_dev, size_t bytes_per_ch,
                         void *pcm_dev, void *pcm_count_dev,
                         void *magnitude_dev, void *signal_present_dev);

int iqd_synchronize(iqd_t *e);

/* Replaces {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData(int8_t *bufferPtr, uint32_t bufferLength) - AmDemodulator.h:30,
 * FmDemodulator.h:30, WbFmDemodulator.h:31 (WbFmDemodulator.cc:383-411), SsbDemodulator.h:33 - the demodulator's own
 * entry, which the processor calls behind its squelch (IqDataProcessor.cc:793-836) and the reference's offline harness
 * calls directly (demodulatorResearch/demodulators/demod.cc:262-285):
 *   iq            [n_ch][bytes_per_ch] SIGNED interleaved I/Q as the processor hands it over (after the -128 and the
 *                 rotation); NOT modified (the reference's WBFM chain filters its buffer in place)
 *   bytes_per_ch  any positive multiple of 64 (32 samples = one PCM sample); the reference's own limit of 32768 per call
 *                 (demodulatedData[16384]) does not apply
 *   pcm           [n_ch][bytes_per_ch/64]: every sample reaches the demodulator - no squelch, no signal or magnitude
 *                 notification, no AGC or scanner step; the channel's squelch tracker does not move
 * The demodulators are the ones the channel's processor uses (Radio.cc:150-181): their filter state is shared with
 * iqd_accept_iq of the same channel, in either order.  demod = IQD_DEMOD_*; the SSB sideband is the one last selected
 * by iqd_set_mode(LSB / USB) or iqd_demod_set_sideband - SsbDemodulator::setLsbDemodulationMode / setUsbDemodulationMode
 * (SsbDemodulator.cc:333-367; the reference's constructor starts in LSB, :143). */
int iqd_demod_accept(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod,
                     const int8_t *iq, size_t bytes_per_ch, int16_t *pcm);
int iqd_demod_set_sideband(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int lsb);

/* The front end alone, IqDataProcessor.cc:735-749: s = u - 128, then the channel's rotation; [n_ch][bytes_per_ch]
 * signed bytes out.  This is what the reference leaves in the caller's buffer (it works in place) and what its
 * IQ dump tap streams (:756-760, UdpClient::sendData); iqd_accept_* itself never modifies its input. */
int iqd_front_end(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const uint8_t *iq, size_t bytes_per_ch, int8_t *out);
int iqd_front_end_device(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const void *iq_dev, size_t bytes_per_ch,
                         void *out_dev);
/* Replaces IqDataProcessor::upconvertByFsOver4 (direction +1, IqDataProcessor.cc:567-611) and
 * downconvertByFsOver4 (direction -1, :487-531; hdr_diags/IqDataProcessor.h:32-33) as stand-alone calls: `buffer`
 * holds SIGNED interleaved I/Q, is rotated in place (-(-128) stays -128), byte_count a multiple of 8. */
int iqd_convert_fs_over_4(iqd_t *e, int direction, int8_t *buffer, size_t byte_count);

/* Diagnostics (the reference's displayInternalInformation() text dumps, e.g.
 * IqDataProcessor.cc:860-926, become queryable values). */
typedef struct iqd_stats {
    uint64_t accepts;            /* iqd_accept_* calls */
    uint64_t samples;            /* IQ samples accepted, all channels */
    uint64_t kernel_launches;    /* chain-kernel launches */
    uint64_t state_checks;       /* tile hand-offs verified bit-exact (WBFM de-emphasis) */
    uint64_t state_repairs;      /* tiles (WBFM) / rows (AM, SSB DC removal) re-run from the exact carried state */
    double chain_kernel_ms;      /* HIP-event time of the chain kernels while profiling is on */
    uint64_t chain_kernel_count; /* launches covered by chain_kernel_ms */
    uint64_t segment_repairs;    /* extra in-kernel passes of the segmented de-emphasis (a segment's warm-up had not
                                    reached its neighbour's exact state yet) */
    uint64_t stream_launches;    /* chain launches (any family) that ran as a streaming pipeline */
    uint64_t device_launches;    /* kernel-launch calls queued by iqd_accept_* (all kinds) */
    uint64_t device_copies;      /* memcpy / memset operations queued by iqd_accept_* */
    uint64_t mixed_launches;     /* calls whose families' streaming pipelines ran as ranges of one launch */
} iqd_stats;

int iqd_get_stats(iqd_t *e, iqd_stats *out);
int iqd_set_profiling(iqd_t *e, int enabled);  /* HIP events around the chain kernels */
int iqd_get_channel_mode(iqd_t *e, uint32_t ch, int *mode);
int iqd_get_channel_gain(iqd_t *e, uint32_t ch, int demod, float *gain);
const char *iqd_last_error(iqd_t *e);
const char *iqd_strerror(int status);
uint32_t iqd_abi_version(void);

/* The reference's stand-alone resampler classes (not used by any demodulator), block-wise and for
 * n_channels independent streams at once.  A resampler replaces, per channel, one object of
 *   IQD_RESAMPLE_DECIMATE_F32     Decimator(filterLength, coefficients, decimationFactor), Filters/Decimator.cc:
 *                                 one output per `factor` inputs, when the last of them arrives (:283-321);
 *   IQD_RESAMPLE_INTERPOLATE_F32  Interpolator(...), Filters/Interpolator.cc: `factor` outputs per input (:340-364);
 *   IQD_RESAMPLE_INTERPOLATE_I16  Interpolator_int16(...), Filters/Int16/Interpolator_int16.cc: the same in Q15
 *                                 with the per-MAC clamp (:203-246); int16 in and out,
 * and a call with n_in samples per channel replaces n_in decimate()/interpolate() calls; the filter state (and the
 * decimator's commutator phase) carries over to the next call.  Layouts: in [n_channels][n_in], out
 * [n_channels][iqd_resampler_out_count(r, n_in)].  iqd_resampler_reset = resetFilterState(). */
enum iqd_resample_kind { IQD_RESAMPLE_DECIMATE_F32 = 0, IQD_RESAMPLE_INTERPOLATE_F32 = 1, IQD_RESAMPLE_INTERPOLATE_I16 = 2 };
typedef struct iqd_resampler iqd_resampler_t;
int iqd_resampler_create(iqd_t *e, int kind, const float *taps, uint32_t n_taps, uint32_t factor, uint32_t n_channels,
                         iqd_resampler_t **out);
void iqd_resampler_destroy(iqd_resampler_t *r);
int iqd_resampler_reset(iqd_resampler_t *r);
size_t iqd_resampler_out_count(const iqd_resampler_t *r, size_t n_in);
int iqd_resampler_run(iqd_resampler_t *r, const void *in, size_t n_in, void *out);                 /* host pointers */
int iqd_resampler_run_device(iqd_resampler_t *r, const void *in_dev, size_t n_in, void *out_dev);  /* queued on the engine's stream */

/* PCM of several engines (one per GPU, one host process each or all in one) into one place over RCCL / xGMI.  The data
 * path itself has no collective - channels are independent, every GPU demodulates its own - this only
 * delivers the audio, 1/32 of the input volume, to one rank.  The reference has no counterpart (one dongle, one
 * process); the sink it stands in front of is radioApp.cc:103-111.
 *   iqd_gather_unique_id   the root makes an id; the host passes it to the other ranks by any means of its own
 *   iqd_gather_create      every rank, with its own engine (its GPU current), the id, its rank, the world size
 *   iqd_gather_pcm         every rank: its bytes_per_rank[rank] bytes at send_dev go to row `rank` of recv_dev on the
 *                          root (rows row_stride bytes apart; recv_dev is ignored elsewhere).  Queued on the engine's
 *                          stream behind the accept that produced the PCM; the next accept queues behind it.
 * librccl.so is loaded when the first of these is called.  IQD_ENODEV: no RCCL on this host. */
#define IQD_GATHER_ID_BYTES 128
typedef struct iqd_gather iqd_gather_t;
int iqd_gather_unique_id(uint8_t *id128);
int iqd_gather_create(iqd_t *e, const uint8_t *id128, uint32_t rank, uint32_t world, uint32_t root, iqd_gather_t **out);
int iqd_gather_pcm(iqd_gather_t *g, const void *send_dev, const size_t *bytes_per_rank, void *recv_dev, size_t row_stride);
void iqd_gather_destroy(iqd_gather_t *g);
/* RCCL's version code, the rank count the communicator reports (ncclCommCount; g may be NULL: 0), and whether an RCCL the
 * process had loaded already (a torch host's) was reused instead of opening a second copy.  Any pointer may be NULL. */
int iqd_gather_info(iqd_gather_t *g, int *rccl_version, int *comm_ranks, int *library_reused);

/* Small device-memory helpers so that non-HIP hosts (ctypes, cgo, JNI) can stage
 * device-resident buffers for iqd_accept_iq_device without linking the HIP runtime. */
int iqd_dev_alloc(iqd_t *e, size_t bytes, void **out);
int iqd_dev_free(iqd_t *e, void *p);
int iqd_dev_upload(iqd_t *e, void *dst_dev, const void *src_host, size_t bytes);
int iqd_dev_download(iqd_t *e, void *dst_host, const void *src_dev, size_t bytes);
/* Page-locked host memory for iqd_accept_iq: with it the uploads are true DMA and, for calls of 64 MiB and more,
 * overlap the kernels (the call is cut into ~32 MiB slices through two device staging sets; this stands in for
 * the reference's ring of receive buffers, DataConsumer.cc:220-352).  Pageable buffers work too, more slowly. */
int iqd_host_alloc(iqd_t *e, size_t bytes, void **out);
int iqd_host_free(iqd_t *e, void *p);
/* Fills dst_dev with `total` bytes by repeating the first `period` bytes already there. */
int iqd_dev_tile(iqd_t *e, void *dst_dev, size_t period, size_t total);
void *iqd_stream(iqd_t *e); /* the engine's hipStream_t */
int iqd_device_count(void); /* HIP devices this process can see (0 without a usable runtime): what iqd_config::device may name */
int iqd_get_device(iqd_t *e, int *device); /* the HIP device ordinal the engine lives on (iqd_config::device resolved) */
/* Diagnostic builds (-DIQD_STAMPS) only: cycle sums per kernel phase; zeros otherwise. */
int iqd_debug_stamps(iqd_t *e, unsigned long long *out16);
int iqd_debug_stamps_ext(iqd_t *e, unsigned long long *out, uint32_t n);   /* the first n <= 32768 of them (IQD_ST_TIMING / IQD_ST_WAITSTAT / IQD_ST_TRACE builds) */

#ifdef __cplusplus
}
#endif
#endif /* IQDEMOD_H */
