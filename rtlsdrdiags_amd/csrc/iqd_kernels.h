// Launch descriptors and host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "iqd_device.h"

namespace iqd {

struct WbfmStart;
struct WbfmRecord;

enum Counter { CNT_TILE_MISMATCH = 0, CNT_TILE_CHECKS = 1, CNT_SEG_REPAIRS = 2, CNT_DC_REDO = 3, CNT_TILE_REPAIRS = 4, CNT_STREAM_MISMATCH = 5, CNT_COUNT = 8 };
constexpr uint32_t MAX_MISMATCH_LIST = 1024;

// One chain launch = the channels of one demodulator family inside one accept call.
struct ChainLaunch {
    const uint8_t *iq;            // [n_ch][ch_stride_bytes] this call's input (device)
    size_t ch_stride_bytes;
    const uint32_t *ch_list;      // channels (call-relative) handled by this launch
    uint32_t n_list;
    uint32_t first_ch;            // engine channel of call-relative channel 0
    uint32_t vlen;                // samples per channel when no squelch can close
    const uint32_t *vlen_gated;   // else: open samples per call-relative channel
    const uint32_t *blk_lists;    //       and [n_ch][n_blocks] open-block indices
    uint32_t n_blocks;
    uint32_t block_samples, block_magic;
    uint32_t tile_len, tiles_per_ch;
    uint8_t *tails;               // [engine ch][FAM_COUNT][TAIL_BYTES]
    const ChanParams *params;     // [engine ch]
    WbfmCarry *wbfm_carry;        // [engine ch]
    DcCarry *dc_carry;            // [engine ch][2] (AM, SSB)
    GainEpoch *epochs;            // [engine ch]: samples since the WBFM / FM gain last changed
    const float *atan_lut;        // 256x256
    const float *fm_lut;          // 283x283
    int16_t *pcm;                 // [n_ch][pcm_stride]
    size_t pcm_stride;
    uint32_t *mag_sums;           // [n_ch][n_blocks]
    WbfmRecord *records;          // [n_list][tiles_per_ch]
    uint32_t *repair_flags;       // [n_list]: set by the verification, cleared by the repair (zero between calls)
    uint32_t *epoch_report;       // WBFM, optional (page-locked host word): set to 1 by the tail update of any channel whose newest gain
                                  // change still lies inside its kept tail after this call (the host ages its mirror from it)
    uint32_t verify_at_end;       // streaming launches: a cold segment's y_in is its state at its own start, to be compared
                                  // with its predecessor's y_end (tile launches: both taken FORCED_BACK earlier, y_out)
    uint32_t *counters;           // [CNT_COUNT]
    int32_t *base8k;              // AM/SSB detector input at 8 kS/s, n_ch_call * pcm_stride ints
    size_t base_stride_ch, base_stride_t;   // its layout: channel-major or time-major
    uint32_t det16;               // round 6: the streaming pipelines leave the detector input as int16 IN THE PCM ROWS (|x| <= 546) and the
                                  // one-wave DC pass runs over them in place - half the bytes of the int32 stream, no second buffer
    uint32_t n_ch_call;           // channels in this accept call
    void *dc_records;             // long AM/SSB rows: [n_list][dc_tiles] DcRecord, then [n_list] redo flags
    uint32_t dc_tiles;            // tiles per channel of the many-wave DC pass (0: not used)
    unsigned long long *stamps;   // diagnostic builds (IQD_STAMPS): [16] phase cycle sums
    // A launch that runs several families' streaming pipelines side by side (iqd_stream_mixed.hip) gives each a range of
    // its workgroups; a kernel of its own leaves both zero (= the whole grid).
    uint32_t wg_first, wg_count;
};
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t chain_wg(const ChainLaunch &a) { return blockIdx.x - a.wg_first; }
__device__ __forceinline__ uint32_t chain_wgs(const ChainLaunch &a) { return a.wg_count ? a.wg_count : gridDim.x; }
#endif

struct SquelchLaunch {
    uint32_t n_ch, first_ch, n_blocks, block_samples;
    const ChanParams *params;
    const uint32_t *mag_sums;     // [n_ch][n_blocks]
    uint32_t *tracker;            // [engine ch] SignalTracker state
    uint32_t *magnitude;          // out, optional
    uint8_t *allowed;             // out, optional
    uint32_t *blk_lists;          // out, optional
    uint32_t *vlen_out;           // out, optional
    uint32_t *pcm_count;          // out, optional
    const AgcConfig *agc_cfg;     // [engine ch]
    AgcState *agc;                // [engine ch]: the IF gain the squelch uses, and the AGC behind it
    uint32_t any_agc;             // some channel of the call has its AGC enabled
    uint32_t *gain_trace;         // out, optional [n_ch][n_blocks]: the IF gain each block's squelch saw
    const ScanConfig *scan_cfg;   // [engine ch]
    ScanState *scan;              // [engine ch]
    unsigned long long *freq_trace;   // out, optional [n_ch][n_blocks]: the tuned frequency after each block
    uint32_t zero_sums_after;     // squelch_block_kernel is the sums' only reader in this call: it clears them behind itself
    uint32_t *closed_any;         // optional: counts the channels of the call that had a block rejected
};

hipError_t upload_consts(const Consts &c, hipStream_t s);
hipError_t launch_wbfm(const ChainLaunch &a, bool gated, bool mag, uint32_t n_blocks, hipStream_t s);
hipError_t launch_fm(const ChainLaunch &a, bool gated, bool mag, uint32_t n_blocks, hipStream_t s);
hipError_t launch_am(const ChainLaunch &a, int family, bool gated, bool mag, uint32_t n_blocks, hipStream_t s);
hipError_t launch_reset(uint8_t *tails, WbfmCarry *wc, DcCarry *dc, uint32_t first_ch, uint32_t n_ch,
                        uint32_t family_mask, hipStream_t s);
hipError_t launch_wbfm_verify(const ChainLaunch &a, hipStream_t s);
struct StreamArgs;
hipError_t init_wbfm_stream_kernels();   // LDS attribute of the streaming kernels on the current device (iqd_create)
hipError_t init_d4_stream_kernels();
hipError_t launch_wbfm_stream(const ChainLaunch &a, const StreamArgs &sa, int rotation, bool mag, bool epochs, uint32_t grid, hipStream_t s);
struct D4Args;
hipError_t launch_d4_stream(const ChainLaunch &a, const D4Args &da, int mode, bool mag, uint32_t grid, hipStream_t s);
hipError_t launch_am_dc(const ChainLaunch &a, int family, hipStream_t s, bool channel_major = false);   // the DC-removal passes behind an AM/SSB chain launch
hipError_t launch_wbfm_stream_fixup(const ChainLaunch &a, const StreamArgs &sa, hipStream_t s);
hipError_t launch_retail(uint8_t *tails, const ChanParams *params, uint32_t n_ch, hipStream_t s);
// kind: 0 float decimator, 1 float interpolator, 2 int16 interpolator
hipError_t launch_resample(int kind, const void *in, void *out, const void *hist, void *hist_next, const void *taps,
                           uint32_t n_ch, uint32_t n_in, uint32_t n_out, uint32_t hist_len, uint32_t n_taps,
                           uint32_t factor, uint32_t first, hipStream_t s);
hipError_t launch_wbfm_repair(const ChainLaunch &a, bool gated, hipStream_t s);
hipError_t launch_front_end(const uint8_t *iq, int8_t *out, const ChanParams *params, uint32_t first_ch, uint32_t n_ch,
                            size_t bytes_per_ch, hipStream_t s);
hipError_t launch_rotate_signed(int8_t *buf_dev, size_t bytes, int rotation, hipStream_t s);
hipError_t launch_agc_apply(const AgcConfig *cfg, AgcState *st, const ScanConfig *scfg, ScanState *sst,
                            ChanParams *params, GainEpoch *epochs, uint32_t n_ch, hipStream_t s);
hipError_t launch_tail_update(const ChainLaunch &a, int family, hipStream_t s);
hipError_t launch_write_word(uint32_t *word, uint32_t value, hipStream_t s);   // one store, in stream order
hipError_t launch_magnitude(const uint8_t *iq, size_t ch_stride_bytes, const uint32_t *ch_list, uint32_t n_list,
                            uint32_t block_samples, uint32_t n_blocks, uint32_t *mag_sums, hipStream_t s);
// tail_of: a one-family call hands its pending tail update to the squelch launch (tail_squelch_kernel)
// tail_dc: the riding family is AM or SSB on its streaming pipeline and its one-wave DC-removal pass rides along too
hipError_t launch_squelch(const SquelchLaunch &q, bool always_open, hipStream_t s, const ChainLaunch *tail_of = nullptr, int tail_family = 0, bool tail_dc = false);
hipError_t launch_tile_fill(uint8_t *dst, size_t period, size_t total, hipStream_t s);

}  // namespace iqd
