// gfx950 kernels of the IQ demodulation engine.  hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "iqd_kernels.h"
#include "iqd_chains.h"
#include "iqd_wbfm.h"
#include "iqd_stream_fix.h"

namespace iqd {

__constant__ Consts g_consts;

hipError_t upload_consts(const Consts &c, hipStream_t s)
{
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_consts), &c, sizeof(Consts), 0, hipMemcpyHostToDevice, s);
}

// SIMT machine for the phase functions (see the host twin in tests/emu).
struct DeviceExec {
    int tid;
    template <class F> __device__ __forceinline__ void all(F f) { f(tid); __syncthreads(); }
    __device__ __forceinline__ bool in_wave0() const { return tid < 64; }
    __device__ __forceinline__ void wave_fence() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    template <class F> __device__ __forceinline__ void wave0(F f) { f(tid); wave_fence(); }
    template <class F> __device__ __forceinline__ bool wave0_all(F f)
    {
        const bool ok = f(tid);
        wave_fence();
        return __all(ok);
    }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    template <class F> __device__ __forceinline__ void others(F f) { if (tid >= 64) f(tid); }
    template <class F> __device__ __forceinline__ void all_nosync(F f) { f(tid); }
    // Rendezvous of waves 1-3 only (wave 0 is elsewhere, in its serial phase): a monotonic arrival counter in
    // LDS.  All waves of a workgroup are resident, so spinning cannot starve the ones being waited for.
    uint32_t sync_epoch = 0;
    __device__ __forceinline__ void others_sync(uint32_t *ctr)
    {
        if (tid < 64) return;
        sync_epoch += 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((tid & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < sync_epoch) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // value held by the lane below (undefined in lane 0 of a wave): v_mov_b32_dpp wave_shr:1
    template <int SLOT> __device__ __forceinline__ uint32_t shr1(int, uint32_t v) const
    {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
    }
    template <class T> struct Local {   // per-thread values that live across phases: registers here
        T v;
        __device__ __forceinline__ T &at(int) { return v; }
    };
    // the single-wave IIR phase is the workgroup's critical path: let it win VALU arbitration
    __device__ __forceinline__ void critical(bool on) const { if (on) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#ifdef IQD_STAMPS   // diagnostic build only: cycles per phase, summed over workgroups
    uint32_t last = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 32-bit: a workgroup lives far less than 2^32 cycles
    __device__ __forceinline__ void stamp(int k)
    {
        const uint32_t now = (uint32_t)__builtin_readcyclecounter() | 1u;
        if (last) acc[k] += now - last;
        last = now;
    }
#else
    __device__ __forceinline__ void stamp(int) const {}
#endif
};

__device__ __forceinline__ void rotation_selectors(int rotation, WbfmTile &t)
{
    if (rotation == 0) {
        t.sel_i = 0x06040200u; t.sel_q = 0x07050301u; t.neg_i = 0u; t.neg_q = 0u;
    } else {
        t.sel_i = 0x07040300u; t.sel_q = 0x06050201u;
        t.neg_i = rotation > 0 ? 0x00ffff00u : 0xffff0000u;
        t.neg_q = rotation > 0 ? 0xffff0000u : 0x00ffff00u;
    }
}

// One tile of one channel: set-up, the chain, the hand-off record.  `start` is the state the tile begins with.
template <bool GATED, bool MAG>
__device__ __forceinline__ void wbfm_run_tile(const ChainLaunch &a, WbfmLds &lds, uint32_t li, uint32_t tile,
                                              uint32_t ch, uint32_t vlen, const WbfmStart &start)
{
    const uint32_t ech = a.first_ch + ch;         // engine channel
    const int64_t v0 = (int64_t)tile * a.tile_len;
    const ChanParams &p = a.params[ech];
    WbfmTile t;
    t.iq_ch = a.iq + (size_t)ch * a.ch_stride_bytes;
    t.tail = a.tails + ((size_t)ech * FAM_COUNT + FAM_WBFM) * TAIL_BYTES;
    t.blk_list = GATED ? a.blk_lists + (size_t)ch * a.n_blocks : nullptr;
    t.block_samples = a.block_samples;
    t.block_magic = a.block_magic;
    t.v0 = v0;
    t.tlen = (int32_t)(((int64_t)vlen - v0) < (int64_t)a.tile_len ? ((int64_t)vlen - v0) : (int64_t)a.tile_len);
    rotation_selectors(p.rotation, t);
    t.k = p.wbfm_k;
    // only a call's first tile reaches back before the call, where earlier gains may apply
    t.epochs = tile == 0 && a.epochs[ech].wbfm.since[0] < (uint32_t)TAIL ? &a.epochs[ech].wbfm : nullptr;
    float kmax;
    epoch_k_range(t.epochs, t.k, kmax, t.k_min);
    t.bounded = (kmax * 3.1730f < 2147483648.0f) ? 1u : 0u;
    t.lut = a.atan_lut;
    t.pcm_row = a.pcm + (size_t)ch * a.pcm_stride;
    t.mag_row = MAG ? a.mag_sums + (size_t)ch * a.n_blocks : nullptr;
    DeviceExec ex{(int)threadIdx.x};
#ifdef IQD_WBFM_SERIAL_PHASES   // the first driver: IIR phase not overlapped (kept for A/B measurements)
    wbfm_tile<GATED, MAG>(ex, t, g_consts, lds, start, &a.records[(size_t)li * a.tiles_per_ch + tile]);
#else
    wbfm_tile_pipe<GATED, MAG>(ex, t, g_consts, lds, start, &a.records[(size_t)li * a.tiles_per_ch + tile]);
#endif
    if (ex.tid == 0 && lds.repair_count) atomicAdd(&a.counters[CNT_SEG_REPAIRS], lds.repair_count);
#ifdef IQD_STAMPS
    if (a.stamps && (ex.tid == 0 || ex.tid == 64))
        for (int k = 0; k < 8; k++) atomicAdd(&a.stamps[(ex.tid ? 8 : 0) + k], (unsigned long long)ex.acc[k]);
#endif
}

#ifndef IQD_WBFM_MIN_WAVES
#define IQD_WBFM_MIN_WAVES 1
#endif
template <bool GATED, bool MAG>
__global__ __launch_bounds__(WB_THREADS, IQD_WBFM_MIN_WAVES) void wbfm_chain_kernel(const ChainLaunch a)
{
    __shared__ WbfmLds lds;
    const uint32_t li = blockIdx.x / a.tiles_per_ch, tile = blockIdx.x - li * a.tiles_per_ch;
    const uint32_t ch = a.ch_list[li];            // channel index inside this call
    const uint32_t vlen = GATED ? a.vlen_gated[ch] : a.vlen;
    if ((int64_t)tile * a.tile_len >= (int64_t)vlen) return;
    WbfmStart start;
    if (tile == 0) {
        const WbfmCarry cy = a.wbfm_carry[a.first_ch + ch];
        start.y = cy.y; start.u = cy.u; start.back = cy.back; start.cold = 0;
    } else {
        start.y = 0.f; start.u = 0.f; start.back = 0; start.cold = 1;
    }
    wbfm_run_tile<GATED, MAG>(a, lds, li, tile, ch, vlen, start);
}

// Hand-off repair, on the device.  wbfm_verify_kernel flags the channels in which some cold tile's state at its
// restart point differs from what the tile before recorded there (seen only on strictly periodic input, where
// two trajectories can stay one ulp apart for ever).  One workgroup per flagged channel walks its tiles in
// order and re-runs every tile that does not chain up from its predecessor's recorded exact state, which makes
// that tile's own record exact for the next comparison.  Magnitudes are not touched (the first run added them).
__device__ __forceinline__ void tail_update_body(const ChainLaunch &a, int family, uint32_t li);
static_assert(WB_THREADS == 256, "the repair kernel ends with the tail update, which moves 16 bytes per thread");

// (The channel's state commit and new tail follow in the same launch: they need the repaired records, and a launch of
// their own cost more than both together.)
template <bool GATED>
__device__ __forceinline__ void wbfm_repair_body(const ChainLaunch &a, WbfmLds &lds, const uint32_t li)
{
    const uint32_t ch = a.ch_list[li];
    const uint32_t vlen = GATED ? a.vlen_gated[ch] : a.vlen;
    const uint32_t ntiles = (vlen + a.tile_len - 1) / a.tile_len;
    // (squelch-gated streaming launches: the host cannot know how many hand-offs the fix-up kernel verified)
    if (a.verify_at_end == 2 && threadIdx.x == 0 && ntiles > 1) atomicAdd(&a.counters[CNT_TILE_CHECKS], ntiles - 1);
    if (!a.repair_flags[li]) {
        tail_update_body(a, FAM_WBFM, li);
        return;
    }
    WbfmRecord *r = a.records + (size_t)li * a.tiles_per_ch;
    const bool strong = a.params[a.first_ch + ch].wbfm_k >= 1.0f;
    // Which hand-offs do not chain up?  All of them are looked at in parallel first (a long row has tens of thousands
    // of tiles; walking their records one dependent load after the other cost tens of milliseconds for one bad tile);
    // then the bad ones are re-run in order - a re-run makes that tile's own record exact, so the hand-off behind it is
    // looked at again when its turn comes.
    __shared__ uint32_t first_bad;
    uint32_t from = 1;
    for (;;) {
        if (threadIdx.x == 0) first_bad = ntiles;
        __syncthreads();
        for (uint32_t tile = from + threadIdx.x; tile < ntiles; tile += blockDim.x)
            if (!iir_states_agree(r[tile].y_in, a.verify_at_end ? r[tile - 1].y_end : r[tile - 1].y_out, strong)) {
                atomicMin(&first_bad, tile);
                break;                                 // (this thread's later tiles lie behind a bad one anyway)
            }
        __syncthreads();
        uint32_t tile = first_bad;
        __syncthreads();
        if (tile >= ntiles) break;
        // re-run `tile` from its predecessor's recorded exact state; streaming launches: the tile after a re-run one took
        // its first PCM samples from histories that the failed run may have left inexact - it is re-run too
        bool redo_next = false;
        for (; tile < ntiles; tile++) {
            const WbfmRecord prev = r[tile - 1];
            const bool ok = iir_states_agree(r[tile].y_in, a.verify_at_end ? prev.y_end : prev.y_out, strong);
            if (ok && !redo_next) break;
            redo_next = a.verify_at_end && !ok;
            // (a re-run rewrites the tile's record: if it was the one that kept the channel's restart state for a short last
            //  segment - WbfmRecord::pad - the last one is re-run as well, and its own record, exact then, is what the commit takes)
            if (a.verify_at_end && tile + 2 == ntiles && vlen - (ntiles - 1) * a.tile_len < (uint32_t)FORCED_BACK) redo_next = true;
            WbfmStart start;
            start.y = prev.y_out; start.u = prev.u_out; start.back = prev.back_out; start.cold = 0;
            wbfm_run_tile<GATED, false>(a, lds, li, tile, ch, vlen, start);
            __threadfence();
            __syncthreads();     // the new record is written by one lane; everybody reads it next
            if (threadIdx.x == 0) atomicAdd(&a.counters[CNT_TILE_REPAIRS], 1u);
        }
        from = tile;                                   // everything before `tile` chains up now
        if (from >= ntiles) break;
    }
    if (threadIdx.x == 0) a.repair_flags[li] = 0;
    tail_update_body(a, FAM_WBFM, li);
}

template <bool GATED>
__global__ __launch_bounds__(WB_THREADS, IQD_WBFM_MIN_WAVES) void wbfm_repair_kernel(const ChainLaunch a)
{
    __shared__ WbfmLds lds;
    wbfm_repair_body<GATED>(a, lds, blockIdx.x);
}

// Common tile set-up of the FM / AM / SSB kernels (no carried float state: FIR chains only).
__device__ __forceinline__ bool setup_tile(const ChainLaunch &a, bool gated, int family, Tile &t, uint32_t &ch, uint32_t &ech)
{
    const uint32_t li = blockIdx.x / a.tiles_per_ch, tile = blockIdx.x - li * a.tiles_per_ch;
    ch = a.ch_list[li];
    ech = a.first_ch + ch;
    const uint32_t vlen = gated ? a.vlen_gated[ch] : a.vlen;
    const int64_t v0 = (int64_t)tile * a.tile_len;
    if (v0 >= (int64_t)vlen) return false;
    const ChanParams &p = a.params[ech];
    t.iq_ch = a.iq + (size_t)ch * a.ch_stride_bytes;
    t.tail = a.tails + ((size_t)ech * FAM_COUNT + family) * TAIL_BYTES;
    t.blk_list = gated ? a.blk_lists + (size_t)ch * a.n_blocks : nullptr;
    t.block_samples = a.block_samples;
    t.block_magic = a.block_magic;
    t.v0 = v0;
    t.tlen = (int32_t)(((int64_t)vlen - v0) < (int64_t)a.tile_len ? ((int64_t)vlen - v0) : (int64_t)a.tile_len);
    rotation_selectors(p.rotation, t);
    t.k = p.fm_k;
    t.epochs = family == FAM_FM && t.v0 == 0 && a.epochs[ech].fm.since[0] < (uint32_t)TAIL ? &a.epochs[ech].fm : nullptr;
    float kmax;
    epoch_k_range(t.epochs, t.k, kmax, t.k_min);
    t.bounded = (kmax * 6.35f < 2147483648.0f) ? 1u : 0u;   // |K * dtheta| <= |K| * 2 pi
    t.lut = nullptr;
    t.pcm_row = a.pcm + (size_t)ch * a.pcm_stride;
    t.mag_row = a.mag_sums + (size_t)ch * a.n_blocks;
    return true;
}

template <bool GATED, bool MAG>
__global__ __launch_bounds__(WB_THREADS) void fm_chain_kernel(const ChainLaunch a)
{
    __shared__ FmLds lds;
    Tile t;
    uint32_t ch, ech;
    if (!setup_tile(a, GATED, FAM_FM, t, ch, ech)) return;
    DeviceExec ex{(int)threadIdx.x};
    fm_tile<GATED, MAG>(ex, t, g_consts, lds, a.fm_lut);
}

template <bool GATED, bool MAG>
__global__ __launch_bounds__(WB_THREADS) void am_chain_kernel(const ChainLaunch a, int family)
{
    __shared__ AmLds lds;
    Tile t;
    uint32_t ch, ech;
    if (!setup_tile(a, GATED, family, t, ch, ech)) return;
    DeviceExec ex{(int)threadIdx.x};
    am_tile<GATED, MAG>(ex, t, g_consts, lds, family == FAM_SSB, a.params[ech].ssb_lsb,
                        a.base8k + (size_t)ch * a.base_stride_ch, (int)a.base_stride_t);
}

// AM / SSB DC-removal IIR with the exact carried state for batches of short streams: one lane per
// channel.  The detector input was written time-major ([t][channel]) by am_chain_kernel, so a wave
// reads one coalesced row per step; the next 8 rows are fetched while the current 8 are filtered.
// Each lane writes its PCM 4 samples (8 bytes) at a time straight into its channel's row.
__global__ __launch_bounds__(64) void dc_kernel(const ChainLaunch a, int family)
{
    const uint32_t li = blockIdx.x * 64 + threadIdx.x;
    if (li >= a.n_list) return;
    const uint32_t which = family == FAM_SSB ? 1 : 0;
    const uint32_t ch = a.ch_list[li];
    const uint32_t ech = a.first_ch + ch;
    const uint32_t n = (a.vlen_gated ? a.vlen_gated[ch] : a.vlen) / 32;   // any count: a short block is a multiple of 64 bytes
    const float gain = a.params[ech].gain[family];
    DcCarry st = a.dc_carry[2 * (size_t)ech + which];
    const float a1 = g_consts.dc_a1;
    const int32_t *src = a.base8k + ch;                 // + t * base_stride_t
    int16_t *row = a.pcm + (size_t)ch * a.pcm_stride;
    u32x2 *dst = (u32x2 *)row;
    // four samples per store where every row starts on an 8-byte boundary; else - rows of a PCM count that is not a multiple
    // of 4, i.e. short blocks of 64 / 128 / 192 bytes more than a multiple of 256 - sample by sample, and never past the row's
    // end (round 4: the wide store of a 26-sample row's tail used to land on the next channel's first two samples)
    const bool wide = (a.pcm_stride & 3) == 0 && ((uintptr_t)a.pcm & 7) == 0;
    const size_t last = a.pcm_stride - 1;
    constexpr int B = 32;                               // steps per batch; one batch of row loads in flight
    int32_t cur[B], nxt[B];
#pragma unroll
    for (int k = 0; k < B; k++) cur[k] = src[(size_t)((size_t)k < last ? k : last) * a.base_stride_t];
    for (uint32_t t0 = 0; t0 < n; t0 += B) {
#pragma unroll
        for (int k = 0; k < B; k++) {                   // rows past the end are clamped and unused
            const size_t t = (size_t)t0 + B + k;
            nxt[k] = src[(t < last ? t : last) * a.base_stride_t];
        }
#pragma unroll
        for (int q = 0; q < B; q += 4) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float xf = (float)cur[q + k];
                const float tn = xf - st.x_prev;
                const float r = a1 * st.y_prev;
                const float y = tn - r;
                w[k] = (uint32_t)cast_i16(gain * y);
                if (t0 + q + k < n) { st.x_prev = xf; st.y_prev = y; }
            }
            if (wide && t0 + q + 4 <= n) {
                dst[(t0 + q) / 4] = u32x2{pack_lo16(w[0], w[1]), pack_lo16(w[2], w[3])};
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (t0 + q + k < n) row[t0 + q + k] = (int16_t)w[k];
            }
        }
#pragma unroll
        for (int k = 0; k < B; k++) cur[k] = nxt[k];
    }
    a.dc_carry[2 * (size_t)ech + which] = st;
}

// The one-wave DC pass of channel li (the workgroup's first wave): over the int32 detector stream, or - behind a streaming
// pipeline (ChainLaunch::det16) - over the int16 detector values the pipeline left in the PCM row, in place.
__device__ __forceinline__ void dc_channel_wave(const ChainLaunch &a, int family, uint32_t li, DcLds &lds)
{
    const uint32_t ch = a.ch_list[li], ech = a.first_ch + ch;
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[ch] : a.vlen;
    const ChanParams &p = a.params[ech];
    DcCarry st = a.dc_carry[2 * (size_t)ech + (family == FAM_SSB ? 1 : 0)];
    DeviceExec ex{(int)threadIdx.x};
    int16_t *row = a.pcm + (size_t)ch * a.pcm_stride;
    if (a.det16) dc_block_wave(ex, g_consts, lds, (const int16_t *)row, (int)(vlen / 32), p.gain[family], st, row);
    else dc_block_wave(ex, g_consts, lds, a.base8k + (size_t)ch * a.base_stride_ch, (int)(vlen / 32), p.gain[family], st, row);
    if (threadIdx.x == 0) a.dc_carry[2 * (size_t)ech + (family == FAM_SSB ? 1 : 0)] = st;
}

// The same for long streams: one wave per channel, segmented with exact verification.
__global__ __launch_bounds__(64) void dc_wave_kernel(const ChainLaunch a, int family)
{
    __shared__ DcLds lds;
    dc_channel_wave(a, family, blockIdx.x, lds);
}

// Long rows: many waves per channel (dc_tile), the chain-up check, and the one-wave pass again for the
// channels that did not chain up (never seen on live signals; a noiseless decaying tail provokes it).
__global__ __launch_bounds__(64) void dc_tiled_kernel(const ChainLaunch a, int family)
{
    __shared__ DcLds lds;
    const uint32_t li = blockIdx.y, tile = blockIdx.x;
    const uint32_t ch = a.ch_list[li], ech = a.first_ch + ch;
    const uint32_t n = (a.vlen_gated ? a.vlen_gated[ch] : a.vlen) / 32;
    if ((size_t)tile * DC_TILE >= n) return;
    const ChanParams &p = a.params[ech];
    const DcCarry carried = a.dc_carry[2 * (size_t)ech + (family == FAM_SSB ? 1 : 0)];
    DeviceExec ex{(int)threadIdx.x};
    DcRecord rec;
    dc_tile(ex, g_consts, lds, a.base8k + (size_t)ch * a.base_stride_ch, (int)n, (int)tile, p.gain[family], carried,
            a.pcm + (size_t)ch * a.pcm_stride, rec);
    if (threadIdx.x == 0) ((DcRecord *)a.dc_records)[(size_t)li * a.dc_tiles + tile] = rec;
}

// one thread per (channel, tile boundary): flags the channel when the boundary does not chain up
__global__ void dc_chainup_kernel(const ChainLaunch a, int family)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t li = idx / a.dc_tiles, tile = idx - li * a.dc_tiles;
    if (li >= a.n_list || tile == 0) return;
    const uint32_t ch = a.ch_list[li], ech = a.first_ch + ch;
    const uint32_t n = (a.vlen_gated ? a.vlen_gated[ch] : a.vlen) / 32;
    if ((size_t)tile * DC_TILE >= n) return;
    const DcRecord *rec = (const DcRecord *)a.dc_records + (size_t)li * a.dc_tiles;
    if (!iir_states_agree(rec[tile].y_start, rec[tile - 1].y_end, fabsf(a.params[ech].gain[family]) <= 1e6f)) {
        uint32_t *redo = (uint32_t *)((DcRecord *)a.dc_records + (size_t)a.n_list * a.dc_tiles);
        redo[li] = 1;
    }
}

// per channel: commit the last tile's state - or, if some boundary was flagged, redo the row with the one-wave pass
__global__ __launch_bounds__(64) void dc_redo_kernel(const ChainLaunch a, int family)
{
    __shared__ DcLds lds;
    const uint32_t li = blockIdx.x;
    uint32_t *redo = (uint32_t *)((DcRecord *)a.dc_records + (size_t)a.n_list * a.dc_tiles);
    const uint32_t ch = a.ch_list[li], ech = a.first_ch + ch;
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[ch] : a.vlen;
    const uint32_t n = vlen / 32;
    DcCarry *carry = &a.dc_carry[2 * (size_t)ech + (family == FAM_SSB ? 1 : 0)];
    if (!redo[li]) {
        if (threadIdx.x == 0 && n) {
            const DcRecord last = ((const DcRecord *)a.dc_records)[(size_t)li * a.dc_tiles + (n + DC_TILE - 1) / DC_TILE - 1];
            *carry = DcCarry{last.x_end, last.y_end};
        }
        return;
    }
    const ChanParams &p = a.params[ech];
    DcCarry st = *carry;
    DeviceExec ex{(int)threadIdx.x};
    dc_block_wave(ex, g_consts, lds, a.base8k + (size_t)ch * a.base_stride_ch, (int)n, p.gain[family], st,
                  a.pcm + (size_t)ch * a.pcm_stride);
    if (threadIdx.x == 0) {
        *carry = st;
        redo[li] = 0;   // zero between calls
        atomicAdd(&a.counters[CNT_DC_REDO], 1u);
    }
}

// Hand-off check between consecutive tiles of a channel: a cold tile's own state at its
// restart point must equal, bit for bit, what the tile before it recorded there.
__global__ void wbfm_verify_kernel(const ChainLaunch a)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t li = idx / a.tiles_per_ch, tile = idx % a.tiles_per_ch;
    if (li >= a.n_list) return;
    const uint32_t ch = a.ch_list[li];
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[ch] : a.vlen;
    const uint32_t ntiles = (vlen + a.tile_len - 1) / a.tile_len;
    if (tile >= ntiles) return;
    const WbfmRecord *r = a.records + (size_t)li * a.tiles_per_ch;
    if (tile > 0) {
        if (!iir_states_agree(r[tile].y_in, a.verify_at_end ? r[tile - 1].y_end : r[tile - 1].y_out, a.params[a.first_ch + ch].wbfm_k >= 1.0f)) {
            atomicAdd(&a.counters[CNT_TILE_MISMATCH], 1u);
            a.repair_flags[li] = 1;
        } else {
            atomicAdd(&a.counters[CNT_TILE_CHECKS], 1u);
        }
    }
}

// New tail = last TAIL samples of [old tail | this call's open blocks] for one family.  For WBFM the
// same launch commits the restart state of each channel's last tile (after verification and repair).
__device__ __forceinline__ void tail_update_body(const ChainLaunch &a, int family, uint32_t li)
{
    const uint32_t ch = a.ch_list[li];
    const uint32_t ech = a.first_ch + ch;
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[ch] : a.vlen;
    if (family == FAM_WBFM && a.epoch_report && threadIdx.x == 0 &&
        (uint64_t)a.epochs[ech].wbfm.since[0] + vlen < (uint64_t)TAIL)
        *(volatile uint32_t *)a.epoch_report = 1u;   // (every writer stores the same value: no atomic needed)
    if (vlen == 0) return;
    if (family == FAM_WBFM && threadIdx.x == 0) {
        const uint32_t ntiles = (vlen + a.tile_len - 1) / a.tile_len;
        // (a streamed last segment shorter than FORCED_BACK took its own record inside its lead-in, before its state had become
        //  exact: the segment before it kept the state at the restart point - wbfm_pick_carry, iqd_wbfm.h)
        const WbfmRecord r = a.records[(size_t)li * a.tiles_per_ch + ntiles - 1];
        const WbfmRecord p = a.records[(size_t)li * a.tiles_per_ch + (ntiles >= 2 ? ntiles - 2 : 0)];
        a.wbfm_carry[ech] = wbfm_pick_carry(r, p, ntiles, vlen, a.tile_len, a.verify_at_end);
    }
    if (threadIdx.x < EPOCHS && (family == FAM_WBFM || family == FAM_FM)) {   // this call's samples now lie behind the gain changes
        uint32_t *since = (family == FAM_WBFM ? a.epochs[ech].wbfm.since : a.epochs[ech].fm.since) + threadIdx.x;
        const uint64_t total = (uint64_t)*since + vlen;
        *since = total < (uint64_t)TAIL ? (uint32_t)total : (uint32_t)TAIL;
    }
    uint8_t *tail = a.tails + ((size_t)ech * FAM_COUNT + family) * TAIL_BYTES;
    const uint8_t *iq_ch = a.iq + (size_t)ch * a.ch_stride_bytes;
    const uint32_t *blk_list = a.vlen_gated ? a.blk_lists + (size_t)ch * a.n_blocks : nullptr;
    // thread i moves 8 samples (16 bytes) of the last `keep` the family can reach back for (tail_keep, iqd_device.h): kept sample
    // 8i.. = virtual vlen-keep+8i, at the END of the channel's tail buffer (readers count back from there)
    const int keep = family == FAM_WBFM ? tail_keep(FAM_WBFM) : family == FAM_FM ? tail_keep(FAM_FM) : family == FAM_AM ? tail_keep(FAM_AM) : tail_keep(FAM_SSB);
    const bool mine = 8 * (int)threadIdx.x < keep;
    const int64_t v = (int64_t)vlen - keep + 8 * (int64_t)(mine ? threadIdx.x : 0);
    const uint8_t *src;
    if (v < 0) src = tail + TAIL_BYTES + 2 * v;
    else if (!blk_list) src = iq_ch + 2 * v;
    else {
        const uint32_t blk = (uint32_t)(v / a.block_samples);
        const uint32_t off = (uint32_t)(v - (int64_t)blk * a.block_samples);
        src = iq_ch + ((int64_t)blk_list[blk] * a.block_samples + off) * 2;
    }
    uint4 val{};
    if (mine) val = *(const uint4 *)src;
    __syncthreads();
#ifndef IQD_NO_WT_STORES   // write-through, like the DC pass's PCM (iqd_chains.h: dc_store)
    if (mine) {
        typedef uint32_t wt4 __attribute__((ext_vector_type(4)));
        const wt4 vv = {val.x, val.y, val.z, val.w};
        const void *dst = tail + TAIL_BYTES - 2 * keep + 16 * (int)threadIdx.x;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(vv) : "memory");
    }
#else
    if (mine) ((uint4 *)(tail + TAIL_BYTES - 2 * keep))[threadIdx.x] = val;
#endif
}

__global__ __launch_bounds__(256) void tail_update_kernel(const ChainLaunch a, int family)
{
    tail_update_body(a, family, blockIdx.x);
}

// Per-block squelch magnitude sums for channels whose chain kernel does not produce them
// (mode None, squelch-gated calls).  One workgroup per (block, channel).
__global__ __launch_bounds__(256) void magnitude_kernel(const uint8_t *iq, size_t ch_stride_bytes,
                                                        const uint32_t *ch_list, uint32_t block_samples,
                                                        uint32_t n_blocks, uint32_t *mag_sums)
{
    const uint32_t ch = ch_list ? ch_list[blockIdx.y] : blockIdx.y;
    const uint32_t blk = blockIdx.x;
    const uint4 *src = (const uint4 *)(iq + (size_t)ch * ch_stride_bytes + (size_t)blk * block_samples * 2);
    const uint32_t n16 = block_samples / 8;  // 16-byte groups
    uint32_t m = 0;
    for (uint32_t i = threadIdx.x; i < n16; i += 256) {
        uint4 r = src[i];
        m += magnitude2(r.x ^ 0x80808080u) + magnitude2(r.y ^ 0x80808080u) +
             magnitude2(r.z ^ 0x80808080u) + magnitude2(r.w ^ 0x80808080u);
    }
    for (int off = 32; off > 0; off >>= 1) m += __shfl_down(m, off);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) mag_sums[(size_t)ch * n_blocks + blk] = part[0] + part[1] + part[2] + part[3];
}

// Squelch, part 1 (SignalDetector.cc:249-271): per (channel, block) average magnitude.  When no channel's
// squelch can close (always_open) this is the whole squelch: every block is allowed and the tracker ends in its
// Tracking state.  Otherwise the decisions are taken per channel in block order by squelch_track_kernel.
__device__ __forceinline__ void squelch_block_body(const SquelchLaunch &q, int always_open, uint32_t idx)
{
    if (idx >= q.n_ch * q.n_blocks) return;
    const uint32_t ch = idx / q.n_blocks, b = idx - ch * q.n_blocks;
    const uint32_t ech = q.first_ch + ch;
    const ChanParams &p = q.params[ech];
    if (q.magnitude) q.magnitude[idx] = q.mag_sums[idx] / q.block_samples;
    // (the last reader of the sums when nothing tracks them block by block: leaves them zero for the next call's
    // chain kernels to add into, which saves that call a memset launch)
    if (q.zero_sums_after) const_cast<uint32_t *>(q.mag_sums)[idx] = 0;
    if (q.gain_trace) q.gain_trace[idx] = q.agc[ech].rx_gain;   // channels with a running AGC overwrite theirs
    if (q.freq_trace) q.freq_trace[idx] = q.scan[ech].current_hz;   // scanning, gated channels overwrite theirs
    if (always_open) {
        if (q.allowed) q.allowed[idx] = 1;
        if (b == q.n_blocks - 1) q.tracker[ech] = 1u;
        if (b == 0 && q.pcm_count) q.pcm_count[ch] = p.mode == 0 ? 0u : q.n_blocks * q.block_samples / 32u;
    }
}

__global__ void squelch_block_kernel(const SquelchLaunch q, int always_open)
{
    squelch_block_body(q, always_open, blockIdx.x * blockDim.x + threadIdx.x);
}

// A call with one demodulator family: the family's tail update (workgroups 0 .. n_list-1) and the squelch pass's first
// part (the rest) share a launch - they are independent, tiny, and each launch of their own costs about as much as both.
__global__ __launch_bounds__(256) void tail_squelch_kernel(const ChainLaunch a, int family, const SquelchLaunch q, int always_open)
{
    if (blockIdx.x < a.n_list) tail_update_body(a, family, blockIdx.x);
    else squelch_block_body(q, always_open, (blockIdx.x - a.n_list) * 256u + threadIdx.x);
}

// ... and for a call whose one family is AM or SSB on its streaming pipeline (round 5): the DC-removal pass of the channels - the
// one-wave pass of dc_wave_kernel, which used to be a launch of its own between the pipeline and this one (0.019 ms + a queue gap
// of the 0.2 ms step) - is a third role: workgroups 0 .. n_list - 1, their first wave.  It depends on the pipeline only.
__global__ __launch_bounds__(256) void tail_dc_squelch_kernel(const ChainLaunch a, int family, const SquelchLaunch q, int always_open)
{
    __shared__ DcLds lds;
    // (round 6: the DC passes - one serial recurrence per row, the launch's long pole - take the FIRST workgroups, so that they are
    //  dispatched first and the tail updates fill in beside them: AM 4096 x 2^16 -2 %, 4096 x 2^14 -5 %, profiles/r6_dc_first_ab.txt)
    if (blockIdx.x < a.n_list) {
        if (threadIdx.x >= 64) return;
        dc_channel_wave(a, family, blockIdx.x, lds);
    } else if (blockIdx.x < 2 * a.n_list) {
        tail_update_body(a, family, blockIdx.x - a.n_list);
    } else {
        squelch_block_body(q, always_open, (blockIdx.x - 2 * a.n_list) * 256u + threadIdx.x);
    }
}

// The same for a call whose one family is WBFM: hand-off repair check, state commit and tail (workgroups 0 .. n_list-1)
// beside the squelch pass's first part.
__global__ __launch_bounds__(WB_THREADS, IQD_WBFM_MIN_WAVES) void wbfm_repair_squelch_kernel(const ChainLaunch a, const SquelchLaunch q, int always_open)
{
    __shared__ WbfmLds lds;
    if (blockIdx.x < a.n_list) wbfm_repair_body<false>(a, lds, blockIdx.x);
    else squelch_block_body(q, always_open, (blockIdx.x - a.n_list) * 256u + threadIdx.x);
}

// What follows a launch of several families' streaming pipelines (iqd_stream_mixed.hip), again as one launch: every
// workgroup has one role - WBFM boundary fix-up (8 segments), DC removal of one AM or SSB channel (its first wave; the
// one-wave pass of dc_wave_kernel), tail update of one AM / FM / SSB channel.  All of them depend on the stream launch only;
// the WBFM repair check and state commit, which need the fix-up's verdicts, ride in the squelch launch behind this one.
struct MixedTailRoles { uint32_t end[6]; };
// (each launch descriptor a kernel argument of its own: see mixed_stream_kernel)
__global__ __launch_bounds__(256) void mixed_tail_kernel(const ChainLaunch a_wbfm, const StreamArgs sa, const ChainLaunch a_am,
                                                         const ChainLaunch a_fm, const ChainLaunch a_ssb, const MixedTailRoles roles)
{
    __shared__ union Lds { DcLds dc; FixLds fix; __device__ Lds() {} } u;   // (a workgroup has one role)
    DcLds &lds = u.dc;
    const uint32_t b = blockIdx.x;
    if (b < roles.end[0]) {
        wbfm_stream_fixup_body(a_wbfm, sa, b, u.fix);
    } else if (b < roles.end[2]) {
        if (threadIdx.x >= 64) return;
        const int family = b < roles.end[1] ? FAM_AM : FAM_SSB;
        const ChainLaunch &a = family == FAM_AM ? a_am : a_ssb;
        const uint32_t li = b - (family == FAM_AM ? roles.end[0] : roles.end[1]);
        dc_channel_wave(a, family, li, lds);
    } else if (b < roles.end[3]) {
        tail_update_body(a_am, FAM_AM, b - roles.end[2]);
    } else if (b < roles.end[4]) {
        tail_update_body(a_fm, FAM_FM, b - roles.end[3]);
    } else if (b < roles.end[5]) {
        tail_update_body(a_ssb, FAM_SSB, b - roles.end[4]);
    }
}

// Squelch, part 2, one thread per channel, blocks in order: the "signal present" comparison
// (SignalDetector.cc:259-266) with the IF gain in force, the two-state tracker with its one-block tail
// (SignalTracker.cc:104-145, Squelch.cc:240-269), the list of open blocks - and the magnitude callback into the
// channel's AGC (IqDataProcessor.cc:781-790), whose gain the NEXT block's comparison sees.
__global__ void squelch_track_kernel(const SquelchLaunch q, int always_open)
{
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= q.n_ch) return;
    const uint32_t ech = q.first_ch + ch;
    const AgcConfig cfg = q.agc_cfg[ech];
    if (always_open && !cfg.enabled) return;   // squelch_block_kernel did everything
    const ChanParams &p = q.params[ech];
    AgcState st = q.agc[ech];
    uint32_t gain = st.rx_gain;
    uint32_t tracking = q.tracker[ech];
    uint32_t open = 0;
    const ScanConfig sc = q.scan_cfg[ech];
    ScanState ss = q.scan[ech];
    for (uint32_t b = 0; b < q.n_blocks; b++) {
        const size_t idx = (size_t)ch * q.n_blocks + b;
        const uint32_t avg = q.mag_sums[idx] / q.block_samples;
        if (!always_open) {
            const uint32_t m = avg > 127u ? 127u : avg;  // DbfsCalculator.cc:122-125, full scale 127
            int32_t dbfs = g_consts.db_table[m] - 42;
            dbfs = (int32_t)((uint32_t)dbfs - gain);
            const uint32_t present = dbfs >= p.threshold ? 1u : 0u;
            const uint32_t allowed = present | tracking;
            tracking = present;
            if (q.allowed) q.allowed[idx] = (uint8_t)allowed;
            if (allowed) {
                q.blk_lists[(size_t)ch * q.n_blocks + open] = b;
                open++;
            } else if (sc.scanning) {
                scanner_step(sc, ss);
            }
            if (q.freq_trace) q.freq_trace[idx] = ss.current_hz;
        }
        if (q.gain_trace) q.gain_trace[idx] = gain;
        if (cfg.enabled) gain = agc_run(g_consts, cfg, st, avg, gain);
    }
    st.rx_gain = gain;
    q.agc[ech] = st;
    if (!always_open) {
        q.tracker[ech] = tracking;
        q.scan[ech] = ss;
        const uint32_t vlen = (p.mode == 0) ? 0u : open * q.block_samples;
        q.vlen_out[ch] = vlen;
        if (q.pcm_count) q.pcm_count[ch] = vlen / 32u;
        if (q.closed_any && p.mode != 0 && open != q.n_blocks) atomicAdd(q.closed_any, 1u);
    }
}

// The same for rows of many blocks: one wave per channel, 64 blocks at a time.  The lanes fetch the 64 averages
// together; with a fixed gain the decisions are independent (allowed = present | present of the block before) and
// the open-block list is a ballot/popcount compaction; with a running AGC the wave steps through the 64 values in
// order from registers (the recurrence stays serial, the memory latency is gone).
#ifndef IQD_AGC_RUNS
#define IQD_AGC_RUNS 1   // 0: every block of a channel with a running AGC takes the serial step (A/B builds)
#endif
__global__ __launch_bounds__(64) void squelch_track_wave_kernel(const SquelchLaunch q, int always_open)
{
    const uint32_t ch = blockIdx.x, lane = threadIdx.x;
    const uint32_t ech = q.first_ch + ch;
    // everything the block loop carries is the same in all lanes: say so (readfirstlane), and the integer part of the
    // per-block state machine runs on the scalar unit
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    AgcConfig cfg = q.agc_cfg[ech];
    cfg.enabled = uni(cfg.enabled); cfg.type = uni(cfg.type); cfg.operating_point = (int32_t)uni((uint32_t)cfg.operating_point);
    cfg.deadband = (int32_t)uni((uint32_t)cfg.deadband); cfg.blanking_limit = uni(cfg.blanking_limit);
    cfg.alpha = u2f(uni(f2u(cfg.alpha)));
    if (always_open && !cfg.enabled) return;
    const ChanParams &p = q.params[ech];
    const int32_t threshold = (int32_t)uni((uint32_t)p.threshold);
    AgcState st = q.agc[ech];
    st.if_gain = uni(st.if_gain); st.blank_ctr = uni(st.blank_ctr); st.adjusted = uni(st.adjusted);
    st.filtered = u2f(uni(f2u(st.filtered)));
    uint32_t gain = uni(st.rx_gain);
    uint32_t tracking = uni(q.tracker[ech]);
    uint32_t open = 0;
    const ScanConfig sc = q.scan_cfg[ech];
    ScanState ss = q.scan[ech];
    const unsigned long long below = (1ull << lane) - 1ull;
    auto sum_at = [&](uint32_t b) { return q.mag_sums[(size_t)ch * q.n_blocks + (b < q.n_blocks ? b : q.n_blocks - 1)]; };
    // two trips to memory per group of 64 blocks (the sums, then the dB table) are asked for one group ahead each, so that a
    // long row's walk does not stand still for them
    uint32_t avg_next = sum_at(lane) / q.block_samples;
    int32_t sig_next = magnitude_dbfs(g_consts, avg_next);
    uint32_t sum_next2 = sum_at(lane + 64);
    for (uint32_t base = 0; base < q.n_blocks; base += 64) {
        const uint32_t b = base + lane;
        const bool valid = b < q.n_blocks;
        const size_t idx = (size_t)ch * q.n_blocks + (valid ? b : q.n_blocks - 1);
        const uint32_t avg = avg_next;
        const int32_t my_sig = sig_next;
        avg_next = sum_next2 / q.block_samples;
        sig_next = magnitude_dbfs(g_consts, avg_next);
        sum_next2 = sum_at(b + 128);
        const uint32_t count = q.n_blocks - base < 64 ? q.n_blocks - base : 64;
        uint32_t my_gain = gain, allowed = 0;
        unsigned long long my_freq = ss.current_hz;
        if (!cfg.enabled && !sc.scanning) {   // nothing moves from block to block but the tracker's one-block tail
            if (!always_open) {
                const int32_t dbfs = (int32_t)((uint32_t)my_sig - gain);
                const uint32_t present = (valid && dbfs >= threshold) ? 1u : 0u;
                uint32_t before = (uint32_t)__shfl_up((int)present, 1);
                if (lane == 0) before = tracking;
                allowed = present | before;
                tracking = (uint32_t)__shfl((int)present, (int)count - 1);
            }
        } else if (IQD_AGC_RUNS && cfg.enabled && !sc.scanning) {
            // A running AGC (round 4).  The recurrence over blocks is serial only where the AGC MOVES: while the error stays
            // inside the deadband (or is pinned at a gain limit) a step changes nothing but the two "last seen" values, and
            // while the measurements behind an adjustment are blanked it only counts.  Runs of such blocks are found with one
            // ballot over the 64 blocks in the lanes and taken at once - the squelch's presence / tracker for them in parallel,
            // the gain being constant - and only the blocks that adjust the gain take the serial step
            // (AutomaticGainControl.cc:663-748; a settled AGC on a 2^28-sample row: 3.7 -> 0.8 ms per step).
            uint32_t j0 = 0;
            while (j0 < count) {
                const bool adjusted = st.adjusted != 0;
                uint32_t run = 0;
                bool blanked = false;
                if (adjusted && st.blank_ctr < cfg.blanking_limit) {     // blanked measurements: nothing but the counter moves
                    run = cfg.blanking_limit - st.blank_ctr;
                    run = run < count - j0 ? run : count - j0;
                    blanked = true;
                } else {
                    // would a step with error 0 leave the filter where it is?  Harris: filtered + alpha * 0; lowpass: only at its
                    // fixed point.  (in range, so that the clamp does nothing; not -0.0, which the addition would turn into +0.0)
                    const float f = st.filtered;
                    const float f1 = cfg.type == 0 ? (cfg.alpha * (float)(int32_t)gain) + ((1 - cfg.alpha) * f) : f + (cfg.alpha * 0.f);
                    const bool steady = f2u(f1) == f2u(f) && f >= 0.f && f <= (float)AGC_MAX_GAIN && f2u(f) != 0x80000000u;
                    if (steady) {
                        int32_t error = cfg.operating_point - my_sig;
                        const bool at_max = gain == AGC_MAX_GAIN, at_min = gain == 0;
                        error = (at_max && error > 0) ? 0 : error;
                        error = (!at_max && at_min && error < 0) ? 0 : error;
                        error = ((error < 0 ? -error : error) <= cfg.deadband) ? 0 : error;
                        const unsigned long long moves = __ballot(error != 0 || lane >= count) >> j0;   // bit k: block j0 + k adjusts (or is not there)
                        run = moves ? (uint32_t)__builtin_ctzll(moves) : 64u - j0;
                    }
                }
                if (run) {
                    const uint32_t j1 = j0 + run;
                    uint32_t al = 1;
                    if (!always_open) {
                        const int32_t dbfs = (int32_t)((uint32_t)my_sig - gain);
                        const uint32_t present = dbfs >= threshold ? 1u : 0u;
                        uint32_t before = (uint32_t)__shfl_up((int)present, 1);
                        if (lane == j0) before = tracking;
                        al = present | before;
                        tracking = (uint32_t)__builtin_amdgcn_readlane((int)present, (int)j1 - 1);
                    }
                    if (lane >= j0 && lane < j1) { allowed = al; my_gain = gain; my_freq = ss.current_hz; }
                    if (blanked) {
                        st.if_gain = gain;
                        st.blank_ctr += run;
                    } else {
                        st.blank_ctr = adjusted ? 0u : st.blank_ctr;
                        st.adjusted = 0;
                        st.signal_magnitude = (uint32_t)__builtin_amdgcn_readlane((int)avg, (int)j1 - 1);
                        st.normalized = (int32_t)((uint32_t)__builtin_amdgcn_readlane(my_sig, (int)j1 - 1) - gain);
                        st.if_gain = (uint32_t)st.filtered;
                    }
                    j0 = j1;
                    continue;
                }
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)avg, (int)j0);
                const int32_t sig = __builtin_amdgcn_readlane(my_sig, (int)j0);
                uint32_t al = 1;
                if (!always_open) {
                    const int32_t dbfs = (int32_t)((uint32_t)sig - gain);
                    const uint32_t present = dbfs >= threshold ? 1u : 0u;
                    al = present | tracking;
                    tracking = present;
                }
                if (lane == j0) { allowed = al; my_gain = gain; my_freq = ss.current_hz; }
                gain = agc_run_dbfs(cfg, st, a, sig, gain);
                j0++;
            }
        } else {
            for (uint32_t j = 0; j < count; j++) {   // wave-uniform state, block j's values from lane j
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)avg, (int)j);
                const int32_t sig = __builtin_amdgcn_readlane(my_sig, (int)j);
                uint32_t al = 1;
                if (!always_open) {
                    const int32_t dbfs = (int32_t)((uint32_t)sig - gain);
                    const uint32_t present = dbfs >= threshold ? 1u : 0u;
                    al = present | tracking;
                    tracking = present;
                    if (!al && sc.scanning) scanner_step(sc, ss);
                }
                if (lane == j) { allowed = al; my_gain = gain; my_freq = ss.current_hz; }
                if (cfg.enabled) gain = agc_run_dbfs(cfg, st, a, sig, gain);
            }
        }
        if (valid) {
            if (q.gain_trace) q.gain_trace[idx] = my_gain;
            if (!always_open) {
                if (q.allowed) q.allowed[idx] = (uint8_t)allowed;
                if (q.freq_trace) q.freq_trace[idx] = my_freq;
            }
        }
        if (!always_open) {
            const unsigned long long ball = __ballot(valid && allowed);
            if (valid && allowed) q.blk_lists[(size_t)ch * q.n_blocks + open + (uint32_t)__popcll(ball & below)] = b;
            open += (uint32_t)__popcll(ball);
        }
    }
    if (lane != 0) return;
    st.rx_gain = gain;
    q.agc[ech] = st;
    if (!always_open) {
        q.tracker[ech] = tracking;
        q.scan[ech] = ss;
        const uint32_t vlen = (p.mode == 0) ? 0u : open * q.block_samples;
        q.vlen_out[ch] = vlen;
        if (q.pcm_count) q.pcm_count[ch] = vlen / 32u;
        if (q.closed_any && p.mode != 0 && open != q.n_blocks) atomicAdd(q.closed_any, 1u);
    }
}

// The operator's one-shot commands, applied before the next block: a manual IF gain
// (Radio::setReceiveIfGainInDb), resetBlankingSystem() (AutomaticGainControl.cc:625-634), the scanner's jump to
// its end frequency when it starts with new parameters.
__global__ void agc_apply_kernel(const AgcConfig *cfg, AgcState *st, const ScanConfig *scfg, ScanState *sst,
                                 ChanParams *params, GainEpoch *epochs, uint32_t n_ch)
{
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_ch) return;
    // a demodulator gain changed since the last accept: the new one starts with the next sample; the one before it is
    // remembered for the histories - unless no sample was consumed since the previous change, whose "before" still holds
    const uint32_t changed = params[ch].k_changed;
    // (a change with no sample consumed since the one before it replaces nothing: that one's "before" still holds)
    auto push = [](GainEpochList &l, float k_before) {
        if (l.since[0] == 0) return;
        for (int i = EPOCHS - 1; i > 0; i--) { l.since[i] = l.since[i - 1]; l.k_before[i] = l.k_before[i - 1]; }
        l.since[0] = 0;
        l.k_before[0] = k_before;
    };
    if (changed & 1u) push(epochs[ch].wbfm, params[ch].wbfm_k_prev);
    if (changed & 2u) push(epochs[ch].fm, params[ch].fm_k_prev);
    if (changed) params[ch].k_changed = 0;
    const AgcConfig c = cfg[ch];
    if (c.set_gain != 0xffffffffu) st[ch].rx_gain = c.set_gain;
    if (c.reset_blanking) { st[ch].blank_ctr = 0; st[ch].adjusted = 0; }
    if (scfg[ch].set_current_flag) {   // FrequencyScanner::start() after new parameters, FrequencyScanner.cc:250-257
        sst[ch].current_hz = scfg[ch].set_current;
        sst[ch].tune_count++;
    }
}

// resetDemodulator() for a channel range: histories become zero signal; the WBFM
// de-emphasis state survives (WbFmDemodulator.cc:304-320) and restarts at the stream end.
__global__ void reset_kernel(uint8_t *tails, WbfmCarry *wc, DcCarry *dc, uint32_t first_ch, uint32_t n_ch,
                             uint32_t family_mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_tail = TAIL_BYTES / 16;
    const size_t n16 = (size_t)n_ch * FAM_COUNT * per_tail;
    uint4 *t = (uint4 *)(tails + (size_t)first_ch * FAM_COUNT * TAIL_BYTES);
    for (size_t k = i; k < n16; k += (size_t)gridDim.x * blockDim.x) {
        const uint32_t fam = (uint32_t)((k / per_tail) % FAM_COUNT);
        if (family_mask & (1u << fam)) t[k] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    }
    if (i < n_ch) {
        if (family_mask & (1u << FAM_WBFM)) {
            WbfmCarry c = wc[first_ch + i];
            c.y = c.y_end; c.u = c.u_end; c.back = 0;
            wc[first_ch + i] = c;
        }
        if (family_mask & (1u << FAM_AM)) dc[2 * (size_t)(first_ch + i)] = DcCarry{0.f, 0.f};
        if (family_mask & (1u << FAM_SSB)) dc[2 * (size_t)(first_ch + i) + 1] = DcCarry{0.f, 0.f};
    }
}

// One group of 4 signed samples (I0 Q0 I1 Q1 | I2 Q2 I3 Q3) through upconvertByFsOver4 (rotation > 0,
// IqDataProcessor.cc:567-611), downconvertByFsOver4 (< 0, :496-540) or neither.
__device__ __forceinline__ u32x2 rotate_group(u32x2 g, int rotation)
{
    if (rotation == 0) return g;
    // sample 1: (-Q, I) up / (Q, -I) down; sample 2: (-I, -Q); sample 3: (Q, -I) up / (-Q, I) down
    const uint32_t a_sw = perm(g.x, g.x, 0x02030100u);   // I0 Q0 Q1 I1
    const uint32_t b_sw = perm(g.y, g.y, 0x02030100u);   // I2 Q2 Q3 I3
    if (rotation > 0) return u32x2{neg_bytes(a_sw, 0x00ff0000u), neg_bytes(b_sw, 0xff00ffffu)};   // I0 Q0 -Q1 I1 | -I2 -Q2 Q3 -I3
    return u32x2{neg_bytes(a_sw, 0xff000000u), neg_bytes(b_sw, 0x00ffffffu)};                      // I0 Q0 Q1 -I1 | -I2 -Q2 -Q3 I3
}

// The front end alone (IqDataProcessor.cc:735-749): u8 -> s8, then the channel's rotation, interleaved bytes
// out - what the reference leaves in the caller's buffer and streams from its IQ dump tap (:756-760).
// One thread per 4 samples (8 bytes); the rotation phase restarts with every block, and blocks are multiples
// of 4 samples, so it is the sample index modulo 4.
// params == nullptr: every row is rotated by `fixed_rotation` and holds SIGNED bytes already (the stand-alone
// up/downconvertByFsOver4 of IqDataProcessor.cc:487-611, which work on the converted buffer).
__global__ void front_end_kernel(const uint8_t *iq, int8_t *out, const ChanParams *params, uint32_t first_ch,
                                 uint32_t n_ch, size_t bytes_per_ch, int fixed_rotation)
{
    const size_t groups_per_ch = bytes_per_ch / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups_per_ch * n_ch) return;
    const uint32_t ch = (uint32_t)(idx / groups_per_ch);
    const int rotation = params ? params[first_ch + ch].rotation : fixed_rotation;
    const uint32_t to_signed = params ? 0x80808080u : 0u;
    const u32x2 raw = ((const u32x2 *)iq)[idx];
    ((u32x2 *)out)[idx] = rotate_group(u32x2{raw.x ^ to_signed, raw.y ^ to_signed}, rotation);
}

// A rotation selector changed between two calls.  The filter histories live on as raw tail bytes that every tile
// re-rotates with the channel's (now new) selector, while the reference's histories hold what the OLD selector
// produced.  Rotations are exactly invertible on int8 (negation is an involution, -128 its fixed point), so the
// tails are rewritten once: tail' = rot(-new)(rot(old)(tail)); read back through the new selector they give the
// old-rotated history bit for bit.  One workgroup per channel, all four families' tails.
__global__ __launch_bounds__(256) void retail_kernel(uint8_t *tails, const ChanParams *params, uint32_t n_ch)
{
    const uint32_t ch = blockIdx.x;
    if (ch >= n_ch || !(params[ch].k_changed & 4u)) return;
    const int old_rot = params[ch].rotation_prev, new_rot = params[ch].rotation;
    u32x2 *t = (u32x2 *)(tails + (size_t)ch * FAM_COUNT * TAIL_BYTES);
    for (uint32_t i = threadIdx.x; i < FAM_COUNT * TAIL_BYTES / 8; i += blockDim.x) {   // groups of 4 samples, phase 0 first
        const u32x2 raw = t[i];
        u32x2 s = rotate_group(u32x2{raw.x ^ 0x80808080u, raw.y ^ 0x80808080u}, old_rot);
        s = rotate_group(s, -new_rot);
        t[i] = u32x2{s.x ^ 0x80808080u, s.y ^ 0x80808080u};
    }
}

// ---- float / int16 resamplers (Filters/Decimator.cc, Interpolator.cc, Int16/Interpolator_int16.cc) ----------
// FIR only: every output is its own sequential sum over the samples behind it, so one thread per output and
// channel; the samples come from [history | this call's input].  `hist` holds the last `hist_len` samples of
// the stream before this call (zeros at the stream start).  Accumulation order and rounding are the reference's.
struct ResampleLaunch {
    const void *in;        // [n_ch][n_in]
    void *out;             // [n_ch][n_out]
    const void *hist;      // [n_ch][hist_len]
    void *hist_next;       // [n_ch][hist_len]: the history after this call
    const void *taps;      // float taps, or int16 Q15 taps for the int16 interpolator (polyphase order for both interpolators)
    uint32_t n_ch, n_in, n_out, hist_len, n_taps, factor, first;   // first: input index of the first decimator output
};

template <class T>
__device__ __forceinline__ T resample_sample(const ResampleLaunch &a, uint32_t ch, int64_t idx)   // idx < 0: history
{
    if (idx >= 0) return ((const T *)a.in)[(size_t)ch * a.n_in + idx];
    const int64_t h = (int64_t)a.hist_len + idx;
    return h >= 0 ? ((const T *)a.hist)[(size_t)ch * a.hist_len + h] : (T)0;
}

// Decimator::decimate -> filterData (Decimator.cc:176-214, :283-321): y = 0; y = y + h[k] * x[n-k]
__global__ void decimate_f32_kernel(const ResampleLaunch a)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.n_ch * a.n_out) return;
    const uint32_t ch = (uint32_t)(t / a.n_out), j = (uint32_t)(t - (size_t)ch * a.n_out);
    const int64_t n = (int64_t)a.first + (int64_t)j * a.factor;
    const float *h = (const float *)a.taps;
    float y = 0;
    for (uint32_t k = 0; k < a.n_taps; k++) y = y + (h[k] * resample_sample<float>(a, ch, n - k));
    ((float *)a.out)[t] = y;
}

// Interpolator::interpolate -> filterData(sub-filter) (Interpolator.cc:166-196, :340-364); taps in polyphase order
__global__ void interpolate_f32_kernel(const ResampleLaunch a)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.n_ch * a.n_out) return;
    const uint32_t ch = (uint32_t)(t / a.n_out), j = (uint32_t)(t - (size_t)ch * a.n_out);
    const uint32_t n = j / a.factor, p = j - n * a.factor;
    const float *h = (const float *)a.taps + (size_t)p * a.n_taps;   // n_taps = taps per sub-filter
    float y = 0;
    for (uint32_t k = 0; k < a.n_taps; k++) y = y + (h[k] * resample_sample<float>(a, ch, (int64_t)n - k));
    ((float *)a.out)[t] = y;
}

// Interpolator_int16::filterData (Interpolator_int16.cc:203-246): Q15, rounding term, clamp after every MAC
__global__ void interpolate_i16_kernel(const ResampleLaunch a)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.n_ch * a.n_out) return;
    const uint32_t ch = (uint32_t)(t / a.n_out), j = (uint32_t)(t - (size_t)ch * a.n_out);
    const uint32_t n = j / a.factor, p = j - n * a.factor;
    const int16_t *h = (const int16_t *)a.taps + (size_t)p * a.n_taps;
    int32_t acc = 1 << 14;
    for (uint32_t k = 0; k < a.n_taps; k++) {
        acc = acc + ((int32_t)h[k] * (int32_t)resample_sample<int16_t>(a, ch, (int64_t)n - k));
        acc = clamp_q30(acc);
    }
    ((int16_t *)a.out)[t] = (int16_t)(acc >> 15);
}

// The history after the call: the last hist_len samples of [history | input].
template <class T>
__global__ void resample_history_kernel(const ResampleLaunch a)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.n_ch * a.hist_len) return;
    const uint32_t ch = (uint32_t)(t / a.hist_len), k = (uint32_t)(t - (size_t)ch * a.hist_len);
    ((T *)a.hist_next)[t] = resample_sample<T>(a, ch, (int64_t)a.n_in - (int64_t)a.hist_len + k);
}

hipError_t launch_resample(int kind, const void *in, void *out, const void *hist, void *hist_next, const void *taps,
                           uint32_t n_ch, uint32_t n_in, uint32_t n_out, uint32_t hist_len, uint32_t n_taps,
                           uint32_t factor, uint32_t first, hipStream_t s)
{
    ResampleLaunch a{in, out, hist, hist_next, taps, n_ch, n_in, n_out, hist_len, n_taps, factor, first};
    const size_t n = (size_t)n_ch * n_out;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (n) {
        if (kind == 0) hipLaunchKernelGGL(decimate_f32_kernel, grid, block, 0, s, a);
        else if (kind == 1) hipLaunchKernelGGL(interpolate_f32_kernel, grid, block, 0, s, a);
        else hipLaunchKernelGGL(interpolate_i16_kernel, grid, block, 0, s, a);
    }
    const size_t nh = (size_t)n_ch * hist_len;
    if (nh) {
        const dim3 gh((unsigned)((nh + 255) / 256));
        if (kind == 2) hipLaunchKernelGGL(resample_history_kernel<int16_t>, gh, block, 0, s, a);
        else hipLaunchKernelGGL(resample_history_kernel<float>, gh, block, 0, s, a);
    }
    return hipGetLastError();
}

// Repeats the first `period` bytes of a buffer over the rest (bench input staging).
__global__ void tile_fill_kernel(uint8_t *dst, size_t period, size_t total)
{
    const size_t n16 = total / 16, p16 = period / 16;
    for (size_t i = p16 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16;
         i += (size_t)gridDim.x * blockDim.x)
        ((uint4 *)dst)[i] = ((const uint4 *)dst)[i % p16];
}

// ---- launch wrappers ---------------------------------------------------------------------
hipError_t launch_wbfm(const ChainLaunch &a, bool gated, bool mag, uint32_t n_blocks, hipStream_t s)
{
    dim3 grid(n_blocks), block(WB_THREADS);
    if (gated) {
        if (mag) hipLaunchKernelGGL((wbfm_chain_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wbfm_chain_kernel<true, false>), grid, block, 0, s, a);
    } else {
        if (mag) hipLaunchKernelGGL((wbfm_chain_kernel<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wbfm_chain_kernel<false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_fm(const ChainLaunch &a, bool gated, bool mag, uint32_t n_blocks, hipStream_t s)
{
    dim3 grid(n_blocks), block(WB_THREADS);
    if (gated) hipLaunchKernelGGL((fm_chain_kernel<true, false>), grid, block, 0, s, a);
    else if (mag) hipLaunchKernelGGL((fm_chain_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((fm_chain_kernel<false, false>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_am(const ChainLaunch &a_in, int family, bool gated, bool mag, uint32_t n_blocks, hipStream_t s)
{
    dim3 grid(n_blocks), block(WB_THREADS);
    ChainLaunch a = a_in;
    // DC pass: one lane per channel for rows of up to one 32768-byte block (512 outputs), one wave per channel
    // above that (measured at 4096 channels: 2^14 samples 0.123 against 0.153 ms, 2^16 samples 0.417 against 0.317)
    const bool batch = a.vlen / 32 <= 512;
    a.base_stride_ch = batch ? 1 : a.pcm_stride; // time-major scratch for it, channel-major otherwise
    a.base_stride_t = batch ? a.n_ch_call : 1;
    if (gated) hipLaunchKernelGGL((am_chain_kernel<true, false>), grid, block, 0, s, a, family);
    else if (mag) hipLaunchKernelGGL((am_chain_kernel<false, true>), grid, block, 0, s, a, family);
    else hipLaunchKernelGGL((am_chain_kernel<false, false>), grid, block, 0, s, a, family);
    return launch_am_dc(a, family, s, false);
}

hipError_t launch_am_dc(const ChainLaunch &a_in, int family, hipStream_t s, bool channel_major)
{
    ChainLaunch a = a_in;
    const bool batch = !channel_major && a.vlen / 32 <= 512;   // (the streaming pipelines write the detector stream channel-major)
    a.base_stride_ch = batch ? 1 : a.pcm_stride;
    a.base_stride_t = batch ? a.n_ch_call : 1;
    // short streams: one lane per channel; long streams: one wave per channel, segmented
    if (batch) {
        hipLaunchKernelGGL(dc_kernel, dim3((a.n_list + 63) / 64), dim3(64), 0, s, a, family);
    } else if (a.dc_tiles >= 2 && a.dc_records) {   // rows longer than one tile: many waves per channel
        hipLaunchKernelGGL(dc_tiled_kernel, dim3(a.dc_tiles, a.n_list), dim3(64), 0, s, a, family);
        hipLaunchKernelGGL(dc_chainup_kernel, dim3((a.n_list * a.dc_tiles + 255) / 256), dim3(256), 0, s, a, family);
        hipLaunchKernelGGL(dc_redo_kernel, dim3(a.n_list), dim3(64), 0, s, a, family);
    } else {
        hipLaunchKernelGGL(dc_wave_kernel, dim3(a.n_list), dim3(64), 0, s, a, family);
    }
    return hipGetLastError();
}

hipError_t launch_wbfm_verify(const ChainLaunch &a, hipStream_t s)
{
    const uint32_t n = a.n_list * a.tiles_per_ch;
    hipLaunchKernelGGL(wbfm_verify_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_wbfm_repair(const ChainLaunch &a, bool gated, hipStream_t s)
{
    if (gated) hipLaunchKernelGGL((wbfm_repair_kernel<true>), dim3(a.n_list), dim3(WB_THREADS), 0, s, a);
    else hipLaunchKernelGGL((wbfm_repair_kernel<false>), dim3(a.n_list), dim3(WB_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tail_update(const ChainLaunch &a, int family, hipStream_t s)
{
    hipLaunchKernelGGL(tail_update_kernel, dim3(a.n_list), dim3(256), 0, s, a, family);
    return hipGetLastError();
}

__global__ void write_word_kernel(uint32_t *word, uint32_t value) { *(volatile uint32_t *)word = value; }
hipError_t launch_write_word(uint32_t *word, uint32_t value, hipStream_t s)
{
    hipLaunchKernelGGL(write_word_kernel, dim3(1), dim3(1), 0, s, word, value);
    return hipGetLastError();
}

hipError_t launch_magnitude(const uint8_t *iq, size_t ch_stride_bytes, const uint32_t *ch_list, uint32_t n_list,
                            uint32_t block_samples, uint32_t n_blocks, uint32_t *mag_sums, hipStream_t s)
{
    hipLaunchKernelGGL(magnitude_kernel, dim3(n_blocks, n_list), dim3(256), 0, s, iq, ch_stride_bytes, ch_list,
                       block_samples, n_blocks, mag_sums);
    return hipGetLastError();
}

hipError_t launch_squelch(const SquelchLaunch &q, bool always_open, hipStream_t s, const ChainLaunch *tail_of, int tail_family, bool tail_dc)
{
    const uint32_t n = q.n_ch * q.n_blocks;
    if (tail_of && tail_dc)
        hipLaunchKernelGGL(tail_dc_squelch_kernel, dim3(2 * tail_of->n_list + (n + 255) / 256), dim3(256), 0, s, *tail_of, tail_family, q, always_open ? 1 : 0);
    else if (tail_of && tail_family == FAM_WBFM)
        hipLaunchKernelGGL(wbfm_repair_squelch_kernel, dim3(tail_of->n_list + (n + 255) / 256), dim3(WB_THREADS), 0, s, *tail_of, q, always_open ? 1 : 0);
    else if (tail_of)
        hipLaunchKernelGGL(tail_squelch_kernel, dim3(tail_of->n_list + (n + 255) / 256), dim3(256), 0, s, *tail_of, tail_family, q, always_open ? 1 : 0);
    else
        hipLaunchKernelGGL(squelch_block_kernel, dim3((n + 255) / 256), dim3(256), 0, s, q, always_open ? 1 : 0);
    if (!always_open || q.any_agc) {
        if (q.n_blocks >= 32)   // long rows: one wave per channel
            hipLaunchKernelGGL(squelch_track_wave_kernel, dim3(q.n_ch), dim3(64), 0, s, q, always_open ? 1 : 0);
        else
            hipLaunchKernelGGL(squelch_track_kernel, dim3((q.n_ch + 63) / 64), dim3(64), 0, s, q, always_open ? 1 : 0);
    }
    return hipGetLastError();
}

hipError_t launch_mixed_tail(const MixedTailArgs &m, hipStream_t s)
{
    // role k's workgroups end at roles.end[k]: fix-up, DC AM, DC SSB, tails AM, tails FM, tails SSB
    MixedTailRoles roles;
    uint32_t at = 0;
    if (m.a[FAM_WBFM].wg_count) at += (m.sa.n_segments + FIX_SEGS - 1) / FIX_SEGS;
    roles.end[0] = at;
    if (m.a[FAM_AM].wg_count) at += m.a[FAM_AM].n_list;
    roles.end[1] = at;
    if (m.a[FAM_SSB].wg_count) at += m.a[FAM_SSB].n_list;
    roles.end[2] = at;
    if (m.a[FAM_AM].wg_count) at += m.a[FAM_AM].n_list;
    roles.end[3] = at;
    if (m.a[FAM_FM].wg_count) at += m.a[FAM_FM].n_list;
    roles.end[4] = at;
    if (m.a[FAM_SSB].wg_count) at += m.a[FAM_SSB].n_list;
    roles.end[5] = at;
    if (at) hipLaunchKernelGGL(mixed_tail_kernel, dim3(at), dim3(256), 0, s, m.a[FAM_WBFM], m.sa, m.a[FAM_AM], m.a[FAM_FM], m.a[FAM_SSB], roles);
    return hipGetLastError();
}

hipError_t launch_agc_apply(const AgcConfig *cfg, AgcState *st, const ScanConfig *scfg, ScanState *sst,
                            ChanParams *params, GainEpoch *epochs, uint32_t n_ch, hipStream_t s)
{
    hipLaunchKernelGGL(agc_apply_kernel, dim3((n_ch + 255) / 256), dim3(256), 0, s, cfg, st, scfg, sst, params, epochs, n_ch);
    return hipGetLastError();
}

hipError_t launch_reset(uint8_t *tails, WbfmCarry *wc, DcCarry *dc, uint32_t first_ch, uint32_t n_ch,
                        uint32_t family_mask, hipStream_t s)
{
    uint32_t blocks = (n_ch + 255) / 256;
    if (blocks < 64) blocks = 64;
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(256), 0, s, tails, wc, dc, first_ch, n_ch, family_mask);
    return hipGetLastError();
}

hipError_t launch_front_end(const uint8_t *iq, int8_t *out, const ChanParams *params, uint32_t first_ch, uint32_t n_ch,
                            size_t bytes_per_ch, hipStream_t s)
{
    const size_t n = bytes_per_ch / 8 * n_ch;
    hipLaunchKernelGGL(front_end_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, iq, out, params, first_ch,
                       n_ch, bytes_per_ch, 0);
    return hipGetLastError();
}

hipError_t launch_rotate_signed(int8_t *buf, size_t bytes, int rotation, hipStream_t s)
{
    const size_t n = bytes / 8;
    hipLaunchKernelGGL(front_end_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *)buf, buf,
                       (const ChanParams *)nullptr, 0u, 1u, bytes, rotation);
    return hipGetLastError();
}

hipError_t launch_retail(uint8_t *tails, const ChanParams *params, uint32_t n_ch, hipStream_t s)
{
    hipLaunchKernelGGL(retail_kernel, dim3(n_ch), dim3(256), 0, s, tails, params, n_ch);
    return hipGetLastError();
}

hipError_t launch_tile_fill(uint8_t *dst, size_t period, size_t total, hipStream_t s)
{
    hipLaunchKernelGGL(tile_fill_kernel, dim3(2048), dim3(256), 0, s, dst, period, total);
    return hipGetLastError();
}

}  // namespace iqd
