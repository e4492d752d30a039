// WBFM streaming pipeline: what a lane knows about its segment, and the boundary fix-up that follows the stream kernel
// (iqd_stream.hip).  A header because the fix-up also runs as one role of the launch that finishes several families'
// streaming kernels at once (iqd_kernels.hip: mixed_tail_kernel).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "iqd_kernels.h"
#include "iqd_stream.h"
#include "iqd_stream_mixed.h"
#include "iqd_wbfm.h"
#include "iqd_mfma.h"

namespace iqd {

struct StSeg {           // what a lane knows about its segment
    uint32_t valid, li, tile, ch, ech;
    int32_t v0, tlen;
    int32_t vlen;        // samples the channel consumes in this call: all of them, or those of its open blocks (squelch-gated call)
};

__device__ __forceinline__ StSeg st_segment(const ChainLaunch &a, uint32_t sid, uint32_t n_segments)
{
    StSeg s;
    s.valid = sid < n_segments ? 1u : 0u;
    const uint32_t id = s.valid ? sid : 0u;
    s.li = id / a.tiles_per_ch;
    s.tile = id - s.li * a.tiles_per_ch;
    s.ch = a.ch_list[s.li];
    s.ech = a.first_ch + s.ch;
    // a squelch-gated call: the chain sees the concatenation of the channel's open blocks (IqDataProcessor.cc:793), a
    // virtual stream of vlen_gated[ch] samples, and the segments are cut on that axis
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[s.ch] : a.vlen;
    s.vlen = (int32_t)vlen;
    const int64_t v0 = (int64_t)s.tile * a.tile_len;
    if (v0 >= (int64_t)vlen) { s.valid = 0; s.v0 = 0; s.tlen = 0; return s; }
    s.v0 = (int32_t)v0;
    const int64_t rest = (int64_t)vlen - v0;
    s.tlen = (int32_t)(rest < (int64_t)a.tile_len ? rest : (int64_t)a.tile_len);
    if (!s.valid) s.tlen = 0;
    return s;
}

// The same for a launch whose ids are grouped by rotation selector (StreamArgs::grouped); rot = the id's selector.
// (IQD_ST_NO_GROUPS: a translation unit whose launches are never grouped - the one that holds all four families' pipelines,
// iqd_stream_mixed.hip, where every scalar register counts - compiles the plain mapping only.)
#ifndef IQD_ST_NO_GROUPS
#define IQD_ST_NO_GROUPS 0
#endif
__device__ __forceinline__ StSeg st_segment_of(const ChainLaunch &a, const StreamArgs &sa, uint32_t sid, int &rot)
{
    if (IQD_ST_NO_GROUPS || !sa.grouped) { rot = 0; return st_segment(a, sid, sa.n_segments); }
    const uint32_t r = (sid >= sa.group_start[1] ? 1u : 0u) + (sid >= sa.group_start[2] ? 1u : 0u);
    rot = 1 - (int)r;
    const uint32_t local = sid - sa.group_start[r];
    const bool there = sid < sa.group_start[3] && local < sa.group_nseg[r];
    // (an id that is padding maps to no segment: st_segment() then hands back the launch's first channel with length 0)
    return st_segment(a, there ? sa.group_li0[r] * a.tiles_per_ch + local : sa.n_segments, sa.n_segments);
}
// A segment that lets its ring take the IIR wave's fast path: not there at all, or cold (not its channel's first), of full length
// and not the keeper of its channel's restart state (the one before a last segment shorter than FORCED_BACK).  (st_iir_wave)
__device__ __forceinline__ bool st_cold_and_full(const StSeg &s, uint32_t tile_len)
{
    if (!s.valid) return true;
    const bool not_last = s.vlen - s.v0 > s.tlen;
    const bool keeps_restart = not_last && s.vlen - FORCED_BACK >= s.v0 && s.vlen - FORCED_BACK < s.v0 + s.tlen;
    return s.tile != 0 && s.tlen == (int32_t)tile_len && !keeps_restart;
}
__device__ __forceinline__ uint32_t st_id_count(const StreamArgs &sa) { return !IQD_ST_NO_GROUPS && sa.grouped ? sa.group_start[3] : sa.n_segments; }

// The first ST_FIX_PCM PCM samples of every cold segment, recomputed with the exact histories its predecessor left
// (StHist): stage-1 output 0, stage-2 outputs 0..2, then the 40-tap audio decimator in the reference's MAC order with
// the clamp after every MAC (Decimator_int16.cc:176-238) - which is also what the clamp-free fast path equals when
// no clamp can fire.  A workgroup takes FIX_SEGS consecutive segments: their FIX_SEGS + 1 records (the predecessor of the first
// included) are one contiguous read into LDS (16 + 1 records: 4352 bytes); then FIX_LANES threads per segment, a PCM sample (or two) each.
// (round 6: IQD_FIX_LANES threads per segment - 32 in rounds 2-5, one per PCM sample with 11 idle; 16 with two samples for the first
// five halve the launch's waves, and its workgroups then nearly fit the chip at once: the kernel is its trips to memory)
#ifndef IQD_FIX_LANES
#define IQD_FIX_LANES 16
#endif
constexpr int FIX_LANES = IQD_FIX_LANES;
constexpr int FIX_SEGS = 256 / FIX_LANES;
constexpr int FIX_THREADS = FIX_LANES * FIX_SEGS;
static_assert(FIX_LANES == 16 || FIX_LANES == 32, "threads per boundary");
struct FixLds {
    uint32_t rec[(FIX_SEGS + 1) * 64];
    uint32_t y2x[FIX_SEGS][41];             // per segment: stage-2 pairs -40 .. 41 with the boundary ones exact
};
__device__ __forceinline__ void wbfm_stream_fixup_body(const ChainLaunch &a, const StreamArgs &sa, const uint32_t block, FixLds &fl)
{
    uint32_t (&rec)[(FIX_SEGS + 1) * 64] = fl.rec;
    uint32_t (&y2x)[FIX_SEGS][41] = fl.y2x;
    const uint32_t sid0 = block * FIX_SEGS;
    const int tid = (int)threadIdx.x, sl = tid / FIX_LANES, i = tid % FIX_LANES;
    // Everything this workgroup needs from memory is asked for at once - the segment's channel, the two states of the
    // hand-off, the nine records - instead of one dependent trip after the other (round 2: 17.6 us of latency for
    // 47 662 boundaries).
    const uint32_t sid = sid0 + (uint32_t)sl;
    const StSeg sg = st_segment(a, sid < sa.n_segments ? sid : 0u, sa.n_segments);
    const bool live = sid < sa.n_segments && sg.valid && sg.tile != 0;
    float y_in = 0.f, y_end = 0.f;
    if (live && i == FIX_LANES - 1) {
        const WbfmRecord *r = a.records + (size_t)sg.li * a.tiles_per_ch;
        y_in = r[sg.tile].y_in;
        y_end = r[sg.tile - 1].y_end;
    }
    {
        const uint32_t *src = (const uint32_t *)sa.hist + ((size_t)sid0 - (sid0 ? 1 : 0)) * 64;
        const uint32_t n_avail = (sa.n_segments - sid0 < (uint32_t)FIX_SEGS ? sa.n_segments - sid0 : (uint32_t)FIX_SEGS) + (sid0 ? 1u : 0u);
        for (uint32_t k = (uint32_t)tid; k < n_avail * 64; k += FIX_THREADS) rec[k + (sid0 ? 0 : 64)] = src[k];
    }
    __syncthreads();
    const StHist &prev = *(const StHist *)&rec[sl * 64], &own = *(const StHist *)&rec[(sl + 1) * 64];
    if (live && i < 3) {                               // stage-2 output i (0..2) of the segment, from y1[4i-8 .. 4i+3]
        int acc = 1 << 14;                             // stage-1 output 0: w[-4 .. 3]
        acc = dot2(prev.w_last[0], sa.d1p[0], acc);
        acc = dot2(prev.w_last[1], sa.d1p[1], acc);
        acc = dot2(own.w_first[0], sa.d1p[2], acc);
        acc = dot2(own.w_first[1], sa.d1p[3], acc);
        uint32_t y1[10];                               // pairs: outputs -8 .. 11
#pragma unroll
        for (int k = 0; k < 4; k++) y1[k] = prev.y1_last[k];
#pragma unroll
        for (int k = 0; k < 6; k++) y1[4 + k] = own.y1_first[k];
        y1[4] = (y1[4] & 0xffff0000u) | ((uint32_t)(acc >> 15) & 0xffffu);
        int s2 = 1 << 14;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t d = i == 0 ? y1[5 - q] : (i == 1 ? y1[7 - q] : y1[9 - q]);   // pairs 2i .. 2i+5, newest first
            s2 = dot2(d, sa.p12p[q], s2);
        }
        ((int16_t *)&y2x[sl][20])[i] = (int16_t)(s2 >> 15);   // outputs 0, 1 -> pair 20; output 2 -> low half of pair 21
    } else if (live) {                                 // the other pairs as they are (each thread a few)
        for (int p = i - 3; p < 41; p += FIX_LANES - 3) {
            if (p == 20) continue;
            if (p == 21) ((int16_t *)&y2x[sl][21])[1] = (int16_t)(own.y2_first[1] >> 16);
            else y2x[sl][p] = p < 20 ? prev.y2_last[p] : own.y2_first[p - 20];
        }
    }
    // Hand-off verification of the same segments (what wbfm_verify_kernel does for tile launches): a cold segment's
    // warmed-up state at its start must be its predecessor's end state, bit for bit.  Only mismatches touch the
    // device counters (thousands of workgroups adding to one word would cost more than the whole kernel); the host
    // knows how many hand-offs a launch has.
    if (live && i == FIX_LANES - 1 && f2u(y_in) != f2u(y_end)) {   // (bit-equal is the rule; the sub-2^-100 exception needs the channel's K: rare)
        if (!iir_states_agree(y_in, y_end, a.params[sg.ech].wbfm_k >= 1.0f)) {
            atomicAdd(&a.counters[CNT_TILE_MISMATCH], 1u);
            atomicAdd(&a.counters[CNT_STREAM_MISMATCH], 1u);
            a.repair_flags[sg.li] = 1;
        }
    }
    __syncthreads();   // y2x[sl] was written by this segment's 32 threads, each reads the others' entries (uniform control flow up to here)
    if (!live) return;
    for (int j = i; j < ST_FIX_PCM && j < sg.tlen / 32; j += FIX_LANES) {
        // PCM j from stage-2 outputs 2j-38 .. 2j+1: pairs j+1 .. j+20, newest first
        int s3 = 1 << 14;
#pragma unroll
        for (int q = 0; q < 20; q++) {
            const uint32_t pair = y2x[sl][20 + j - q];
            s3 = clamp_q30(dot2(pair, sa.a40p[q] & 0xffff0000u, s3));
            s3 = clamp_q30(dot2(pair, sa.a40p[q] & 0x0000ffffu, s3));
        }
        a.pcm[(size_t)sg.ch * a.pcm_stride + (sg.v0 >> 5) + j] = (int16_t)(s3 >> 15);
    }
}

}  // namespace iqd
