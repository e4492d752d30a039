// FM, AM and SSB chains (FmDemodulator.cc:376-560, AmDemodulator.cc:339-504,
// SsbDemodulator.cc:462-598 behind IqDataProcessor.cc:735-749) as per-thread phase functions,
// in the same tile / chunk scheme as the WBFM chain (iqd_wbfm.h).
//
// Every decimating stage here is a FIR: with the raw tail in front of a tile all histories are
// rebuilt exactly from bytes, so tiles are independent.  The only recurrence left is the
// 8 kS/s DC-removal IIR of AM and SSB; the tile kernels stop in front of it (they emit its
// integer input at 8 kS/s) and dc_kernel runs it with the exact carried state.
//
// Overflow: the reference clamps its Q15 accumulators after every MAC.  Fed with int8 samples
// the AM/SSB stages and the FM tuner can be bounded once and for all (|acc| stays below 2^30:
// see the bounds next to each stage), so those sums use v_dot4 / v_dot2 in any order; the FM
// post-demodulation stages see arbitrary int16 values and keep the sequential clamp.
#pragma once
#include "iqd_wbfm.h"

namespace iqd {

typedef WbfmTile Tile;

#ifndef IQD_CH_CHUNK
#define IQD_CH_CHUNK 8192
#endif
constexpr int CH_CHUNK = IQD_CH_CHUNK;       // samples per chunk of the FM / AM / SSB tile kernels
// (raw history a tile rebuilds its FIR states from: fir_halo(family), iqd_device.h)

// ---- shared front end: raw u8 -> signed -> rotated rail dwords in LDS, + squelch magnitude ----
template <bool GATED, bool MAG, class Lds>
IQD_DEV void front_rotate(const Tile &t, Lds &lds, const ChunkBlocks &cb, int cstart, int clen, int tid,
                          int rail_hist)
{
    const int ngroups = clen >> 4;
    for (int g = tid; g < ngroups; g += WB_THREADS) {
        const int64_t v = t.v0 + cstart + 16 * g;
        const u32x4 *po = raw_group<GATED>(t, v);
        const u32x4 r2 = po[0], r3 = po[1];
        uint32_t s[8] = {r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
        uint32_t xi[4], xq[4];
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] ^= 0x80808080u;
#pragma unroll
        for (int j = 0; j < 4; j++) rotate4(t, s[2 * j], s[2 * j + 1], xi[j], xq[j]);
        *(u32x4 *)&lds.xi[rail_hist + 4 * g] = u32x4{xi[0], xi[1], xi[2], xi[3]};
        *(u32x4 *)&lds.xq[rail_hist + 4 * g] = u32x4{xq[0], xq[1], xq[2], xq[3]};
        if (MAG && cstart >= 0) {
            const uint32_t m = magnitude16(s);
            const uint32_t slot = div_block(t, cb.in_blk + (uint32_t)(16 * g));
#if IQD_ON_DEVICE
            atomicAdd(&lds.mag[slot], m);
#else
            lds.mag[slot] += m;
#endif
        }
    }
}

template <class Lds>
IQD_DEV void flush_mag(const Tile &t, Lds &lds, const ChunkBlocks &cb, int cstart, int clen, int tid)
{
    if (cstart < 0) return;
    const uint32_t nslots = div_block(t, cb.in_blk + (uint32_t)clen - 1) + 1;
    if ((uint32_t)tid < nslots) {
        const uint32_t m = lds.mag[tid];
        lds.mag[tid] = 0;
#if IQD_ON_DEVICE
        if (m) atomicAdd(&t.mag_row[cb.base_blk + tid], m);
#else
        t.mag_row[cb.base_blk + tid] += m;
#endif
    }
}

// Q15 dot product without the clamp (only where the bound proves it cannot fire);
// buf is an int16 array, `newest` the index of x[n], taps newest first.
template <int L>
IQD_DEV int q15_free(const int16_t *h, const uint32_t *buf, int newest)
{
    int acc = 1 << 14;
#pragma unroll
    for (int k = 0; k < L; k++) {
        const int idx = newest - k;
        const uint32_t pair = buf[idx >> 1];
        const uint32_t tap = (idx & 1) ? ((uint32_t)(uint16_t)h[k] << 16) : (uint32_t)(uint16_t)h[k];
        acc = dot2(pair, tap, acc);
    }
    return acc >> 15;
}

// Moves the last `n_dwords` of a buffer's data region (which holds `used` dwords after a
// history region of n_dwords) to its front.  Ranges never overlap when used >= n_dwords.
IQD_DEV void shift_hist(uint32_t *buf, int n_dwords, int used, int tid, int first_tid)
{
    const int k = tid - first_tid;
    if (k < 0 || k >= n_dwords) return;
    if (used >= n_dwords) buf[k] = buf[used + k];
}
IQD_DEV void shift_hist_serial(uint32_t *buf, int n_dwords, int used)  // used < n_dwords
{
    for (int k = 0; k < n_dwords; k++) buf[k] = buf[used + k];
}

// =============================================================================================
// FM
// =============================================================================================
struct FmLds {
    alignas(16) uint32_t xi[8 + CH_CHUNK / 4];               // rail dwords (4 samples each), 8 of history
    alignas(16) uint32_t xq[8 + CH_CHUNK / 4];
    alignas(16) uint32_t theta[4 + CH_CHUNK / 4];            // 64 kS/s phase (float bits), 4 of history
    alignas(16) uint32_t e[(12 + CH_CHUNK / 4) / 2];         // int16 (int16)(K*dtheta), 12 of history
    alignas(16) uint32_t y2[(40 + CH_CHUNK / 16) / 2];       // int16 16 kS/s, 40 of history
    alignas(16) uint32_t mag[CH_CHUNK / SEG + 2];
    uint32_t e_peak, e_peak_hist, y2_peak, y2_peak_hist;   // loudness flags for the clamp-free fast paths
};

// Tuner decimator /4, 32 taps (FmDemodulator.cc:389-419) + phase angle (:476).
// Bound: |acc| <= 16384 + 35938*128 < 2^23, no clamp; |y| <= 141 -> the theta table.
IQD_DEV void fm_stage1(const Tile &t, const Consts &c, FmLds &lds, int clen, int tid, const float *fm_lut)
{
    const int ngroups = clen >> 4;
    const int round = 1 << 14;   // Q15 rounding term, kept in one VGPR for every chain head
    for (int g = tid; g < ngroups; g += WB_THREADS) {
        const u32x4 *pi = (const u32x4 *)&lds.xi[4 * g], *pq = (const u32x4 *)&lds.xq[4 * g];
        const u32x4 a0 = pi[0], a1 = pi[1], a2 = pi[2], b0 = pq[0], b1 = pq[1], b2 = pq[2];
        const uint32_t di[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
        const uint32_t dq[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
        float th[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {   // output m = 4g + r uses rail dwords [m-7, m] = d[1+r .. 8+r]
            int li = 0, hi = 0, lq = 0, hq = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                li = q == 0 ? dot4_first(di[1 + r], c.fm_tuner_lo[0], round) : dot4(di[1 + r + q], c.fm_tuner_lo[q], li);
                hi = q == 0 ? dot4_first0(di[1 + r], c.fm_tuner_hi[0]) : dot4(di[1 + r + q], c.fm_tuner_hi[q], hi);
                lq = q == 0 ? dot4_first(dq[1 + r], c.fm_tuner_lo[0], round) : dot4(dq[1 + r + q], c.fm_tuner_lo[q], lq);
                hq = q == 0 ? dot4_first0(dq[1 + r], c.fm_tuner_hi[0]) : dot4(dq[1 + r + q], c.fm_tuner_hi[q], hq);
            }
            const int yi = (li + (int)((uint32_t)hi << 8)) >> 15;
            const int yq = (lq + (int)((uint32_t)hq << 8)) >> 15;
            th[r] = fm_lut[(yq + FM_LUT_R) * FM_LUT_W + (yi + FM_LUT_R)];
        }
        *(u32x4 *)&lds.theta[4 + 4 * g] = u32x4{f2u(th[0]), f2u(th[1]), f2u(th[2]), f2u(th[3])};
    }
    (void)t;
}

// Differentiator theta[n-2] - theta[n-4] (FmDemodulator.cc:113-122,479), branch cut, scale,
// (int16) cast (:485-498, :540).
IQD_DEV void fm_discriminate(const Tile &t, FmLds &lds, int clen, int tid)
{
    // eight consecutive outputs per lane: theta[m-4 .. m+5] in three 16-byte reads, the int16 results in two 8-byte writes
    const int ngroups = clen >> 5;   // clen is a multiple of 128
    uint32_t peak = 0;
    for (int g = tid; g < ngroups; g += WB_THREADS) {
        const u32x4 *p = (const u32x4 *)&lds.theta[8 * g];   // theta of output m sits at index 4 + m
        const u32x4 a = p[0], b = p[1], c4 = p[2];
        const float th[10] = {u2f(a.x), u2f(a.y), u2f(a.z), u2f(a.w), u2f(b.x), u2f(b.y), u2f(b.z), u2f(b.w), u2f(c4.x), u2f(c4.y)};
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {   // output m = 8g + k: theta[m-2] - theta[m-4] = th[k+2] - th[k]
            float d = th[k + 2] - th[k];
            d = wrap_delta(d);
            const float v = t.k * d;
            const int ev = (int)(int16_t)(uint16_t)(t.bounded ? cast_i16_bounded(v) : (uint32_t)cast_i16(v));
            e[k] = (uint32_t)ev;
            const uint32_t m = (uint32_t)(ev < 0 ? -ev : ev);
            peak = m > peak ? m : peak;
        }
        u32x2 *dst = (u32x2 *)&lds.e[6 + 4 * g];   // int16 index 12 + 8g
        dst[0] = u32x2{pack_lo16(e[0], e[1]), pack_lo16(e[2], e[3])};
        dst[1] = u32x2{pack_lo16(e[4], e[5]), pack_lo16(e[6], e[7])};
    }
    if (peak > (uint32_t)POST12_SAFE) lds_max(&lds.e_peak, peak);
}

IQD_DEV void fm_post(const Consts &c, FmLds &lds, int clen, int tid)   // /4, 12 taps (:545)
{
    const int nout = clen >> 4;
    const bool quiet = lds.e_peak <= (uint32_t)POST12_SAFE && lds.e_peak_hist <= (uint32_t)POST12_SAFE;
    uint32_t peak = 0;
    for (int j = tid; j < nout; j += WB_THREADS) {
        const int y = quiet ? q15_pairs<12>(c.post12, lds.e, 12 + 4 * j + 3) : q15_seq<12>(c.post12, lds.e, 12 + 4 * j + 3);
        put_i16(lds.y2, 40 + j, y);
        const uint32_t a = (uint32_t)(y < 0 ? -y : y);
        peak = a > peak ? a : peak;
    }
    if (peak > (uint32_t)AUDIO40_SAFE) lds_max(&lds.y2_peak, peak);
}

IQD_DEV void fm_audio(const Consts &c, FmLds &lds, const Tile &t, int cstart, int clen, int tid)  // /2, 40 taps
{
    const int nout = clen >> 5;
    const bool quiet = lds.y2_peak <= (uint32_t)AUDIO40_SAFE && lds.y2_peak_hist <= (uint32_t)AUDIO40_SAFE;
    for (int i = tid; i < nout; i += WB_THREADS) {
        const int y = quiet ? q15_pairs<40>(c.audio40, lds.y2, 40 + 2 * i + 1)
                            : q15_seq<40>(c.audio40, lds.y2, 40 + 2 * i + 1);
        if (cstart >= 0) t.pcm_row[((t.v0 + cstart) >> 5) + i] = (int16_t)y;
    }
}

// Histories for the next chunk.  Each buffer's tail moves to its front in the phase that follows its last reader
// (that phase touches other buffers only), so no phase of its own is needed:
//   rails after stage 1 (beside the discriminator), theta beside the post decimator, e and its loudness flags beside
//   the audio decimator, y2 and its flags beside the next chunk's front end.
IQD_DEV void fm_shift_rails(FmLds &lds, int clen, int tid)
{
    shift_hist(lds.xi, 8, clen >> 2, tid, 0);
    shift_hist(lds.xq, 8, clen >> 2, tid, 8);
}
IQD_DEV void fm_shift_theta(FmLds &lds, int clen, int tid) { shift_hist(lds.theta, 4, clen >> 2, tid, 16); }
IQD_DEV void fm_shift_e(FmLds &lds, int clen, int tid)
{
    if (tid == 64) {   // conservative loudness flag for what stays in reach of the next chunk
        const uint32_t keep_e = (clen >> 2) >= 12 ? 0u : lds.e_peak_hist;
        lds.e_peak_hist = lds.e_peak > keep_e ? lds.e_peak : keep_e;
        lds.e_peak = 0;
    }
    // (a 32-sample chunk - a gain change 32 samples before a call - brings 4 dwords against 6 of history: the ranges
    // overlap; round 4: shift_hist() then did nothing at all and the next chunk read a stale history - found by the
    // short-block fuzzer, tests/test_gpu_gain_epochs.py::test_gain_change_then_64_byte_calls)
    if ((clen >> 3) >= 6) shift_hist(lds.e, 6, clen >> 3, tid, 24);
    else if (tid == 24) shift_hist_serial(lds.e, 6, clen >> 3);
}
IQD_DEV void fm_shift_y2(FmLds &lds, int clen, int tid)
{
    if (tid == 64) {
        const uint32_t keep_y = (clen >> 4) >= 40 ? 0u : lds.y2_peak_hist;
        lds.y2_peak_hist = lds.y2_peak > keep_y ? lds.y2_peak : keep_y;
        lds.y2_peak = 0;
    }
    if ((clen >> 5) >= 20) shift_hist(lds.y2, 20, clen >> 5, tid, 32);
    else if (tid == 32) shift_hist_serial(lds.y2, 20, clen >> 5);
}

template <bool GATED, bool MAG, class Exec>
IQD_DEV void fm_tile(Exec &ex, const Tile &t, const Consts &c, FmLds &lds, const float *fm_lut)
{
    ex.all([&](int tid) {
        if (tid < 8) lds.xi[tid] = 0, lds.xq[tid] = 0;
        if (tid < 4) lds.theta[tid] = 0;
        if (tid < 6) lds.e[tid] = 0;
        if (tid < 20) lds.y2[tid] = 0;
        if (tid < CH_CHUNK / SEG + 2) lds.mag[tid] = 0;
        if (tid == 0) lds.e_peak = lds.e_peak_hist = lds.y2_peak = lds.y2_peak_hist = 0;
    });
    int prev_clen = 0;
    for (int cstart = -fir_halo(FAM_FM); cstart < t.tlen;) {
        const int clen = wbfm_chunk_len(t, cstart, CH_CHUNK);   // the lead-in splits where the gain last changed
        Tile tc = t;
        tc.k = wbfm_chunk_gain(t, cstart);
        const ChunkBlocks cb = chunk_blocks(t, cstart);
        ex.all([&](int tid) {
            if (prev_clen) fm_shift_y2(lds, prev_clen, tid);
            front_rotate<GATED, MAG>(t, lds, cb, cstart, clen, tid, 8);
        });
        ex.all([&](int tid) {
            if (MAG) flush_mag(t, lds, cb, cstart, clen, tid);
            fm_stage1(t, c, lds, clen, tid, fm_lut);
        });
        ex.all([&](int tid) {
            fm_discriminate(tc, lds, clen, tid);
            fm_shift_rails(lds, clen, tid);
        });
        ex.all([&](int tid) {
            fm_post(c, lds, clen, tid);
            fm_shift_theta(lds, clen, tid);
        });
        ex.all([&](int tid) {
            fm_audio(c, lds, t, cstart, clen, tid);
            fm_shift_e(lds, clen, tid);
        });
        prev_clen = clen;
        cstart += clen;
    }
}

// =============================================================================================
// AM and SSB (shared /32 front end)
// =============================================================================================
struct AmLds {
    alignas(16) uint32_t xi[4 + CH_CHUNK / 4];               // rail dwords, 4 of history (1 needed)
    alignas(16) uint32_t xq[4 + CH_CHUNK / 4];
    alignas(16) uint32_t s1i[(12 + CH_CHUNK / 4) / 2];       // int16 64 kS/s, 12 of history (8 needed)
    alignas(16) uint32_t s1q[(12 + CH_CHUNK / 4) / 2];
    alignas(16) uint32_t s2i[(16 + CH_CHUNK / 16) / 2];      // int16 16 kS/s, 16 of history (14 needed)
    alignas(16) uint32_t s2q[(16 + CH_CHUNK / 16) / 2];
    alignas(16) uint32_t s3i[(32 + CH_CHUNK / 32) / 2];      // int16 8 kS/s, 32 of history (30 needed by SSB)
    alignas(16) uint32_t s3q[(32 + CH_CHUNK / 32) / 2];
    alignas(16) uint32_t mag[CH_CHUNK / SEG + 2];
};

// Stage 1: /4, 8 taps.  |acc| <= 16384 + 29002*128 < 2^22; |y| <= 113.
IQD_DEV void am_stage1(const Consts &c, AmLds &lds, int clen, int tid)
{
    const int ngroups = clen >> 4;
    const int round = 1 << 14;
    for (int g = tid; g < ngroups; g += WB_THREADS) {
        // outputs m = 4g + r use rail dwords [m-1, m] = indices 4 + m - 1, 4 + m
        uint32_t di[5], dq[5];
        di[0] = lds.xi[4 + 4 * g - 1];
        dq[0] = lds.xq[4 + 4 * g - 1];
        const u32x4 a = *(const u32x4 *)&lds.xi[4 + 4 * g], b = *(const u32x4 *)&lds.xq[4 + 4 * g];
        di[1] = a.x; di[2] = a.y; di[3] = a.z; di[4] = a.w;
        dq[1] = b.x; dq[2] = b.y; dq[3] = b.z; dq[4] = b.w;
        int yi[4], yq[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int li = dot4_first(di[r], c.am_s1_lo[0], round), hi = dot4_first0(di[r], c.am_s1_hi[0]);
            int lq = dot4_first(dq[r], c.am_s1_lo[0], round), hq = dot4_first0(dq[r], c.am_s1_hi[0]);
            li = dot4(di[r + 1], c.am_s1_lo[1], li);
            hi = dot4(di[r + 1], c.am_s1_hi[1], hi);
            lq = dot4(dq[r + 1], c.am_s1_lo[1], lq);
            hq = dot4(dq[r + 1], c.am_s1_hi[1], hq);
            yi[r] = (li + (int)((uint32_t)hi << 8)) >> 15;
            yq[r] = (lq + (int)((uint32_t)hq << 8)) >> 15;
        }
        *(u32x2 *)&lds.s1i[6 + 2 * g] = u32x2{pack_lo16((uint32_t)yi[0], (uint32_t)yi[1]), pack_lo16((uint32_t)yi[2], (uint32_t)yi[3])};
        *(u32x2 *)&lds.s1q[6 + 2 * g] = u32x2{pack_lo16((uint32_t)yq[0], (uint32_t)yq[1]), pack_lo16((uint32_t)yq[2], (uint32_t)yq[3])};
    }
}

// Stage 2: /4, 12 taps.  |x| <= 113 -> |acc| <= 16384 + 34926*113 < 2^22; |y| <= 121.
IQD_DEV void am_stage2(const Consts &c, AmLds &lds, int clen, int tid)
{
    const int nout = clen >> 4;
    for (int j = tid; j < nout; j += WB_THREADS) {
        put_i16(lds.s2i, 16 + j, q15_free<12>(c.am_s2, lds.s1i, 12 + 4 * j + 3));
        put_i16(lds.s2q, 16 + j, q15_free<12>(c.am_s2, lds.s1q, 12 + 4 * j + 3));
    }
}

// Stage 3: /2, 16 taps.  |x| <= 121 -> |acc| <= 16384 + 48394*121 < 2^23; |y| <= 179.
IQD_DEV void am_stage3(const Consts &c, AmLds &lds, int clen, int tid)
{
    const int nout = clen >> 5;
    for (int i = tid; i < nout; i += WB_THREADS) {
        put_i16(lds.s3i, 32 + i, q15_free<16>(c.am_s3, lds.s2i, 16 + 2 * i + 1));
        put_i16(lds.s3q, 32 + i, q15_free<16>(c.am_s3, lds.s2q, 16 + 2 * i + 1));
    }
}

// Detector input at 8 kS/s -> global scratch (the DC-removal IIR runs in dc_kernel).
//   AM  (AmDemodulator.cc:446-459): max(|i|,|q|) + min(|i|,|q|)/2 in int16 arithmetic
//   SSB (SsbDemodulator.cc:574-588): delayed I (the 1.0 tap is -32768 in Q15, so -i[n-15])
//       -+ Hilbert-transformed Q.  |x| <= 179 -> Hilbert |acc| <= 16384 + 67250*179 < 2^24.
IQD_DEV void am_detect(const Consts &c, AmLds &lds, const Tile &t, int cstart, int clen, int tid,
                       int ssb, int lsb, int32_t *base_row, int base_stride_t)
{
    const int nout = clen >> 5;
    for (int i = tid; i < nout; i += WB_THREADS) {
        int x;
        if (!ssb) {
            const int iv = get_i16(lds.s3i, 32 + i), qv = get_i16(lds.s3q, 32 + i);
            const int im = (int)(int16_t)(iv < 0 ? -iv : iv), qm = (int)(int16_t)(qv < 0 ? -qv : qv);
            x = (int)(int16_t)((im > qm) ? im + (qm >> 1) : qm + (im >> 1));
        } else {
            const int idl = q15_free<16>(c.ssb_delay, lds.s3i, 32 + i);
            const int qh = q15_free<31>(c.ssb_hilbert, lds.s3q, 32 + i);
            x = lsb ? (int)(int16_t)idl - (int)(int16_t)qh : (int)(int16_t)idl + (int)(int16_t)qh;
        }
        if (cstart >= 0) base_row[(size_t)(((t.v0 + cstart) >> 5) + i) * base_stride_t] = x;
    }
}

// Histories for the next chunk, each moved in the phase that follows its buffer's last reader (see fm_shift_*).
IQD_DEV void am_shift_rails(AmLds &lds, int clen, int tid)
{
    shift_hist(lds.xi, 4, clen >> 2, tid, 0);
    shift_hist(lds.xq, 4, clen >> 2, tid, 4);
}
IQD_DEV void am_shift_s1(AmLds &lds, int clen, int tid)
{
    shift_hist(lds.s1i, 6, clen >> 3, tid, 8);
    shift_hist(lds.s1q, 6, clen >> 3, tid, 16);
}
IQD_DEV void am_shift_s2(AmLds &lds, int clen, int tid)
{
    shift_hist(lds.s2i, 8, clen >> 5, tid, 24);
    shift_hist(lds.s2q, 8, clen >> 5, tid, 32);
}
IQD_DEV void am_shift_s3(AmLds &lds, int clen, int tid)
{
    if ((clen >> 6) >= 16) {
        shift_hist(lds.s3i, 16, clen >> 6, tid, 40);
        shift_hist(lds.s3q, 16, clen >> 6, tid, 56);
    } else if (tid == 40) {
        shift_hist_serial(lds.s3i, 16, clen >> 6);
        shift_hist_serial(lds.s3q, 16, clen >> 6);
    }
}

template <bool GATED, bool MAG, class Exec>
IQD_DEV void am_tile(Exec &ex, const Tile &t, const Consts &c, AmLds &lds, int ssb, int lsb, int32_t *base_row,
                     int base_stride_t)
{
    ex.all([&](int tid) {
        if (tid < 4) lds.xi[tid] = 0, lds.xq[tid] = 0;
        if (tid < 6) lds.s1i[tid] = 0, lds.s1q[tid] = 0;
        if (tid < 8) lds.s2i[tid] = 0, lds.s2q[tid] = 0;
        if (tid < 16) lds.s3i[tid] = 0, lds.s3q[tid] = 0;
        if (tid < CH_CHUNK / SEG + 2) lds.mag[tid] = 0;
    });
    int prev_clen = 0;
    for (int cstart = ssb ? -fir_halo(FAM_SSB) : -fir_halo(FAM_AM); cstart < t.tlen;) {
        const int clen = cstart < 0 ? -cstart : (t.tlen - cstart < CH_CHUNK ? t.tlen - cstart : CH_CHUNK);
        const ChunkBlocks cb = chunk_blocks(t, cstart);
        ex.all([&](int tid) {
            if (prev_clen) am_shift_s3(lds, prev_clen, tid);
            front_rotate<GATED, MAG>(t, lds, cb, cstart, clen, tid, 4);
        });
        ex.all([&](int tid) {
            if (MAG) flush_mag(t, lds, cb, cstart, clen, tid);
            am_stage1(c, lds, clen, tid);
        });
        ex.all([&](int tid) {
            am_stage2(c, lds, clen, tid);
            am_shift_rails(lds, clen, tid);
        });
        ex.all([&](int tid) {
            am_stage3(c, lds, clen, tid);
            am_shift_s1(lds, clen, tid);
        });
        ex.all([&](int tid) {
            am_detect(c, lds, t, cstart, clen, tid, ssb, lsb, base_row, base_stride_t);
            am_shift_s2(lds, clen, tid);
        });
        prev_clen = clen;
        cstart += clen;
    }
}

// DC-removal IIR + gain + (int16) cast for one channel (AmDemodulator.cc:462-465,
// SsbDemodulator.cc:590-592; IirFilter.cc:161-176 with b = {1, -1}, a1 = -0.95):
//   y[n] = ((0 + 1*x[n]) + (-1)*x[n-1]) - (0 + a1*y[n-1]);  pcm = (int16_t)(gain * y[n])
template <class SRC>
IQD_DEV void dc_block_run(const SRC *x, int n, float gain, float a1, DcCarry &st, int16_t *pcm)   // (x may be pcm: x[i] is read before pcm[i] is written)
{
    float xp = st.x_prev, yp = st.y_prev;
    for (int i = 0; i < n; i++) {
        const float xf = (float)x[i];
        const float tn = xf - xp;        // exact: both are small integers
        const float r = a1 * yp;
        const float y = tn - r;
        if (pcm) pcm[i] = (int16_t)cast_i16(gain * y);
        xp = xf;
        yp = y;
    }
    st.x_prev = xp;
    st.y_prev = yp;
}


// ---- the same IIR for long streams: one wave per channel, 64 segments of DC_S samples at a time ----
// Same idea as the WBFM de-emphasis (iqd_wbfm.h): every lane first rebuilds the state at the start of its segment
// over the DC_WSEG segments in front of it from a close guess - two trajectories of this contraction (pole 0.95) that
// start a fraction of a unit in the last place apart become bit-identical within those 128 steps - and is accepted
// only if it reproduces its left neighbour's exact end state bit for bit; otherwise it restarts from that state.
// The input differences x[n] - x[n-1] are formed once (they are exact small integers: the detector's output is within
// +-546, AM / SSB chains on int8 input), and the PCM leaves through LDS as whole dwords: per pass of 2048 samples a lane
// runs 32 + 128 + 32 steps.
// Round 6: a segment's row in LDS holds its 32 differences as int16 pairs until its lane has run them, then its 16 PCM
// pairs IN THE SAME WORDS (5 KB per workgroup instead of 13: every channel's workgroup of a 4096- or 8192-row launch is
// resident at once, where 11 per CU were - configs[4]'s pass, which runs beside the next call's magnitude pre-pass, took three
// rounds of them); a pass whose hand-off check fails (a few per cent) fills its rows again before the lanes re-run.  The
// source is the detector stream as int32 (the tile kernels' scratch) or as int16 in the PCM row itself (the streaming
// pipelines, in place: x[n] is read before pcm[n] is written, pass by pass).
constexpr int DC_S = 32;                    // samples per segment
constexpr int DC_WSEG = 4;                  // segments of warm-up
constexpr int DC_GUESS = 12;                // zero-state responses summed for the guess (0.95^(32*12) ~ 3e-9)
constexpr int DC_SUPER = 64 * DC_S;         // PCM samples per pass of the wave
constexpr int DC_ROW = DC_S / 2 + 1;        // words per segment row (padded: the lanes' rows start in different banks)
struct DcLds {
    uint32_t row[64 * DC_ROW];              // per segment: int16 pairs - its input differences, then its PCM
    float z[DC_GUESS + 64], g[64], e[64];
    float x_carry, y_carry;                 // state entering the pass
    float x_last;                           // the pass's last input sample (the next pass's x_carry; taken before the row is overwritten)
};

IQD_DEV float dc_lo(uint32_t pair) { return (float)(int)(int16_t)(pair & 0xffffu); }
IQD_DEV float dc_hi(uint32_t pair) { return (float)((int)pair >> 16); }

template <class SRC>
IQD_DEV void dc_fill(DcLds &lds, const SRC *x, int at, int nseg, int lane, bool first_fill)
{
    // a lane takes PER consecutive samples (16 bytes) out of every 64 * PER; all of a pass's loads are issued before the first use
    // (one after the other they cost a trip to memory each, which was most of this pass's time)
    constexpr int PER = 16 / (int)sizeof(SRC), LOADS = DC_SUPER / (64 * PER);
    u32x4 v[LOADS];
    int32_t before[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; k++) {                 // (clamped, not predicated: nothing to wait for in between)
        const int i = PER * lane + 64 * PER * k, ic = i < nseg * DC_S ? i : nseg * DC_S - PER;
#if IQD_ON_DEVICE
        v[k] = *(const u32x4 *)(x + at + ic);         // rows and passes start on multiples of 8 samples
#else
        memcpy(&v[k], x + at + ic, 16);
#endif
        before[k] = (int32_t)x[at + ic > 0 ? at + ic - 1 : 0];
    }
#if IQD_ON_DEVICE
    __builtin_amdgcn_sched_barrier(0);                // keep the loads together, ahead of the first use
#endif
#pragma unroll
    for (int k = 0; k < LOADS; k++) {
        const int i = PER * lane + 64 * PER * k;
        if (i >= nseg * DC_S) break;
        int32_t xs[PER];
        if (sizeof(SRC) == 4) { xs[0] = (int32_t)v[k].x; xs[1] = (int32_t)v[k].y; xs[2] = (int32_t)v[k].z; xs[3] = (int32_t)v[k].w; }
        else {
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int q = 0; q < 4; q++) { xs[(2 * q) % PER] = (int)(int16_t)(w[q] & 0xffffu); xs[(2 * q + 1) % PER] = (int)w[q] >> 16; }
        }
        // (the sample in front of a pass's first one is the carried x - an integer-valued float: it is always (float) of a detector
        //  output - ; in place x[at - 1] already holds the previous pass's PCM)
        int32_t prev = i == 0 ? (int32_t)lds.x_carry : before[k];
        uint32_t *dst = &lds.row[(i / DC_S) * DC_ROW + (i % DC_S) / 2];
#pragma unroll
        for (int q = 0; q < PER; q += 2) {
            const int32_t d0 = xs[q] - prev, d1 = xs[q + 1] - xs[q];
            prev = xs[q + 1];
            dst[q / 2] = ((uint32_t)d0 & 0xffffu) | ((uint32_t)d1 << 16);
        }
        if (i + PER == nseg * DC_S) lds.x_last = (float)xs[PER - 1];
    }
    if (first_fill && lane < DC_GUESS) lds.z[lane] = lane == DC_GUESS - 1 ? lds.y_carry : 0.f;
}

IQD_DEV void dc_guess(const Consts &c, DcLds &lds, int nseg, int lane)
{
    if (lane >= nseg) return;
    const float cc = -c.dc_a1;
    const uint32_t *src = &lds.row[lane * DC_ROW];
    float z = 0.f;
    for (int i = 0; i < DC_S / 2; i++) {
        z = __builtin_fmaf(cc, z, dc_lo(src[i]));
        z = __builtin_fmaf(cc, z, dc_hi(src[i]));
    }
    lds.z[DC_GUESS + lane] = z;
}

IQD_DEV void dc_warm(const Consts &c, DcLds &lds, int nseg, int lane)
{
    if (lane >= nseg) return;
    if (lane == 0) { lds.g[0] = lds.y_carry; return; }
    const int m = lane > DC_WSEG ? lane - DC_WSEG : 0;       // first segment of the warm-up
    float y = lds.y_carry;
    if (m > 0) {                                             // state entering segment m, to float rounding
        const float a = c.dc_cseg;
        const float *zz = &lds.z[DC_GUESS + m - 1];
        y = zz[-(DC_GUESS - 1)];
        for (int j = DC_GUESS - 2; j >= 0; j--) y = __builtin_fmaf(a, y, zz[-j]);
    }
    const float a1 = c.dc_a1;
    for (int sgm = m; sgm < lane; sgm++) {
        const uint32_t *src = &lds.row[sgm * DC_ROW];
        for (int i = 0; i < DC_S / 2; i++) {
            float r = a1 * y;
            y = dc_lo(src[i]) - r;
            r = a1 * y;
            y = dc_hi(src[i]) - r;
        }
    }
    lds.g[lane] = y;
}

// the lane's own segment for real: its differences out of its row, its PCM pairs into the same words
IQD_DEV void dc_real(const Consts &c, DcLds &lds, int nseg, int lane, float gain)
{
    if (lane >= nseg) return;
    uint32_t *row = &lds.row[lane * DC_ROW];
    float y = lds.g[lane];
    const float a1 = c.dc_a1;
    for (int i = 0; i < DC_S / 2; i++) {
        const uint32_t d = row[i];
        float r = a1 * y;
        y = dc_lo(d) - r;
        const uint32_t lo = (uint32_t)cast_i16(gain * y);
        r = a1 * y;
        y = dc_hi(d) - r;
        row[i] = pack_lo16(lo, (uint32_t)cast_i16(gain * y));
    }
    lds.e[lane] = y;
}

// (As for the de-emphasis, two states below 2^-100 count as agreeing when |gain| <= 1e6: the next nonzero input
// of this recurrence is an integer difference, which absorbs them, and gain * y stays far below one PCM step.
// A constant detector output - an unmodulated carrier - leaves the true state stuck at a denormal.)
IQD_DEV bool dc_check(DcLds &lds, int nseg, int lane, bool tiny_ok = false)
{
    if (lane == 0 || lane >= nseg) return true;
    const float want = lds.e[lane - 1];
    if (iir_states_agree(want, lds.g[lane], tiny_ok)) return true;
    lds.g[lane] = want;
    return false;
}

IQD_DEV void dc_store(const DcLds &lds, int nseg, int lane, int16_t *pcm /* of the pass, 4-byte aligned */)
{
    uint32_t *dst = (uint32_t *)pcm;
    for (int d = lane; d < nseg * (DC_S / 2); d += 64) {
        const uint32_t w = lds.row[(d / (DC_S / 2)) * DC_ROW + (d % (DC_S / 2))];
#if !defined(IQD_NO_WT_STORES) && IQD_ON_DEVICE
        // write-through (sc1): what a launch leaves dirty in the L2s is written back at its end, in front of whatever follows it on
        // the stream (B / 6 TB/s: MI355X_MICROARCH.md, kernel boundaries) - the closing launch's PCM and tails go out while it still
        // runs instead (round 6, profiles/r6_wt_ab.txt: AM 4096 x 2^16 -0.8 %, 4096 x 2^14 -1.7 %, configs[4] -0.9 %; the pipelines'
        // own 8-byte stores the same way: +0.5 to +4 %, narrow write-throughs - not taken)
        __hip_atomic_store(dst + d, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        dst[d] = w;
#endif
    }
}

// One channel, n PCM samples (multiple of 4; the last pass may be partial, its last segment too).  x may be the PCM row itself
// (SRC = int16_t, in place).
template <class Exec, class SRC>
IQD_DEV void dc_block_wave(Exec &ex, const Consts &c, DcLds &lds, const SRC *x, int n, float gain,
                           DcCarry &st, int16_t *pcm)
{
#if defined(IQD_DC_TIMING) && IQD_ON_DEVICE   // measurement build (tools/variant.sh dctiming iqd_kernels.hip -DIQD_DC_TIMING=1, tools/r6/r6_dctiming.sh):
    long long tm[8];                           // where a row's pass goes, in shader clocks; profiles/r6_dc_timing.txt
    const long long wall0 = wall_clock64();
#define DC_T(K) tm[K] = clock64()
    int redos = 0;
#else
#define DC_T(K)
#endif
    DC_T(0);
    ex.wave0([&](int lane) { if (lane == 0) { lds.x_carry = st.x_prev; lds.y_carry = st.y_prev; } });
    for (int base = 0; base < n; base += DC_SUPER) {
        const int len = n - base < DC_SUPER ? n - base : DC_SUPER;
        const int nfull = len / DC_S;             // whole segments: the segmented scheme
        if (nfull > 0) {
            ex.wave0([&](int lane) { dc_fill(lds, x, base, nfull, lane, true); });
            DC_T(1);
            ex.wave0([&](int lane) { dc_guess(c, lds, nfull, lane); });
            DC_T(2);
            ex.wave0([&](int lane) { dc_warm(c, lds, nfull, lane); });
            DC_T(3);
            for (bool again = false;; again = true) {
#if defined(IQD_DC_TIMING) && IQD_ON_DEVICE
                redos += again;
#endif
                if (again) ex.wave0([&](int lane) { dc_fill(lds, x, base, nfull, lane, false); });   // (the rows hold PCM by now)
                ex.wave0([&](int lane) { dc_real(c, lds, nfull, lane, gain); });
                if (ex.wave0_all([&](int lane) { return dc_check(lds, nfull, lane, __builtin_fabsf(gain) <= 1e6f); })) break;
            }
            DC_T(4);
            ex.wave0([&](int lane) {
                if (pcm) dc_store(lds, nfull, lane, pcm + base);
                if (lane == 0) {
                    lds.y_carry = lds.e[nfull - 1];
                    lds.x_carry = lds.x_last;
                }
            });
        }
        const int rest = len - nfull * DC_S;      // a tail shorter than a segment: one lane, serial
        if (rest > 0) {
            ex.wave0([&](int lane) {
                if (lane != 0) return;
                DcCarry t2{lds.x_carry, lds.y_carry};
                dc_block_run(x + base + nfull * DC_S, rest, gain, c.dc_a1, t2, pcm ? pcm + base + nfull * DC_S : nullptr);
                lds.x_carry = t2.x_prev;
                lds.y_carry = t2.y_prev;
            });
        }
    }
    st.x_prev = lds.x_carry;
    st.y_prev = lds.y_carry;
#if defined(IQD_DC_TIMING) && IQD_ON_DEVICE
    DC_T(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DC_T(6);
    if ((threadIdx.x & 63) == 0 && (blockIdx.x % 331) == 7 && n == DC_SUPER)
        printf("dc row wg %u: wall start %lld (x10 ns) | fill %lld guess %lld warm %lld real+check %lld (redos %d) store-issue %lld store-done %lld | total %lld clocks, wall %lld\n",
               blockIdx.x, wall0, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], redos, tm[5] - tm[4], tm[6] - tm[5], tm[6] - tm[0],
               wall_clock64() - wall0);
#endif
#undef DC_T
}

// ---- long streams, many waves per channel -------------------------------------------------------------
// A channel's 8 kS/s stream is cut into tiles of DC_TILE samples, one wave each.  Tile 0 starts from the exact
// carried state; every other tile first runs the DC_WARM samples in front of it from the zero state - two
// trajectories of this contraction (pole 0.95) become bit-identical after a few hundred steps - and then its
// own samples.  It records the state it started from and the state it ended in; the channel is accepted
// only if every tile's start state agrees with its predecessor's end state (dc_chainup_kernel), which by
// induction from tile 0 makes every tile the serial result.  Otherwise (e.g. a decaying tail that has not reached zero)
// the one-wave pass above redoes the channel from the carried state.
constexpr int DC_TILE = 8192;
constexpr int DC_WARM = 2048;
struct DcRecord { float y_start, y_end, x_end; uint32_t pad; };

template <class Exec>
IQD_DEV void dc_tile(Exec &ex, const Consts &c, DcLds &lds, const int32_t *x, int n, int tile, float gain,
                     const DcCarry &carried, int16_t *pcm, DcRecord &rec)
{
    const int start = tile * DC_TILE;
    const int len = n - start < DC_TILE ? n - start : DC_TILE;
    DcCarry st = carried;
    if (tile > 0) {
        st.x_prev = (float)x[start - DC_WARM - 1];   // exact; DC_WARM < DC_TILE, so the index is >= 0
        st.y_prev = 0.f;
        dc_block_wave(ex, c, lds, x + start - DC_WARM, DC_WARM, gain, st, nullptr);
    }
    rec.y_start = st.y_prev;
    dc_block_wave(ex, c, lds, x + start, len, gain, st, pcm + start);
    rec.y_end = st.y_prev;
    rec.x_end = st.x_prev;
    rec.pad = 0;
}

// ---- per-block control loops carried by the squelch pass (host + device, like the chains) -----------------
// AutomaticGainControl::run (AutomaticGainControl.cc:663-741) with runLowpass (:743-889) / runHarris
// (:935-1062): one block magnitude in, the receiver's IF gain out.  binary32 arithmetic in the reference's order.
// convertMagnitudeToDbFs for the AGC and the squelch (DbfsCalculator.cc:111-147 with full scale 127)
IQD_DEV int32_t magnitude_dbfs(const Consts &c, uint32_t magnitude)
{
    const uint32_t m = magnitude > 127u ? 127u : magnitude;   // :122-125
    return c.db_table[m] - 42;
}

IQD_DEV uint32_t agc_run_dbfs(const AgcConfig &cfg, AgcState &st, uint32_t magnitude, int32_t signal, uint32_t gain);

IQD_DEV uint32_t agc_run(const Consts &c, const AgcConfig &cfg, AgcState &st, uint32_t magnitude, uint32_t gain)
{
    return agc_run_dbfs(cfg, st, magnitude, magnitude_dbfs(c, magnitude), gain);
}

// the same with the magnitude's dBFS value already looked up (the wave kernel does that for 64 blocks at once).
// Written with selects instead of branches: the long-row kernel runs it once per block on wave-uniform values, where
// every branch costs more than the arithmetic it skips.
IQD_DEV uint32_t agc_run_dbfs(const AgcConfig &cfg, AgcState &st, uint32_t magnitude, int32_t signal, uint32_t gain)
{
    st.if_gain = gain;   // run(): follow the operator's manual changes (:682-688)
    // blank the measurements that follow an adjustment (:690-707)
    const bool adjusted = st.adjusted != 0;
    const bool blank = adjusted && st.blank_ctr < cfg.blanking_limit;
    st.blank_ctr = blank ? st.blank_ctr + 1 : (adjusted ? 0u : st.blank_ctr);
    const bool allowed = !blank;
    // runLowpass / runHarris on the side, committed only if allowed
    int32_t error = cfg.operating_point - signal;
    const bool at_max = st.if_gain == AGC_MAX_GAIN, at_min = st.if_gain == 0;
    error = (at_max && error > 0) ? 0 : error;
    error = (!at_max && at_min && error < 0) ? 0 : error;
    error = ((error < 0 ? -error : error) <= cfg.deadband) ? 0 : error;
    const int32_t target = (int32_t)(st.if_gain + (uint32_t)error);
    const float lowpass = (cfg.alpha * (float)target) + ((1 - cfg.alpha) * st.filtered);
    const float harris = st.filtered + (cfg.alpha * (float)error);
    float filtered = cfg.type == 0 ? lowpass : harris;
    filtered = filtered > (float)AGC_MAX_GAIN ? (float)AGC_MAX_GAIN : (filtered < 0 ? 0.f : filtered);
    const uint32_t if_gain = (uint32_t)filtered;
    const bool changed = allowed && error != 0;
    st.signal_magnitude = allowed ? magnitude : st.signal_magnitude;
    st.normalized = allowed ? (int32_t)((uint32_t)signal - st.if_gain) : st.normalized;
    st.filtered = allowed ? filtered : st.filtered;
    st.if_gain = allowed ? if_gain : st.if_gain;
    st.adjusted = changed ? 1u : (blank ? 1u : 0u);
    return changed ? if_gain : gain;   // Radio::setReceiveIfGainInDb(0, ifGainInDb), Radio.cc:851-861
}

// FrequencyScanner::run on a rejected block (signalStateCallback -> run, FrequencyScanner.cc:47-62, :378-404)
IQD_DEV void scanner_step(const ScanConfig &sc, ScanState &ss)
{
    ss.current_hz = ss.current_hz + sc.increment_hz;
    if (ss.current_hz > sc.end_hz) ss.current_hz = sc.start_hz;
    ss.tune_count++;
}

}  // namespace iqd
