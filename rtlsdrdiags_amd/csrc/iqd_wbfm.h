// WBFM chain (WbFmDemodulator.cc:383-562 behind IqDataProcessor.cc:735-749), written as
// per-thread phase functions.  A workgroup of 256 threads owns one tile (a run of consecutive
// samples of one channel) and walks it chunk by chunk; LDS carries every filter history from
// chunk to chunk, so inside a tile the arithmetic is the reference's, sample for sample.
//
//   phase1   raw u8 -> s8 -> +-Fs/4 rotation -> 16-tap Q15 FIR on I and Q (v_dot4, taps split
//            into low/high bytes; |acc| < 2^23 so the per-MAC clamp can never fire)
//            -> int8 wrap -> 256x256 atan2 table -> delta-theta, branch cut -> u = b0*(K*d)
//   iir_*    75 us de-emphasis, y[n] = (u[n] + u[n-1]) - a1*y[n-1] evaluated op by op in
//            binary32.  The recurrence is serial, so 64 lanes run 64 segments of 128 samples:
//            each lane first runs the previous segment from a guessed state (a float
//            approximation), and the result is accepted only when every lane's warmed-up
//            state equals its left neighbour's exact end state BIT FOR BIT; otherwise the
//            mismatching lanes restart from the neighbour's state until it does
//            (by induction from lane 0, whose start is the carried exact state).
//   stage1-3 (int16)y -> /4 (8 taps) -> /4 (12) -> /2 (40), Q15 with the reference's per-MAC
//            clamp order (Decimator_int16.cc:176-238) -> PCM
//
// The same functions are compiled for the GPU (iqd_kernels.hip) and, for tests only, for the
// host by tests/emu (IQD_HOST_EMU), where a loop over thread ids replaces the SIMT machine.
#pragma once
#include "iqd_device.h"
#include "iqd_prims.h"

namespace iqd {

constexpr int WB_THREADS = 256;
constexpr int WB_BIAS = 16384 + (128 << 15);  // Q15 rounding term + 128 for the table index
constexpr int TGRAN = TSTRIDE / 4;            // 16-byte granules per segment of the IIR input

// Granule gi (0..31) of a segment sits at gi ^ ((gi >> 3) & 3): phase 1 writes 16 consecutive
// samples per lane (8 lanes -> 8 different bank groups) and the IIR lanes, one segment each,
// stay conflict-free because the segment stride is 33 granules.
IQD_DEV int t_slot(int seg, int gi) { return seg * TGRAN + (gi ^ ((gi >> 3) & 3)); }

struct WbfmLds {
    u32x4 t4[WBFM_NSEG * TGRAN];           // u[n] = b0 * (K * dtheta[n]) as float bits: per segment
                                           // 32 granules of 4 samples (XOR-swizzled) + 1 pad
    alignas(16) uint32_t w[WBFM_NSEG * WSTRIDE];       // (int16)y[n], two per dword, segment-strided
    alignas(16) uint32_t y1[(8 + WBFM_CHUNK / 4) / 2];   // stage-1 output with 8 samples of history
    alignas(16) uint32_t y2[(40 + WBFM_CHUNK / 16) / 2]; // stage-2 output with 40 samples of history
    uint32_t whist[2];                     // the 4 w samples before the chunk
    float z[WBFM_NSEG + 4];                // zero-state segment responses; z[3] = carried y
    float g[WBFM_NSEG];                    // state entering each segment
    float e[WBFM_NSEG];                    // state leaving each segment
    float y_carry, u_carry;                // state entering the chunk
    uint32_t mag[WBFM_CHUNK / SEG + 2];    // squelch magnitude partial sums per block slot
    uint32_t repair_count;
};

// Per 16-sample group: sum c^(15-k) u[k], input of the IIR state guess.  Lives in the data part of
// y1 (dead between the end of a chunk's decimation and the next chunk's stage 1).
IQD_DEV float *lds_part(WbfmLds &lds) { return (float *)&lds.y1[4]; }
static_assert(WBFM_NSEG * 8 * 4 <= (WBFM_CHUNK / 4 / 2) * 4, "part[] must fit in y1's data region");

struct WbfmTile {
    const uint8_t *iq_ch;        // the channel's input row (virtual sample 0)
    const uint8_t *tail;         // the channel's WBFM tail: virtual samples [-TAIL, 0)
    const uint32_t *blk_list;    // squelch-gated runs: open-block indices of this channel
    uint32_t block_samples;
    uint32_t block_magic;        // ceil(2^32 / block_samples)
    int64_t v0;                  // tile start, virtual samples
    int32_t tlen;                // tile length
    uint32_t sel_i, sel_q, neg_i, neg_q;  // rotation as byte selectors / negate masks
    float k;                     // (gain / 75000) * 32767
    uint32_t bounded;            // |k| * pi * 1.01 < 2^31: (int16) casts cannot hit the indefinite value
    const float *lut;            // atan2 table, lut[y * 256 + x]
    int16_t *pcm_row;            // PCM of virtual sample 0
    uint32_t *mag_row;           // per-block magnitude sums of this channel
};

// Address of 32 raw bytes (16 samples) starting at virtual sample v (multiple of 16).
template <bool GATED>
IQD_DEV const u32x4 *raw_group(const WbfmTile &t, int64_t v)
{
    if (v < 0) return (const u32x4 *)(t.tail + (int64_t)TAIL_BYTES + 2 * v);
    if (!GATED) return (const u32x4 *)(t.iq_ch + 2 * v);
    uint32_t blk = (uint32_t)(v / t.block_samples);
    uint32_t off = (uint32_t)(v - (int64_t)blk * t.block_samples);
    return (const u32x4 *)(t.iq_ch + ((int64_t)t.blk_list[blk] * t.block_samples + off) * 2);
}

// Rotation of 4 samples held in two dwords (I0 Q0 I1 Q1 | I2 Q2 I3 Q3, signed bytes) into
// one dword of I' and one of Q' (IqDataProcessor.cc:567-611).
IQD_DEV void rotate4(const WbfmTile &t, uint32_t w0, uint32_t w1, uint32_t &xi, uint32_t &xq)
{
    xi = neg_bytes(perm(w1, w0, t.sel_i), t.neg_i);
    xq = neg_bytes(perm(w1, w0, t.sel_q), t.neg_q);
}

// 17 outputs (samples -1 .. 15 of the group) of the 16-tap FIR over 32 bytes of one rail.
// Output idx uses window bytes idx .. idx+15; the result carries WB_BIAS.
IQD_DEV void fir16_window(const uint32_t (&x)[8], const Consts &c, int (&acc)[17])
{
    uint32_t y[3][7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        y[0][j] = alignbyte(x[j + 1], x[j], 1);
        y[1][j] = alignbyte(x[j + 1], x[j], 2);
        y[2][j] = alignbyte(x[j + 1], x[j], 3);
    }
#pragma unroll
    for (int idx = 0; idx < 17; idx++) {
        const int s = idx & 3, j0 = idx >> 2;
        int lo = WB_BIAS, hi = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = (s == 0) ? x[j0 + q] : y[s - 1][j0 + q];
            lo = dot4(d, c.pre_lo[q], lo);
            hi = dot4(d, c.pre_hi[q], hi);
        }
        acc[idx] = lo + (int)((uint32_t)hi << 8);
    }
}

// max(|I|,|Q|) + min(|I|,|Q|)/2 summed over the 2 samples of a dword of signed bytes
// (SignalDetector.cc:227-247; the value does not depend on the rotation).
IQD_DEV uint32_t magnitude2(uint32_t s)
{
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int i = (int)(int8_t)(s >> (16 * k)), q = (int)(int8_t)(s >> (16 * k + 8));
        uint32_t a = (uint32_t)(i < 0 ? -i : i), b = (uint32_t)(q < 0 ? -q : q);
        sum += (a > b) ? a + (b >> 1) : b + (a >> 1);
    }
    return sum;
}

IQD_DEV uint32_t div_block(const WbfmTile &t, uint32_t x)  // x / block_samples, x < 65536
{
    return (uint32_t)(((uint64_t)x * t.block_magic) >> 32);
}

// Where a chunk sits relative to the squelch blocks (uniform per chunk).
struct ChunkBlocks { uint32_t base_blk, in_blk; };

IQD_DEV ChunkBlocks chunk_blocks(const WbfmTile &t, int cstart)
{
    ChunkBlocks cb;
    const int64_t v = t.v0 + (cstart < 0 ? 0 : cstart);
    cb.base_blk = (uint32_t)(v / t.block_samples);
    cb.in_blk = (uint32_t)(v - (int64_t)cb.base_blk * t.block_samples);
    return cb;
}

template <bool GATED, bool MAG>
IQD_DEV void wbfm_phase1(const WbfmTile &t, const Consts &c, WbfmLds &lds, const ChunkBlocks &cb,
                         int cstart, int clen, int tid)
{
    const int ngroups = clen >> 4;
    for (int g = tid; g < ngroups; g += WB_THREADS) {
        const int n0 = cstart + 16 * g;
        const int64_t v = t.v0 + n0;
        const u32x4 *ph = raw_group<GATED>(t, v - 16);
        const u32x4 *po = raw_group<GATED>(t, v);
        const u32x4 r0 = ph[0], r1 = ph[1], r2 = po[0], r3 = po[1];
        uint32_t s[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w,
                          r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
        uint32_t xi[8], xq[8];
#pragma unroll
        for (int j = 0; j < 16; j++) s[j] ^= 0x80808080u;  // offset binary -> signed (:735-738)
#pragma unroll
        for (int j = 0; j < 8; j++) rotate4(t, s[2 * j], s[2 * j + 1], xi[j], xq[j]);

        int ai[17], aq[17];
        fir16_window(xi, c, ai);
        fir16_window(xq, c, aq);

        float th[17];
#pragma unroll
        for (int k = 0; k < 17; k++) {
            // byte offset of lut[(uint8)(Q'+128)][(uint8)(I'+128)]: bits 15..22 of each accumulator
            const uint32_t off = (((uint32_t)aq[k] >> 5) & 0x3fc00u) | (((uint32_t)ai[k] >> 13) & 0x3fcu);
            th[k] = *(const float *)((const char *)t.lut + off);
        }
        const int p = n0 - cstart;  // position inside the chunk
        const int seg = p >> 7, gq = (p & 127) >> 4;
        float u[16];
        const float cc = c.deemph_c;
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float d = th[k + 1] - th[k];
            d = wrap_delta(d);
            const float v1 = t.k * d;
            u[k] = c.deemph_b0 * v1;
            part = __builtin_fmaf(cc, part, u[k]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            lds.t4[t_slot(seg, 4 * gq + q)] =
                u32x4{f2u(u[4 * q]), f2u(u[4 * q + 1]), f2u(u[4 * q + 2]), f2u(u[4 * q + 3])};
        lds_part(lds)[8 * seg + gq] = part;
        if (MAG && cstart >= 0) {
            uint32_t m = 0;
#pragma unroll
            for (int j = 8; j < 16; j++) m += magnitude2(s[j]);
            // block slot relative to the block holding the chunk's first sample
            const uint32_t slot = div_block(t, cb.in_blk + (uint32_t)p);
#if IQD_ON_DEVICE
            atomicAdd(&lds.mag[slot], m);
#else
            lds.mag[slot] += m;
#endif
        }
    }
}

// Flushes the chunk's magnitude partial sums (call after a barrier; then barrier again).
IQD_DEV void wbfm_flush_mag(const WbfmTile &t, WbfmLds &lds, const ChunkBlocks &cb,
                            int cstart, int clen, int tid)
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

// ---- de-emphasis IIR -----------------------------------------------------------------------
IQD_DEV float t_last(const WbfmLds &lds, int seg) { return u2f(lds.t4[t_slot(seg, 31)].w); }

IQD_DEV float iir_u_before(const WbfmLds &lds, int seg)  // u[n-1] at the start of a segment
{
    return seg == 0 ? lds.u_carry : t_last(lds, seg - 1);
}

// lane j: approximate zero-state response of segment j to t[n] = u[n] + u[n-1], from the
// per-group partial sums phase 1 left (plain float arithmetic: this is only the state GUESS).
//   S = sum c^(127-i) u[i];  response = S + c^127 u[-1] + (S - u[127]) / c
IQD_DEV void iir_guess(const Consts &c, WbfmLds &lds, int nseg, int lane)
{
    if (lane == 0) lds.z[3] = lds.y_carry, lds.z[2] = 0.f, lds.z[1] = 0.f, lds.z[0] = 0.f;
    if (lane >= nseg) return;
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) sum = __builtin_fmaf(c.deemph_c16, sum, lds_part(lds)[8 * lane + g]);
    const float z = sum + c.deemph_c127 * iir_u_before(lds, lane) + (sum - t_last(lds, lane)) * c.deemph_cinv;
    lds.z[4 + lane] = z;
}

// One de-emphasis step, exactly as IirFilter::filterData evaluates it (IirFilter.cc:161-176):
// y = (b0 x[n] + b1 x[n-1]) - (a1 y[n-1]), every operation rounded to binary32.
#define IQD_IIR_STEP(U)            \
    {                              \
        const float tn_ = (U) + up; \
        const float r_ = a1 * y;   \
        y = tn_ - r_;              \
        up = (U);                  \
    }

// lane j >= 1: run segment j-1 from the guessed state to get the state entering segment j.
IQD_DEV void iir_warm(const Consts &c, WbfmLds &lds, int nseg, int lane)
{
    if (lane >= nseg) return;
    if (lane == 0) { lds.g[0] = lds.y_carry; return; }
    const float a = c.deemph_c128;
    // approx y at the end of segment lane-2:  z[l-2] + A z[l-3] + A^2 z[l-4] + A^3 z[l-5]
    const float *zz = &lds.z[4 + lane - 2];
    float y = zz[0] + a * (zz[-1] + a * (zz[-2] + a * zz[-3]));
    if (lane == 1) y = lds.y_carry;  // exact
    float up = iir_u_before(lds, lane - 1);
    const float a1 = c.deemph_a1;
#pragma unroll 2
    for (int gi = 0; gi < 32; gi += 2) {
        const u32x4 a4 = lds.t4[t_slot(lane - 1, gi)], b4 = lds.t4[t_slot(lane - 1, gi + 1)];
        IQD_IIR_STEP(u2f(a4.x)) IQD_IIR_STEP(u2f(a4.y)) IQD_IIR_STEP(u2f(a4.z)) IQD_IIR_STEP(u2f(a4.w))
        IQD_IIR_STEP(u2f(b4.x)) IQD_IIR_STEP(u2f(b4.y)) IQD_IIR_STEP(u2f(b4.z)) IQD_IIR_STEP(u2f(b4.w))
    }
    lds.g[lane] = y;
}

// lane j: the real pass over segment j from g[j]; writes (int16)y and e[j].
// bounded: the host proved |y| < 2^31 for this launch (|K| pi * 1.01 < 2^31), so the cast needs
// no "integer indefinite" handling.
IQD_DEV void iir_real(const Consts &c, WbfmLds &lds, int nseg, int lane, bool bounded)
{
    if (lane >= nseg) return;
    u32x4 *dst = (u32x4 *)&lds.w[lane * WSTRIDE];
    float up = iir_u_before(lds, lane), y = lds.g[lane];
    const float a1 = c.deemph_a1;
    if (bounded) {
#pragma unroll 2
        for (int gi = 0; gi < 32; gi += 2) {
            const u32x4 a4 = lds.t4[t_slot(lane, gi)], b4 = lds.t4[t_slot(lane, gi + 1)];
            uint32_t w[8];
            IQD_IIR_STEP(u2f(a4.x)) w[0] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(a4.y)) w[1] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(a4.z)) w[2] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(a4.w)) w[3] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(b4.x)) w[4] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(b4.y)) w[5] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(b4.z)) w[6] = cast_i16_bounded(y);
            IQD_IIR_STEP(u2f(b4.w)) w[7] = cast_i16_bounded(y);
            dst[gi >> 1] = u32x4{pack_lo16(w[0], w[1]), pack_lo16(w[2], w[3]),
                                 pack_lo16(w[4], w[5]), pack_lo16(w[6], w[7])};
        }
        lds.e[lane] = y;
        return;
    }
#pragma unroll 2
    for (int gi = 0; gi < 32; gi += 2) {
        const u32x4 a4 = lds.t4[t_slot(lane, gi)], b4 = lds.t4[t_slot(lane, gi + 1)];
        uint32_t w[8];
        IQD_IIR_STEP(u2f(a4.x)) w[0] = (uint32_t)cast_i16(y) & 0xffffu;
        IQD_IIR_STEP(u2f(a4.y)) w[1] = (uint32_t)cast_i16(y) << 16;
        IQD_IIR_STEP(u2f(a4.z)) w[2] = (uint32_t)cast_i16(y) & 0xffffu;
        IQD_IIR_STEP(u2f(a4.w)) w[3] = (uint32_t)cast_i16(y) << 16;
        IQD_IIR_STEP(u2f(b4.x)) w[4] = (uint32_t)cast_i16(y) & 0xffffu;
        IQD_IIR_STEP(u2f(b4.y)) w[5] = (uint32_t)cast_i16(y) << 16;
        IQD_IIR_STEP(u2f(b4.z)) w[6] = (uint32_t)cast_i16(y) & 0xffffu;
        IQD_IIR_STEP(u2f(b4.w)) w[7] = (uint32_t)cast_i16(y) << 16;
        dst[gi >> 1] = u32x4{w[0] | w[1], w[2] | w[3], w[4] | w[5], w[6] | w[7]};
    }
    lds.e[lane] = y;
}

// lane j >= 1: true when the state it started from is its neighbour's exact end state.
// On a mismatch the lane adopts the neighbour's state for the next iir_real().
IQD_DEV bool iir_check(WbfmLds &lds, int nseg, int lane)
{
    if (lane == 0 || lane >= nseg) return true;
    const float want = lds.e[lane - 1];
    if (f2u(want) == f2u(lds.g[lane])) return true;
    lds.g[lane] = want;
    return false;
}

// ---- decimators ----------------------------------------------------------------------------
IQD_DEV u32x2 w_group(const WbfmLds &lds, int g)  // 4 consecutive w samples, group index g
{
    if (g < 0) return u32x2{lds.whist[0], lds.whist[1]};
    const uint32_t *p = &lds.w[(g >> 5) * WSTRIDE + 2 * (g & 31)];
    return u32x2{p[0], p[1]};
}

IQD_DEV void put_i16(uint32_t *buf, int idx, int v)  // buf as int16[]; exclusive owner of idx
{
    int16_t *p = (int16_t *)buf;
    p[idx] = (int16_t)v;
}

IQD_DEV int get_i16(const uint32_t *buf, int idx)
{
    const int16_t *p = (const int16_t *)buf;
    return p[idx];
}

// /4, 8 taps (WbFmDemodulator.cc:535-537).  sum|hq| = 29126 < 2^15: no clamp can fire,
// so the sum is formed with v_dot2 in any order.
IQD_DEV void wbfm_stage1(const Consts &c, WbfmLds &lds, int clen, int tid)
{
    const int nout = clen >> 2;
    for (int m = tid; m < nout; m += WB_THREADS) {
        const u32x2 a = w_group(lds, m - 1), b = w_group(lds, m);
        // window ascending x[4m-4 .. 4m+3] <-> taps h[7 .. 0]
        int acc = 1 << 14;
        acc = dot2(a.x, ((uint32_t)(uint16_t)c.wbfm_d1[7]) | ((uint32_t)(uint16_t)c.wbfm_d1[6] << 16), acc);
        acc = dot2(a.y, ((uint32_t)(uint16_t)c.wbfm_d1[5]) | ((uint32_t)(uint16_t)c.wbfm_d1[4] << 16), acc);
        acc = dot2(b.x, ((uint32_t)(uint16_t)c.wbfm_d1[3]) | ((uint32_t)(uint16_t)c.wbfm_d1[2] << 16), acc);
        acc = dot2(b.y, ((uint32_t)(uint16_t)c.wbfm_d1[1]) | ((uint32_t)(uint16_t)c.wbfm_d1[0] << 16), acc);
        put_i16(lds.y1, 8 + m, acc >> 15);
    }
}

// Sequential Q15 dot product with the per-MAC clamp, newest sample first.
// buf is an int16 array; `newest` is the index of x[n].
template <int L>
IQD_DEV int q15_seq(const int16_t *h, const uint32_t *buf, int newest)
{
    int acc = 1 << 14;
#pragma unroll
    for (int k = 0; k < L; k++) {
        const int idx = newest - k;
        const uint32_t pair = buf[idx >> 1];
        const uint32_t tap = (idx & 1) ? ((uint32_t)(uint16_t)h[k] << 16) : (uint32_t)(uint16_t)h[k];
        acc = clamp_q30(dot2(pair, tap, acc));
    }
    return acc >> 15;
}

// /4, 12 taps (WbFmDemodulator.cc:541)
IQD_DEV void wbfm_stage2(const Consts &c, WbfmLds &lds, int clen, int tid)
{
    const int nout = clen >> 4;
    for (int j = tid; j < nout; j += WB_THREADS)
        put_i16(lds.y2, 40 + j, q15_seq<12>(c.post12, lds.y1, 8 + 4 * j + 3));
}

// /2, 40 taps (WbFmDemodulator.cc:546) -> PCM
IQD_DEV void wbfm_stage3(const Consts &c, WbfmLds &lds, const WbfmTile &t, int cstart, int clen, int tid)
{
    const int nout = clen >> 5;
    for (int i = tid; i < nout; i += WB_THREADS) {
        const int y = q15_seq<40>(c.audio40, lds.y2, 40 + 2 * i + 1);
        if (cstart >= 0) t.pcm_row[((t.v0 + cstart) >> 5) + i] = (int16_t)y;
    }
}

// Histories for the next chunk (call after a barrier that follows stage3).
IQD_DEV void wbfm_shift_history(WbfmLds &lds, int clen, int tid)
{
    const int n1 = clen >> 2, n2 = clen >> 4;
    if (tid < 4) {  // 8 int16 = 4 dwords
        lds.y1[tid] = lds.y1[(n1 >> 1) + tid];
    } else if (tid >= 64 && tid < 64 + 20 && (n2 >> 1) >= 20) {  // 40 int16 = 20 dwords
        const int k = tid - 64;
        lds.y2[k] = lds.y2[(n2 >> 1) + k];
    } else if (tid == 96 && (n2 >> 1) < 20) {  // short chunk: ranges overlap, move in order
        for (int k = 0; k < 20; k++) lds.y2[k] = lds.y2[(n2 >> 1) + k];
    } else if (tid == 128) {
        const u32x2 last = w_group(lds, n1 - 1);
        lds.whist[0] = last.x;
        lds.whist[1] = last.y;
    }
}


// ---- tile driver ----------------------------------------------------------------------------
// Exec abstracts the SIMT machine: all(f) runs f(tid) for the 256 threads and then barriers;
// wave0(f) runs f(lane) on the first wave only, followed by a wave-level LDS fence;
// wave0_all(f) additionally and-reduces the lanes' results.
struct WbfmStart {
    float y, u;      // de-emphasis state at the restart point
    int32_t back;    // restart distance before the tile (multiple of SEG; 0 = at the tile start)
    int32_t cold;    // 1: no carried state, warm up from zero COLD_HALO samples back
};

struct WbfmRecord {  // what the tile reports for hand-off verification and the next restart
    float y_in;      // cold tiles: own y at the restart point FORCED_BACK before the tile
    float y_out, u_out;
    int32_t back_out;
    float y_end, u_end;  // state after the tile's last sample (for resetDemodulator())
    uint32_t pad[2];
};

template <bool GATED, bool MAG, class Exec>
IQD_DEV void wbfm_tile(Exec &ex, const WbfmTile &t, const Consts &c, WbfmLds &lds,
                       const WbfmStart &start, WbfmRecord *rec_out)
{
    const int halo = start.cold ? COLD_HALO : start.back;
    ex.all([&](int tid) {
        if (tid < 4) lds.y1[tid] = 0;
        if (tid < 20) lds.y2[tid] = 0;
        if (tid < 2) lds.whist[tid] = 0;
        if (tid < WBFM_CHUNK / SEG + 2) lds.mag[tid] = 0;
        if (tid == 0) {
            lds.y_carry = start.cold ? 0.f : start.y;
            lds.u_carry = start.cold ? 0.f : start.u;
            lds.repair_count = 0;
        }
    });
    // restart point for whoever continues this stream: FORCED_BACK before the end when the
    // tile (plus its exact lead-in) is long enough, else the tile's own restart point.
    int rec_pos = t.tlen - FORCED_BACK;
    if (rec_pos < -halo) rec_pos = -halo;
    WbfmRecord rec;
    rec.y_in = start.y;
    rec.y_out = start.y;
    rec.u_out = start.u;
    rec.back_out = t.tlen - rec_pos;

    for (int cstart = -halo; cstart < t.tlen;) {
        const int clen = cstart < 0 ? -cstart : (t.tlen - cstart < WBFM_CHUNK ? t.tlen - cstart : WBFM_CHUNK);
        const int nseg = clen / SEG;
        const ChunkBlocks cb = chunk_blocks(t, cstart);
        ex.stamp(7);
        ex.all([&](int tid) { wbfm_phase1<GATED, MAG>(t, c, lds, cb, cstart, clen, tid); });
        ex.stamp(0);
        if (ex.in_wave0()) {
            if (rec_pos == cstart) { rec.y_out = lds.y_carry; rec.u_out = lds.u_carry; }
            ex.wave0([&](int lane) { iir_guess(c, lds, nseg, lane); });
            ex.stamp(1);
            ex.wave0([&](int lane) { iir_warm(c, lds, nseg, lane); });
            ex.stamp(2);
            int rounds = 0;
            do {
                ex.wave0([&](int lane) { iir_real(c, lds, nseg, lane, t.bounded != 0); });
                rounds++;
            } while (!ex.wave0_all([&](int lane) { return iir_check(lds, nseg, lane); }));
            ex.stamp(3);
            if (rec_pos > cstart && rec_pos < cstart + clen) {
                const int seg = (rec_pos - cstart) / SEG - 1;
                rec.y_out = lds.e[seg];
                rec.u_out = t_last(lds, seg);
            }
            if (start.cold && cstart < 0) rec.y_in = lds.e[(COLD_HALO - FORCED_BACK) / SEG - 1];
            ex.wave0([&](int lane) {
                if (lane == 0) {
                    lds.y_carry = lds.e[nseg - 1];
                    lds.u_carry = t_last(lds, nseg - 1);
                    lds.repair_count += (uint32_t)(rounds - 1);
                }
            });
        }
        ex.sync();
        ex.stamp(4);
        ex.all([&](int tid) {
            if (MAG) wbfm_flush_mag(t, lds, cb, cstart, clen, tid);
            wbfm_stage1(c, lds, clen, tid);
        });
        ex.stamp(5);
        ex.all([&](int tid) { wbfm_stage2(c, lds, clen, tid); });
        ex.all([&](int tid) { wbfm_stage3(c, lds, t, cstart, clen, tid); });
        ex.all([&](int tid) { wbfm_shift_history(lds, clen, tid); });
        ex.stamp(6);
        cstart += clen;
    }
    if (ex.in_wave0() && rec_out) {
        rec.y_end = lds.y_carry;
        rec.u_end = lds.u_carry;
        rec.pad[0] = rec.pad[1] = 0;
        ex.wave0([&](int lane) { if (lane == 0) *rec_out = rec; });
    }
}

}  // namespace iqd
