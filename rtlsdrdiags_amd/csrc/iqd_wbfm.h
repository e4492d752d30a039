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

#ifndef IQD_WB_THREADS
#define IQD_WB_THREADS 256
#endif
constexpr int WB_THREADS = IQD_WB_THREADS;   // threads per workgroup of the tile kernels
constexpr int WB_BIAS = 2 * (16384 + (128 << 15));  // doubled: Q15 rounding term + 128 for the table index
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
    uint32_t whist[4];                     // the 4 w samples before the chunk, two alternating slots
    float z[WBFM_NSEG + 4];                // zero-state segment responses; z[3] = carried y
    float g[WBFM_NSEG];                    // state entering each segment
    float e[WBFM_NSEG];                    // state leaving each segment
    float y_carry, u_carry;                // state entering the chunk
    uint32_t mag[WBFM_CHUNK / SEG + 2];    // squelch magnitude partial sums per block slot
    uint32_t repair_count;
    uint32_t sync_ctr;                     // arrivals at the waves-1-3 rendezvous (monotonic)
    float part[WBFM_CHUNK / 16];           // per 16-sample group: sum c^(15-k) u[k], input of the IIR state guess
    uint32_t y2_peak, y2_peak_hist;        // max |y2| of this chunk (if loud) / reaching into the next
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
    const GainEpochList *epochs; // gain changes still inside the tail (tile 0 of a call only; else nullptr):
                                 // samples before position -since[i] (call-relative) ran with k_before[i]
    float k_min;                 // smallest K among k and the k_before in reach (the tiny-state rule wants all >= 1)
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
// Output idx uses window bytes idx .. idx+15.  The result is TWICE the Q15 accumulator plus WB_BIAS, so
// that byte 2 of it is (uint8)((int8)(acc >> 15) + 128), the atan2 table index.
template <int FIRST = 0>
IQD_DEV void fir16_window(const uint32_t (&x)[8], const Consts &c, int (&acc)[17])
{
    const int bias = WB_BIAS;   // lives in one VGPR for all chains
    uint32_t y[3][7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        y[0][j] = alignbyte(x[j + 1], x[j], 1);
        y[1][j] = alignbyte(x[j + 1], x[j], 2);
        y[2][j] = alignbyte(x[j + 1], x[j], 3);
    }
#pragma unroll
    for (int idx = FIRST; idx < 17; idx++) {
        const int s = idx & 3, j0 = idx >> 2;
        int lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = (s == 0) ? x[j0 + q] : y[s - 1][j0 + q];
            lo = q == 0 ? dot4_first(d, c.pre_lo[0], bias) : dot4(d, c.pre_lo[q], lo);
            hi = q == 0 ? dot4_first0(d, c.pre_hi[0]) : dot4(d, c.pre_hi[q], hi);
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

// The same for 8 dwords (16 samples) with packed 16-bit arithmetic: two samples per instruction.
// Per lane of 16 bits the partial sums stay below 8 * 192, so they cannot carry into the neighbour.
IQD_DEV uint32_t magnitude16(const uint32_t (&s)[8])
{
#if IQD_ON_DEVICE
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 acc = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const s2 w = __builtin_bit_cast(s2, s[j]);
        const s2 q = w >> 8;                       // sign-extended high bytes (Q)
        const s2 i = (s2)(w << 8) >> 8;            // sign-extended low bytes (I)
        const us2 a = __builtin_bit_cast(us2, __builtin_elementwise_max(i, (s2)(-i)));
        const us2 b = __builtin_bit_cast(us2, __builtin_elementwise_max(q, (s2)(-q)));
        const us2 mx = __builtin_elementwise_max(a, b), mn = __builtin_elementwise_min(a, b);
        acc += mx + (mn >> 1);
    }
    const uint32_t r = __builtin_bit_cast(uint32_t, acc);
    return (r & 0xffffu) + (r >> 16);
#else
    uint32_t m = 0;
    for (int j = 0; j < 8; j++) m += magnitude2(s[j]);
    return m;
#endif
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

// ---- phase 1, split so that two 16-sample groups per lane can be interleaved: the second
// group's FIR arithmetic runs while the first group's table gathers are in flight ----------
struct P1Raw { u32x4 r0, r1, r2, r3; };   // 16 lead-in samples + the group's own 16

template <bool GATED>
IQD_DEV P1Raw p1_load(const WbfmTile &t, int64_t v)
{
    const u32x4 *ph = raw_group<GATED>(t, v - 16);
    const u32x4 *po = raw_group<GATED>(t, v);
    return P1Raw{ph[0], ph[1], po[0], po[1]};
}

// raw -> signed -> rotated rails -> 17 FIR outputs per rail -> byte offsets into the atan2 table
IQD_DEV void p1_front(const WbfmTile &t, const Consts &c, const P1Raw &r, uint32_t (&off)[17])
{
    uint32_t s[16] = {r.r0.x, r.r0.y, r.r0.z, r.r0.w, r.r1.x, r.r1.y, r.r1.z, r.r1.w,
                      r.r2.x, r.r2.y, r.r2.z, r.r2.w, r.r3.x, r.r3.y, r.r3.z, r.r3.w};
    uint32_t xi[8], xq[8];
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] ^= 0x80808080u;  // offset binary -> signed (:735-738)
#pragma unroll
    for (int j = 0; j < 8; j++) rotate4(t, s[2 * j], s[2 * j + 1], xi[j], xq[j]);
    int ai[17], aq[17];
#ifdef IQD_ABL_NOFIR   // diagnostic build: skip the FIR arithmetic, keep its inputs alive
#pragma unroll
    for (int k = 0; k < 17; k++) { ai[k] = (int)(xi[k & 7] + k); aq[k] = (int)(xq[k & 7] ^ k); }
#else
    fir16_window(xi, c, ai);
    fir16_window(xq, c, aq);
#endif
#pragma unroll
    for (int k = 0; k < 17; k++)  // float index of lut[(uint8)(Q'+128)][(uint8)(I'+128)]: byte 2 of each sum
        off[k] = perm((uint32_t)aq[k], (uint32_t)ai[k], 0x0c0c0602u);
}

IQD_DEV void p1_gather(const WbfmTile &t, const uint32_t (&off)[17], float (&th)[17])
{
#pragma unroll
    for (int k = 0; k < 17; k++) {
#ifdef IQD_ABL_NOLUT   // diagnostic build: no table gather
        th[k] = u2f(0x3f000000u | (off[k] & 0xffu));
#else
        th[k] = t.lut[lut_index(off[k])];
#endif
    }
}

IQD_DEV uint32_t p1_magnitude(const P1Raw &r)
{
    const uint32_t k = 0x80808080u;
    const uint32_t own[8] = {r.r2.x ^ k, r.r2.y ^ k, r.r2.z ^ k, r.r2.w ^ k, r.r3.x ^ k, r.r3.y ^ k, r.r3.z ^ k, r.r3.w ^ k};
    return magnitude16(own);
}

// What phase 1 produces for one 16-sample group, held in registers until it may be stored.
struct P1Out {
    float u[16];     // b0 * (K * dtheta)
    float part;      // sum c^(15-k) u[k], for the IIR state guess
    uint32_t mag;    // squelch magnitude sum of the group
    int p;           // position of the group inside its chunk
    int valid;
};

// theta -> delta theta, branch cut, K*d, b0*v (+ the partial sum for the IIR state guess)
IQD_DEV void p1_make(const WbfmTile &t, const Consts &c, const float (&th)[17], P1Out &o)
{
    const float c2 = c.deemph_c * c.deemph_c;
    float pe = 0.f, po = 0.f;   // sum c^(15-k) u[k] as two interleaved chains (even / odd k)
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        float d0 = th[k + 1] - th[k], d1 = th[k + 2] - th[k + 1];
        d0 = wrap_delta(d0);
        d1 = wrap_delta(d1);
        const float v0 = t.k * d0, v1 = t.k * d1;
        o.u[k] = c.deemph_b0 * v0;
        o.u[k + 1] = c.deemph_b0 * v1;
        pe = __builtin_fmaf(c2, pe, o.u[k]);
        po = __builtin_fmaf(c2, po, o.u[k + 1]);
    }
    o.part = __builtin_fmaf(c.deemph_c, pe, po);
}

IQD_DEV void p1_store_t(WbfmLds &lds, const P1Out &o)
{
    if (!o.valid) return;
    const int seg = o.p >> 7, gq = (o.p & 127) >> 4;
#pragma unroll
    for (int q = 0; q < 4; q++)
        lds.t4[t_slot(seg, 4 * gq + q)] =
            u32x4{f2u(o.u[4 * q]), f2u(o.u[4 * q + 1]), f2u(o.u[4 * q + 2]), f2u(o.u[4 * q + 3])};
}

IQD_DEV void p1_store_part(float *part, const P1Out &o)
{
    if (o.valid) part[8 * (o.p >> 7) + ((o.p & 127) >> 4)] = o.part;
}

IQD_DEV void p1_finish(const WbfmTile &t, const Consts &c, WbfmLds &lds, const float (&th)[17], int p, bool valid)
{
    P1Out o;
    o.p = p;
    o.valid = valid;
    p1_make(t, c, th, o);
    p1_store_t(lds, o);
    p1_store_part(lds_part(lds), o);
}

IQD_DEV void p1_add_mag(const WbfmTile &t, WbfmLds &lds, const ChunkBlocks &cb, int p, uint32_t m)
{
    const uint32_t slot = div_block(t, cb.in_blk + (uint32_t)p);  // block slot relative to the chunk's first block
#if IQD_ON_DEVICE
    atomicAdd(&lds.mag[slot], m);
#else
    lds.mag[slot] += m;
#endif
}

template <bool GATED, bool MAG>
IQD_DEV void wbfm_phase1(const WbfmTile &t, const Consts &c, WbfmLds &lds, const ChunkBlocks &cb,
                         int cstart, int clen, int tid)
{
    const int ngroups = clen >> 4;
#ifdef IQD_ABL_NOMAG
    const bool want_mag = false;
#else
    const bool want_mag = MAG && cstart >= 0;
#endif
#ifdef IQD_P1_SINGLE
    for (int g0 = tid; (g0 & ~63) < ngroups; g0 += WB_THREADS) {
#else
    for (int g0 = tid; (g0 & ~63) < ngroups; g0 += 2 * WB_THREADS) {   // wave-uniform trip count
#endif
        // lanes past the end redo the chunk's last group and drop the result (no divergence)
        const int ga = g0 < ngroups ? g0 : ngroups - 1;
        const bool va = g0 < ngroups;
#ifdef IQD_P1_SINGLE   // experiment: one group per lane and pass (fewer VGPRs)
        const bool wave_has_b = false;
#else
        const bool wave_has_b = ((g0 & ~63) + WB_THREADS) < ngroups;
#endif
        const int gb = g0 + WB_THREADS < ngroups ? g0 + WB_THREADS : ngroups - 1;
        const bool vb = g0 + WB_THREADS < ngroups;
        const P1Raw ra = p1_load<GATED>(t, t.v0 + cstart + 16 * ga);
        uint32_t off[17];
        float tha[17];
        if (wave_has_b) {
            const P1Raw rb = p1_load<GATED>(t, t.v0 + cstart + 16 * gb);
            float thb[17];
            p1_front(t, c, ra, off);
            p1_gather(t, off, tha);
            const uint32_t ma = want_mag ? p1_magnitude(ra) : 0u;
            p1_front(t, c, rb, off);
            p1_gather(t, off, thb);
            const uint32_t mb = want_mag ? p1_magnitude(rb) : 0u;
            p1_finish(t, c, lds, tha, 16 * ga, va);
            p1_finish(t, c, lds, thb, 16 * gb, vb);
            if (want_mag && va) p1_add_mag(t, lds, cb, 16 * ga, ma);
            if (want_mag && vb) p1_add_mag(t, lds, cb, 16 * gb, mb);
        } else {
            p1_front(t, c, ra, off);
            p1_gather(t, off, tha);
            const uint32_t ma = want_mag ? p1_magnitude(ra) : 0u;
            p1_finish(t, c, lds, tha, 16 * ga, va);
            if (want_mag && va) p1_add_mag(t, lds, cb, 16 * ga, ma);
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

IQD_DEV float t_at(const WbfmLds &lds, int seg, int i)   // u of sample i (0..127) of a segment
{
    const u32x4 g = lds.t4[t_slot(seg, i >> 2)];
    const uint32_t w[4] = {g.x, g.y, g.z, g.w};
    return u2f(w[i & 3]);
}

// A chunk is whole 128-sample segments - except that a call may end (and a gain may have changed) on any multiple of 32
// samples, and that a carried restart point may lie any multiple of 32 samples back (round 4: WBFM calls in 64-byte units,
// IqDataProcessor.cc:586 strides 8 bytes, WbFmDemodulator.cc:383-411 takes what it is given).  Both are lane-local:
//   last_len    samples of the chunk's LAST segment that exist (32 .. 128); its lane stops there, and the state it
//               leaves is the chunk's end state
//   first_skip  samples at the head of the tile's FIRST segment that lie before the carried restart point (0 .. 96): the
//               lead-in is cut on the segment grid, the exact carried state applies first_skip samples into it, and what
//               lies before is silence (no output can reach that far back: the decimators remember 696 samples)
struct IirShape {
    int nseg, last_len, first_skip;
};
IQD_DEV IirShape iir_shape(int clen, int first_skip)
{
    IirShape s;
    s.nseg = (clen + SEG - 1) / SEG;
    s.last_len = clen - SEG * (s.nseg - 1);
    s.first_skip = first_skip;
    return s;
}

IQD_DEV float iir_u_before(const WbfmLds &lds, int seg)  // u[n-1] at the start of a segment
{
    return seg == 0 ? lds.u_carry : t_last(lds, seg - 1);
}

// lane j: approximate zero-state response of segment j to t[n] = u[n] + u[n-1], from the
// per-group partial sums phase 1 left (plain float arithmetic: this is only the state GUESS).
//   S = sum c^(127-i) u[i];  response = S + c^127 u[-1] + (S - u[127]) / c
IQD_DEV void iir_guess(const Consts &c, WbfmLds &lds, int nseg, int lane, const float *part = nullptr)
{
    if (!part) part = lds_part(lds);
    if (lane == 0) lds.z[3] = lds.y_carry, lds.z[2] = 0.f, lds.z[1] = 0.f, lds.z[0] = 0.f;
    if (lane >= nseg) return;
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) sum = __builtin_fmaf(c.deemph_c16, sum, part[8 * lane + g]);
    const float z = sum + c.deemph_c127 * iir_u_before(lds, lane) + (sum - t_last(lds, lane)) * c.deemph_cinv;
    lds.z[4 + lane] = z;
}

// One de-emphasis step, exactly as IirFilter::filterData evaluates it (IirFilter.cc:161-176):
// y = (b0 x[n] + b1 x[n-1]) - (a1 y[n-1]), every operation rounded to binary32.
#ifndef IQD_RELAXED_TOL
#define IQD_RELAXED_TOL 0   // 1: TIMING A/B ONLY ("is bit-exactness what caps the roofline?") - the recurrence with one
#endif                      // fused multiply-add, K b0 folded into one factor: PCM within +-1 LSB of the reference instead of identical.  Never shipped.
#if IQD_RELAXED_TOL
#define IQD_IIR_STEP(U)                              \
    {                                                \
        y = __builtin_fmaf(-a1, y, (U) + up);        \
        up = (U);                                    \
    }
#else
#define IQD_IIR_STEP(U)            \
    {                              \
        const float tn_ = (U) + up; \
        const float r_ = a1 * y;   \
        y = tn_ - r_;              \
        up = (U);                  \
    }
#endif

// lane j >= 1: run segment j-1 from the guessed state to get the state entering segment j.
IQD_DEV void iir_warm(const Consts &c, WbfmLds &lds, int nseg, int lane, int first_skip = 0)
{
    if (lane >= nseg) return;
    if (lane == 0) { lds.g[0] = lds.y_carry; return; }
    const float a = c.deemph_c128;
    // approx y at the end of segment lane-2:  z[l-2] + A z[l-3] + A^2 z[l-4] + A^3 z[l-5]
    const float *zz = &lds.z[4 + lane - 2];
    float y = zz[0] + a * (zz[-1] + a * (zz[-2] + a * zz[-3]));
    if (lane == 1) y = lds.y_carry;  // exact (the carried state applies first_skip samples into segment 0)
    float up = iir_u_before(lds, lane - 1);
    const float a1 = c.deemph_a1;
#pragma unroll 2
    for (int gi = lane == 1 ? first_skip >> 2 : 0; gi < 32; gi += 2) {
        const u32x4 a4 = lds.t4[t_slot(lane - 1, gi)], b4 = lds.t4[t_slot(lane - 1, gi + 1)];
        IQD_IIR_STEP(u2f(a4.x)) IQD_IIR_STEP(u2f(a4.y)) IQD_IIR_STEP(u2f(a4.z)) IQD_IIR_STEP(u2f(a4.w))
        IQD_IIR_STEP(u2f(b4.x)) IQD_IIR_STEP(u2f(b4.y)) IQD_IIR_STEP(u2f(b4.z)) IQD_IIR_STEP(u2f(b4.w))
    }
    lds.g[lane] = y;
}

// lane j: the real pass over segment j from g[j]; writes (int16)y and e[j].
// bounded: the host proved |y| < 2^31 for this launch (|K| pi * 1.01 < 2^31), so the cast needs
// no "integer indefinite" handling.
// mark: the restart record of the tile falls INSIDE a segment of this chunk (a call that does not end on the 128-sample grid):
// the lane that runs that segment leaves the state entering granule `gi` in z[0] (y) and z[1] (u) - two cells that only the
// state guess uses, and rewrites before it does (iir_guess) - for wbfm_rec_in_chunk.
struct IirMark { int lane, gi; };
IQD_DEV void iir_real(const Consts &c, WbfmLds &lds, int nseg, int lane, bool bounded, int first_skip = 0, int last_len = SEG,
                      IirMark mark = IirMark{-1, 0})
{
    if (lane >= nseg) return;
    u32x4 *dst = (u32x4 *)&lds.w[lane * WSTRIDE];
    float up = iir_u_before(lds, lane), y = lds.g[lane];
    const float a1 = c.deemph_a1;
    // granules (4 samples) of this lane's segment that are run: all 32, but for the head of the tile's first segment
    // (silence before the carried restart point) and the tail of a chunk's last one (IirShape)
    const int gi0 = lane == 0 ? first_skip >> 2 : 0, gi1 = lane == nseg - 1 ? last_len >> 2 : 32;
    for (int gi = 0; gi < gi0; gi += 2) dst[gi >> 1] = u32x4{0u, 0u, 0u, 0u};
    if (bounded) {
#pragma unroll 2
        for (int gi = gi0; gi < gi1; gi += 2) {
            if (lane == mark.lane && gi == mark.gi) { lds.z[0] = y; lds.z[1] = up; }
            const u32x4 a4 = lds.t4[t_slot(lane, gi)], b4 = lds.t4[t_slot(lane, gi + 1)];
            uint32_t w[4];
            float ye;
            IQD_IIR_STEP(u2f(a4.x)) ye = y;
            IQD_IIR_STEP(u2f(a4.y)) w[0] = cast_pack_i16_bounded(ye, y);
            IQD_IIR_STEP(u2f(a4.z)) ye = y;
            IQD_IIR_STEP(u2f(a4.w)) w[1] = cast_pack_i16_bounded(ye, y);
            IQD_IIR_STEP(u2f(b4.x)) ye = y;
            IQD_IIR_STEP(u2f(b4.y)) w[2] = cast_pack_i16_bounded(ye, y);
            IQD_IIR_STEP(u2f(b4.z)) ye = y;
            IQD_IIR_STEP(u2f(b4.w)) w[3] = cast_pack_i16_bounded(ye, y);
            dst[gi >> 1] = u32x4{w[0], w[1], w[2], w[3]};
        }
        lds.e[lane] = y;
        return;
    }
#pragma unroll 2
    for (int gi = gi0; gi < gi1; gi += 2) {
        if (lane == mark.lane && gi == mark.gi) { lds.z[0] = y; lds.z[1] = up; }
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
// Two de-emphasis states "cannot be told apart at the output" when both are below 2^-100 and K >= 1: the first
// nonzero recurrence input (at least 1e-14 in magnitude then) absorbs either of them completely, and until it
// arrives every (int16)y is 0.  Exactly constant input (digital silence, a noiseless carrier) makes this matter:
// the true state then sticks at a denormal (0.949 k rounds back to k for |k| <= 9 units of 2^-149) while a warm-up
// from zero sits at 0, and insisting on bit equality would serialise everything for no audible sample.
IQD_DEV bool iir_states_agree(float a, float b, bool tiny_ok)
{
    if (f2u(a) == f2u(b)) return true;
    return tiny_ok && __builtin_fabsf(a) < 0x1p-100f && __builtin_fabsf(b) < 0x1p-100f;
}

IQD_DEV bool iir_check(WbfmLds &lds, int nseg, int lane, bool tiny_ok = false)
{
    if (lane == 0 || lane >= nseg) return true;
    const float want = lds.e[lane - 1];
    if (iir_states_agree(want, lds.g[lane], tiny_ok)) return true;
    lds.g[lane] = want;
    return false;
}

// ---- decimators ----------------------------------------------------------------------------
IQD_DEV u32x2 w_group(const WbfmLds &lds, int g, int hslot = 0)  // 4 consecutive w samples, group index g
{
    if (g < 0) return u32x2{lds.whist[2 * hslot], lds.whist[2 * hslot + 1]};
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
IQD_DEV void wbfm_stage1(const Consts &c, WbfmLds &lds, int clen, int tid, int hslot = 0)
{
    // Four consecutive outputs per lane and pass: w groups m0-1 .. m0+3 (a group = 4 samples = 2 dwords;
    // 32 groups per segment, so groups m0 .. m0+3 never straddle segments), two passes issued together
    // so that the LDS latency is paid once.
    const int nquad = clen >> 4;
    const uint32_t t76 = ((uint32_t)(uint16_t)c.wbfm_d1[7]) | ((uint32_t)(uint16_t)c.wbfm_d1[6] << 16);
    const uint32_t t54 = ((uint32_t)(uint16_t)c.wbfm_d1[5]) | ((uint32_t)(uint16_t)c.wbfm_d1[4] << 16);
    const uint32_t t32 = ((uint32_t)(uint16_t)c.wbfm_d1[3]) | ((uint32_t)(uint16_t)c.wbfm_d1[2] << 16);
    const uint32_t t10 = ((uint32_t)(uint16_t)c.wbfm_d1[1]) | ((uint32_t)(uint16_t)c.wbfm_d1[0] << 16);
    for (int k0 = tid; k0 < nquad; k0 += 2 * WB_THREADS) {
        u32x2 prev[2];
        u32x4 lo[2], hi[2];
        bool ok[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int k = k0 + r * WB_THREADS;
            ok[r] = k < nquad;
            const int m0 = 4 * (ok[r] ? k : k0);
            prev[r] = w_group(lds, m0 - 1, hslot);
            const u32x4 *p = (const u32x4 *)&lds.w[(m0 >> 5) * WSTRIDE + 2 * (m0 & 31)];
            lo[r] = p[0];
            hi[r] = p[1];
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (!ok[r]) continue;
            const int m0 = 4 * (k0 + r * WB_THREADS);
            // window of output m: x[4m-4 .. 4m+3] ascending <-> taps h[7 .. 0]
            const uint32_t g[10] = {prev[r].x, prev[r].y, lo[r].x, lo[r].y, lo[r].z, lo[r].w,
                                    hi[r].x, hi[r].y, hi[r].z, hi[r].w};
            uint32_t y[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                int acc = 1 << 14;
                acc = dot2(g[2 * o], t76, acc);
                acc = dot2(g[2 * o + 1], t54, acc);
                acc = dot2(g[2 * o + 2], t32, acc);
                acc = dot2(g[2 * o + 3], t10, acc);
                y[o] = (uint32_t)(acc >> 15);
            }
            *(u32x2 *)&lds.y1[(8 + m0) >> 1] = u32x2{pack_lo16(y[0], y[1]), pack_lo16(y[2], y[3])};
        }
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

// The same dot product when the bound proves that no clamp can fire: whole pairs per v_dot2,
// any order.  L even, `newest` odd (the window then starts and ends on dword boundaries).
template <int L>
IQD_DEV int q15_pairs(const int16_t *h, const uint32_t *buf, int newest)
{
    int acc = 1 << 14;
    const uint32_t *p = buf + (newest >> 1);
#pragma unroll
    for (int q = 0; q < L / 2; q++)
        acc = dot2(p[-q], (uint32_t)(uint16_t)h[2 * q + 1] | ((uint32_t)(uint16_t)h[2 * q] << 16), acc);
    return acc >> 15;
}

constexpr int AUDIO40_SAFE = 16061;   // (2^30 - 16384) / 66852: below it the 40-tap sum cannot clamp
constexpr int POST12_SAFE = 29210;    // (2^30 - 16384) / 36758

IQD_DEV void lds_max(uint32_t *slot, uint32_t v)
{
#if IQD_ON_DEVICE
    atomicMax(slot, v);
#else
    if (v > *slot) *slot = v;
#endif
}

// /4, 12 taps (WbFmDemodulator.cc:541).  Its input is stage 1's output, |y1| <= 29126 whatever the
// data (sum|h1| = 29126 < 2^15), and 16384 + 36758 * 29126 < 2^30: the clamp can never fire here.
IQD_DEV void wbfm_stage2(const Consts &c, WbfmLds &lds, int clen, int tid, int nthreads = WB_THREADS)
{
    // two consecutive outputs per lane: y1 dwords 2*j0 .. 2*j0+7 (j0 even -> 16-byte aligned)
    const int npair = clen >> 5;
    uint32_t taps[6];
#pragma unroll
    for (int q = 0; q < 6; q++)   // pair q counts back from the newest dword: (lo: h[2q+1], hi: h[2q])
        taps[q] = (uint32_t)(uint16_t)c.post12[2 * q + 1] | ((uint32_t)(uint16_t)c.post12[2 * q] << 16);
    uint32_t peak = 0;
    for (int k = tid; k < npair; k += nthreads) {
        const u32x4 *p = (const u32x4 *)&lds.y1[4 * k];
        const u32x4 a = p[0], b = p[1];
        const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        int y[2];
#pragma unroll
        for (int o = 0; o < 2; o++) {   // output j = 2k + o: newest dword = 2j + 5 = d[2*o + 5]
            int acc = 1 << 14;
#pragma unroll
            for (int q = 0; q < 6; q++) acc = dot2(d[2 * o + 5 - q], taps[q], acc);
            y[o] = acc >> 15;
            const uint32_t m = (uint32_t)(y[o] < 0 ? -y[o] : y[o]);
            peak = m > peak ? m : peak;
        }
        lds.y2[20 + k] = pack_lo16((uint32_t)y[0], (uint32_t)y[1]);
    }
    if (peak > (uint32_t)AUDIO40_SAFE) lds_max(&lds.y2_peak, peak);   // rare: loud audio only
}

// /2, 40 taps (WbFmDemodulator.cc:546) -> PCM
IQD_DEV void wbfm_stage3(const Consts &c, WbfmLds &lds, const WbfmTile &t, int cstart, int clen, int tid,
                         int nthreads = WB_THREADS)
{
    const int nout = clen >> 5;
    // the reference's per-MAC clamp matters only when some |y2| in reach exceeds AUDIO40_SAFE
    const bool quiet = lds.y2_peak <= (uint32_t)AUDIO40_SAFE && lds.y2_peak_hist <= (uint32_t)AUDIO40_SAFE;
    for (int i = tid; i < nout; i += nthreads) {
        const int y = quiet ? q15_pairs<40>(c.audio40, lds.y2, 40 + 2 * i + 1)
                            : q15_seq<40>(c.audio40, lds.y2, 40 + 2 * i + 1);
        if (cstart >= 0) t.pcm_row[((t.v0 + cstart) >> 5) + i] = (int16_t)y;
    }
}

// Histories for the next chunk (call after a barrier that follows stage3).
// Histories for the next chunk.  Part A (y1 tail, last w group) runs beside stage 3, which touches
// neither; part B (y2 tail, loudness flags) runs at the head of the next chunk's phase 1, when
// stage 3 has finished reading y2.
IQD_DEV void wbfm_shift_a(WbfmLds &lds, int clen, int tid)
{
    const int n1 = clen >> 2;
    if (tid >= 100 && tid < 104) {  // 8 int16 = 4 dwords
        const int k = tid - 100;
        lds.y1[k] = lds.y1[(n1 >> 1) + k];
    } else if (tid == 110) {
        const u32x2 last = w_group(lds, n1 - 1);
        lds.whist[0] = last.x;
        lds.whist[1] = last.y;
    }
}

IQD_DEV void wbfm_shift_b(WbfmLds &lds, int clen, int tid)
{
    const int n2 = clen >> 4;
    if (tid >= 64 && tid < 64 + 20 && (n2 >> 1) >= 20) {  // 40 int16 = 20 dwords
        const int k = tid - 64;
        lds.y2[k] = lds.y2[(n2 >> 1) + k];
    } else if (tid == 96 && (n2 >> 1) < 20) {  // short chunk: ranges overlap, move in order
        for (int k = 0; k < 20; k++) lds.y2[k] = lds.y2[(n2 >> 1) + k];
    } else if (tid == 97) {   // conservative: the 40 samples kept may contain this chunk's peak
        const uint32_t keep = n2 >= 40 ? 0u : lds.y2_peak_hist;
        lds.y2_peak_hist = lds.y2_peak > keep ? lds.y2_peak : keep;
        lds.y2_peak = 0;
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
    // Streaming launches (iqd_stream.hip).  A cold segment's state is exact from its own start on (verified there), not inside its
    // 768-sample lead-in - so when a channel's LAST segment is shorter than FORCED_BACK, the restart point vlen - FORCED_BACK lies in
    // the segment before it, which then keeps the state there in pad[0] (y) and pad[1] (u); the last segment's own record carries
    // WBFM_REC_STREAMED in pad[0], so that the commit can tell it from the record a tile-kernel repair writes (exact as it is).
    uint32_t pad[2];
};
constexpr uint32_t WBFM_REC_STREAMED = 0x53545245u;

// Where a STREAMED segment (iqd_stream.hip: st_iir_wave) takes its restart record, as plain arithmetic - shared by the kernel and by
// the CPU tier's model of the restart state's journey from call to call (tests/test_emu_restart_model.py; round 4's three
// restart-state bugs all lived in these few lines).  A segment's positions count from its own first sample; its lead-in is
// FORCED_BACK samples (a channel's first segment: from the carried exact state, `carried_back` before the call).
//   rec_pos       where the segment's own record (y_out, u_out) is taken: FORCED_BACK before its end, or at the carried state
//                 itself when the segment is shorter than that (then back_out says how far before the new end that lies)
//   keeps_restart the channel's restart point vlen - FORCED_BACK lies in THIS segment although it is not the last - because the
//                 last one is shorter than FORCED_BACK and would take it inside its lead-in, where a cold segment's state is still
//                 converging: this segment parks the state there (WbfmRecord::pad; `park_pos`), its own record stays where
//                 every full segment's is
struct StRecPlan { int32_t rec_pos, back_out, park_pos; uint32_t keeps_restart; };
IQD_DEV StRecPlan st_rec_plan(uint32_t valid, uint32_t tile, int32_t v0, int32_t tlen, int32_t vlen, int32_t carried_back)
{
    StRecPlan r;
    const int32_t halo = tile == 0 ? carried_back : FORCED_BACK;
    r.rec_pos = tlen - FORCED_BACK;
    if (r.rec_pos < -halo) r.rec_pos = -halo;
    r.back_out = tlen - r.rec_pos;
    const bool not_last = vlen - v0 > tlen;
    r.keeps_restart = valid && not_last && vlen - FORCED_BACK >= v0 && vlen - FORCED_BACK < v0 + tlen ? 1u : 0u;
    r.park_pos = r.keeps_restart ? vlen - FORCED_BACK - v0 : r.rec_pos;
    return r;
}

// The restart state a channel carries out of a call: its last tile's / segment's record - or, when that last one was a streamed
// segment shorter than FORCED_BACK, what the segment before it parked (tail_update_body; `before` = that segment's record).
IQD_DEV WbfmCarry wbfm_pick_carry(const WbfmRecord &last, const WbfmRecord &before, uint32_t ntiles, uint32_t vlen, uint32_t tile_len,
                                  uint32_t verify_at_end)
{
    WbfmCarry cy;
    cy.y = last.y_out; cy.u = last.u_out; cy.back = last.back_out;
    if (verify_at_end && ntiles >= 2 && last.pad[0] == WBFM_REC_STREAMED && vlen - (ntiles - 1) * tile_len < (uint32_t)FORCED_BACK) {
        cy.y = u2f(before.pad[0]); cy.u = u2f(before.pad[1]); cy.back = FORCED_BACK;
    }
    cy.y_end = last.y_end; cy.u_end = last.u_end;
    cy.pad[0] = cy.pad[1] = cy.pad[2] = 0;
    return cy;
}

// Chunk boundaries of a tile: the lead-in [-halo, 0) is one chunk - or two, split where the demodulator gain last
// changed (the channel's GainEpochList), so that every chunk has one gain.
// The gain in force at call-relative position v < 0: the list is ordered, most recent change first.
IQD_DEV float epoch_gain(const GainEpochList *ep, float k_now, int v)
{
    float k = k_now;
    if (ep)
        for (int i = 0; i < EPOCHS; i++) {
            if (ep->since[i] >= (uint32_t)TAIL || v >= -(int)ep->since[i]) break;
            k = ep->k_before[i];
        }
    return k;
}
// The same without a data-dependent loop, for the streaming kernels' piece loops: `since` ascends with the index (older
// changes have consumed more; entries out of reach sit at TAIL, beyond any position a lead-in can have), so the number
// m of changes that lie AFTER position v - since[i] < -v - is found by a binary search of log2(EPOCHS) steps, each a load
// and a select; the gain is k_before[m - 1], or the current one when m = 0.  (Round 4: with 64 entries the linear scan
// above became a real, divergent loop inside the FM pipeline's software-pipelined piece body, and that build faulted on
// the device at launch geometries with few workgroups per family - tools/dbg_mixed.py, gpurun_out: "fm wgs 56".)
IQD_DEV float epoch_gain_search(const GainEpochList *ep, float k_now, int v)
{
    static_assert((EPOCHS & (EPOCHS - 1)) == 0, "a power of two");
    const uint32_t want = (uint32_t)(v < 0 ? -v : 0);            // entries with since < want lie after v
    int m = 0;
#pragma unroll
    for (int step = EPOCHS / 2; step >= 1; step >>= 1)
        m += ep->since[m + step - 1] < want ? step : 0;
    // m in [0, EPOCHS - 1] entries qualify among the first EPOCHS - 1; the last one, if the list is full:
    m += m == EPOCHS - 1 && ep->since[EPOCHS - 1] < want ? 1 : 0;
    const float kb = ep->k_before[m > 0 ? m - 1 : 0];
    return m > 0 ? kb : k_now;
}

IQD_DEV int wbfm_chunk_len(const WbfmTile &t, int cs, int chunk, int lead_from = -TAIL)
{
    if (cs < 0) {   // the lead-in ends at the tile start or at the next gain change, whichever comes first
        int next = 0;   // (a change at or before lead_from, the carried restart point, lies in the silence before it)
        if (t.epochs)
            for (int i = 0; i < EPOCHS; i++) {
                const int sw = -(int)t.epochs->since[i];
                if (t.epochs->since[i] < (uint32_t)TAIL && sw > cs && sw > lead_from && sw < next) next = sw;
            }
        return next - cs;
    }
    return t.tlen - cs < chunk ? t.tlen - cs : chunk;
}

// The restart point a tile leaves for whoever continues its stream: the last position at or before tlen - FORCED_BACK
// where the exact state is at hand - a chunk's start (the carried state) or a segment boundary inside a chunk - and not
// before `from`, the point where the tile's own exact state began.  wbfm_rec_at_start / wbfm_rec_in_chunk are called
// chunk by chunk in stream order, so the last candidate that qualifies stays.
struct WbfmRecPos { int target, from, pos; };
IQD_DEV void wbfm_rec_at_start(WbfmRecPos &rp, WbfmRecord &rec, const WbfmLds &lds, int cstart)
{
    if (cstart >= rp.from && cstart <= rp.target) { rp.pos = cstart; rec.y_out = lds.y_carry; rec.u_out = lds.u_carry; }
}
// the target itself, when it lies inside a segment of this chunk (and not before the tile's own exact state began)
IQD_DEV IirMark wbfm_mark_in_chunk(const WbfmRecPos &rp, int cstart, int clen)
{
    const int off = rp.target - cstart;
    if (off <= 0 || off >= clen || rp.target < rp.from || off % SEG == 0) return IirMark{-1, 0};
    return IirMark{off / SEG, (off % SEG) >> 2};
}
IQD_DEV void wbfm_rec_in_chunk(WbfmRecPos &rp, WbfmRecord &rec, const WbfmLds &lds, int cstart, int clen, IirMark mark = IirMark{-1, 0})
{
    if (mark.lane >= 0) {   // (round 4: exactly tlen - FORCED_BACK, so that `back` never exceeds the streaming kernel's lead-in)
        rp.pos = rp.target;
        rec.y_out = lds.z[0];
        rec.u_out = lds.z[1];
        return;
    }
    if (rp.target <= cstart) return;
    int j = (rp.target - cstart) / SEG;                    // whole segments of this chunk that end at or before the target
    if (j > clen / SEG) j = clen / SEG;
    if (SEG * j == clen) j--;                              // (the chunk's very end is the next chunk's start - or the tile's end)
    if (j >= 1 && cstart + SEG * j >= rp.from) { rp.pos = cstart + SEG * j; rec.y_out = lds.e[j - 1]; rec.u_out = t_last(lds, j - 1); }
}
IQD_DEV float wbfm_chunk_gain(const WbfmTile &t, int cs) { return cs < 0 ? epoch_gain(t.epochs, t.k, cs) : t.k; }
// largest |K| and smallest K a tile can meet: now, and every earlier gain still inside its lead-in
IQD_DEV void epoch_k_range(const GainEpochList *ep, float k_now, float &kmax_abs, float &kmin)
{
    kmax_abs = __builtin_fabsf(k_now);
    kmin = k_now;
    if (ep)
        for (int i = 0; i < EPOCHS; i++) {
            if (ep->since[i] >= (uint32_t)TAIL) break;
            const float k = ep->k_before[i], a = __builtin_fabsf(k);
            kmax_abs = a > kmax_abs ? a : kmax_abs;
            kmin = k < kmin ? k : kmin;
        }
}

template <bool GATED, bool MAG, class Exec>
IQD_DEV void wbfm_tile(Exec &ex, const WbfmTile &t, const Consts &c, WbfmLds &lds,
                       const WbfmStart &start, WbfmRecord *rec_out)
{
    // the lead-in starts on the segment grid; a carried restart point that is not on it (a stream of calls in 64-byte
    // units) lies first_skip samples into the first segment (IirShape)
    const int exact_from = start.cold ? -COLD_HALO : -start.back;
    const int halo = start.cold ? COLD_HALO : (start.back + SEG - 1) / SEG * SEG;
    ex.all([&](int tid) {
        if (tid < 4) lds.y1[tid] = 0;
        if (tid < 20) lds.y2[tid] = 0;
        if (tid < 4) lds.whist[tid] = 0;
        if (tid < WBFM_CHUNK / SEG + 2) lds.mag[tid] = 0;
        if (tid == 0) {
            lds.y_carry = start.cold ? 0.f : start.y;
            lds.u_carry = start.cold ? 0.f : start.u;
            lds.repair_count = 0;
            lds.y2_peak = 0;
            lds.y2_peak_hist = 0;
        }
    });
    // restart point for whoever continues this stream: FORCED_BACK before the end when the
    // tile (plus its exact lead-in) is long enough, else the tile's own restart point.
    WbfmRecPos rp{t.tlen - FORCED_BACK, exact_from, exact_from};
    WbfmRecord rec;
    rec.y_in = start.y;
    rec.y_out = start.y;
    rec.u_out = start.u;

    int prev_clen = 0;
    for (int cstart = -halo; cstart < t.tlen;) {
        const int clen = wbfm_chunk_len(t, cstart, WBFM_CHUNK, exact_from);
        WbfmTile tc = t;
        tc.k = wbfm_chunk_gain(t, cstart < exact_from ? exact_from : cstart);   // (the gain at the first sample that counts)
        const IirShape sh = iir_shape(clen, cstart == -halo ? halo + exact_from : 0);
        const int nseg = sh.nseg;
        const ChunkBlocks cb = chunk_blocks(t, cstart);
        ex.stamp(7);
        ex.all([&](int tid) {
            if (prev_clen) wbfm_shift_b(lds, prev_clen, tid);
            wbfm_phase1<GATED, MAG>(tc, c, lds, cb, cstart, clen, tid);
        });
        ex.stamp(0);
        if (ex.in_wave0()) {
            ex.critical(true);
            wbfm_rec_at_start(rp, rec, lds, cstart);
            const IirMark mark = wbfm_mark_in_chunk(rp, cstart, clen);
            ex.wave0([&](int lane) { iir_guess(c, lds, nseg, lane); });
            ex.stamp(1);
            ex.wave0([&](int lane) { iir_warm(c, lds, nseg, lane, sh.first_skip); });
            ex.stamp(2);
            int rounds = 0;
            do {
                ex.wave0([&](int lane) { iir_real(c, lds, nseg, lane, t.bounded != 0, sh.first_skip, sh.last_len, mark); });
                rounds++;
            } while (!ex.wave0_all([&](int lane) { return iir_check(lds, nseg, lane, t.k_min >= 1.0f); }));
            ex.stamp(3);
            wbfm_rec_in_chunk(rp, rec, lds, cstart, clen, mark);
            if (start.cold && cstart < 0) rec.y_in = lds.e[(COLD_HALO - FORCED_BACK) / SEG - 1];
            ex.wave0([&](int lane) {
                if (lane == 0) {
                    lds.y_carry = lds.e[nseg - 1];
                    lds.u_carry = t_at(lds, nseg - 1, sh.last_len - 1);
                    lds.repair_count += (uint32_t)(rounds - 1);
                }
            });
            ex.critical(false);
        }
        ex.sync();
        ex.stamp(4);
        ex.all([&](int tid) {
            if (MAG) wbfm_flush_mag(t, lds, cb, cstart, clen, tid);
            wbfm_stage1(c, lds, clen, tid);
        });
        ex.stamp(5);
        ex.all([&](int tid) { wbfm_stage2(c, lds, clen, tid); });
        ex.all([&](int tid) {
            wbfm_stage3(c, lds, t, cstart, clen, tid);
            wbfm_shift_a(lds, clen, tid);
        });
        prev_clen = clen;
        ex.stamp(6);
        cstart += clen;
    }
    if (ex.in_wave0() && rec_out) {
        rec.back_out = t.tlen - rp.pos;
        rec.y_end = lds.y_carry;
        rec.u_end = lds.u_carry;
        rec.pad[0] = rec.pad[1] = 0;
        ex.wave0([&](int lane) { if (lane == 0) *rec_out = rec; });
    }
}


// ---- pipelined tile driver -------------------------------------------------------------------
// Two workgroup barriers per chunk.  The serial IIR of chunk k (wave 0) runs beside the last two
// decimation stages of chunk k-1 and phase 1 of chunk k+1 (waves 1-3), whose results wait in registers
// until the IIR has released the LDS input buffer:
//
//   X   wave 0:    IIR(k): t4, part -> w; then phase 1 of groups 378.. of chunk k+1
//       waves 1-3: flush magnitudes(k); stage 2(k-1); rendezvous of these three waves;
//                  stage 3(k-1) -> PCM, y1 history; phase 1 of two groups per lane of chunk k+1     | barrier
//   Y   all:       y2 history(k-1); store u(k+1) -> t4, guess sums -> part, magnitude sums;
//                  stage 1(k): w -> y1                                                              | barrier
//
// Nothing one side of X writes is read by the other side before the barrier: the IIR touches t4, part, w,
// z/g/e and the carries; stages 2 and 3 touch y1, y2 and the loudness flags; phase 1 lives in registers.
struct P1Pair { P1Out a, b; };

// ---- phase 1 with neighbour sharing ------------------------------------------------------------
// Consecutive lanes hold consecutive groups, so a lane's 16-sample lead-in is its left neighbour's
// own rotated data and its theta[-1] the neighbour's theta[15]: both arrive by one DPP wave shift
// instead of being recomputed.  Lane 0 of a wave has no neighbour; it therefore redoes the group
// before the wave's first one (1/64 redundancy) purely as a donor: its rotated bytes and theta[15]
// need no lead-in, everything else it computes is dropped.
struct P1Own { u32x4 r2, r3; };   // a group's own 32 raw bytes

constexpr int P1S_PER_WAVE = 63;                       // useful groups per wave and pass
IQD_DEV int p1s_group(int slot, int lane) { return P1S_PER_WAVE * slot + lane - 1; }

template <bool GATED>
IQD_DEV P1Own p1s_load(const WbfmTile &t, int cstart, int ngroups, int g_raw)
{
    const int g = g_raw < ngroups ? g_raw : ngroups - 1;      // g = -1 is the lead-in of the chunk
    const u32x4 *po = raw_group<GATED>(t, t.v0 + cstart + 16 * g);
    return P1Own{po[0], po[1]};
}

template <int SLOT0, class Exec>
IQD_DEV void p1s_front(Exec &ex, int tid, const WbfmTile &t, const Consts &c, const P1Own &r, uint32_t (&off)[17])
{
    uint32_t s[8] = {r.r2.x, r.r2.y, r.r2.z, r.r2.w, r.r3.x, r.r3.y, r.r3.z, r.r3.w};
    uint32_t xi[8], xq[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] ^= 0x80808080u;
#pragma unroll
    for (int j = 0; j < 4; j++) rotate4(t, s[2 * j], s[2 * j + 1], xi[4 + j], xq[4 + j]);
    xi[0] = ex.template shr1<SLOT0 + 0>(tid, xi[4]);
    xi[1] = ex.template shr1<SLOT0 + 1>(tid, xi[5]);
    xi[2] = ex.template shr1<SLOT0 + 2>(tid, xi[6]);
    xi[3] = ex.template shr1<SLOT0 + 3>(tid, xi[7]);
    xq[0] = ex.template shr1<SLOT0 + 4>(tid, xq[4]);
    xq[1] = ex.template shr1<SLOT0 + 5>(tid, xq[5]);
    xq[2] = ex.template shr1<SLOT0 + 6>(tid, xq[6]);
    xq[3] = ex.template shr1<SLOT0 + 7>(tid, xq[7]);
    int ai[17], aq[17];
#ifdef IQD_ABL_NOFIR   // diagnostic build: skip the FIR arithmetic, keep its inputs alive
#pragma unroll
    for (int k = 1; k < 17; k++) { ai[k] = (int)(xi[k & 7] + k); aq[k] = (int)(xq[k & 7] ^ k); }
#else
    fir16_window<1>(xi, c, ai);
    fir16_window<1>(xq, c, aq);
#endif
#pragma unroll
    for (int k = 1; k < 17; k++) off[k] = perm((uint32_t)aq[k], (uint32_t)ai[k], 0x0c0c0602u);
}

IQD_DEV void p1s_gather(const WbfmTile &t, const uint32_t (&off)[17], float (&th)[17])
{
#pragma unroll
    for (int k = 1; k < 17; k++) {
#ifdef IQD_ABL_NOLUT   // diagnostic build: no table gather
        th[k] = u2f(0x3f000000u | (off[k] & 0xffu));
#else
        th[k] = t.lut[lut_index(off[k])];
#endif
    }
}

IQD_DEV uint32_t p1s_magnitude(const P1Own &r)
{
    const uint32_t k = 0x80808080u;
    const uint32_t own[8] = {r.r2.x ^ k, r.r2.y ^ k, r.r2.z ^ k, r.r2.w ^ k, r.r3.x ^ k, r.r3.y ^ k, r.r3.z ^ k, r.r3.w ^ k};
    return magnitude16(own);
}

// Two passes (slots sa, sb) of one wave, interleaved: the second pass's FIR arithmetic runs while the first
// pass's table gathers are in flight.  lane = tid & 63.
template <class Exec>
IQD_DEV void p1s_compute(Exec &ex, int tid, const WbfmTile &t, const Consts &c, const P1Own &ra, const P1Own &rb,
                         int ngroups, int sa, int sb, bool wave_has_b, bool want_mag, P1Pair &r)
{
    const int lane = tid & 63;
    const int ga = p1s_group(sa, lane), gb = p1s_group(sb, lane);
    r.a.valid = lane > 0 && ga < ngroups;
    r.b.valid = wave_has_b && lane > 0 && gb < ngroups;
    r.a.p = 16 * (ga < 0 ? 0 : ga);
    r.b.p = 16 * (gb < 0 ? 0 : gb);
    r.a.mag = r.b.mag = 0;
    uint32_t off[17];
    float tha[17];
    p1s_front<0>(ex, tid, t, c, ra, off);
    p1s_gather(t, off, tha);
    if (want_mag) r.a.mag = p1s_magnitude(ra);
    if (wave_has_b) {
        float thb[17];
        p1s_front<9>(ex, tid, t, c, rb, off);
        p1s_gather(t, off, thb);
        if (want_mag) r.b.mag = p1s_magnitude(rb);
        tha[0] = u2f(ex.template shr1<8>(tid, f2u(tha[16])));
        p1_make(t, c, tha, r.a);
        thb[0] = u2f(ex.template shr1<17>(tid, f2u(thb[16])));
        p1_make(t, c, thb, r.b);
    } else {
        tha[0] = u2f(ex.template shr1<8>(tid, f2u(tha[16])));
        p1_make(t, c, tha, r.a);
    }
}

constexpr int PIPE_OTHERS = WB_THREADS - 64;   // lanes of waves 1-3
// chunks of up to 6 x 63 groups leave wave 0 to the IIR alone; longer ones give it a seventh pass
constexpr bool W0_CAN_SHARE = WBFM_CHUNK / 16 > 6 * P1S_PER_WAVE;
static_assert(WBFM_CHUNK / 16 <= 7 * P1S_PER_WAVE, "a chunk must fit seven wave passes of 63 groups");

template <bool GATED, bool MAG, class Exec>
IQD_DEV void wbfm_tile_pipe(Exec &ex, const WbfmTile &t, const Consts &c, WbfmLds &lds,
                            const WbfmStart &start, WbfmRecord *rec_out)
{
    const int exact_from = start.cold ? -COLD_HALO : -start.back;   // (see wbfm_tile)
    const int halo = start.cold ? COLD_HALO : (start.back + SEG - 1) / SEG * SEG;
    ex.all([&](int tid) {
        if (tid < 4) lds.y1[tid] = 0;
        if (tid < 20) lds.y2[tid] = 0;
        if (tid < 4) lds.whist[tid] = 0;
        if (tid < WBFM_CHUNK / SEG + 2) lds.mag[tid] = 0;
        if (tid == 0) {
            lds.y_carry = start.cold ? 0.f : start.y;
            lds.u_carry = start.cold ? 0.f : start.u;
            lds.repair_count = 0;
            lds.sync_ctr = 0;
            lds.y2_peak = 0;
            lds.y2_peak_hist = 0;
        }
    });
    WbfmRecPos rp{t.tlen - FORCED_BACK, exact_from, exact_from};
    WbfmRecord rec;
    rec.y_in = start.y;
    rec.y_out = start.y;
    rec.u_out = start.u;

    typename Exec::template Local<P1Pair> regs;
    typename Exec::template Local<P1Own> raw0;   // wave 0's prefetched group
    auto chunk_len = [&](int cs) { return wbfm_chunk_len(t, cs, WBFM_CHUNK, exact_from); };
    const bool tiny_ok = t.k_min >= 1.0f;
#ifdef IQD_ABL_NOMAG
    const bool mag_on = false;
#else
    const bool mag_on = MAG;
#endif

    // iteration -1 only produces chunk 0; iteration k runs the IIR of chunk k, phase 1 of chunk k+1 and the
    // last two decimation stages of chunk k-1; one more iteration drains the last chunk's decimation
    int cstart = -halo - 1, clen = 0;          // "no current chunk"
    int next_cstart = -halo, next_clen = chunk_len(-halo);
    int st_cstart = 0, st_clen = 0;            // chunk whose stage-1 output awaits stages 2 and 3
    int parity = 0;
    bool has_cur = false, has_next = true;
    while (has_cur || has_next || st_clen) {
        const IirShape sh = iir_shape(clen, cstart == -halo ? halo + exact_from : 0);
        const int nseg = sh.nseg;
        const ChunkBlocks ccb = chunk_blocks(t, cstart), ncb = chunk_blocks(t, next_cstart);
        const bool cur_mag = mag_on && has_cur && cstart >= 0;
        const bool next_mag = mag_on && has_next && next_cstart >= 0;
        const bool w0_share = W0_CAN_SHARE && has_next && (next_clen >> 4) > P1S_PER_WAVE * 6;   // slot 6: groups 378 ..
        WbfmTile tn = t;   // phase 1 of the next chunk runs with that chunk's gain
        tn.k = wbfm_chunk_gain(t, next_cstart < exact_from ? exact_from : next_cstart);   // (the gain at the first sample that counts)
        // ---- X ----
        ex.stamp(7);
        if (ex.in_wave0()) {
            if (w0_share)   // fetch the raw bytes now: the loads fly during the IIR
                ex.wave0([&](int lane) { raw0.at(lane) = p1s_load<GATED>(t, next_cstart, next_clen >> 4, p1s_group(6, lane)); });
            if (has_cur) {
                wbfm_rec_at_start(rp, rec, lds, cstart);
                const IirMark mark = wbfm_mark_in_chunk(rp, cstart, clen);
                ex.wave0([&](int lane) { iir_guess(c, lds, nseg, lane, lds.part); });
                ex.wave0([&](int lane) { iir_warm(c, lds, nseg, lane, sh.first_skip); });
                int rounds = 0;
                do {
                    ex.wave0([&](int lane) { iir_real(c, lds, nseg, lane, t.bounded != 0, sh.first_skip, sh.last_len, mark); });
                    rounds++;
                } while (!ex.wave0_all([&](int lane) { return iir_check(lds, nseg, lane, tiny_ok); }));
                wbfm_rec_in_chunk(rp, rec, lds, cstart, clen, mark);
                if (start.cold && cstart < 0) rec.y_in = lds.e[(COLD_HALO - FORCED_BACK) / SEG - 1];
                ex.wave0([&](int lane) {
                    if (lane == 0) {
                        lds.y_carry = lds.e[nseg - 1];
                        lds.u_carry = t_at(lds, nseg - 1, sh.last_len - 1);
                        lds.repair_count += (uint32_t)(rounds - 1);
                    }
                });
            }
            ex.stamp(1);
            if (w0_share)
                ex.wave0([&](int lane) {
                    p1s_compute(ex, lane, tn, c, raw0.at(lane), raw0.at(lane), next_clen >> 4, 6, 6, false, next_mag, regs.at(lane));
                });
        }
        if (cur_mag || st_clen)
            ex.others([&](int tid) {
                if (cur_mag) wbfm_flush_mag(t, lds, ccb, cstart, clen, tid - 64);
                if (st_clen) wbfm_stage2(c, lds, st_clen, tid - 64, PIPE_OTHERS);
            });
        if (st_clen) {
            ex.others_sync(&lds.sync_ctr);   // stage 3 reads what all three waves' stage 2 wrote
            ex.others([&](int tid) {
                wbfm_stage3(c, lds, t, st_cstart, st_clen, tid - 64, PIPE_OTHERS);
                const int n1 = st_clen >> 2;
                if (tid >= 100 && tid < 104) lds.y1[tid - 100] = lds.y1[(n1 >> 1) + tid - 100];
            });
        }
        ex.stamp(0);
        if (has_next)
            ex.others([&](int tid) {
                const int w = (tid - 64) >> 6, lane = tid & 63;    // waves 1-3 take slots w and w+3
                const int ng = next_clen >> 4;
                if (P1S_PER_WAVE * w >= ng) { regs.at(tid).a.valid = 0; regs.at(tid).b.valid = 0; return; }   // idle wave
                const bool wave_has_b = P1S_PER_WAVE * (w + 3) < ng;
                const P1Own ra = p1s_load<GATED>(t, next_cstart, ng, p1s_group(w, lane));
                const P1Own rb = wave_has_b ? p1s_load<GATED>(t, next_cstart, ng, p1s_group(w + 3, lane)) : ra;
                p1s_compute(ex, tid, tn, c, ra, rb, ng, w, w + 3, wave_has_b, next_mag, regs.at(tid));
            });
        ex.stamp(2);
        ex.sync();
        ex.stamp(3);
        // ---- Y ----
        ex.all([&](int tid) {
            if (st_clen) wbfm_shift_b(lds, st_clen, tid);   // that chunk's stage 3 finished in X
            if (has_next) {
                const bool w0 = tid < 64;
                if (!w0 || w0_share) {
                    P1Pair &r = regs.at(tid);
                    p1_store_t(lds, r.a);
                    p1_store_part(lds.part, r.a);
                    if (next_mag && r.a.valid) p1_add_mag(t, lds, ncb, r.a.p, r.a.mag);
                    if (!w0) {
                        p1_store_t(lds, r.b);
                        p1_store_part(lds.part, r.b);
                        if (next_mag && r.b.valid) p1_add_mag(t, lds, ncb, r.b.p, r.b.mag);
                    }
                }
            }
            if (has_cur) {
                wbfm_stage1(c, lds, clen, tid, parity ^ 1);
                if (tid == 110) {   // the last w group of this chunk, for the next chunk's stage 1
                    const u32x2 last = w_group(lds, (clen >> 2) - 1);
                    lds.whist[2 * parity] = last.x;
                    lds.whist[2 * parity + 1] = last.y;
                }
            }
        });
        ex.stamp(4);
        // advance
        st_cstart = cstart;
        st_clen = has_cur ? clen : 0;
        if (has_cur) parity ^= 1;
        has_cur = has_next;
        cstart = next_cstart;
        clen = next_clen;
        next_cstart = cstart + clen;
        has_next = has_cur && next_cstart < t.tlen;
        next_clen = has_next ? chunk_len(next_cstart) : 0;
    }
    ex.sync();
    if (ex.in_wave0() && rec_out) {
        rec.back_out = t.tlen - rp.pos;
        rec.y_end = lds.y_carry;
        rec.u_end = lds.u_carry;
        rec.pad[0] = rec.pad[1] = 0;
        ex.wave0([&](int lane) { if (lane == 0) *rec_out = rec; });
    }
}

}  // namespace iqd
