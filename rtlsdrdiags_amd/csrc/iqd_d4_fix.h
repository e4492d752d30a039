// FM / AM / SSB streaming pipelines without their long lead-ins (round 6): boundary records and the fix-up that uses them.
//
// Every stage of these chains is a FIR, so a segment that starts cold is exact once its filters' windows lie inside what it has
// run itself.  Rounds 2-5 gave EVERY segment the lead-in that takes (AM 384, FM 768, SSB 1280 samples: the last decimator's 16 or
// 40 taps at 16 kS/s, the Hilbert transformer's 31 at 8 kS/s) - 7 / 14 / 23 % of a 5.5 k-sample segment, a third to a half of the
// pieces at the reference's own operating point (one 64 ms block per call, DataConsumer.cc:333-346).  Now every segment runs a
// lead-in of D4_HALO_SHORT = 128 samples (which the first two stages need), leaves a *record* of the few intermediate values
// its successor's first outputs reach back for, and those outputs - 4 (AM), 34 (SSB), 18 (FM) per boundary - are recomputed.
// WHERE: a channel's segments have consecutive segment ids, i.e. they sit in neighbouring lanes of a consumer wave, so at the end
// of its run every lane takes the lane below's end state (DPP) and replays its own first outputs from inputs it kept in LDS
// (iqd_stream2.hip: d4_am_wave, d4_fm_wave) - no memory traffic.  Only where the predecessor sits in another wave (every 64th
// segment id) the two lanes leave the records below in global memory and the launch that closes the step recomputes those
// outputs (this file; before the DC-removal pass reads the detector stream / beside the tail update).  The reference carries exactly this state from call to call in its filters' ring
// buffers (Decimator_int16.cc:310-351, FirFilter_int16.cc:151-213); here it travels from segment to segment of one call.
//
// A channel's FIRST segment has no predecessor in the call: its histories come from the kept raw tail, as before.  So that all
// segments of a launch still run the same number of pieces (they advance in lock step), segment t of a channel covers the
// samples [t * tile_len - D, (t + 1) * tile_len - D) with D = full lead-in - 128 (D4Args::lead_shift): segment 0 starts D samples
// BEFORE the call's first sample - those and its 128-sample lead-in are the old full lead-in, read from the tail - and what it
// computes for positions before 0 is not stored.
//
// Piece numbering below: a segment's run is pieces 0 .. N-1 of 32 samples, N = (128 + tile_len) / 32; pieces 0-3 are its lead-in,
// piece 4 + j produces its output j (one PCM / detector sample per piece).  A record holds, as int16 pairs (older sample in the
// low half, the pipelines' register layout):
//   AM / SSB  y2 (16 kS/s, both rails): the pairs of pieces 4-7 ("head") and of the last 7 pieces ("tail")
//   SSB       the 8 kS/s rails i, q of pieces 8-39 (head) and of the last 32 pieces (tail)
//   FM        y2 (16 kS/s): the pairs of pieces 4-23 (head) and of the last 20 pieces (tail)
// Host + device (tests/emu steps the fix-up against the oracle on the CPU tier).
#pragma once
#include <stdint.h>

#include "iqd_device.h"
#include "iqd_stream.h"
#include "iqd_prims.h"
#include "iqd_wbfm.h"     // q15_seq
#include "iqd_chains.h"   // q15_free

namespace iqd {

constexpr int D4_LEAD_PIECES = D4_HALO_SHORT / 32;
constexpr int D4_FIX_AM = 4, D4_FIX_SSB = 34, D4_FIX_FM = 18;   // outputs per boundary that reach into the predecessor

struct D4RecAm {
    uint32_t y2_head[2][4];      // [rail i, q][pieces 4..7]
    uint32_t y2_tail[2][8];      // [rail][pieces N-7 .. N-1], [7] unused
};
struct D4RecSsb {
    D4RecAm y2;
    uint32_t head_i[16], head_q[16];   // 8 kS/s rails of pieces 8..39, int16 pairs
    uint32_t tail_i[16], tail_q[16];   // ... of pieces N-32 .. N-1
};
struct D4RecFm {
    uint32_t y2_head[20];        // pieces 4..23
    uint32_t y2_tail[20];        // pieces N-20 .. N-1
};
static_assert(sizeof(D4RecAm) == 96 && sizeof(D4RecSsb) == 352 && sizeof(D4RecFm) == 160, "records are whole 16-byte units");

constexpr uint32_t d4_rec_bytes(int family) { return family == FAM_FM ? sizeof(D4RecFm) : family == FAM_AM ? sizeof(D4RecAm) : sizeof(D4RecSsb); }

// Segment t of a channel that consumes vlen samples in the call: first sample, length (0: not there).
struct D4Span { int32_t v0, tlen; };
IQD_DEV D4Span d4_span(uint32_t tile, uint32_t tile_len, uint32_t shift, uint32_t vlen)
{
    const int64_t v0 = (int64_t)tile * tile_len - (int64_t)shift;
    if (vlen == 0 || v0 >= (int64_t)vlen) return D4Span{0, 0};
    const int64_t rest = (int64_t)vlen - v0;
    return D4Span{(int32_t)v0, (int32_t)(rest < (int64_t)tile_len ? rest : (int64_t)tile_len)};
}
IQD_DEV uint32_t d4_tiles(uint32_t tile_len, uint32_t shift, uint32_t vlen)
{
    return vlen ? (uint32_t)(((uint64_t)vlen + shift + tile_len - 1) / tile_len) : 0u;
}

// ---- the fix-up of one channel ------------------------------------------------------------------------------
// Staging in LDS (plain arrays on the host), FIX_BATCH boundaries at a time: the FIR windows as contiguous int16 arrays, so
// that the taps run over them with the chains' own q15_free / q15_seq.
constexpr int D4_FIX_BATCH = 12;
struct D4FixLds {
    uint32_t y2[D4_FIX_BATCH][2][12];      // AM / SSB: y2 pairs of pieces -3 .. 7 per rail ([11] unused); FM: see fm[]
    uint32_t ri[D4_FIX_BATCH][32];         // SSB: rail i of pieces -26 .. 37
    uint32_t rq[D4_FIX_BATCH][32];
};
struct D4FixFmLds {
    uint32_t y2[D4_FIX_BATCH][40];         // y2 pairs of pieces -15 .. 21 ([37..39] unused)
};

IQD_DEV int am_detector(int iv, int qv)    // AmDemodulator.cc:446-459
{
    const int im = (int)(int16_t)(iv < 0 ? -iv : iv), qm = (int)(int16_t)(qv < 0 ? -qv : qv);
    return (int)(int16_t)((im > qm) ? im + (qm >> 1) : qm + (im >> 1));
}

// AM and SSB.  `rec` = the channel's records (one per segment, in segment order), n_tiles >= 2, out = the channel's detector
// stream (8 kS/s, one int per PCM sample, stride out_stride).  Phases separated by the machine's barrier (ex.all).
template <int FAMILY, class Exec>
IQD_DEV void d4_fix_am_ssb(Exec &ex, const Consts &c, D4FixLds &lds, const void *rec_v, uint32_t n_tiles, uint32_t tile_len,
                           uint32_t shift, uint32_t vlen, int lsb, int32_t *out, size_t out_stride, int nthr, uint32_t t_first = 1, uint32_t t_step = 1)
{
    constexpr bool SSB = FAMILY == FAM_SSB;
    constexpr uint32_t STRIDE = SSB ? sizeof(D4RecSsb) : sizeof(D4RecAm);
    const uint8_t *rec = (const uint8_t *)rec_v;
    // the boundaries to do: in front of the segments t_first, t_first + t_step, ... (all of them: 1, 1; the pipelines fix what lies
    // inside a consumer wave themselves and leave every 64th segment id's - d4_am_wave)
    const uint32_t n_todo = t_first < n_tiles ? (n_tiles - t_first + t_step - 1) / t_step : 0u;
    for (uint32_t k0 = 0; k0 < n_todo; k0 += D4_FIX_BATCH) {
        const int nb = (int)(n_todo - k0 < (uint32_t)D4_FIX_BATCH ? n_todo - k0 : (uint32_t)D4_FIX_BATCH);
        auto tile_of = [&](int b) { return t_first + (k0 + (uint32_t)b) * t_step; };
        ex.all([&](int tid) {   // windows from the records: the predecessor's tail, then this segment's head
            for (int it = tid; it < nb * 2 * 11; it += nthr) {
                const int b = it / 22, r = (it % 22) / 11, k = it % 11;
                const D4RecAm *own = (const D4RecAm *)(rec + (size_t)tile_of(b) * STRIDE), *pred = (const D4RecAm *)(rec + (size_t)(tile_of(b) - 1) * STRIDE);
                lds.y2[b][r][k] = k < 7 ? pred->y2_tail[r][k] : own->y2_head[r][k - 7];
            }
            if (SSB)
                for (int it = tid; it < nb * 2 * 30; it += nthr) {   // rails of pieces -26..3 (dwords 0..14) and 8..37 (dwords 17..31)
                    const int b = it / 60, r = (it % 60) / 30, k = it % 30;
                    const D4RecSsb *own = (const D4RecSsb *)(rec + (size_t)tile_of(b) * STRIDE), *pred = (const D4RecSsb *)(rec + (size_t)(tile_of(b) - 1) * STRIDE);
                    uint32_t *dst = r ? lds.rq[b] : lds.ri[b];
                    if (k < 15) dst[k] = (r ? pred->tail_q : pred->tail_i)[k + 1];
                    else dst[k + 2] = (r ? own->head_q : own->head_i)[k - 15];
                }
        });
        ex.all([&](int tid) {   // the rails of pieces 4..7 (/2, 16 taps over y2: AmDemodulator.cc:388-398) - AM: the detector right away
            for (int it = tid; it < nb * 4; it += nthr) {
                const int b = it >> 2, j = it & 3;
                const D4Span sp = d4_span(tile_of(b), tile_len, shift, vlen);
                const int iv = q15_free<16>(c.am_s3, lds.y2[b][0], 2 * (j + 7) + 1), qv = q15_free<16>(c.am_s3, lds.y2[b][1], 2 * (j + 7) + 1);
                if (!SSB) {
                    if (32 * j < sp.tlen) out[(size_t)((sp.v0 >> 5) + j) * out_stride] = am_detector(iv, qv);
                } else {   // piece 4 + j sits at int16 index 30 + j of the rail windows
                    put_i16(lds.ri[b], 30 + j, iv);
                    put_i16(lds.rq[b], 30 + j, qv);
                }
            }
        });
        if (SSB)
            ex.all([&](int tid) {   // SsbDemodulator.cc:574-588: -i[n-15] -+ Hilbert31(q)
                for (int it = tid; it < nb * D4_FIX_SSB; it += nthr) {
                    const int b = it / D4_FIX_SSB, j = it % D4_FIX_SSB;
                    const D4Span sp = d4_span(tile_of(b), tile_len, shift, vlen);
                    if (32 * j >= sp.tlen) continue;
                    const int idl = q15_free<16>(c.ssb_delay, lds.ri[b], 30 + j);
                    const int qh = q15_free<31>(c.ssb_hilbert, lds.rq[b], 30 + j);
                    out[(size_t)((sp.v0 >> 5) + j) * out_stride] = lsb ? (int)(int16_t)idl - (int)(int16_t)qh : (int)(int16_t)idl + (int)(int16_t)qh;
                }
            });
    }
}

// FM: the first 18 PCM samples of every segment but the channel's first (/2, 40 taps with the per-MAC clamp, FmDemodulator.cc:548-556).
template <class Exec>
IQD_DEV void d4_fix_fm(Exec &ex, const Consts &c, D4FixFmLds &lds, const D4RecFm *rec, uint32_t n_tiles, uint32_t tile_len, uint32_t shift,
                       uint32_t vlen, int16_t *pcm_row, int nthr, uint32_t t_first = 1, uint32_t t_step = 1)
{
    const uint32_t n_todo = t_first < n_tiles ? (n_tiles - t_first + t_step - 1) / t_step : 0u;
    for (uint32_t k0 = 0; k0 < n_todo; k0 += D4_FIX_BATCH) {
        const int nb = (int)(n_todo - k0 < (uint32_t)D4_FIX_BATCH ? n_todo - k0 : (uint32_t)D4_FIX_BATCH);
        auto tile_of = [&](int b) { return t_first + (k0 + (uint32_t)b) * t_step; };
        ex.all([&](int tid) {   // pairs of pieces -15..3 from the predecessor's tail ([1..19]), 4..21 from this segment's head
            for (int it = tid; it < nb * 37; it += nthr) {
                const int b = it / 37, k = it % 37;
                lds.y2[b][k] = k < 19 ? rec[tile_of(b) - 1].y2_tail[k + 1] : rec[tile_of(b)].y2_head[k - 19];
            }
        });
        ex.all([&](int tid) {
            for (int it = tid; it < nb * D4_FIX_FM; it += nthr) {
                const int b = it / D4_FIX_FM, j = it % D4_FIX_FM;
                const D4Span sp = d4_span(tile_of(b), tile_len, shift, vlen);
                if (32 * j >= sp.tlen) continue;
                pcm_row[(sp.v0 >> 5) + j] = (int16_t)q15_seq<40>(c.audio40, lds.y2[b], 2 * (j + 19) + 1);
            }
        });
    }
}

}  // namespace iqd
