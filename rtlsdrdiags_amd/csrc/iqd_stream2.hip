// FM, AM and SSB chains as streaming pipelines (gfx950), the shape of iqd_stream.hip without the table and without
// a full-rate recurrence: one persistent 15-wave workgroup per CU, 192 segments in lock step, 32 samples (a piece)
// at a time.
//
//   12 P waves, 16 segments each: raw bytes (four pieces of input in flight per wave, iqd_mfma.h: gload16_untracked)
//       -> squelch magnitude of the raw samples, one quad-SAD per dword (rounds 3-4: through a 68 KiB table in LDS; since round 5
//       that LDS holds rings of 16 pieces instead of 8, IQD_D4_MAGLUT / IQD_D4_SLOTS) -> signed, rotation signs (SDWA) -> the
//       chain's first /4 decimator on both rails as v_mfma_i32_16x16x64_i8 (FM: 32-tap tuner filter, AM/SSB: 8 taps;
//       the part of the window that lies in the previous piece through a second, chained MFMA) ->
//       AM/SSB: 8 + 8 int16 outputs per piece into the ring
//       FM: the exact phase angle of each output (283 x 283 table in global memory, L2-resident), then the
//           discriminator right here, two outputs per lane: theta[m] - theta[m-2] (the lane 16 below holds theta[m-2]),
//           branch cut, K, (int16) - one dword per lane into the ring
//   3 consumer waves, 64 segments each (one per lane): the remaining stages with their histories in registers
//       AM   /4 (12 taps) /2 (16 taps) on both rails -> max + min/2 -> detector input at 8 kS/s
//       SSB  the same -> -i[n-15] -+ Hilbert31(q) -> detector input at 8 kS/s
//       FM   /4 (12 taps) -> /2 (40 taps) -> PCM (the P waves' stream runs two samples early: one pair is kept a piece)
//   AM/SSB stop in front of the 8 kS/s DC-removal IIR like the tile kernel does (dc_* kernels run it exactly).
//
// Every stage is a FIR: a segment rebuilds its histories over a lead-in (AM 384, FM 768, SSB 1280 samples) and is
// exact by construction - no verification, no records.  Segments of channels with different rotation selectors sit
// in groups padded to 16, so that a P wave's 16 segments always share one (its tap matrices and byte negations; the
// piece loop is instantiated per selector).
//
// Reference: FmDemodulator.cc:376-560, AmDemodulator.cc:339-504, SsbDemodulator.cc:462-598 behind
// IqDataProcessor.cc:735-749.  hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "iqd_kernels.h"
#include "iqd_stream.h"
#include "iqd_wbfm.h"
#include "iqd_mfma.h"
#include "iqd_taps.h"

// build-time experiments (tools/variant.sh): all off in the shipped library
#ifndef IQD_D4_SPLIT
#define IQD_D4_SPLIT 0
#endif
#ifndef IQD_D4_PRIO
#define IQD_D4_PRIO 0
#endif
#ifndef IQD_D4_TIMING
#define IQD_D4_TIMING 0
#endif
// 1: a segment's four lanes LOAD as neighbours (lane 4 c + k: 64 contiguous bytes per segment and load instruction) and four
// ds_bpermute put the bytes where the matrix instruction wants them (lane 16 k + c).  Built in round 3 and then ±2 %: the LDS
// pipe was busy with the magnitude table's gathers.  With those gone (IQD_D4_MAGLUT=3) the permutes are cheap and the better
// read pattern shows: FM 0.2439 -> 0.2207 ms, USB 0.1961 -> 0.1849, AM 0.1731 -> 0.1707 (profiles/r5_d4_ring_depth.txt).  0: the A/B.
#ifndef IQD_D4_TRANSPOSE
#define IQD_D4_TRANSPOSE 1
#endif
// Squelch magnitudes of the P waves: 3 = one quad-SAD per dword on the raw bytes (iqd_mfma.h: st_mag_raw_chunk), no table; 1 = the
// 68 KB table in LDS of rounds 3-4.  Round 5 (profiles/r5_d4_ring_depth.txt, one box, interleaved): these kernels do not wait for
// issue slots - 32 idle vector instructions per piece in the P waves or in the consumer waves cost +0.5 % - so the SADs' extra
// instructions are free, and the LDS the table leaves behind holds rings of 16 pieces instead of 8: AM 0.1913 -> 0.1826 (no
// table) -> 0.1821 (16 slots), USB 0.2057 -> 0.1982 -> 0.1915, FM 0.2495 -> 0.2443 -> 0.2435 ms.
#ifndef IQD_D4_MAGLUT
#define IQD_D4_MAGLUT 3
#endif
#ifndef IQD_D4_WAITSTAT
#define IQD_D4_WAITSTAT 0
#endif
// The FM consumer lanes' decimator taps as literals of the v_dot2c instructions (iqd_taps.h: STREAM_TAPS, the designs are fixed)
// instead of 26 scalar registers (the FM kernels had 33-35 of their scalar registers spilled to vector lanes).
// IQD_D4_TAPS_IN_SGPRS: from the kernel arguments as before (the A/B).
#ifdef IQD_D4_TAPS_IN_SGPRS
#define D4_TAP(WHICH, Q) da.WHICH[Q]
#else
#define D4_TAP(WHICH, Q) (uint32_t)taps::STREAM_TAPS.WHICH[Q]
#endif
#ifndef IQD_D4_SLEEP_P      // s_sleep argument (x 64 cycles) between two looks at a ring counter: P waves / consumer waves
#define IQD_D4_SLEEP_P 1
#endif
#ifndef IQD_D4_SLEEP_C
#define IQD_D4_SLEEP_C 1
#endif

namespace iqd {

#ifndef IQD_D4_SLOTS
#define IQD_D4_SLOTS 16
#endif
constexpr int D4_SLOTS = IQD_D4_SLOTS;            // ring depth in pieces: whole quads, a power of two of them
constexpr int D4_QUADS = D4_SLOTS / 4;            // the waves shake hands once per quad (4 pieces = 128 samples), not per piece
// per ring a "full" counter per quad of slots (each of the 4 P waves adds 1 when it has stored the quad's last piece), then
// a "consumed" counter per ring (quads the consumer wave has read)
constexpr int D4_SYNC_WORDS = ST_RINGS * D4_QUADS + 8;
static_assert(D4_SLOTS % 4 == 0 && (D4_QUADS & (D4_QUADS - 1)) == 0 && D4_QUADS >= 2, "ring depth");
// Pieces of input a P wave keeps in flight: 4.  (FM's two angle loads per piece are counted with them.  IQD_D4_AHEAD_AM=8: AM / SSB
// with eight - built and bit-exact in round 5 on the thought that kernels which do not wait for issue slots wait for memory; they
// do not: AM 0.1735 -> 0.1811 ms, USB 0.1931 -> 0.1936, profiles/r5_d4_ring_depth.txt.)
#ifndef IQD_D4_AHEAD_AM
#define IQD_D4_AHEAD_AM 4
#endif
template <int MODE> constexpr int d4_ahead() { return MODE == D4_FM ? 4 : IQD_D4_AHEAD_AM; }
constexpr int D4_SLOT_BYTES = 64 * 32;            // 64 segments x (4 lanes x 8 bytes) per piece
constexpr int D4_MAGLUT_OFF = (ST_RINGS * D4_SLOTS * D4_SLOT_BYTES + D4_SYNC_WORDS * 4 + 15) & ~15;   // squelch magnitude table (iqd_mfma.h)
// Round 6 (iqd_stream.h, d4_geom): every consumer lane keeps the inputs of its segment's first outputs here - the y2 pairs of pieces 4..7
// (AM / SSB, both rails: 8 words), SSB's 8 kS/s rails of pieces 8..39 (32 words), FM's y2 pairs of pieces 4..23 (20 words) - and
// replays those outputs at the end of its run with its predecessor's end state, which sits in the lane below.  [word][segment].
constexpr int D4_HEAD_WORDS = 40;   // per segment: AM 2 x 4 pairs; SSB those + 32 rails; FM 20 pairs
static_assert(D4_HEAD_WORDS >= 2 * D4_REPLAY_PAIRS_AMSSB + (D4_REPLAY_SSB - D4_REPLAY_PAIRS_AMSSB) && D4_HEAD_WORDS >= D4_REPLAY_FM, "head store");
static_assert(D4_RAILS_FROM_PIECE == 4 + D4_REPLAY_PAIRS_AMSSB && D4_REPLAY_AM == D4_REPLAY_PAIRS_AMSSB, "replay layout");
constexpr int D4_HEAD_OFF = D4_MAGLUT_OFF + (IQD_D4_MAGLUT == 1 ? ST_MAGLUT_BYTES : 0);
constexpr int D4_LDS_BYTES = D4_HEAD_OFF + D4_HEAD_WORDS * ST_SEGS * 4;
static_assert(D4_LDS_BYTES <= 160 * 1024, "rings + head store must fit the CU's LDS");

struct D4Seg {
    uint32_t valid, li, tile, ch, ech;
    int32_t v0, tlen;
    int32_t vlen;          // samples this channel consumes in the call: all of them, or those of its open blocks (squelch)
    int rot;
    int32_t skip;          // a cold segment (a channel's first, a consumer wave's lane 0: iqd_stream.h, d4_geom) stores nothing before v0 + skip
    uint32_t cold;
};

// segment id -> (rotation group, channel, tile).  Groups in the order +Fs/4, none, -Fs/4, each padded to 16 ids.
__device__ __forceinline__ D4Seg d4_segment(const ChainLaunch &a, const D4Args &da, uint32_t sid)
{
    D4Seg s;
    const uint32_t r = (sid >= da.group_start[1] ? 1u : 0u) + (sid >= da.group_start[2] ? 1u : 0u);
    const uint32_t local = sid - da.group_start[r];
    s.rot = 1 - (int)r;
    s.valid = sid < da.group_start[3] && local < da.group_nseg[r] ? 1u : 0u;
    const uint32_t id = s.valid ? local : 0u;
    s.li = da.group_li0[r] + id / a.tiles_per_ch;
    if (!s.valid) s.li = 0;
    s.tile = id % a.tiles_per_ch;
    s.ch = a.ch_list[s.li];
    s.ech = a.first_ch + s.ch;
    // a squelch-gated call: the channel's chain sees the concatenation of its open blocks (IqDataProcessor.cc:793), a
    // virtual stream of vlen_gated[ch] samples; the segments are cut on that axis, those beyond its end are not there
    const uint32_t vlen = a.vlen_gated ? a.vlen_gated[s.ch] : a.vlen;
    s.vlen = (int32_t)vlen;
    // (round 6, iqd_stream.h: d4_geom - warm segments run 128 samples of lead-in and take their predecessor's state from the lane
    //  below; cold ones start lead_shift samples before their first output, the channel's first one in the kept tail)
    const D4Geom g = d4_geom(sid, s.tile, a.tile_len, da.lead_shift);
    s.skip = (int32_t)g.skip;
    s.cold = g.cold;
    const int64_t v0 = g.v0;
    if (vlen == 0 || v0 >= (int64_t)vlen) { s.valid = 0; s.v0 = 0; s.tlen = 0; return s; }
    s.v0 = (int32_t)v0;
    const int64_t rest = (int64_t)vlen - v0;
    s.tlen = s.valid ? (int32_t)(rest < (int64_t)a.tile_len ? rest : (int64_t)a.tile_len) : 0;
    return s;
}

__device__ __forceinline__ uint32_t d4_ring_off(uint32_t row, uint32_t g) { return row * 32u + ((g ^ (row & 3u)) << 3); }

// ---- P wave ---------------------------------------------------------------------------------------------
// One round of one P wave: its 16 segments from lead-in to end, with the rotation as a template parameter so that the
// piece loop is straight-line code (the selector is uniform over the wave: groups are padded to 16).
// A lane has two identities.  It LOADS as lane 4 c + k: bytes 16 k .. 16 k + 15 of segment c's piece, so that the four
// lanes of a segment are neighbours and ask for 64 contiguous bytes (tools/ubench/streams.hip: 4.4 TB/s against 2.9
// when they sit 16 lanes apart, which is how the matrix instruction wants them); squelch magnitudes are taken in this
// arrangement.  Four ds_bpermute then put the bytes where v_mfma expects them - lane 16 k + c - and everything behind
// the matrix instruction (sg, g) belongs to that arrangement.
// GATED: the launch follows a squelch pre-pass and some channel lost blocks: virtual sample v of a channel lives in its
// open block number v / block_samples (ChainLaunch::blk_lists).  A lane keeps the block it is in (virtual range and
// where its bytes are) and looks the next one up when it leaves it - once per block_samples / 32 pieces.
template <int MODE, bool MAG, int ROT, bool GATED>
__device__ __forceinline__ void d4_p_round(const ChainLaunch &a, const D4Args &da, const D4Seg &sg, const D4Seg &sgl, uint8_t *ring_base,
                                           const uint32_t *full, const uint32_t *consumed, uint32_t *sync, uint32_t wr_off,
                                           int g, int lane, uint32_t &pg)
{
#if IQD_D4_TRANSPOSE
        const int gl = lane & 3;                                 // k group of the bytes this lane loads
        const int from_lane4 = (4 * (lane & 15) + (lane >> 4)) << 2;   // ds_bpermute address: who loaded this lane's operand
#else
        const int gl = g;
#endif
        const int rot = ROT;
        const v4i cround = {1 << 14, 1 << 14, 1 << 14, 1 << 14}, czero = {0, 0, 0, 0};
        uint32_t zero = 0, four = 4;
        asm volatile("" : "+v"(zero), "+v"(four));
        const int n_pieces = (da.halo + (int)a.tile_len) >> 5;
        const v4i *am = (const v4i *)da.amat + (size_t)(rot + 1) * 4 * 64;
        const v4i A0 = am[0 * 64 + lane], A1 = am[1 * 64 + lane], A2 = am[2 * 64 + lane], A3 = am[3 * 64 + lane];
        const uint8_t *iq_ch = a.iq + (size_t)sgl.ch * a.ch_stride_bytes;
        const int fam = MODE == D4_FM ? FAM_FM : (MODE == D4_AM ? FAM_AM : FAM_SSB);
        const uint8_t *tail = a.tails + ((size_t)sgl.ech * FAM_COUNT + fam) * TAIL_BYTES + TAIL_BYTES;
        const int32_t vlane = sgl.v0 + 8 * gl;
        const uint8_t *base_iq = iq_ch + 2 * (int64_t)vlane, *base_tail = tail + 2 * (int64_t)vlane;
        const int32_t pos_max = sgl.vlen - 8 - vlane;
        const uint32_t *blk_list = GATED ? a.blk_lists + (size_t)sgl.ch * a.n_blocks : nullptr;
        int32_t gb_v0 = 0, gb_v1 = 0;                           // GATED: virtual range of the open block this lane is in
        const uint8_t *gb_base = iq_ch;                          // ... and the address its virtual sample 0 would have
        auto load_piece = [&](int pos) -> v4u {
            const int32_t pc = pos < pos_max ? pos : pos_max;
            if (!GATED) {
#ifdef IQD_D4_PROBE_CACHED   // TIMING PROBE ONLY (wrong results): every lane re-reads a 512-byte window of its segment - the loads hit L2
                return gload16_untracked(base_iq + 2 * (int64_t)(pc & 0xff));
#endif
                const uint8_t *base = pc < -vlane ? base_tail : base_iq;
                return gload16_untracked(base + 2 * (int64_t)pc);
            }
            const int32_t vv = vlane + pc;                       // virtual sample of this lane's 8
            if (vv >= 0 && (vv >= gb_v1 || vv < gb_v0)) {        // (rare) into another open block
                const uint32_t blk = (uint32_t)vv / a.block_samples;
                gb_v0 = (int32_t)(blk * a.block_samples);
                gb_v1 = gb_v0 + (int32_t)a.block_samples;
                gb_base = iq_ch + 2 * ((int64_t)blk_list[blk] * a.block_samples - (int64_t)gb_v0);
            }
            const uint8_t *addr = vv < 0 ? base_tail + 2 * (int64_t)pc : gb_base + 2 * (int64_t)vv;
            return gload16_untracked(addr);
        };
#if IQD_D4_TRANSPOSE
        auto front = [&](uint4 raw) -> uint4 {                   // signed bytes, rotation signs, then to the matrix arrangement
            const uint4 f = st_front<ROT>(raw, zero);
            return uint4{(uint32_t)__builtin_amdgcn_ds_bpermute(from_lane4, (int)f.x), (uint32_t)__builtin_amdgcn_ds_bpermute(from_lane4, (int)f.y),
                         (uint32_t)__builtin_amdgcn_ds_bpermute(from_lane4, (int)f.z), (uint32_t)__builtin_amdgcn_ds_bpermute(from_lane4, (int)f.w)};
        };
#else
        auto front = [&](uint4 raw) -> uint4 { return st_front<ROT>(raw, zero); };
#endif
        uint32_t *mag_row = MAG ? a.mag_sums + (size_t)sgl.ch * a.n_blocks : nullptr;
        const bool mcount = MAG && sgl.valid;
        const int32_t mlimit = sgl.tlen - 8 * gl;
        const int32_t mfirst = sgl.skip;                          // (a cold segment: what lies before its first output is lead-in)
#if IQD_D4_TIMING
        long long t_ring_wait = 0;
#endif
        uint32_t macc = 0;
        uint32_t mblk = sgl.v0 < 0 ? 0u : (uint32_t)(sgl.v0 + 8 * gl) / a.block_samples;
        int32_t minblk = sgl.v0 + 8 * gl - (int32_t)(mblk * a.block_samples);   // position in the block (negative: before the call)

        // A piece into its ring slot.  `sq` (known after unrolling) is the slot's place in its quad: room is checked
        // before a quad's first store - its four slots are free once the consumer has read the quad D4_QUADS back - and
        // the quad is published after its last.
        auto hand_over = [&](u32x2 payload, int sq) {
#ifdef IQD_D4_DYNPRIO
            // A P wave that runs ahead of its ring's consumer yields to the ones that lag (the SIMD serves oldest first: the in-kernel
            // timing shows the youngest P wave of a ring getting what the three older ones leave - 2200 cycles per piece against their
            // 900 - while those then wait for ring space): priority by lag, once per quad, from the counter the wave reads anyway.
            if (sq == 0) {
                const uint32_t seen0 = lds_load_relaxed(consumed);
                if ((int32_t)(pg - seen0) >= IQD_D4_DYNPRIO) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(2);
            }
#endif
            if (sq == 0 && pg >= (uint32_t)D4_QUADS) {
                uint32_t seen = lds_load_relaxed(consumed);
#if IQD_D4_TIMING
                const long long tw0 = clock64();
#endif
                while ((int32_t)(seen - (pg - (D4_QUADS - 1))) < 0) {
#if IQD_D4_WAITSTAT
                    if (lane == 0) atomicAdd(&sync[D4_SYNC_WORDS - 2], 1u);
#endif
                    __builtin_amdgcn_s_sleep(IQD_D4_SLEEP_P);
                    seen = lds_load_relaxed(consumed);
                }
#if IQD_D4_TIMING
                t_ring_wait += clock64() - tw0;
#endif
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            uint8_t *slot = ring_base + ((pg & (D4_QUADS - 1)) * 4 + (uint32_t)sq) * D4_SLOT_BYTES + wr_off;
            if (MODE == D4_FM) *(uint32_t *)slot = payload.x;
            else *(u32x2 *)slot = payload;
            if (sq == 3) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_signal(&full[pg & (D4_QUADS - 1)]);
                pg++;
            }
        };
        // FM: the discriminator runs here too, two outputs per lane (theta[m] - theta[m-2]: the lane 16 below holds
        // theta[m-2], for the first lane group it is what the last one held a piece ago).  The two angles come from the
        // 283 x 283 table in L2, a long way off for an in-order wave: a piece asks for its angles and finishes the
        // piece before it, whose angles have had a whole piece's arithmetic to arrive.
        const int src_lane4 = ((lane - 16) & 63) << 2;
        float last_a = 0.f, last_b = 0.f;
        const float k_now = MODE == D4_FM ? a.params[sg.ech].fm_k : 0.f;
        const GainEpochList *ep = &a.epochs[sg.ech].fm;
        const bool ep_reach = MODE == D4_FM && ep->since[0] < (uint32_t)TAIL && (int64_t)sg.v0 - da.halo - 32 < -(int64_t)ep->since[0];
        {   // have these arrive now: a first use inside the piece loop would put a wait for everything in flight there
            const uint32_t reach = ep_reach;
            asm volatile("" :: "v"(k_now), "v"(reach));
        }
        uint32_t asked_a[2] = {0, 0}, asked_b[2] = {0, 0};
#if IQD_D4_TIMING
        long long t_wait_raw = 0, t_front_mag = 0, t_post_hand = 0;
#endif
        auto fm_finish = [&](int pos, uint32_t ta_bits, uint32_t tb_bits, int sq) {
            const float ta = u2f(ta_bits), tb = u2f(tb_bits);
            const float give_a = g == 3 ? last_a : ta, give_b = g == 3 ? last_b : tb;
            const float before_a = u2f((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)f2u(give_a)));
            const float before_b = u2f((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)f2u(give_b)));
            last_a = ta;
            last_b = tb;
            // theta[m] - theta[m-2] is the reference's discriminator output two tuner samples later
            // (theta[n-2] - theta[n-4], FmDemodulator.cc:113-122, 479): the last lane group's pair belongs to the
            // next piece, gain included.  (A segment whose lead-in reaches back before a gain change runs that part
            // with the gain of the time; changes fall on call boundaries = multiples of 128 samples.)
            float kk = k_now;
            if (ep_reach) kk = epoch_gain_search(ep, k_now, sg.v0 + pos + (g == 3 ? 32 : 0));
            // branch cut, K, cast; (int16) = the low half
            hand_over(u32x2{cast_pack_i16_bounded(kk * wrap_delta(ta - before_a), kk * wrap_delta(tb - before_b)), 0u}, sq);
        };
        // Four pieces of input in flight - a piece's arithmetic is much shorter than a trip to HBM - in four named
        // buffers of a loop unrolled by four: handing a buffer on with register moves would wait for the load it has
        // just issued.  Waits count the loads issued since (iqd_mfma.h): per piece one of these, for FM two angles more.
        constexpr int PER_PIECE = MODE == D4_FM ? 3 : 1;
        constexpr int D4_AHEAD = d4_ahead<MODE>();             // (4 or 8: whole quads; the ring slot of buffer j is j & 3)
        static_assert(D4_AHEAD % 4 == 0 && (PER_PIECE == 1 || D4_AHEAD == 4), "prefetch depth");
        v4u raw_before = load_piece(-da.halo - 32);
        v4u raw[D4_AHEAD];
#pragma unroll
        for (int j = 0; j < D4_AHEAD; j++) raw[j] = load_piece(-da.halo + 32 * j);
        gload_wait<D4_AHEAD>(raw_before);
        uint4 prev = front(as_uint4(raw_before));
        // One piece.  `first` (a type) says whether this is the first trip through the four buffers, where fewer loads
        // have been issued since raw[j]'s: the two cases are separate instantiations, not a branch - a wait chosen at run
        // time made the compiler copy the buffer, still in flight, in front of one of the two waits.
        auto piece = [&](auto first, int q0, int j) {
            const int pos = -da.halo + 32 * (q0 + j);
#ifdef IQD_D4_BURN_SIMD   // measurement build: the P waves of ONE SIMD (hardware wave % 4) issue IQD_D4_BURN_N idle vector instructions per piece
            if (((int)(threadIdx.x >> 6) & 3) == IQD_D4_BURN_SIMD) {
                float burn = 1.0f;
#pragma unroll
                for (int k = 0; k < IQD_D4_BURN_N; k++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(burn));
            }
#endif
#if IQD_D4_TIMING
            const long long tA = clock64();
#endif
            // younger than raw[j]: the other three buffers' loads, plus the angles asked for since its own issue (in
            // the first trip: since the start)
            if (PER_PIECE == 1 || !decltype(first)::value) gload_wait<D4_AHEAD - 1 + (PER_PIECE - 1) * D4_AHEAD>(raw[j]);
            else gload_wait_n(raw[j], D4_AHEAD - 1 + (PER_PIECE - 1) * j);
            const uint4 rawj = as_uint4(raw[j]);
#if IQD_D4_TIMING
            const long long tB = clock64();
            t_wait_raw += tB - tA;
#endif
            const uint4 cur = front(rawj);
            const v4i bc = {(int)cur.x, (int)cur.y, (int)cur.z, (int)cur.w};
            const v4i bp = {(int)prev.x, (int)prev.y, (int)prev.z, (int)prev.w};
            v4i lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, bc, cround, 0, 0, 0);
            v4i hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, bc, czero, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2, bp, lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A3, bp, hi, 0, 0, 0);
            if (MAG && pos >= 0) {
#if IQD_D4_MAGLUT == 1
                const uint32_t m = st_maglut_chunk_at((uint32_t)D4_MAGLUT_OFF, rawj, four);
#elif IQD_D4_MAGLUT == 2   // masked SADs on the raw bytes (iqd_mfma.h: st_mag_raw_dword), no table
                uint32_t m16 = st_mag_raw_dword(rawj.x, 0u);
                m16 = st_mag_raw_dword(rawj.y, m16);
                m16 = st_mag_raw_dword(rawj.z, m16);
                m16 = st_mag_raw_dword(rawj.w, m16);
                const uint32_t m = (m16 & 0xffffu) + (m16 >> 16);
#elif IQD_D4_MAGLUT == 3   // one quad-SAD per dword (iqd_mfma.h: st_mag_raw_dword_q), no table: AM -0.5 %, USB -1 %, FM +5 % (round 4)
                const uint32_t m16 = st_mag_raw_chunk(rawj, 0u);
                const uint32_t m = (m16 & 0xffffu) + (m16 >> 16);
#else
                const uint32_t m = st_mag_chunk(cur);
#endif
                macc += mcount && pos < mlimit && pos >= mfirst ? m : 0u;
                if ((j & 3) == 3) {   // blocks, segments and lead-ins are whole quads: the boundary test once per quad
                    minblk += 128;
                    if (minblk >= (int32_t)a.block_samples) {
                        if (macc) atomicAdd(&mag_row[mblk], macc);
                        macc = 0;
                        mblk++;
                        minblk -= (int32_t)a.block_samples;
                    }
                }
            }
#if IQD_D4_TIMING
            const long long tC = clock64();
            t_front_mag += tC - tB;
#endif
            // the buffer's next load only now, after the last use of its old contents: while those are live the new
            // load would get other registers and the loop would have to move it back - reading registers in flight
            raw[j] = load_piece(pos + 32 * D4_AHEAD);
            // rows 4g'+r: r even = I' rail, odd = Q' rail, output 2g' + (r >> 1) of the piece
            int y[4];
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = (lo[r] + (int)((uint32_t)hi[r] << 8)) >> 15;
            if (MODE == D4_FM) {   // |y| <= 141: the exact theta table (FmDemodulator.cc:476)
                asked_a[j & 1] = gload4_untracked(&da.fm_lut[(y[1] + FM_LUT_R) * FM_LUT_W + (y[0] + FM_LUT_R)]);
                asked_b[j & 1] = gload4_untracked(&da.fm_lut[(y[3] + FM_LUT_R) * FM_LUT_W + (y[2] + FM_LUT_R)]);
                if (!decltype(first)::value || j > 0) {  // the piece before: younger than its second angle are this piece's input load and two angles
                    gload_wait<3>(asked_a[(j & 1) ^ 1]);
                    gload_wait<3>(asked_b[(j & 1) ^ 1]);
                    fm_finish(pos - 32, asked_a[(j & 1) ^ 1], asked_b[(j & 1) ^ 1], (j + 3) & 3);
                }
            } else {
                // I' outputs 2g, 2g+1 | Q'
                hand_over(u32x2{pack_lo16((uint32_t)y[0], (uint32_t)y[2]), pack_lo16((uint32_t)y[1], (uint32_t)y[3])}, j & 3);
            }
            prev = cur;
#if IQD_D4_TIMING
            const long long tD = clock64();
            t_post_hand += tD - tC;
#endif
        };
#if IQD_D4_TIMING
        const long long t_round0 = clock64();
#endif
        int q_start = 0;
        if (PER_PIECE > 1) {                                    // (n_pieces is a multiple of 4 and at least 12)
#pragma unroll
            for (int j = 0; j < D4_AHEAD; j++) piece(std::true_type{}, 0, j);
            q_start = D4_AHEAD;
        }
        int q0 = q_start;
        for (; q0 + D4_AHEAD <= n_pieces; q0 += D4_AHEAD) {
#pragma unroll
            for (int j = 0; j < D4_AHEAD; j++) piece(std::false_type{}, q0, j);
        }
        // eight in flight and an odd number of quads: the last quad is the start of another trip (buffers 0 .. 3; the loads
        // younger than each of them are the same seven as in every trip)
        if (D4_AHEAD > 4 && q0 < n_pieces) {
#pragma unroll
            for (int j = 0; j < 4; j++) piece(std::false_type{}, q0, j);
        }
        // the loads asked for beyond the last piece (clamped re-reads) must have landed before their registers move on
#pragma unroll
        for (int j = 0; j < D4_AHEAD; j++) gload_wait<0>(raw[j]);
        if (MODE == D4_FM) {       // the last piece
            gload_wait<0>(asked_a[(D4_AHEAD - 1) & 1]);
            gload_wait<0>(asked_b[(D4_AHEAD - 1) & 1]);
            fm_finish(-da.halo + 32 * (n_pieces - 1), asked_a[(D4_AHEAD - 1) & 1], asked_b[(D4_AHEAD - 1) & 1], 3);
        }
        if (MAG && macc) atomicAdd(&mag_row[mblk], macc);
#if IQD_D4_TIMING
        if ((blockIdx.x & 63) == 7 && lane == 0)
            printf("wg %u P wave %d: round %lld cycles for %d pieces: wait_raw %lld front+mfma+mag %lld post+handover %lld (of which ring-space wait %lld)\n", blockIdx.x,
                   (int)(threadIdx.x >> 6), clock64() - t_round0, n_pieces, t_wait_raw, t_front_mag, t_post_hand, t_ring_wait);
#endif
}

template <int MODE, bool MAG, bool GATED>
__device__ __forceinline__ void d4_p_wave(const ChainLaunch &a, const D4Args &da, uint8_t *lds, uint32_t *sync, int pw, int lane)
{
    // a ring's four P waves are every third wave, not four in a row: the hardware issues oldest wave first, and with
    // rings of neighbouring waves ring 0 ran a third ahead of ring 2 (per-wave end times 115 / 137 / 155 us), which left
    // the last ring to finish on a nearly empty CU.  Now every ring has a wave of each age.
#if IQD_RINGS_IN_A_ROW
    const int ring = pw / ST_P_PER_RING, cg = pw % ST_P_PER_RING;
#else
    const int ring = pw % ST_RINGS, cg = pw / ST_RINGS;
#endif
    if (ring >= (int)da.rings) return;                         // (a workgroup of fewer rings, D4Args::rings: this wave's is not there)
#ifdef IQD_D4_PRIO_YOUNG   // measurement build: the youngest P wave of every ring (hardware waves 12-14, the fourth wave of SIMDs 0-2) at a raised priority
    if (pw >= 3 * (ST_P_PER_RING - 1)) __builtin_amdgcn_s_setprio(IQD_D4_PRIO_YOUNG);
#endif
    const uint32_t wg_segs = 64u * da.rings;
    const int g = lane >> 4, c = lane & 15;
    const uint32_t row = (uint32_t)(16 * cg + c);
    uint8_t *ring_base = lds + ring * (D4_SLOTS * D4_SLOT_BYTES);
    const uint32_t *full = sync + ring * D4_QUADS;
    const uint32_t *consumed = sync + ST_RINGS * D4_QUADS + ring;
    const uint32_t wr_off = MODE == D4_FM ? row * 16u + 4u * (uint32_t)g : d4_ring_off(row, (uint32_t)g);   // FM: one dword per lane
    uint32_t pg = 0;                                           // quads this ring has seen (all rounds)
    for (uint32_t round = 0; round < da.rounds; round++) {
        if ((round * chain_wgs(a) + chain_wg(a)) * wg_segs >= da.group_start[3]) break;   // nothing left for this workgroup
        const uint32_t sid0 = (round * chain_wgs(a) + chain_wg(a)) * wg_segs + ring * 64 + 16 * cg;
        const D4Seg sg = d4_segment(a, da, sid0 + (uint32_t)c);
#if IQD_D4_TRANSPOSE
        const D4Seg sgl = d4_segment(a, da, sid0 + (uint32_t)(lane >> 2));
#else
        const D4Seg &sgl = sg;
#endif
        const int rot = __builtin_amdgcn_readfirstlane(sg.rot);
        if (rot == 0) d4_p_round<MODE, MAG, 0, GATED>(a, da, sg, sgl, ring_base, full, consumed, sync, wr_off, g, lane, pg);
        else if (rot > 0) d4_p_round<MODE, MAG, 1, GATED>(a, da, sg, sgl, ring_base, full, consumed, sync, wr_off, g, lane, pg);
        else d4_p_round<MODE, MAG, -1, GATED>(a, da, sg, sgl, ring_base, full, consumed, sync, wr_off, g, lane, pg);
    }
}

__device__ __forceinline__ uint32_t pk_abs_i16(uint32_t v)
{
    uint32_t n, r;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(v));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(v), "v"(n));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- consumer waves ---------------------------------------------------------------------------------------
// one ring row: the 4 P lanes' payloads of this lane's segment, un-swizzled
__device__ __forceinline__ void d4_read_row(const uint8_t *slot, uint32_t row, u32x2 (&p)[4])
{
#pragma unroll
    for (int g = 0; g < 4; g++) p[g] = *(const u32x2 *)(slot + d4_ring_off(row, (uint32_t)g));
}

#if IQD_D4_WAITSTAT || IQD_D4_TIMING
__device__ __forceinline__ uint32_t *d4_stat_word(int i)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t d4_lds[];
    return (uint32_t *)(d4_lds + ST_RINGS * D4_SLOTS * D4_SLOT_BYTES) + i;
}
#endif
__device__ __forceinline__ void d4_wait_quad(const uint32_t *full, uint32_t pg)   // all four P waves have stored quad pg
{
    const uint32_t target = 4u * ((pg / D4_QUADS) + 1u);
#if IQD_D4_TIMING
    const long long tw0 = clock64();
#endif
    while ((int32_t)(lds_load_relaxed(&full[pg & (D4_QUADS - 1)]) - target) < 0) {
#if IQD_D4_WAITSTAT
        if ((threadIdx.x & 63) == 0) atomicAdd(d4_stat_word(D4_SYNC_WORDS - 1), 1u);
#endif
        __builtin_amdgcn_s_sleep(IQD_D4_SLEEP_C);
    }
#if IQD_D4_TIMING   // cycles this consumer wave waited for its P waves, per hardware wave 0..2 (= ring)
    if ((threadIdx.x & 63) == 0) atomicAdd(d4_stat_word(ST_RINGS * D4_QUADS + ST_RINGS + (int)(threadIdx.x >> 6)), (uint32_t)(clock64() - tw0));
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// piece V of quad pg: its slot; the consumer waits before a quad's first piece and hands the quad back after its last
template <int V>
__device__ __forceinline__ const uint8_t *d4_take_piece(const uint8_t *ring_base, const uint32_t *full, uint32_t pg)
{
    if (V == 0) d4_wait_quad(full, pg);
    return ring_base + ((pg & (D4_QUADS - 1)) * 4 + V) * D4_SLOT_BYTES;
}
template <int V>
__device__ __forceinline__ void d4_piece_taken(uint32_t *consumed, uint32_t &pg)
{
    if (V == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_signal(consumed);
        pg++;
    }
}

// the value the lane below holds (lane 0: its own) - v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ uint32_t d4_from_lane_below(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}

// a consumer lane's 8-byte store of four outputs (IQD_D4_WT_STORES: measurement build, write-through - 64 narrow write-throughs per
// instruction: USB 4096 x 2^16 +4 %, configs[4] +2.6 %, profiles/r6_wt_ab.txt; the closing launch's coalesced stores do go out that way)
#ifdef IQD_D4_WT_STORES
#define d4_store4(P, LO, HI) __hip_atomic_store((unsigned long long *)(P), (unsigned long long)(LO) | ((unsigned long long)(HI) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#elif defined(IQD_D4_NT_STORES)   // (non-temporal)
typedef uint32_t d4_v2u __attribute__((ext_vector_type(2)));
#define d4_store4(P, LO, HI) __builtin_nontemporal_store(d4_v2u{LO, HI}, (d4_v2u *)(P))
#else
#define d4_store4(P, LO, HI) (*(u32x2 *)(P) = u32x2{LO, HI})
#endif

// AM / SSB -----------------------------------------------------------------------------------------------
struct D4Rail {
    uint32_t y1h[4];      // the last 8 stage-1 outputs of the rail
    uint32_t y2[11];      // stage-2 outputs as pairs; variant V of a piece uses [V .. V+7], its new pair is [V+7]
};

// stage 3 alone (/2, 16 taps over the y2 pairs) with this piece's pair given: the boundary replay
template <int V>
__device__ __forceinline__ int d4_s3(const D4Args &da, D4Rail &r, uint32_t pair, int c14)
{
    r.y2[V + 7] = pair;
    int s3 = c14;
#pragma unroll
    for (int q = 0; q < 8; q++) s3 = dot2(r.y2[V + 7 - q], da.s3p[q], s3);
    return s3 >> 15;
}

// one piece of one rail: 8 stage-1 outputs in, 2 stage-2 outputs (a pair) kept, 1 stage-3 output returned
// (c14, c15: the rounding terms 1 << 14 and 1 << 15 in registers - as constants the compiler moves them into the
// accumulator in front of every chain; stage 2 runs with doubled taps, D4Args::s2p, and rounds with 1 << 15)
template <int V>
__device__ __forceinline__ int d4_am_rail(const D4Args &da, D4Rail &r, const uint32_t (&n1)[4], int c14, int c15)
{
    const uint32_t w[8] = {r.y1h[0], r.y1h[1], r.y1h[2], r.y1h[3], n1[0], n1[1], n1[2], n1[3]};
    int a0 = c15, a1 = c15;                            // /4, 12 taps: output k from y1[4k-8 .. 4k+3]
#pragma unroll
    for (int q = 0; q < 6; q++) {
        a0 = dot2(w[5 - q], da.s2p[q], a0);
        a1 = dot2(w[7 - q], da.s2p[q], a1);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) r.y1h[k] = n1[k];
    r.y2[V + 7] = pack_hi16((uint32_t)a0, (uint32_t)a1);
    int s3 = c14;                                      // /2, 16 taps: from y2[2i-14 .. 2i+1]
#if IQD_D4_SPLIT
    int s3b = 0;                                       // (two half-length chains: int32 sums wrap, so any order is exact)
#pragma unroll
    for (int q = 0; q < 4; q++) { s3 = dot2(r.y2[V + 7 - q], da.s3p[q], s3); s3b = dot2(r.y2[V + 3 - q], da.s3p[q + 4], s3b); }
    s3 += s3b;
#else
#pragma unroll
    for (int q = 0; q < 8; q++) s3 = dot2(r.y2[V + 7 - q], da.s3p[q], s3);
#endif
    return s3 >> 15;
}

struct D4Ssb {             // 8 kS/s histories of the phasing detector, split by sample parity (the Hilbert taps at odd
    uint32_t qe[9], qo[9]; // distances are zero): pairs of same-parity samples, older in the low half, newest pair last
    uint32_t ie[5], io[5];
};

// Hilbert transformer over one parity stream whose newest sample sits in the HIGH half of p[8] (full == true) or in
// the LOW half of p[8] (full == false: that pair's high half is not there yet).  taps t[j] = h[2j], j = 0..15.
template <bool FULL>
__device__ __forceinline__ int d4_hilbert(const D4Args &da, const uint32_t (&p)[9])
{
    int acc = 1 << 14;
    if (FULL) {
#pragma unroll
        for (int i = 0; i < 8; i++)   // pair p[8-i] = (S[m-2i-1], S[m-2i]): taps (t[2i+1], t[2i])
            acc = dot2(p[8 - i], (da.hilb[2 * i + 1] & 0xffffu) | (da.hilb[2 * i] << 16), acc);
    } else {
        acc = dot2(p[8], da.hilb[0] & 0xffffu, acc);   // S[m] alone in the low half
#pragma unroll
        for (int i = 1; i <= 8; i++)  // pair p[8-i] = (S[m-2i], S[m-2i+1]): taps (t[2i], t[2i-1])
            acc = dot2(p[8 - i], ((2 * i < 16 ? da.hilb[2 * i] : 0u) & 0xffffu) | (da.hilb[2 * i - 1] << 16), acc);
    }
    return acc >> 15;
}

// the detector stage of one piece from its 8 kS/s rail values (the pipeline's pieces and the boundary replay)
template <int MODE, int V>
__device__ __forceinline__ int d4_detect(const D4Args &da, D4Ssb &sb, int iv, int qv, int lsb)
{
    if (MODE == D4_AM) {   // AmDemodulator.cc:446-459: max(|i|,|q|) + min(|i|,|q|)/2 in int16 arithmetic
        const int im = (int)(int16_t)(iv < 0 ? -iv : iv), qm = (int)(int16_t)(qv < 0 ? -qv : qv);
        return (int)(int16_t)((im > qm) ? im + (qm >> 1) : qm + (im >> 1));
    }
    // SSB (SsbDemodulator.cc:574-588): delayed I - the 1.0 tap is -32768 in Q15, so (16384 - 32768 i[n-15]) >> 15 =
    // -i[n-15] - and the 31-tap Hilbert transformer on Q; sample n of this piece has parity V & 1, and within its
    // parity stream it is the low (V < 2) or the high (V >= 2) half of the newest pair.
    int idl, qh;
    if ((V & 1) == 0) {
        if (V < 2) { sb.qe[8] = (uint32_t)qv & 0xffffu; sb.ie[4] = (uint32_t)iv & 0xffffu; }
        else { sb.qe[8] = pack_lo16(sb.qe[8], (uint32_t)qv); sb.ie[4] = pack_lo16(sb.ie[4], (uint32_t)iv); }
        qh = V < 2 ? d4_hilbert<false>(da, sb.qe) : d4_hilbert<true>(da, sb.qe);
        // i[n-15] lies in the odd stream, 7 samples before its newest one (n-1): that one is the high half of io[3]
        // when V == 0 (the pair completed in the previous run of 4 pieces), the low half of io[4] when V == 2
        idl = V < 2 ? -(int)(int16_t)(sb.io[0] & 0xffffu) : -(int)(int16_t)(sb.io[0] >> 16);
    } else {
        if (V < 2) { sb.qo[8] = (uint32_t)qv & 0xffffu; sb.io[4] = (uint32_t)iv & 0xffffu; }
        else { sb.qo[8] = pack_lo16(sb.qo[8], (uint32_t)qv); sb.io[4] = pack_lo16(sb.io[4], (uint32_t)iv); }
        qh = V < 2 ? d4_hilbert<false>(da, sb.qo) : d4_hilbert<true>(da, sb.qo);
        // even stream, 7 before its newest sample (n-1): the low half of ie[4] when V == 1, the high half of ie[4] when V == 3
        idl = V < 2 ? -(int)(int16_t)(sb.ie[0] >> 16) : -(int)(int16_t)(sb.ie[1] & 0xffffu);
    }
    return lsb ? (int)(int16_t)idl - (int)(int16_t)qh : (int)(int16_t)idl + (int)(int16_t)qh;
}

template <int MODE, int V>
__device__ __forceinline__ int d4_am_piece(const D4Args &da, const uint8_t *ring_base, const uint32_t *full, uint32_t *consumed,
                                           uint32_t &pg, uint32_t row, int lane, D4Rail &ri, D4Rail &rq, D4Ssb &sb, int lsb, int c14, int c15,
                                           uint32_t &rails)   // (out: this piece's 8 kS/s rail values, i in the low half - SSB's boundary replay)
{
    u32x2 p[4];
    d4_read_row(d4_take_piece<V>(ring_base, full, pg), row, p);
    d4_piece_taken<V>(consumed, pg);
    const uint32_t ni[4] = {p[0].x, p[1].x, p[2].x, p[3].x}, nq[4] = {p[0].y, p[1].y, p[2].y, p[3].y};
#ifdef IQD_D4_BURN_CONSUMER   // measurement build: the consumer waves issue this many idle vector instructions per piece
    {
        float burn = 1.0f;
#pragma unroll
        for (int k = 0; k < IQD_D4_BURN_CONSUMER; k++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(burn));
    }
#endif
    const int iv = d4_am_rail<V>(da, ri, ni, c14, c15), qv = d4_am_rail<V>(da, rq, nq, c14, c15);
    if (MODE == D4_SSB) rails = pack_lo16((uint32_t)iv, (uint32_t)qv);
    return d4_detect<MODE, V>(da, sb, iv, qv, lsb);
}

// A quad of the boundary replay: the pieces' y2 pairs (first quad of the body: h != nullptr at q4 == 0) or their rails (SSB, later
// quads) come from the lane's head store; stage 3 / the detector run with the state handed in.
template <int MODE, int V>
__device__ __forceinline__ int d4_replay_piece(const D4Args &da, D4Rail &ri, D4Rail &rq, D4Ssb &sb, int lsb, int c14, bool from_y2, uint32_t wi, uint32_t wq)
{
    int iv, qv;
    if (from_y2) {   // wi, wq: this piece's y2 pairs of the two rails
        iv = d4_s3<V>(da, ri, wi, c14);
        qv = d4_s3<V>(da, rq, wq, c14);
    } else {         // wi: the rails, i in the low half
        iv = (int)(int16_t)(wi & 0xffffu);
        qv = (int)(int16_t)(wi >> 16);
    }
    return d4_detect<MODE, V>(da, sb, iv, qv, lsb);
}

template <int MODE>
__device__ __forceinline__ void d4_am_wave(const ChainLaunch &a, const D4Args &da, uint8_t *lds, uint32_t *sync, int ring, int lane)
{
    const uint8_t *ring_base = lds + ring * (D4_SLOTS * D4_SLOT_BYTES);
    const uint32_t *full = sync + ring * D4_QUADS;
    uint32_t *consumed = sync + ST_RINGS * D4_QUADS + ring;
    const int n_pieces = (da.halo + (int)a.tile_len) >> 5;
    if (ring >= (int)da.rings) return;
    const uint32_t wg_segs = 64u * da.rings;
    uint32_t pg = 0;
    for (uint32_t round = 0; round < da.rounds; round++) {
        if ((round * chain_wgs(a) + chain_wg(a)) * wg_segs >= da.group_start[3]) break;
        const uint32_t sid = (round * chain_wgs(a) + chain_wg(a)) * wg_segs + ring * 64 + lane;
        const D4Seg sg = d4_segment(a, da, sid);
        const int lsb = a.params[sg.ech].ssb_lsb;
        int32_t *base_row = a.base8k + (size_t)sg.ch * a.base_stride_ch;
        int16_t *det_row = a.pcm + (size_t)sg.ch * a.pcm_stride;
        D4Rail ri, rq;
        D4Ssb sb;
        int c14 = 1 << 14, c15 = 1 << 15;
        asm volatile("" : "+v"(c14), "+v"(c15));
#pragma unroll
        for (int k = 0; k < 4; k++) ri.y1h[k] = rq.y1h[k] = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) ri.y2[k] = rq.y2[k] = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) sb.qe[k] = sb.qo[k] = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) sb.ie[k] = sb.io[k] = 0;
        // Short lead-ins (iqd_stream.h, d4_geom): the inputs of this segment's first outputs go into the lane's head store as they pass -
        // the y2 pairs of pieces 4..7, SSB's rails of pieces 8..39 - and those outputs are replayed behind the loop with the
        // predecessor's end state, which is the lane below's.
        const bool lf = da.lead_shift != 0;
        uint32_t *head = (uint32_t *)(lds + D4_HEAD_OFF) + ring * 64 + lane;       // [word][ST_SEGS]
        for (int pq = 0; pq < n_pieces; pq += 4) {
            const int pos = -da.halo + 32 * pq;
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            const int x0 = d4_am_piece<MODE, 0>(da, ring_base, full, consumed, pg, (uint32_t)lane, lane, ri, rq, sb, lsb, c14, c15, r0);
            const int x1 = d4_am_piece<MODE, 1>(da, ring_base, full, consumed, pg, (uint32_t)lane, lane, ri, rq, sb, lsb, c14, c15, r1);
            const int x2 = d4_am_piece<MODE, 2>(da, ring_base, full, consumed, pg, (uint32_t)lane, lane, ri, rq, sb, lsb, c14, c15, r2);
            const int x3 = d4_am_piece<MODE, 3>(da, ring_base, full, consumed, pg, (uint32_t)lane, lane, ri, rq, sb, lsb, c14, c15, r3);
            // 128 samples = 4 detector inputs = one store (segments start and end on multiples of 128): int16 into the PCM row, where
            // the DC pass runs in place (ChainLaunch::det16; |x| <= 546), or int32 into the detector stream's own buffer
            if (sg.valid && pos >= sg.skip && pos < sg.tlen) {
                if (a.det16) d4_store4(det_row + ((sg.v0 + pos) >> 5), pack_lo16((uint32_t)x0, (uint32_t)x1), pack_lo16((uint32_t)x2, (uint32_t)x3));
                else *(u32x4 *)(base_row + ((sg.v0 + pos) >> 5)) = u32x4{(uint32_t)x0, (uint32_t)x1, (uint32_t)x2, (uint32_t)x3};
            }
            if (lf && pos == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) { head[k * ST_SEGS] = ri.y2[7 + k]; head[(4 + k) * ST_SEGS] = rq.y2[7 + k]; }
            }
            if (lf && MODE == D4_SSB && pos >= 128 && pos < 32 * D4_REPLAY_SSB) {     // pieces 8..39
                uint32_t *h = head + (8 + ((pos - 128) >> 5)) * ST_SEGS;
                h[0] = r0; h[ST_SEGS] = r1; h[2 * ST_SEGS] = r2; h[3 * ST_SEGS] = r3;
            }
#pragma unroll
            for (int k = 0; k < 7; k++) { ri.y2[k] = ri.y2[k + 4]; rq.y2[k] = rq.y2[k + 4]; }
            if (MODE == D4_SSB) {   // each parity stream gained one pair
#pragma unroll
                for (int k = 0; k < 8; k++) { sb.qe[k] = sb.qe[k + 1]; sb.qo[k] = sb.qo[k + 1]; }
#pragma unroll
                for (int k = 0; k < 4; k++) { sb.ie[k] = sb.ie[k + 1]; sb.io[k] = sb.io[k + 1]; }
            }
        }
        if (!lf) continue;
        // ---- the boundary: this lane's state now is its segment's END state = what its successor's first outputs reach back
        // for.  The successor is the next segment id = the lane above (a channel's segments have consecutive ids), so every lane
        // takes the state of the lane below and replays its own first outputs from its head store: 4 (AM), 36 (SSB: 34 needed).
        // (Lane 0 and a channel's first segment have no such neighbour: they are the cold segments, which ran their full lead-in.)
        // the lane below's end state
#pragma unroll
        for (int k = 0; k < 7; k++) { ri.y2[k] = d4_from_lane_below(ri.y2[k]); rq.y2[k] = d4_from_lane_below(rq.y2[k]); }
        if (MODE == D4_SSB) {
#pragma unroll
            for (int k = 0; k < 8; k++) { sb.qe[k] = d4_from_lane_below(sb.qe[k]); sb.qo[k] = d4_from_lane_below(sb.qo[k]); }
#pragma unroll
            for (int k = 0; k < 4; k++) { sb.ie[k] = d4_from_lane_below(sb.ie[k]); sb.io[k] = d4_from_lane_below(sb.io[k]); }
        }
        const bool mine = sg.valid && !sg.cold;
        constexpr int REPLAY_QUADS = (MODE == D4_SSB ? D4_REPLAY_SSB : D4_REPLAY_AM) / 4;
        for (int q4 = 0; q4 < REPLAY_QUADS; q4++) {
            uint32_t wi[4], wq[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wi[k] = q4 == 0 ? head[k * ST_SEGS] : head[(8 + 4 * (q4 - 1) + k) * ST_SEGS];
                wq[k] = q4 == 0 ? head[(4 + k) * ST_SEGS] : 0u;
            }
            const int x0 = d4_replay_piece<MODE, 0>(da, ri, rq, sb, lsb, c14, q4 == 0, wi[0], wq[0]);
            const int x1 = d4_replay_piece<MODE, 1>(da, ri, rq, sb, lsb, c14, q4 == 0, wi[1], wq[1]);
            const int x2 = d4_replay_piece<MODE, 2>(da, ri, rq, sb, lsb, c14, q4 == 0, wi[2], wq[2]);
            const int x3 = d4_replay_piece<MODE, 3>(da, ri, rq, sb, lsb, c14, q4 == 0, wi[3], wq[3]);
            if (mine && 128 * q4 < sg.tlen) {
                if (a.det16) d4_store4(det_row + ((sg.v0 + 128 * q4) >> 5), pack_lo16((uint32_t)x0, (uint32_t)x1), pack_lo16((uint32_t)x2, (uint32_t)x3));
                else *(u32x4 *)(base_row + ((sg.v0 + 128 * q4) >> 5)) = u32x4{(uint32_t)x0, (uint32_t)x1, (uint32_t)x2, (uint32_t)x3};
            }
#pragma unroll
            for (int k = 0; k < 7; k++) { ri.y2[k] = ri.y2[k + 4]; rq.y2[k] = rq.y2[k + 4]; }
            if (MODE == D4_SSB) {
#pragma unroll
                for (int k = 0; k < 8; k++) { sb.qe[k] = sb.qe[k + 1]; sb.qo[k] = sb.qo[k + 1]; }
#pragma unroll
                for (int k = 0; k < 4; k++) { sb.ie[k] = sb.ie[k + 1]; sb.io[k] = sb.io[k + 1]; }
            }
        }
    }
}

// FM -----------------------------------------------------------------------------------------------------
struct D4Fm {
    uint32_t eh[4];        // the last 8 (int16)(K dtheta) of the previous piece
    uint32_t early;        // the pair the P waves delivered a piece ahead (their stream runs two samples early)
    uint32_t y2p[24];      // /4 outputs as pairs; variant V of a piece uses [V+1 .. V+20]
    int loud_e, loud_y2;   // pieces for which a value above the clamp-free bound stays in a window's reach
};

// the last stage of one piece (/2, 40 taps -> PCM, FmDemodulator.cc:548-556) from the piece's pair of /4 outputs: the pipeline's
// pieces and the boundary replay
template <int V>
__device__ __forceinline__ int d4_fm_audio(D4Fm &s, uint32_t pair)
{
    const int y0 = (int)(int16_t)(pair & 0xffffu), y1 = (int)(int16_t)(pair >> 16);
    const uint32_t m0 = (uint32_t)(y0 < 0 ? -y0 : y0), m1 = (uint32_t)(y1 < 0 ? -y1 : y1);
    if ((m0 > m1 ? m0 : m1) > (uint32_t)AUDIO40_SAFE) s.loud_y2 = 21;
    s.y2p[V + 20] = pair;
    int acc = 1 << 14;                                 // /2, 40 taps -> PCM
    if (!__any(s.loud_y2 > 0)) {
#if IQD_D4_SPLIT
        int accb = 0, accc = 0, accd = 0;              // (four chains of 5: without the clamps int32 sums wrap, any order is exact)
#pragma unroll
        for (int q = 0; q < 5; q++) {
            acc = dot2(s.y2p[V + 20 - q], D4_TAP(a40p, q), acc);
            accb = dot2(s.y2p[V + 15 - q], D4_TAP(a40p, q + 5), accb);
            accc = dot2(s.y2p[V + 10 - q], D4_TAP(a40p, q + 10), accc);
            accd = dot2(s.y2p[V + 5 - q], D4_TAP(a40p, q + 15), accd);
        }
        acc = (acc + accb) + (accc + accd);
#else
#pragma unroll
        for (int q = 0; q < 20; q++) acc = dot2(s.y2p[V + 20 - q], D4_TAP(a40p, q), acc);
#endif
    } else {
#pragma unroll
        for (int q = 0; q < 20; q++) {
            acc = clamp_q30(dot2(s.y2p[V + 20 - q], D4_TAP(a40p, q) & 0xffff0000u, acc));
            acc = clamp_q30(dot2(s.y2p[V + 20 - q], D4_TAP(a40p, q) & 0x0000ffffu, acc));
        }
    }
    if (s.loud_y2 > 0) s.loud_y2--;
    return acc >> 15;
}

template <int V>
__device__ __forceinline__ int d4_fm_piece(const D4Args &da, const uint8_t *ring_base, const uint32_t *full, uint32_t *consumed,
                                           uint32_t &pg, uint32_t row, D4Fm &s)
{
    const u32x4 n = *(const u32x4 *)(d4_take_piece<V>(ring_base, full, pg) + row * 16u);
    d4_piece_taken<V>(consumed, pg);
    // this piece's 8 discriminator outputs: the pair that came early with the previous piece, then three of the four new
    const uint32_t w[8] = {s.eh[0], s.eh[1], s.eh[2], s.eh[3], s.early, n.x, n.y, n.z};
    s.early = n.w;
    uint32_t peak = 0;     // max |e| of the piece, on the pairs (|-32768| stays 0x8000 = 32768 unsigned)
#pragma unroll
    for (int j = 4; j < 8; j++) peak = pk_max_u16(peak, pk_abs_i16(w[j]));
    peak = (peak & 0xffffu) > (peak >> 16) ? (peak & 0xffffu) : (peak >> 16);
    if (peak > (uint32_t)POST12_SAFE) s.loud_e = 4;   // stays inside a 12-sample window for this piece and the next
    int a0 = 1 << 14, a1 = 1 << 14;                    // /4, 12 taps (FmDemodulator.cc:545): output j from e[4j-8 .. 4j+3]
    if (!__any(s.loud_e > 0)) {
#pragma unroll
        for (int q = 0; q < 6; q++) {
            a0 = dot2(w[5 - q], D4_TAP(p12p, q), a0);
            a1 = dot2(w[7 - q], D4_TAP(p12p, q), a1);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 6; q++) {
            a0 = clamp_q30(dot2(w[5 - q], D4_TAP(p12p, q) & 0xffff0000u, a0));
            a0 = clamp_q30(dot2(w[5 - q], D4_TAP(p12p, q) & 0x0000ffffu, a0));
            a1 = clamp_q30(dot2(w[7 - q], D4_TAP(p12p, q) & 0xffff0000u, a1));
            a1 = clamp_q30(dot2(w[7 - q], D4_TAP(p12p, q) & 0x0000ffffu, a1));
        }
    }
    if (s.loud_e > 0) s.loud_e--;
#pragma unroll
    for (int j = 0; j < 4; j++) s.eh[j] = w[4 + j];
    return d4_fm_audio<V>(s, pack_lo16((uint32_t)(a0 >> 15), (uint32_t)(a1 >> 15)));
}

__device__ __forceinline__ void d4_fm_wave(const ChainLaunch &a, const D4Args &da, uint8_t *lds, uint32_t *sync, int ring, int lane)
{
    const uint8_t *ring_base = lds + ring * (D4_SLOTS * D4_SLOT_BYTES);
    const uint32_t *full = sync + ring * D4_QUADS;
    uint32_t *consumed = sync + ST_RINGS * D4_QUADS + ring;
    const int n_pieces = (da.halo + (int)a.tile_len) >> 5;
    if (ring >= (int)da.rings) return;
    const uint32_t wg_segs = 64u * da.rings;
    uint32_t pg = 0;
    for (uint32_t round = 0; round < da.rounds; round++) {
        if ((round * chain_wgs(a) + chain_wg(a)) * wg_segs >= da.group_start[3]) break;
        const uint32_t sid = (round * chain_wgs(a) + chain_wg(a)) * wg_segs + ring * 64 + lane;
        const D4Seg sg = d4_segment(a, da, sid);
        int16_t *pcm_row = a.pcm + (size_t)sg.ch * a.pcm_stride;
        D4Fm s;
#pragma unroll
        for (int j = 0; j < 4; j++) s.eh[j] = 0;
        s.early = 0;
#pragma unroll
        for (int j = 0; j < 24; j++) s.y2p[j] = 0;
        s.loud_e = s.loud_y2 = 0;
        // Short lead-ins (iqd_stream.h, d4_geom): the y2 pairs of pieces 4..23 go into the lane's head store as they pass; the first 20 PCM
        // samples are replayed behind the loop with the predecessor's end state - the lane below's (see d4_am_wave).
        const bool lf = da.lead_shift != 0;
        uint32_t *head = (uint32_t *)(lds + D4_HEAD_OFF) + ring * 64 + lane;       // [word][ST_SEGS]
        for (int pq = 0; pq < n_pieces; pq += 4) {
            const int pos = -da.halo + 32 * pq;
            int pcm[4];
            pcm[0] = d4_fm_piece<0>(da, ring_base, full, consumed, pg, (uint32_t)lane, s);
            pcm[1] = d4_fm_piece<1>(da, ring_base, full, consumed, pg, (uint32_t)lane, s);
            pcm[2] = d4_fm_piece<2>(da, ring_base, full, consumed, pg, (uint32_t)lane, s);
            pcm[3] = d4_fm_piece<3>(da, ring_base, full, consumed, pg, (uint32_t)lane, s);
            if (sg.valid && pos >= sg.skip && pos < sg.tlen)
                d4_store4(pcm_row + ((sg.v0 + pos) >> 5), pack_lo16((uint32_t)pcm[0], (uint32_t)pcm[1]), pack_lo16((uint32_t)pcm[2], (uint32_t)pcm[3]));
            if (lf && pos >= 0 && pos < 32 * D4_REPLAY_FM) {
                uint32_t *h = head + (pos >> 5) * ST_SEGS;
                h[0] = s.y2p[20]; h[ST_SEGS] = s.y2p[21]; h[2 * ST_SEGS] = s.y2p[22]; h[3 * ST_SEGS] = s.y2p[23];
            }
#pragma unroll
            for (int j = 0; j < 20; j++) s.y2p[j] = s.y2p[j + 4];
        }
        if (!lf) continue;
        // the lane below's end state: its last 20 pairs and how long a loud value stays in reach
#pragma unroll
        for (int j = 0; j < 20; j++) s.y2p[j] = d4_from_lane_below(s.y2p[j]);
        s.loud_y2 = (int)d4_from_lane_below((uint32_t)s.loud_y2);
        const bool mine = sg.valid && !sg.cold;
        for (int q4 = 0; q4 < D4_REPLAY_FM / 4; q4++) {
            const uint32_t *h = head + 4 * q4 * ST_SEGS;
            int pcm[4];
            pcm[0] = d4_fm_audio<0>(s, h[0]);
            pcm[1] = d4_fm_audio<1>(s, h[ST_SEGS]);
            pcm[2] = d4_fm_audio<2>(s, h[2 * ST_SEGS]);
            pcm[3] = d4_fm_audio<3>(s, h[3 * ST_SEGS]);
            if (mine && 128 * q4 < sg.tlen)
                d4_store4(pcm_row + ((sg.v0 + 128 * q4) >> 5), pack_lo16((uint32_t)pcm[0], (uint32_t)pcm[1]), pack_lo16((uint32_t)pcm[2], (uint32_t)pcm[3]));
#pragma unroll
            for (int j = 0; j < 20; j++) s.y2p[j] = s.y2p[j + 4];
        }
    }
}

#ifndef IQD_D4_AM_TWO_WGS
#define IQD_D4_AM_TWO_WGS 0
#endif
#define D4_WAVES_PER_SIMD(MODE) ((MODE) == D4_AM && IQD_D4_AM_TWO_WGS ? 8 : 4)
// (a function of its own: the kernel below calls it, and so does the launch that runs several families side by side,
// iqd_stream_mixed.hip)
template <int MODE, bool MAG, bool GATED>
__device__ __forceinline__ void d4_stream_body(const ChainLaunch &a, const D4Args &da, uint8_t *d4_lds)
{
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)d4_lds != 0u) __builtin_trap();   // st_maglut_chunk_at()
    uint32_t *sync = (uint32_t *)(d4_lds + ST_RINGS * D4_SLOTS * D4_SLOT_BYTES);
    const int tid = (int)threadIdx.x;
    if (tid < D4_SYNC_WORDS) sync[tid] = 0;
    if (MAG && IQD_D4_MAGLUT == 1) st_maglut_build(d4_lds + D4_MAGLUT_OFF, tid, ST_THREADS);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#if IQD_D4_TIMING == 2
    const long long t_wave0 = clock64();
    const long long t_real0 = wall_clock64();
#endif
#ifdef IQD_D4_CONSUMERS_ON_SIMD3   // measurement build: the consumer waves are hardware waves 3, 7, 11 (all on the SIMD that holds three waves)
    const bool is_consumer = (wave & 3) == 3;
    const int c_ring = wave >> 2, p_index = wave - (wave > 3) - (wave > 7) - (wave > 11);
#else
    const bool is_consumer = wave < ST_RINGS;
    const int c_ring = wave, p_index = wave - ST_RINGS;
#endif
    if (is_consumer) {
        if (IQD_D4_PRIO) __builtin_amdgcn_s_setprio(IQD_D4_PRIO);
#if IQD_D4_TIMING == 1
        const long long t_c0 = clock64();
#endif
        if (MODE == D4_FM) d4_fm_wave(a, da, d4_lds, sync, c_ring, lane);
        else d4_am_wave<MODE>(a, da, d4_lds, sync, c_ring, lane);
#if IQD_D4_TIMING == 1
        if ((blockIdx.x & 63) == 7 && lane == 0)
            printf("wg %u consumer wave %d: %lld cycles for %d pieces, of which waiting for a full quad %u\n", blockIdx.x, wave, clock64() - t_c0,
                   (da.halo + (int)a.tile_len) >> 5, sync[ST_RINGS * D4_QUADS + ST_RINGS + wave]);
#endif
    } else {
        d4_p_wave<MODE, MAG, GATED>(a, da, d4_lds, sync, p_index, lane);
    }
#if IQD_D4_TIMING == 2
    {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const long long t_end = wall_clock64();
        __shared__ long long t_rec[32];            // per wave: run time in 10 ns ticks, HW_ID (which SIMD it sat on)
        if (lane == 0) {
            t_rec[wave * 2] = t_end - t_real0;
            t_rec[wave * 2 + 1] = hwid;
        }
        __syncthreads();
        if (tid == 0) {
            const long long *r = t_rec;
            unsigned long long simds = 0;
            for (int w = 0; w < 15; w++) simds |= (unsigned long long)((r[2 * w + 1] >> 4) & 3) << (4 * w);
            printf("T wg %u simds %llx cu %lld se %lld t %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", blockIdx.x, simds,
                   (r[1] >> 8) & 15, (r[1] >> 13) & 7, r[0], r[2], r[4], r[6], r[8], r[10], r[12], r[14], r[16], r[18], r[20], r[22], r[24], r[26], r[28]);
        }
    }
#endif
#if IQD_D4_WAITSTAT
    __syncthreads();
    if (tid == 0 && (blockIdx.x & 63) == 5)
        printf("wg %u: P sleeps %u (12 waves), consumer sleeps %u (3 waves), pieces %d\n", blockIdx.x, sync[D4_SYNC_WORDS - 2], sync[D4_SYNC_WORDS - 1], (da.halo + (int)a.tile_len) >> 5);
#endif
}

#ifndef IQD_STREAM_BODIES_ONLY
template <int MODE, bool MAG, bool GATED>
__global__ __launch_bounds__(ST_THREADS, D4_WAVES_PER_SIMD(MODE)) void d4_stream_kernel(const ChainLaunch a, const D4Args da)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t d4_lds[];
    d4_stream_body<MODE, MAG, GATED>(a, da, d4_lds);
}

typedef void (*D4Kernel)(const ChainLaunch, const D4Args);
// [chain][0: no magnitudes, 1: squelch magnitudes in the kernel, 2: squelch-gated launch (magnitudes taken by the pre-pass)]
static const D4Kernel d4_kernels[3][3] = {{d4_stream_kernel<D4_AM, false, false>, d4_stream_kernel<D4_AM, true, false>, d4_stream_kernel<D4_AM, false, true>},
                                          {d4_stream_kernel<D4_SSB, false, false>, d4_stream_kernel<D4_SSB, true, false>, d4_stream_kernel<D4_SSB, false, true>},
                                          {d4_stream_kernel<D4_FM, false, false>, d4_stream_kernel<D4_FM, true, false>, d4_stream_kernel<D4_FM, false, true>}};

// The streaming kernels want more LDS than the 64 KiB a kernel gets without asking.  The attribute belongs to the
// current device's code object: iqd_create calls this once per engine, serialised (no unsynchronised statics
// on the launch path).
hipError_t init_d4_stream_kernels()
{
    for (int m = 0; m < 3; m++)
        for (int g = 0; g < 3; g++) {
            const hipError_t e = hipFuncSetAttribute((const void *)d4_kernels[m][g], hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

hipError_t launch_d4_stream(const ChainLaunch &a, const D4Args &da, int mode, bool mag, uint32_t grid, hipStream_t s)
{
    const D4Kernel (&ks)[3][3] = d4_kernels;
    const bool gated = a.vlen_gated != nullptr;
    const int m = mode == D4_AM ? 0 : (mode == D4_SSB ? 1 : 2);
    hipLaunchKernelGGL(ks[m][gated ? 2 : (mag ? 1 : 0)], dim3(grid), dim3(ST_THREADS), D4_LDS_BYTES, s, a, da);
    return hipGetLastError();
}

#endif   // IQD_STREAM_BODIES_ONLY

}  // namespace iqd
