// WBFM chain as a streaming pipeline: one persistent 15-wave workgroup per CU (see iqd_stream.h).
// hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "iqd_kernels.h"
#include "iqd_stream.h"
#include "iqd_wbfm.h"
#include "iqd_mfma.h"
#include "iqd_stream_fix.h"
#include "iqd_taps.h"

namespace iqd {

// IQD_ST_WT_STORES (measurement build, tools/variant.sh): the IIR lanes' 16-byte stores - a lane's PCM leaves as whole 32-byte sectors,
// its boundary record in 16-byte parts - as write-through stores, so that the launch leaves nothing dirty in the L2s for the
// kernel boundary to write back (28.6 MB per 2^28-sample launch: PCM + records)
#ifdef IQD_ST_WT_STORES
__device__ __forceinline__ void st_store16(void *p, v4u v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
#define ST_STORE16(P, A, B, C, D) st_store16((void *)(P), v4u{A, B, C, D})
#elif defined(IQD_ST_NT_STORES)   // the same as non-temporal stores (the compiler's own: no register tuple forced on the lanes)
#define ST_STORE16(P, A, B, C, D) __builtin_nontemporal_store(v4u{A, B, C, D}, (v4u *)(P))
#else
#define ST_STORE16(P, A, B, C, D) (*(u32x4 *)(P) = u32x4{A, B, C, D})
#endif


// The angle table starts at LDS address 0 (the kernel has no static LDS; checked when the kernel starts), so a table
// cell's byte offset IS its LDS address: through the generic `lds + offset` form the compiler adds the base - a
// relocated zero - to every one of the eight addresses of a piece.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t st_table_read(uint32_t byte_offset) { return *(lds_cu32 *)(uintptr_t)byte_offset; }

#ifndef IQD_ST_WAITSTAT   // diagnostic build: how often each side of a ring sleeps on the other (per-wave counts, added once
#define IQD_ST_WAITSTAT 0 // per wave to ChainLaunch::stamps[0..3] = P sleeps, IIR sleeps, P pieces, IIR pieces; iqd_debug_stamps)
#endif

// How long a wave sleeps between two looks at its ring counter (units of 64 cycles).  A waiting wave's polls are
// instructions like any others, on a SIMD whose other waves are not waiting.
// s_sleep argument (x 64 cycles) between two looks at a ring counter.  A waiting wave's poll loop takes issue slots from the
// waves it waits for: with the counters read by ds_read_b32 (round 3: a flat_load before) one round of the loop is short, the
// waiters spun twice as often, and 8 % of the kernel's instructions were polls.  0.343 -> 0.332 ms from 1 / 1 to 6 / 10
// (3 / 3: 0.335, 10 / 10: 0.336, 16 / 16: 0.347, 24 / 24: 0.370).
#ifndef IQD_ST_SLEEP_P
#define IQD_ST_SLEEP_P 6
#endif
#ifndef IQD_ST_SLEEP_I
#define IQD_ST_SLEEP_I 10
#endif
#ifndef IQD_ST_RUNPTR     // 1: the P waves' input addresses as a running pointer (0: piece_address() per piece, A/B builds)
#define IQD_ST_RUNPTR 1
#endif
// The IIR lanes' decimator taps: literals of the v_dot2c instructions (the designs are fixed, iqd_taps.h: STREAM_TAPS) instead of
// 30 scalar registers that the wave could not keep (85 of its scalar registers spilled to vector lanes, 14 v_readlane per piece).
// IQD_ST_TAPS_IN_SGPRS: from the kernel arguments as before (the A/B).
#ifdef IQD_ST_TAPS_IN_SGPRS
#define ST_TAP(WHICH, Q) sa.WHICH[Q]
#else
#define ST_TAP(WHICH, Q) (uint32_t)taps::STREAM_TAPS.WHICH[Q]
#endif
#ifndef IQD_ST_DOT2_FROM   // a chain's first product in the three-address form (iqd_prims.h: dot2_from); 0: the A/B
#define IQD_ST_DOT2_FROM 1
#endif
#ifndef IQD_ST_YOUNG_SHIFT
#define IQD_ST_YOUNG_SHIFT 0
#endif
#ifndef IQD_ST_LEVEL_PROBE
#define IQD_ST_LEVEL_PROBE 0
#endif
// Lead-in of a ring whose 64 segments are all cold and of full length (the IIR wave's fast path): a cold segment's lead-in only
// has to make its de-emphasis state exact (its decimators' histories are replaced by the boundary fix-up), which takes 319 steps
// on average, p99.9 446, maximum 554 over 10^6 starts (tools/deemph_convergence.py); a state that has not converged is caught by
// the hand-off verification and repaired by the tile kernel.  768 = every segment runs the full ST_HALO (rounds 2-4).
#ifndef IQD_ST_COLD_HALO
#define IQD_ST_COLD_HALO 768
#endif
static_assert(IQD_ST_COLD_HALO % 128 == 0 && IQD_ST_COLD_HALO <= 768 && IQD_ST_COLD_HALO >= 256, "cold lead-in: whole quads of pieces within ST_HALO");
#ifndef IQD_ST_TRACE      // diagnostic build: workgroup 5 writes clock64() of (hardware wave, piece, event k) to stamps[64 + ((wave * 256 + piece) * 4 + k)]
#define IQD_ST_TRACE 0
#endif
#if IQD_ST_TRACE
#define ST_TRACE(st, wave, piece, k) do { if (blockIdx.x == 5 && lane == 0 && (piece) < 256) (st)[64 + (((wave) * 256 + (int)(piece)) * 4 + (k))] = (unsigned long long)clock64(); } while (0)
#else
#define ST_TRACE(st, wave, piece, k) do { } while (0)
#endif
#ifndef IQD_ST_TIMING     // diagnostic build: where a P wave's and an IIR wave's cycles go, per SIMD (stamps[16 + 8 simd + k])
#define IQD_ST_TIMING 0
#endif
#if IQD_ST_TIMING
#define ST_T(var) const long long var = clock64()
__device__ __forceinline__ int st_simd_id() { return (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11)) ) & 3; }   // HW_ID[5:4]
#else
#define ST_T(var) do { } while (0)
#endif

// ---- P wave: 16 segments, raw bytes -> u[n] -------------------------------------------------------
// EPOCHS: some channel of the launch has a gain change whose samples a lead-in can still reach (GainEpochList; the host
// knows: iqd_set_gain).  The piecewise-gain lookup lives only in that instantiation - in the common one it would sit
// in the piece loop as a 16-deep ladder twice over, costing registers (it spilled) and instruction cache for nothing.
// GATED: a squelch-gated launch (some channel lost blocks): virtual sample v of a channel lives in its open block number
// v / block_samples (ChainLaunch::blk_lists); a lane keeps the block it is in and looks the next one up when it leaves it.
// BYGROUP: the launch holds channels of several rotation selectors (StreamArgs::grouped); this call takes the rounds in
// which the wave's 16 segments belong to selector ROT and leaves the others to the calls for the other two selectors
// (st_p_wave_any: +1, 0, -1 in turn - the order of the groups, so a wave still meets its rounds in ascending order).
// pc: pieces the wave's ring has seen so far, carried through those calls.
template <int ROT, bool MAG, bool EPOCHS, bool GATED, bool BYGROUP = false>
__device__ __forceinline__ void st_p_wave(const ChainLaunch &a, const StreamArgs &sa, uint8_t *lds, uint32_t *sync,
                                          int pw, int lane, uint32_t &pc)
{
    // a ring's four P waves are every third wave, not four in a row: the hardware issues oldest wave first, and with
    // rings of neighbouring waves ring 0 ran a third ahead of ring 2 (per-wave end times 115 / 137 / 155 us), which left
    // the last ring to finish on a nearly empty CU.  Now every ring has a wave of each age.
#if IQD_RINGS_IN_A_ROW
    const int ring = pw / ST_P_PER_RING, cg = pw % ST_P_PER_RING;
#else
    // (IQD_ST_YOUNG_SHIFT: the three youngest P waves - hardware waves 12-14, each the last-served wave of a SIMD that also
    // carries an IIR wave - feed the ring of ANOTHER SIMD's IIR wave: a ring's IIR wave starts its burst when the ring's last
    // writer has signalled, i.e. exactly when that writer begins its next piece)
    const int cg = pw / ST_RINGS, ring = (pw + (cg == ST_P_PER_RING - 1 ? IQD_ST_YOUNG_SHIFT : 0)) % ST_RINGS;
#endif
    if (ring >= (int)sa.rings) return;                           // (a workgroup of fewer rings: this wave's is not there)
    const uint32_t wg_segs = 64u * sa.rings;                     // segments per workgroup and round
    const int g = lane >> 4, c = lane & 15;
    const uint32_t row = (uint32_t)(16 * cg + c);                // ring row of this lane's segment
    uint8_t *ring_base = lds + ST_TABLE_BYTES + ring * (ST_RING_SLOTS * ST_SLOT_BYTES);
    const uint32_t *full = sync + ring * 8;        // [slot of the ring]: pieces written into it, x 4 P waves
    asm volatile("" : "+v"(full));                 // (the LDS address stays in a register: else a move per use)
    const uint32_t *consumed = full + 4;           // pieces the IIR wave has read (one register for both: an offset in the instruction)
    const uint32_t wr_off = st_slot_off(row, (uint32_t)g);
    const int src_lane4 = ((lane - 16) & 63) << 2;               // whose theta[3] precedes this lane's theta[0]

    v4i A[8];
#pragma unroll
    for (int m = 0; m < 8; m++) A[m] = ((const v4i *)(BYGROUP ? sa.amat3[1 - ROT] : sa.amat))[m * 64 + lane];
    v4i cbias = {WB_BIAS, WB_BIAS, WB_BIAS, WB_BIAS};
    const v4i czero = {0, 0, 0, 0};
    asm volatile("" : "+v"(cbias));  // four VGPRs for the whole kernel: else the compiler rebuilds the quad from scalars for every piece
    uint32_t zero = 0;
    asm volatile("" : "+v"(zero));   // a VGPR holding 0 for the SDWA negations

    const int n_pieces = (ST_HALO + (int)a.tile_len) >> 5;
    uint64_t inv_2pi = 0x3e22f9843e22f984ull;                    // (float)(1 / (2 pi)) twice: the scalar operand of v_pk_mul_f32
    asm volatile("" : "+s"(inv_2pi));
    for (uint32_t round = 0; round < sa.rounds; round++) {
        if ((round * chain_wgs(a) + chain_wg(a)) * wg_segs >= st_id_count(sa)) break;   // nothing left for this workgroup
        const uint32_t sid = (round * chain_wgs(a) + chain_wg(a)) * wg_segs + ring * 64 + row;
        int rot_of_id = ROT;
        const StSeg sg = BYGROUP ? st_segment_of(a, sa, sid, rot_of_id) : st_segment(a, sid, sa.n_segments);
        if (BYGROUP && __builtin_amdgcn_readfirstlane(rot_of_id) != ROT) continue;   // (uniform: groups are padded to 16 ids)
        // pieces of the lead-in this ring skips (IQD_ST_COLD_HALO): the IIR wave's `fast` predicate, evaluated here over the ring's
        // 64 ids, one per lane - the ring's five waves must agree on it
        int q_first = 0;
        if (IQD_ST_COLD_HALO < ST_HALO && !BYGROUP && !GATED && !EPOCHS) {
            const StSeg ps = st_segment(a, sid - row + (uint32_t)lane, sa.n_segments);
            q_first = __all(st_cold_and_full(ps, a.tile_len)) ? (ST_HALO - IQD_ST_COLD_HALO) / 32 : 0;
        }
        const int lead = ST_HALO - 32 * q_first;                 // this ring's lead-in, samples
        const ChanParams &p = a.params[sg.ech];
        const uint8_t *iq_ch = a.iq + (size_t)sg.ch * a.ch_stride_bytes;
        const uint8_t *tail = a.tails + ((size_t)sg.ech * FAM_COUNT + FAM_WBFM) * TAIL_BYTES + TAIL_BYTES;
        const int32_t vmax = sg.vlen - 8;
        const float kneg = -p.wbfm_k;
        // segments whose lead-in reaches back before the call's start may meet earlier gains (GainEpochList)
        const GainEpochList *ep = &a.epochs[sg.ech].wbfm;
        const bool ep_reach = EPOCHS && ep->since[0] < (uint32_t)TAIL && (int64_t)sg.v0 - ST_HALO - 32 < -(int64_t)ep->since[0];
        const bool ep_any = EPOCHS && __any(ep_reach);
        // squelch magnitude bookkeeping, once per group of ST_AHEAD pieces: segments start on multiples of 128 samples of their
        // channel's stream, blocks and segment lengths are whole 128-sample units (iqd_create: block_bytes % 256 == 0; the
        // engine streams only rows of whole 128-sample units), so a group never straddles a block boundary or a segment's
        // end, and "inside the segment" is the same for the four lanes of a segment
        const int32_t mlim = MAG && sg.valid ? sg.tlen : INT32_MIN;   // groups at pos < mlim lie inside the segment
        uint32_t macc = 0;
        const uint32_t mblk0 = (uint32_t)sg.v0 / a.block_samples;
        uint32_t minblk = (uint32_t)sg.v0 - mblk0 * a.block_samples;
        uint32_t *mag_at = MAG ? a.mag_sums + (size_t)sg.ch * a.n_blocks + mblk0 : nullptr;   // the block's sum
        uint32_t gm16 = 0;                                       // the group's magnitudes, two 16-bit partial sums

        // this lane's 8 samples of the piece at `pos`: virtual sample v = v0 + 8 g + pos, from the kept tail while v < 0
        const int32_t vlane = sg.v0 + 8 * g;
        const uint8_t *base_iq = iq_ch + 2 * (int64_t)vlane, *base_tail = tail + 2 * (int64_t)vlane;
        const int32_t pos_max = vmax - vlane;
        const uint32_t *blk_list = GATED ? a.blk_lists + (size_t)sg.ch * a.n_blocks : nullptr;
        int32_t gb_v0 = 0, gb_v1 = 0;                            // GATED: virtual range of the open block this lane is in
        const uint8_t *gb_base = iq_ch;                          // ... and the address its virtual sample 0 would have
        auto piece_address = [&](int pos) -> const uint8_t * {
            const int32_t pc = pos < pos_max ? pos : pos_max;
            if (!GATED) {
                const uint8_t *base = pc < -vlane ? base_tail : base_iq;
                return base + 2 * (int64_t)pc;
            }
            const int32_t vv = vlane + pc;
            if (vv >= 0 && (vv >= gb_v1 || vv < gb_v0)) {        // (rare) into another open block
                const uint32_t blk = (uint32_t)vv / a.block_samples;
                gb_v0 = (int32_t)(blk * a.block_samples);
                gb_v1 = gb_v0 + (int32_t)a.block_samples;
                gb_base = iq_ch + 2 * ((int64_t)blk_list[blk] * a.block_samples - (int64_t)gb_v0);
            }
            return vv < 0 ? base_tail + 2 * (int64_t)pc : gb_base + 2 * (int64_t)vv;
        };
        // ST_AHEAD pieces of input in flight per wave (a piece's arithmetic is about as long as a trip to HBM under load:
        // with one piece asked for in advance the wave stood at the loop's head waiting - and 12 waves x 1 KB in flight per
        // CU cap the chip near 1.5 TB/s whatever the arithmetic costs).  The loads are the untracked ones of iqd_mfma.h:
        // the compiler's own wait placement would join the loop's back edge to vmcnt(0).  A buffer is re-asked right after
        // the last use of its old contents, in ST_AHEAD copies of the loop body with named buffers.
        // Not GATED: the address of the next piece to ask for is a running pointer - 64 bytes on per piece, held at the
        // last piece that lies wholly inside the channel's samples (what is computed from a repeated piece lies beyond the
        // segment's end and is never stored or counted), and moved from the kept tail to the call's own samples when a
        // first segment's lead-in ends (below, once per segment) - instead of piece_address()'s six operations per piece.
        const int32_t pos_last = pos_max & ~31;
        const bool from_tail = sg.v0 == 0;                       // (tile_len >= ST_MIN_TILE = ST_HALO: only a first segment's lead-in reads the tail inside the loop)
        const uint8_t *nxt = piece_address(-lead + 32 * ST_AHEAD);
        uint4 prev = st_front<ROT>(*(const uint4 *)piece_address(-lead - 32), zero);
        v4u raw[ST_AHEAD];
#pragma unroll
        for (int j = 0; j < ST_AHEAD; j++) raw[j] = gload16_untracked(piece_address(-lead + 32 * j));
        // theta' of the sample before the lead-in (a warm segment's carried state applies from the lead-in's very first
        // sample, whose delta theta needs it): the last output of the piece before, from that piece's "N" window
        float last_prev;                                         // theta'[3] of this lane's previous window
        {
            const v4i bn = v4i{(int)prev.x, (int)prev.y, (int)prev.z, (int)prev.w};
            const v4i ilo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], bn, cbias, 0, 0, 0);
            const v4i ihi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], bn, czero, 0, 0, 0);
            const v4i qlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], bn, cbias, 0, 0, 0);
            const v4i qhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], bn, czero, 0, 0, 0);
            const uint32_t ti = (uint32_t)ilo[3] + ((uint32_t)ihi[3] << 8), tq = (uint32_t)qlo[3] + ((uint32_t)qhi[3] << 8);
            // Only row 3 of the four results is used.  Left to itself the allocator hands the twelve dead registers out again
            // at once - in the squelch-gated instantiations to an UNTRACKED load issued right behind the matrix instruction,
            // which the compiler does not guard against the matrix unit's pending write (tools/isa_lint.py: lint_mfma_readers;
            // benign in practice, the load's bytes arrive long after, but not by construction).  Kept alive until the
            // compiler's own first use above has waited for the results.
            asm volatile("" :: "v"(ilo), "v"(ihi), "v"(qlo), "v"(qhi), "v"(ti), "v"(tq));
            uint32_t rr;
            asm("v_msad_u8 %0, %1, %2, 0" : "=v"(rr) : "v"(tq), "s"(0x00800000u));
            const uint32_t t = st_table_read(rr * (uint32_t)(ST_ROW_FLOATS * 4) + (bfe(ti, 16, 8) << 2));
            last_prev = u2f((t & 0x7fffffffu) | ((tq << 8) & 0x80000000u));
        }
        // one piece: `prev` = the signed bytes of the piece before, `cur` = this piece's (output).  The loop below runs two
        // copies of the body with the two buffers swapped, so that neither is ever copied.
        uint32_t n_sleeps = 0;                                   // (IQD_ST_WAITSTAT)
#if IQD_ST_TIMING
        long long tt[4] = {0, 0, 0, 0};
#endif
        auto do_piece = [&](const int q, const int j, const uint4 &prev, uint4 &cur) __attribute__((always_inline)) {
            const int pos = -ST_HALO + 32 * q;
            ST_TRACE(a.stamps, pw + 3, q, 0);
            ST_T(t0);
#ifdef IQD_ST_BURN_SIMD   // measurement build: the P waves of ONE SIMD (hardware wave % 4) issue IQD_ST_BURN_N idle vector instructions per piece
            if (((pw + ST_RINGS) & 3) == IQD_ST_BURN_SIMD) {
                float burn = 1.0f;
#pragma unroll
                for (int k = 0; k < IQD_ST_BURN_N; k++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(burn));
            }
#endif
            gload_wait<ST_AHEAD - 1>(raw[j]);                    // younger than this buffer's load: the other buffers' loads
            ST_T(t1);
            const uint4 raw_cur = as_uint4(raw[j]);              // (offset binary, as loaded: the squelch magnitudes below)
            cur = st_front<ROT>(raw_cur, zero);
            // both windows' MFMAs go out first: the second window's run under the first window's index arithmetic.
            // "S" window (the piece's first 16 outputs): lanes 0-31 this piece's first 32 bytes, lanes 32-63 the
            // previous piece's last 32; "N" window: the piece itself.
            const v4i bs = v4i{(int)(lane < 32 ? cur.x : prev.x), (int)(lane < 32 ? cur.y : prev.y),
                               (int)(lane < 32 ? cur.z : prev.z), (int)(lane < 32 ? cur.w : prev.w)};
            const v4i bn = v4i{(int)cur.x, (int)cur.y, (int)cur.z, (int)cur.w};
            v4i acc[2][4];
            acc[0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[4], bs, cbias, 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[5], bs, czero, 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[6], bs, cbias, 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[7], bs, czero, 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], bn, cbias, 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], bn, czero, 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], bn, cbias, 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], bn, czero, 0, 0, 0);
            // phase A, both windows: table addresses and the gathers.  Twice the Q15 sums + WB_BIAS: byte 2 is
            // (uint8)((int8)(acc >> 15) + 128), the table index.
            uint32_t traw[2][4], tqs[2][4];
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t ti = (uint32_t)acc[half][0][r] + ((uint32_t)acc[half][1][r] << 8);
                    const uint32_t tq = (uint32_t)acc[half][2][r] + ((uint32_t)acc[half][3][r] << 8);
                    uint32_t rr;                                 // |y| = |byte 2 of tq - 128|
                    asm("v_msad_u8 %0, %1, %2, 0" : "=v"(rr) : "v"(tq), "s"(0x00800000u));
                    const uint32_t x4 = bfe(ti, 16 + IQD_ST_FAKE_SHIFT, 8 - IQD_ST_FAKE_SHIFT) << 2;
                    uint32_t addr;                               // row |y|, column x of the half table
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(rr), "s"((uint32_t)(ST_LDS_ROW_FLOATS * 4)), "v"(x4));
                    traw[half][r] = st_table_read(addr);
                    tqs[half][r] = tq;
                }
            }
            uint32_t seen = lds_load_relaxed(consumed);          // asked early, needed only before the ring stores
            // phase B: squelch magnitudes of this lane's 8 samples, while the gathers are in flight (two packed 16-bit sums,
            // folded and booked once per group of pieces: at most 4 x 4 x 192 per half)
            if (MAG && pos >= 0) {
#if IQD_ST_LEVEL_PROBE   // TIMING PROBE ONLY (wrong magnitudes): what levelling the SIMDs could buy at most - the P waves of the three
                         // SIMDs that carry an IIR wave skip the magnitude arithmetic, those of the fourth do it four times
                if (IQD_ST_LEVEL_PROBE == 1 && (pw & 3) == 0) {   // (2: nobody does them)
                    uint4 rc = raw_cur;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        asm volatile("" : "+v"(rc.x), "+v"(rc.y), "+v"(rc.z), "+v"(rc.w));
                        gm16 = st_mag_raw_chunk(rc, gm16);
                    }
                }
#else
                gm16 = st_mag_raw_chunk(raw_cur, gm16);
#endif
            }
            // (after the last use of the buffer's old contents)
            if (GATED || !IQD_ST_RUNPTR) {
                raw[j] = gload16_untracked(piece_address(pos + 32 * ST_AHEAD));
            } else {
                raw[j] = gload16_untracked(nxt);
                nxt += pos + 32 * ST_AHEAD < pos_last ? 64 : 0;
            }
            // phase C, both windows: sign, delta theta, branch cut, K, b0
            float u[2][4];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int wpos = pos + 16 * half;
                float th[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {   // the table holds |theta|; theta' = -theta carries the sign bit of y >= 0 (index bit 7)
                    uint32_t bits;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bits) : "s"(0x7fffffffu), "v"(traw[half][r]), "v"(tqs[half][r] << 8));
                    th[r] = u2f(bits);
                }
                const float give = g == 3 ? last_prev : th[3];
                const float before = u2f((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)f2u(give)));
                last_prev = th[3];
                float kk = kneg;
                if (EPOCHS && ep_any && ep_reach) kk = -epoch_gain_search(ep, p.wbfm_k, sg.v0 + wpos);   // (rare: right after a gain change)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {   // two samples per packed operation; wrap_delta() of iqd_prims.h operation by operation
                    v2f d = {th[r] - (r == 0 ? before : th[r - 1]), th[r + 1] - th[r]};   // = -(delta theta)
                    v2f m = pk_mul_s(d, inv_2pi);
                    m.x = __builtin_rintf(m.x);
                    m.y = __builtin_rintf(m.y);
                    d = __builtin_elementwise_fma(-m, v2f{6.28318548202514648f, 6.28318548202514648f}, d);
                    d = __builtin_elementwise_fma(-m, v2f{-1.74845553146951715e-7f, -1.74845553146951715e-7f}, d);
#if IQD_RELAXED_TOL   // (timing A/B only: one rounding instead of two)
                    const v2f w = d * v2f{kk * sa.b0, kk * sa.b0};
#else
                    const v2f v = d * v2f{kk, kk};
                    const v2f w = v2f{sa.b0, sa.b0} * v;
#endif
                    u[half][r] = w.x;
                    u[half][r + 1] = w.y;
                }
            }
            // hand the 2 x 4 samples to the IIR wave: the ring's two slots hold one piece (window 0, window 1), free once
            // the IIR wave has read the previous piece.  One signal per piece in each direction.
            ST_T(t2);
            ST_TRACE(a.stamps, pw + 3, q, 1);
            while ((int32_t)(seen + (uint32_t)(ST_DEPTH - 1) - pc) < 0) {   // piece pc - ST_DEPTH has been read: its slots are free
                __builtin_amdgcn_s_sleep(IQD_ST_SLEEP_P);
                seen = lds_load_relaxed(consumed);
                if (IQD_ST_WAITSTAT) n_sleeps++;
            }
            ST_T(t3);
            ST_TRACE(a.stamps, pw + 3, q, 2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint8_t *const slot = ring_base + (ST_DEPTH > 1 ? (pc & (uint32_t)(ST_DEPTH - 1)) * (2 * ST_SLOT_BYTES) : 0u) + wr_off;
            *(u32x4 *)slot = u32x4{f2u(u[0][0]), f2u(u[0][1]), f2u(u[0][2]), f2u(u[0][3])};
            *(u32x4 *)(slot + ST_SLOT_BYTES) = u32x4{f2u(u[1][0]), f2u(u[1][1]), f2u(u[1][2]), f2u(u[1][3])};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            lds_signal(ST_DEPTH > 1 ? full + (pc & (uint32_t)(ST_DEPTH - 1)) : full);
            ST_TRACE(a.stamps, pw + 3, q, 3);
            pc++;
#if IQD_ST_TIMING
            { ST_T(t4); tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2; tt[3] += t4 - t3; }
#endif
        };
        uint4 other;
        for (int q = q_first; q < n_pieces; q += ST_AHEAD) {   // (n_pieces and q_first are multiples of 4)
#pragma unroll
            for (int j = 0; j < ST_AHEAD; j += 2) {
                do_piece(q + j, j, prev, other);
                do_piece(q + j + 1, j + 1, other, prev);
            }
            const int gpos = -ST_HALO + 32 * q;                  // the group's first sample
            if (!GATED && IQD_ST_RUNPTR && gpos + 64 * ST_AHEAD == 0)   // (recomputed here, once per segment: two registers fewer through the loop)
                nxt = from_tail ? a.iq + (size_t)sg.ch * a.ch_stride_bytes + 16 * g : nxt;   // the next piece asked for is the one at position 0
            if (MAG && gpos >= 0) {
                const uint32_t m = (gm16 & 0xffffu) + (gm16 >> 16);
                gm16 = 0;
                macc += gpos < mlim ? m : 0u;
                minblk += 32 * ST_AHEAD;
                if (minblk >= a.block_samples) {                 // the next group belongs to the next block
                    if (macc) atomicAdd(mag_at, macc);
                    macc = 0;
                    mag_at++;
                    minblk -= a.block_samples;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ST_AHEAD; j++) gload_wait<0>(raw[j]);   // the loads asked for beyond the last piece: drained before their registers move on
        if (MAG && macc) atomicAdd(mag_at, macc);
        if (IQD_ST_WAITSTAT && lane == 0) {
            atomicAdd(&a.stamps[0], (unsigned long long)n_sleeps);
            atomicAdd(&a.stamps[2], (unsigned long long)n_pieces);
        }
#if IQD_ST_TIMING
        if (lane == 0) {   // per P wave of the workgroup: [16 + pw] compute, [32 + pw] ring wait, [48 + pw] raw wait + hand-over
            atomicAdd(&a.stamps[16 + pw], (unsigned long long)tt[1]);
            atomicAdd(&a.stamps[32 + pw], (unsigned long long)tt[2]);
            atomicAdd(&a.stamps[48 + pw], (unsigned long long)(tt[0] + tt[3]));
        }
#endif
    }
}

// ---- IIR wave: 64 segments, u[n] -> PCM ------------------------------------------------------------
struct StIir {
    float y, up;
    uint32_t wlast[2];     // the last 4 (int16)y
    uint32_t wq0[2];       // the first 4 (int16)y of the window just processed
    uint32_t y1h[4];       // the last 8 stage-1 outputs
    uint32_t y2p[24];      // stage-2 outputs as pairs; variant V of a piece uses [V+1 .. V+20]
    uint32_t y2lo;         // first stage-2 output of the current piece
    int loud;              // pieces for which a |y2| > AUDIO40_SAFE stays in reach of the 40-tap window
    uint32_t n_sleeps;     // (IQD_ST_WAITSTAT)
    int ring;              // (IQD_ST_TRACE)
    unsigned long long *stamps;
#if IQD_ST_TIMING
    long long t_wait, t_seen, t_to_consumed;
#endif
};

// 16 samples of one segment: de-emphasis (IirFilter.cc:161-176 op by op), (int16), /4 with 8 taps, then one
// /4 output with 12 taps.  Returns that stage-2 output.
__device__ __forceinline__ int st_iir_window(const StreamArgs &sa, StIir &s, const float (&u)[16], int c14, int c15)
{
    uint32_t wq[10];                                   // 20 samples: the last quad, then this window's four
    float y = s.y, up = s.up;
    const float a1 = sa.a1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        IQD_IIR_STEP(u[2 * k])
        const float y_even = y;
        IQD_IIR_STEP(u[2 * k + 1])
        wq[2 + k] = cast_pack_i16_bounded(y_even, y);
    }
    s.y = y;
    s.up = up;
    wq[0] = s.wlast[0];
    wq[1] = s.wlast[1];
    s.wlast[0] = wq[8];
    s.wlast[1] = wq[9];
    s.wq0[0] = wq[2];
    s.wq0[1] = wq[3];
    // stage 1 with DOUBLED taps (|2 h| <= 12892 fits int16; the sums stay below 2^31: 2 (16384 + 29126 * 32768)): the
    // output (acc >> 15) then sits in the high half of the accumulator and two of them pack with one v_perm_b32, no shifts.
    // The accumulators start from a register holding the rounding term (c15 = 1 << 15, c14 = 1 << 14): as constants the
    // compiler re-materialises them with a move in front of every chain.
    uint32_t y1[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {                      // window x[4m-4 .. 4m+3] <-> taps h[7 .. 0]
#if IQD_ST_DOT2_FROM
        int acc = dot2_from(wq[2 * o], sa.d1p2[0], c15);
#else
        int acc = c15;
        acc = dot2(wq[2 * o], ST_TAP(d1p2, 0), acc);
#endif
        acc = dot2(wq[2 * o + 1], ST_TAP(d1p2, 1), acc);
        acc = dot2(wq[2 * o + 2], ST_TAP(d1p2, 2), acc);
        acc = dot2(wq[2 * o + 3], ST_TAP(d1p2, 3), acc);
        y1[o] = (uint32_t)acc;
    }
    // stage 2: output k from y1[4k-8 .. 4k+3], 12 taps, newest pair first
    const uint32_t d[6] = {s.y1h[0], s.y1h[1], s.y1h[2], s.y1h[3], pack_hi16(y1[0], y1[1]), pack_hi16(y1[2], y1[3])};
#if IQD_ST_DOT2_FROM
    int acc = dot2_from(d[5], sa.p12p[0], c14);
#pragma unroll
    for (int q = 1; q < 6; q++) acc = dot2(d[5 - q], ST_TAP(p12p, q), acc);
#else
    int acc = c14;
#pragma unroll
    for (int q = 0; q < 6; q++) acc = dot2(d[5 - q], ST_TAP(p12p, q), acc);
#endif
    s.y1h[0] = d[2];
    s.y1h[1] = d[3];
    s.y1h[2] = d[4];
    s.y1h[3] = d[5];
    return acc >> 15;
}

// /2, 40 taps over the pairs p[1..20] (p[20] newest): without clamps when no loud value is in reach, else in
// the reference's order with the clamp after every MAC (Decimator_int16.cc:176-238)
template <int V>
__device__ __forceinline__ int st_audio(const StreamArgs &sa, const StIir &s, bool quiet, int c14)
{
    int acc = c14;
    if (quiet) {
#if IQD_ST_DOT2_FROM
        acc = dot2_from(s.y2p[V + 20], sa.a40p[0], c14);
#pragma unroll
        for (int q = 1; q < 20; q++) acc = dot2(s.y2p[V + 20 - q], ST_TAP(a40p, q), acc);
#else
#pragma unroll
        for (int q = 0; q < 20; q++) acc = dot2(s.y2p[V + 20 - q], ST_TAP(a40p, q), acc);
#endif
    } else {
#pragma unroll
        for (int q = 0; q < 20; q++) {
            acc = clamp_q30(dot2(s.y2p[V + 20 - q], ST_TAP(a40p, q) & 0xffff0000u, acc));   // newer sample of the pair: h[2q]
            acc = clamp_q30(dot2(s.y2p[V + 20 - q], ST_TAP(a40p, q) & 0x0000ffffu, acc));   // older: h[2q+1]
        }
    }
    return acc >> 15;
}

struct StIirSeg {
    StSeg sg;
    int32_t back;          // tile 0: the carried exact state sits `back` samples before the tile; else -1
    float cy_y, cy_u;
    int32_t rec_pos;
    WbfmRecord rec;
    int16_t *pcm_row;
    StHist *hist;          // this segment's boundary record
    int c14, c15;          // 1 << 14, 1 << 15 in registers (the decimators' rounding terms)
};

// (q.sg.valid bit 1: the segment keeps the channel's restart state - WbfmRecord::pad - at q.rec_pos, and takes its own record
//  where every full cold segment does, rec_pos_uniform)
__device__ __forceinline__ void st_iir_marks(const ChainLaunch &a, StIirSeg &q, StIir &s, int pos, int rec_pos_uniform)
{
    if (q.back >= 0) {                                 // warm segment: silence before the carried state applies
        if (pos == -q.back) { s.y = q.cy_y; s.up = q.cy_u; }
        else if (pos < -q.back) { s.y = 0.f; s.up = 0.f; }
    } else if (pos == 0) {                             // cold segment: the warmed-up state, checked against the predecessor's end
        q.rec.y_in = s.y;
    }
    if (pos == q.rec_pos) {
        if (q.sg.valid & 2u) {   // (parked in the spare words of the segment's boundary record, whose address is at hand: no register for it)
            q.hist->pad[0] = f2u(s.y);
            q.hist->pad[1] = f2u(s.up);
        } else {
            q.rec.y_out = s.y;
            q.rec.u_out = s.up;
        }
    }
    if ((q.sg.valid & 2u) && pos == rec_pos_uniform) { q.rec.y_out = s.y; q.rec.u_out = s.up; }
    if (pos == q.sg.tlen) {                            // (a multiple of 128: the pair history sits in y2p[0..19] here)
        q.rec.y_end = s.y;
        q.rec.u_end = s.up;
        if (q.sg.valid) {
            ST_STORE16(q.hist->y1_last, s.y1h[0], s.y1h[1], s.y1h[2], s.y1h[3]);
            *(u32x2 *)q.hist->w_last = u32x2{s.wlast[0], s.wlast[1]};
#pragma unroll
            for (int k = 0; k < 20; k += 4) ST_STORE16(&q.hist->y2_last[k], s.y2p[k], s.y2p[k + 1], s.y2p[k + 2], s.y2p[k + 3]);
        }
    }
}

// FAST: every segment of the wave is cold and of full length (or not there at all), so the few things that happen at
// particular positions happen at the SAME position in every lane - one scalar compare per window instead of a dozen
// per-lane compare-and-select operations - and the lead-in has already been run by st_iir_lead_in().
template <int V, bool FAST>
__device__ __forceinline__ int st_iir_piece(const ChainLaunch &a, const StreamArgs &sa, uint8_t *ring_base, const uint32_t *full, uint32_t *consumed,
                                             uint32_t &wg, StIirSeg &q, StIir &s, int pos, uint32_t rd_off0, uint32_t rd_swz, int lane,
                                             int rec_pos_uniform)
{
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int wpos = pos + 16 * half;
        if (half == 0) {   // the four P waves of the ring have written piece `wg` (both windows) when `full` reaches 4 (wg + 1)
            const uint32_t target = 4u * (wg / (uint32_t)ST_DEPTH + 1u);
            const uint32_t *const fullp = ST_DEPTH > 1 ? full + (wg & (uint32_t)(ST_DEPTH - 1)) : full;
            ST_T(tw0);
            while ((int32_t)(lds_load_relaxed(fullp) - target) < 0) {
                __builtin_amdgcn_s_sleep(IQD_ST_SLEEP_I);
                if (IQD_ST_WAITSTAT) s.n_sleeps++;
            }
#if IQD_ST_TIMING
            { ST_T(tw1); s.t_wait += tw1 - tw0; s.t_seen = tw1; }
#endif
            ST_TRACE(s.stamps, s.ring, wg, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint8_t *slot = ring_base + (ST_DEPTH > 1 ? (wg & (uint32_t)(ST_DEPTH - 1)) * (2 * ST_SLOT_BYTES) : 0u) + half * ST_SLOT_BYTES + rd_off0;
        float u[16];
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const u32x4 v = *(const u32x4 *)(slot + (((uint32_t)gq ^ rd_swz) << 4));
            u[4 * gq] = u2f(v.x); u[4 * gq + 1] = u2f(v.y); u[4 * gq + 2] = u2f(v.z); u[4 * gq + 3] = u2f(v.w);
        }
        if (half == 1) {   // both windows read: the ring is free for the next piece
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the reads have returned
            lds_signal(consumed);
            ST_TRACE(s.stamps, s.ring, wg, 1);
            wg++;
#if IQD_ST_TIMING
            { ST_T(tc); s.t_to_consumed += tc - s.t_seen; }
#endif
        }
        if (!FAST) st_iir_marks(a, q, s, wpos, rec_pos_uniform);
        else if (wpos == rec_pos_uniform) { q.rec.y_out = s.y; q.rec.u_out = s.up; }
        const int y2 = st_iir_window(sa, s, u, q.c14, q.c15);
        if (wpos >= 0 && wpos < 48 && q.sg.valid) {    // (uniform) the segment's first values, for the boundary fix-up
            if (wpos == 0) ST_STORE16(q.hist->w_first, s.wq0[0], s.wq0[1], s.y1h[2], s.y1h[3]);   // (w_first, y1_first[0..1])
            else *(u32x2 *)&q.hist->y1_first[wpos >> 3] = u32x2{s.y1h[2], s.y1h[3]};
        }
        const uint32_t mag = (uint32_t)(y2 < 0 ? -y2 : y2);
        if (mag > (uint32_t)AUDIO40_SAFE) s.loud = 21;
        if (half == 0) s.y2lo = (uint32_t)y2;
        else s.y2p[V + 20] = pack_lo16(s.y2lo, (uint32_t)y2);
    }
    if (V == 3 && pos >= 96 && pos < 768 && q.sg.valid)   // the four pairs of this run of 128 samples (pos = its last piece)
        ST_STORE16(&q.hist->y2_first[(pos - 96) >> 5], s.y2p[20], s.y2p[21], s.y2p[22], s.y2p[23]);
    ST_TRACE(s.stamps, s.ring, wg - 1, 2);
    const bool quiet = !__any(s.loud > 0);
    const int pcm = quiet ? st_audio<V>(sa, s, true, q.c14) : st_audio<V>(sa, s, false, q.c14);
    if (s.loud > 0) s.loud--;
    return pcm;
}

// The lead-in of a wave whose segments are all cold: only the de-emphasis recurrence (its state is what the lead-in is
// for; a cold segment's decimators start with histories that the boundary fix-up replaces anyway).
__device__ __forceinline__ void st_iir_lead_in(const StreamArgs &sa, uint8_t *ring_base, const uint32_t *full, uint32_t *consumed,
                                               uint32_t &wg, StIir &s, uint32_t rd_off0, uint32_t rd_swz, int lead)
{
    float y = s.y, up = s.up;
    const float a1 = sa.a1;
    for (int piece = 0; piece < lead / 32; piece++) {   // (the ring's P waves skip the same pieces: st_p_wave, q_first)
        const uint32_t target = 4u * (wg / (uint32_t)ST_DEPTH + 1u);
        while ((int32_t)(lds_load_relaxed(ST_DEPTH > 1 ? full + (wg & (uint32_t)(ST_DEPTH - 1)) : full) - target) < 0) __builtin_amdgcn_s_sleep(IQD_ST_SLEEP_I);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        u32x4 v[8];
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++)
                v[4 * half + gq] = *(const u32x4 *)(ring_base + (ST_DEPTH > 1 ? (wg & (uint32_t)(ST_DEPTH - 1)) * (2 * ST_SLOT_BYTES) : 0u) +
                                                    half * ST_SLOT_BYTES + rd_off0 + (((uint32_t)gq ^ rd_swz) << 4));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the reads have returned
        lds_signal(consumed);
        wg++;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            IQD_IIR_STEP(u2f(v[k].x)) IQD_IIR_STEP(u2f(v[k].y)) IQD_IIR_STEP(u2f(v[k].z)) IQD_IIR_STEP(u2f(v[k].w))
        }
    }
    s.y = y;
    s.up = up;
}

// lead_cold: the lead-in of a ring whose segments are all cold and of full length (IQD_ST_COLD_HALO where the launch's P waves
// take it, ST_HALO otherwise)
__device__ __forceinline__ void st_iir_wave(const ChainLaunch &a, const StreamArgs &sa, uint8_t *lds, uint32_t *sync,
                                            int ring, int lane, int lead_cold)
{
    uint8_t *ring_base = lds + ST_TABLE_BYTES + ring * (ST_RING_SLOTS * ST_SLOT_BYTES);
    const uint32_t *full = sync + ring * 8;
    uint32_t *consumed = sync + ring * 8 + 4;
    const uint32_t rd_off0 = (uint32_t)lane * 64u, rd_swz = ((uint32_t)lane >> 2) & 3u;
    const int n_pieces = (ST_HALO + (int)a.tile_len) >> 5;       // a multiple of 4
    if (ring >= (int)sa.rings) return;
    const uint32_t wg_segs = 64u * sa.rings;
    uint32_t wg = 0;                                             // pieces read so far (all rounds)
    for (uint32_t round = 0; round < sa.rounds; round++) {
        if ((round * chain_wgs(a) + chain_wg(a)) * wg_segs >= st_id_count(sa)) break;
        const uint32_t sid = (round * chain_wgs(a) + chain_wg(a)) * wg_segs + ring * 64 + lane;
        StIirSeg q;
        int rot_unused;
        q.sg = st_segment_of(a, sa, sid, rot_unused);
        q.pcm_row = a.pcm + (size_t)q.sg.ch * a.pcm_stride;
        q.hist = sa.hist + (q.sg.valid ? (size_t)q.sg.li * a.tiles_per_ch + q.sg.tile : 0);
        q.back = -1;
        q.cy_y = q.cy_u = 0.f;
        q.c14 = 1 << 14;
        q.c15 = 1 << 15;
        asm volatile("" : "+v"(q.c14), "+v"(q.c15));
        int32_t carried_back = ST_HALO;
        if (q.sg.tile == 0) {
            const WbfmCarry cy = a.wbfm_carry[q.sg.ech];
            q.back = cy.back;
            q.cy_y = cy.y;
            q.cy_u = cy.u;
            carried_back = cy.back;
        }
        // where the segment's record is taken, and whether it keeps the channel's restart state for a short last segment behind it
        // (outside the fast path): iqd_wbfm.h, st_rec_plan - shared with the CPU tier's model of the hand-over
        const StRecPlan rp = st_rec_plan(q.sg.valid, q.sg.tile, q.sg.v0, q.sg.tlen, q.sg.vlen, carried_back);
        q.rec_pos = rp.park_pos;   // (= rec_pos unless the segment keeps: then its own record sits where every full segment's does, rec_pos_uniform)
        q.rec.y_in = q.cy_y;
        q.rec.y_out = q.cy_y;
        q.rec.u_out = q.cy_u;
        q.rec.back_out = rp.back_out;
        q.rec.y_end = 0.f;
        q.rec.u_end = 0.f;
        q.rec.pad[0] = q.rec.pad[1] = 0;
        const bool keeps_restart = rp.keeps_restart != 0;
        if (keeps_restart) q.sg.valid |= 2u;
        StIir s;
        s.y = 0.f; s.up = 0.f;
        s.wlast[0] = s.wlast[1] = 0;
        s.wq0[0] = s.wq0[1] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) s.y1h[k] = 0;
#pragma unroll
        for (int k = 0; k < 24; k++) s.y2p[k] = 0;
        s.y2lo = 0;
        s.loud = 0;
        s.n_sleeps = 0;
        s.ring = ring;
        s.stamps = a.stamps;
#if IQD_ST_TIMING
        s.t_wait = 0;
        s.t_seen = 0;
        s.t_to_consumed = 0;
        const long long t_iir0 = clock64();
#endif
        uint32_t pbuf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // wide stores need whole 512-sample groups per lane (tile_len a multiple of 512) and 32-byte aligned rows
        const bool wide = (a.tile_len & 511u) == 0 && (((uintptr_t)a.pcm | (a.pcm_stride * 2)) & 31u) == 0;
        // (a segment that is not there has tlen 0 and stores nothing: it may run along with any kind of wave)
        // (the P waves evaluate the same predicate, st_cold_and_full(): a ring's waves must agree on its lead-in)
        const bool fast = __all(!q.sg.valid || (q.back < 0 && q.sg.tlen == (int32_t)a.tile_len && !keeps_restart)) != 0;
        const int rec_pos_uniform = (int)a.tile_len - FORCED_BACK;
        int pq0 = 0;
        if (fast) {
            st_iir_lead_in(sa, ring_base, full, consumed, wg, s, rd_off0, rd_swz, lead_cold);
            q.rec.y_in = s.y;                                    // the warmed-up state, checked against the predecessor's end
            pq0 = ST_HALO / 32;
        }
        for (int pq = pq0; pq < n_pieces; pq += 4) {
            const int pos = -ST_HALO + 32 * pq;
            int p0, p1, p2, p3;
            if (fast) {
                p0 = st_iir_piece<0, true>(a, sa, ring_base, full, consumed, wg, q, s, pos, rd_off0, rd_swz, lane, rec_pos_uniform);
                p1 = st_iir_piece<1, true>(a, sa, ring_base, full, consumed, wg, q, s, pos + 32, rd_off0, rd_swz, lane, rec_pos_uniform);
                p2 = st_iir_piece<2, true>(a, sa, ring_base, full, consumed, wg, q, s, pos + 64, rd_off0, rd_swz, lane, rec_pos_uniform);
                p3 = st_iir_piece<3, true>(a, sa, ring_base, full, consumed, wg, q, s, pos + 96, rd_off0, rd_swz, lane, rec_pos_uniform);
            } else {
                p0 = st_iir_piece<0, false>(a, sa, ring_base, full, consumed, wg, q, s, pos, rd_off0, rd_swz, lane, rec_pos_uniform);
                p1 = st_iir_piece<1, false>(a, sa, ring_base, full, consumed, wg, q, s, pos + 32, rd_off0, rd_swz, lane, rec_pos_uniform);
                p2 = st_iir_piece<2, false>(a, sa, ring_base, full, consumed, wg, q, s, pos + 64, rd_off0, rd_swz, lane, rec_pos_uniform);
                p3 = st_iir_piece<3, false>(a, sa, ring_base, full, consumed, wg, q, s, pos + 96, rd_off0, rd_swz, lane, rec_pos_uniform);
            }
            // 128 samples = 4 PCM samples = 8 bytes.  Where the row allows it they are collected over 512 samples and
            // leave as one aligned 32-byte sector (segments start on multiples of 512 samples of their channel's
            // stream, so the phase below is the same for every lane); else 8 bytes at a time.
            const uint32_t w0 = pack_lo16((uint32_t)p0, (uint32_t)p1), w1 = pack_lo16((uint32_t)p2, (uint32_t)p3);
            const int phase = (pos >> 7) & 3;
            if (phase == 0) { pbuf[0] = w0; pbuf[1] = w1; }
            else if (phase == 1) { pbuf[2] = w0; pbuf[3] = w1; }
            else if (phase == 2) { pbuf[4] = w0; pbuf[5] = w1; }
            else { pbuf[6] = w0; pbuf[7] = w1; }
            if (q.sg.valid) {
                if (!wide) {
                    if (pos >= 0 && pos < q.sg.tlen) *(u32x2 *)(q.pcm_row + ((q.sg.v0 + pos) >> 5)) = u32x2{w0, w1};
                } else if (phase == 3) {
                    const int g0 = pos - 384;                    // the group [g0, g0 + 512)
                    int16_t *dst = q.pcm_row + ((q.sg.v0 + g0) >> 5);
                    if (g0 >= 0 && g0 + 512 <= q.sg.tlen) {
                        ST_STORE16(dst, pbuf[0], pbuf[1], pbuf[2], pbuf[3]);
                        ST_STORE16(dst + 8, pbuf[4], pbuf[5], pbuf[6], pbuf[7]);
                    } else {                                     // a segment's ragged end (or the lead-in): quarter by quarter
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (g0 + 128 * k >= 0 && g0 + 128 * k < q.sg.tlen) ((u32x2 *)dst)[k] = u32x2{pbuf[2 * k], pbuf[2 * k + 1]};
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 20; k++) s.y2p[k] = s.y2p[k + 4];
        }
        st_iir_marks(a, q, s, (int)a.tile_len, rec_pos_uniform);
        if (q.sg.valid) {
            WbfmRecord *const r = a.records + (size_t)q.sg.li * a.tiles_per_ch + q.sg.tile;
            if (q.sg.valid & 2u) {   // (the restart state this segment kept: st_iir_marks)
                q.rec.pad[0] = q.hist->pad[0];
                q.rec.pad[1] = q.hist->pad[1];
            } else if (!(q.sg.vlen - q.sg.v0 > q.sg.tlen)) {
                q.rec.pad[0] = WBFM_REC_STREAMED;   // the channel's last segment
            }
            *r = q.rec;
        }
        if (IQD_ST_WAITSTAT && lane == 0) {
            atomicAdd(&a.stamps[1], (unsigned long long)s.n_sleeps);
            atomicAdd(&a.stamps[3], (unsigned long long)n_pieces);
        }
#if IQD_ST_TIMING
        if (lane == 0) {   // per IIR wave: [28 + ring] wait, [44 + ring] total
            atomicAdd(&a.stamps[28 + ring], (unsigned long long)s.t_wait);
            atomicAdd(&a.stamps[44 + ring], (unsigned long long)(clock64() - t_iir0));
            atomicAdd(&a.stamps[60 + ring], (unsigned long long)s.t_to_consumed);
        }
#endif
    }
}

// The workgroup's whole life: table into LDS, then the waves take their roles.  (A function of its own: the kernel below
// calls it, and so does the launch that runs several families side by side, iqd_stream_mixed.hip.)
template <int ROT, bool MAG, bool EPOCHS, bool GATED>
__device__ __forceinline__ void wbfm_stream_body(const ChainLaunch &a, const StreamArgs &sa, uint8_t *st_lds)
{
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)st_lds != 0u) __builtin_trap();   // st_table_read()
    uint32_t *sync = (uint32_t *)(st_lds + ST_TABLE_BYTES + ST_RINGS * ST_RING_SLOTS * ST_SLOT_BYTES);
    const int tid = (int)threadIdx.x;
    {   // the table: every thread's loads in flight together, then its stores (round 5: written as one loop, the compiler waited for
        // each load before the next - nine trips to L2 one after the other at the start of every launch, 256 workgroups at once)
        constexpr int N16 = ST_TABLE_BYTES / 16, TRIPS = (N16 + ST_THREADS - 1) / ST_THREADS;
        uint4 t[TRIPS];
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = tid + k * ST_THREADS;
            t[k] = ((const uint4 *)sa.half_lut)[i < N16 ? i : N16 - 1];
        }
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = tid + k * ST_THREADS;
            if (i < N16) ((uint4 *)st_lds)[i] = t[k];
        }
    }
    if (tid < ST_SYNC_WORDS) sync[tid] = 0;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    uint32_t pc = 0;                                             // pieces this wave's ring has seen (all rounds)
    if (wave < ST_RINGS) st_iir_wave(a, sa, st_lds, sync, wave, lane, ROT != 2 && !GATED && !EPOCHS ? IQD_ST_COLD_HALO : ST_HALO);
    else if (ROT != 2) st_p_wave<ROT, MAG, EPOCHS, GATED>(a, sa, st_lds, sync, wave - ST_RINGS, lane, pc);
    else {   // channels of several rotation selectors: the groups in their order
        st_p_wave<1, MAG, EPOCHS, GATED, true>(a, sa, st_lds, sync, wave - ST_RINGS, lane, pc);
        st_p_wave<0, MAG, EPOCHS, GATED, true>(a, sa, st_lds, sync, wave - ST_RINGS, lane, pc);
        st_p_wave<-1, MAG, EPOCHS, GATED, true>(a, sa, st_lds, sync, wave - ST_RINGS, lane, pc);
    }
}

#ifndef IQD_STREAM_BODIES_ONLY
template <int ROT, bool MAG, bool EPOCHS, bool GATED>
__global__ __launch_bounds__(ST_THREADS, 4) void wbfm_stream_kernel(const ChainLaunch a, const StreamArgs sa)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t st_lds[];
    wbfm_stream_body<ROT, MAG, EPOCHS, GATED>(a, sa, st_lds);
}
#endif

#ifndef IQD_STREAM_BODIES_ONLY
__global__ __launch_bounds__(FIX_THREADS) void wbfm_stream_fixup_kernel(const ChainLaunch a, const StreamArgs sa)
{
    __shared__ FixLds fl;
    wbfm_stream_fixup_body(a, sa, blockIdx.x, fl);
}

hipError_t launch_wbfm_stream_fixup(const ChainLaunch &a, const StreamArgs &sa, hipStream_t s)
{
    hipLaunchKernelGGL(wbfm_stream_fixup_kernel, dim3((sa.n_segments + FIX_SEGS - 1) / FIX_SEGS), dim3(FIX_THREADS), 0, s, a, sa);
    return hipGetLastError();
}

typedef void (*StKernel)(const ChainLaunch, const StreamArgs);
// [rotation selector -1, 0, +1][0: no magnitudes, 1: squelch magnitudes in the kernel, 2: squelch-gated launch][gain epochs
// within reach of a lead-in]
#define ST_K(R) {{wbfm_stream_kernel<R, false, false, false>, wbfm_stream_kernel<R, false, true, false>}, \
                 {wbfm_stream_kernel<R, true, false, false>, wbfm_stream_kernel<R, true, true, false>},   \
                 {wbfm_stream_kernel<R, false, false, true>, wbfm_stream_kernel<R, false, true, true>}}
static const StKernel st_kernels[4][3][2] = {ST_K(-1), ST_K(0), ST_K(1), ST_K(2)};   // [3]: selectors by group (StreamArgs::grouped)
#undef ST_K

// One workgroup takes nearly all of a CU's LDS; the attribute belongs to the current device's code object and is set
// once per engine by iqd_create (serialised there).
hipError_t init_wbfm_stream_kernels()
{
    for (int r = 0; r < 4; r++)
        for (int g = 0; g < 6; g++) {
            const hipError_t e = hipFuncSetAttribute((const void *)st_kernels[r][g >> 1][g & 1], hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

hipError_t launch_wbfm_stream(const ChainLaunch &a, const StreamArgs &sa, int rotation, bool mag, bool epochs, uint32_t grid, hipStream_t s)
{
    const int variant = a.vlen_gated ? 2 : (mag ? 1 : 0);
    hipLaunchKernelGGL(st_kernels[sa.grouped ? 3 : rotation < 0 ? 0 : rotation > 0 ? 2 : 1][variant][epochs ? 1 : 0], dim3(grid), dim3(ST_THREADS), ST_LDS_BYTES, s, a, sa);
    return hipGetLastError();
}

#endif   // IQD_STREAM_BODIES_ONLY

}  // namespace iqd
