// plan_call(): see iqd_plan.h.  Host-only, no HIP; also compiled into tests/emu for the CPU test tier.
#include "iqd_plan.h"

#include <algorithm>

#include "iqdemod.h"
#include "iqd_stream.h"

namespace iqd {

// samples per segment from which a launch of family f takes its streaming pipeline (see STREAM_MIN_SEG_*)
// several_families: 0 a call of one family, 1 several families in one launch, 2 several families as kernels on streams
uint64_t stream_min_seg(const PlanKnobs &k, int f, uint64_t vlen, int several_families)
{
    if (k.env_stream_min_seg) return k.env_stream_min_seg;
    if (several_families == 2) return STREAM_MIN_SEG_FORKED;
    if (several_families) return vlen < 8192 ? STREAM_MIN_SEG_MIXED_SHORT : STREAM_MIN_SEG_MIXED;
    if (f == FAM_WBFM) return STREAM_MIN_SEG_WBFM;
    if (f == FAM_FM) return STREAM_MIN_SEG_FM;
    if (f == FAM_AM) return vlen <= 16384 ? STREAM_MIN_SEG_AM_SHORT : STREAM_MIN_SEG_AM;
    return vlen <= 16384 ? STREAM_MIN_SEG_SSB_SHORT : STREAM_MIN_SEG_SSB;
}

namespace {

// lead-in samples every segment of family f's streaming launch runs when its cold segments are shifted by `shift` (d4_geom)
uint32_t halo_of(int f, uint32_t shift)
{
    return f == FAM_WBFM ? (uint32_t)ST_HALO : (uint32_t)d4_full_halo(f) - shift;
}

// the segment ids of a streaming launch grouped by rotation selector, each group padded to a multiple of 16
uint32_t group_ids(const FamilyShape &s, FamilyPlan &p)
{
    uint32_t at = 0, li0 = 0;
    for (int r = 0; r < 3; r++) {   // the channel list is sorted +Fs/4, none, -Fs/4 (rebuild_lists)
        p.group_start[r] = at;
        p.group_li0[r] = li0;
        p.group_nseg[r] = s.rot_count[r] * p.tiles_per_ch;
        at += (p.group_nseg[r] + 15u) / 16u * 16u;
        li0 += s.rot_count[r];
    }
    p.group_start[3] = at;
    return at;
}

// The segments a streaming launch of `slots` segment slots cuts family s's rows into - the ONE search both rings_of() and
// plan_once() use (they each had a copy): the shortest segments that fit one round (plan_stream); when the rotation
// groups' padding to 16 ids pushes an exact fit into a second round, once more with 48 slots spare.  Fills p.tile_len,
// p.tiles_per_ch and (grouped) the group_* fields; returns the segment ids the launch needs.
uint32_t fit_stream(const FamilyShape &s, uint32_t vlen, uint32_t slots, uint32_t granule, bool grouped, uint32_t shift, FamilyPlan &p)
{
    for (uint32_t spare = 0;; spare += 48) {
        const TilePlan sp = plan_stream(vlen, s.n_list, grouped && slots > spare ? slots - spare : slots, granule, shift);
        p.tile_len = sp.tile_len;
        p.tiles_per_ch = sp.tiles_per_ch;
        const uint32_t ids = grouped ? group_ids(s, p) : s.n_list * sp.tiles_per_ch;
        if (!grouped || ids <= slots || spare >= 48 || s.n_list >= slots) return ids;
    }
}

// Rings of 64 segments per workgroup of family f's streaming launch (StreamArgs::rings), round 5.  A launch too small to give
// every CU a full workgroup of three rings used to fill a part of the chip with full workgroups of minimum-length segments;
// spread over all CUs as workgroups of one or two rings its segments are longer (less lead-in per sample) and a ring's pieces go
// by faster with fewer rings beside it: per piece 0.80 (two rings) and 0.62 (one) of the three-ring time for FM and WBFM, 0.72
// and 0.50 for AM / SSB (whose P waves are the lighter), fitted to tools/rings_probe.sh on 256 CUs (256 to 16 384 channels x
// 2^14 .. 2^16; profiles/r5_rings_probe.txt: the rule picks the fastest arrangement in 14 of the 15 cases measured, second by
// 2 % in the other).  The estimate is pieces per segment x that factor.  Streaming launches of 1024 channels x 2^14: AM 0.076 ->
// 0.064 ms per step, FM 0.087 -> 0.076, WBFM 0.124 -> 0.104; 4096 x 2^14: AM 0.085 -> 0.081, WBFM 0.164 -> 0.158; larger: three rings.
uint32_t rings_of(const PlanKnobs &k, const CallShape &c, int f, uint32_t wgs, bool fused, bool grouped, uint32_t shift)
{
    if (fused) return (uint32_t)ST_RINGS;   // (the shares of the one launch are planned in workgroups of three rings: plan_fused_by_time)
    if (k.env_rings >= 1 && k.env_rings <= (uint32_t)ST_RINGS) return k.env_rings;
    static const float heavy[4] = {0.f, 0.62f, 0.80f, 1.0f}, light[4] = {0.f, 0.50f, 0.72f, 1.0f};
    const float *per_piece = f == FAM_AM || f == FAM_SSB ? light : heavy;
    const FamilyShape &s = c.fam[f];
    const uint32_t granule = f == FAM_WBFM ? k.env_stream_gran : k.env_d4_gran;
    uint32_t best = (uint32_t)ST_RINGS;
    float best_t = 0.f;
    for (uint32_t r = (uint32_t)ST_RINGS; r >= 1; r--) {
        FamilyPlan tmp;
        const uint32_t ids = fit_stream(s, c.vlen, wgs * 64u * r, granule, grouped, shift, tmp);
        if (ids > wgs * 64u * r) continue;                                 // (a second round: never better)
        const float t = per_piece[r] * (float)(tmp.tile_len + halo_of(f, shift));
        if (r == (uint32_t)ST_RINGS || t < (shift ? 1.0f : 0.99f) * best_t) { best = r; best_t = t; }   // (short lead-ins: a tie goes to the fewer rings - AM 1024 x 2^14: 0.0584 ms on two, round 6)
    }
    return best;
}

// One pass over the call with the one-launch arrangement allowed or not.  Returns false when `allow_fused` was taken and a
// family then fell off its streaming pipeline - which the predicates below exclude by construction; plan_call() then plans
// again without it, BEFORE anything is queued (round 4 had this as a guard inside the launch loop, behind the pre-pass).
bool plan_once(const PlanKnobs &k, const CallShape &c, bool allow_fused, CallPlan &out)
{
    out = CallPlan{};
    const uint32_t vlen = c.vlen;
    for (int f = 0; f < FAM_COUNT; f++) {
        out.fam[f].present = c.fam[f].n_list != 0;
        out.n_fams += out.fam[f].present ? 1 : 0;
    }
    out.forked = out.n_fams > 1;

    // families by estimated cost, dearest first
    float cost[FAM_COUNT], total = 0.f;
    for (int f = 0; f < FAM_COUNT; f++) {
        cost[f] = k.fam_weight[f] * (float)c.fam[f].n_list;
        total += cost[f];
        out.order[f] = f;
    }
    std::sort(out.order, out.order + FAM_COUNT, [&](int x, int y) { return cost[x] > cost[y]; });

    // Several families side by side: each one's persistent workgroups take a share of the CUs in proportion to its
    // estimated cost, so that the families' streaming kernels run at the same time (a CU's LDS holds one such
    // workgroup) on longer segments - less lead-in overhead, which is what small families pay most for.  (Mixed
    // configuration, 4096 channels x 2^16: 0.45 ms per step with every family on all CUs in turn, 0.39 with shares.)
    // Only when every family of the call will take its streaming kernel - here that means: brings enough samples for ITS
    // share of the CUs; tile kernels know nothing of shares, and a streaming kernel held to its share beside them lost
    // 12-16 % at 2500-3000 mixed channels.
    uint32_t share[FAM_COUNT];
    for (int f = 0; f < FAM_COUNT; f++) share[f] = k.n_cus;
    auto every_family_streams = [&](int arrangement) {   // 1: as ranges of one launch, 2: as kernels on streams (stream_min_seg)
        bool all = !(k.flags & IQD_F_WBFM_TILES) && vlen % 128 == 0 && k.env_path >= 0;
        for (int f = 0; f < FAM_COUNT && all; f++) {
            const uint64_t n_f = c.fam[f].n_list;
            if (!n_f) continue;
            const bool forced = (k.flags & IQD_F_WBFM_STREAM) != 0;
            const float due = total > 0.f ? (float)(k.n_cus - 16) * cost[f] / total : (float)k.n_cus;
            if (!forced && (float)((uint64_t)vlen * n_f) < due * (float)(ST_SEGS * stream_min_seg(k, f, vlen, arrangement))) all = false;
            if ((f == FAM_AM || f == FAM_SSB) && vlen / 32 < k.env_am_stream_min) all = false;
        }
        return all;
    };
    bool shares_on = out.forked && every_family_streams(1) && total > 0.f && k.n_cus >= 64 && !k.env_full_grid;
    // ONE launch for all of them (iqd_stream_mixed.hip) when each family can take its streaming pipeline in the plain
    // instantiation: the WBFM channels of one rotation selector and without a gain change in reach of a lead-in,
    // gains below the "integer indefinite" bounds, AM / SSB rows that the one-wave DC pass takes
    bool fused = allow_fused && shares_on && !k.env_mixed_forked && k.env_path == 0 && !k.env_stream_wgs && !(k.flags & IQD_F_WBFM_STREAM);
    if (fused && c.fam[FAM_WBFM].n_list) {
        const FamilyShape &w = c.fam[FAM_WBFM];
        const bool one_selector = w.rot_count[0] == w.n_list || w.rot_count[1] == w.n_list || w.rot_count[2] == w.n_list;
        fused = k.stream_ok && one_selector && w.cast_bounded && !w.epochs_in_reach;
    }
    if (fused && c.fam[FAM_FM].n_list) fused = c.fam[FAM_FM].cast_bounded;
    if (fused && (c.fam[FAM_AM].n_list || c.fam[FAM_SSB].n_list))
        fused = vlen / 32 >= k.env_am_stream_min && (c.pcm_per_ch + k.dc_tile - 1) / k.dc_tile < 2;
    if (shares_on && !fused) shares_on = every_family_streams(2);   // (the kernels-on-streams arrangement pays from larger calls only)
    if (shares_on && fused) {
        FusedFamily ff[FAM_COUNT];
        for (int f = 0; f < FAM_COUNT; f++) {
            for (int r = 0; r < 3; r++) ff[f].rot_count[r] = c.fam[f].n_list ? c.fam[f].rot_count[r] : 0u;
            ff[f].shift = f != FAM_WBFM && k.d4_leadfree == 2 ? (uint32_t)d4_lead_shift(f) : 0u;   // (the one launch: full lead-ins unless asked)
            ff[f].halo = halo_of(f, ff[f].shift);
            ff[f].granule = f == FAM_WBFM ? k.env_stream_gran : k.env_d4_gran;
            ff[f].ns_per_sample = k.fam_ns[f];
        }
        if (k.env_shares_by_cost || !plan_fused_by_time(vlen, FAM_COUNT, ff, k.n_cus, share))
            plan_fused_shares(cost, FAM_COUNT, k.n_cus, share);
    } else if (shares_on && !plan_family_shares(cost, FAM_COUNT, k.n_cus, share)) {
        shares_on = false;
    }
    if (!shares_on) fused = false;
    out.shares_on = shares_on;
    out.fused = fused;

    float lane_load[4] = {0.f, 0.f, 0.f, 0.f};   // 0: the engine's stream, 1 to 3: the side streams
    for (int oi = 0; oi < FAM_COUNT; oi++) {
        const int f = out.order[oi];
        const FamilyShape &s = c.fam[f];
        FamilyPlan &p = out.fam[f];
        if (!s.n_list) continue;
        // More than one family as kernels of their own: up to three side streams beside the engine's (the runtime multiplexes
        // equal-priority streams onto a handful of hardware queues); families go to the least loaded lane, dearest first
        if (out.forked && !fused) {
            for (int l = 1; l < 4; l++)
                if (lane_load[l] < lane_load[p.lane]) p.lane = l;
            lane_load[p.lane] += cost[f];
        }
        p.wgs = k.env_stream_wgs ? k.env_stream_wgs : share[f];
        // workgroups a CU holds at once: WBFM 3 (LDS), the others 4 (registers)
        const TilePlan tp = f == FAM_WBFM ? plan_tiles(vlen, s.n_list, k.wbfm_chunk, k.wbfm_cold_halo, 3 * k.n_cus, k.env_plan_chunks)
                                          : plan_tiles(vlen, s.n_list, k.ch_chunk, (uint32_t)fir_halo(f), 4 * k.n_cus, k.env_plan_chunks);
        p.tile_len = tp.tile_len;
        p.tiles_per_ch = tp.tiles_per_ch;
        int want = 0;   // 0 choose, 1 stream, -1 tiles
        if (k.flags & IQD_F_WBFM_STREAM) want = 1;
        if (k.env_path) want = k.env_path;
        const uint64_t work = (uint64_t)vlen * s.n_list;
        const bool enough = want > 0 || shares_on ||
                            work >= (uint64_t)(shares_on ? p.wgs : k.n_cus) * ST_SEGS * stream_min_seg(k, f, vlen, 0);
        const bool whole_units = !(k.flags & IQD_F_WBFM_TILES) && vlen % 128 == 0;
        if (f == FAM_WBFM) {
            // WBFM: the streaming pipeline (iqd_stream.hip) when the launch can fill the chip with it and nothing it does
            // not handle is in play: a K so large that (int16)y can hit the "integer indefinite" value, rows that are not whole
            // 128-sample units.  Results are identical either way.
            const bool mixed_selectors = !(s.rot_count[0] == s.n_list || s.rot_count[1] == s.n_list || s.rot_count[2] == s.n_list);
            bool ok = k.stream_ok && whole_units && s.cast_bounded;
            if (fused && mixed_selectors) ok = false;   // (the one-launch arrangement holds the single-selector instantiations only)
            if (ok && want >= 0 && enough) {
                p.path = PLAN_STREAM;
                p.grouped = mixed_selectors;
                p.rings = rings_of(k, c, f, p.wgs, fused, p.grouped, 0u);
                for (;;) {
                    const uint32_t wg_segs = 64u * p.rings;
                    const uint32_t ids = fit_stream(s, vlen, p.wgs * wg_segs, k.env_stream_gran, p.grouped, 0u, p);
                    const uint32_t wgs_needed = (ids + wg_segs - 1) / wg_segs;
                    p.grid = wgs_needed < p.wgs ? wgs_needed : p.wgs;
                    p.rounds = (wgs_needed + p.grid - 1) / p.grid;
                    if (p.rounds == 1 || p.rings == (uint32_t)ST_RINGS || k.env_rings) break;
                    p.rings = (uint32_t)ST_RINGS;   // (workgroups of fewer rings are for launches of one round)
                }
                p.epochs = s.epochs_in_reach;
            }
        } else if (whole_units && (f == FAM_FM || vlen / 32 >= k.env_am_stream_min)) {
            // FM / AM / SSB: the streaming pipelines of iqd_stream2.hip under the same conditions (channels of different
            // rotation selectors are fine here: the list is sorted by selector and the groups are padded)
            const bool ok = f != FAM_FM || s.cast_bounded;
            if (ok && want >= 0 && enough) {
                p.path = PLAN_STREAM;
                p.grouped = true;
                // short lead-ins (d4_geom) where they pay: see PlanKnobs::d4_leadfree
                uint32_t shift = (uint32_t)d4_lead_shift(f);
                if (k.d4_leadfree == 0 || (fused && k.d4_leadfree != 2)) shift = 0;
                for (;;) {
                    p.lead_shift = shift;
                    p.halo = halo_of(f, shift);
                    p.rings = rings_of(k, c, f, p.wgs, fused, true, shift);
                    for (;;) {
                        const uint32_t wg_segs = 64u * p.rings;
                        // (these pipelines store 8 or 16 bytes per 128 samples: no wide stores to keep whole)
                        fit_stream(s, vlen, p.wgs * wg_segs, k.env_d4_gran, true, shift, p);
                        const uint32_t wgs_needed = (p.group_start[3] + wg_segs - 1) / wg_segs;
                        p.grid = wgs_needed < p.wgs ? wgs_needed : p.wgs;
                        p.rounds = (wgs_needed + p.grid - 1) / p.grid;
                        if (p.rounds == 1 || p.rings == (uint32_t)ST_RINGS || k.env_rings) break;
                        p.rings = (uint32_t)ST_RINGS;   // (workgroups of fewer rings are for launches of one round)
                    }
                    if (!shift || k.d4_leadfree >= 0 || p.tiles_per_ch >= D4_LEADFREE_MIN_TILES) break;
                    shift = 0;   // (the default rule: a channel of only a handful of segments keeps its full lead-ins)
                }
            }
        }
        if (fused) {
            if (p.path != PLAN_STREAM) return false;
            p.wg_first = out.mix_wgs;
            out.mix_wgs += p.grid;
        }
    }
    return true;
}

}  // namespace

void plan_call(const PlanKnobs &k, const CallShape &c, CallPlan &out)
{
    if (!plan_once(k, c, true, out)) plan_once(k, c, false, out);
}

}  // namespace iqd
