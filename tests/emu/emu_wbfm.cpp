// tests/emu/emu_wbfm.cpp — TEST INFRASTRUCTURE ONLY.
//
// Compiles the WBFM chain's phase functions (rtlsdrdiags_amd/csrc/iqd_wbfm.h) for the host
// with IQD_HOST_EMU and steps them with a loop over thread ids in place of the SIMT machine,
// so the tiling / hand-off / history logic of the GPU kernel can be checked against the
// oracle in the CPU-only test tier.  It mirrors what iqd_engine.cpp does for one channel
// with the squelch open.  Nothing in the product links or loads this file.
#define IQD_HOST_EMU 1
#include <stdint.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "iqd_host.h"
#include "iqd_plan.h"
#include "iqd_stream.h"
#include "iqd_taps.h"
#include "iqd_chains.h"
#include "iqd_wbfm.h"

namespace {

struct HostExec {
    template <class F> void all(F f) { for (int t = 0; t < iqd::WB_THREADS; t++) f(t); }
    bool in_wave0() const { return true; }
    template <class F> void wave0(F f) { for (int l = 0; l < 64; l++) f(l); }
    template <class F> bool wave0_all(F f)
    {
        bool ok = true;
        for (int l = 0; l < 64; l++) ok = f(l) && ok;
        return ok;
    }
    void sync() const {}
    template <class F> void others(F f) { for (int t = 64; t < iqd::WB_THREADS; t++) f(t); }
    template <class F> void all_nosync(F f) { for (int t = 0; t < iqd::WB_THREADS; t++) f(t); }
    void others_sync(uint32_t *) {}   // phases run one after the other here
    // wave shift by one lane: lanes run in ascending order here, so the lane below has already published
    uint32_t pub[18][iqd::WB_THREADS];
    template <int SLOT> uint32_t shr1(int tid, uint32_t v)
    {
        pub[SLOT][tid] = v;
        return (tid & 63) ? pub[SLOT][tid - 1] : 0xdeadbeefu;
    }
    template <class T> struct Local {
        std::vector<T> v = std::vector<T>(iqd::WB_THREADS);
        T &at(int tid) { return v[tid]; }
    };
    void stamp(int) const {}
    void critical(bool) const {}
};

}  // namespace

static int g_serial_phases = 0;

extern "C" {

void emu_wbfm_driver(int serial_phases) { g_serial_phases = serial_phases; }

// One accept call for one WBFM channel (squelch open).  tail/carry are in-out state.
// guess_bias perturbs the de-emphasis state guess (to force the segment repair path).
// Returns the number of tile hand-off mismatches; seg_repairs counts repair rounds.
int emu_wbfm_accept(const uint8_t *iq, uint32_t n_samples, uint32_t tile_len, uint32_t block_samples,
                    int rotation, float gain, uint8_t *tail, iqd::WbfmCarry *carry,
                    int16_t *pcm, uint32_t *mag_sums, uint32_t *seg_repairs, float guess_scale)
{
    using namespace iqd;
    static Consts c;
    static std::vector<float> lut;
    static bool ready = false;
    if (!ready) { build_consts(c); build_atan2_lut(lut); ready = true; }
    Consts cc = c;
    cc.deemph_c128 *= guess_scale;  // 1.0 = the product's guess

    ChanParams p;
    default_params(p);
    p.gain[FAM_WBFM] = gain;
    p.rotation = rotation;
    derive_params(p);

    const uint32_t ntiles = (n_samples + tile_len - 1) / tile_len;
    std::vector<WbfmRecord> recs(ntiles);
    static WbfmLds lds;
    uint32_t repairs = 0;
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        WbfmTile t;
        t.iq_ch = iq;
        t.tail = tail;
        t.blk_list = nullptr;
        t.block_samples = block_samples;
        t.block_magic = block_magic(block_samples);
        t.v0 = (int64_t)tile * tile_len;
        t.tlen = (int32_t)((n_samples - t.v0) < tile_len ? (n_samples - t.v0) : tile_len);
        if (rotation == 0) { t.sel_i = 0x06040200u; t.sel_q = 0x07050301u; t.neg_i = 0; t.neg_q = 0; }
        else {
            t.sel_i = 0x07040300u; t.sel_q = 0x06050201u;
            t.neg_i = rotation > 0 ? 0x00ffff00u : 0xffff0000u;
            t.neg_q = rotation > 0 ? 0xffff0000u : 0x00ffff00u;
        }
        t.k = p.wbfm_k;
        t.epochs = nullptr;
        t.k_min = p.wbfm_k;
        t.bounded = (fabsf(p.wbfm_k) * 3.1730f < 2147483648.0f) ? 1u : 0u;
        t.lut = lut.data();
        t.pcm_row = pcm;
        t.mag_row = mag_sums;
        WbfmStart start;
        if (tile == 0) { start.y = carry->y; start.u = carry->u; start.back = carry->back; start.cold = 0; }
        else { start.y = 0; start.u = 0; start.back = 0; start.cold = 1; }
        HostExec ex;
        memset(&lds, 0xcd, sizeof(lds));  // poison: the kernel must initialise what it reads
        if (g_serial_phases) wbfm_tile<false, true>(ex, t, cc, lds, start, &recs[tile]);
        else wbfm_tile_pipe<false, true>(ex, t, cc, lds, start, &recs[tile]);
        repairs += lds.repair_count;
    }
    int mismatches = 0;
    for (uint32_t tile = 1; tile < ntiles; tile++)
        if (f2u(recs[tile].y_in) != f2u(recs[tile - 1].y_out)) mismatches++;
    const WbfmRecord &r = recs[ntiles - 1];
    carry->y = r.y_out; carry->u = r.u_out; carry->back = r.back_out;
    carry->y_end = r.y_end; carry->u_end = r.u_end;
    // tail update (tail_update_kernel)
    std::vector<uint8_t> nt(TAIL_BYTES);
    for (int i = 0; i < TAIL; i++) {
        const int64_t v = (int64_t)n_samples - TAIL + i;
        const uint8_t *src = v < 0 ? tail + TAIL_BYTES + 2 * v : iq + 2 * v;
        nt[2 * i] = src[0];
        nt[2 * i + 1] = src[1];
    }
    memcpy(tail, nt.data(), TAIL_BYTES);
    if (seg_repairs) *seg_repairs = repairs;
    return mismatches;
}

void emu_wbfm_reset(uint8_t *tail, iqd::WbfmCarry *carry)
{
    memset(tail, 0x80, iqd::TAIL_BYTES);
    carry->y = carry->y_end;
    carry->u = carry->u_end;
    carry->back = 0;
}

uint32_t emu_lds_bytes(void) { return (uint32_t)sizeof(iqd::WbfmLds); }

// One accept call for one FM (family 1), AM (0) or SSB (3) channel with the squelch open.
// tail and dc are in-out state; base8k is caller scratch of n_samples/32 ints.
void emu_chain_accept(int family, int lsb, const uint8_t *iq, uint32_t n_samples, uint32_t tile_len,
                      uint32_t block_samples, int rotation, float gain, uint8_t *tail, iqd::DcCarry *dc,
                      int16_t *pcm, uint32_t *mag_sums, int32_t *base8k)
{
    using namespace iqd;
    static Consts c;
    static std::vector<float> fm_lut;
    static bool ready = false;
    if (!ready) { build_consts(c); build_fm_lut(fm_lut); ready = true; }
    ChanParams p;
    default_params(p);
    p.gain[family] = gain;
    p.rotation = rotation;
    derive_params(p);
    const uint32_t ntiles = (n_samples + tile_len - 1) / tile_len;
    static FmLds flds;
    static AmLds alds;
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        Tile t;
        t.iq_ch = iq;
        t.tail = tail;
        t.blk_list = nullptr;
        t.block_samples = block_samples;
        t.block_magic = block_magic(block_samples);
        t.v0 = (int64_t)tile * tile_len;
        t.tlen = (int32_t)((n_samples - t.v0) < tile_len ? (n_samples - t.v0) : tile_len);
        if (rotation == 0) { t.sel_i = 0x06040200u; t.sel_q = 0x07050301u; t.neg_i = 0; t.neg_q = 0; }
        else {
            t.sel_i = 0x07040300u; t.sel_q = 0x06050201u;
            t.neg_i = rotation > 0 ? 0x00ffff00u : 0xffff0000u;
            t.neg_q = rotation > 0 ? 0xffff0000u : 0x00ffff00u;
        }
        t.k = p.fm_k;
        t.epochs = nullptr;
        t.k_min = p.fm_k;
        t.bounded = (fabsf(p.fm_k) * 6.35f < 2147483648.0f) ? 1u : 0u;
        t.lut = nullptr;
        t.pcm_row = pcm;
        t.mag_row = mag_sums;
        HostExec ex;
        if (family == FAM_FM) {
            memset(&flds, 0xcd, sizeof(flds));
            fm_tile<false, true>(ex, t, c, flds, fm_lut.data());
        } else {
            memset(&alds, 0xcd, sizeof(alds));
            am_tile<false, true>(ex, t, c, alds, family == FAM_SSB, lsb, base8k, 1);
        }
    }
    if (family != FAM_FM) {
        if (n_samples / 32 <= 256) {
            dc_block_run(base8k, (int)(n_samples / 32), p.gain[family], c.dc_a1, *dc, pcm);
        } else {   // the segmented wave-per-channel variant
            static DcLds dlds;
            memset(&dlds, 0xcd, sizeof(dlds));
            HostExec ex;
            dc_block_wave(ex, c, dlds, base8k, (int)(n_samples / 32), p.gain[family], *dc, pcm);
        }
    }
    std::vector<uint8_t> nt(TAIL_BYTES);
    for (int i = 0; i < TAIL; i++) {
        const int64_t v = (int64_t)n_samples - TAIL + i;
        const uint8_t *src = v < 0 ? tail + TAIL_BYTES + 2 * v : iq + 2 * v;
        nt[2 * i] = src[0];
        nt[2 * i + 1] = src[1];
    }
    // (what the closing launch does not keep, tail_update_body: poisoned, so that a tile reaching back further than tail_keep shows)
    const int keep = family == FAM_FM ? tail_keep(FAM_FM) : family == FAM_AM ? tail_keep(FAM_AM) : tail_keep(FAM_SSB);
    for (int i = 0; i < 2 * (TAIL - keep); i++) nt[i] = (uint8_t)(0x5a + 37 * i);
    memcpy(tail, nt.data(), TAIL_BYTES);
}

// The device's AGC step (iqd_chains.h: agc_run) on the host: n magnitudes through one channel's AGC.
// cfg: {enabled, type, operating_point, deadband, alpha, blanking_limit}; state in/out: {rx_gain, if_gain,
// filtered, blank_ctr, adjusted}; gains_out[i] = the IF gain after magnitude i.
void emu_agc_run(uint32_t type, int32_t operating_point, int32_t deadband, float alpha, uint32_t blanking_limit,
                 uint32_t *rx_gain, uint32_t *if_gain, float *filtered, uint32_t *blank_ctr, uint32_t *adjusted,
                 const uint32_t *magnitudes, uint32_t n, uint32_t *gains_out)
{
    iqd::Consts c;
    iqd::build_consts(c);
    iqd::AgcConfig cfg{};
    cfg.enabled = 1; cfg.type = type; cfg.operating_point = operating_point; cfg.deadband = deadband;
    cfg.alpha = alpha; cfg.blanking_limit = blanking_limit;
    iqd::AgcState st{};
    st.rx_gain = *rx_gain; st.if_gain = *if_gain; st.filtered = *filtered; st.blank_ctr = *blank_ctr; st.adjusted = *adjusted;
    uint32_t gain = st.rx_gain;
    for (uint32_t i = 0; i < n; i++) {
        gain = iqd::agc_run(c, cfg, st, magnitudes[i], gain);
        gains_out[i] = gain;
    }
    *rx_gain = gain; *if_gain = st.if_gain; *filtered = st.filtered; *blank_ctr = st.blank_ctr; *adjusted = st.adjusted;
}

// host planning logic of the engine, for the CPU tier
void emu_plan_fused_shares(const float *cost, int n, uint32_t n_wgs, uint32_t *share)
{
    iqd::plan_fused_shares(cost, n, n_wgs, share);
}

int emu_plan_family_shares(const float *cost, int n, uint32_t n_cus, uint32_t *share)
{
    return iqd::plan_family_shares(cost, n, n_cus, share) ? 1 : 0;
}

// fam: per family {rot_count[3], halo, granule} as six uint32 + ns_per_sample as a float array
int emu_plan_fused_by_time(uint32_t vlen, int n, const uint32_t *fam6, const float *ns, uint32_t n_wgs, uint32_t *share)
{
    iqd::FusedFamily ff[8];
    for (int f = 0; f < n && f < 8; f++) {
        for (int r = 0; r < 3; r++) ff[f].rot_count[r] = fam6[6 * f + r];
        ff[f].halo = fam6[6 * f + 3];
        ff[f].granule = fam6[6 * f + 4];
        ff[f].ns_per_sample = ns[f];
    }
    return iqd::plan_fused_by_time(vlen, n, ff, n_wgs, share) ? 1 : 0;
}

void emu_plan_stream(uint32_t vlen, uint32_t n_channels, uint32_t streams, uint32_t *tile_len, uint32_t *tiles_per_ch)
{
    const iqd::TilePlan p = iqd::plan_stream(vlen, n_channels, streams);
    *tile_len = p.tile_len;
    *tiles_per_ch = p.tiles_per_ch;
}

// plan_call (iqd_plan.cpp) on plain arrays.  knobs: {flags, n_cus, stream_ok, env_path (+1 / 0 / -1 as 1 / 0 / 2), env_stream_min_seg,
// env_am_stream_min (0 = default), env_mixed_forked, env_shares_by_cost, env_stream_wgs, env_full_grid, env_rings}; fam: per family
// {n_list, rot_count[3], cast_bounded, epochs_in_reach}; out: {n_fams, forked, shares_on, fused, mix_wgs, order[4]} then per family
// {present, path, lane, wgs, tile_len, tiles_per_ch, grouped, group_start[4], group_nseg[3], grid, rounds, wg_first, epochs, rings, halo, lead_shift}
// (21 words).  knobs[11]: 0 = the default rule for short lead-ins, 1 + v = IQD_D4_LEADFREE=v.
void emu_plan_call(const uint32_t *knobs, uint32_t vlen, uint32_t pcm_per_ch, uint32_t gated, const uint32_t *fam, uint32_t *out)
{
    iqd::PlanKnobs k;
    k.flags = knobs[0]; k.n_cus = knobs[1]; k.stream_ok = knobs[2] != 0;
    k.env_path = knobs[3] == 1 ? 1 : knobs[3] == 2 ? -1 : 0;
    k.env_stream_min_seg = knobs[4];
    if (knobs[5]) k.env_am_stream_min = knobs[5];
    k.env_mixed_forked = knobs[6] != 0; k.env_shares_by_cost = knobs[7] != 0; k.env_stream_wgs = knobs[8]; k.env_full_grid = knobs[9] != 0;
    k.env_rings = knobs[10];
    k.d4_leadfree = knobs[11] == 0 ? -1 : (int)knobs[11] - 1;   // 0: the default rule; 1 / 2 / 3: IQD_D4_LEADFREE = 0 / 1 / 2
    k.wbfm_chunk = iqd::WBFM_CHUNK; k.wbfm_cold_halo = iqd::COLD_HALO; k.ch_chunk = iqd::CH_CHUNK; k.dc_tile = iqd::DC_TILE;
    iqd::CallShape c;
    c.vlen = vlen; c.pcm_per_ch = pcm_per_ch; c.gated = gated != 0;
    for (int f = 0; f < iqd::FAM_COUNT; f++) {
        c.fam[f].n_list = fam[6 * f];
        for (int r = 0; r < 3; r++) c.fam[f].rot_count[r] = fam[6 * f + 1 + r];
        c.fam[f].cast_bounded = fam[6 * f + 4] != 0;
        c.fam[f].epochs_in_reach = fam[6 * f + 5] != 0;
    }
    iqd::CallPlan p;
    iqd::plan_call(k, c, p);
    uint32_t *o = out;
    *o++ = (uint32_t)p.n_fams; *o++ = p.forked; *o++ = p.shares_on; *o++ = p.fused; *o++ = p.mix_wgs;
    for (int f = 0; f < iqd::FAM_COUNT; f++) *o++ = (uint32_t)p.order[f];
    for (int f = 0; f < iqd::FAM_COUNT; f++) {
        const iqd::FamilyPlan &q = p.fam[f];
        *o++ = q.present; *o++ = (uint32_t)q.path; *o++ = (uint32_t)q.lane; *o++ = q.wgs; *o++ = q.tile_len; *o++ = q.tiles_per_ch; *o++ = q.grouped;
        for (int r = 0; r < 4; r++) *o++ = q.group_start[r];
        for (int r = 0; r < 3; r++) *o++ = q.group_nseg[r];
        *o++ = q.grid; *o++ = q.rounds; *o++ = q.wg_first; *o++ = q.epochs; *o++ = q.rings; *o++ = q.halo; *o++ = q.lead_shift;
    }
}

// ---- the AM / SSB DC-removal pass of one channel row, one wave, IN PLACE over int16 detector values (round 6: what runs behind the
// streaming pipelines, iqd_kernels.hip: dc_channel_wave) or from an int32 stream into a PCM row.  tests/test_emu_chains.py
void emu_dc_row(int in_place, const int32_t *x32, int16_t *row, int n, float gain, iqd::DcCarry *st)
{
    using namespace iqd;
    static Consts c;
    static bool ready = false;
    if (!ready) { build_consts(c); ready = true; }
    static DcLds lds;
    HostExec ex;
    if (in_place) dc_block_wave(ex, c, lds, (const int16_t *)row, n, gain, *st, row);
    else dc_block_wave(ex, c, lds, x32, n, gain, *st, row);
}
void emu_dc_serial(const int32_t *x32, int16_t *row, int n, float gain, iqd::DcCarry *st)
{
    static iqd::Consts c;
    static bool ready = false;
    if (!ready) { iqd::build_consts(c); ready = true; }
    iqd::dc_block_run(x32, n, gain, c.dc_a1, *st, row);
}

// ---- FM / AM / SSB segments with short lead-ins: the geometry the kernels and the plan share (iqd_stream.h: d4_geom) ----------
void emu_d4_geom(uint32_t sid, uint32_t tile, uint32_t tile_len, uint32_t shift, int64_t *v0, uint32_t *skip, uint32_t *cold)
{
    const iqd::D4Geom g = iqd::d4_geom(sid, tile, tile_len, shift);
    *v0 = g.v0; *skip = g.skip; *cold = g.cold;
}
void emu_plan_stream2(uint32_t vlen, uint32_t n_channels, uint32_t streams, uint32_t granule, uint32_t shift, uint32_t *tile_len, uint32_t *tiles_per_ch)
{
    const iqd::TilePlan p = iqd::plan_stream(vlen, n_channels, streams, granule, shift);
    *tile_len = p.tile_len;
    *tiles_per_ch = p.tiles_per_ch;
}
uint32_t emu_d4_const(int which)
{
    switch (which) {
    case 3: return iqd::D4_HALO_SHORT;
    case 4: return (uint32_t)iqd::d4_lead_shift(iqd::FAM_AM);
    case 5: return (uint32_t)iqd::d4_lead_shift(iqd::FAM_FM);
    case 6: return (uint32_t)iqd::d4_lead_shift(iqd::FAM_SSB);
    case 7: return iqd::TAIL;
    case 8: return (uint32_t)iqd::d4_replay_outputs(iqd::FAM_AM);
    case 9: return (uint32_t)iqd::d4_replay_outputs(iqd::FAM_FM);
    case 10: return (uint32_t)iqd::d4_replay_outputs(iqd::FAM_SSB);
    case 11: return (uint32_t)iqd::D4_REPLAY_PAIRS_AMSSB;
    case 12: return (uint32_t)iqd::D4_RAILS_FROM_PIECE;
    case 13: return (uint32_t)iqd::tail_keep(iqd::FAM_AM);
    case 14: return (uint32_t)iqd::tail_keep(iqd::FAM_FM);
    case 15: return (uint32_t)iqd::tail_keep(iqd::FAM_SSB);
    }
    return 0;
}

// ---- the WBFM restart state's journey from call to call (tests/test_emu_restart_model.py) ----------------------------------
// st_rec_plan / wbfm_pick_carry are the kernel's own (iqd_wbfm.h): where a streamed segment takes its record, what the commit picks.
void emu_st_rec_plan(uint32_t valid, uint32_t tile, int32_t v0, int32_t tlen, int32_t vlen, int32_t carried_back, int32_t *out4)
{
    const iqd::StRecPlan r = iqd::st_rec_plan(valid, tile, v0, tlen, vlen, carried_back);
    out4[0] = r.rec_pos; out4[1] = r.back_out; out4[2] = r.park_pos; out4[3] = (int32_t)r.keeps_restart;
}

// records: 8 words each {y_in, y_out, u_out, back_out, y_end, u_end, pad[0], pad[1]} (iqd::WbfmRecord)
void emu_wbfm_pick_carry(const uint32_t *last8, const uint32_t *before8, uint32_t ntiles, uint32_t vlen, uint32_t tile_len,
                         uint32_t verify_at_end, iqd::WbfmCarry *out)
{
    static_assert(sizeof(iqd::WbfmRecord) == 32, "record layout");
    iqd::WbfmRecord a, b;
    memcpy(&a, last8, sizeof(a));
    memcpy(&b, before8, sizeof(b));
    *out = iqd::wbfm_pick_carry(a, b, ntiles, vlen, tile_len, verify_at_end);
}

// {FORCED_BACK, ST_HALO, ST_MIN_TILE, WBFM_REC_STREAMED, TAIL, WBFM_CHUNK}; a1, b0 of the de-emphasis filter; K for a demodulator gain
void emu_restart_consts(uint32_t *out6, float *a1_b0, float gain, float *k_out)
{
    out6[0] = iqd::FORCED_BACK; out6[1] = iqd::ST_HALO; out6[2] = iqd::ST_MIN_TILE; out6[3] = iqd::WBFM_REC_STREAMED;
    out6[4] = iqd::TAIL; out6[5] = iqd::WBFM_CHUNK;
    iqd::Consts c;
    iqd::build_consts(c);
    a1_b0[0] = c.deemph_a1; a1_b0[1] = c.deemph_b0;
    iqd::ChanParams p;
    iqd::default_params(p);
    p.gain[iqd::FAM_WBFM] = gain;
    iqd::derive_params(p);
    *k_out = p.wbfm_k;
}

// The decimator taps the streaming kernels carry as instruction literals (iqd_taps.h: STREAM_TAPS, compile time) beside the ones
// the host builds for the kernel arguments (build_consts + build_stream_taps): 30 words each, d1p2 | p12p | a40p.
void emu_stream_taps(uint32_t *literal30, uint32_t *host30)
{
    for (int q = 0; q < 4; q++) literal30[q] = iqd::taps::STREAM_TAPS.d1p2[q];
    for (int q = 0; q < 6; q++) literal30[4 + q] = iqd::taps::STREAM_TAPS.p12p[q];
    for (int q = 0; q < 20; q++) literal30[10 + q] = iqd::taps::STREAM_TAPS.a40p[q];
    iqd::Consts c;
    iqd::build_consts(c);
    iqd::StreamArgs sa{};
    iqd::build_stream_taps(c.wbfm_d1, c.post12, c.audio40, sa);
    for (int q = 0; q < 4; q++) host30[q] = sa.d1p2[q];
    for (int q = 0; q < 6; q++) host30[4 + q] = sa.p12p[q];
    for (int q = 0; q < 20; q++) host30[10 + q] = sa.a40p[q];
}

uint32_t emu_plan_const(int which)   // geometry constants the plan tests need
{
    switch (which) {
    case 0: return iqd::ST_SEGS;
    case 1: return iqd::ST_MIN_TILE;
    case 2: return iqd::ST_HALO;
    case 3: return iqd::DC_TILE;
    default: return 0;
    }
}

}  // extern "C"
