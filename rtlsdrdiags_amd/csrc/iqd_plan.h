// The plan of one accept call: which kernels run for which demodulator family, on how many workgroups, cut into which
// segments - decided ONCE, on the host, from plain data, before anything is queued (plan, then execute).
// iqd_engine.cpp describes the call (describe_call), plan_call() decides, the queue_* functions only carry the plan out.
// No HIP in here: tests/emu compiles this file for the CPU tier (tests/test_host_planning.py enumerates plans).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "iqd_device.h"
#include "iqd_host.h"

namespace iqd {

// A chain launch takes its streaming kernel when it brings this many samples per segment of the persistent workgroups
// (n_cus x 192 segments).  Round 4 re-measured the crossovers against the tile kernels on 256 CUs (tools/minseg_probe.sh,
// ms per step streaming / tiles): FM 512 x 2^16 0.085 / 0.069, 1024 x 2^16 0.091 / 0.118, 4096 x 2^13 0.078 / 0.084; WBFM
// 1 x 2^25 0.122 / 0.131, 512 x 2^16 0.129 / 0.142, 1 x 2^24 0.115 / 0.089; AM 1024 x 2^16 0.077 / 0.084, 512 x 2^16 0.075 / 0.055,
// and rows of one or two blocks (where the tile path's DC pass has a lane per channel and nothing to hide its latency
// behind) 1024 x 2^14 0.074 / 0.079, 2048 x 2^13 0.063 / 0.063.  A call with SEVERAL families takes the one-launch
// arrangement (iqd_stream_mixed.hip) from the smallest sizes probed: 512 x 2^14 0.102 / 0.141, 4096 x 2^14 0.130 / 0.230,
// 1400 x 2^16 0.130 / 0.268 - round 3's rule (1024 per segment for every family's share) dated from the kernels-on-streams
// arrangement and kept such calls on the tile kernels.  IQD_STREAM_MIN_SEG overrides all of them (measurement runs).
// (second probe, around the thresholds: FM 640 x 2^16 0.085 / 0.088, 768 x 2^16 0.087 / 0.095; WBFM 384 x 2^16 0.120 / 0.099; AM / USB
// 768 x 2^16 0.075 / 0.074 and 0.087 / 0.082, 896 x 2^16 0.075 / 0.083 and 0.086 / 0.093, 1024 x 2^14 0.074 / 0.079 and 0.084 / 0.082;
// several families: 128 x 2^14 0.099 / 0.125, 16 x 2^16 0.100 / 0.120, 1024 x 2^12 0.098 / 0.101, 512 x 2^12 0.096 / 0.076)
// Round 5 (workgroups of fewer rings for small launches, rings_of(); tools/r5/r5_fourth.sh, profiles/r5_threshold_probe.txt, default /
// tiles / stream): USB 1024 x 2^14 0.0828 / 0.0808 / 0.0693 and 768 x 2^16 0.0814 / 0.0823 / 0.0776 - the SSB thresholds came down
// (1100 -> 1000, short rows 450 -> 330); every other crossover stayed where it was (FM 512 x 2^16 0.0697 tiles / 0.0848 stream,
// 768 x 2^16 0.0945 / 0.0859; AM 512 x 2^16 0.0545 / 0.0662, 1024 x 2^14 0.0784 / 0.0626; WBFM 256 x 2^16 0.0844 / 0.1015, 512 x 2^16 0.1469 / 0.1164).
constexpr uint64_t STREAM_MIN_SEG_WBFM = 600, STREAM_MIN_SEG_FM = 900, STREAM_MIN_SEG_AM = 1000, STREAM_MIN_SEG_SSB = 1000,
                   STREAM_MIN_SEG_AM_SHORT = 320, STREAM_MIN_SEG_SSB_SHORT = 330,   // rows of up to 2^14 samples
                   STREAM_MIN_SEG_MIXED = 16, STREAM_MIN_SEG_MIXED_SHORT = 96,      // one launch for all families; rows below 2^13 samples
                   STREAM_MIN_SEG_FORKED = 1024;                                    // several families as kernels on streams
// AM / SSB rows at least this long (PCM samples) may take their streaming pipeline; the DC pass behind it is then the
// one-wave pass whatever the row length (round 4: the rule used to be > 512, which kept the reference's own operating
// point - one 64 ms block per channel per call, 512 PCM samples - on the tile kernels at 0.14 of the HBM peak).  IQD_AM_STREAM_MIN.
constexpr uint32_t AM_STREAM_MIN_PCM = 128;
constexpr uint32_t D4_LEADFREE_MIN_TILES = 8;

// Engine-wide settings: iqd_config::flags and the IQD_* measurement knobs, read once by iqd_create.
struct PlanKnobs {
    uint32_t flags = 0;                  // IQD_F_*
    uint32_t n_cus = 256;
    bool stream_ok = true;               // the half table's symmetry holds on this host's libm (iqd_create)
    int env_path = 0;                    // IQD_WBFM_PATH: +1 stream, -1 tiles, 0 choose
    uint64_t env_stream_min_seg = 0;     // IQD_STREAM_MIN_SEG (0: the measured per-family thresholds)
    uint32_t env_am_stream_min = AM_STREAM_MIN_PCM;   // IQD_AM_STREAM_MIN (measurement runs: 513 = the rule of rounds 2-3)
    uint32_t env_d4_gran = 128;          // IQD_D4_GRAN: segment-length granule of the FM / AM / SSB pipelines (measurement runs)
    // FM / AM / SSB streaming segments with 128-sample lead-ins (round 6, iqd_stream.h: d4_geom).  IQD_D4_LEADFREE: 0 = every segment
    // with its family's full lead-in (rounds 2-5), 1 = short lead-ins wherever the family streams as a kernel of its own, 2 = in
    // the one launch for several families too; default (no variable): where a channel is cut into at least D4_LEADFREE_MIN_TILES
    // segments - a channel's first segment and the lane-0 ones pay their full lead-in out of the segments' length, which costs
    // more than it saves when a channel has only a handful of them (8192 channels x 2^16: six; profiles/r6_leadfree_ab.txt)
    int d4_leadfree = -1;
    bool env_full_grid = false;          // IQD_FULL_GRID
    bool env_mixed_forked = false;       // IQD_MIXED=forked: several families as kernels of their own side by side (A/B runs)
    bool env_shares_by_cost = false;     // IQD_SHARES=cost: round 3's proportional shares
    uint32_t env_stream_wgs = 0, env_plan_chunks = 0, env_stream_gran = 0;   // IQD_STREAM_WGS, IQD_PLAN_CHUNKS, IQD_STREAM_GRAN
    uint32_t env_rings = 0;              // IQD_RINGS=1|2|3: rings per workgroup of every streaming launch (0: choose; measurement runs)
    // ns per sample of a segment (lead-in included) of one workgroup's 192 segments in lock step, inside the one launch that
    // holds all four pipelines: WBFM 224 us for 3072 + 768, FM 225 for 5120 + 768, AM 214 for 9472 + 384, SSB 226 for 9472 + 1280
    // (tools/mixed_probe.py, 4096 channels x 2^16).  IQD_FAMILY_NS=am,fm,wbfm,ssb
    float fam_ns[FAM_COUNT] = {21.7f, 38.2f, 58.3f, 21.0f};
    // relative cost per channel-sample of the streaming pipelines: AM, FM, WBFM, SSB (the families' workgroups side by side: 214 /
    // 228 / 227 / 259 us on 32 / 56 / 96 / 56 CUs for 819 / 819 / 820 / 1638 channels x 2^16, profiles/r3_mixed_4096_kernel_stats.csv)
    float fam_weight[FAM_COUNT] = {3.4f, 6.3f, 10.8f, 3.6f};
    // geometry of the tile kernels and the DC pass (iqd_device.h, iqd_chains.h; handed in so that this file needs no kernel header)
    uint32_t wbfm_chunk = 0, wbfm_cold_halo = 0, ch_chunk = 0, dc_tile = 0;
};

// One demodulator family's channels in the call.
struct FamilyShape {
    uint32_t n_list = 0;                 // channels (0: the family is not in the call)
    uint32_t rot_count[3] = {0, 0, 0};   // ... per rotation selector +Fs/4, none, -Fs/4 (the list is sorted like that)
    int32_t rot_first = 0;               // selector of the list's first channel
    bool cast_bounded = true;            // WBFM / FM: every channel's |K| keeps (int16)y clear of the "integer indefinite" value
    bool epochs_in_reach = false;        // WBFM: some channel has a gain change that a lead-in can still reach (GainEpochList)
};
struct CallShape {
    uint32_t vlen = 0;                   // samples per channel
    uint32_t pcm_per_ch = 0;             // bytes_per_ch / 64
    bool gated = false;                  // a squelch that can close: the chains walk each channel's open blocks
    FamilyShape fam[FAM_COUNT];
};

enum { PLAN_TILES = 0, PLAN_STREAM = 1 };
struct FamilyPlan {
    bool present = false;
    int path = PLAN_TILES;               // tile kernel, or the family's streaming pipeline
    int lane = 0;                        // 0: the engine's stream; 1..3: a side stream (several families as kernels of their own)
    uint32_t wgs = 0;                    // the family's share of the CUs / of the one launch's workgroups
    uint32_t tile_len = 0, tiles_per_ch = 0;
    // streaming pipelines
    uint32_t halo = 0, lead_shift = 0;   // FM / AM / SSB: lead-in samples every segment runs; what a cold segment skips of its own length (iqd_stream.h: d4_geom)
    bool grouped = false;                // segment ids grouped by rotation selector, each group padded to 16 (WBFM: only if mixed)
    uint32_t group_start[4] = {0, 0, 0, 0}, group_li0[3] = {0, 0, 0}, group_nseg[3] = {0, 0, 0};
    uint32_t grid = 0, rounds = 0;       // workgroups launched, rounds each runs
    uint32_t rings = 3;                  // rings of 64 segments per workgroup (StreamArgs::rings)
    uint32_t wg_first = 0;               // one launch for all families: the family's first workgroup
    bool epochs = false;                 // WBFM: the instantiation with the piecewise-gain lookup
};
struct CallPlan {
    int n_fams = 0;
    bool forked = false;                 // more than one family
    bool shares_on = false;              // the families' streaming kernels side by side, each on a share of the CUs
    bool fused = false;                  // ... as ranges of ONE launch's workgroups (iqd_stream_mixed.hip)
    int order[FAM_COUNT] = {0, 1, 2, 3}; // families by estimated cost, dearest first (the order they are queued in)
    FamilyPlan fam[FAM_COUNT];
    uint32_t mix_wgs = 0;                // fused: the launch's grid
};

uint64_t stream_min_seg(const PlanKnobs &k, int f, uint64_t vlen, int several_families);
void plan_call(const PlanKnobs &k, const CallShape &c, CallPlan &out);

}  // namespace iqd
