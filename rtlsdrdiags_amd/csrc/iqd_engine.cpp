// libiqdemod.so — C-ABI (include/iqdemod.h) over the gfx950 kernels.
//
// Host orchestration only: parameter mirrors, per-call bucketing of channels by demodulator
// family, launch planning, the exact-state verification of the WBFM tile hand-offs and the
// copies of the host-pointer entry point.  There is no CPU data path in this library: every
// sample is demodulated by the HIP kernels in iqd_kernels.hip, and creation fails when no HIP
// device is usable.
#include "iqdemod.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "iqd_host.h"
#include "iqd_kernels.h"
#include "iqd_stream.h"
#include "iqd_stream_mixed.h"
#include "iqd_taps.h"
#include "iqd_wbfm.h"
#include "iqd_chains.h"

using namespace iqd;

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
static const uint64_t STREAM_MIN_SEG_WBFM = 600, STREAM_MIN_SEG_FM = 900, STREAM_MIN_SEG_AM = 1000, STREAM_MIN_SEG_SSB = 1100,
                      STREAM_MIN_SEG_AM_SHORT = 320, STREAM_MIN_SEG_SSB_SHORT = 450,   // rows of up to 2^14 samples
                      STREAM_MIN_SEG_MIXED = 16, STREAM_MIN_SEG_MIXED_SHORT = 96,      // one launch for all families; rows below 2^13 samples
                      STREAM_MIN_SEG_FORKED = 1024;                                    // several families as kernels on streams
// AM / SSB rows at least this long (PCM samples) may take their streaming pipeline; the DC pass behind it is then the
// one-wave pass whatever the row length (round 4: the rule used to be > 512, which kept the reference's own operating
// point - one 64 ms block per channel per call, 512 PCM samples - on the tile kernels at 0.14 of the HBM peak).  IQD_AM_STREAM_MIN.



namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return (T *)p; }
};

}  // namespace

// AM / SSB rows at least this long (PCM samples) may take their streaming pipeline (see STREAM_MIN_PER_SEGMENT)
static const uint32_t AM_STREAM_MIN_PCM = 128;

struct iqd_engine {
    std::mutex mu;             // parameter mirror + dirty flags (setters vs. accept)
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n_ch = 0, block_bytes = 0, block_samples = 0, flags = 0;
    Consts consts;

    std::vector<ChanParams> h_params;
    bool params_dirty = true;
    bool lists_dirty = true;
    uint32_t list_first = 0, list_n = 0;
    std::vector<uint32_t> h_lists[FAM_COUNT + 1];  // per family; [FAM_COUNT] = mode None
    uint32_t rot_count[FAM_COUNT][3] = {};         // channels of each family per rotation group (+Fs/4, none, -Fs/4)
    uint32_t n_cus = 256;
    // measurement knobs, read from the environment ONCE at creation (include/iqdemod.h): IQD_WBFM_PATH=stream|tiles
    // (+1 / -1, 0 = choose), IQD_FULL_GRID, IQD_STREAM_WGS=<n>, IQD_PLAN_CHUNKS=<k>
    int env_path = 0;
    uint64_t env_stream_min_seg = 0;     // IQD_STREAM_MIN_SEG (0: the measured per-family thresholds)
    uint32_t env_am_stream_min = AM_STREAM_MIN_PCM;   // IQD_AM_STREAM_MIN (measurement runs: 513 = the rule of rounds 2-3)
    uint32_t env_d4_gran = 128;            // IQD_D4_GRAN: segment-length granule of the FM / AM / SSB pipelines (measurement runs)
    bool env_full_grid = false;
    bool env_mixed_forked = false;         // IQD_MIXED=forked: several families as kernels of their own side by side (A/B runs)
    // ns per sample of a segment (lead-in included) of one workgroup's 192 segments in lock step, inside the one launch that
    // holds all four pipelines: WBFM 224 us for 3072 + 768, FM 225 for 5120 + 768, AM 214 for 9472 + 384, SSB 226 for 9472 + 1280
    // (tools/mixed_probe.py, 4096 channels x 2^16).  IQD_FAMILY_NS=am,fm,wbfm,ssb; IQD_SHARES=cost keeps the proportional shares.
    float fam_ns[FAM_COUNT] = {21.7f, 38.2f, 58.3f, 21.0f};
    bool env_shares_by_cost = false;
    float fam_weight[FAM_COUNT] = {3.4f, 6.3f, 10.8f, 3.6f};   // relative cost per channel-sample of the streaming pipelines: AM, FM, WBFM, SSB
    uint32_t env_stream_wgs = 0, env_plan_chunks = 0, env_stream_gran = 0;
    size_t dcr_layout[2][2] = {{~(size_t)0, 0}, {~(size_t)0, 0}};   // AM / SSB: where the DC redo flags sit in their buffer, and how many
    bool any_gated = false, any_agc = false;
    bool demod_bypass = false;             // inside iqd_demod_accept: the demodulator alone - no squelch, tracker, AGC, scanner, magnitudes
    std::vector<AgcConfig> h_agc;           // per channel; the one-shot fields are cleared once applied
    std::vector<uint8_t> agc_touched;       // the device may have moved this channel's IF gain
    bool agc_dirty = true;
    std::vector<ScanConfig> h_scan;         // per channel; one-shot fields cleared once applied
    std::vector<uint8_t> scan_new_cfg;      // FrequencyScanner::newConfigurationAvailable
    bool trace_on = false;
    uint32_t trace_first = 0, trace_n = 0, trace_blocks = 0;   // what gain_trace holds

    // persistent device state
    ChanParams *d_params = nullptr;
    uint8_t *d_tails = nullptr;
    WbfmCarry *d_wcarry = nullptr;
    DcCarry *d_dc = nullptr;
    uint32_t *d_tracker = nullptr;
    AgcConfig *d_agc_cfg = nullptr;
    AgcState *d_agc = nullptr;
    GainEpoch *d_epochs = nullptr;
    std::vector<float> k_applied;           // [n_ch][2]: the WBFM / FM K the device last ran with
    // [n_ch]: WBFM samples a channel still has to consume before its latest gain change is out of every lead-in's reach
    // (GainEpochList on the device; this mirror only decides which instantiation of the streaming kernel runs: the one
    // with the piecewise-gain lookup while any channel of the launch is non-zero here).  Never cleared early: it goes
    // down only by samples the channel's WBFM chain really consumed.
    std::vector<uint32_t> wbfm_epoch_left;
    // ... aged by arithmetic for ungated calls (every channel consumes the whole call); a squelch-gated call's consumption is
    // known to the device only, which then reports: its WBFM tail updates set word [0] of a page-locked pair if any
    // channel's newest change still lies inside its tail, a one-store kernel at the call's end writes the call's number to
    // word [1], and a later call that finds its number there and word [0] clear drops the mirror of that call's channels.
    std::vector<uint32_t> wbfm_epoch_seq;   // the first accept (accept_seq) whose chain ran with the channel's newest change
    uint32_t accept_seq = 0;
    uint32_t *h_epoch_report = nullptr;
    bool epoch_report_pending = false;
    uint32_t epoch_report_seq = 0;
    std::vector<uint32_t> epoch_report_channels;
    uint32_t wbfm_epochs_live = 0;          // channels with wbfm_epoch_left != 0
    std::vector<int32_t> rot_applied;       // [n_ch]: the rotation the device last ran with
    bool rot_changed = false;               // some channel's tails need rewriting (retail_kernel)
    ScanConfig *d_scan_cfg = nullptr;
    ScanState *d_scan = nullptr;
    float *d_atan = nullptr, *d_fmlut = nullptr;
    float *d_half_lut = nullptr;         // streaming WBFM kernel: |atan2| half table, tap matrices per rotation (-1, 0, +1)
    uint32_t *d_amat[3] = {nullptr, nullptr, nullptr};
    uint32_t *d_amat4 = nullptr;         // FM tuner / AM-SSB stage 1 as MFMA operands: [fm: 3 rotations][am: 3 rotations][4][64][4]
    D4Args d4_args{};
    std::vector<float> fm_kmax;          // [n_ch]: like wbfm_kmax, for the FM chain
    bool stream_ok = false;              // the half table's symmetry holds on this host's libm
    std::vector<float> wbfm_kmax;        // [n_ch]: largest |K| a channel has run with since creation (casts stay bounded)
    StreamArgs stream_args{};
    uint64_t stream_handoffs = 0;        // cold segments launched so far (their verification counts only mismatches)
    uint32_t *d_counters = nullptr;      // cumulative, read by iqd_get_stats
    unsigned long long *d_stamps = nullptr;
    uint32_t *h_counters = nullptr;  // pinned

    // per-call scratch
    DevBuf stream_hist;   // boundary records of the streaming WBFM kernel, one StHist per segment
    DevBuf lists[FAM_COUNT + 1], mag_sums, blk_lists, vlen, records, base8k, base8k2, gain_trace, freq_trace, dc_records, dc_records2, repair_flags;
    size_t mag_sums_zero = 0;            // leading elements of mag_sums known to be zero (left so by the last squelch pass)
    // IQD_F_PREPASS_OVERLAP: a squelch-gated call's pre-pass (magnitudes of every block, decisions, open-block lists) runs on a
    // stream of its own, one call ahead of the pipelines: pre-pass(N + 1) overlaps chain(N).  Two sets of its buffers.
    hipStream_t pre_stream = nullptr;
    hipEvent_t ev_pre_done[2] = {nullptr, nullptr}, ev_chain_done[2] = {nullptr, nullptr}, ev_main_decisions = nullptr;
    DevBuf g_sums[2], g_blk[2], g_vlen[2];
    int gate_set = 0;
    bool chain_pending[2] = {false, false};   // ev_chain_done[set] has been recorded: the set's last reader may still run
    bool decisions_on_main = false;           // the last squelch decision pass ran on the main stream (ev_main_decisions)
    bool in_host_path = false;                // inside iqd_accept_iq: its staging copies are ordered on the main stream only
    DevBuf st_iq, st_pcm, st_count, st_mag, st_allowed;  // staging for host-pointer accepts
    // sliced host-pointer accepts: two staging sets, so that slice k+1 crosses PCIe while slice k runs
    DevBuf sl_iq[2], sl_pcm[2], sl_count[2], sl_mag[2], sl_allowed[2];
    hipStream_t copy_stream = nullptr;
    // mixed-mode calls: two side streams beside the engine's, so that the families' kernels share the GPU
    hipStream_t fam_stream[3] = {};     // side streams: with the engine's own, one lane per demodulator family
    hipEvent_t fam_fork = nullptr, fam_join[3] = {};
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    uint32_t *h_slice_counts = nullptr;  // pinned, [2][n_ch of a slice]
    uint32_t *d_closed = nullptr, *h_closed = nullptr;   // squelch-gated calls: did any channel lose a block? (device word, pinned copy)
    size_t h_slice_counts_cap = 0;
    uint8_t *h_small = nullptr;          // page-locked staging of small host-pointer calls: [input | pcm | counts | magnitudes | flags]
    size_t h_small_cap = 0;

    bool profiling = false;
    // profiling: one event pair per timed launch, read back lazily so that accepts stay asynchronous
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free_pairs, ev_pending;
    iqd_stats stats{};
    std::string last_error;

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error = buf;
        return code;
    }
};

// samples per segment from which a launch of family f takes its streaming pipeline (see STREAM_MIN_SEG_*)
// several_families: 0 a call of one family, 1 several families in one launch, 2 several families as kernels on streams
static uint64_t stream_min_seg(const iqd_engine *e, int f, uint64_t vlen, int several_families)
{
    if (e->env_stream_min_seg) return e->env_stream_min_seg;
    if (several_families == 2) return STREAM_MIN_SEG_FORKED;
    if (several_families) return vlen < 8192 ? STREAM_MIN_SEG_MIXED_SHORT : STREAM_MIN_SEG_MIXED;
    if (f == FAM_WBFM) return STREAM_MIN_SEG_WBFM;
    if (f == FAM_FM) return STREAM_MIN_SEG_FM;
    if (f == FAM_AM) return vlen <= 16384 ? STREAM_MIN_SEG_AM_SHORT : STREAM_MIN_SEG_AM;
    return vlen <= 16384 ? STREAM_MIN_SEG_SSB_SHORT : STREAM_MIN_SEG_SSB;
}

#define HIP_TRY(e, call)                                                                          \
    do {                                                                                          \
        hipError_t err_ = (call);                                                                 \
        if (err_ != hipSuccess)                                                                   \
            return (e)->fail(IQD_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err_),    \
                             __FILE__, __LINE__);                                                 \
    } while (0)

// the same, counting what is queued on the device (iqd_stats.device_launches / device_copies: launch_* calls and
// memcpy / memset operations; bench.py --config 0 reports them per block)
#define HIP_LAUNCH(e, call) do { (e)->stats.device_launches++; HIP_TRY(e, call); } while (0)
#define HIP_COPY(e, call) do { (e)->stats.device_copies++; HIP_TRY(e, call); } while (0)

static int family_of_mode(int mode)
{
    switch (mode) {
    case IQD_MODE_AM: return FAM_AM;
    case IQD_MODE_FM: return FAM_FM;
    case IQD_MODE_WBFM: return FAM_WBFM;
    case IQD_MODE_LSB: case IQD_MODE_USB: return FAM_SSB;
    default: return FAM_COUNT;
    }
}

static bool range_ok(const iqd_t *e, uint32_t first, uint32_t n)
{
    return e && n >= 1 && first < e->n_ch && n <= e->n_ch - first;
}

extern "C" {

uint32_t iqd_abi_version(void) { return IQD_ABI_VERSION; }

const char *iqd_strerror(int status)
{
    switch (status) {
    case IQD_OK: return "ok";
    case IQD_EALREADY: return "already in the requested state";
    case IQD_EINVAL: return "invalid argument";
    case IQD_ENODEV: return "no usable HIP device";
    case IQD_ENOMEM: return "out of memory";
    case IQD_EHIP: return "HIP runtime error";
    case IQD_ESTATE: return "exact-state verification failed";
    default: return "unknown status";
    }
}

const char *iqd_last_error(iqd_t *e) { return e ? e->last_error.c_str() : "null engine"; }

int iqd_create(const iqd_config *cfg, iqd_t **out)
{
    if (!cfg || !out || cfg->abi_version != IQD_ABI_VERSION || cfg->n_channels == 0) return IQD_EINVAL;
    uint32_t bb = cfg->block_bytes ? cfg->block_bytes : 32768u;
    if (bb % 256u != 0 || bb > 32768u) return IQD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IQD_ENODEV;
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return IQD_ENODEV;
    }
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess) return IQD_ENODEV;

    iqd_t *e = new (std::nothrow) iqd_engine;
    if (!e) return IQD_ENOMEM;
    e->device = dev;
    {
        hipDeviceProp_t prop;
        e->n_cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    }
    e->n_ch = cfg->n_channels;
    e->block_bytes = bb;
    e->block_samples = bb / 2;
    e->flags = cfg->flags;
    if (const char *env = getenv("IQD_WBFM_PATH")) e->env_path = env[0] == 's' ? 1 : env[0] == 't' ? -1 : 0;
    e->env_full_grid = getenv("IQD_FULL_GRID") != nullptr;
    if (const char *env = getenv("IQD_SHARES")) e->env_shares_by_cost = env[0] == 'c';
    if (const char *env = getenv("IQD_FAMILY_NS")) {
        float w[FAM_COUNT];
        if (sscanf(env, "%f,%f,%f,%f", &w[0], &w[1], &w[2], &w[3]) == 4 && w[0] > 0.f && w[1] > 0.f && w[2] > 0.f && w[3] > 0.f)
            for (int f = 0; f < FAM_COUNT; f++) e->fam_ns[f] = w[f];
    }
    if (const char *env = getenv("IQD_D4_GRAN")) e->env_d4_gran = (uint32_t)atoi(env);
    if (const char *env = getenv("IQD_STREAM_MIN_SEG")) e->env_stream_min_seg = atoi(env) > 0 ? (uint64_t)atoi(env) : 0;
    if (const char *env = getenv("IQD_AM_STREAM_MIN")) e->env_am_stream_min = atoi(env) > 0 ? (uint32_t)atoi(env) : AM_STREAM_MIN_PCM;
    if (const char *env = getenv("IQD_MIXED")) e->env_mixed_forked = env[0] == 'f' && env[1] == 'o';
    if (const char *env = getenv("IQD_FAMILY_WEIGHTS")) {   // "am,fm,wbfm,ssb" (measurement runs)
        float w[FAM_COUNT];
        if (sscanf(env, "%f,%f,%f,%f", &w[0], &w[1], &w[2], &w[3]) == 4 && w[0] > 0 && w[1] > 0 && w[2] > 0 && w[3] > 0)
            for (int f = 0; f < FAM_COUNT; f++) e->fam_weight[f] = w[f];
    }
    if (const char *env = getenv("IQD_STREAM_WGS")) e->env_stream_wgs = atoi(env) > 0 ? (uint32_t)atoi(env) : 0u;
    if (const char *env = getenv("IQD_STREAM_GRAN")) e->env_stream_gran = (uint32_t)atoi(env);
    if (const char *env = getenv("IQD_PLAN_CHUNKS")) e->env_plan_chunks = atoi(env) > 0 ? (uint32_t)atoi(env) : 0u;
    build_consts(e->consts);
    e->h_params.resize(e->n_ch);
    for (auto &p : e->h_params) default_params(p);
    // AutomaticGainControl constructor defaults (AutomaticGainControl.cc:113-189; operating point: Radio.cc:184)
    AgcConfig agc0{};
    agc0.enabled = 0; agc0.type = 1; agc0.operating_point = -12; agc0.deadband = 1; agc0.alpha = 0.8f;
    agc0.blanking_limit = 1; agc0.reset_blanking = 0; agc0.set_gain = 0xffffffffu;
    e->h_agc.assign(e->n_ch, agc0);
    e->agc_touched.assign(e->n_ch, 0);
    AgcState st0{};
    st0.rx_gain = 24; st0.if_gain = 24; st0.filtered = 24.f; st0.normalized = -24; st0.signal_magnitude = 64;
    std::vector<AgcState> agc_states(e->n_ch, st0);
    e->k_applied.resize(2 * (size_t)e->n_ch);
    e->wbfm_epoch_left.assign(e->n_ch, 0u);
    e->wbfm_epoch_seq.assign(e->n_ch, 0u);
    for (uint32_t c = 0; c < e->n_ch; c++) {
        e->k_applied[2 * c] = e->h_params[c].wbfm_k;
        e->k_applied[2 * c + 1] = e->h_params[c].fm_k;
        e->h_params[c].wbfm_k_prev = e->h_params[c].wbfm_k;
        e->h_params[c].fm_k_prev = e->h_params[c].fm_k;
        e->h_params[c].k_changed = 0;
        e->h_params[c].rotation_prev = e->h_params[c].rotation;
    }
    e->rot_applied.resize(e->n_ch);
    for (uint32_t c = 0; c < e->n_ch; c++) e->rot_applied[c] = e->h_params[c].rotation;
    // FrequencyScanner constructor defaults (FrequencyScanner.cc:96-131)
    ScanConfig sc0{};
    sc0.start_hz = sc0.end_hz = 162550000ull;
    e->h_scan.assign(e->n_ch, sc0);
    e->scan_new_cfg.assign(e->n_ch, 0);
    ScanState ss0{162550000ull, 0ull};
    std::vector<ScanState> scan_states(e->n_ch, ss0);

    std::vector<float> atan_lut, fm_lut;
    build_atan2_lut(atan_lut);
    build_fm_lut(fm_lut);
    std::vector<float> half_lut((size_t)129 * ST_ROW_FLOATS);
    e->stream_ok = build_half_lut(half_lut.data());
    std::vector<uint32_t> amat[3];
    {
        int16_t pre[16];
        quantize_q15(taps::WBFM_PRE, 16, pre);
        for (int r = 0; r < 3; r++) {
            amat[r].assign(8 * 64 * 4, 0u);
            build_stream_amat(r - 1, pre, amat[r].data());
        }
    }
    std::vector<uint32_t> amat4((size_t)2 * 3 * 4 * 64 * 4, 0u);
    for (int r = 0; r < 3; r++) {
        build_decim4_amat(r - 1, e->consts.fm_tuner, 32, amat4.data() + (size_t)r * 4 * 64 * 4);
        build_decim4_amat(r - 1, e->consts.am_s1, 8, amat4.data() + (size_t)(3 + r) * 4 * 64 * 4);
    }
    build_d4_taps(e->consts, e->d4_args);
    e->fm_kmax.resize(e->n_ch);
    for (uint32_t c = 0; c < e->n_ch; c++) e->fm_kmax[c] = fabsf(e->h_params[c].fm_k);
    build_stream_taps(e->consts.wbfm_d1, e->consts.post12, e->consts.audio40, e->stream_args);
    e->stream_args.b0 = e->consts.deemph_b0;
    e->stream_args.a1 = e->consts.deemph_a1;
    e->wbfm_kmax.resize(e->n_ch);
    for (uint32_t c = 0; c < e->n_ch; c++) e->wbfm_kmax[c] = fabsf(e->h_params[c].wbfm_k);

    const size_t n = e->n_ch;
    bool ok = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess;
    {
        static std::mutex attr_mu;   // engines may be created from several threads
        std::lock_guard<std::mutex> lk(attr_mu);
        ok = ok && init_wbfm_stream_kernels() == hipSuccess && init_d4_stream_kernels() == hipSuccess && init_mixed_stream_kernels() == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&e->d_params, n * sizeof(ChanParams)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_tails, n * FAM_COUNT * TAIL_BYTES) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_wcarry, n * sizeof(WbfmCarry)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_dc, n * 2 * sizeof(DcCarry)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_tracker, n * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_agc_cfg, n * sizeof(AgcConfig)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_agc, n * sizeof(AgcState)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_epochs, n * sizeof(GainEpoch)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_scan_cfg, n * sizeof(ScanConfig)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_scan, n * sizeof(ScanState)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_atan, atan_lut.size() * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_fmlut, fm_lut.size() * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_half_lut, half_lut.size() * sizeof(float)) == hipSuccess;
    for (int r = 0; r < 3; r++) ok = ok && hipMalloc((void **)&e->d_amat[r], amat[r].size() * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_amat4, amat4.size() * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_counters, CNT_COUNT * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(e->d_counters, 0, CNT_COUNT * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_stamps, 32768 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(e->d_stamps, 0, 32768 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&e->h_counters, CNT_COUNT * sizeof(uint32_t)) == hipSuccess;
    if (ok) {
        ok = hipMemsetAsync(e->d_tails, 0x80, n * FAM_COUNT * TAIL_BYTES, e->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(e->d_wcarry, 0, n * sizeof(WbfmCarry), e->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(e->d_dc, 0, n * 2 * sizeof(DcCarry), e->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(e->d_tracker, 0, n * sizeof(uint32_t), e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->d_agc, agc_states.data(), n * sizeof(AgcState), hipMemcpyHostToDevice,
                                  e->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(e->d_epochs, 0x7f, n * sizeof(GainEpoch), e->stream) == hipSuccess;   // "long ago"
        ok = ok && hipMemcpyAsync(e->d_scan, scan_states.data(), n * sizeof(ScanState), hipMemcpyHostToDevice,
                                  e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->d_atan, atan_lut.data(), atan_lut.size() * sizeof(float),
                                  hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->d_fmlut, fm_lut.data(), fm_lut.size() * sizeof(float),
                                  hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->d_half_lut, half_lut.data(), half_lut.size() * sizeof(float),
                                  hipMemcpyHostToDevice, e->stream) == hipSuccess;
        for (int r = 0; r < 3; r++)
            ok = ok && hipMemcpyAsync(e->d_amat[r], amat[r].data(), amat[r].size() * sizeof(uint32_t),
                                      hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->d_amat4, amat4.data(), amat4.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  e->stream) == hipSuccess;
        ok = ok && upload_consts(e->consts, e->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;
    }
    if (!ok) {
        iqd_destroy(e);
        return IQD_ENOMEM;
    }
    *out = e;
    return IQD_OK;
}

void iqd_destroy(iqd_t *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    void *ptrs[] = {e->d_params, e->d_tails, e->d_wcarry, e->d_dc, e->d_tracker, e->d_agc_cfg, e->d_agc, e->d_epochs, e->d_scan_cfg, e->d_scan,
                    e->d_atan, e->d_fmlut, e->d_counters, e->d_stamps, e->d_half_lut, e->d_amat[0], e->d_amat[1], e->d_amat[2], e->d_amat4};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    for (auto &b : e->lists) b.release();
    DevBuf *bufs[] = {&e->stream_hist, &e->mag_sums, &e->blk_lists, &e->vlen, &e->records, &e->base8k, &e->base8k2, &e->gain_trace, &e->freq_trace, &e->dc_records, &e->dc_records2, &e->repair_flags,
                      &e->st_iq, &e->st_pcm, &e->st_count, &e->st_mag, &e->st_allowed};
    for (DevBuf *b : bufs) b->release();
    for (int b = 0; b < 2; b++) {
        e->sl_iq[b].release(); e->sl_pcm[b].release(); e->sl_count[b].release();
        e->sl_mag[b].release(); e->sl_allowed[b].release();
        if (e->ev_in[b]) (void)hipEventDestroy(e->ev_in[b]);
        if (e->ev_free[b]) (void)hipEventDestroy(e->ev_free[b]);
    }
    if (e->pre_stream) {
        (void)hipStreamSynchronize(e->pre_stream);
        (void)hipStreamDestroy(e->pre_stream);
        for (int k = 0; k < 2; k++) {
            (void)hipEventDestroy(e->ev_pre_done[k]);
            (void)hipEventDestroy(e->ev_chain_done[k]);
            e->g_sums[k].release(); e->g_blk[k].release(); e->g_vlen[k].release();
        }
        (void)hipEventDestroy(e->ev_main_decisions);
    }
    if (e->h_slice_counts) (void)hipHostFree(e->h_slice_counts);
    if (e->h_small) (void)hipHostFree(e->h_small);
    if (e->h_epoch_report) (void)hipHostFree(e->h_epoch_report);
    if (e->d_closed) (void)hipFree(e->d_closed);
    if (e->h_closed) (void)hipHostFree(e->h_closed);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    for (int f = 0; f < 3; f++) {
        if (e->fam_stream[f]) (void)hipStreamDestroy(e->fam_stream[f]);
        if (e->fam_join[f]) (void)hipEventDestroy(e->fam_join[f]);
    }
    if (e->fam_fork) (void)hipEventDestroy(e->fam_fork);
    for (auto *v : {&e->ev_free_pairs, &e->ev_pending})
        for (auto &pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int iqd_set_mode(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int mode)
{
    if (!range_ok(e, first_ch, n_ch) || mode < IQD_MODE_NONE || mode > IQD_MODE_USB) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        e->h_params[c].mode = mode;
        if (mode == IQD_MODE_LSB) e->h_params[c].ssb_lsb = 1;  // IqDataProcessor.cc:244-256
        if (mode == IQD_MODE_USB) e->h_params[c].ssb_lsb = 0;
    }
    e->params_dirty = e->lists_dirty = true;
    return IQD_OK;
}

int iqd_set_gain(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod, float gain)
{
    if (!range_ok(e, first_ch, n_ch) || demod < IQD_DEMOD_AM || demod > IQD_DEMOD_SSB) return IQD_EINVAL;
    static const int fam[5] = {0, FAM_AM, FAM_FM, FAM_WBFM, FAM_SSB};
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        e->h_params[c].gain[fam[demod]] = gain;
        derive_params(e->h_params[c]);
        const float ak = fabsf(e->h_params[c].wbfm_k), fk = fabsf(e->h_params[c].fm_k);
        if (!(ak <= e->wbfm_kmax[c])) e->wbfm_kmax[c] = ak;   // (NaN included: never "bounded" again)
        if (!(fk <= e->fm_kmax[c])) e->fm_kmax[c] = fk;
    }
    e->params_dirty = true;
    return IQD_OK;
}

int iqd_set_squelch(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int32_t threshold)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) e->h_params[c].threshold = threshold;
    e->params_dirty = e->lists_dirty = true;
    return IQD_OK;
}

int iqd_set_rx_gain_db(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t gain_db)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        e->h_params[c].rx_gain_db = gain_db;
        e->h_agc[c].set_gain = gain_db;                      // the device holds the gain in force
        if (!e->h_agc[c].enabled) e->agc_touched[c] = 0;     // and nothing there will move it
    }
    e->params_dirty = e->lists_dirty = e->agc_dirty = true;
    return IQD_OK;
}

// ---- AutomaticGainControl: the reference's setters one to one (they validate like the reference does) ----
extern "C++" {
template <class F>
static int agc_update(iqd_t *e, uint32_t first_ch, uint32_t n_ch, bool valid, F f)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    if (!valid) return e->fail(IQD_EINVAL, "AGC parameter out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) f(e->h_agc[c], c);
    e->agc_dirty = e->lists_dirty = true;
    return IQD_OK;
}
}  // extern "C++"

int iqd_agc_set_type(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t type)
{
    return agc_update(e, first_ch, n_ch, type == IQD_AGC_LOWPASS || type == IQD_AGC_HARRIS,
                      [&](AgcConfig &a, uint32_t) { a.type = type; });
}

int iqd_agc_set_deadband(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t deadband_db)
{
    return agc_update(e, first_ch, n_ch, deadband_db <= 10, [&](AgcConfig &a, uint32_t) { a.deadband = (int32_t)deadband_db; });
}

int iqd_agc_set_blanking_limit(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t limit)
{
    return agc_update(e, first_ch, n_ch, limit <= 10, [&](AgcConfig &a, uint32_t) {
        a.blanking_limit = limit;
        a.reset_blanking = 1;   // setBlankingLimit() also resets the blanking system
    });
}

int iqd_agc_set_operating_point(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int32_t dbfs)
{
    return agc_update(e, first_ch, n_ch, true, [&](AgcConfig &a, uint32_t) { a.operating_point = dbfs; });
}

int iqd_agc_set_filter_coefficient(iqd_t *e, uint32_t first_ch, uint32_t n_ch, float coefficient)
{
    // the reference compares the float against double literals
    return agc_update(e, first_ch, n_ch, (coefficient >= 0.001) && (coefficient < 0.999),
                      [&](AgcConfig &a, uint32_t) { a.alpha = coefficient; });
}

int iqd_agc_enable(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int enabled)
{
    uint32_t changed = 0;
    int rc = agc_update(e, first_ch, n_ch, true, [&](AgcConfig &a, uint32_t c) {
        if (enabled && !a.enabled) {
            a.reset_blanking = 1;   // enable() of a disabled AGC resets the blanking system
            a.enabled = 1;
            e->agc_touched[c] = 1;
            changed++;
        } else if (!enabled && a.enabled) {
            a.enabled = 0;
            changed++;
        }
    });
    if (rc == IQD_OK && !changed) return IQD_EALREADY;   // enable()/disable() return false
    return rc;
}

// Uploads the channel parameters if they changed.  A WBFM / FM gain that differs from what the device last ran with
// hands the old K over for the histories (GainEpoch); agc_sync() then applies and clears the flags.  Call with e->mu held.
static int upload_params(iqd_t *e)
{
    if (!e->params_dirty) return IQD_OK;
    for (uint32_t c = 0; c < e->n_ch; c++) {
        ChanParams &p = e->h_params[c];
        // (a change still pending - uploaded by a front-end call, not yet applied by agc_sync - keeps ITS "before":
        // the histories on the device were made with that one, whatever the value was set to in between)
        if (f2u(p.wbfm_k) != f2u(e->k_applied[2 * c])) {
            if (!(p.k_changed & 1u)) p.wbfm_k_prev = e->k_applied[2 * c];
            p.k_changed |= 1u;
            e->k_applied[2 * c] = p.wbfm_k;
            if (!e->wbfm_epoch_left[c]) e->wbfm_epochs_live++;
            e->wbfm_epoch_left[c] = (uint32_t)TAIL;
            e->wbfm_epoch_seq[c] = e->accept_seq + 1;
        }
        if (f2u(p.fm_k) != f2u(e->k_applied[2 * c + 1])) {
            if (!(p.k_changed & 2u)) p.fm_k_prev = e->k_applied[2 * c + 1];
            p.k_changed |= 2u;
            e->k_applied[2 * c + 1] = p.fm_k;
        }
        if (p.rotation != e->rot_applied[c]) {
            if (!(p.k_changed & 4u)) p.rotation_prev = e->rot_applied[c];
            p.k_changed |= 4u;
            e->rot_applied[c] = p.rotation;
            e->rot_changed = true;
        }
        if (p.k_changed) e->agc_dirty = true;
    }
    HIP_TRY(e, hipMemcpyAsync(e->d_params, e->h_params.data(), e->n_ch * sizeof(ChanParams), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));  // the mirror may change once the lock is dropped
    e->params_dirty = false;
    return IQD_OK;
}

// Uploads the AGC configuration and applies the pending one-shot commands.  Call with e->mu held.
static int agc_sync(iqd_t *e)
{
    if (!e->agc_dirty) return IQD_OK;
    hipStream_t s = e->stream;
    HIP_TRY(e, hipMemcpyAsync(e->d_agc_cfg, e->h_agc.data(), e->n_ch * sizeof(AgcConfig), hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(e->d_scan_cfg, e->h_scan.data(), e->n_ch * sizeof(ScanConfig), hipMemcpyHostToDevice, s));
    if (e->rot_changed) {   // before the flags are cleared: rewrite the tails of the channels whose rotation changed
        HIP_TRY(e, launch_retail(e->d_tails, e->d_params, e->n_ch, s));
        e->rot_changed = false;
    }
    HIP_TRY(e, launch_agc_apply(e->d_agc_cfg, e->d_agc, e->d_scan_cfg, e->d_scan, e->d_params, e->d_epochs, e->n_ch, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    for (auto &a : e->h_agc) { a.reset_blanking = 0; a.set_gain = 0xffffffffu; }
    for (auto &c : e->h_scan) c.set_current_flag = 0;
    for (auto &p : e->h_params) p.k_changed = 0;   // (the kernel cleared the device copies)
    e->agc_dirty = false;
    return IQD_OK;
}

int iqd_agc_get_state(iqd_t *e, uint32_t ch, iqd_agc_state *out)
{
    if (!e || ch >= e->n_ch || !out) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    AgcState st;
    AgcConfig cfg;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        int rc = agc_sync(e);
        if (rc != IQD_OK) return rc;
        cfg = e->h_agc[ch];
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(&st, e->d_agc + ch, sizeof(st), hipMemcpyDeviceToHost));
    out->enabled = cfg.enabled; out->type = cfg.type; out->operating_point_dbfs = cfg.operating_point;
    out->deadband_db = (uint32_t)cfg.deadband; out->blanking_limit = cfg.blanking_limit; out->alpha = cfg.alpha;
    out->rx_gain_db = st.rx_gain; out->if_gain_db = st.if_gain; out->filtered_if_gain_db = st.filtered;
    out->blanking_counter = st.blank_ctr; out->gain_was_adjusted = st.adjusted;
    out->normalized_level_dbfs = st.normalized; out->signal_magnitude = st.signal_magnitude;
    return IQD_OK;
}

int iqd_get_rx_gain_db(iqd_t *e, uint32_t ch, uint32_t *gain_db)
{
    iqd_agc_state st;
    if (!gain_db) return IQD_EINVAL;
    int rc = iqd_agc_get_state(e, ch, &st);
    if (rc == IQD_OK) *gain_db = st.rx_gain_db;
    return rc;
}

// ---- FrequencyScanner: the reference's methods one to one -------------------------------------------
int iqd_scanner_set_parameters(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint64_t start_hz, uint64_t end_hz,
                               uint64_t increment_hz)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    uint32_t changed = 0;
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        ScanConfig &sc = e->h_scan[c];
        if (sc.scanning) continue;           // setScanParameters() returns false while scanning
        sc.start_hz = start_hz; sc.end_hz = end_hz; sc.increment_hz = increment_hz;
        e->scan_new_cfg[c] = 1;
        changed++;
    }
    if (!changed) return IQD_EALREADY;
    e->agc_dirty = true;
    return IQD_OK;
}

int iqd_scanner_start(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int start)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    uint32_t changed = 0;
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        ScanConfig &sc = e->h_scan[c];
        if (start && !sc.scanning) {
            if (e->scan_new_cfg[c]) {        // start(): jump to the end frequency and tune there
                sc.set_current = sc.end_hz;
                sc.set_current_flag = 1;
                e->scan_new_cfg[c] = 0;
            }
            sc.scanning = 1;
            changed++;
        } else if (!start && sc.scanning) {
            sc.scanning = 0;
            changed++;
        }
    }
    if (!changed) return IQD_EALREADY;       // start()/stop() return false
    e->agc_dirty = true;
    return IQD_OK;
}

int iqd_scanner_get(iqd_t *e, uint32_t ch, uint64_t *current_hz, uint64_t *tune_count, int *scanning)
{
    if (!e || ch >= e->n_ch) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    {
        std::lock_guard<std::mutex> lk(e->mu);
        int rc = agc_sync(e);
        if (rc != IQD_OK) return rc;
        if (scanning) *scanning = (int)e->h_scan[ch].scanning;
    }
    ScanState ss;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(&ss, e->d_scan + ch, sizeof(ss), hipMemcpyDeviceToHost));
    if (current_hz) *current_hz = ss.current_hz;
    if (tune_count) *tune_count = ss.tune_count;
    return IQD_OK;
}

int iqd_get_frequency_trace(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint64_t *out, size_t n_blocks)
{
    if (!e || !out) return IQD_EINVAL;
    if (!e->trace_n || first_ch < e->trace_first || n_ch == 0 || first_ch + n_ch > e->trace_first + e->trace_n ||
        n_blocks != e->trace_blocks)
        return e->fail(IQD_EINVAL, "no frequency trace for that range (tracing on? same channels and block count as the last accept?)");
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out, e->freq_trace.as<unsigned long long>() + (size_t)(first_ch - e->trace_first) * n_blocks,
                         (size_t)n_ch * n_blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return IQD_OK;
}

int iqd_set_gain_trace(iqd_t *e, int enabled)
{
    if (!e) return IQD_EINVAL;
    e->trace_on = enabled != 0;
    if (!e->trace_on) e->trace_n = 0;
    return IQD_OK;
}

int iqd_get_gain_trace(iqd_t *e, uint32_t first_ch, uint32_t n_ch, uint32_t *out, size_t n_blocks)
{
    if (!e || !out) return IQD_EINVAL;
    if (!e->trace_n || first_ch < e->trace_first || n_ch == 0 || first_ch + n_ch > e->trace_first + e->trace_n ||
        n_blocks != e->trace_blocks)
        return e->fail(IQD_EINVAL, "no gain trace for that range (tracing on? same channels and block count as the last accept?)");
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out, e->gain_trace.as<uint32_t>() + (size_t)(first_ch - e->trace_first) * n_blocks,
                         (size_t)n_ch * n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return IQD_OK;
}

int iqd_set_rotation(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int rotation)
{
    if (!range_ok(e, first_ch, n_ch) || rotation < -1 || rotation > 1) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) e->h_params[c].rotation = rotation;
    e->params_dirty = e->lists_dirty = true;   // (the families' channel lists are grouped by selector for the streaming kernels)
    return IQD_OK;
}

int iqd_reset(iqd_t *e, uint32_t first_ch, uint32_t n_ch)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, launch_reset(e->d_tails, e->d_wcarry, e->d_dc, first_ch, n_ch, 0xfu, e->stream));
    return IQD_OK;
}

int iqd_reset_demod(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod)
{
    if (!range_ok(e, first_ch, n_ch) || demod < IQD_DEMOD_AM || demod > IQD_DEMOD_SSB) return IQD_EINVAL;
    static const int fam[5] = {0, FAM_AM, FAM_FM, FAM_WBFM, FAM_SSB};
    (void)hipSetDevice(e->device);
    HIP_TRY(e, launch_reset(e->d_tails, e->d_wcarry, e->d_dc, first_ch, n_ch, 1u << fam[demod], e->stream));
    return IQD_OK;
}

int iqd_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int iqd_get_device(iqd_t *e, int *device)
{
    if (!e || !device) return IQD_EINVAL;
    *device = e->device;
    return IQD_OK;
}

int iqd_get_channel_mode(iqd_t *e, uint32_t ch, int *mode)
{
    if (!e || ch >= e->n_ch || !mode) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    *mode = e->h_params[ch].mode;
    return IQD_OK;
}

int iqd_get_channel_gain(iqd_t *e, uint32_t ch, int demod, float *gain)
{
    if (!e || ch >= e->n_ch || !gain || demod < IQD_DEMOD_AM || demod > IQD_DEMOD_SSB) return IQD_EINVAL;
    static const int fam[5] = {0, FAM_AM, FAM_FM, FAM_WBFM, FAM_SSB};
    std::lock_guard<std::mutex> lk(e->mu);
    *gain = e->h_params[ch].gain[fam[demod]];
    return IQD_OK;
}

int iqd_set_profiling(iqd_t *e, int enabled)
{
    if (!e) return IQD_EINVAL;
    e->profiling = enabled != 0;
    return IQD_OK;
}

int iqd_get_stats(iqd_t *e, iqd_stats *out)
{
    if (!e || !out) return IQD_EINVAL;
    // the verification / repair counters live on the device (accepts do not wait for them): wait and read
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(e->h_counters, e->d_counters, CNT_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (auto &pr : e->ev_pending) {   // the stream is idle: every recorded pair has completed
        float ms = 0.f;
        HIP_TRY(e, hipEventElapsedTime(&ms, pr.first, pr.second));
        e->stats.chain_kernel_ms += ms;
        e->stats.chain_kernel_count++;
        e->ev_free_pairs.push_back(pr);
    }
    e->ev_pending.clear();
    e->stats.state_checks = e->h_counters[CNT_TILE_CHECKS] + e->stream_handoffs - e->h_counters[CNT_STREAM_MISMATCH];
    e->stats.segment_repairs = e->h_counters[CNT_SEG_REPAIRS];
    e->stats.state_repairs = (uint64_t)e->h_counters[CNT_TILE_REPAIRS] + e->h_counters[CNT_DC_REDO];
    *out = e->stats;
    return IQD_OK;
}

int iqd_synchronize(iqd_t *e)
{
    if (!e) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return IQD_OK;
}

void *iqd_stream(iqd_t *e) { return e ? (void *)e->stream : nullptr; }

// Diagnostic builds (-DIQD_STAMPS) only: per-phase cycle sums of the chain kernel.
int iqd_debug_stamps(iqd_t *e, unsigned long long *out16)
{
    if (!e || !out16) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out16, e->d_stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return IQD_OK;
}

int iqd_debug_stamps_ext(iqd_t *e, unsigned long long *out, uint32_t n)
{
    if (!e || !out || n > 32768) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out, e->d_stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return IQD_OK;
}

int iqd_dev_alloc(iqd_t *e, size_t bytes, void **out)
{
    if (!e || !out) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    if (hipMalloc(out, bytes) != hipSuccess) return e->fail(IQD_ENOMEM, "hipMalloc(%zu) failed", bytes);
    return IQD_OK;
}

int iqd_dev_free(iqd_t *e, void *p)
{
    if (!e) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipFree(p));
    return IQD_OK;
}

int iqd_dev_upload(iqd_t *e, void *dst, const void *src, size_t bytes)
{
    if (!e) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return IQD_OK;
}

int iqd_dev_download(iqd_t *e, void *dst, const void *src, size_t bytes)
{
    if (!e) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return IQD_OK;
}

// A call is k whole blocks of block_bytes (k consecutive acceptIqData calls) - or ONE short block: the reference
// forwards whatever rtlsdr_read_sync returned (Radio.cc:1895-1906; DataConsumer.cc:238-242 only counts short
// reads) and acceptIqData averages the squelch over that call's samples (IqDataProcessor.cc:722-749).  A short
// block is a whole number of 64-byte units (32 samples: the period of the chains' /32 commutators; USB reads are
// multiples of 512 bytes) in every mode - the WBFM chain's 128-sample de-emphasis segments take a ragged head and tail
// since round 4 (iqd_wbfm.h: IirShape).  Returns the block size in force for the call, 0 if the length is not acceptable.
static uint32_t call_block_bytes(iqd_t *e, uint32_t first_ch, uint32_t n_ch, size_t bytes_per_ch)
{
    (void)first_ch;
    (void)n_ch;
    if (bytes_per_ch == 0) return 0;
    if (bytes_per_ch % e->block_bytes == 0) return e->block_bytes;
    if (bytes_per_ch >= e->block_bytes || bytes_per_ch % 64 != 0) return 0;
    return (uint32_t)bytes_per_ch;
}
#define IQD_LEN_MSG "bytes_per_ch (%zu) must be a positive multiple of block_bytes (%u), or one short block: a multiple of 64 below it"

// The front end alone: what the reference leaves in the caller's buffer / sends from its IQ dump tap.
int iqd_front_end_device(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const void *iq_dev, size_t bytes_per_ch,
                         void *out_dev)
{
    if (!range_ok(e, first_ch, n_ch) || !iq_dev || !out_dev) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    if (bytes_per_ch == 0 || bytes_per_ch % 8 != 0)   // the rotation pattern spans 4 samples (IqDataProcessor.cc:567-611)
        return e->fail(IQD_EINVAL, "bytes_per_ch (%zu) must be a positive multiple of 8", bytes_per_ch);
    if ((((uintptr_t)iq_dev) | ((uintptr_t)out_dev)) & 7) return e->fail(IQD_EINVAL, "buffers must be 8-byte aligned");
    (void)hipSetDevice(e->device);
    {
        std::lock_guard<std::mutex> lk(e->mu);
        int rc = upload_params(e);
        if (rc != IQD_OK) return rc;
    }
    HIP_TRY(e, launch_front_end((const uint8_t *)iq_dev, (int8_t *)out_dev, e->d_params, first_ch, n_ch, bytes_per_ch, e->stream));
    return IQD_OK;
}

int iqd_front_end(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const uint8_t *iq, size_t bytes_per_ch, int8_t *out)
{
    if (!range_ok(e, first_ch, n_ch) || !iq || !out) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    (void)hipSetDevice(e->device);
    const size_t bytes = (size_t)n_ch * bytes_per_ch;
    HIP_TRY(e, e->st_iq.ensure(bytes));
    HIP_TRY(e, e->st_pcm.ensure(bytes));
    HIP_TRY(e, hipMemcpyAsync(e->st_iq.p, iq, bytes, hipMemcpyHostToDevice, e->stream));
    int rc = iqd_front_end_device(e, first_ch, n_ch, e->st_iq.p, bytes_per_ch, e->st_pcm.p);
    if (rc != IQD_OK) return rc;
    HIP_TRY(e, hipMemcpyAsync(out, e->st_pcm.p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return IQD_OK;
}

// ---- resamplers (Filters/Decimator.cc, Interpolator.cc, Int16/Interpolator_int16.cc) ---------------------
}  // extern "C"

struct iqd_resampler {
    iqd_t *e = nullptr;
    int kind = 0;
    uint32_t n_ch = 0, factor = 1, n_taps = 0 /* per output */, hist_len = 0, elem = 4;
    uint64_t count = 0;          // samples accepted so far (the decimator's commutator phase)
    DevBuf taps, hist[2], st_in, st_out;
    int cur = 0;
};

extern "C" {

int iqd_resampler_create(iqd_t *e, int kind, const float *taps, uint32_t n_taps, uint32_t factor, uint32_t n_channels,
                         iqd_resampler_t **out)
{
    if (!e || !taps || !out || kind < IQD_RESAMPLE_DECIMATE_F32 || kind > IQD_RESAMPLE_INTERPOLATE_I16 || n_taps == 0 ||
        factor == 0 || n_channels == 0)
        return e ? e->fail(IQD_EINVAL, "bad resampler parameters") : IQD_EINVAL;
    (void)hipSetDevice(e->device);
    iqd_resampler *r = new (std::nothrow) iqd_resampler;
    if (!r) return IQD_ENOMEM;
    r->e = e; r->kind = kind; r->n_ch = n_channels; r->factor = factor;
    r->elem = kind == IQD_RESAMPLE_INTERPOLATE_I16 ? 2 : 4;
    std::vector<float> tf;
    std::vector<int16_t> tq;
    if (kind == IQD_RESAMPLE_DECIMATE_F32) {
        r->n_taps = n_taps;
        tf.assign(taps, taps + n_taps);
    } else {   // polyphase order: sub-filter p holds h[p], h[p+L], ... (createPolyphaseCoefficients)
        const uint32_t q = n_taps / factor;
        if (q == 0) { delete r; return e->fail(IQD_EINVAL, "an interpolator needs at least `factor` taps"); }
        r->n_taps = q;
        std::vector<int16_t> hq(n_taps);
        if (kind == IQD_RESAMPLE_INTERPOLATE_I16) quantize_q15(taps, (int)n_taps, hq.data());
        for (uint32_t p = 0; p < factor; p++)
            for (uint32_t k = 0; k < q; k++) {
                if (kind == IQD_RESAMPLE_INTERPOLATE_I16) tq.push_back(hq[p + k * factor]);
                else tf.push_back(taps[p + k * factor]);
            }
    }
    r->hist_len = r->n_taps;   // one more than strictly needed; keeps the indexing plain
    const void *src = tq.empty() ? (const void *)tf.data() : (const void *)tq.data();
    const size_t tbytes = tq.empty() ? tf.size() * sizeof(float) : tq.size() * sizeof(int16_t);
    bool ok = r->taps.ensure(tbytes) == hipSuccess;
    for (int b = 0; b < 2 && ok; b++) ok = r->hist[b].ensure((size_t)r->n_ch * r->hist_len * r->elem) == hipSuccess;
    ok = ok && hipMemcpy(r->taps.p, src, tbytes, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(r->hist[0].p, 0, (size_t)r->n_ch * r->hist_len * r->elem) == hipSuccess;
    if (!ok) { iqd_resampler_destroy(r); return e->fail(IQD_ENOMEM, "resampler allocation failed"); }
    *out = r;
    return IQD_OK;
}

void iqd_resampler_destroy(iqd_resampler_t *r)
{
    if (!r) return;
    (void)hipSetDevice(r->e->device);
    (void)hipStreamSynchronize(r->e->stream);
    r->taps.release(); r->hist[0].release(); r->hist[1].release(); r->st_in.release(); r->st_out.release();
    delete r;
}

int iqd_resampler_reset(iqd_resampler_t *r)   // resetFilterState()
{
    if (!r) return IQD_EINVAL;
    (void)hipSetDevice(r->e->device);
    HIP_TRY(r->e, hipMemsetAsync(r->hist[r->cur].p, 0, (size_t)r->n_ch * r->hist_len * r->elem, r->e->stream));
    r->count = 0;
    return IQD_OK;
}

size_t iqd_resampler_out_count(const iqd_resampler_t *r, size_t n_in)
{
    if (!r) return 0;
    if (r->kind != IQD_RESAMPLE_DECIMATE_F32) return n_in * r->factor;
    return (size_t)((r->count % r->factor + n_in) / r->factor);
}

int iqd_resampler_run_device(iqd_resampler_t *r, const void *in_dev, size_t n_in, void *out_dev)
{
    if (!r || !in_dev || !out_dev) return IQD_EINVAL;
    iqd_t *e = r->e;
    if (n_in == 0) return IQD_OK;
    if (n_in > 0x7fffffffu / r->factor) return e->fail(IQD_EINVAL, "too many samples in one call");
    (void)hipSetDevice(e->device);
    const uint32_t n_out = (uint32_t)iqd_resampler_out_count(r, n_in);
    const uint32_t phase = (uint32_t)(r->count % r->factor);
    const uint32_t first = r->factor - 1 - phase;   // decimator: input index that completes the first group
    HIP_TRY(e, launch_resample(r->kind, in_dev, out_dev, r->hist[r->cur].p, r->hist[r->cur ^ 1].p, r->taps.p, r->n_ch,
                               (uint32_t)n_in, n_out, r->hist_len, r->n_taps, r->factor, first, e->stream));
    r->cur ^= 1;
    r->count += n_in;
    return IQD_OK;
}

int iqd_resampler_run(iqd_resampler_t *r, const void *in, size_t n_in, void *out)
{
    if (!r || !in || !out) return IQD_EINVAL;
    iqd_t *e = r->e;
    (void)hipSetDevice(e->device);
    const size_t n_out = iqd_resampler_out_count(r, n_in);
    const size_t ib = (size_t)r->n_ch * n_in * r->elem, ob = (size_t)r->n_ch * n_out * r->elem;
    HIP_TRY(e, r->st_in.ensure(ib ? ib : 16));
    HIP_TRY(e, r->st_out.ensure(ob ? ob : 16));
    HIP_TRY(e, hipMemcpyAsync(r->st_in.p, in, ib, hipMemcpyHostToDevice, e->stream));
    int rc = iqd_resampler_run_device(r, r->st_in.p, n_in, r->st_out.p);
    if (rc != IQD_OK) return rc;
    if (ob) HIP_TRY(e, hipMemcpyAsync(out, r->st_out.p, ob, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return IQD_OK;
}

int iqd_host_alloc(iqd_t *e, size_t bytes, void **out)
{
    if (!e || !out || bytes == 0) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        *out = nullptr;
        return e->fail(IQD_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
    }
    return IQD_OK;
}

int iqd_host_free(iqd_t *e, void *p)
{
    if (!e) return IQD_EINVAL;
    if (p) (void)hipHostFree(p);
    return IQD_OK;
}

int iqd_dev_tile(iqd_t *e, void *dst, size_t period, size_t total)
{
    if (!e || !dst || period == 0 || period % 16 || total % 16 || total < period) return IQD_EINVAL;
    (void)hipSetDevice(e->device);
    HIP_TRY(e, launch_tile_fill((uint8_t *)dst, period, total, e->stream));
    return IQD_OK;
}

// IqDataProcessor::upconvertByFsOver4 / downconvertByFsOver4 (IqDataProcessor.cc:487-611) as the reference offers
// them: in place, on signed bytes, a multiple of 8 of them.
int iqd_convert_fs_over_4(iqd_t *e, int direction, int8_t *buffer, size_t byte_count)
{
    if (!e || !buffer) return e ? e->fail(IQD_EINVAL, "NULL buffer") : IQD_EINVAL;
    if (direction != 1 && direction != -1) return e->fail(IQD_EINVAL, "direction must be +1 (up) or -1 (down)");
    if (byte_count == 0 || byte_count % 8 != 0) return e->fail(IQD_EINVAL, "byte_count (%zu) must be a positive multiple of 8", byte_count);
    (void)hipSetDevice(e->device);
    hipStream_t s = e->stream;
    HIP_TRY(e, e->st_iq.ensure(byte_count));
    HIP_TRY(e, hipMemcpyAsync(e->st_iq.p, buffer, byte_count, hipMemcpyHostToDevice, s));
    HIP_TRY(e, launch_rotate_signed((int8_t *)e->st_iq.p, byte_count, direction, s));
    HIP_TRY(e, hipMemcpyAsync(buffer, e->st_iq.p, byte_count, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    return IQD_OK;
}

// ---- accept ------------------------------------------------------------------------------------

static void rebuild_lists(iqd_t *e, uint32_t first_ch, uint32_t n_ch)
{
    for (auto &l : e->h_lists) l.clear();
    e->any_gated = e->any_agc = false;
    for (uint32_t c = 0; c < n_ch; c++) {
        const ChanParams &p = e->h_params[first_ch + c];
        e->h_lists[family_of_mode(p.mode)].push_back(c);
        ChanParams worst = p;   // a channel whose AGC runs (or ran) may sit at any gain up to the maximum
        if (e->agc_touched[first_ch + c] && worst.rx_gain_db < AGC_MAX_GAIN) worst.rx_gain_db = AGC_MAX_GAIN;
        if (!squelch_always_open(worst, e->consts)) e->any_gated = true;
        if (e->h_agc[first_ch + c].enabled) e->any_agc = true;
    }
    // Each family's channels in the order +Fs/4, no rotation, -Fs/4: the streaming kernels want the segments of one
    // rotation selector next to each other (a P wave's 16 segments share tap matrices); nothing else minds the order.
    for (int f = 0; f < FAM_COUNT; f++) {
        auto &l = e->h_lists[f];
        auto grp = [&](uint32_t c) { return 1 - e->h_params[first_ch + c].rotation; };
        std::stable_sort(l.begin(), l.end(), [&](uint32_t x, uint32_t y) { return grp(x) < grp(y); });
        for (int r = 0; r < 3; r++) e->rot_count[f][r] = 0;
        for (uint32_t c : l) e->rot_count[f][grp(c)]++;
    }
    e->list_first = first_ch;
    e->list_n = n_ch;
    e->lists_dirty = false;
}

int iqd_accept_iq_device(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const void *iq_dev, size_t bytes_per_ch,
                         void *pcm_dev, void *pcm_count_dev, void *magnitude_dev, void *signal_present_dev)
{
    if (!range_ok(e, first_ch, n_ch) || !iq_dev || !pcm_dev) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    const uint32_t call_bb = call_block_bytes(e, first_ch, n_ch, bytes_per_ch);
    if (!call_bb) return e->fail(IQD_EINVAL, IQD_LEN_MSG, bytes_per_ch, e->block_bytes);
    const uint32_t call_bs = call_bb / 2;
    if (bytes_per_ch / 2 > 0x7fff0000ull) return e->fail(IQD_EINVAL, "bytes_per_ch too large");
    if (((uintptr_t)iq_dev & 15) != 0) return e->fail(IQD_EINVAL, "iq_dev must be 16-byte aligned");
    (void)hipSetDevice(e->device);
    hipStream_t s = e->stream;

    const uint32_t n_blocks = (uint32_t)(bytes_per_ch / call_bb);
    const uint32_t vlen = (uint32_t)(bytes_per_ch / 2);
    bool gated, any_agc;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->epoch_report_pending && ((volatile uint32_t *)e->h_epoch_report)[1] == e->epoch_report_seq) {   // (see wbfm_epoch_seq)
            if (((volatile uint32_t *)e->h_epoch_report)[0] == 0u)
                for (uint32_t c : e->epoch_report_channels)
                    if (e->wbfm_epoch_left[c] && e->wbfm_epoch_seq[c] <= e->epoch_report_seq) {
                        e->wbfm_epoch_left[c] = 0;
                        e->wbfm_epochs_live--;
                    }
            e->epoch_report_pending = false;
        }
        {
            int rc = upload_params(e);
            if (rc != IQD_OK) return rc;
        }
        e->accept_seq++;
        if (e->lists_dirty || e->list_first != first_ch || e->list_n != n_ch) {
            rebuild_lists(e, first_ch, n_ch);
            for (int f = 0; f <= FAM_COUNT; f++) {
                const auto &l = e->h_lists[f];
                if (l.empty()) continue;
                HIP_TRY(e, e->lists[f].ensure(l.size() * sizeof(uint32_t)));
                HIP_COPY(e, hipMemcpyAsync(e->lists[f].p, l.data(), l.size() * sizeof(uint32_t),
                                          hipMemcpyHostToDevice, s));
            }
            HIP_TRY(e, hipStreamSynchronize(s));
        }
        int rc = agc_sync(e);
        if (rc != IQD_OK) return rc;
        gated = e->any_gated;
        any_agc = e->any_agc;
    }
    if (e->demod_bypass) {   // {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData: nothing of the processor's squelch path runs or moves
        gated = any_agc = false;
        pcm_count_dev = magnitude_dev = signal_present_dev = nullptr;
    }
    const bool want_mag = !e->demod_bypass && (gated || any_agc || !(e->flags & IQD_F_NO_MAGNITUDE) || magnitude_dev);

    // the pre-pass of a gated call one call ahead, on its own stream (include/iqdemod.h: IQD_F_PREPASS_OVERLAP)
    const bool pre_overlap = gated && (e->flags & IQD_F_PREPASS_OVERLAP) && !e->trace_on && !e->in_host_path;
    if (pre_overlap && !e->pre_stream) {
        HIP_TRY(e, hipStreamCreateWithFlags(&e->pre_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_pre_done[k], hipEventDisableTiming));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_chain_done[k], hipEventDisableTiming));
        }
        HIP_TRY(e, hipEventCreateWithFlags(&e->ev_main_decisions, hipEventDisableTiming));
    }
    if (!pre_overlap) {   // the sums start at zero: by memset, unless the previous call's squelch pass left this many of them zero
        const void *before = e->mag_sums.p;
        HIP_TRY(e, e->mag_sums.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        if (e->mag_sums.p != before) e->mag_sums_zero = 0;
        if (want_mag && e->mag_sums_zero < (size_t)n_ch * n_blocks)
            HIP_COPY(e, hipMemsetAsync(e->mag_sums.p, 0, (size_t)n_ch * n_blocks * sizeof(uint32_t), s));
        e->mag_sums_zero = 0;   // from here on the call writes into them
    }

    SquelchLaunch q{};
    q.n_ch = n_ch; q.first_ch = first_ch; q.n_blocks = n_blocks; q.block_samples = call_bs;
    q.params = e->d_params;
    q.mag_sums = e->mag_sums.as<uint32_t>();
    q.tracker = e->d_tracker;
    q.magnitude = (uint32_t *)magnitude_dev;
    q.allowed = (uint8_t *)signal_present_dev;
    q.pcm_count = (uint32_t *)pcm_count_dev;
    q.agc_cfg = e->d_agc_cfg;
    q.agc = e->d_agc;
    q.any_agc = any_agc ? 1u : 0u;
    q.scan_cfg = e->d_scan_cfg;
    q.scan = e->d_scan;
    if (e->trace_on) {
        HIP_TRY(e, e->gain_trace.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        q.gain_trace = e->gain_trace.as<uint32_t>();
        HIP_TRY(e, e->freq_trace.ensure((size_t)n_ch * n_blocks * sizeof(unsigned long long)));
        q.freq_trace = e->freq_trace.as<unsigned long long>();
        e->trace_first = first_ch; e->trace_n = n_ch; e->trace_blocks = n_blocks;
    }

    const bool chain_gated = gated;   // the chain kernels walk each channel's open blocks (blk_lists, vlen_gated)
    DevBuf *gate_blk = &e->blk_lists, *gate_vlen = &e->vlen;
    int pre_set = -1;
    if (pre_overlap) {
        // magnitudes, decisions and open-block lists of THIS call on the pre-pass stream, which the previous call's pipelines
        // (main stream) do not hold up: it waits for the decision pass before it (its own stream order - or the main stream's,
        // where the previous call was not gated), and for the last reader of the buffer set it is about to overwrite
        hipStream_t ps = e->pre_stream;
        pre_set = e->gate_set ^= 1;
        DevBuf &sums = e->g_sums[pre_set];
        gate_blk = &e->g_blk[pre_set];
        gate_vlen = &e->g_vlen[pre_set];
        if (e->chain_pending[pre_set]) HIP_TRY(e, hipStreamWaitEvent(ps, e->ev_chain_done[pre_set], 0));
        if (e->decisions_on_main) {
            HIP_TRY(e, hipStreamWaitEvent(ps, e->ev_main_decisions, 0));
            e->decisions_on_main = false;
        }
        HIP_TRY(e, sums.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, gate_blk->ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, gate_vlen->ensure((size_t)n_ch * sizeof(uint32_t)));
        HIP_COPY(e, hipMemsetAsync(sums.p, 0, (size_t)n_ch * n_blocks * sizeof(uint32_t), ps));
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)iq_dev, bytes_per_ch, nullptr, n_ch, call_bs, n_blocks, sums.as<uint32_t>(), ps));
        q.mag_sums = sums.as<uint32_t>();
        q.blk_lists = gate_blk->as<uint32_t>();
        q.vlen_out = gate_vlen->as<uint32_t>();
        HIP_LAUNCH(e, launch_squelch(q, false, ps));
        HIP_TRY(e, hipEventRecord(e->ev_pre_done[pre_set], ps));
        HIP_TRY(e, hipStreamWaitEvent(s, e->ev_pre_done[pre_set], 0));   // the pipelines read the lists
    } else if (gated) {
        // pass 1: magnitudes of every block, then the squelch decisions and open-block lists
        if (e->pre_stream) HIP_TRY(e, hipStreamSynchronize(e->pre_stream));   // (a call of the overlapped kind before this one)
        HIP_TRY(e, e->blk_lists.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, e->vlen.ensure((size_t)n_ch * sizeof(uint32_t)));
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)iq_dev, bytes_per_ch, nullptr, n_ch, call_bs,
                                    n_blocks, e->mag_sums.as<uint32_t>(), s));
        q.blk_lists = e->blk_lists.as<uint32_t>();
        q.vlen_out = e->vlen.as<uint32_t>();
        // (Round 2 read one word back here - did any channel lose a block? - to let an all-open call take the streaming
        // kernels, which could not gate.  They can now (a virtual sample axis over each channel's open blocks), so the
        // call stays asynchronous whatever the squelch decides.)
        HIP_LAUNCH(e, launch_squelch(q, false, s));
    }
    if (!gated && e->pre_stream) HIP_TRY(e, hipStreamWaitEvent(s, e->ev_pre_done[e->gate_set], 0));   // (the tracker / AGC state this call's squelch pass continues from)

    ChainLaunch base{};
    base.iq = (const uint8_t *)iq_dev;
    base.ch_stride_bytes = bytes_per_ch;
    base.first_ch = first_ch;
    base.vlen = vlen;
    base.vlen_gated = chain_gated ? gate_vlen->as<uint32_t>() : nullptr;
    base.blk_lists = chain_gated ? gate_blk->as<uint32_t>() : nullptr;
    base.n_blocks = n_blocks;
    base.block_samples = call_bs;
    base.block_magic = block_magic(call_bs);
    base.tails = e->d_tails;
    base.params = e->d_params;
    base.wbfm_carry = e->d_wcarry;
    base.dc_carry = e->d_dc;
    base.epochs = e->d_epochs;
    base.atan_lut = e->d_atan;
    base.fm_lut = e->d_fmlut;
    base.pcm = (int16_t *)pcm_dev;
    base.pcm_stride = bytes_per_ch / 64;
    base.mag_sums = e->mag_sums.as<uint32_t>();
    base.counters = e->d_counters;
    base.stamps = e->d_stamps;
    base.n_ch_call = n_ch;

    const bool fused_mag = want_mag && !gated;
    bool timed = false;
    std::pair<hipEvent_t, hipEvent_t> evp{nullptr, nullptr};
    // More than one demodulator family in the call: the (often small) per-family launches run side by side on up
    // to three lanes - the engine's stream and two side streams of other priorities between a fork and a join event
    // (the runtime multiplexes equal-priority streams onto a handful of hardware queues; streams that land on one
    // queue run one after the other) - instead of each draining the GPU in turn.  Families go to the least loaded
    // lane, longest first.
    int n_fams = 0;
    for (int f = 0; f < FAM_COUNT; f++) n_fams += e->h_lists[f].empty() ? 0 : 1;
    const bool forked = n_fams > 1;
    hipStream_t const s_main = s;
    if (forked) {
        if (!e->fam_fork) {
            int lo = 0, hi = 0;   // numerically lower = higher priority
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(e, hipEventCreateWithFlags(&e->fam_fork, hipEventDisableTiming));
            for (int k = 0; k < 3; k++) {
                HIP_TRY(e, hipStreamCreateWithPriority(&e->fam_stream[k], hipStreamNonBlocking, k == 0 ? hi : lo));
                HIP_TRY(e, hipEventCreateWithFlags(&e->fam_join[k], hipEventDisableTiming));
            }
        }
        // before the fork.  One buffer per family: short rows are written time-major by the tile kernels and channel-major by the
        // streaming pipelines, and the two families of a call may take different paths (round 4's fuzzer: an AM family that
        // streamed beside an SSB family on the tile kernels overwrote its detector stream)
        if (!e->h_lists[FAM_AM].empty()) HIP_TRY(e, e->base8k.ensure((size_t)n_ch * base.pcm_stride * sizeof(int32_t)));
        if (!e->h_lists[FAM_SSB].empty()) HIP_TRY(e, e->base8k2.ensure((size_t)n_ch * base.pcm_stride * sizeof(int32_t)));
    }
    // relative cost per channel-sample: AM, FM, WBFM, SSB (the families' workgroups side by side: 214 / 228 / 227 / 259 us on
    // 32 / 56 / 96 / 56 CUs for 819 / 819 / 820 / 1638 channels x 2^16, profiles/r3_mixed_4096_kernel_stats.csv)
    const float *weight = e->fam_weight;
    int order[FAM_COUNT] = {0, 1, 2, 3};
    float cost[FAM_COUNT];
    for (int f = 0; f < FAM_COUNT; f++) cost[f] = weight[f] * (float)e->h_lists[f].size();
    std::sort(order, order + FAM_COUNT, [&](int x, int y) { return cost[x] > cost[y]; });
    // Several families side by side: each one's persistent workgroups take a share of the CUs in proportion to its
    // estimated cost, so that the families' streaming kernels run at the same time (a CU's LDS holds one such
    // workgroup) on longer segments - less lead-in overhead, which is what small families pay most for.  (Mixed
    // configuration, 4096 channels x 2^16: 0.45 ms per step with every family on all CUs in turn, 0.39 with shares.)
    uint32_t fam_share[FAM_COUNT];
    bool shares_on = false;   // the families of this call run their streaming kernels side by side, each on a share of the CUs
    bool fused = false;       // ... as ranges of one launch's workgroups
    {
        float total = 0.f;
        for (int f = 0; f < FAM_COUNT; f++) total += cost[f];
        for (int f = 0; f < FAM_COUNT; f++) fam_share[f] = e->n_cus;
        // (only when every family of the call will take its streaming kernel - here that means: brings enough samples
        // for ITS share of the CUs; tile kernels know nothing of shares, and a streaming kernel held to its share beside
        // them lost 12-16 % at 2500-3000 mixed channels)
        auto every_family_streams = [&](int arrangement) {   // 1: as ranges of one launch, 2: as kernels on streams (stream_min_seg)
            bool all = !(e->flags & IQD_F_WBFM_TILES) && vlen % 128 == 0 && e->env_path >= 0;
            for (int f = 0; f < FAM_COUNT && all; f++) {
                const uint64_t n_f = e->h_lists[f].size();
                if (!n_f) continue;
                const bool forced = (e->flags & IQD_F_WBFM_STREAM) != 0;
                const float due = total > 0.f ? (float)(e->n_cus - 16) * cost[f] / total : (float)e->n_cus;
                if (!forced && (float)((uint64_t)vlen * n_f) < due * (float)(ST_SEGS * stream_min_seg(e, f, vlen, arrangement))) all = false;
                if ((f == FAM_AM || f == FAM_SSB) && vlen / 32 < e->env_am_stream_min) all = false;
            }
            return all;
        };
        const bool all_stream = every_family_streams(1);
        shares_on = forked && all_stream && total > 0.f && e->n_cus >= 64 && !e->env_full_grid;
        // ONE launch for all of them (iqd_stream_mixed.hip) when each family can take its streaming pipeline in the plain
        // instantiation: the WBFM channels of one rotation selector and without a gain change in reach of a lead-in,
        // gains below the "integer indefinite" bounds, AM / SSB rows that the one-wave DC pass takes
        fused = shares_on && !e->env_mixed_forked && e->env_path == 0 && !e->env_stream_wgs &&
                !(e->flags & IQD_F_WBFM_STREAM);
        if (fused && !e->h_lists[FAM_WBFM].empty()) {
            const auto &l = e->h_lists[FAM_WBFM];
            const int rot0 = e->h_params[first_ch + l[0]].rotation;
            fused = e->stream_ok;
            for (uint32_t c : l) {
                fused = fused && e->h_params[first_ch + c].rotation == rot0 && e->wbfm_kmax[first_ch + c] * 3.1730f < 2147483648.0f;
                fused = fused && !(e->wbfm_epochs_live && e->wbfm_epoch_left[first_ch + c] != 0);
            }
        }
        if (fused)
            for (uint32_t c : e->h_lists[FAM_FM]) fused = fused && e->fm_kmax[first_ch + c] * 6.35f < 2147483648.0f;
        if (fused && (!e->h_lists[FAM_AM].empty() || !e->h_lists[FAM_SSB].empty()))
            fused = vlen / 32 >= e->env_am_stream_min && (bytes_per_ch / 64 + DC_TILE - 1) / DC_TILE < 2;
        if (shares_on && !fused) shares_on = every_family_streams(2);   // (the kernels-on-streams arrangement pays from larger calls only)
        if (shares_on && fused) {
            FusedFamily ff[FAM_COUNT];
            for (int f = 0; f < FAM_COUNT; f++) {
                for (int r = 0; r < 3; r++) ff[f].rot_count[r] = e->h_lists[f].empty() ? 0u : e->rot_count[f][r];
                ff[f].halo = f == FAM_WBFM ? (uint32_t)ST_HALO : f == FAM_FM ? (uint32_t)D4_HALO_FM : f == FAM_AM ? (uint32_t)D4_HALO_AM : (uint32_t)D4_HALO_SSB;
                ff[f].granule = f == FAM_WBFM ? e->env_stream_gran : e->env_d4_gran;
                ff[f].ns_per_sample = e->fam_ns[f];
            }
            if (e->env_shares_by_cost || !plan_fused_by_time((uint32_t)vlen, FAM_COUNT, ff, e->n_cus, fam_share))
                plan_fused_shares(cost, FAM_COUNT, e->n_cus, fam_share);
        }
        else if (shares_on && !plan_family_shares(cost, FAM_COUNT, e->n_cus, fam_share)) shares_on = false;
        if (!shares_on) fused = false;
    }
    MixedStreamArgs mix{};
    uint32_t mix_wgs = 0;
    bool epoch_report_now = false;   // this call's WBFM tail updates report whether a gain change is still in reach (h_epoch_report)
    if (forked && !fused) HIP_TRY(e, hipEventRecord(e->fam_fork, s_main));
    float lane_load[4] = {0.f, 0.f, 0.f, 0.f};   // 0: the engine's stream, 1 to 3: the side streams
    bool lane_used[4] = {false, false, false, false};
    size_t (*dcr_layout)[2] = e->dcr_layout;
    ChainLaunch tail_a{};
    int tail_f = 0;
    bool tail_pending = false;
    for (int oi = 0; oi < FAM_COUNT; oi++) {
        const int f = order[oi];
        const uint32_t n_list = (uint32_t)e->h_lists[f].size();
        if (!n_list) continue;
        int lane = 0;
        if (forked && !fused) {
            for (int k = 1; k < 4; k++)
                if (lane_load[k] < lane_load[lane]) lane = k;
            lane_load[lane] += cost[f];
        }
        s = lane == 0 ? s_main : e->fam_stream[lane - 1];
        if (lane != 0 && !lane_used[lane]) HIP_TRY(e, hipStreamWaitEvent(s, e->fam_fork, 0));
        lane_used[lane] = true;
        uint32_t fam_wgs = fam_share[f];
        if (e->env_stream_wgs) fam_wgs = e->env_stream_wgs;   // (experiments)
        ChainLaunch a = base;
        a.ch_list = e->lists[f].as<uint32_t>();
        a.n_list = n_list;
        if (f == FAM_WBFM && chain_gated && e->wbfm_epochs_live && !e->epoch_report_pending) {
            bool any = false;
            for (uint32_t c : e->h_lists[FAM_WBFM]) any = any || e->wbfm_epoch_left[first_ch + c] != 0;
            if (any) {
                if (!e->h_epoch_report) HIP_TRY(e, hipHostMalloc((void **)&e->h_epoch_report, 2 * sizeof(uint32_t), hipHostMallocDefault));
                ((volatile uint32_t *)e->h_epoch_report)[0] = 0u;
                a.epoch_report = e->h_epoch_report;
                e->epoch_report_channels.clear();
                for (uint32_t c : e->h_lists[FAM_WBFM]) e->epoch_report_channels.push_back(first_ch + c);
                epoch_report_now = true;
            }
        }
        // workgroups a CU holds at once: WBFM 3 (LDS), the others 4 (registers)
        const TilePlan plan = f == FAM_WBFM ? plan_tiles(vlen, n_list, WBFM_CHUNK, COLD_HALO, 3 * e->n_cus, e->env_plan_chunks)
                                            : plan_tiles(vlen, n_list, CH_CHUNK, FIR_HALO, 4 * e->n_cus, e->env_plan_chunks);
        a.tile_len = plan.tile_len;
        a.tiles_per_ch = plan.tiles_per_ch;
        // WBFM: the streaming pipeline (iqd_stream.hip) when the launch can fill the chip with it and nothing it does
        // not handle is in play: squelch-gated rows, channels with different rotation selectors, a K so large that
        // (int16)y can hit the "integer indefinite" value.  Results are identical either way.
        bool use_stream = false;
        int stream_rot = 0;
        bool stream_grouped = false;   // channels of several rotation selectors: segment ids grouped by selector (StreamArgs::grouped)
        uint32_t st_group_start[4] = {0, 0, 0, 0}, st_group_li0[3] = {0, 0, 0}, st_group_nseg[3] = {0, 0, 0};
        if (f == FAM_WBFM && e->stream_ok && !(e->flags & IQD_F_WBFM_TILES) && vlen % 128 == 0) {
            const auto &l = e->h_lists[FAM_WBFM];
            stream_rot = e->h_params[first_ch + l[0]].rotation;
            bool ok = true;
            for (uint32_t c : l) {
                const ChanParams &p = e->h_params[first_ch + c];
                stream_grouped = stream_grouped || p.rotation != stream_rot;
                ok = ok && e->wbfm_kmax[first_ch + c] * 3.1730f < 2147483648.0f;
            }
            if (fused && stream_grouped) ok = false;   // (the one-launch arrangement holds the single-selector instantiations only)
            int want = 0;   // 0 auto, 1 stream, -1 tiles
            if (e->flags & IQD_F_WBFM_STREAM) want = 1;
            if (e->env_path) want = e->env_path;
            const uint64_t work = (uint64_t)vlen * n_list;
            if (ok && want >= 0 && (want > 0 || work >= (uint64_t)(shares_on ? fam_wgs : e->n_cus) * ST_SEGS * stream_min_seg(e, f, vlen, 0) || shares_on)) {
                for (uint32_t spare = 0;; spare += 48) {   // (the groups' padding may push an exact fit into a second round)
                    const TilePlan sp = plan_stream(vlen, n_list, fam_wgs * ST_SEGS - (stream_grouped ? spare : 0u), e->env_stream_gran);
                    a.tile_len = sp.tile_len;
                    a.tiles_per_ch = sp.tiles_per_ch;
                    if (!stream_grouped) break;
                    uint32_t at = 0, li0 = 0;
                    for (int r = 0; r < 3; r++) {   // the channel list is sorted +Fs/4, none, -Fs/4 (rebuild_lists)
                        st_group_start[r] = at;
                        st_group_li0[r] = li0;
                        st_group_nseg[r] = e->rot_count[FAM_WBFM][r] * a.tiles_per_ch;
                        at += (st_group_nseg[r] + 15u) / 16u * 16u;
                        li0 += e->rot_count[FAM_WBFM][r];
                    }
                    st_group_start[3] = at;
                    if (at <= fam_wgs * ST_SEGS || spare >= 48 || n_list * 1u >= fam_wgs * ST_SEGS) break;
                }
                use_stream = true;
            }
        }
        // FM / AM / SSB: the streaming pipelines of iqd_stream2.hip under the same conditions (channels of different
        // rotation selectors are fine here: the list is sorted by selector and the groups are padded)
        bool use_d4 = false;
        uint32_t d4_wgs = fam_wgs;
        D4Args d4 = e->d4_args;
        if (f != FAM_WBFM && !(e->flags & IQD_F_WBFM_TILES) && vlen % 128 == 0 && (f == FAM_FM || vlen / 32 >= e->env_am_stream_min)) {
            bool ok = true;
            if (f == FAM_FM)
                for (uint32_t c : e->h_lists[f]) ok = ok && e->fm_kmax[first_ch + c] * 6.35f < 2147483648.0f;
            int want = 0;
            if (e->flags & IQD_F_WBFM_STREAM) want = 1;
            if (e->env_path) want = e->env_path;
            const uint64_t work = (uint64_t)vlen * n_list;
            if (ok && want >= 0 && (want > 0 || work >= (uint64_t)(shares_on ? fam_wgs : e->n_cus) * ST_SEGS * stream_min_seg(e, f, vlen, 0) || shares_on)) {
                d4_wgs = fam_wgs;
                for (uint32_t spare = 0;; spare += 48) {   // (the rotation groups' padding may push an exact fit into a second round)
                    const TilePlan sp = plan_stream(vlen, n_list, d4_wgs * ST_SEGS - spare, e->env_d4_gran);   // (these pipelines store 8 or 16 bytes per 128 samples: no wide stores to keep whole)
                    a.tile_len = sp.tile_len;
                    a.tiles_per_ch = sp.tiles_per_ch;
                    uint32_t at = 0, li0 = 0;
                    for (int r = 0; r < 3; r++) {
                        d4.group_start[r] = at;
                        d4.group_li0[r] = li0;
                        d4.group_nseg[r] = e->rot_count[f][r] * a.tiles_per_ch;
                        at += (d4.group_nseg[r] + 15u) / 16u * 16u;
                        li0 += e->rot_count[f][r];
                    }
                    d4.group_start[3] = at;
                    if (at <= d4_wgs * ST_SEGS || spare >= 48 || n_list * 1u >= d4_wgs * ST_SEGS) break;
                }
                d4.amat = e->d_amat4 + (size_t)(f == FAM_FM ? 0 : 3) * 4 * 64 * 4;
                d4.fm_lut = e->d_fmlut;
                d4.halo = f == FAM_FM ? D4_HALO_FM : (f == FAM_AM ? D4_HALO_AM : D4_HALO_SSB);
                use_d4 = true;
            }
        }
        if (fused && !((f == FAM_WBFM && use_stream) || (f != FAM_WBFM && use_d4))) return e->fail(IQD_EINVAL, "a family of a fused launch fell off its streaming pipeline");
        if (e->profiling && !timed && !fused) {
            if (e->ev_free_pairs.empty()) {
                hipEvent_t a0, a1;
                HIP_TRY(e, hipEventCreate(&a0));
                HIP_TRY(e, hipEventCreate(&a1));
                e->ev_free_pairs.emplace_back(a0, a1);
            }
            evp = e->ev_free_pairs.back();
            e->ev_free_pairs.pop_back();
            HIP_TRY(e, hipEventRecord(evp.first, s));
        }
        if (f == FAM_WBFM) {
            HIP_TRY(e, e->records.ensure((size_t)n_list * a.tiles_per_ch * sizeof(WbfmRecord)));
            a.records = e->records.as<WbfmRecord>();
            if (e->repair_flags.cap < n_list * sizeof(uint32_t)) {   // zero between calls: the repair kernel clears what it used
                HIP_TRY(e, e->repair_flags.ensure(n_list * sizeof(uint32_t)));
                HIP_COPY(e, hipMemsetAsync(e->repair_flags.p, 0, e->repair_flags.cap, s));
            }
            a.repair_flags = e->repair_flags.as<uint32_t>();
            if (use_stream) {
                StreamArgs sa = e->stream_args;
                sa.amat = e->d_amat[stream_rot + 1];
                sa.half_lut = e->d_half_lut;
                sa.n_segments = n_list * a.tiles_per_ch;
                sa.grouped = stream_grouped ? 1u : 0u;
                for (int r = 0; r < 3; r++) {
                    sa.group_start[r] = st_group_start[r];
                    sa.group_li0[r] = st_group_li0[r];
                    sa.group_nseg[r] = st_group_nseg[r];
                    sa.amat3[r] = e->d_amat[2 - r];          // d_amat[] is indexed by selector + 1; the groups run +1, 0, -1
                }
                sa.group_start[3] = st_group_start[3];
                const uint32_t wgs_needed = ((stream_grouped ? sa.group_start[3] : sa.n_segments) + ST_SEGS - 1) / ST_SEGS;
                const uint32_t grid = wgs_needed < fam_wgs ? wgs_needed : fam_wgs;
                sa.rounds = (wgs_needed + grid - 1) / grid;
                HIP_TRY(e, e->stream_hist.ensure((size_t)sa.n_segments * sizeof(StHist)));
                sa.hist = e->stream_hist.as<StHist>();
                a.verify_at_end = chain_gated ? 2u : 1u;   // (2: the hand-offs are counted on the device - how many tiles a channel has depends on its squelch)
                bool epochs_live = false;
                if (e->wbfm_epochs_live)
                    for (uint32_t c : e->h_lists[FAM_WBFM]) epochs_live = epochs_live || e->wbfm_epoch_left[first_ch + c] != 0;
                if (fused) {   // a range of the one launch's workgroups; the fix-up rides in the launch behind it
                    a.wg_first = mix_wgs;
                    a.wg_count = grid;
                    mix_wgs += grid;
                    mix.a[f] = a;
                    mix.sa = sa;
                    mix.wbfm_rot = stream_rot;
                } else {
                    HIP_LAUNCH(e, launch_wbfm_stream(a, sa, stream_rot, fused_mag, epochs_live, grid, s));
                    HIP_LAUNCH(e, launch_wbfm_stream_fixup(a, sa, s));
                }
                if (!chain_gated) e->stream_handoffs += (uint64_t)n_list * ((vlen + a.tile_len - 1) / a.tile_len - 1);
                e->stats.stream_launches++;
            } else {
                HIP_LAUNCH(e, launch_wbfm(a, chain_gated, fused_mag, n_list * a.tiles_per_ch, s));
            }
        } else if (f == FAM_FM) {
            if (use_d4) {
                const uint32_t wgs_needed = (d4.group_start[3] + ST_SEGS - 1) / ST_SEGS;
                const uint32_t grid = wgs_needed < d4_wgs ? wgs_needed : d4_wgs;
                d4.rounds = (wgs_needed + grid - 1) / grid;
                if (fused) {
                    a.wg_first = mix_wgs;
                    a.wg_count = grid;
                    mix_wgs += grid;
                    mix.a[f] = a;
                    mix.d4[f] = d4;
                } else {
                    HIP_LAUNCH(e, launch_d4_stream(a, d4, D4_FM, fused_mag, grid, s));
                }
                e->stats.stream_launches++;
            } else {
                HIP_LAUNCH(e, launch_fm(a, chain_gated, fused_mag, n_list * a.tiles_per_ch, s));
            }
        } else if (use_d4) {
            DevBuf &b8 = f == FAM_SSB ? e->base8k2 : e->base8k;
            HIP_TRY(e, b8.ensure((size_t)n_ch * base.pcm_stride * sizeof(int32_t)));
            a.base8k = b8.as<int32_t>();
            a.base_stride_ch = base.pcm_stride;   // channel-major (rows longer than 512 PCM samples)
            a.base_stride_t = 1;
            a.dc_tiles = (uint32_t)((base.pcm_stride + DC_TILE - 1) / DC_TILE);
            DevBuf &dcr = f == FAM_SSB ? e->dc_records2 : e->dc_records;
            {
                const size_t rec_bytes = (size_t)n_list * a.dc_tiles * sizeof(DcRecord);
                const bool grown = dcr.cap < rec_bytes + n_list * sizeof(uint32_t);
                HIP_TRY(e, dcr.ensure(rec_bytes + n_list * sizeof(uint32_t)));
                a.dc_records = dcr.p;
                if (grown || dcr_layout[f == FAM_SSB][0] != rec_bytes || dcr_layout[f == FAM_SSB][1] < n_list) {
                    HIP_COPY(e, hipMemsetAsync((char *)dcr.p + rec_bytes, 0, n_list * sizeof(uint32_t), s));
                    dcr_layout[f == FAM_SSB][0] = rec_bytes;
                    dcr_layout[f == FAM_SSB][1] = n_list;
                }
            }
            const uint32_t wgs_needed = (d4.group_start[3] + ST_SEGS - 1) / ST_SEGS;
            const uint32_t grid = wgs_needed < d4_wgs ? wgs_needed : d4_wgs;
            d4.rounds = (wgs_needed + grid - 1) / grid;
            if (fused) {
                a.wg_first = mix_wgs;
                a.wg_count = grid;
                mix_wgs += grid;
                mix.a[f] = a;
                mix.d4[f] = d4;
            } else {
                HIP_LAUNCH(e, launch_d4_stream(a, d4, f == FAM_AM ? D4_AM : D4_SSB, fused_mag, grid, s));
                HIP_LAUNCH(e, launch_am_dc(a, f, s, true));   // (the pipeline wrote the detector stream channel-major)
            }
            e->stats.stream_launches++;
        } else {
            DevBuf &b8 = f == FAM_SSB ? e->base8k2 : e->base8k;
            HIP_TRY(e, b8.ensure((size_t)n_ch * base.pcm_stride * sizeof(int32_t)));
            a.base8k = b8.as<int32_t>();
            a.dc_tiles = (uint32_t)((base.pcm_stride + DC_TILE - 1) / DC_TILE);
            DevBuf &dcr = f == FAM_SSB ? e->dc_records2 : e->dc_records;   // AM and SSB may run side by side
            {   // records, then one redo flag per channel: zero between calls (dc_redo_kernel clears what it used)
                const size_t rec_bytes = (size_t)n_list * a.dc_tiles * sizeof(DcRecord);
                const bool grown = dcr.cap < rec_bytes + n_list * sizeof(uint32_t);
                HIP_TRY(e, dcr.ensure(rec_bytes + n_list * sizeof(uint32_t)));
                a.dc_records = dcr.p;
                // the flags sit behind the records, whose extent changes with the call: clear them whenever it may have
                // (same offset but more channels than last time: the new flags lie over old record bytes)
                if (grown || dcr_layout[f == FAM_SSB][0] != rec_bytes || dcr_layout[f == FAM_SSB][1] < n_list) {
                    HIP_COPY(e, hipMemsetAsync((char *)dcr.p + rec_bytes, 0, n_list * sizeof(uint32_t), s));
                    dcr_layout[f == FAM_SSB][0] = rec_bytes;
                    dcr_layout[f == FAM_SSB][1] = n_list;
                }
            }
            HIP_LAUNCH(e, launch_am(a, f, chain_gated, fused_mag, n_list * a.tiles_per_ch, s));
        }
        if (e->profiling && !timed && !fused) {
            HIP_TRY(e, hipEventRecord(evp.second, s));
            e->ev_pending.push_back(evp);
            timed = true;
        }
        e->stats.kernel_launches++;
        if (fused) continue;   // (one launch for all the families and one for what follows them, behind this loop)
        if (f == FAM_WBFM && !chain_gated && e->wbfm_epochs_live)   // every channel of the family has consumed vlen samples
            for (uint32_t c : e->h_lists[FAM_WBFM]) {
                uint32_t &left = e->wbfm_epoch_left[first_ch + c];
                if (!left) continue;
                left = left > vlen ? left - vlen : 0u;
                if (!left) e->wbfm_epochs_live--;
            }
        const bool rides_with_squelch = !forked && !gated && !e->demod_bypass && (want_mag || pcm_count_dev || signal_present_dev || e->trace_on);
        if (f == FAM_WBFM) {
            // hand-off verification (streaming launches: done by their fix-up kernel), repair of what it flags
            // (normally an immediate exit), then state commit + tail - in the squelch launch below when the call has
            // no other family
            if (!use_stream) HIP_LAUNCH(e, launch_wbfm_verify(a, s));
            if (rides_with_squelch) {
                tail_a = a;
                tail_f = f;
                tail_pending = true;
            } else {
                HIP_LAUNCH(e, launch_wbfm_repair(a, chain_gated, s));   // (ends with the channels' state commit and tail update)
            }
        } else if (rides_with_squelch) {
            tail_a = a;        // the only family of the call: its tail update rides in the squelch launch below
            tail_f = f;
            tail_pending = true;
        } else {
            HIP_LAUNCH(e, launch_tail_update(a, f, s));
        }
    }
    s = s_main;
    if (fused) {
        if (e->profiling) {
            if (e->ev_free_pairs.empty()) {
                hipEvent_t a0, a1;
                HIP_TRY(e, hipEventCreate(&a0));
                HIP_TRY(e, hipEventCreate(&a1));
                e->ev_free_pairs.emplace_back(a0, a1);
            }
            evp = e->ev_free_pairs.back();
            e->ev_free_pairs.pop_back();
            HIP_TRY(e, hipEventRecord(evp.first, s));
        }
        HIP_LAUNCH(e, launch_mixed_stream(mix, fused_mag, chain_gated, mix_wgs, s));
        e->stats.mixed_launches++;
        MixedTailArgs mt{};
        for (int f = 0; f < FAM_COUNT; f++) mt.a[f] = mix.a[f];
        mt.sa = mix.sa;
        HIP_LAUNCH(e, launch_mixed_tail(mt, s));
        if (e->profiling) {   // (the timed region: the pipelines AND their followers - fix-up, DC passes, tails)
            HIP_TRY(e, hipEventRecord(evp.second, s));
            e->ev_pending.push_back(evp);
        }
        if (mix.a[FAM_WBFM].wg_count) {   // repair check, state commit and tail of the WBFM channels: in the squelch launch below
            tail_a = mix.a[FAM_WBFM];
            tail_f = FAM_WBFM;
            tail_pending = true;
        }
    }
    for (int k = 1; k < 4; k++)
        if (lane_used[k]) {
            HIP_TRY(e, hipEventRecord(e->fam_join[k - 1], e->fam_stream[k - 1]));
            HIP_TRY(e, hipStreamWaitEvent(s_main, e->fam_join[k - 1], 0));
        }

    // channels in mode None still report their magnitudes
    if (fused_mag && !e->h_lists[FAM_COUNT].empty())
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)iq_dev, bytes_per_ch, e->lists[FAM_COUNT].as<uint32_t>(),
                                    (uint32_t)e->h_lists[FAM_COUNT].size(), call_bs, n_blocks,
                                    e->mag_sums.as<uint32_t>(), s));
    if (!gated && !e->demod_bypass && (want_mag || pcm_count_dev || signal_present_dev || e->trace_on)) {
        q.zero_sums_after = any_agc ? 0u : 1u;   // (a running AGC reads them again in the tracking pass)
        HIP_LAUNCH(e, launch_squelch(q, true, s, tail_pending ? &tail_a : nullptr, tail_f));
        tail_pending = false;
        if (q.zero_sums_after) e->mag_sums_zero = (size_t)n_ch * n_blocks;
    }
    if (tail_pending) {   // (no squelch launch to ride in)
        if (tail_f == FAM_WBFM) HIP_LAUNCH(e, launch_wbfm_repair(tail_a, chain_gated, s));
        else HIP_LAUNCH(e, launch_tail_update(tail_a, tail_f, s));
    }

    if (pre_set >= 0) {   // this call's pipelines are the last readers of its buffer set
        HIP_TRY(e, hipEventRecord(e->ev_chain_done[pre_set], s_main));
        e->chain_pending[pre_set] = true;
    } else if (!gated && e->pre_stream && !e->demod_bypass) {   // a squelch pass on the main stream: the next pre-pass continues from it
        HIP_TRY(e, hipEventRecord(e->ev_main_decisions, s_main));
        e->decisions_on_main = true;
    }
    if (epoch_report_now) {   // behind everything this call queued (the side streams have joined)
        HIP_LAUNCH(e, launch_write_word(e->h_epoch_report + 1, e->accept_seq, s_main));
        e->epoch_report_pending = true;
        e->epoch_report_seq = e->accept_seq;
    }
    e->stats.accepts++;
    e->stats.samples += (uint64_t)vlen * n_ch;
    return IQD_OK;
}

// Large host-pointer accepts are cut into slices of about SLICE_BYTES that go through two sets of device
// staging buffers: the upload of slice k+1 (copy stream) overlaps the kernels and the download of slice k
// (engine stream).  A slice is either a time range of every channel of the call (few long rows) or a range of
// whole rows (many channels).  Per-channel state is carried from slice to slice exactly as from one accept call
// to the next, so the result equals the unsliced call; squelch-gated rows are compacted on the host at the end.
static const size_t SLICE_BYTES = (size_t)32 << 20;
static const size_t SMALL_CALL_BYTES = (size_t)256 << 10;   // host-pointer calls up to this size run straight out of page-locked host memory

static int accept_sliced(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const uint8_t *iq, size_t bytes_per_ch, uint32_t call_bb,
                         int16_t *pcm, uint32_t *pcm_count, uint32_t *magnitude, uint8_t *signal_present)
{
    hipStream_t s = e->stream;
    if (!e->copy_stream) {
        HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_in[b], hipEventDisableTiming));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_free[b], hipEventDisableTiming));
        }
    }
    const size_t bb = call_bb;   // the block size in force for THIS call: one short block has bb = bytes_per_ch
    const size_t row_blocks = bytes_per_ch / bb;
    // slice shape: sc channels x st bytes of each row
    size_t sc, st;
    if ((size_t)n_ch * bb <= SLICE_BYTES && row_blocks > 1) {   // time slices of all channels
        sc = n_ch;
        st = std::max<size_t>(1, SLICE_BYTES / ((size_t)n_ch * bb)) * bb;
        if (st > bytes_per_ch) st = bytes_per_ch;
    } else {                                                    // ranges of whole rows
        st = bytes_per_ch;
        sc = std::max<size_t>(1, SLICE_BYTES / bytes_per_ch);
        if (sc > n_ch) sc = n_ch;
    }
    const size_t n_tslices = (bytes_per_ch + st - 1) / st, n_cslices = (n_ch + sc - 1) / sc;
    const size_t n_slices = n_tslices * n_cslices;
    const size_t st_blocks = st / bb;
    if (e->h_slice_counts_cap < 2 * sc) {
        if (e->h_slice_counts) (void)hipHostFree(e->h_slice_counts);
        e->h_slice_counts = nullptr;
        e->h_slice_counts_cap = 0;
        HIP_TRY(e, hipHostMalloc((void **)&e->h_slice_counts, 2 * sc * sizeof(uint32_t), hipHostMallocDefault));
        e->h_slice_counts_cap = 2 * sc;
    }
    for (int b = 0; b < 2; b++) {
        HIP_TRY(e, e->sl_iq[b].ensure(sc * st));
        HIP_TRY(e, e->sl_pcm[b].ensure(sc * (st / 64) * sizeof(int16_t)));
        HIP_TRY(e, e->sl_count[b].ensure(sc * sizeof(uint32_t)));
        HIP_TRY(e, e->sl_mag[b].ensure(sc * st_blocks * sizeof(uint32_t)));
        HIP_TRY(e, e->sl_allowed[b].ensure(sc * st_blocks));
    }
    std::vector<uint32_t> filled(n_ch, 0);   // PCM samples already in place at the front of each row
    const size_t row_pcm = bytes_per_ch / 64;

    struct Slice { size_t c0, nc, t0, tb; };
    auto slice_at = [&](size_t k) {
        Slice x;
        const size_t ci = k / n_tslices, ti = k % n_tslices;
        x.c0 = ci * sc; x.nc = std::min(sc, (size_t)n_ch - x.c0);
        x.t0 = ti * st; x.tb = std::min(st, bytes_per_ch - x.t0);
        return x;
    };
    auto upload = [&](size_t k) -> int {
        const Slice x = slice_at(k);
        const int b = (int)(k & 1);
        if (k >= 2) HIP_TRY(e, hipStreamWaitEvent(e->copy_stream, e->ev_free[b], 0));
        HIP_COPY(e, hipMemcpy2DAsync(e->sl_iq[b].p, x.tb, iq + x.c0 * bytes_per_ch + x.t0, bytes_per_ch, x.tb, x.nc,
                                    hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(e, hipEventRecord(e->ev_in[b], e->copy_stream));
        return IQD_OK;
    };
    // what slice k-1 left in the pinned count buffer: compact its PCM behind what the rows already hold
    auto settle = [&](size_t k) {
        const Slice x = slice_at(k);
        const uint32_t *cnt = e->h_slice_counts + (k & 1) * sc;
        for (size_t c = 0; c < x.nc; c++) {
            int16_t *row = pcm + (x.c0 + c) * row_pcm;
            uint32_t &have = filled[x.c0 + c];
            const size_t landed = x.t0 / 64;
            if (cnt[c] && have != landed) memmove(row + have, row + landed, cnt[c] * sizeof(int16_t));
            have += cnt[c];
        }
    };

    int rc = upload(0);
    if (rc != IQD_OK) return rc;
    for (size_t k = 0; k < n_slices; k++) {
        const Slice x = slice_at(k);
        const int b = (int)(k & 1);
        if (k + 1 < n_slices && (rc = upload(k + 1)) != IQD_OK) return rc;
        HIP_TRY(e, hipStreamWaitEvent(s, e->ev_in[b], 0));
        HIP_COPY(e, hipMemsetAsync(e->sl_pcm[b].p, 0, x.nc * (x.tb / 64) * sizeof(int16_t), s));
        rc = iqd_accept_iq_device(e, first_ch + (uint32_t)x.c0, (uint32_t)x.nc, e->sl_iq[b].p, x.tb, e->sl_pcm[b].p,
                                  e->sl_count[b].p, magnitude ? e->sl_mag[b].p : nullptr,
                                  signal_present ? e->sl_allowed[b].p : nullptr);
        if (rc != IQD_OK) return rc;
        if (e->demod_bypass)   // (no squelch pass, nothing dropped: every row of the slice is full)
            HIP_COPY(e, hipMemsetD32Async((hipDeviceptr_t)e->sl_count[b].p, (int)(x.tb / 64), x.nc, s));
        if (k >= 1) {   // the previous slice's downloads are complete once its count copy is
            HIP_TRY(e, hipEventSynchronize(e->ev_free[b ^ 1]));
            settle(k - 1);
        }
        const size_t tpcm = x.tb / 64, tblk = x.tb / bb;
        HIP_COPY(e, hipMemcpy2DAsync(pcm + x.c0 * row_pcm + x.t0 / 64, row_pcm * sizeof(int16_t), e->sl_pcm[b].p,
                                    tpcm * sizeof(int16_t), tpcm * sizeof(int16_t), x.nc, hipMemcpyDeviceToHost, s));
        if (magnitude)
            HIP_COPY(e, hipMemcpy2DAsync(magnitude + x.c0 * row_blocks + x.t0 / bb, row_blocks * sizeof(uint32_t),
                                        e->sl_mag[b].p, tblk * sizeof(uint32_t), tblk * sizeof(uint32_t), x.nc,
                                        hipMemcpyDeviceToHost, s));
        if (signal_present)
            HIP_COPY(e, hipMemcpy2DAsync(signal_present + x.c0 * row_blocks + x.t0 / bb, row_blocks, e->sl_allowed[b].p,
                                        tblk, tblk, x.nc, hipMemcpyDeviceToHost, s));
        HIP_COPY(e, hipMemcpyAsync(e->h_slice_counts + b * sc, e->sl_count[b].p, x.nc * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipEventRecord(e->ev_free[b], s));
    }
    HIP_TRY(e, hipStreamSynchronize(s));
    settle(n_slices - 1);
    for (uint32_t c = 0; c < n_ch; c++)   // like the unsliced path: zeros behind the valid samples
        if (filled[c] < row_pcm) memset(pcm + c * row_pcm + filled[c], 0, (row_pcm - filled[c]) * sizeof(int16_t));
    if (pcm_count) memcpy(pcm_count, filled.data(), n_ch * sizeof(uint32_t));
    return IQD_OK;
}

int iqd_accept_iq(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const uint8_t *iq, size_t bytes_per_ch,
                  int16_t *pcm, uint32_t *pcm_count, uint32_t *magnitude, uint8_t *signal_present)
{
    if (!range_ok(e, first_ch, n_ch) || !iq || !pcm) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    const uint32_t call_bb = call_block_bytes(e, first_ch, n_ch, bytes_per_ch);
    if (!call_bb) return e->fail(IQD_EINVAL, IQD_LEN_MSG, bytes_per_ch, e->block_bytes);
    (void)hipSetDevice(e->device);
    hipStream_t s = e->stream;
    struct HostPath { iqd_t *e; HostPath(iqd_t *e_) : e(e_) { e->in_host_path = true; } ~HostPath() { e->in_host_path = false; } } host_path(e);
    const size_t in_bytes = (size_t)n_ch * bytes_per_ch;
    const size_t pcm_bytes = (size_t)n_ch * (bytes_per_ch / 64) * sizeof(int16_t);
    const size_t nb = (size_t)n_ch * (bytes_per_ch / call_bb);
    if (in_bytes >= 2 * SLICE_BYTES)
        return accept_sliced(e, first_ch, n_ch, iq, bytes_per_ch, call_bb, pcm, pcm_count, magnitude, signal_present);
    if (in_bytes <= SMALL_CALL_BYTES) {
        // A small call - the reference's own operating point is ONE 32768-byte block per call (DataConsumer.cc:333-346) -
        // is all latency: through the staging path it was an upload, two fills, two kernels and up to four downloads,
        // ~90 us per block.  Here the kernels read the block from, and write their results to, page-locked host memory
        // that the device addresses directly: no copy operation is queued at all, two launches and one wait remain.
        const size_t off_cnt = (pcm_bytes + 15) & ~(size_t)15, off_mag = off_cnt + (((size_t)n_ch * 4 + 15) & ~(size_t)15);
        const size_t off_al = off_mag + ((nb * 4 + 15) & ~(size_t)15), out_bytes = off_al + ((nb + 15) & ~(size_t)15);
        if (e->h_small_cap < in_bytes + out_bytes + 16) {
            if (e->h_small) (void)hipHostFree(e->h_small);
            e->h_small = nullptr;
            e->h_small_cap = 0;
            const size_t want = 2 * (in_bytes + out_bytes) + 4096;
            HIP_TRY(e, hipHostMalloc((void **)&e->h_small, want, hipHostMallocDefault));
            e->h_small_cap = want;
        }
        uint8_t *h_in = e->h_small, *h_out = e->h_small + ((in_bytes + 15) & ~(size_t)15);
        memcpy(h_in, iq, in_bytes);
        memset(h_out, 0, pcm_bytes);   // (zeros behind the valid samples, as the staging path leaves them)
        int rc = iqd_accept_iq_device(e, first_ch, n_ch, h_in, bytes_per_ch, h_out, h_out + off_cnt,
                                      magnitude ? h_out + off_mag : nullptr, signal_present ? h_out + off_al : nullptr);
        if (rc != IQD_OK) return rc;
        HIP_TRY(e, hipStreamSynchronize(s));
        memcpy(pcm, h_out, pcm_bytes);
        if (pcm_count) memcpy(pcm_count, h_out + off_cnt, (size_t)n_ch * sizeof(uint32_t));
        if (magnitude) memcpy(magnitude, h_out + off_mag, nb * sizeof(uint32_t));
        if (signal_present) memcpy(signal_present, h_out + off_al, nb);
        return IQD_OK;
    }
    HIP_TRY(e, e->st_iq.ensure(in_bytes));
    HIP_TRY(e, e->st_pcm.ensure(pcm_bytes));
    HIP_TRY(e, e->st_count.ensure(n_ch * sizeof(uint32_t)));
    HIP_TRY(e, e->st_mag.ensure(nb * sizeof(uint32_t)));
    HIP_TRY(e, e->st_allowed.ensure(nb));
    HIP_COPY(e, hipMemcpyAsync(e->st_iq.p, iq, in_bytes, hipMemcpyHostToDevice, s));
    HIP_COPY(e, hipMemsetAsync(e->st_pcm.p, 0, pcm_bytes, s));
    int rc = iqd_accept_iq_device(e, first_ch, n_ch, e->st_iq.p, bytes_per_ch, e->st_pcm.p, e->st_count.p,
                                  magnitude ? e->st_mag.p : nullptr, signal_present ? e->st_allowed.p : nullptr);
    if (rc != IQD_OK) return rc;
    HIP_COPY(e, hipMemcpyAsync(pcm, e->st_pcm.p, pcm_bytes, hipMemcpyDeviceToHost, s));
    if (pcm_count) HIP_COPY(e, hipMemcpyAsync(pcm_count, e->st_count.p, n_ch * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (magnitude) HIP_COPY(e, hipMemcpyAsync(magnitude, e->st_mag.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (signal_present) HIP_COPY(e, hipMemcpyAsync(signal_present, e->st_allowed.p, nb, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    return IQD_OK;
}

// ---- the demodulators' own entry ----------------------------------------------------------------
// {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData(int8_t *bufferPtr, uint32_t bufferLength), e.g. WbFmDemodulator.cc:383-411:
// SIGNED bytes as the processor hands them over (after the -128 and the rotation), straight into the demodulator -
// no squelch, no notification, no AGC.  The demodulator objects are the ones the processor uses (Radio.cc:150-181), so
// its filter state is shared with iqd_accept_iq of the same channel.  Run here as an accept of the same channel with
// the front end neutralised: offset-binary bytes s ^ 0x80, rotation selector 0 (a channel's kept tail is stored as raw
// bytes under its selector; retail_kernel rewrites it exactly when the selector moves, and back afterwards), the
// squelch path bypassed.
int iqd_demod_set_sideband(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int lsb)
{
    if (!range_ok(e, first_ch, n_ch)) return IQD_EINVAL;
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        ChanParams &p = e->h_params[c];
        p.ssb_lsb = lsb ? 1 : 0;
        if (p.mode == IQD_MODE_LSB || p.mode == IQD_MODE_USB) p.mode = lsb ? IQD_MODE_LSB : IQD_MODE_USB;   // (one flag in the reference: SsbDemodulator.cc:333-367)
    }
    e->params_dirty = e->lists_dirty = true;
    return IQD_OK;
}

int iqd_demod_accept(iqd_t *e, uint32_t first_ch, uint32_t n_ch, int demod, const int8_t *iq, size_t bytes_per_ch, int16_t *pcm)
{
    if (!range_ok(e, first_ch, n_ch) || !iq || !pcm) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    if (demod < IQD_DEMOD_AM || demod > IQD_DEMOD_SSB) return e->fail(IQD_EINVAL, "demod must be IQD_DEMOD_AM .. IQD_DEMOD_SSB");
    if (bytes_per_ch == 0 || bytes_per_ch % 64 != 0)
        return e->fail(IQD_EINVAL, "bytes_per_ch (%zu) must be a positive multiple of 64", bytes_per_ch);
    // the channels as this call needs them; what the caller had set comes back afterwards
    std::vector<int32_t> mode0(n_ch), rot0(n_ch);
    {
        std::lock_guard<std::mutex> lk(e->mu);
        for (uint32_t c = 0; c < n_ch; c++) {
            ChanParams &p = e->h_params[first_ch + c];
            mode0[c] = p.mode;
            rot0[c] = p.rotation;
            p.mode = demod == IQD_DEMOD_AM ? IQD_MODE_AM : demod == IQD_DEMOD_FM ? IQD_MODE_FM : demod == IQD_DEMOD_WBFM ? IQD_MODE_WBFM
                                           : (p.ssb_lsb ? IQD_MODE_LSB : IQD_MODE_USB);
            p.rotation = 0;
        }
        e->params_dirty = e->lists_dirty = true;
    }
    e->demod_bypass = true;
    int rc = IQD_OK;
    // whole blocks first, then what is left as one short block; at most ~16 MiB of input per accept (the staging path)
    const size_t bb = e->block_bytes, row_pcm = bytes_per_ch / 64;
    size_t max_t = ((size_t)16 << 20) / n_ch / bb * bb;
    if (max_t < bb) max_t = bb;
    std::vector<uint8_t> u8;
    std::vector<int16_t> part;
    for (size_t t0 = 0; t0 < bytes_per_ch && rc == IQD_OK;) {
        const size_t rest = bytes_per_ch - t0;
        const size_t t = rest >= bb ? std::min(rest / bb * bb, max_t) : rest;
        u8.resize((size_t)n_ch * t);
        part.resize((size_t)n_ch * (t / 64));
        for (uint32_t c = 0; c < n_ch; c++) {
            const uint8_t *src = (const uint8_t *)iq + (size_t)c * bytes_per_ch + t0;
            uint8_t *dst = u8.data() + (size_t)c * t;
            for (size_t k = 0; k < t; k++) dst[k] = src[k] ^ 0x80u;
        }
        rc = iqd_accept_iq(e, first_ch, n_ch, u8.data(), t, part.data(), nullptr, nullptr, nullptr);
        if (rc == IQD_OK)
            for (uint32_t c = 0; c < n_ch; c++)
                memcpy(pcm + (size_t)c * row_pcm + t0 / 64, part.data() + (size_t)c * (t / 64), (t / 64) * sizeof(int16_t));
        t0 += t;
    }
    e->demod_bypass = false;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        for (uint32_t c = 0; c < n_ch; c++) {
            ChanParams &p = e->h_params[first_ch + c];
            p.mode = mode0[c];
            p.rotation = rot0[c];
        }
        e->params_dirty = e->lists_dirty = true;
    }
    return rc;
}

}  // extern "C"
