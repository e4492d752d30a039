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
#include "iqd_plan.h"
#include "iqd_kernels.h"
#include "iqd_stream.h"
#include "iqd_stream_mixed.h"
#include "iqd_taps.h"
#include "iqd_wbfm.h"
#include "iqd_chains.h"

using namespace iqd;

// (the streaming thresholds STREAM_MIN_SEG_* and every tile / stream / one-launch decision: iqd_plan.h, iqd_plan.cpp)

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
    // iqd_config::flags and the measurement knobs, read from the environment ONCE at creation (include/iqdemod.h): what
    // plan_call() decides from.  The plan of the last call is kept while the call's shape repeats (planning a
    // 2^28-sample FM + WBFM call cost more host time than its kernels run).
    PlanKnobs knobs;
    CallShape plan_shape;
    CallPlan plan;
    bool plan_valid = false;
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
    std::vector<uint32_t> mode_gen, rot_gen;   // [n_ch]: bumped by iqd_set_mode / iqd_set_rotation (iqd_demod_accept restores only what nobody set meanwhile)
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
    PlanKnobs &kn = e->knobs;
    kn.flags = cfg->flags;
    kn.n_cus = e->n_cus;
    kn.wbfm_chunk = WBFM_CHUNK; kn.wbfm_cold_halo = COLD_HALO; kn.ch_chunk = CH_CHUNK; kn.dc_tile = DC_TILE;
    if (const char *env = getenv("IQD_WBFM_PATH")) kn.env_path = env[0] == 's' ? 1 : env[0] == 't' ? -1 : 0;
    kn.env_full_grid = getenv("IQD_FULL_GRID") != nullptr;
    if (const char *env = getenv("IQD_SHARES")) kn.env_shares_by_cost = env[0] == 'c';
    if (const char *env = getenv("IQD_FAMILY_NS")) {
        float w[FAM_COUNT];
        if (sscanf(env, "%f,%f,%f,%f", &w[0], &w[1], &w[2], &w[3]) == 4 && w[0] > 0.f && w[1] > 0.f && w[2] > 0.f && w[3] > 0.f)
            for (int f = 0; f < FAM_COUNT; f++) kn.fam_ns[f] = w[f];
    }
    if (const char *env = getenv("IQD_D4_GRAN")) kn.env_d4_gran = (uint32_t)atoi(env);
    if (const char *env = getenv("IQD_D4_LEADFREE")) kn.d4_leadfree = atoi(env);
    if (const char *env = getenv("IQD_STREAM_MIN_SEG")) kn.env_stream_min_seg = atoi(env) > 0 ? (uint64_t)atoi(env) : 0;
    if (const char *env = getenv("IQD_AM_STREAM_MIN")) kn.env_am_stream_min = atoi(env) > 0 ? (uint32_t)atoi(env) : AM_STREAM_MIN_PCM;
    if (const char *env = getenv("IQD_MIXED")) kn.env_mixed_forked = env[0] == 'f' && env[1] == 'o';
    if (const char *env = getenv("IQD_FAMILY_WEIGHTS")) {   // "am,fm,wbfm,ssb" (measurement runs)
        float w[FAM_COUNT];
        if (sscanf(env, "%f,%f,%f,%f", &w[0], &w[1], &w[2], &w[3]) == 4 && w[0] > 0 && w[1] > 0 && w[2] > 0 && w[3] > 0)
            for (int f = 0; f < FAM_COUNT; f++) kn.fam_weight[f] = w[f];
    }
    if (const char *env = getenv("IQD_STREAM_WGS")) kn.env_stream_wgs = atoi(env) > 0 ? (uint32_t)atoi(env) : 0u;
    if (const char *env = getenv("IQD_STREAM_GRAN")) kn.env_stream_gran = (uint32_t)atoi(env);
    if (const char *env = getenv("IQD_PLAN_CHUNKS")) kn.env_plan_chunks = atoi(env) > 0 ? (uint32_t)atoi(env) : 0u;
    if (const char *env = getenv("IQD_RINGS")) kn.env_rings = atoi(env) > 0 ? (uint32_t)atoi(env) : 0u;
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
    e->knobs.stream_ok = e->stream_ok;
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
    if (e->mode_gen.size() < e->h_params.size()) e->mode_gen.resize(e->h_params.size(), 0u);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        e->h_params[c].mode = mode;
        e->mode_gen[c]++;
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
    if (e->rot_gen.size() < e->h_params.size()) e->rot_gen.resize(e->h_params.size(), 0u);
    for (uint32_t c = first_ch; c < first_ch + n_ch; c++) {
        e->h_params[c].rotation = rotation;
        e->rot_gen[c]++;
    }
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

// ---- accept: describe, plan, then queue ----------------------------------------------------------
// iqd_accept_iq_device() = prepare_call (parameter mirror, channel lists, under the lock) -> describe_call (the call as plain
// data) -> plan_call (iqd_plan.cpp: every tile / stream / one-launch / share decision, nothing queued yet; cached while the
// shape repeats) -> queue_prepass -> queue_family per family -> queue_commit.  The queue_* functions decide nothing.

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

namespace {
// What one accept call carries from phase to phase.
struct CallCtx {
    uint32_t first_ch = 0, n_ch = 0;
    const void *iq_dev = nullptr;
    size_t bytes_per_ch = 0;
    void *pcm_dev = nullptr, *pcm_count_dev = nullptr, *magnitude_dev = nullptr, *signal_present_dev = nullptr;
    uint32_t call_bb = 0, call_bs = 0, n_blocks = 0, vlen = 0;
    bool gated = false, any_agc = false, want_mag = false, pre_overlap = false, fused_mag = false;
    int pre_set = -1;                               // IQD_F_PREPASS_OVERLAP: the buffer set this call's pre-pass filled
    DevBuf *gate_blk = nullptr, *gate_vlen = nullptr;
    SquelchLaunch q{};
    ChainLaunch base{};
    const CallPlan *plan = nullptr;
    hipStream_t s_main = nullptr;
    bool lane_used[4] = {false, false, false, false};
    bool timed = false, evp_open = false;           // profiling: the event pair of the call
    std::pair<hipEvent_t, hipEvent_t> evp{nullptr, nullptr};
    MixedStreamArgs mix{};
    bool epoch_report_now = false;                  // this call's WBFM tail updates report whether a gain change is still in reach (h_epoch_report)
    ChainLaunch tail_a{};                           // a tail update (WBFM: repair check + state commit + tail) that rides in the squelch launch
    int tail_f = 0;
    bool tail_pending = false;
    bool tail_dc = false;                           // ... and the family's one-wave DC-removal pass with it (AM / SSB streaming launches)
};

int take_event_pair(iqd_t *e, CallCtx &x, hipStream_t s)
{
    if (e->ev_free_pairs.empty()) {
        hipEvent_t a0, a1;
        HIP_TRY(e, hipEventCreate(&a0));
        HIP_TRY(e, hipEventCreate(&a1));
        e->ev_free_pairs.emplace_back(a0, a1);
    }
    x.evp = e->ev_free_pairs.back();
    e->ev_free_pairs.pop_back();
    HIP_TRY(e, hipEventRecord(x.evp.first, s));
    return IQD_OK;
}
}  // namespace

// Under the lock: the epoch mirror's ageing report, parameter upload, channel lists, AGC configuration.
static int prepare_call(iqd_t *e, CallCtx &x)
{
    hipStream_t s = e->stream;
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
    if (e->lists_dirty || e->list_first != x.first_ch || e->list_n != x.n_ch) {
        rebuild_lists(e, x.first_ch, x.n_ch);
        for (int f = 0; f <= FAM_COUNT; f++) {
            const auto &l = e->h_lists[f];
            if (l.empty()) continue;
            HIP_TRY(e, e->lists[f].ensure(l.size() * sizeof(uint32_t)));
            HIP_COPY(e, hipMemcpyAsync(e->lists[f].p, l.data(), l.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        }
        HIP_TRY(e, hipStreamSynchronize(s));
    }
    int rc = agc_sync(e);
    if (rc != IQD_OK) return rc;
    x.gated = e->any_gated;
    x.any_agc = e->any_agc;
    return IQD_OK;
}

// The call as plain data for plan_call(): per family how many channels, per rotation selector, whether every gain keeps
// the (int16) casts bounded, whether a WBFM gain change is still in reach of a lead-in.
static void describe_call(const iqd_t *e, const CallCtx &x, CallShape &c)
{
    c = CallShape{};
    c.vlen = x.vlen;
    c.pcm_per_ch = (uint32_t)(x.bytes_per_ch / 64);
    c.gated = x.gated;
    for (int f = 0; f < FAM_COUNT; f++) {
        const auto &l = e->h_lists[f];
        FamilyShape &s = c.fam[f];
        s.n_list = (uint32_t)l.size();
        if (l.empty()) continue;
        for (int r = 0; r < 3; r++) s.rot_count[r] = e->rot_count[f][r];
        // (from the snapshot prepare_call took under the lock, not from the live parameters a setter may be writing: ADVICE r5.
        //  The list is sorted +Fs/4, none, -Fs/4, so its first channel's selector is the first non-empty group's)
        s.rot_first = s.rot_count[0] ? 1 : (s.rot_count[1] ? 0 : -1);
        if (f == FAM_WBFM)
            for (uint32_t ch : l) {
                s.cast_bounded = s.cast_bounded && e->wbfm_kmax[x.first_ch + ch] * 3.1730f < 2147483648.0f;
                s.epochs_in_reach = s.epochs_in_reach || (e->wbfm_epochs_live && e->wbfm_epoch_left[x.first_ch + ch] != 0);
            }
        if (f == FAM_FM)
            for (uint32_t ch : l) s.cast_bounded = s.cast_bounded && e->fm_kmax[x.first_ch + ch] * 6.35f < 2147483648.0f;
    }
}

static bool same_shape(const CallShape &a, const CallShape &b)
{
    if (a.vlen != b.vlen || a.pcm_per_ch != b.pcm_per_ch || a.gated != b.gated) return false;
    for (int f = 0; f < FAM_COUNT; f++) {
        const FamilyShape &p = a.fam[f], &q = b.fam[f];
        if (p.n_list != q.n_list || p.rot_count[0] != q.rot_count[0] || p.rot_count[1] != q.rot_count[1] || p.rot_count[2] != q.rot_count[2] ||
            p.rot_first != q.rot_first || p.cast_bounded != q.cast_bounded || p.epochs_in_reach != q.epochs_in_reach)
            return false;
    }
    return true;
}

// The squelch's part in front of the pipelines: the block sums start at zero; a gated call's magnitudes, decisions and
// open-block lists (inline on the engine's stream, or one call ahead on the pre-pass stream); the launch descriptor
// every family's launch starts from.
static int queue_prepass(iqd_t *e, CallCtx &x)
{
    hipStream_t s = x.s_main;
    const uint32_t n_ch = x.n_ch, n_blocks = x.n_blocks;
    // the pre-pass of a gated call one call ahead, on its own stream (include/iqdemod.h: IQD_F_PREPASS_OVERLAP)
    if (x.pre_overlap && !e->pre_stream) {
        HIP_TRY(e, hipStreamCreateWithFlags(&e->pre_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_pre_done[k], hipEventDisableTiming));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_chain_done[k], hipEventDisableTiming));
        }
        HIP_TRY(e, hipEventCreateWithFlags(&e->ev_main_decisions, hipEventDisableTiming));
        // whatever squelch pass ran before this one ran on the engine's stream, unrecorded: the new stream starts behind it
        HIP_TRY(e, hipEventRecord(e->ev_main_decisions, s));
        e->decisions_on_main = true;
    }
    if (!x.pre_overlap) {   // the sums start at zero: by memset, unless the previous call's squelch pass left this many of them zero
        const void *before = e->mag_sums.p;
        HIP_TRY(e, e->mag_sums.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        if (e->mag_sums.p != before) e->mag_sums_zero = 0;
        if (x.want_mag && e->mag_sums_zero < (size_t)n_ch * n_blocks)
            HIP_COPY(e, hipMemsetAsync(e->mag_sums.p, 0, (size_t)n_ch * n_blocks * sizeof(uint32_t), s));
        e->mag_sums_zero = 0;   // from here on the call writes into them
    }

    SquelchLaunch &q = x.q;
    q.n_ch = n_ch; q.first_ch = x.first_ch; q.n_blocks = n_blocks; q.block_samples = x.call_bs;
    q.params = e->d_params;
    q.mag_sums = e->mag_sums.as<uint32_t>();
    q.tracker = e->d_tracker;
    q.magnitude = (uint32_t *)x.magnitude_dev;
    q.allowed = (uint8_t *)x.signal_present_dev;
    q.pcm_count = (uint32_t *)x.pcm_count_dev;
    q.agc_cfg = e->d_agc_cfg;
    q.agc = e->d_agc;
    q.any_agc = x.any_agc ? 1u : 0u;
    q.scan_cfg = e->d_scan_cfg;
    q.scan = e->d_scan;
    if (e->trace_on) {
        HIP_TRY(e, e->gain_trace.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        q.gain_trace = e->gain_trace.as<uint32_t>();
        HIP_TRY(e, e->freq_trace.ensure((size_t)n_ch * n_blocks * sizeof(unsigned long long)));
        q.freq_trace = e->freq_trace.as<unsigned long long>();
        e->trace_first = x.first_ch; e->trace_n = n_ch; e->trace_blocks = n_blocks;
    }

    x.gate_blk = &e->blk_lists;
    x.gate_vlen = &e->vlen;
    if (x.pre_overlap) {
        // magnitudes, decisions and open-block lists of THIS call on the pre-pass stream, which the previous call's pipelines
        // (main stream) do not hold up: it waits for the decision pass before it (its own stream order - or the main stream's,
        // where the previous call's pass ran there), and for the last reader of the buffer set it is about to overwrite
        hipStream_t ps = e->pre_stream;
        x.pre_set = e->gate_set ^= 1;
        DevBuf &sums = e->g_sums[x.pre_set];
        x.gate_blk = &e->g_blk[x.pre_set];
        x.gate_vlen = &e->g_vlen[x.pre_set];
        if (e->chain_pending[x.pre_set]) HIP_TRY(e, hipStreamWaitEvent(ps, e->ev_chain_done[x.pre_set], 0));
        if (e->decisions_on_main) {
            HIP_TRY(e, hipStreamWaitEvent(ps, e->ev_main_decisions, 0));
            e->decisions_on_main = false;
        }
        HIP_TRY(e, sums.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, x.gate_blk->ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, x.gate_vlen->ensure((size_t)n_ch * sizeof(uint32_t)));
        HIP_COPY(e, hipMemsetAsync(sums.p, 0, (size_t)n_ch * n_blocks * sizeof(uint32_t), ps));
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)x.iq_dev, x.bytes_per_ch, nullptr, n_ch, x.call_bs, n_blocks, sums.as<uint32_t>(), ps));
        q.mag_sums = sums.as<uint32_t>();
        q.blk_lists = x.gate_blk->as<uint32_t>();
        q.vlen_out = x.gate_vlen->as<uint32_t>();
        HIP_LAUNCH(e, launch_squelch(q, false, ps));
        HIP_TRY(e, hipEventRecord(e->ev_pre_done[x.pre_set], ps));
        HIP_TRY(e, hipStreamWaitEvent(s, e->ev_pre_done[x.pre_set], 0));   // the pipelines read the lists
    } else if (x.gated) {
        // pass 1: magnitudes of every block, then the squelch decisions and open-block lists
        if (e->pre_stream) HIP_TRY(e, hipStreamSynchronize(e->pre_stream));   // (a call of the overlapped kind before this one)
        HIP_TRY(e, e->blk_lists.ensure((size_t)n_ch * n_blocks * sizeof(uint32_t)));
        HIP_TRY(e, e->vlen.ensure((size_t)n_ch * sizeof(uint32_t)));
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)x.iq_dev, x.bytes_per_ch, nullptr, n_ch, x.call_bs, n_blocks, e->mag_sums.as<uint32_t>(), s));
        q.blk_lists = e->blk_lists.as<uint32_t>();
        q.vlen_out = e->vlen.as<uint32_t>();
        // (Round 2 read one word back here - did any channel lose a block? - to let an all-open call take the streaming
        // kernels, which could not gate.  They can now (a virtual sample axis over each channel's open blocks), so the
        // call stays asynchronous whatever the squelch decides.)
        HIP_LAUNCH(e, launch_squelch(q, false, s));
    }
    if (!x.gated && e->pre_stream) HIP_TRY(e, hipStreamWaitEvent(s, e->ev_pre_done[e->gate_set], 0));   // (the tracker / AGC state this call's squelch pass continues from)

    ChainLaunch &base = x.base;
    base.iq = (const uint8_t *)x.iq_dev;
    base.ch_stride_bytes = x.bytes_per_ch;
    base.first_ch = x.first_ch;
    base.vlen = x.vlen;
    base.vlen_gated = x.gated ? x.gate_vlen->as<uint32_t>() : nullptr;   // the chain kernels walk each channel's open blocks
    base.blk_lists = x.gated ? x.gate_blk->as<uint32_t>() : nullptr;
    base.n_blocks = n_blocks;
    base.block_samples = x.call_bs;
    base.block_magic = block_magic(x.call_bs);
    base.tails = e->d_tails;
    base.params = e->d_params;
    base.wbfm_carry = e->d_wcarry;
    base.dc_carry = e->d_dc;
    base.epochs = e->d_epochs;
    base.atan_lut = e->d_atan;
    base.fm_lut = e->d_fmlut;
    base.pcm = (int16_t *)x.pcm_dev;
    base.pcm_stride = x.bytes_per_ch / 64;
    base.mag_sums = e->mag_sums.as<uint32_t>();
    base.counters = e->d_counters;
    base.stamps = e->d_stamps;
    base.n_ch_call = n_ch;
    return IQD_OK;
}

// AM / SSB: the detector stream's buffer and the DC pass's records of this launch.  One buffer per family: short rows are
// written time-major by the tile kernels and channel-major by the streaming pipelines, and the two families of a call may
// take different paths (round 4's fuzzer: an AM family that streamed beside an SSB family on the tile kernels overwrote its
// detector stream).
static int attach_dc_buffers(iqd_t *e, CallCtx &x, int f, ChainLaunch &a, hipStream_t s)
{
    DevBuf &b8 = f == FAM_SSB ? e->base8k2 : e->base8k;
    HIP_TRY(e, b8.ensure((size_t)x.n_ch * x.base.pcm_stride * sizeof(int32_t)));
    a.base8k = b8.as<int32_t>();
    a.dc_tiles = (uint32_t)((x.base.pcm_stride + DC_TILE - 1) / DC_TILE);
    DevBuf &dcr = f == FAM_SSB ? e->dc_records2 : e->dc_records;   // AM and SSB may run side by side
    // records, then one redo flag per channel: zero between calls (dc_redo_kernel clears what it used)
    const size_t rec_bytes = (size_t)a.n_list * a.dc_tiles * sizeof(DcRecord);
    const bool grown = dcr.cap < rec_bytes + a.n_list * sizeof(uint32_t);
    HIP_TRY(e, dcr.ensure(rec_bytes + a.n_list * sizeof(uint32_t)));
    a.dc_records = dcr.p;
    // the flags sit behind the records, whose extent changes with the call: clear them whenever it may have
    // (same offset but more channels than last time: the new flags lie over old record bytes)
    size_t (&layout)[2] = e->dcr_layout[f == FAM_SSB];
    if (grown || layout[0] != rec_bytes || layout[1] < a.n_list) {
        HIP_COPY(e, hipMemsetAsync((char *)dcr.p + rec_bytes, 0, a.n_list * sizeof(uint32_t), s));
        layout[0] = rec_bytes;
        layout[1] = a.n_list;
    }
    return IQD_OK;
}

// One family's launches as the plan says: its tile kernel or streaming pipeline (as a kernel of its own, or as a range of the
// one launch that queue_commit() starts), then what follows it - hand-off verification, repair, DC pass, tail update - unless
// that rides in the squelch launch at the call's end.
static int queue_family(iqd_t *e, CallCtx &x, int f)
{
    const CallPlan &plan = *x.plan;
    const FamilyPlan &fp = plan.fam[f];
    const bool fused = plan.fused, forked = plan.forked, streams = fp.path == PLAN_STREAM;
    const uint32_t n_list = (uint32_t)e->h_lists[f].size();
    hipStream_t s = fp.lane == 0 ? x.s_main : e->fam_stream[fp.lane - 1];
    if (fp.lane != 0 && !x.lane_used[fp.lane]) HIP_TRY(e, hipStreamWaitEvent(s, e->fam_fork, 0));
    x.lane_used[fp.lane] = true;

    ChainLaunch a = x.base;
    a.ch_list = e->lists[f].as<uint32_t>();
    a.n_list = n_list;
    a.tile_len = fp.tile_len;
    a.tiles_per_ch = fp.tiles_per_ch;
    if (f == FAM_WBFM && x.gated && e->wbfm_epochs_live && !e->epoch_report_pending) {
        bool any = false;
        for (uint32_t c : e->h_lists[FAM_WBFM]) any = any || e->wbfm_epoch_left[x.first_ch + c] != 0;
        if (any) {
            if (!e->h_epoch_report) HIP_TRY(e, hipHostMalloc((void **)&e->h_epoch_report, 2 * sizeof(uint32_t), hipHostMallocDefault));
            ((volatile uint32_t *)e->h_epoch_report)[0] = 0u;
            a.epoch_report = e->h_epoch_report;
            e->epoch_report_channels.clear();
            for (uint32_t c : e->h_lists[FAM_WBFM]) e->epoch_report_channels.push_back(x.first_ch + c);
            x.epoch_report_now = true;
        }
    }
    if (e->profiling && !x.timed && !fused) {
        int rc = take_event_pair(e, x, s);
        if (rc != IQD_OK) return rc;
    }
    if (fused) {   // a range of the one launch's workgroups
        a.wg_first = fp.wg_first;
        a.wg_count = fp.grid;
    }
    if (f == FAM_WBFM) {
        HIP_TRY(e, e->records.ensure((size_t)n_list * a.tiles_per_ch * sizeof(WbfmRecord)));
        a.records = e->records.as<WbfmRecord>();
        if (e->repair_flags.cap < n_list * sizeof(uint32_t)) {   // zero between calls: the repair kernel clears what it used
            HIP_TRY(e, e->repair_flags.ensure(n_list * sizeof(uint32_t)));
            HIP_COPY(e, hipMemsetAsync(e->repair_flags.p, 0, e->repair_flags.cap, s));
        }
        a.repair_flags = e->repair_flags.as<uint32_t>();
        if (streams) {
            const int stream_rot = e->plan_shape.fam[FAM_WBFM].rot_first;   // (the plan's snapshot: queue_* decide nothing and read no live parameter)
            StreamArgs sa = e->stream_args;
            sa.amat = e->d_amat[stream_rot + 1];
            sa.half_lut = e->d_half_lut;
            sa.n_segments = n_list * a.tiles_per_ch;
            sa.grouped = fp.grouped ? 1u : 0u;
            for (int r = 0; r < 3; r++) {
                sa.group_start[r] = fp.grouped ? fp.group_start[r] : 0u;
                sa.group_li0[r] = fp.grouped ? fp.group_li0[r] : 0u;
                sa.group_nseg[r] = fp.grouped ? fp.group_nseg[r] : 0u;
                sa.amat3[r] = e->d_amat[2 - r];          // d_amat[] is indexed by selector + 1; the groups run +1, 0, -1
            }
            sa.group_start[3] = fp.grouped ? fp.group_start[3] : 0u;
            sa.rounds = fp.rounds;
            sa.rings = fp.rings;
            HIP_TRY(e, e->stream_hist.ensure((size_t)sa.n_segments * sizeof(StHist)));
            sa.hist = e->stream_hist.as<StHist>();
            a.verify_at_end = x.gated ? 2u : 1u;   // (2: the hand-offs are counted on the device - how many tiles a channel has depends on its squelch)
            if (fused) {   // the fix-up rides in the launch behind the one launch
                x.mix.a[f] = a;
                x.mix.sa = sa;
                x.mix.wbfm_rot = stream_rot;
            } else {
                HIP_LAUNCH(e, launch_wbfm_stream(a, sa, stream_rot, x.fused_mag, fp.epochs, fp.grid, s));
                HIP_LAUNCH(e, launch_wbfm_stream_fixup(a, sa, s));
            }
            if (!x.gated) e->stream_handoffs += (uint64_t)n_list * ((x.vlen + a.tile_len - 1) / a.tile_len - 1);
            e->stats.stream_launches++;
        } else {
            HIP_LAUNCH(e, launch_wbfm(a, x.gated, x.fused_mag, n_list * a.tiles_per_ch, s));
        }
    } else {
        D4Args d4 = e->d4_args;
        if (streams) {
            for (int r = 0; r < 3; r++) {
                d4.group_start[r] = fp.group_start[r];
                d4.group_li0[r] = fp.group_li0[r];
                d4.group_nseg[r] = fp.group_nseg[r];
            }
            d4.group_start[3] = fp.group_start[3];
            d4.amat = e->d_amat4 + (size_t)(f == FAM_FM ? 0 : 3) * 4 * 64 * 4;
            d4.fm_lut = e->d_fmlut;
            d4.halo = (int32_t)fp.halo;
            d4.lead_shift = fp.lead_shift;
            d4.rounds = fp.rounds;
            d4.rings = fp.rings;
        }
        if (f != FAM_FM) {
            int rc = attach_dc_buffers(e, x, f, a, s);
            if (rc != IQD_OK) return rc;
            if (streams) {   // (the pipeline writes the detector stream channel-major)
                a.base_stride_ch = x.base.pcm_stride;
                a.base_stride_t = 1;
                // ... as int16 into the PCM rows themselves when the DC pass behind it is the one-wave pass (in place, round 6);
                // rows so long that they take the many-wave pass keep the int32 stream: its tiles warm up over their neighbours' input
                static const bool no_det16 = getenv("IQD_NO_DET16") != nullptr;   // (the A/B)
                a.det16 = a.dc_tiles < 2 && !no_det16 ? 1u : 0u;
            }
        }
        if (streams) {
            if (fused) {
                x.mix.a[f] = a;
                x.mix.d4[f] = d4;
            } else {
                HIP_LAUNCH(e, launch_d4_stream(a, d4, f == FAM_FM ? D4_FM : f == FAM_AM ? D4_AM : D4_SSB, x.fused_mag, fp.grid, s));
                // the DC-removal pass: a role of the call's closing launch where there is one to ride in and the rows take the
                // one-wave pass (round 5: 0.019 ms and a queue gap less per step); else its own launches
                x.tail_dc = f != FAM_FM && a.dc_tiles < 2 && !forked && !x.gated && !e->demod_bypass &&
                            (x.want_mag || x.pcm_count_dev || x.signal_present_dev || e->trace_on);
                if (f != FAM_FM && !x.tail_dc) HIP_LAUNCH(e, launch_am_dc(a, f, s, true));
            }
            e->stats.stream_launches++;
        } else if (f == FAM_FM) {
            HIP_LAUNCH(e, launch_fm(a, x.gated, x.fused_mag, n_list * a.tiles_per_ch, s));
        } else {
            HIP_LAUNCH(e, launch_am(a, f, x.gated, x.fused_mag, n_list * a.tiles_per_ch, s));
        }
    }
    if (e->profiling && !x.timed && !fused) {
        // the pair closes behind the step's last launch (queue_commit), so that the timed region holds EVERY launch of the step
        // (the repair-check / commit / squelch launch was left out)
        x.evp_open = true;   // (forked plans opened theirs in front of the fork: iqd_accept_iq_device)
        x.timed = true;
    }
    e->stats.kernel_launches++;
    if (fused) return IQD_OK;   // (one launch for all the families and one for what follows them: queue_commit)
    if (f == FAM_WBFM && !x.gated && e->wbfm_epochs_live)   // every channel of the family has consumed vlen samples
        for (uint32_t c : e->h_lists[FAM_WBFM]) {
            uint32_t &left = e->wbfm_epoch_left[x.first_ch + c];
            if (!left) continue;
            left = left > x.vlen ? left - x.vlen : 0u;
            if (!left) e->wbfm_epochs_live--;
        }
    const bool rides_with_squelch = !forked && !x.gated && !e->demod_bypass &&
                                    (x.want_mag || x.pcm_count_dev || x.signal_present_dev || e->trace_on);
    if (f == FAM_WBFM) {
        // hand-off verification (streaming launches: done by their fix-up kernel), repair of what it flags
        // (normally an immediate exit), then state commit + tail - in the squelch launch at the call's end when the call has
        // no other family
        if (!streams) HIP_LAUNCH(e, launch_wbfm_verify(a, s));
        if (rides_with_squelch) {
            x.tail_a = a;
            x.tail_f = f;
            x.tail_pending = true;
        } else {
            HIP_LAUNCH(e, launch_wbfm_repair(a, x.gated, s));   // (ends with the channels' state commit and tail update)
        }
    } else if (rides_with_squelch) {
        x.tail_a = a;        // the only family of the call: its tail update rides in the squelch launch
        x.tail_f = f;
        x.tail_pending = true;
    } else {
        HIP_LAUNCH(e, launch_tail_update(a, f, s));
    }
    return IQD_OK;
}

// Behind the families: the one launch that holds all of them (fused plans) and its follower, the side streams' join, the
// magnitudes of channels in mode None, the squelch pass with whatever rides in it, the call's events and counters.
static int queue_commit(iqd_t *e, CallCtx &x)
{
    const CallPlan &plan = *x.plan;
    hipStream_t s = x.s_main;
    if (plan.fused) {
        if (e->profiling) {
            int rc = take_event_pair(e, x, s);
            if (rc != IQD_OK) return rc;
        }
        HIP_LAUNCH(e, launch_mixed_stream(x.mix, x.fused_mag, x.gated, plan.mix_wgs, s));
        e->stats.mixed_launches++;
        MixedTailArgs mt{};
        for (int f = 0; f < FAM_COUNT; f++) mt.a[f] = x.mix.a[f];
        mt.sa = x.mix.sa;
        HIP_LAUNCH(e, launch_mixed_tail(mt, s));
        if (e->profiling) x.evp_open = true;   // (the timed region: the pipelines AND every follower - fix-up, DC passes, tails, the squelch launch)
        if (x.mix.a[FAM_WBFM].wg_count) {   // repair check, state commit and tail of the WBFM channels: in the squelch launch below
            x.tail_a = x.mix.a[FAM_WBFM];
            x.tail_f = FAM_WBFM;
            x.tail_pending = true;
        }
        // (the WBFM family's epoch mirror: a fused plan has no gain change in reach, nothing to age)
    }
    for (int k = 1; k < 4; k++)
        if (x.lane_used[k]) {
            HIP_TRY(e, hipEventRecord(e->fam_join[k - 1], e->fam_stream[k - 1]));
            HIP_TRY(e, hipStreamWaitEvent(s, e->fam_join[k - 1], 0));
        }

    // channels in mode None still report their magnitudes
    if (x.fused_mag && !e->h_lists[FAM_COUNT].empty())
        HIP_LAUNCH(e, launch_magnitude((const uint8_t *)x.iq_dev, x.bytes_per_ch, e->lists[FAM_COUNT].as<uint32_t>(),
                                    (uint32_t)e->h_lists[FAM_COUNT].size(), x.call_bs, x.n_blocks, e->mag_sums.as<uint32_t>(), s));
    if (!x.gated && !e->demod_bypass && (x.want_mag || x.pcm_count_dev || x.signal_present_dev || e->trace_on)) {
        x.q.zero_sums_after = x.any_agc ? 0u : 1u;   // (a running AGC reads them again in the tracking pass)
        HIP_LAUNCH(e, launch_squelch(x.q, true, s, x.tail_pending ? &x.tail_a : nullptr, x.tail_f, x.tail_pending && x.tail_dc));
        x.tail_pending = false;
        if (x.q.zero_sums_after) e->mag_sums_zero = (size_t)x.n_ch * x.n_blocks;
    }
    if (x.tail_pending) {   // (no squelch launch to ride in)
        if (x.tail_f == FAM_WBFM) HIP_LAUNCH(e, launch_wbfm_repair(x.tail_a, x.gated, s));
        else HIP_LAUNCH(e, launch_tail_update(x.tail_a, x.tail_f, s));
    }

    if (x.evp_open) {
        HIP_TRY(e, hipEventRecord(x.evp.second, s));
        e->ev_pending.push_back(x.evp);
    }
    if (x.pre_set >= 0) {   // this call's pipelines are the last readers of its buffer set
        HIP_TRY(e, hipEventRecord(e->ev_chain_done[x.pre_set], s));
        e->chain_pending[x.pre_set] = true;
    } else if (e->pre_stream && !e->demod_bypass) {
        // a squelch pass on the main stream - this call's closing pass, or a gated call's inline decisions (those left
        // no event behind) - : the next pre-pass continues from it
        HIP_TRY(e, hipEventRecord(e->ev_main_decisions, s));
        e->decisions_on_main = true;
    }
    if (x.epoch_report_now) {   // behind everything this call queued (the side streams have joined)
        HIP_LAUNCH(e, launch_write_word(e->h_epoch_report + 1, e->accept_seq, s));
        e->epoch_report_pending = true;
        e->epoch_report_seq = e->accept_seq;
    }
    e->stats.accepts++;
    e->stats.samples += (uint64_t)x.vlen * x.n_ch;
    return IQD_OK;
}

int iqd_accept_iq_device(iqd_t *e, uint32_t first_ch, uint32_t n_ch, const void *iq_dev, size_t bytes_per_ch,
                         void *pcm_dev, void *pcm_count_dev, void *magnitude_dev, void *signal_present_dev)
{
    if (!range_ok(e, first_ch, n_ch) || !iq_dev || !pcm_dev) return e ? e->fail(IQD_EINVAL, "bad channel range or NULL buffer") : IQD_EINVAL;
    const uint32_t call_bb = call_block_bytes(e, first_ch, n_ch, bytes_per_ch);
    if (!call_bb) return e->fail(IQD_EINVAL, IQD_LEN_MSG, bytes_per_ch, e->block_bytes);
    if (bytes_per_ch / 2 > 0x7fff0000ull) return e->fail(IQD_EINVAL, "bytes_per_ch too large");
    if (((uintptr_t)iq_dev & 15) != 0) return e->fail(IQD_EINVAL, "iq_dev must be 16-byte aligned");
    (void)hipSetDevice(e->device);

    CallCtx x;
    x.first_ch = first_ch; x.n_ch = n_ch; x.iq_dev = iq_dev; x.bytes_per_ch = bytes_per_ch;
    x.pcm_dev = pcm_dev; x.pcm_count_dev = pcm_count_dev; x.magnitude_dev = magnitude_dev; x.signal_present_dev = signal_present_dev;
    x.call_bb = call_bb;
    x.call_bs = call_bb / 2;
    x.n_blocks = (uint32_t)(bytes_per_ch / call_bb);
    x.vlen = (uint32_t)(bytes_per_ch / 2);
    x.s_main = e->stream;
    int rc = prepare_call(e, x);
    if (rc != IQD_OK) return rc;
    if (e->demod_bypass) {   // {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData: nothing of the processor's squelch path runs or moves
        x.gated = x.any_agc = false;
        x.pcm_count_dev = x.magnitude_dev = x.signal_present_dev = nullptr;
    }
    x.want_mag = !e->demod_bypass && (x.gated || x.any_agc || !(e->flags & IQD_F_NO_MAGNITUDE) || x.magnitude_dev);
    x.pre_overlap = x.gated && (e->flags & IQD_F_PREPASS_OVERLAP) && !e->trace_on && !e->in_host_path;
    x.fused_mag = x.want_mag && !x.gated;

    // the plan: decided before anything of the call is queued, kept while the call's shape repeats
    CallShape shape;
    describe_call(e, x, shape);
    if (!e->plan_valid || !same_shape(shape, e->plan_shape)) {
        plan_call(e->knobs, shape, e->plan);
        e->plan_shape = shape;
        e->plan_valid = true;
    }
    x.plan = &e->plan;
    const CallPlan &plan = e->plan;

    rc = queue_prepass(e, x);
    if (rc != IQD_OK) return rc;
    if (plan.forked) {
        if (!e->fam_fork) {
            int lo = 0, hi = 0;   // numerically lower = higher priority
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(e, hipEventCreateWithFlags(&e->fam_fork, hipEventDisableTiming));
            for (int k = 0; k < 3; k++) {
                HIP_TRY(e, hipStreamCreateWithPriority(&e->fam_stream[k], hipStreamNonBlocking, k == 0 ? hi : lo));
                HIP_TRY(e, hipEventCreateWithFlags(&e->fam_join[k], hipEventDisableTiming));
            }
        }
        // the detector-stream buffers before the fork (an allocation that grows one must not free what a side stream still reads)
        if (!e->h_lists[FAM_AM].empty()) HIP_TRY(e, e->base8k.ensure((size_t)n_ch * x.base.pcm_stride * sizeof(int32_t)));
        if (!e->h_lists[FAM_SSB].empty()) HIP_TRY(e, e->base8k2.ensure((size_t)n_ch * x.base.pcm_stride * sizeof(int32_t)));
        // several families as kernels of their own on side streams: the timed region opens in front of the fork and closes behind
        // the join and the closing launch (queue_commit), like every other arrangement's - every launch of the step (ADVICE r5:
        // it used to hold the first family's launches only, so kernel_ms was not comparable across arrangements)
        if (!plan.fused && e->profiling) {
            rc = take_event_pair(e, x, x.s_main);
            if (rc != IQD_OK) return rc;
            x.timed = true;
            x.evp_open = true;
        }
        if (!plan.fused) HIP_TRY(e, hipEventRecord(e->fam_fork, x.s_main));
    }
    for (int oi = 0; oi < FAM_COUNT; oi++) {
        const int f = plan.order[oi];
        if (!plan.fam[f].present) continue;
        rc = queue_family(e, x, f);
        if (rc != IQD_OK) return rc;
    }
    return queue_commit(e, x);
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
    // The channels as this call needs them; what the caller had set comes back afterwards.  A channel that already is in the
    // wanted mode with selector 0 is left alone: a bare demodulator object driven call after call (demod.cc, one block per
    // call) then costs no parameter upload, no list rebuild and no tail rewrite at all.
    std::vector<int32_t> mode0(n_ch), rot0(n_ch), mode1(n_ch);
    std::vector<uint32_t> mgen(n_ch), rgen(n_ch);   // the setters' generation counters as this call found them
    bool touched = false;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->mode_gen.size() < e->h_params.size()) e->mode_gen.resize(e->h_params.size(), 0u);
        if (e->rot_gen.size() < e->h_params.size()) e->rot_gen.resize(e->h_params.size(), 0u);
        for (uint32_t c = 0; c < n_ch; c++) {
            mgen[c] = e->mode_gen[first_ch + c];
            rgen[c] = e->rot_gen[first_ch + c];
            ChanParams &p = e->h_params[first_ch + c];
            mode0[c] = p.mode;
            rot0[c] = p.rotation;
            mode1[c] = demod == IQD_DEMOD_AM ? IQD_MODE_AM : demod == IQD_DEMOD_FM ? IQD_MODE_FM : demod == IQD_DEMOD_WBFM ? IQD_MODE_WBFM
                                             : (p.ssb_lsb ? IQD_MODE_LSB : IQD_MODE_USB);
            touched = touched || p.mode != mode1[c] || p.rotation != 0;
            p.mode = mode1[c];
            p.rotation = 0;
        }
        if (touched) e->params_dirty = e->lists_dirty = true;
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
    if (touched) {
        // only what this call wrote and no setter has written since - by the setters' generation counters, not by value: a
        // set_mode to the demodulator's own mode or a set_rotation(0) issued during the call keeps its value too
        std::lock_guard<std::mutex> lk(e->mu);
        for (uint32_t c = 0; c < n_ch; c++) {
            ChanParams &p = e->h_params[first_ch + c];
            if (e->mode_gen[first_ch + c] == mgen[c]) p.mode = mode0[c];
            if (e->rot_gen[first_ch + c] == rgen[c]) p.rotation = rot0[c];
        }
        e->params_dirty = e->lists_dirty = true;
    }
    return rc;
}

}  // extern "C"
