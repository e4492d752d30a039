// Several demodulator families' streaming pipelines in one launch (configs[3]: AM / FM / WBFM / LSB / USB channels in one
// call).  Each family's persistent workgroups used to be a kernel of their own on a stream of their own, held to a share
// of the CUs (iqd_host.cpp: plan_family_shares): a fork and a join of four streams per call (~40 us of event latency),
// shares in steps of 8 CUs (one per XCD, or a workgroup waits for a whole kernel behind another family's), 16 CUs kept
// free for the same reason.  Here the families are ranges of ONE grid: see MixedStreamArgs (iqd_stream.h).
//
// The workgroup bodies are the stream kernels' own (iqd_stream.hip, iqd_stream2.hip), compiled into this translation unit
// a second time; a workgroup runs exactly one of them.
#define IQD_STREAM_BODIES_ONLY 1
#define IQD_ST_NO_GROUPS 1        // (its WBFM family has one rotation selector: iqd_engine.cpp plans it so)
#include "iqd_stream.hip"
#include "iqd_stream2.hip"
#include "iqd_stream_mixed.h"

#ifndef IQD_MIXED_TIMING
#define IQD_MIXED_TIMING 0
#endif

namespace iqd {

// (Every family's launch descriptor is a kernel argument of its own: as members of one 2.6 KB argument the compiler copied
// the lot to scratch and read tap tables from there inside the piece loops.)
// GATED: the call is squelch-gated (a pre-pass took the magnitudes and the decisions; the pipelines walk each channel's
// open blocks).
template <bool MAG, bool GATED>
__global__ __launch_bounds__(ST_THREADS, 4) void mixed_stream_kernel(const ChainLaunch a_wbfm, const StreamArgs sa, const int32_t wbfm_rot,
                                                                     const ChainLaunch a_fm, const D4Args d_fm,
                                                                     const ChainLaunch a_ssb, const D4Args d_ssb,
                                                                     const ChainLaunch a_am, const D4Args d_am)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t mx_lds[];
    const uint32_t b = blockIdx.x;
#if IQD_MIXED_TIMING   // diagnostic build: stamps[16 + workgroup] = its run time in 10 ns ticks, family in the top byte, stamps[528 + workgroup] = when it started (tools/mixed_probe.py)
    const unsigned long long t_wg0 = wall_clock64();
    const int fam_of_wg = b - a_wbfm.wg_first < a_wbfm.wg_count ? FAM_WBFM : b - a_fm.wg_first < a_fm.wg_count ? FAM_FM : b - a_ssb.wg_first < a_ssb.wg_count ? FAM_SSB : FAM_AM;
#endif
    // (WBFM first: its workgroups start with 134 KB of table to fetch)
    if (b - a_wbfm.wg_first < a_wbfm.wg_count) {
        if (wbfm_rot == 0) wbfm_stream_body<0, MAG, false, GATED>(a_wbfm, sa, mx_lds);
        else if (wbfm_rot > 0) wbfm_stream_body<1, MAG, false, GATED>(a_wbfm, sa, mx_lds);
        else wbfm_stream_body<-1, MAG, false, GATED>(a_wbfm, sa, mx_lds);
    } else if (b - a_fm.wg_first < a_fm.wg_count) {
        d4_stream_body<D4_FM, MAG, GATED>(a_fm, d_fm, mx_lds);
    } else if (b - a_ssb.wg_first < a_ssb.wg_count) {
        d4_stream_body<D4_SSB, MAG, GATED>(a_ssb, d_ssb, mx_lds);
    } else if (b - a_am.wg_first < a_am.wg_count) {
        d4_stream_body<D4_AM, MAG, GATED>(a_am, d_am, mx_lds);
    }
#if IQD_MIXED_TIMING
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long dt = wall_clock64() - t_wg0;
        a_wbfm.stamps[16 + b] = dt | ((unsigned long long)fam_of_wg << 56);
        a_wbfm.stamps[16 + 512 + b] = t_wg0;
    }
#endif
}

constexpr int MX_LDS_BYTES = ST_LDS_BYTES > D4_LDS_BYTES ? ST_LDS_BYTES : D4_LDS_BYTES;

typedef void (*MxKernel)(const ChainLaunch, const StreamArgs, const int32_t, const ChainLaunch, const D4Args, const ChainLaunch, const D4Args,
                         const ChainLaunch, const D4Args);
// [0: no magnitudes, 1: squelch magnitudes in the kernel, 2: squelch-gated call]
static const MxKernel mx_kernels[3] = {mixed_stream_kernel<false, false>, mixed_stream_kernel<true, false>, mixed_stream_kernel<false, true>};

hipError_t init_mixed_stream_kernels()
{
    for (int k = 0; k < 3; k++) {
        const hipError_t e = hipFuncSetAttribute((const void *)mx_kernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_mixed_stream(const MixedStreamArgs &m, bool mag, bool gated, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(mx_kernels[gated ? 2 : (mag ? 1 : 0)], dim3(grid), dim3(ST_THREADS), MX_LDS_BYTES, s, m.a[FAM_WBFM], m.sa, m.wbfm_rot,
                       m.a[FAM_FM], m.d4[FAM_FM], m.a[FAM_SSB], m.d4[FAM_SSB], m.a[FAM_AM], m.d4[FAM_AM]);
    return hipGetLastError();
}

}  // namespace iqd
