// Launch descriptors of the launches that run several demodulator families' streaming pipelines side by side
// (iqd_stream_mixed.hip) and finish them (iqd_kernels.hip: mixed_tail_kernel).
#pragma once
#include "iqd_kernels.h"
#include "iqd_stream.h"

namespace iqd {

// Several families' streaming pipelines in ONE launch (iqd_stream_mixed.hip): family f owns the workgroups
// [a[f].wg_first, a[f].wg_first + a[f].wg_count) of the grid.  Workgroup i of a launch goes to XCD i % 8, so every family's
// contiguous range is spread evenly over the XCDs whatever its size, each workgroup has a CU to itself (the grid is at most
// the CU count), and the families end together when their shares are in proportion to their work - no fork and join of
// streams, no share granularity, no CUs kept free for workgroups of the other kernels.
struct MixedStreamArgs {
    ChainLaunch a[FAM_COUNT];    // by family (FAM_AM, FAM_FM, FAM_WBFM, FAM_SSB); wg_count 0: the family is not in the call
    StreamArgs sa;               // WBFM
    D4Args d4[FAM_COUNT];        // AM, FM, SSB ([FAM_WBFM] unused)
    int32_t wbfm_rot;            // the WBFM channels' common rotation selector
};
// What follows it, again in one launch (iqd_kernels.hip: mixed_tail_kernel, 256-thread workgroups with one role each): the
// WBFM boundary fix-up, the DC-removal pass of the AM and SSB channels, every family's tail update.  Families with
// wg_count 0 are not in the call.
struct MixedTailArgs {
    ChainLaunch a[FAM_COUNT];
    StreamArgs sa;
};
hipError_t init_mixed_stream_kernels();
hipError_t launch_mixed_stream(const MixedStreamArgs &m, bool mag, bool gated, uint32_t grid, hipStream_t s);
hipError_t launch_mixed_tail(const MixedTailArgs &m, hipStream_t s);

}  // namespace iqd
