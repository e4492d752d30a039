// How fast can 49152 interleaved 64-byte-at-a-time streams be read?  (The access pattern of the streaming kernels'
// P waves: 256 workgroups x 12 waves x 16 segments, a lane quartet takes 64 consecutive bytes of its segment per piece.)
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/streams tools/ubench/streams.hip && /tmp/streams
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int BYTES_PER_LANE, int AHEAD>
__global__ __launch_bounds__(768) void streams(const uint8_t *in, uint32_t *out, uint32_t seg_bytes, int n_pieces)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t seg = (blockIdx.x * 12 + wave) * 16 + c;
    const uint8_t *p = in + (size_t)seg * seg_bytes + g * BYTES_PER_LANE;
    constexpr int PIECE = 4 * BYTES_PER_LANE;   // bytes of a segment per piece
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[AHEAD][BYTES_PER_LANE / 16];
#pragma unroll
    for (int j = 0; j < AHEAD; j++)
#pragma unroll
        for (int k = 0; k < BYTES_PER_LANE / 16; k++) buf[j][k] = *(const uint4 *)(p + j * PIECE + 16 * k);
    for (int q = 0; q < n_pieces; q += AHEAD) {
#pragma unroll
        for (int j = 0; j < AHEAD; j++) {
#pragma unroll
            for (int k = 0; k < BYTES_PER_LANE / 16; k++) {
                const uint4 v = buf[j][k];
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                int nq = q + j + AHEAD;
                nq = nq < n_pieces ? nq : n_pieces - 1;
                buf[j][k] = *(const uint4 *)(p + (size_t)nq * PIECE + 16 * k);
            }
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// 4 segments per load instruction, 256 contiguous bytes each (16 lanes x 16 B); a wave still owns 16 segments and
// visits them in 4 groups per step of 256 bytes
template <int AHEAD>
__global__ __launch_bounds__(768) void streams4(const uint8_t *in, uint32_t *out, uint32_t seg_bytes, int n_steps)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, s4 = lane >> 4, l16 = lane & 15;
    const uint32_t seg0 = (blockIdx.x * 12 + wave) * 16;
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[AHEAD][4];
    const uint8_t *p[4];
#pragma unroll
    for (int grp = 0; grp < 4; grp++) p[grp] = in + (size_t)(seg0 + 4 * grp + s4) * seg_bytes + 16 * l16;
#pragma unroll
    for (int j = 0; j < AHEAD; j++)
#pragma unroll
        for (int grp = 0; grp < 4; grp++) buf[j][grp] = *(const uint4 *)(p[grp] + j * 256);
    for (int q = 0; q < n_steps; q += AHEAD) {
#pragma unroll
        for (int j = 0; j < AHEAD; j++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const uint4 v = buf[j][grp];
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                int nq = q + j + AHEAD;
                nq = nq < n_steps ? nq : n_steps - 1;
                buf[j][grp] = *(const uint4 *)(p[grp] + (size_t)nq * 256);
            }
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int A>
static void run4(const uint8_t *d, uint32_t *o, size_t total, const char *what)
{
    const uint32_t nseg = 256 * 12 * 16, seg_bytes = 11264;
    total = (size_t)seg_bytes * nseg;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((streams4<A>), dim3(256), dim3(768), 0, 0, d, o, seg_bytes, (int)(seg_bytes / 256));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", what, best, total / (best * 1e-3) / 1e12);
}

// SEGS segments per load instruction, 1024 / SEGS contiguous bytes each; a wave owns 16 segments
template <int SEGS, int AHEAD>
__global__ __launch_bounds__(768) void streams_n(const uint8_t *in, uint32_t *out, uint32_t seg_bytes, int n_steps)
{
    constexpr int LANES = 64 / SEGS, CHUNK = 16 * LANES, GROUPS = 16 / SEGS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sx = lane / LANES, lx = lane % LANES;
    const uint32_t seg0 = (blockIdx.x * 12 + wave) * 16;
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[AHEAD][GROUPS];
    const uint8_t *p[GROUPS];
#pragma unroll
    for (int grp = 0; grp < GROUPS; grp++) p[grp] = in + (size_t)(seg0 + SEGS * grp + sx) * seg_bytes + 16 * lx;
#pragma unroll
    for (int j = 0; j < AHEAD; j++)
#pragma unroll
        for (int grp = 0; grp < GROUPS; grp++) buf[j][grp] = *(const uint4 *)(p[grp] + j * CHUNK);
    for (int q = 0; q < n_steps; q += AHEAD) {
#pragma unroll
        for (int j = 0; j < AHEAD; j++) {
#pragma unroll
            for (int grp = 0; grp < GROUPS; grp++) {
                const uint4 v = buf[j][grp];
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                int nq = q + j + AHEAD;
                nq = nq < n_steps ? nq : n_steps - 1;
                buf[j][grp] = *(const uint4 *)(p[grp] + (size_t)nq * CHUNK);
            }
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int SEGS, int A>
static void run_n(const uint8_t *d, uint32_t *o, const char *what)
{
    const uint32_t nseg = 256 * 12 * 16, seg_bytes = 11264;
    const size_t total = (size_t)seg_bytes * nseg;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((streams_n<SEGS, A>), dim3(256), dim3(768), 0, 0, d, o, seg_bytes, (int)(seg_bytes / (1024 / SEGS)));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", what, best, total / (best * 1e-3) / 1e12);
}


// The P waves' pattern with the loads of BURST consecutive pieces of a segment issued back to back (BURST x 64 bytes per
// segment reach the memory system together), SETS such bursts in flight, and SLEEP x 64 cycles of "work" per piece
// (round 5: does the DRAM see longer runs per stream if the pieces are asked for in bursts rather than one per piece time?)
template <int BURST, int SETS, int SLEEP>
__global__ __launch_bounds__(768) void streams_burst(const uint8_t *in, uint32_t *out, uint32_t seg_bytes, int n_pieces)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t seg = (blockIdx.x * 12 + wave) * 16 + c;
    const uint8_t *p = in + (size_t)seg * seg_bytes + g * 16;
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[SETS][BURST];
#pragma unroll
    for (int s = 0; s < SETS; s++)
#pragma unroll
        for (int j = 0; j < BURST; j++) buf[s][j] = *(const uint4 *)(p + (s * BURST + j) * 64);
    for (int q = 0; q < n_pieces; q += SETS * BURST) {
#pragma unroll
        for (int s = 0; s < SETS; s++) {
            uint4 v[BURST];
#pragma unroll
            for (int j = 0; j < BURST; j++) {
                v[j] = buf[s][j];
                acc.x ^= v[j].x; acc.y += v[j].y; acc.z ^= v[j].z; acc.w += v[j].w;
                if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < BURST; j++) {
                int nq = q + (s + SETS) * BURST + j;
                nq = nq < n_pieces ? nq : n_pieces - 1;
                buf[s][j] = *(const uint4 *)(p + (size_t)nq * 64);
            }
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int BURST, int SETS, int SLEEP>
static void run_burst(const uint8_t *d, uint32_t *o, const char *what, uint32_t seg_bytes = 19712)   // (9472 + 384) samples
{
    const uint32_t nseg = 256 * 12 * 16;
    const size_t total = (size_t)seg_bytes * nseg;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((streams_burst<BURST, SETS, SLEEP>), dim3(256), dim3(768), 0, 0, d, o, seg_bytes, (int)(seg_bytes / 64));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", what, best, total / (best * 1e-3) / 1e12);
}

// the same bytes, fully coalesced: a workgroup streams through its contiguous share, a wave takes 1 KiB at a time
template <int AHEAD>
__global__ __launch_bounds__(768) void coalesced(const uint8_t *in, uint32_t *out, uint32_t wg_bytes)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint8_t *p = in + (size_t)blockIdx.x * wg_bytes + wave * 1024 + lane * 16;
    const int n = wg_bytes / (12 * 1024);
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[AHEAD];
#pragma unroll
    for (int j = 0; j < AHEAD; j++) buf[j] = *(const uint4 *)(p + (size_t)j * 12 * 1024);
    for (int q = 0; q < n; q += AHEAD) {
#pragma unroll
        for (int j = 0; j < AHEAD; j++) {
            const uint4 v = buf[j];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
            int nq = q + j + AHEAD;
            nq = nq < n ? nq : n - 1;
            buf[j] = *(const uint4 *)(p + (size_t)nq * 12 * 1024);
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int A>
static void run_coalesced(const uint8_t *d, uint32_t *o, size_t total, int wgs, const char *what)
{
    const uint32_t wg_bytes = (uint32_t)(total / wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((coalesced<A>), dim3(wgs), dim3(768), 0, 0, d, o, wg_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", what, best, total / (best * 1e-3) / 1e12);
}

template <int B, int A>
static void run(const uint8_t *d, uint32_t *o, size_t total, const char *what, uint32_t seg_override = 0)
{
    const uint32_t nseg = 256 * 12 * 16;
    const uint32_t seg_bytes = seg_override ? seg_override : (uint32_t)(total / nseg);
    total = (size_t)seg_bytes * nseg;
    const int n_pieces = seg_bytes / (4 * B);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((streams<B, A>), dim3(256), dim3(768), 0, 0, d, o, seg_bytes, n_pieces);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", what, best, total / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t total = (size_t)1 << 29;   // 2^28 samples
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, 2 * total); hipMalloc(&o, 2048 * 768 * 4);
    hipMemset(d, 1, 2 * total);
    run<16, 1>(d, o, total, "64 B per segment per piece, 1 ahead");
    run<16, 2>(d, o, total, "64 B per segment per piece, 2 ahead");
    run<16, 4>(d, o, total, "64 B per segment per piece, 4 ahead");
    run<16, 8>(d, o, total, "64 B per segment per piece, 8 ahead");
    run<32, 2>(d, o, total, "128 B per segment per piece, 2 ahead");
    run<32, 4>(d, o, total, "128 B per segment per piece, 4 ahead");
    run<64, 2>(d, o, total, "256 B per segment per piece, 2 ahead");
    for (uint32_t sb : {8192u, 10240u, 10496u, 10752u, 11008u, 11264u, 11520u, 11776u, 12032u, 12288u, 16384u, 11264u + 64u, 11264u + 128u}) {
        char what[64];
        snprintf(what, sizeof what, "64 B pieces, segments %u bytes apart", sb);
        run<16, 4>(d, o, total, what, sb);
    }
    run_n<16, 2>(d, o, "16 segments x 64 B per load, 2 ahead");
    run_n<8, 1>(d, o, "8 segments x 128 B per load, 1 ahead");
    run_n<8, 2>(d, o, "8 segments x 128 B per load, 2 ahead");
    run_n<4, 2>(d, o, "4 segments x 256 B per load, 2 ahead");
    run_n<2, 1>(d, o, "2 segments x 512 B per load, 1 ahead");
    run4<1>(d, o, total, "4 segments x 256 B per load, 1 step ahead");
    run4<2>(d, o, total, "4 segments x 256 B per load, 2 steps ahead");
    run_burst<1, 4, 0>(d, o, "burst 1 x 4 sets, no work, 11264 B segments", 11264);
    run_burst<1, 4, 0>(d, o, "burst 1 x 4 sets, no work, 15360 B segments", 15360);
    run_burst<1, 4, 0>(d, o, "burst 1 x 4 sets, no work, 18944 B segments", 18944);
    run<16, 4>(d, o, total, "64 B pieces, 4 ahead, segments 19712 bytes", 19712);
    run<16, 4>(d, o, total, "64 B pieces, 4 ahead, segments 18944 bytes", 18944);
    run<16, 8>(d, o, total, "64 B pieces, 8 ahead, segments 19712 bytes", 19712);
    run_burst<1, 4, 0>(d, o, "burst 1 x 4 sets, no work");
    run_burst<1, 8, 0>(d, o, "burst 1 x 8 sets, no work");
    run_burst<2, 2, 0>(d, o, "burst 2 x 2 sets, no work");
    run_burst<2, 4, 0>(d, o, "burst 2 x 4 sets, no work");
    run_burst<4, 1, 0>(d, o, "burst 4 x 1 set, no work");
    run_burst<4, 2, 0>(d, o, "burst 4 x 2 sets, no work");
    run_burst<8, 1, 0>(d, o, "burst 8 x 1 set, no work");
    run_burst<1, 4, 12>(d, o, "burst 1 x 4 sets, 768 cycles per piece");
    run_burst<1, 8, 12>(d, o, "burst 1 x 8 sets, 768 cycles per piece");
    run_burst<2, 2, 12>(d, o, "burst 2 x 2 sets, 768 cycles per piece");
    run_burst<2, 4, 12>(d, o, "burst 2 x 4 sets, 768 cycles per piece");
    run_burst<4, 1, 12>(d, o, "burst 4 x 1 set, 768 cycles per piece");
    run_burst<4, 2, 12>(d, o, "burst 4 x 2 sets, 768 cycles per piece");
    run_burst<8, 1, 12>(d, o, "burst 8 x 1 set, 768 cycles per piece");
    run_burst<1, 4, 16>(d, o, "burst 1 x 4 sets, 1024 cycles per piece");
    run_burst<4, 2, 16>(d, o, "burst 4 x 2 sets, 1024 cycles per piece");
    run_burst<8, 1, 16>(d, o, "burst 8 x 1 set, 1024 cycles per piece");
    run_coalesced<4>(d, o, total, 256, "coalesced, 256 workgroups, 4 ahead");
    run_coalesced<8>(d, o, total, 256, "coalesced, 256 workgroups, 8 ahead");
    run_coalesced<4>(d, o, total, 1024, "coalesced, 1024 workgroups, 4 ahead");
    run_coalesced<8>(d, o, total, 2048, "coalesced, 2048 workgroups, 8 ahead");
    return 0;
}
