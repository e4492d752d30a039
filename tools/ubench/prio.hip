// Diagnostic tool, not product: four waves per SIMD issue vector instructions more slowly than three (tools/ubench/classes.hip:
// 1.98 against 1.52 cycles per wave-instruction).  Does giving the four waves of a SIMD different priorities (s_setprio)
// change that, and what does a SIMD's oldest wave get when all four are hungry?
// Build: hipcc --offload-arch=gfx950 -O3 -o prio prio.hip      Run: ./prio
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// MODE 0: v_add_f32 chains; 1: v_dot2_i32_i16 chains.  PRIO 0: none; 1: wave w of the workgroup takes priority (w >> 2) & 3
// (the four waves of a SIMD - w, w + 4, w + 8, w + 12 - all differ); 2: only waves 0..3 (the oldest of each SIMD) raised to 3.
template <int MODE, int PRIO>
__global__ void k(uint32_t *out, unsigned long long *cyc, int iters, uint32_t b, uint32_t c)
{
    uint32_t a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 977u + i * 131u + b;
    const int w = threadIdx.x >> 6;
    if (PRIO == 1) {
        switch ((w >> 2) & 3) {
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
        }
    } else if (PRIO == 2) {
        if (w < 4) __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b), "v"(c));
                else asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = memtime();
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE, int PRIO>
static void run(uint32_t *d_out, unsigned long long *d_cyc, int ncu, const char *what)
{
    const int iters = 1500;
    printf("%-44s", what);
    for (int wps = 4; wps >= 3; wps--) {
        const int threads = 256 * wps, wpw = threads / 64, nw = ncu * wpw;
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, 50, 0x01020304u, 0x00000080u);
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, iters, 0x01020304u, 0x00000080u);
        hipDeviceSynchronize();
        std::vector<unsigned long long> cyc(nw);
        hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost);
        // per wave age (w >> 2): the median over the workgroups of that age's run time, in cycles per own instruction
        printf("  %d w/SIMD:", wps);
        double slowest = 0;
        for (int age = 0; age < wps; age++) {
            std::vector<unsigned long long> v;
            for (int wg = 0; wg < ncu; wg++)
                for (int s = 0; s < 4; s++) v.push_back(cyc[wg * wpw + age * 4 + s]);
            std::sort(v.begin(), v.end());
            const double per = (double)v[v.size() / 2] / (iters * 64.0);
            printf(" %5.2f", per);
            slowest = std::max(slowest, per);
        }
        printf("  -> %.2f per SIMD", slowest / wps);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    uint32_t *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 4u << 20);
    hipMalloc(&d_cyc, 1u << 20);
    printf("cycles per own instruction of the waves by age (oldest first), then the SIMD's cycles per wave-instruction\n");
    run<0, 0>(d_out, d_cyc, ncu, "v_add_f32, no priorities");
    run<0, 1>(d_out, d_cyc, ncu, "v_add_f32, priority = age");
    run<0, 2>(d_out, d_cyc, ncu, "v_add_f32, oldest wave of each SIMD at 3");
    run<1, 0>(d_out, d_cyc, ncu, "v_dot2_i32_i16, no priorities");
    run<1, 1>(d_out, d_cyc, ncu, "v_dot2_i32_i16, priority = age");
    run<1, 2>(d_out, d_cyc, ncu, "v_dot2_i32_i16, oldest wave of each SIMD at 3");
    return 0;
}
