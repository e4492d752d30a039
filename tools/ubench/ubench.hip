// Microbenchmarks that settle design questions of the WBFM chain on gfx950 (diagnostic tool, not product):
//   1. vector-instruction issue rate per SIMD at 1/2/4 waves per SIMD, per opcode
//   2. dependent mul->sub latency (the de-emphasis recurrence)
//   3. v_mfma_i32_16x16x64_i8: operand/result lane maps checked against a host product; issue rate
//   4. ds_read_b32 gathers from a 134 KB LDS table: uniform, patch and ring address patterns
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip     Run: ./ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

enum Op {
    OP_FMA, OP_PK_FMA, OP_PK_MUL, OP_PK_ADD, OP_MUL, OP_DOT4, OP_PERM, OP_MQSAD_PK, OP_MSAD, OP_SAD, OP_SDWA_SUB,
    OP_XAD, OP_PK_MAX_U16, OP_PK_ADD_U16, OP_RNDNE, OP_CVT_I32, OP_BFI, OP_MAD_U24, OP_LSHL_ADD, OP_AND_OR, OP_CNDMASK,
    OP_ALIGNBIT, OP_DPP_ROWSHR, OP_PERMLANE32, OP_BPERMUTE, OP_DOT2, OP_MED3, OP_PK_LSHR_U16, OP_ADD_U32, OP_COUNT
};
static const char *op_names[] = {
    "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_dot4_i32_i8", "v_perm_b32",
    "v_mqsad_pk_u16_u8", "v_msad_u8", "v_sad_u8", "v_sub_u32_sdwa(byte)", "v_xad_u32", "v_pk_max_u16", "v_pk_add_u16",
    "v_rndne_f32", "v_cvt_i32_f32", "v_bfi_b32", "v_mad_u32_u24", "v_lshl_add_u32", "v_and_or_b32", "v_cndmask_b32",
    "v_alignbit_b32", "v_mov_dpp row_shr:1", "v_permlane32_swap", "ds_bpermute_b32", "v_dot2_i32_i16", "v_med3_i32",
    "v_pk_lshrrev_b16", "v_add_u32"};

template <int OP>
__device__ __forceinline__ void step(uint32_t (&a)[8], unsigned long long (&w)[4], uint32_t b, uint32_t c)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (OP == OP_DOT4) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_DOT2) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_MSAD) asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_SAD) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_SDWA_SUB) asm volatile("v_sub_u32_sdwa %0, %1, %0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a[k]) : "v"(b));
        if (OP == OP_XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (OP == OP_PK_LSHR_U16) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(a[k]));
        if (OP == OP_RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[k]));
        if (OP == OP_CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
        if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a[k]) : "v"(b));
        if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(0x00ff00ff00ff00ffull));
        if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 14" : "+v"(a[k]) : "v"(b));
        if (OP == OP_DPP_ROWSHR) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        if (OP == OP_MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (OP == OP_BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
    }
    if (OP == OP_PERMLANE32) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[k + 1]));
#pragma unroll
        for (int k = 0; k < 8; k += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[k + 1]));
    }
    if (OP == OP_BPERMUTE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[k]) : "v"(w[(k + 1) & 3]));
        if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[k]) : "v"(w[(k + 1) & 3]));
        if (OP == OP_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[k]) : "v"(w[(k + 1) & 3]));
        if (OP == OP_MQSAD_PK) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(w[k]) : "v"(w[(k + 1) & 3]), "v"(b));
    }
    if (OP == OP_PK_FMA || OP == OP_PK_MUL || OP == OP_PK_ADD || OP == OP_MQSAD_PK) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[k]) : "v"(w[(k + 2) & 3]));
            if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[k]) : "v"(w[(k + 2) & 3]));
            if (OP == OP_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[k]) : "v"(w[(k + 2) & 3]));
            if (OP == OP_MQSAD_PK) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(w[k]) : "v"(w[(k + 2) & 3]), "v"(b));
        }
    }
}

// every wave: `iters` x 8 instructions; cycles per wave into cyc[]
template <int OP>
__global__ void k_rate(uint32_t *out, unsigned long long *cyc, int iters, uint32_t b, uint32_t c)
{
    uint32_t a[8];
    unsigned long long w[4];
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 977u + k * 131u + b;
    for (int k = 0; k < 4; k++) w[k] = ((unsigned long long)__float_as_uint(1.0f + k * 0.001f) << 32) | __float_as_uint(0.999f);
    __syncthreads();
    const unsigned long long r0 = realtime();
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) step<OP>(a, w, b, c);
    const unsigned long long t1 = memtime();
    const unsigned long long r1 = realtime();
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) s += a[k];
    for (int k = 0; k < 4; k++) s += (uint32_t)w[k] + (uint32_t)(w[k] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((threadIdx.x & 63) == 0) cyc[65536 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = r1 - r0;
}

// dependent chain: y = t - a1*y, one wave, N steps, optionally K independent chains interleaved
template <int CH>
__global__ void k_chain(float *out, unsigned long long *cyc, int iters, float a1, float t)
{
    float y[CH];
    for (int k = 0; k < CH; k++) y[k] = threadIdx.x * 0.001f + k;
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float r;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a1), "v"(y[k]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(y[k]) : "v"(t), "v"(r));
            }
        }
    }
    const unsigned long long t1 = memtime();
    float s = 0;
    for (int k = 0; k < CH; k++) s += y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ---- MFMA ---------------------------------------------------------------------------------------------
__global__ void k_mfma_layout(const v4i *a, const v4i *b, v4i *d)
{
    v4i c = {0, 0, 0, 0};
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}

// MFMA issue rate, optionally with `NV` independent VALU instructions per MFMA
template <int NV>
__global__ void k_mfma_rate(uint32_t *out, unsigned long long *cyc, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    uint32_t x[8];
    for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
    __syncthreads();
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
    }
    const unsigned long long t1 = memtime();
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c0[0] + c1[1] + c2[2] + c3[3];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ---- LDS gather -----------------------------------------------------------------------------------------
constexpr int ROWS = 129, ROW_STRIDE_DW = 260;   // 1040 bytes per row: bank = (x + 4 r) mod 32
constexpr int TABLE_DW = ROWS * ROW_STRIDE_DW;
// pattern 0: uniform cells; 1: patch |x-128| <= 3, r <= 6; 2: ring of radius 60; 3: all lanes the same cell;
// 4: x = 128 fixed, r random (the bank skew at work)
__global__ void k_lds_gather(uint32_t *out, unsigned long long *cyc, int iters, int pattern, const uint32_t *rnd)
{
    extern __shared__ uint32_t table[];
    for (int i = threadIdx.x; i < TABLE_DW; i += blockDim.x) table[i] = i * 2654435761u;
    uint32_t addr[16];
    for (int k = 0; k < 16; k++) {
        const uint32_t r0 = rnd[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + k];
        uint32_t x = r0 & 255, r = (r0 >> 8) % 129;
        if (pattern == 1) { x = 125 + (r0 & 255) % 7; r = (r0 >> 8) % 7; }
        if (pattern == 2) {
            const float ang = (r0 & 0xffff) * (6.2831853f / 65536.f);
            const int xx = (int)(60.f * cosf(ang)), yy = (int)(60.f * sinf(ang));
            x = 128 + xx; r = yy < 0 ? -yy : yy;
        }
        if (pattern == 3) { x = 77; r = 33; }
        if (pattern == 4) { x = 128; }
        addr[k] = (r * ROW_STRIDE_DW + x) * 4;
    }
    __syncthreads();
    uint32_t acc[16];
    for (int k = 0; k < 16; k++) acc[k] = 0;
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("ds_read_b32 %0, %1" : "=v"(v[k]) : "v"(addr[k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(v[k]));
    }
    const unsigned long long t1 = memtime();
    uint32_t s = 0;
    for (int k = 0; k < 16; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// misaligned ds_read_b32: what comes back?
__global__ void k_lds_misaligned(uint32_t *out)
{
    __shared__ uint32_t t[64];
    t[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x;
    __syncthreads();
    asm volatile("" :: "v"(t[threadIdx.x ^ 1]));   // keep the stores alive: the asm read below is invisible to the compiler
    uint32_t v, addr = (uint32_t)(size_t)t + 16 + (threadIdx.x & 3);
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v;
}

static double median_cycles(std::vector<unsigned long long> &v)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

#include <algorithm>

template <int OP>
static void run_rate(uint32_t *d_out, unsigned long long *d_cyc, int ncu)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int threads = wps == 8 ? 1024 : 256 * wps, iters = 20000;
        const int blocks = wps == 8 ? 2 * ncu : ncu;
        const int nw = blocks * threads / 64;
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 100, 0x01020304u, 0x00000080u);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters, 0x01020304u, 0x00000080u);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(nw), rt(nw);
        CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rt.data(), d_cyc + 65536, nw * 8, hipMemcpyDeviceToHost));
        const double med = median_cycles(cyc);
        const double n_instr = iters * 8.0;
        printf("  [in-kernel clock %.0f MHz, wave lifetime %.3f ms] ", med / median_cycles(rt) * 100.0, median_cycles(rt) * 1e-5);
        printf("  %-22s waves/SIMD %d: %6.2f ticks per wave-instr per wave -> %5.2f ticks, %5.2f ns per instr per SIMD (wall %.3f ms, %.0f MHz if tick=cycle)\n",
               op_names[OP], wps, med / n_instr, med / n_instr / wps, ms * 1e6 / (n_instr * wps), ms, med / (ms * 1e3));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    uint32_t *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, 4u << 20));
    CHECK(hipMalloc(&d_cyc, 1u << 20));

    printf("== 1. VALU issue (8 independent chains per wave) ==\n");
    run_rate<OP_FMA>(d_out, d_cyc, ncu);
    run_rate<OP_MUL>(d_out, d_cyc, ncu);
    run_rate<OP_PK_FMA>(d_out, d_cyc, ncu);
    run_rate<OP_PK_MUL>(d_out, d_cyc, ncu);
    run_rate<OP_PK_ADD>(d_out, d_cyc, ncu);
    run_rate<OP_DOT4>(d_out, d_cyc, ncu);
    run_rate<OP_DOT2>(d_out, d_cyc, ncu);
    run_rate<OP_PERM>(d_out, d_cyc, ncu);
    run_rate<OP_MQSAD_PK>(d_out, d_cyc, ncu);
    run_rate<OP_MSAD>(d_out, d_cyc, ncu);
    run_rate<OP_SAD>(d_out, d_cyc, ncu);
    run_rate<OP_SDWA_SUB>(d_out, d_cyc, ncu);
    run_rate<OP_XAD>(d_out, d_cyc, ncu);
    run_rate<OP_PK_MAX_U16>(d_out, d_cyc, ncu);
    run_rate<OP_PK_ADD_U16>(d_out, d_cyc, ncu);
    run_rate<OP_PK_LSHR_U16>(d_out, d_cyc, ncu);
    run_rate<OP_RNDNE>(d_out, d_cyc, ncu);
    run_rate<OP_CVT_I32>(d_out, d_cyc, ncu);
    run_rate<OP_BFI>(d_out, d_cyc, ncu);
    run_rate<OP_MAD_U24>(d_out, d_cyc, ncu);
    run_rate<OP_LSHL_ADD>(d_out, d_cyc, ncu);
    run_rate<OP_AND_OR>(d_out, d_cyc, ncu);
    run_rate<OP_CNDMASK>(d_out, d_cyc, ncu);
    run_rate<OP_ALIGNBIT>(d_out, d_cyc, ncu);
    run_rate<OP_MED3>(d_out, d_cyc, ncu);
    run_rate<OP_ADD_U32>(d_out, d_cyc, ncu);
    run_rate<OP_DPP_ROWSHR>(d_out, d_cyc, ncu);
    run_rate<OP_PERMLANE32>(d_out, d_cyc, ncu);
    run_rate<OP_BPERMUTE>(d_out, d_cyc, ncu);

    printf("== 2. dependent mul->sub chain, one wave per SIMD ==\n");
    {
        const int iters = 2000;
        std::vector<unsigned long long> cyc(ncu * 4);
        hipLaunchKernelGGL(k_chain<1>, dim3(ncu), dim3(256), 0, 0, (float *)d_out, d_cyc, iters, -0.9492274f, 0.37f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        printf("  1 chain : %.2f cycles per step (mul+sub)\n", median_cycles(cyc) / (iters * 8.0));
        hipLaunchKernelGGL(k_chain<2>, dim3(ncu), dim3(256), 0, 0, (float *)d_out, d_cyc, iters, -0.9492274f, 0.37f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        printf("  2 chains: %.2f cycles per step pair\n", median_cycles(cyc) / (iters * 8.0));
        hipLaunchKernelGGL(k_chain<4>, dim3(ncu), dim3(256), 0, 0, (float *)d_out, d_cyc, iters, -0.9492274f, 0.37f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        printf("  4 chains: %.2f cycles per 4 steps\n", median_cycles(cyc) / (iters * 8.0));
        std::vector<unsigned long long> cyc4(ncu * 16);
        hipLaunchKernelGGL(k_chain<1>, dim3(ncu), dim3(1024), 0, 0, (float *)d_out, d_cyc, iters, -0.9492274f, 0.37f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cyc4.data(), d_cyc, cyc4.size() * 8, hipMemcpyDeviceToHost));
        printf("  1 chain, 4 waves per SIMD: %.2f cycles per step per wave\n", median_cycles(cyc4) / (iters * 8.0));
    }

    printf("== 3. v_mfma_i32_16x16x64_i8 ==\n");
    {
        std::vector<int8_t> A(64 * 16), B(64 * 16);   // [lane][byte]
        srand(7);
        for (auto &v : A) v = (int8_t)(rand() % 255 - 127);
        for (auto &v : B) v = (int8_t)(rand() % 255 - 127);
        v4i *da, *db, *dd;
        CHECK(hipMalloc(&da, 1024)); CHECK(hipMalloc(&db, 1024)); CHECK(hipMalloc(&dd, 1024));
        CHECK(hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
        CHECK(hipDeviceSynchronize());
        int D[64][4];
        CHECK(hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost));
        // hypothesis: A row i = lane&15, B col n = lane&15, both pair (lane>>4, byte j) <-> the same k;
        // D lane l reg r = D[row 4*(l>>4)+r][col l&15]
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                int s = 0;
                for (int g = 0; g < 4; g++)
                    for (int j = 0; j < 16; j++) s += (int)A[(16 * g + row) * 16 + j] * (int)B[(16 * g + col) * 16 + j];
                if (s != D[l][r]) bad++;
            }
        printf("  layout hypothesis (A row=lane&15, B col=lane&15, k paired by (lane>>4, byte); D row=4*(lane>>4)+reg, col=lane&15): %s (%d mismatches)\n",
               bad ? "WRONG" : "confirmed", bad);
        for (int nv = 0; nv <= 2; nv++)
            for (int wps = 1; wps <= 4; wps *= 2) {
                const int iters = 2000, threads = 256 * wps, nw = ncu * threads / 64;
                if (nv == 0) hipLaunchKernelGGL(k_mfma_rate<0>, dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, iters);
                if (nv == 1) hipLaunchKernelGGL(k_mfma_rate<2>, dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, iters);
                if (nv == 2) hipLaunchKernelGGL(k_mfma_rate<4>, dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, iters);
                CHECK(hipDeviceSynchronize());
                std::vector<unsigned long long> cyc(nw);
                CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
                printf("  mfma + %d valu each, waves/SIMD %d: %.2f cycles per mfma per SIMD\n", nv * 2, wps,
                       median_cycles(cyc) / (iters * 4.0) / wps);
            }
    }

    printf("== 4. ds_read_b32 gathers from a 134 KB LDS table (one workgroup per CU) ==\n");
    {
        const size_t lds_bytes = (size_t)TABLE_DW * 4;
        CHECK(hipFuncSetAttribute((const void *)k_lds_gather, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        std::vector<uint32_t> rnd((size_t)ncu * 1024 * 16);
        srand(11);
        for (auto &v : rnd) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        uint32_t *d_rnd;
        CHECK(hipMalloc(&d_rnd, rnd.size() * 4));
        CHECK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
        const char *pn[] = {"uniform", "patch 7x7", "ring r=60", "one cell", "x=128, r random"};
        for (int pattern = 0; pattern < 5; pattern++)
            for (int waves = 4; waves <= 16; waves *= 2) {
                const int iters = 500, threads = 64 * waves, nw = ncu * waves;
                hipLaunchKernelGGL(k_lds_gather, dim3(ncu), dim3(threads), lds_bytes, 0, d_out, d_cyc, iters, pattern, d_rnd);
                CHECK(hipDeviceSynchronize());
                std::vector<unsigned long long> cyc(nw);
                CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
                printf("  %-16s %2d waves/CU: %.2f cycles per gather instr per CU\n", pn[pattern], waves,
                       median_cycles(cyc) / (iters * 16.0) / waves);
            }
        hipLaunchKernelGGL(k_lds_misaligned, dim3(1), dim3(64), 0, 0, d_out);
        CHECK(hipDeviceSynchronize());
        uint32_t mis[8];
        CHECK(hipMemcpy(mis, d_out, 32, hipMemcpyDeviceToHost));
        printf("  ds_read_b32 at byte 16+{0,1,2,3} (aligned value %08x): %08x %08x %08x %08x\n", 0x03020100u + 0x04040404u * 4,
               mis[0], mis[1], mis[2], mis[3]);
    }
    return 0;
}
