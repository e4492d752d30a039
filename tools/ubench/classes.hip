// Diagnostic tool, not product: vector-instruction issue cost per opcode on gfx950 with long unrolled bodies (tools/ubench/ubench.hip
// loops over 8 instructions, so its one- and two-wave figures carry the loop's branch), and pairs of opcodes interleaved -
// does a mix cost the sum of its parts?  One workgroup per CU; 16, 12 or 8 waves = 4, 3 or 2 per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o classes classes.hip      Run: ./classes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// 16 independent registers, the instruction applied to each in turn, 4 times over: 64 instructions per loop round.
// T1 / T2: asm templates over %0 (read-write), %1, %2 (inputs); T2 empty = a body of T1 only, else T1 and T2 alternate.
#define KERNEL(NAME, T1, T2)                                                                                         \
    __global__ void NAME(uint32_t *out, unsigned long long *cyc, int iters, uint32_t b, uint32_t c)                  \
    {                                                                                                                \
        uint32_t a[16];                                                                                              \
        for (int k = 0; k < 16; k++) a[k] = threadIdx.x * 977u + k * 131u + b;                                       \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = memtime();                                                                     \
        for (int i = 0; i < iters; i++) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                          \
                _Pragma("unroll") for (int k = 0; k < 16; k++) {                                                     \
                    if ((T2)[0] == 0 || (k & 1) == 0) asm volatile(T1 : "+v"(a[k]) : "v"(b), "v"(c));                \
                    else asm volatile(T2 : "+v"(a[k]) : "v"(b), "v"(c));                                             \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        const unsigned long long t1 = memtime();                                                                     \
        uint32_t s = 0;                                                                                              \
        for (int k = 0; k < 16; k++) s += a[k];                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                              \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                    \
    }

#define OPS(X)                                                                                      \
    X(fma_f32, "v_fma_f32 %0, %0, %1, %2", "")                                                      \
    X(add_f32, "v_add_f32 %0, %0, %1", "")                                                          \
    X(sub_f32, "v_sub_f32 %0, %0, %1", "")                                                          \
    X(mul_f32, "v_mul_f32 %0, %0, %1", "")                                                          \
    X(max_f32, "v_max_f32 %0, %0, %1", "")                                                          \
    X(add_u32, "v_add_u32 %0, %0, %1", "")                                                          \
    X(sub_u32, "v_sub_u32 %0, %0, %1", "")                                                          \
    X(and_b32, "v_and_b32 %0, %0, %1", "")                                                          \
    X(or_b32, "v_or_b32 %0, %0, %1", "")                                                            \
    X(xor_b32, "v_xor_b32 %0, %0, %1", "")                                                          \
    X(lshlrev_b32, "v_lshlrev_b32 %0, 8, %0", "")                                                   \
    X(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0", "")                                                   \
    X(ashrrev_i32, "v_ashrrev_i32 %0, 3, %0", "")                                                   \
    X(mov_b32, "v_mov_b32 %0, %1", "")                                                              \
    X(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc", "")                                             \
    X(cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc", "")                                     \
    X(cndmask_sgpr_pair, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "")                              \
    X(cndmask_alternating, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v_cndmask_b32_e64 %0, %1, %0, s[22:23]") \
    X(mix_cndmask_add_f32, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v_add_f32 %0, %0, %1")          \
    X(max_u32, "v_max_u32 %0, %0, %1", "")                                                          \
    X(min_i32, "v_min_i32 %0, %0, %1", "")                                                          \
    X(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", "")                                                  \
    X(mul_i32_i24, "v_mul_i32_i24 %0, %0, %1", "")                                                  \
    X(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", "")                                                    \
    X(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", "")                                              \
    X(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2", "")                                              \
    X(lshl_add_u32, "v_lshl_add_u32 %0, %0, 8, %1", "")                                             \
    X(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 2", "")                                             \
    X(lshl_or_b32, "v_lshl_or_b32 %0, %0, 8, %1", "")                                               \
    X(add3_u32, "v_add3_u32 %0, %0, %1, %2", "")                                                    \
    X(and_or_b32, "v_and_or_b32 %0, %0, %1, %2", "")                                                \
    X(or3_b32, "v_or3_b32 %0, %0, %1, %2", "")                                                      \
    X(bfe_u32, "v_bfe_u32 %0, %0, 16, 8", "")                                                       \
    X(bfi_b32, "v_bfi_b32 %0, %1, %0, %2", "")                                                      \
    X(perm_b32, "v_perm_b32 %0, %0, %1, %2", "")                                                    \
    X(alignbit_b32, "v_alignbit_b32 %0, %0, %1, 14", "")                                            \
    X(msad_u8, "v_msad_u8 %0, %0, %1, %2", "")                                                      \
    X(sad_u8, "v_sad_u8 %0, %0, %1, %2", "")                                                        \
    X(med3_i32, "v_med3_i32 %0, %0, %1, %2", "")                                                    \
    X(dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0", "")                                            \
    X(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0", "")                                              \
    X(cvt_i32_f32, "v_cvt_i32_f32 %0, %0", "")                                                      \
    X(cvt_f32_i32, "v_cvt_f32_i32 %0, %0", "")                                                      \
    X(rndne_f32, "v_rndne_f32 %0, %0", "")                                                          \
    X(pk_add_u16, "v_pk_add_u16 %0, %0, %1", "")                                                    \
    X(pk_max_u16, "v_pk_max_u16 %0, %0, %1", "")                                                    \
    X(pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 1, %0", "")                                             \
    X(lshlrev_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", "")   \
    X(sub_sdwa_byte, "v_sub_u32_sdwa %0, %1, %0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2", "") \
    X(add_sdwa_word, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "")      \
    X(mov_dpp_rowshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "")              \
    X(add_f32_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "")             \
    X(fmac_f32, "v_fmac_f32 %0, %1, %2", "")                                                        \
    X(mix_fma_perm, "v_fma_f32 %0, %0, %1, %2", "v_perm_b32 %0, %0, %1, %2")                        \
    X(mix_add_u32_perm, "v_add_u32 %0, %0, %1", "v_perm_b32 %0, %0, %1, %2")                        \
    X(mix_fma_lshl_add, "v_fma_f32 %0, %0, %1, %2", "v_lshl_add_u32 %0, %0, 8, %1")                 \
    X(mix_mul_f32_dot2, "v_mul_f32 %0, %0, %1", "v_dot2_i32_i16 %0, %1, %2, %0")                    \
    X(mix_fma_add_u32, "v_fma_f32 %0, %0, %1, %2", "v_add_u32 %0, %0, %1")                          \
    X(mix_perm_mad, "v_perm_b32 %0, %0, %1, %2", "v_mad_u32_u24 %0, %0, %1, %2")

#define X(N, A, B) KERNEL(k_##N, A, B)
OPS(X)
#undef X

// the same with T2 in ONE of 16 places (k == 0), T1 in the other 15: the marginal cost of an instruction among plain ones
#define KERNEL1(NAME, T1, T2)                                                                                        \
    __global__ void NAME(uint32_t *out, unsigned long long *cyc, int iters, uint32_t b, uint32_t c)                  \
    {                                                                                                                \
        uint32_t a[16];                                                                                              \
        for (int k = 0; k < 16; k++) a[k] = threadIdx.x * 977u + k * 131u + b;                                       \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = memtime();                                                                     \
        for (int i = 0; i < iters; i++) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                          \
                _Pragma("unroll") for (int k = 0; k < 16; k++) {                                                     \
                    if (k != 0) asm volatile(T1 : "+v"(a[k]) : "v"(b), "v"(c));                                      \
                    else asm volatile(T2 : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");                                     \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        const unsigned long long t1 = memtime();                                                                     \
        uint32_t s = 0;                                                                                              \
        for (int k = 0; k < 16; k++) s += a[k];                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                              \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                    \
    }
#define OPS1(X)                                                                                     \
    X(one_add_f32_in_16_add_f32, "v_add_f32 %0, %0, %1", "v_add_f32 %0, %0, %1")                    \
    X(one_cndmask_vcc_in_16_add_f32, "v_add_f32 %0, %0, %1", "v_cndmask_b32 %0, %0, %1, vcc")       \
    X(one_cndmask_e64_vcc_in_16_add_f32, "v_add_f32 %0, %0, %1", "v_cndmask_b32_e64 %0, %0, %1, vcc") \
    X(one_cndmask_pair_in_16_add_f32, "v_add_f32 %0, %0, %1", "v_cndmask_b32_e64 %0, %0, %1, s[20:21]") \
    X(one_cmp_cndmask_vcc_in_16, "v_add_f32 %0, %0, %1", "v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc") \
    X(one_cmp_cndmask_pair_in_16, "v_add_f32 %0, %0, %1", "v_cmp_lt_u32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]") \
    X(one_addc_vcc_in_16_add_f32, "v_add_f32 %0, %0, %1", "v_addc_co_u32 %0, vcc, %0, %1, vcc")     \
    X(one_cndmask_vcc_in_16_dot2, "v_dot2_i32_i16 %0, %1, %2, %0", "v_cndmask_b32 %0, %0, %1, vcc")
#define X(N, A, B) KERNEL1(k1_##N, A, B)
OPS1(X)
#undef X

typedef void (*Kern)(uint32_t *, unsigned long long *, int, uint32_t, uint32_t);
struct Entry { const char *name; Kern k; };
#define X(N, A, B) {#N, k_##N},
#define X1(N, A, B) {#N, k1_##N},
static const Entry entries[] = {OPS(X) OPS1(X1)};
#undef X1
#undef X

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, 4u << 20));
    CHECK(hipMalloc(&d_cyc, 1u << 20));
    printf("cycles per wave-instruction per SIMD (64 instructions per loop round, one workgroup per CU)\n");
    printf("%-20s %8s %8s %8s %8s\n", "opcode", "4 w/SIMD", "3 w/SIMD", "2 w/SIMD", "1 w/SIMD");
    const int iters = 1500;
    for (const Entry &e : entries) {
        printf("%-20s", e.name);
        for (int wps = 4; wps >= 1; wps--) {
            const int threads = 256 * wps, nw = ncu * threads / 64;
            hipLaunchKernelGGL(e.k, dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, 50, 0x01020304u, 0x00000080u);
            hipLaunchKernelGGL(e.k, dim3(ncu), dim3(threads), 0, 0, d_out, d_cyc, iters, 0x01020304u, 0x00000080u);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> cyc(nw);
            CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
            std::sort(cyc.begin(), cyc.end());
            printf(" %8.2f", (double)cyc[nw / 2] / (iters * 64.0) / wps);
        }
        printf("\n");
    }
    return 0;
}
