// Device-only building blocks shared by the chain kernels that run their first FIR stage on the matrix cores
// (v_mfma_i32_16x16x64_i8): the byte-level front end and the squelch magnitude.  gfx950 only (no host twin).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "iqd_prims.h"

namespace iqd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

typedef float v2f __attribute__((ext_vector_type(2)));
// x * (s, s) as ONE v_pk_mul_f32 with the constant pair in scalar registers (left to itself the compiler multiplies the
// halves one by one: a literal does not fit a packed instruction)
__device__ __forceinline__ v2f pk_mul_s(v2f x, uint64_t s_pair)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(s_pair));
    return r;
}

// ---- small device helpers -----------------------------------------------------------------------
// byte B of x replaced by its two's-complement negation, the other bytes kept (v_sub_u32_sdwa): int8
// wrap, so -(-128) stays -128 like the reference's rotation (IqDataProcessor.cc:594-607)
#define ST_NEG_BYTE(x, B)                                                                                     \
    asm("v_sub_u32_sdwa %0, %1, %0 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_" #B \
        : "+v"(x) : "v"(zero))

// raw offset-binary bytes of 8 samples -> signed bytes with the rotation's SIGNS applied in place (which
// byte feeds which rail is folded into the tap matrices).  +Fs/4: I' = {I0,-Q1,-I2,Q3}, Q' = {Q0,I1,-Q2,-I3};
// -Fs/4: I' = {I0,Q1,-I2,-Q3}, Q' = {Q0,-I1,-Q2,I3} (IqDataProcessor.cc:567-611).
template <int ROT>
__device__ __forceinline__ uint4 st_front(uint4 raw, uint32_t zero)
{
    uint32_t d[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        uint32_t d0 = d[h], d1 = d[h + 1];
        if (ROT == 0) {
            d0 ^= 0x80808080u;
            d1 ^= 0x80808080u;
        } else if (ROT > 0) {
            d0 = (d0 ^ 0x7f808080u) + 0x01000000u;   // byte 3 (Q1): ~s + 1, the carry leaves the register
            d1 ^= 0x80808080u;
            ST_NEG_BYTE(d1, 0);
            ST_NEG_BYTE(d1, 1);
            ST_NEG_BYTE(d1, 2);
        } else {
            d0 ^= 0x80808080u;
            ST_NEG_BYTE(d0, 2);
            d1 = (d1 ^ 0x7f808080u) + 0x01000000u;   // byte 3 (Q3)
            ST_NEG_BYTE(d1, 0);
            ST_NEG_BYTE(d1, 1);
        }
        d[h] = d0;
        d[h + 1] = d1;
    }
    return uint4{d[0], d[1], d[2], d[3]};
}

// sum over the 2 samples of a dword of signed bytes of max(|I|,|Q|) + min(|I|,|Q|)/2, added into two 16-bit
// lanes of acc (SignalDetector.cc:227-247).  |x| of all four bytes at once: (x ^ m) + t with t the sign bits.
__device__ __forceinline__ uint32_t st_mag_dword(uint32_t sx, uint32_t acc)
{
    const uint32_t t = (sx >> 7) & 0x01010101u;
    uint32_t t8 = t << 8;
    asm("" : "+v"(t8));                                    // (else hipcc folds this into a quarter-rate v_mul_lo_u32 by 255)
    const uint32_t m = t8 - t;
    const uint32_t ab = (sx ^ m) + t;                      // bytes |I0| |Q0| |I1| |Q1|, each <= 128
    const us2 a = __builtin_bit_cast(us2, ab & 0x00ff00ffu);
    const us2 b = __builtin_bit_cast(us2, (ab >> 8) & 0x00ff00ffu);
    const us2 mx = __builtin_elementwise_max(a, b), mn = __builtin_elementwise_min(a, b);
    return acc + __builtin_bit_cast(uint32_t, mx) + __builtin_bit_cast(uint32_t, (us2)(mn >> 1));
}


// The same from the RAW bytes (offset binary I0 Q0 I1 Q1): |u - 128| of one byte is one masked v_msad_u8 (bytes whose
// reference byte is 0 do not count), two of them join into a 16-bit pair with one v_lshl_or_b32: ten operations per
// dword against twelve for the byte-parallel arithmetic above.
__device__ __forceinline__ uint32_t st_mag_raw_dword(uint32_t raw, uint32_t acc)
{
    uint32_t a0, b0, a1, b1;
    asm("v_msad_u8 %0, %1, %2, 0" : "=v"(a0) : "v"(raw), "s"(0x00000080u));
    asm("v_msad_u8 %0, %1, %2, 0" : "=v"(b0) : "v"(raw), "s"(0x00008000u));
    asm("v_msad_u8 %0, %1, %2, 0" : "=v"(a1) : "v"(raw), "s"(0x00800000u));
    asm("v_msad_u8 %0, %1, %2, 0" : "=v"(b1) : "v"(raw), "s"(0x80000000u));
    const us2 a = __builtin_bit_cast(us2, a0 | (a1 << 16)), b = __builtin_bit_cast(us2, b0 | (b1 << 16));
    const us2 mx = __builtin_elementwise_max(a, b), mn = __builtin_elementwise_min(a, b);
    return acc + __builtin_bit_cast(uint32_t, mx) + __builtin_bit_cast(uint32_t, (us2)(mn >> 1));
}

// The same with ONE quad-SAD: the dword's bytes are put in the order I0 I1 Q0 Q1 (v_perm_b32) and v_mqsad_pk_u16_u8 against
// the reference 0x00000080 - only its byte 0 counts, at the four byte positions in turn - leaves |I0 - 128|, |I1 - 128| as
// the halves of one register and |Q0 - 128|, |Q1 - 128| of the next: six instructions per dword against ten (the quad-SAD
// issues at a quarter of the rate, tools/ubench: the kernel's time is the same either way, 0.3288 against 0.3324 ms on one
// box).  IQD_ST_MQSAD=0 selects the single SADs below (A/B builds).
__device__ __forceinline__ uint32_t st_mag_raw_dword_q(uint32_t raw, uint32_t acc)
{
    const uint32_t p = __builtin_amdgcn_perm(raw, raw, 0x03010200u);
    uint32_t upper = 0;
    upper = __builtin_nondeterministic_value(upper);             // (bytes 4-6 of the source are compared with masked reference bytes)
    const uint64_t r = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)p | ((uint64_t)upper << 32), 0x00000080u, 0ull);
    const us2 a = __builtin_bit_cast(us2, (uint32_t)r), b = __builtin_bit_cast(us2, (uint32_t)(r >> 32));
    const us2 mx = __builtin_elementwise_max(a, b), mn = __builtin_elementwise_min(a, b);
    return acc + __builtin_bit_cast(uint32_t, mx) + __builtin_bit_cast(uint32_t, (us2)(mn >> 1));
}
#ifndef IQD_ST_MQSAD
#define IQD_ST_MQSAD 1
#endif

// 8 raw samples (one lane's 16 bytes of a piece) added into the two 16-bit partial sums of acc
__device__ __forceinline__ uint32_t st_mag_raw_chunk(const uint4 &raw, uint32_t acc)
{
    if (IQD_ST_MQSAD) {
        acc = st_mag_raw_dword_q(raw.x, acc);
        acc = st_mag_raw_dword_q(raw.y, acc);
        acc = st_mag_raw_dword_q(raw.z, acc);
        return st_mag_raw_dword_q(raw.w, acc);
    }
    // eight |u - 128| at a time, then their packed arithmetic: interleaved dword by dword the compiler pads the SAD ->
    // packed-16 dependences with s_nop; all sixteen first costs eight more live registers (the launch that holds all four
    // pipelines then spills vector registers to scratch)
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        uint32_t a0[2], b0[2], a1[2], b1[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            asm volatile("v_msad_u8 %0, %1, %2, 0" : "=v"(a0[d]) : "v"(w[h + d]), "s"(0x00000080u));
            asm volatile("v_msad_u8 %0, %1, %2, 0" : "=v"(b0[d]) : "v"(w[h + d]), "s"(0x00008000u));
            asm volatile("v_msad_u8 %0, %1, %2, 0" : "=v"(a1[d]) : "v"(w[h + d]), "s"(0x00800000u));
            asm volatile("v_msad_u8 %0, %1, %2, 0" : "=v"(b1[d]) : "v"(w[h + d]), "s"(0x80000000u));
        }
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const us2 a = __builtin_bit_cast(us2, a0[d] | (a1[d] << 16)), b = __builtin_bit_cast(us2, b0[d] | (b1[d] << 16));
            const us2 mx = __builtin_elementwise_max(a, b), mn = __builtin_elementwise_min(a, b);
            acc = acc + __builtin_bit_cast(uint32_t, mx) + __builtin_bit_cast(uint32_t, (us2)(mn >> 1));
        }
    }
    return acc;
}

__device__ __forceinline__ uint32_t st_mag_chunk(const uint4 &s)   // 8 samples of signed bytes
{
    uint32_t m16 = st_mag_dword(s.x, 0u);
    m16 = st_mag_dword(s.y, m16);
    m16 = st_mag_dword(s.z, m16);
    m16 = st_mag_dword(s.w, m16);
    return (m16 & 0xffffu) + (m16 >> 16);
}

// ---- the same through a table in LDS (kernels with 68 KiB of LDS to spare) ----
// One byte per raw sample (I, Q as they come from the tuner, offset binary): max(|I-128|, |Q-128|) + min(..)/2 <= 192.
// Rows are 272 bytes apart, not 256: the bank is then 4 Q + I/4 and weak signals - every lane's I and Q within a few
// counts of 128 - spread over the banks instead of piling onto two of them.  Two operations form a sample's address
// straight from the raw dword (SDWA picks the bytes), one ds_read_u8 fetches the magnitude: 3.5 operations per sample
// with the sums, against 6.5 for the arithmetic above.
constexpr int ST_MAGLUT_PITCH = 272;
constexpr int ST_MAGLUT_BYTES = 256 * ST_MAGLUT_PITCH;
__device__ __forceinline__ void st_maglut_build(uint8_t *lut, int tid, int n_threads)
{
    for (int e = tid; e < 256 * (ST_MAGLUT_PITCH / 4); e += n_threads) {
        const int q = e / (ST_MAGLUT_PITCH / 4), i0 = 4 * (e % (ST_MAGLUT_PITCH / 4));
        const int b = q < 128 ? 128 - q : q - 128;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = (i0 + k) & 255;                        // (columns 256..271 are padding)
            const int a = i < 128 ? 128 - i : i - 128;
            w |= (uint32_t)(a > b ? a + (b >> 1) : b + (a >> 1)) << (8 * k);
        }
        *(uint32_t *)(lut + 4 * e) = w;
    }
}
// magnitudes of the two samples of a raw dword (bytes I0 Q0 I1 Q1)
__device__ __forceinline__ void st_maglut_pair(const uint8_t *lut, uint32_t w, uint32_t four, uint32_t &m0, uint32_t &m1)
{
    uint32_t t0, a0, t1, a1;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(t0) : "v"(four), "v"(w));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a0) : "v"(t0), "v"(w));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t1) : "v"(four), "v"(w));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "v"(t1), "v"(w));
    m0 = lut[a0];                                                // 256 Q + I + 16 Q
    m1 = lut[a1];
}
// The same with the table at a known LDS ADDRESS (the kernel's dynamic LDS starts at address 0, checked at kernel start):
// through a generic pointer the compiler adds the LDS base - a relocated zero - to every one of the eight addresses.
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
__device__ __forceinline__ uint32_t st_maglut_chunk_at(uint32_t lut_lds_address, const uint4 &raw, uint32_t four)   // 8 raw samples
{
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t m[8];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t t0, a0, t1, a1;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(t0) : "v"(four), "v"(w[d]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a0) : "v"(t0), "v"(w[d]));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t1) : "v"(four), "v"(w[d]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "v"(t1), "v"(w[d]));
        m[2 * d] = *(lds_cu8 *)(uintptr_t)(lut_lds_address + a0);
        m[2 * d + 1] = *(lds_cu8 *)(uintptr_t)(lut_lds_address + a1);
    }
    return (m[0] + m[1] + m[2]) + (m[3] + m[4] + m[5]) + (m[6] + m[7]);
}
__device__ __forceinline__ uint32_t st_maglut_chunk(const uint8_t *lut, const uint4 &raw, uint32_t four)   // 8 raw samples
{
    uint32_t m[8];
    st_maglut_pair(lut, raw.x, four, m[0], m[1]);
    st_maglut_pair(lut, raw.y, four, m[2], m[3]);
    st_maglut_pair(lut, raw.z, four, m[4], m[5]);
    st_maglut_pair(lut, raw.w, four, m[6], m[7]);
    return (m[0] + m[1] + m[2]) + (m[3] + m[4] + m[5]) + (m[6] + m[7]);
}

// ---- input prefetch of the streaming kernels ----
// A 16-byte global load the compiler does not track, and the matching wait.  With ordinary loads the compiler's
// own s_waitcnt placement joins the loop's entry and back edge conservatively (vmcnt(0) at the header, i.e. a wait
// for the load issued one piece ago) and hands buffers on with register moves that wait for the youngest load;
// a P wave then pays a trip to HBM per piece however many pieces it asked for in advance.  Loads return in order,
// so "at most N younger loads outstanding" is exactly "this one has arrived"; other memory operations issued in
// between (magnitude atomics) only make the wait stricter.  Nothing that is not inlined may be called while such loads
// are in flight: a callee uses registers as it pleases (tried: a noinline helper in the piece loop corrupted everything).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u gload16_untracked(const void *p)
{
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p));
    return r;
}
// (In place, on the variable the load was issued into: handing the value to the wait by copy invites the compiler to
// make that copy - of registers still in flight - in front of the wait.  tools/isa_lint.py checks the generated code
// for exactly this.)
template <int N_YOUNGER>
__device__ __forceinline__ void gload_wait(v4u &r)
{
    asm volatile("s_waitcnt vmcnt(%1) ; arrived %0" : "+v"(r) : "n"(N_YOUNGER));   // (the comment is for tools/isa_lint.py)
}
// the same with a count that is only known after unrolling (folds to one s_waitcnt)
__device__ __forceinline__ void gload_wait_n(v4u &r, int n_younger)
{
    switch (n_younger) {
    case 1: gload_wait<1>(r); break;
    case 2: gload_wait<2>(r); break;
    case 3: gload_wait<3>(r); break;
    case 4: gload_wait<4>(r); break;
    case 5: gload_wait<5>(r); break;
    case 6: gload_wait<6>(r); break;
    case 7: gload_wait<7>(r); break;
    case 8: gload_wait<8>(r); break;
    case 9: gload_wait<9>(r); break;
    case 10: gload_wait<10>(r); break;
    case 11: gload_wait<11>(r); break;
    default: gload_wait<0>(r); break;
    }
}
__device__ __forceinline__ uint4 as_uint4(v4u r) { return uint4{r.x, r.y, r.z, r.w}; }
__device__ __forceinline__ uint32_t gload4_untracked(const void *p)
{
    uint32_t r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p));
    return r;
}
template <int N_YOUNGER>
__device__ __forceinline__ void gload_wait(uint32_t &r)
{
    asm volatile("s_waitcnt vmcnt(%1) ; arrived %0" : "+v"(r) : "n"(N_YOUNGER));
}

#ifndef IQD_RINGS_IN_A_ROW
#define IQD_RINGS_IN_A_ROW 0
#endif

// ---- producer / consumer plumbing of the streaming kernels (waves of one workgroup talking through LDS rings) ----
// (through the LDS address space: on the generic pointer this is a flat_load, which travels the vector-memory path
// as well and is waited for with vmcnt(0) - i.e. together with every input load the wave has in flight)
__device__ __forceinline__ uint32_t lds_load_relaxed(const uint32_t *p)
{
    const __attribute__((address_space(3))) uint32_t *q = (const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(uint32_t)(uintptr_t)p;
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// lane 0 adds 1 to an LDS word (exec is all ones wherever this is used); the plain HIP form costs a dozen
// instructions of "which lane is first" bookkeeping per call
__device__ __forceinline__ void lds_signal(const uint32_t *p)
{
    const uint32_t addr = (uint32_t)(uintptr_t)p, one = 1u;
    asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(one) : "memory");
}

// byte offset of granule q (4 samples) of ring row j inside a slot: XOR swizzle, conflict-free for the
// P waves' ds_write_b128 and the IIR lanes' ds_read_b128
__device__ __forceinline__ uint32_t st_slot_off(uint32_t j, uint32_t q) { return j * 64u + ((q ^ ((j >> 2) & 3u)) << 4); }


}  // namespace iqd
