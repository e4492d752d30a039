// Scalar / packed primitives of the chain kernels.
//
// Device build (hipcc, gfx950): thin wrappers over CDNA4 instructions
// (v_dot4_i32_i8, v_dot2_i32_i16, v_perm_b32, v_alignbyte_b32, v_med3_i32, v_bfe_u32).
// Host build (IQD_HOST_EMU, used only by tests/emu to single-step the kernels' phase
// functions on the CPU): bit-identical C++ restatements of those instructions.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) && !defined(IQD_HOST_EMU)
#include <hip/hip_runtime.h>
#define IQD_DEV __host__ __device__ __forceinline__
#else
#define IQD_DEV inline
#endif
// 1 only while hipcc compiles the gfx950 side of a translation unit
#if defined(__HIP_DEVICE_COMPILE__) && !defined(IQD_HOST_EMU)
#define IQD_ON_DEVICE 1
#else
#define IQD_ON_DEVICE 0
#endif

namespace iqd {

struct alignas(16) u32x4 { uint32_t x, y, z, w; };
struct alignas(8) u32x2 { uint32_t x, y; };

IQD_DEV uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
IQD_DEV float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// acc + sum of the four signed-byte products (v_dot4_i32_i8)
IQD_DEV int dot4(uint32_t a, int32_t b, int acc)
{
#if IQD_ON_DEVICE
    return __builtin_amdgcn_sdot4((int)a, b, acc, false);
#else
    for (int i = 0; i < 4; i++)
        acc += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)((uint32_t)b >> (8 * i));
    return acc;
#endif
}

// First link of a dot4 chain: c + dot4(a, b) with c in a VGPR / as the constant 0, written to a
// fresh register (VOP3P form).  The compiler would otherwise spend a v_mov per chain on the
// destructive v_dot4c form.
IQD_DEV int dot4_first(uint32_t a, int32_t b, int c_vgpr)
{
#if IQD_ON_DEVICE
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c_vgpr));
    return r;
#else
    return dot4(a, b, c_vgpr);
#endif
}
IQD_DEV int dot4_first0(uint32_t a, int32_t b)
{
#if IQD_ON_DEVICE
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b));
    return r;
#else
    return dot4(a, b, 0);
#endif
}

// acc + a.lo*b.lo + a.hi*b.hi on signed 16-bit halves (v_dot2_i32_i16)
IQD_DEV int dot2(uint32_t a, uint32_t b, int acc)
{
#if IQD_ON_DEVICE
    typedef short short2v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, false);
#else
    uint32_t r = (uint32_t)acc;
    r += (uint32_t)((int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff));
    r += (uint32_t)((int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16));
    return (int)r;
#endif
}

// byte permute (v_perm_b32): selector byte i picks byte 0-3 from lo, 4-7 from hi
IQD_DEV uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if IQD_ON_DEVICE
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t byte = (s < 8) ? (uint32_t)((v >> (8 * s)) & 0xff) : (s == 12 ? 0u : 0xffu);
        r |= byte << (8 * i);
    }
    return r;
#endif
}

// bytes [shift .. shift+3] of the 8-byte value hi:lo (v_alignbyte_b32)
IQD_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t shift)
{
#if IQD_ON_DEVICE
    return __builtin_amdgcn_alignbyte(hi, lo, shift);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8 * (shift & 3)));
#endif
}

IQD_DEV int clamp_q30(int acc)  // folds to v_med3_i32
{
    const int lo = acc < -0x40000000 ? -0x40000000 : acc;
    return lo > 0x3fffffff ? 0x3fffffff : lo;
}

IQD_DEV uint32_t bfe(uint32_t v, uint32_t off, uint32_t width)
{
#if IQD_ON_DEVICE
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return (v >> off) & ((1u << width) - 1u);
#endif
}

// Two's-complement negation of the bytes selected by mask (0xff per selected byte), each
// byte wrapping on its own: -(-128) stays -128, as the reference's int8 rotation does
// (IqDataProcessor.cc:594-607).
IQD_DEV uint32_t neg_bytes(uint32_t x, uint32_t mask)
{
    uint32_t t = x ^ mask;
    return ((t & 0x7f7f7f7fu) + (mask & 0x01010101u)) ^ (t & 0x80808080u);
}

// Where the atan2 table keeps the angle of (x, y), given off = (y << 8) | x.  The table is stored in
// tiles of 4 rows x 8 columns (one 128-byte cache line each: index bits y7..y2 x7..x3 y1 y0 x2 x1 x0), so that the
// ring of cells a constant-envelope signal visits covers 3-4 times fewer lines than with 1 x 32 row pieces
// (measured: bench signal -1.6 %, full-scale signal -8 %, weak signal and uniform noise +2-3 %: four more
// operations per sample against fewer L1 misses).
IQD_DEV uint32_t lut_index(uint32_t off)
{
#ifndef IQD_LUT_ROWS   // (row-major table: the A/B alternative)
    uint32_t t = (((off << 2) ^ off) & 0x03e0u) ^ off;      // bits 9..5 <- x7..x3       (v_lshlrev, v_bfi)
    t = (((off >> 5) ^ t) & 0x0018u) ^ t;                   // bits 4..3 <- y1 y0         (v_lshrrev, v_bfi)
    return t;
#else
    return off;
#endif
}

// Branch-cut handling of the FM discriminators (WbFmDemodulator.cc:472-480,
// FmDemodulator.cc:485-493).  The reference compares in double against M_PI and adds
// -+2*M_PI in double before rounding to float; for |d| <= 2*pi that equals the two float
// operations below exactly (exhaustively checked over every float in [pi, 2*pi], see
// tests/test_host_logic.py::test_wrap_delta_float_only).
IQD_DEV float wrap_delta(float d)
{
    const float PI_F = 3.14159274101257324f;       // smallest float > M_PI
    const float TWO_PI_HI = 6.28318548202514648f;  // (float)(2*M_PI)
    const float TWO_PI_LO = -1.74845553146951715e-7f;  // (float)(2*M_PI - TWO_PI_HI)
    // k = sign(d) when |d| >= PI_F else 0, for |d| <= 2*pi (differences of two table angles), as one multiply and
    // one round-to-nearest-even: the product is monotone in d, equals 0.5 exactly at the float just below PI_F
    // (ties go to the even 0) and 0.50000006 at PI_F, 1.0000001 at 2*pi (tests/test_host_logic.py checks the whole
    // interval).  d - k*HI is exact (Sterbenz), then one rounding.
    const float k = __builtin_rintf(d * 0.15915495157241821f);
    (void)PI_F;
    d = __builtin_fmaf(-k, TWO_PI_HI, d);
    d = __builtin_fmaf(-k, TWO_PI_LO, d);
    return d;
}

// (int16_t)f as x86-64/gcc executes it: cvttss2si to int32 ("integer indefinite"
// 0x80000000 when out of range or NaN), then the low 16 bits.
IQD_DEV int32_t cast_i16(float f)
{
    int32_t wide;
    if (f >= -2147483648.0f && f < 2147483648.0f) wide = (int32_t)f;
    else wide = (int32_t)0x80000000u;
    return (int32_t)(int16_t)(uint16_t)((uint32_t)wide & 0xffffu);
}

// The same cast when the caller has proved |f| < 2^31 (v_cvt_i32_f32 truncates toward zero).
IQD_DEV uint32_t cast_i16_bounded(float f) { return (uint32_t)(int32_t)f; }

// The first product of a chain whose accumulator starts from a rounding term held in a register: the three-address form
// (v_dot2_i32_i16 d, a, b, c) leaves the term where it is; the compiler's choice, the two-address v_dot2c, needs a move of the
// term into the accumulator first.  b: a scalar register (the three-address encoding takes no literal).
IQD_DEV int dot2_from(uint32_t a, uint32_t b_scalar, int start)
{
#if IQD_ON_DEVICE
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_scalar), "v"(start));
    return r;
#else
    return dot2(a, b_scalar, start);
#endif
}

// Two such casts and the packing of their low halves in two instructions instead of three: v_cvt_i32_f32 with an SDWA
// destination puts the low 16 bits of its result into one half of the register and leaves (or clears) the other.
// (round 5; IQD_NO_CVT_SDWA: the two plain conversions and a v_perm_b32, for the A/B)
IQD_DEV uint32_t cast_pack_i16_bounded(float lo, float hi)
{
#if IQD_ON_DEVICE && !defined(IQD_NO_CVT_SDWA)
    uint32_t r;
    asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(r) : "v"(lo));
    asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(hi));
    return r;
#else
    return ((uint32_t)(int32_t)lo & 0xffffu) | ((uint32_t)(int32_t)hi << 16);
#endif
}

// low halves of two dwords -> one dword (a.lo | b.lo << 16), v_perm_b32
IQD_DEV uint32_t pack_lo16(uint32_t a, uint32_t b) { return perm(b, a, 0x05040100u); }
// high halves of two dwords -> one dword (a.hi | b.hi << 16)
IQD_DEV uint32_t pack_hi16(uint32_t a, uint32_t b) { return perm(b, a, 0x07060302u); }

}  // namespace iqd
