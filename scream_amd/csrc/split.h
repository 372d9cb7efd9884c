// Operand splits that give an fp32-accurate product on the 16-bit matrix cores of gfx950, as policy types shared by the
// GEMM kernel (gemm_split.hip) and the layer-tail kernel (tail_split.hip).  Both kernels are templates over one of:
//
//   SplitBf3   x = x0 + x1 + x2 in bf16 (3 x 8 significand bits: exact), six products a2b0 + a1b1 + a0b2 + a1b0 + a0b1 + a0b0
//              on v_mfma_f32_32x32x16_bf16.  bf16 keeps fp32's exponent range: no scaling, scale invariant bit for bit.
//              (Rounds 1-2; still built, selected with SCREAM_GEMM=x3, and the arithmetic of the attention apply.)
//   SplitH2    x * 2^e = x0 + x1 in fp16 (2 x 11 significand bits), THREE products a1b0 + a0b1 + a0b0 on
//              v_mfma_f32_32x32x16_f16 -- half the matrix instructions, two thirds of the operand bytes.  Round 3, the default.
//              fp16 has 5 exponent bits, so every operand carries an exact power-of-two scale 2^e chosen so that
//              |x| 2^e <= 2^15 holds for EVERY value the operand can take (scream_amd/scales.py derives the bounds from the
//              weights: LayerNorm outputs are bounded by their gamma / beta, a projection of one by the norms of its rows, an
//              attention output by its values).  Nothing can overflow, so there is no run-time flag; values below
//              2^-3 / 2^e lose relative (not absolute) precision: 2^-25 / 2^e against a bound of 2^15 / 2^e, i.e. 2^-40 of
//              the operand's range.  Measured against float64 with the hardware's own accumulation
//              (tools/ubench/split_acc.py, profiles/r03_split_acc.txt): same normwise error as SplitBf3.
//              The accumulators are in units of 2^(ea + ew); the kernels fold that exact factor into what follows (a
//              LayerNorm of c z with eps c^2 equals the LayerNorm of z bit for bit), never into an extra rounding.
#pragma once
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SplitBf3 {
    static constexpr int NP = 3;        // operand planes
    static constexpr int NPROD = 6;     // matrix instructions per 16-deep step
    static constexpr bool SCALED = false;
    typedef bf16x8 vec;
    static __device__ __forceinline__ void split1(float x, int i, vec (&p)[3]) {
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        p[0][i] = a;
        p[1][i] = b;
        p[2][i] = (__bf16)(r1 - (float)b);
    }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    // acc (+)= x . y: smallest terms first (plane index sum 2, then 1), the exact leading product last.  x is the MFMA's
    // first operand (M side).  MIRROR walks the equal-magnitude terms from the other end (the GEMM kernel's historical
    // order: its results stay bit-identical to round 2's).  (Orders that keep one operand in place across consecutive
    // instructions draw the same joules.)
    template <bool MIRROR = false>
    static __device__ __forceinline__ void products(f32x16& acc, const vec (&x)[3], const vec (&y)[3], const f32x16& c0) {
        if (MIRROR) {
            acc = mfma(x[2], y[0], c0);
            acc = mfma(x[1], y[1], acc);
            acc = mfma(x[0], y[2], acc);
            acc = mfma(x[1], y[0], acc);
            acc = mfma(x[0], y[1], acc);
        } else {
            acc = mfma(x[0], y[2], c0);
            acc = mfma(x[1], y[1], acc);
            acc = mfma(x[2], y[0], acc);
            acc = mfma(x[0], y[1], acc);
            acc = mfma(x[1], y[0], acc);
        }
        acc = mfma(x[0], y[0], acc);
    }
};

struct SplitH2 {
    static constexpr int NP = 2;
    static constexpr int NPROD = 3;
    static constexpr bool SCALED = true;
    typedef f16x8 vec;
    static __device__ __forceinline__ void split1(float x, int i, vec (&p)[2]) {  // x already carries its 2^e
        const _Float16 a = (_Float16)x;           // round to nearest even (v_cvt_pk_f16_f32)
        p[0][i] = a;
        p[1][i] = (_Float16)(x - (float)a);       // the residual is exact in fp32
    }
    // the same planes of x * s for an exact power of two s, written so that each plane is ONE instruction (v_fma_mix{lo,hi}_f16:
    // an fp32 fma whose result is rounded to fp16 straight into one half of the destination, its addend read as fp16): the product
    // x s is exact, so fp16(fma(x, s, 0)) and fp16(fma(x, s, -x0)) are bit for bit split1(x * s) -- two instructions per element
    // where multiply, convert, convert back, subtract, convert and pack are three and a half (proj_ring.hip: every vector
    // instruction of an epilogue is paid in time at the socket power cap)
    static __device__ __forceinline__ void split1s(float x, float s, int i, vec (&p)[2]) {
        const _Float16 a = (_Float16)__builtin_fmaf(x, s, 0.0f);
        p[0][i] = a;
        p[1][i] = (_Float16)__builtin_fmaf(x, s, -(float)a);
    }
    // elements i (even) and i + 1 at once, the four v_fma_mix instructions spelled out: left to itself hipcc turns half of the
    // pairs back into multiply + v_cvt_pk_f16_f32 + two converts + two fmas + v_cvt_pk (eight instructions where these are four)
    static __device__ __forceinline__ void split2s(float x0, float x1, float s, int i, vec (&p)[2]) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned hi, lo;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(s), "v"(hi));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(s), "v"(hi));
        u32x4 a = __builtin_bit_cast(u32x4, p[0]), b = __builtin_bit_cast(u32x4, p[1]);
        a[i >> 1] = hi;
        b[i >> 1] = lo;
        p[0] = __builtin_bit_cast(vec, a);
        p[1] = __builtin_bit_cast(vec, b);
    }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    template <bool MIRROR = false>
    static __device__ __forceinline__ void products(f32x16& acc, const vec (&x)[2], const vec (&y)[2], const f32x16& c0) {
        // (round 4: the order x1 y0, x0 y0, x0 y1 -- every product sharing one operand with its predecessor -- measured 0.35 % slower)
        acc = mfma(x[0], y[1], c0);
        acc = mfma(x[1], y[0], acc);
        acc = mfma(x[0], y[0], acc);
    }
};

// ONE fp16 plane, one product: NOT an fp32-accurate split -- the mirror of the reference's own reduced-precision mode, the
// `with autocast()` around the KITTI forward (evaluate_kitti.py:37: fp16 matrix products with fp32 accumulation on CUDA, a
// no-op on the CPU path that parity is defined on).  Same kernels, same weight-derived exponents as SplitH2 (which also keep
// every fp16 operand in range); selected only by an explicit autocast=True / gemm_backend "h1", never a default, never the
// headline.  Error per product 2^-11 relative instead of 2^-22.
struct SplitH1 {
    static constexpr int NP = 1;
    static constexpr int NPROD = 1;
    static constexpr bool SCALED = true;
    typedef f16x8 vec;
    static __device__ __forceinline__ void split1(float x, int i, vec (&p)[1]) { p[0][i] = (_Float16)x; }
    static __device__ __forceinline__ void split1s(float x, float s, int i, vec (&p)[1]) { p[0][i] = (_Float16)__builtin_fmaf(x, s, 0.0f); }
    static __device__ __forceinline__ void split2s(float x0, float x1, float s, int i, vec (&p)[1]) {
        split1s(x0, s, i, p);
        split1s(x1, s, i + 1, p);
    }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    template <bool MIRROR = false>
    static __device__ __forceinline__ void products(f32x16& acc, const vec (&x)[1], const vec (&y)[1], const f32x16& c0) {
        acc = mfma(x[0], y[0], c0);
    }
};

// eight consecutive operand values (two f32x4) -> planes (SplitH2: the values already carry the operand's 2^e)
template <class SP>
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, typename SP::vec (&p)[SP::NP]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) SP::split1(i < 4 ? lo[i] : hi[i - 4], i, p);
}

// the same for (lo, hi) * s with an exact power of two s (1 for an operand that carries its scale already): the fp16 splits go
// through split2s (one v_fma_mix per plane and element), the bf16 split ignores s (it is scale free; callers pass 1)
template <class SP>
__device__ __forceinline__ void split8s(const f32x4 lo, const f32x4 hi, float s, typename SP::vec (&p)[SP::NP]) {
    if constexpr (SP::SCALED) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) SP::split2s(i < 4 ? lo[i] : hi[i - 4], i < 4 ? lo[i + 1] : hi[i - 3], s, i, p);
    } else {
        split8<SP>(lo, hi, p);
    }
}

// 2^e as a float, e in [-126, 127]
static inline float exp2i(int e) {
    union { uint32_t u; float f; } v;
    v.u = (uint32_t)(e + 127) << 23;
    return v.f;
}

}  // namespace
