// Shared helpers for the gfx950 kernels of libscream_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scream_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SCREAM_LAUNCH_CHECK()                         \
    do {                                              \
        hipError_t e_ = hipGetLastError();            \
        if (e_ != hipSuccess) return (int)e_;         \
    } while (0)

#define SCREAM_REQUIRE(cond, code) \
    do {                           \
        if (!(cond)) return (code); \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Row index inside a 32x32 MFMA accumulator tile held by lane (lane>>5 = half) in register reg.
// (cdna_hip_programming.md section 3: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half.)
__device__ __forceinline__ int mfma32_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float half_wave_sum(float v) {
    // Sum over the 32 lanes that share lane>>5 (xor masks < 32 never cross the half).
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP inclusive scan (row_shr 1/2/4/8 inside each 16-lane
// row, then row_bcast15 into rows 1/3 and row_bcast31 into rows 2/3) leaves the total in lane 63; six VALU
// instructions and one v_readlane instead of six dependent ds_bpermute round trips through the LDS pipeline.
__device__ __forceinline__ float wave_sum(float v) {
#define SCREAM_DPP_ADD(ctrl, row_mask)                                                                      \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, \
                                                               0xf, true))
    SCREAM_DPP_ADD(0x111, 0xf);  // row_shr:1 (out-of-row lanes read 0: bound_ctrl)
    SCREAM_DPP_ADD(0x112, 0xf);  // row_shr:2
    SCREAM_DPP_ADD(0x114, 0xf);  // row_shr:4
    SCREAM_DPP_ADD(0x118, 0xf);  // row_shr:8  -> lane 15 of each row holds the row sum
    SCREAM_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    SCREAM_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef SCREAM_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double wave_sum_f64(double v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// elu(x) + 1 (models/transformer.py:7-8: the feature map of the linear attention) = x + 1 for x > 0, exp(x) otherwise.  The
// exponential runs on the hardware's exp2 (v_exp_f32, 1 ulp) with the rounding of x * log2(e) -- product and constant -- carried
// into a first-order correction, exp2(hi) (1 + lo ln 2): within 2 ulp of exp(x), where expf() spends 14 vector instructions on
// range handling that an argument <= 0 never needs (2 x 10^9 of these per 32-pair step, in epilogues no MFMA hides).
// Round 4: no compare / select.  exp2's result is CLAMPED to [0, 1] (an output modifier of the instruction: free), so the
// exponential branch is exact for x <= 0 and ~1 for x > 0, and since e^x >= 1 + x everywhere, elu(x) + 1 = max(1 + x, that):
// eight instructions instead of ten (at the socket power cap every vector instruction of an epilogue is paid in time).
// elu1s(t, c): elu(t c) + 1 for an accumulator t in units of 1 / c, c an exact power of two: the constants carry c (a power of
// two commutes with every rounding), so the scaling multiply in front of the epilogue disappears -- bit for bit elu1(t * c).
// Valid for |x| < 2^126 (beyond that x log2(e) overflows and 0 * inf appears).
struct Elu1Consts {
    float c, cl2e, cl2e_lo;  // c, c * log2(e) rounded to fp32, c * (log2(e) - that)
};
static inline Elu1Consts elu1_consts(float c) { return Elu1Consts{c, c * 1.44269502162933349609375f, c * 1.925963033500011e-08f}; }
__device__ __forceinline__ float elu1s(float t, const Elu1Consts& k) {
    const float LN2 = 0.693147182464599609375f;
    const float hi = t * k.cl2e;
    const float lo = __builtin_fmaf(t, k.cl2e_lo, __builtin_fmaf(t, k.cl2e, -hi));
    const float r = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(hi), 0.0f, 1.0f);  // folds into v_exp_f32 ... clamp
    const float e = __builtin_fmaf(r, lo * LN2, r);
    return fmaxf(__builtin_fmaf(t, k.c, 1.0f), e);
}
__device__ __forceinline__ float elu1(float x) {
    return elu1s(x, Elu1Consts{1.0f, 1.44269502162933349609375f, 1.925963033500011e-08f});
}

#ifndef SCREAM_MAX_GRID
#define SCREAM_MAX_GRID 256  // blocks of the persistent grids: one per CU (tuning builds: fewer)
#endif
