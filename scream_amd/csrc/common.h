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

__device__ __forceinline__ float wave_sum(float v) {
    v = half_wave_sum(v);
    v += __shfl_xor(v, 32);
    return v;
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
