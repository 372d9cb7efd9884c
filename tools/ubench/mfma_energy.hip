// Energy per flop of the two bf16 MFMA shapes of gfx950 on register operands that change from instruction to instruction
// (random bf16 data): 1 wave per SIMD, 256 blocks, nothing but MFMAs.  tools/ubench/mfma_energy.py times the launches
// while rocm-smi samples socket power and sclk.   SHAPE 0: v_mfma_f32_32x32x16_bf16 (16 KFLOP x 2, 8 passes);
// SHAPE 1: v_mfma_f32_16x16x32_bf16 (same flops per instruction pair: two of them per 32x32x16).
// SHAPE 2 (round 3): v_mfma_f32_32x32x16_f16 on fp16 operands -- the instruction of the 2-plane / 3-product split.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_burn(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = ops[(i * 256 + tid) % 4096];
        b[i] = ops[((i + 8) * 256 + tid) % 4096];
    }
    if (SHAPE == 2) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[(i + j) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[(i + j) & 3], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[t][e];
        out[blockIdx.x * 256 + tid] = s;
    } else if (SHAPE == 0) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[(i + j) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[(i + j) & 3], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[t][e];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // two 16x16x32 = the flops of one 32x32x16
                    acc[(i + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[(i + j) & 7], 0, 0, 0);
                    acc[(i + j + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[(i + j + 4) & 7], 0, 0, 0);
                }
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[t][e];
        out[blockIdx.x * 256 + tid] = s;
    }
}

extern "C" int mfma_burn_launch(int shape, const void* ops, float* out, int iters, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (shape == 2) mfma_burn<2><<<dim3(256), dim3(256), 0, st>>>(reinterpret_cast<const bf16x8*>(ops), out, iters);
    else if (shape == 0) mfma_burn<0><<<dim3(256), dim3(256), 0, st>>>(reinterpret_cast<const bf16x8*>(ops), out, iters);
    else mfma_burn<1><<<dim3(256), dim3(256), 0, st>>>(reinterpret_cast<const bf16x8*>(ops), out, iters);
    return (int)hipGetLastError();
}
