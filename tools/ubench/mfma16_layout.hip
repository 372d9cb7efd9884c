// Layout check for the 16x16x32 path of tail_x3.hip: E-form operand planes -> v_permlane16_swap -> v_mfma_f32_16x16x32_bf16 ->
// swap back must give Y[row][feature] = sum_k W[feature][k] X[row][k] in the kernel's register convention.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__host__ __device__ int chunk_k(int s2, int half, int j) { return 8 * (2 * s2 + (j >> 2)) + 4 * half + (j & 3); }
__host__ __device__ int perm16(int m) { return (m & 3) | ((m & 4) << 1) | ((m & 8) >> 1); }
__host__ __device__ int k_of16(int kg, int j) { return chunk_k(kg & 1, kg >> 1, j); }
__device__ void swap16(unsigned& a, unsigned& b) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__global__ void k(const float* X, const float* W, float* Y, unsigned* dbg) {
    const int lane = threadIdx.x, r = lane & 31, half = lane >> 5, m16 = lane & 15, kg = lane >> 4;
    bf16x8 R[2], A[2];
    for (int s2 = 0; s2 < 2; ++s2)
        for (int j = 0; j < 8; ++j) R[s2][j] = (__bf16)X[r * 32 + chunk_k(s2, half, j)];
    for (int fb = 0; fb < 2; ++fb)
        for (int j = 0; j < 8; ++j) A[fb][j] = (__bf16)W[(16 * fb + perm16(m16)) * 32 + k_of16(kg, j)];
    // debug: what does the swap do to lane ids?
    unsigned a = lane, b = 100 + lane;
    swap16(a, b);
    dbg[lane] = a; dbg[64 + lane] = b;
    u32x4_t u0 = __builtin_bit_cast(u32x4_t, R[0]), u1 = __builtin_bit_cast(u32x4_t, R[1]);
    for (int i = 0; i < 4; ++i) { unsigned x = u0[i], y = u1[i]; swap16(x, y); u0[i] = x; u1[i] = y; }
    const bf16x8 X0 = __builtin_bit_cast(bf16x8, u0), X1 = __builtin_bit_cast(bf16x8, u1);
    float acc[16];
    for (int fb = 0; fb < 2; ++fb) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[fb], X0, z, 0, 0, 0);
        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[fb], X1, z, 0, 0, 0);
        for (int i = 0; i < 4; ++i) { acc[8 * fb + i] = d0[i]; acc[8 * fb + 4 + i] = d1[i]; }
    }
    for (int fb = 0; fb < 2; ++fb)
        for (int i = 0; i < 4; ++i) {
            unsigned x = __builtin_bit_cast(unsigned, acc[8 * fb + i]), y = __builtin_bit_cast(unsigned, acc[8 * fb + 4 + i]);
            swap16(x, y);
            acc[8 * fb + i] = __builtin_bit_cast(float, x); acc[8 * fb + 4 + i] = __builtin_bit_cast(float, y);
        }
    for (int a4 = 0; a4 < 4; ++a4)
        for (int bb = 0; bb < 4; ++bb) Y[r * 32 + 8 * a4 + 4 * half + bb] = acc[4 * a4 + bb];
}
int main() {
    float hX[32 * 32], hW[32 * 32], hY[32 * 32], ref[32 * 32];
    for (int i = 0; i < 1024; ++i) { hX[i] = (float)((i * 7 + 3) % 13 - 6); hW[i] = (float)((i * 5 + 1) % 11 - 5); }
    for (int r = 0; r < 32; ++r) for (int f = 0; f < 32; ++f) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hW[f * 32 + kk] * hX[r * 32 + kk]; ref[r * 32 + f] = s; }
    float *dX, *dW, *dY; unsigned* dD; unsigned hD[128];
    hipMalloc(&dX, 4096); hipMalloc(&dW, 4096); hipMalloc(&dY, 4096); hipMalloc(&dD, 512);
    hipMemcpy(dX, hX, 4096, hipMemcpyHostToDevice); hipMemcpy(dW, hW, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dX, dW, dY, dD);
    hipMemcpy(hY, dY, 4096, hipMemcpyDeviceToHost); hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hY[i] != ref[i];
    printf("mismatches: %d of 1024\n", bad);
    printf("swap(a = lane, b = 100 + lane): a' lanes 0,16,32,48 = %u %u %u %u ; b' = %u %u %u %u\n", hD[0], hD[16], hD[32], hD[48], hD[64], hD[80], hD[96], hD[112]);
    if (bad) for (int r = 0; r < 2; ++r) { for (int f = 0; f < 32; ++f) printf("%g/%g ", hY[r * 32 + f], ref[r * 32 + f]); printf("\n"); }
    return bad != 0;
}
