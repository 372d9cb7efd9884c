#include <hip/hip_runtime.h>
#include <stdio.h>
#define T_MFMA16 1
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ __forceinline__ int chunk_k(int s2, int half, int j) { return 8 * (2 * s2 + (j >> 2)) + 4 * half + (j & 3); }
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// feature (0 .. 15) of a 16-feature half that row m of the A operand must carry so that the swapped-back accumulator is in
// the layout convention (register 4a + b of lane half h = feature 8a + 4h + b of the 32-block)
__host__ __device__ __forceinline__ int perm16(int m) { return (m & 3) | ((m & 4) << 1) | ((m & 8) >> 1); }
// contraction index (0 .. 31) that element j of lane group kg of a swapped B operand holds
__host__ __device__ __forceinline__ int k_of16(int kg, int j) { return chunk_k(kg & 1, kg >> 1, j); }

// v_permlane16_swap_b32 as an asm statement: through __builtin_amdgcn_permlane16_swap hipcc 7.2 folded sixteen swaps of
// lane-affine values into three and copied the results around (tools/ubench/swap_rt.hip: every component became component 0
// of the partner row).  Four swaps per statement; two wait states in front (a VALU result feeding a permlane swap) and
// behind (its results feeding a VALU or matrix instruction) -- hipcc pads neither side of an asm statement.
__device__ __forceinline__ void swap16x4(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned& b0, unsigned& b1,
                                         unsigned& b2, unsigned& b3) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %4\n\t"
                 "v_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\t"
                 "v_permlane16_swap_b32 %3, %7\n\t"
                 "s_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
}
// (R0, R1) <-> (X0, X1) for the three planes of a 32-deep block (an involution)
__device__ __forceinline__ void swap_planes(bf16x8 (&a)[3], bf16x8 (&b)[3]) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4_t ua = __builtin_bit_cast(u32x4_t, a[p]), ub = __builtin_bit_cast(u32x4_t, b[p]);
        unsigned x0 = ua[0], x1 = ua[1], x2 = ua[2], x3 = ua[3], y0 = ub[0], y1 = ub[1], y2 = ub[2], y3 = ub[3];
        swap16x4(x0, x1, x2, x3, y0, y1, y2, y3);
        ua = u32x4_t{x0, x1, x2, x3};
        ub = u32x4_t{y0, y1, y2, y3};
        a[p] = __builtin_bit_cast(bf16x8, ua);
        b[p] = __builtin_bit_cast(bf16x8, ub);
    }
}
__device__ __forceinline__ void swap_f4(float& a0, float& a1, float& a2, float& a3, float& b0, float& b1, float& b2, float& b3) {
    unsigned x0 = __builtin_bit_cast(unsigned, a0), x1 = __builtin_bit_cast(unsigned, a1), x2 = __builtin_bit_cast(unsigned, a2),
             x3 = __builtin_bit_cast(unsigned, a3), y0 = __builtin_bit_cast(unsigned, b0), y1 = __builtin_bit_cast(unsigned, b1),
             y2 = __builtin_bit_cast(unsigned, b2), y3 = __builtin_bit_cast(unsigned, b3);
    swap16x4(x0, x1, x2, x3, y0, y1, y2, y3);
    a0 = __builtin_bit_cast(float, x0); a1 = __builtin_bit_cast(float, x1); a2 = __builtin_bit_cast(float, x2); a3 = __builtin_bit_cast(float, x3);
    b0 = __builtin_bit_cast(float, y0); b1 = __builtin_bit_cast(float, y1); b2 = __builtin_bit_cast(float, y2); b3 = __builtin_bit_cast(float, y3);
}
// a 32 x 32 accumulator tile between the E form and the four 16 x 16 tiles (sub-tile, feature half) at registers
// 4 (2 fb + sub) .. + 3; also a 4 x f32x4 segment of x or Q' (same register numbering)
__device__ __forceinline__ void swap_tile(f32x16& t) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float a0 = t[8 * fb], a1 = t[8 * fb + 1], a2 = t[8 * fb + 2], a3 = t[8 * fb + 3];
        float b0 = t[8 * fb + 4], b1 = t[8 * fb + 5], b2 = t[8 * fb + 6], b3 = t[8 * fb + 7];
        swap_f4(a0, a1, a2, a3, b0, b1, b2, b3);
        t[8 * fb] = a0; t[8 * fb + 1] = a1; t[8 * fb + 2] = a2; t[8 * fb + 3] = a3;
        t[8 * fb + 4] = b0; t[8 * fb + 5] = b1; t[8 * fb + 6] = b2; t[8 * fb + 7] = b3;
    }
}
__device__ __forceinline__ void swap_seg(f32x4 (&x)[4]) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float a0 = x[2 * fb][0], a1 = x[2 * fb][1], a2 = x[2 * fb][2], a3 = x[2 * fb][3];
        float b0 = x[2 * fb + 1][0], b1 = x[2 * fb + 1][1], b2 = x[2 * fb + 1][2], b3 = x[2 * fb + 1][3];
        swap_f4(a0, a1, a2, a3, b0, b1, b2, b3);
        x[2 * fb] = f32x4{a0, a1, a2, a3};
        x[2 * fb + 1] = f32x4{b0, b1, b2, b3};
    }
}


__global__ void k(float* out) {
    const int lane = threadIdx.x;
    f32x4 x[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 4; ++i) x[a][i] = lane * 100 + 4 * a + i;
    swap_seg(x);
    f32x16 t;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 4; ++i) t[4 * a + i] = x[a][i];
    swap_tile(t);
    for (int e = 0; e < 16; ++e) out[lane * 16 + e] = t[e];
}
int main() {
    float* d; float h[1024];
    hipMalloc(&d, 4096);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) bad += h[l * 16 + e] != l * 100 + e;
    printf("round trip mismatches: %d\n", bad);
    for (int e = 0; e < 16; ++e) printf("%g ", h[17 * 16 + e]);
    printf("\n");
    return 0;
}
