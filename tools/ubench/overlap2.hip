// Which unit does a saturating MFMA stream block?  256-thread blocks (4 waves = one per SIMD); per-wave roles from a table.
//   0 idle, 1 MFMA back-to-back, 2 L2-resident dwordx4 loads, 3 VALU fma chain, 4 LDS reads, 5 MFMA with s_nop 7 x2 between
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Roles { int r[8]; };
__global__ __launch_bounds__(512, 2) void k(Roles roles, int iters, float* buf, float* sink, long long* stamps) {
    __shared__ float lds[8192];
    const long long t0 = __builtin_amdgcn_s_memtime();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles.r[wave];
    lds[threadIdx.x] = (float)lane;
    __syncthreads();
    float* mine = buf + ((size_t)blockIdx.x * 8 + wave) * 4096;
    if (role == 1 || role == 5) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                    if (role == 5) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) sink[0] = s;
    } else if (role == 2) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const volatile f32x4*>(mine + ((it * 8 + u) & 15) * 256 + lane * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
    } else if (role == 3) {
        float x = (float)lane, y = 1.0001f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, y, 0.5f);
        }
        if (x == 123.456f) sink[0] = x;
    } else if (role == 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s += *reinterpret_cast<const volatile f32x4*>(lds + ((it + u) & 7) * 256 + lane * 4);
        }
        if (s[0] == 123.456f) sink[0] = s[0];
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
    float *buf, *sink; long long* stamps; static long long hs[256 * 8];
    hipMalloc(&buf, (size_t)256 * 8 * 4096 * 4); hipMalloc(&sink, 64); hipMalloc(&stamps, sizeof(hs));
    hipMemset(buf, 0, (size_t)256 * 8 * 4096 * 4);
    struct { Roles r; const char* name; } cases[] = {
        {{{1,0,0,0,0,0,0,0}}, "w0 MFMA"}, {{{2,0,0,0,0,0,0,0}}, "w0 L2loads"}, {{{3,0,0,0,0,0,0,0}}, "w0 VALU"}, {{{4,0,0,0,0,0,0,0}}, "w0 LDS"},
        {{{1,0,0,0,2,0,0,0}}, "w0 MFMA, w4 L2loads (same SIMD?)"}, {{{1,2,0,0,0,0,0,0}}, "w0 MFMA, w1 L2loads (other SIMD?)"},
        {{{1,0,0,0,3,0,0,0}}, "w0 MFMA, w4 VALU"}, {{{1,3,0,0,0,0,0,0}}, "w0 MFMA, w1 VALU"},
        {{{1,0,0,0,4,0,0,0}}, "w0 MFMA, w4 LDS"}, {{{1,4,0,0,0,0,0,0}}, "w0 MFMA, w1 LDS"},
        {{{5,0,0,0,0,0,0,0}}, "w0 MFMA+nops"}, {{{5,0,0,0,2,0,0,0}}, "w0 MFMA+nops, w4 L2loads"}, {{{5,0,0,0,3,0,0,0}}, "w0 MFMA+nops, w4 VALU"},
        {{{1,1,1,1,2,2,2,2}}, "w0-3 MFMA, w4-7 L2loads"}, {{{1,1,1,1,0,0,0,0}}, "w0-3 MFMA"}, {{{2,2,2,2,0,0,0,0}}, "w0-3 L2loads"}};
    for (auto& c : cases) {
        k<<<256, 512>>>(c.r, 1000, buf, sink, stamps);
        hipDeviceSynchronize();
        hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
        printf("%-36s", c.name);
        for (int w = 0; w < 8; ++w) { double t = 0; for (int b = 0; b < 256; ++b) t += (double)hs[b * 8 + w]; if (c.r.r[w]) printf("  w%d:%8.0f", w, t / 256); }
        printf("\n");
    }
    return 0;
}
