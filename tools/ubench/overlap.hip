// Microbenchmark: can one wave's vector-memory traffic (stores / loads) hide under the MFMAs of the OTHER wave on the
// same SIMD?  512-thread blocks, one per CU; waves 0-3 run role A, waves 4-7 role B (one of each per SIMD).
//   roles: 0 idle, 1 MFMA chain, 2 dwordx4 stores (1 KiB/wave-instr, streaming), 3 dwordx4 loads (streaming, L2-miss),
//          4 dwordx4 loads from a small (L2-resident) buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void k(int roleA, int roleB, int iters, float* buf, size_t stride_per_block, float* sink, long long* stamps) {
    const long long t_begin = __builtin_amdgcn_s_memtime();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int role = wave < 4 ? roleA : roleB;
    if (role >= 10) { role -= 10; __builtin_amdgcn_s_setprio(3); }  // 1x = the same role at raised priority
    float* mine = buf + (size_t)blockIdx.x * stride_per_block + (size_t)wave * (stride_per_block / 8);
    if (role == 1) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) sink[0] = s;
    } else if (role == 2) {
        f32x4 v = {1.f, 2.f, 3.f, (float)lane};
        const size_t n = stride_per_block / 8 / 256;  // 1 KiB pieces available (power of two)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t piece = ((size_t)it * 8 + u) & (n - 1);
                *reinterpret_cast<f32x4*>(mine + piece * 256 + lane * 4) = v;
            }
        }
    } else if (role == 3 || role == 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const size_t n = role == 3 ? stride_per_block / 8 / 256 : 16;
        for (int it = 0; it < iters; ++it) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t piece = ((size_t)it * 8 + u) & (n - 1);
                t[u] = *reinterpret_cast<const f32x4*>(mine + piece * 256 + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t_begin;
}

int main() {
    const int blocks = 256;
    const size_t per_block = (size_t)8 << 20;  // floats: 32 MiB per block -> 8 GiB total
    float *buf, *sink; long long* stamps; hipMalloc(&stamps, 256 * 8 * 8); long long hs[256 * 8];
    hipMalloc(&buf, per_block * blocks * sizeof(float));
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, per_block * blocks * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int a, b; const char* name; } cases[] = {
        {1, 0, "MFMA (4 waves)"}, {1, 1, "MFMA + MFMA"}, {2, 0, "stores (4 waves)"}, {2, 2, "stores + stores"},
        {1, 2, "MFMA + stores"}, {3, 0, "stream loads (4 waves)"}, {1, 3, "MFMA + stream loads"},
        {4, 0, "L2 loads (4 waves)"}, {1, 4, "MFMA + L2 loads"}, {2, 3, "stores + stream loads"}, {1, 12, "MFMA + stores@prio3"}, {1, 13, "MFMA + stream loads@prio3"}, {1, 14, "MFMA + L2 loads@prio3"}, {12, 1, "stores@prio3 + MFMA"}, {1, 11, "MFMA + MFMA@prio3"}};
    const int iters = 2000;  // 16000 MFMAs (512k cycles) | 16000 KiB stored per wave
    for (auto& c : cases) {
        const int ia = iters, ib = iters;
        (void)ib;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            // role-1 waves run `iters` x 32 MFMAs; memory roles run iters/8 x 8 pieces so both sides take a comparable time alone
            k<<<blocks, 512>>>(c.a, c.b, c.a == 1 || c.b == 1 ? ia : ia, buf, per_block, sink, stamps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
        double ta = 0, tb = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ta : tb) += (double)hs[b * 8 + w];
        printf("%-28s %8.3f ms   role A waves %9.0f cyc   role B waves %9.0f cyc (s_memtime, mean)\n", c.name, best, ta / 1024, tb / 1024);
    }
    return 0;
}
