// Per-CU store throughput: `blocks` workgroups of `waves` waves each stream 1 KiB-per-wave-instruction dwordx4 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* buf, int iters, long long* stamps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float* mine = buf + ((size_t)blockIdx.x * nw + wave) * ((size_t)iters * 8 * 256);
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4* p = reinterpret_cast<f32x4*>(mine + ((size_t)it * 8 + u) * 256 + lane * 4);
            if (MODE == 0) *p = v;
            else __builtin_nontemporal_store(v, p);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
    float* buf; long long* stamps; static long long hs[256 * 8];
    const int iters = 32;  // 256 KiB per wave
    hipMalloc(&buf, (size_t)256 * 8 * iters * 8 * 1024); hipMalloc(&stamps, sizeof(hs));
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {8, 64, 256})
            for (int waves : {1, 2, 4, 8}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) k<0><<<blocks, waves * 64>>>(buf, iters, stamps); else k<1><<<blocks, waves * 64>>>(buf, iters, stamps);
                    hipDeviceSynchronize();
                }
                hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
                double t = 0; int n = 0;
                for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { t += (double)hs[b * 8 + w]; ++n; }
                t /= n;
                printf("%s blocks %3d waves %d: %8.0f cycles for %4d KiB per CU -> %6.1f B/clk/CU\n", mode ? "nontemporal" : "plain      ", blocks, waves, t,
                       waves * iters * 8, waves * iters * 8 * 1024.0 / t);
            }
    return 0;
}
