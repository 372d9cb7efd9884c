// Is "LDS-DMA 16 B per lane from scattered rows -> s_waitcnt vmcnt(0) -> s_barrier -> ds_read_b128 of the own bytes" sound?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ __launch_bounds__(512) void k(const float* A, int64_t lda, int rows, int iters, int mode, unsigned* bad) {
    __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = (int)((blockIdx.x * 977u + it * 131u + wave * 32 + r16) % (unsigned)(rows - 16));
        const float* p = A + (int64_t)row * lda + kg * 8 + (it & 7) * 32;
        const float* p2 = p + 16 * lda;
        char* slot = smem + 48 * 1024 + wave * 4096;
        // make LDS busy with other traffic first (as the epilogue does)
        float* slab = reinterpret_cast<float*>(smem + 96 * 1024) + wave * 2048;
        for (int j = 0; j < 8; ++j) slab[j * 256 + lane] = (float)(it + j);
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)slot, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + 4), (lptr_t)(slot + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)p2, (lptr_t)(slot + 2048), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p2 + 4), (lptr_t)(slot + 3072), 16, 0, 0);
        if (mode & 1) {  // stores in flight, as after an epilogue
            f32x4 v = {1.f, 2.f, 3.f, 4.f};
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(const_cast<float*>(A) + ((int64_t)rows + blockIdx.x * 64 + wave * 8 + j) * lda + lane * 4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
        if (mode & 2) __builtin_amdgcn_s_barrier();
        f32x4 g[4];
        for (int j = 0; j < 4; ++j) g[j] = *reinterpret_cast<const f32x4*>(slot + j * 1024 + lane * 16);
        f32x4 d[4] = {*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), *reinterpret_cast<const f32x4*>(p2), *reinterpret_cast<const f32x4*>(p2 + 4)};
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 4; ++c) nbad += g[j][c] != d[j][c];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // everyone done reading before the next iteration's DMA
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const int rows = 4096, lda = 256 + 64;
    float* A; unsigned* bad;
    hipMalloc(&A, (size_t)(rows + 256 * 64 + 64) * lda * 4); hipMalloc(&bad, 4);
    float* h = new float[(size_t)rows * lda];
    for (size_t i = 0; i < (size_t)rows * lda; ++i) h[i] = (float)(i % 1000003) * 0.5f;
    hipMemcpy(A, h, (size_t)rows * lda * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(bad, 0, 4);
        k<<<256, 512>>>(A, lda, rows, 20000, mode, bad);
        unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("mode %d (stores in flight: %d, barrier before read: %d): mismatching floats %u of %llu\n", mode, mode & 1, (mode >> 1) & 1, hb, 256ull * 512 * 16 * 20000);
    }
    return 0;
}
