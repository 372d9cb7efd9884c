// fp32-accurate GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  on the gfx950 bf16 matrix cores by 3-way operand splitting.
//
//   x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)     (exact: 3 x 8 significand bits = fp32)
//   a.b ~= a2b0 + a1b1 + a0b2 + a1b0 + a0b1 + a0b0                                  (dropped terms <= 2^-24 relative)
//
// Every bf16 x bf16 product is exact in fp32 and the six v_mfma_f32_32x32x16_bf16 of a 16-deep step accumulate in
// fp32, so the result carries fp32-level error -- measured against fp64 it is slightly MORE accurate than the
// fp32-input MFMA path (max |err| / sum|a||b|: 2.0e-7 vs 2.6e-7 at K = 256, 2.8e-7 vs 3.4e-7 at K = 1024,
// tools/x3_bench.py) -- while the matrix pipe spends 6 x 32 = 192 cycles per 32x32x16 block instead of 8 x 64 = 512.
// Same inputs, same outputs, same epilogues and tolerances as gemm_f32.hip; the dtype of the path stays fp32.
//
// Geometry: block tile 256 x 256, 512 threads = 8 waves stacked in M (wave tile 32 x 256, 128 accumulator VGPRs), one
// persistent block per CU.
//   * W is split and re-tiled once (scream_pack_w_x3) into an image that is stored k-tile by k-tile exactly as it sits
//     in LDS: [3 planes][K/32][N][32] bf16, rows of 64 B whose 16-byte chunk c lives at c ^ ((n >> 2) & 3) so that the 16
//     lanes of a ds_read_b128 group cover 16 distinct bank slots.  A k-tile stage is 48 one-KiB LDS-DMA pieces of
//     CONTIGUOUS memory (with a plain [N][K] plane every piece touched 16 lines for half their bytes); double buffered.
//   * A stays fp32 in HBM: lane (r, half) streams its 64 contiguous bytes of row r per k-tile straight into registers
//     (three register sets, requested TWO k-tiles ahead) and splits them there (v_cvt_pk_bf16_f32 + subtract, twice).
//     Lane-half h owns k = 16h + 8s + j of step s for both operands.
//   * One barrier per k-tile with COUNTED waits: requests are issued in a fixed order (D(kt+1), then A(kt+2), pinned
//     with sched_barrier) so that s_waitcnt vmcnt(4) at the barrier covers the W stage and leaves the A loads of the
//     k-tile after next in flight; the barrier is followed directly by MFMAs (the first step's operand split was done
//     at the end of the previous k-tile, the requests go out between the two steps).
//   * The A loads are plain inline asm: hipcc's own vmcnt bookkeeping cannot count across the epilogue's stores and
//     the loop back edge and would wait for vmcnt(0) everywhere.  Counted waits are used only where nothing but loads
//     is in flight; the first barrier of an output tile, behind the previous epilogue's stores, drains the queue.
//   * The two waves of a SIMD are not symmetric: the older one gets the matrix pipe first (s_memtime stamps,
//     tools/x3_stamps.py: 1.8 k cycles for its first 48 MFMAs against 3.8 k for the younger wave's), so the younger
//     waves (4-7) do their operand split AFTER the barrier, where they would be starved anyway, the older ones before.
//   * A dedicated 64 KiB LDS region holds the epilogue slabs, so the first k-tile of the next output tile is already
//     in flight during the epilogue (for every epilogue kind).
// What was measured and rejected on the way (tools/x3_ablate.py, tools/ubench/, profiles/r01_x3_ablation.txt): two
// independent 128-row blocks per CU (doubles the W traffic; same speed), one wave per SIMD with 64-row wave tiles and
// 512 registers (slower: a lone in-order wave does not keep the pipe full), spreading the fragment reads between the
// MFMAs, starting the CUs out of phase, non-temporal stores.  On this chip a wave that issues MFMAs back to back
// starves the LDS and vector-memory instructions of the other wave on its SIMD (not its VALU), which is why the
// k-tile time is close to the SUM of the MFMA, LDS, VMEM and VALU issue times rather than their maximum.
// Tuning aid (tools/x3_ablate.py builds variants): bit 0 no epilogue, 1 no W DMA after the first k-tile, 2 no A loads
// after the first, 3 no MFMAs, 4 no LDS fragment reads, 5 no operand split.  Always 0 in libscream_hip.so.
#ifndef X3_ABLATE
#define X3_ABLATE 0
#endif
#ifndef X3_PARK
#define X3_PARK 1
#endif
#include <type_traits>

#include "../../scream_amd/csrc/gemm_epilogue.h"

#ifdef X3_STAMPS  // tuning aid: s_memtime stamps of one output tile per block (tools/x3_stamps.py)
__device__ long long x3_stamps[256 * 8 * 160];
extern "C" int scream_x3_stamps_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(x3_stamps), sizeof(long long) * 256 * 8 * 160);
}
#define STAMP(slot)                                                                                    \
    do {                                                                                               \
        if (stamp_on && lane == 0 && (slot) < 160)                                                      \
            x3_stamps[((int)blockIdx.x * 8 + wave) * 160 + (slot)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int XBM = 256, XBN = 256, XBK = 32, XTHREADS = 512, XWAVES = 8;
constexpr int PLANE_BYTES = XBN * XBK * 2;     // 16 KiB
constexpr int STAGE_BYTES = 3 * PLANE_BYTES;   // 48 KiB
constexpr int SLAB_BYTES = XWAVES * 8 * 256 * 4;  // 64 KiB
constexpr int X_MAX_GRID = 256;

__device__ __forceinline__ void split3(const f32x4 lo, const f32x4 hi, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? lo[i] : hi[i - 4];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[i] = a;
        p1[i] = b;
        p2[i] = (__bf16)(r1 - (float)b);
    }
}

// s_waitcnt vmcnt(N) lgkmcnt(0) + workgroup barrier: the N youngest vector-memory operations of this wave (the A
// loads of the k-tile after next) stay in flight across the barrier.
template <int N>
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
}

// ---- epilogues for the 16x16 accumulator layout of this kernel ---------------------------------------------------
// acc[rt][ct][e] = C[row = rt*16 + 4*kg + e][col = ct*16 + r16] of the wave's 32 x 256 tile (r16 = lane & 15,
// kg = lane >> 4).  The row-wise epilogue is the one of gemm_epilogue.h (8-row slabs, float4 rows, LayerNorm statistics
// inside the wave); only the slab write differs: rows 8g .. 8g+7 live in the lanes with kg >> 1 == (g & 1).
template <int EPI>
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[2][16], float* slabs, int wave, int lane, bool rows_exist,
                                           int64_t m0_cur, int n0_cur, const EpiArgs& ep, float* __restrict__ C, int64_t ldc) {
    if (!rows_exist) return;
    const int r16 = lane & 15, kg = lane >> 4;
    constexpr int SLAB_LD = 256;
    float* slab = slabs + wave * (8 * SLAB_LD);
    const int col = n0_cur + lane * 4;
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};  // bias | gamma, beta
    if (EPI == SCREAM_EPI_BIAS_RELU) p0 = ld4(ep.bias + col);
    if (EPI == SCREAM_EPI_RES_LN) {
        p0 = ld4(ep.gamma + col);
        p1 = ld4(ep.beta + col);
    }
    const bool act = n0_cur < ep.n_act;
    f32x4 rsd[2][4];
    if (EPI == SCREAM_EPI_RES_LN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rsd[0][i] = ld4(ep.residual + (m0_cur + wave * 32 + i) * ep.ldr + col);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // rows 8g .. 8g+7 of the wave's 32
        if ((kg >> 1) == (g & 1)) {
#pragma unroll
            for (int ct = 0; ct < 16; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e)  // rows 4..7 are stored with column bit 4 flipped: the two 16-lane groups hit disjoint banks
                    slab[(4 * (kg & 1) + e) * SLAB_LD + ((ct * 16 + r16) ^ ((kg & 1) << 4))] = acc[g >> 1][ct][e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vv[i] = ld4(slab + (hh * 4 + i) * SLAB_LD + ((lane * 4) ^ (hh << 4)));
            const int64_t row0 = m0_cur + wave * 32 + 8 * g + 4 * hh;
            if (EPI == SCREAM_EPI_RES_LN) {
                const int hcur = (2 * g + hh) & 1;
                if (2 * g + hh + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rsd[hcur ^ 1][i] = ld4(ep.residual + (row0 + 4 + i) * ep.ldr + col);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vv[i] += rsd[hcur][i];
                    const float mean = wave_sum((vv[i][0] + vv[i][1]) + (vv[i][2] + vv[i][3])) * (1.0f / 256.0f);
                    const f32x4 d = vv[i] - mean;
                    const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f);
                    const float rstd = 1.0f / sqrtf(var + 1e-5f);
                    vv[i] = d * rstd * p0 + p1;
                }
            } else if (EPI == SCREAM_EPI_ELU1 || EPI == SCREAM_EPI_QKV) {
                if (act) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c) vv[i][c] = vv[i][c] > 0.f ? vv[i][c] + 1.0f : expf(vv[i][c]);
                }
            } else if (EPI == SCREAM_EPI_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) vv[i][c] = fmaxf(vv[i][c], 0.f);
            } else if (EPI == SCREAM_EPI_BIAS_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) vv[i][c] = fmaxf(vv[i][c] + p0[c], 0.f);
            }
#if X3_ABLATE
            if (X3_ABLATE & 64) {
                if (vv[0][0] + vv[1][1] + vv[2][2] + vv[3][3] != 123.456f) continue;
            }
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(C + (row0 + i) * ldc + col) = vv[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Fused K^T V (see gemm_epilogue.h) for the 16x16 layout: a key/value tile holds K (columns 0-127 = ct 0-7) and V
// (ct 8-15) of four heads for the same tokens; lane (r16, kg) of K'acc[rt][ct][e] holds token rt*16 + 4*kg + e of
// feature column ct*16 + r16 -- the operand layout of v_mfma_f32_16x16x4_f32 (lane & 15 = row/column, lane >> 4 = k)
// with the TOKEN as the contraction index, for K' (as A, rows = d) and V (as B, columns = v) alike.
__device__ __forceinline__ void kv_epilogue16(f32x4 (&acc)[2][16], float* slabs, int wave, int lane, int tid, bool rows_exist,
                                              int64_t m0_cur, int n0_cur, const EpiArgs& ep) {
    const int r16 = lane & 15, kg = lane >> 4;
    const int grp = wave >> 2, wg = wave & 3;  // four waves = one 128-row tile = one cloud
    const int64_t mrow = m0_cur + grp * SCREAM_ROW_TILE;
    int valid_w = 0;
    float* part = ep.kv_partial;
    if (rows_exist) {
        const int cloud = ep.tile_cloud[(ep.row_base + mrow) / SCREAM_ROW_TILE];
        valid_w = ep.cloud_len[cloud] - (int)(ep.row_base + mrow - ep.cloud_row0[cloud]) - wg * 32;  // real tokens in this wave's rows
        const int hb = (n0_cur - ep.n_act) / XBN * 4;
        part += ((int64_t)(mrow / SCREAM_ROW_TILE) * SCREAM_NHEAD + hb) * KV_ELEMS;
    }
#pragma unroll
    for (int hq = 0; hq < 4; ++hq) {  // one head per LDS round: 8 x KV_ELEMS floats of scratch
        f32x4 kv[2][2];  // [d tile][v tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[a][b][e] = 0.f;
        float ks[2] = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool real = rt * 16 + 4 * kg + e < valid_w;  // padding rows do not exist
                float k2[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    float a = acc[rt][2 * hq + dt][e];
                    a = a > 0.f ? a + 1.0f : expf(a);  // elu(k) + 1
                    k2[dt] = real ? a : 0.f;
                    ks[dt] += k2[dt];
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt)
                        kv[dt][vt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k2[dt], acc[rt][8 + 2 * hq + vt][e], kv[dt][vt], 0, 0, 0);
            }
        float* sw = slabs + (grp * 4 + wg) * KV_ELEMS;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            ks[dt] += __shfl_xor(ks[dt], 16);
            ks[dt] += __shfl_xor(ks[dt], 32);
#pragma unroll
            for (int vt = 0; vt < 2; ++vt)
#pragma unroll
                for (int e = 0; e < 4; ++e) sw[(dt * 16 + 4 * kg + e) * 32 + vt * 16 + r16] = kv[dt][vt][e];  // [d][v]
            if (kg == 0) sw[32 * 32 + dt * 16 + r16] = ks[dt];
        }
        lds_barrier();
        if (rows_exist) {
            const int t4 = tid & 255;
            for (int i = t4; i < KV_ELEMS; i += 256) {
                const float* s4 = slabs + grp * 4 * KV_ELEMS + i;
                part[hq * KV_ELEMS + i] = (s4[0] + s4[KV_ELEMS]) + (s4[2 * KV_ELEMS] + s4[3 * KV_ELEMS]);
            }
        }
        lds_barrier();
    }
}

template <int EPI>
__global__ __launch_bounds__(XTHREADS, 2) void gemm_x3_kernel(const float* __restrict__ A, int64_t lda,
                                                             const __bf16* __restrict__ Wp, float* __restrict__ C,
                                                             int64_t ldc, int64_t M, int n_tiles, unsigned total_tiles,
                                                             int N, int K, EpiArgs ep) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + SLAB_BYTES];  // 160 KiB
    float* slabs = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const unsigned q8 = total_tiles >> 3, rem = total_tiles & 7u;
    auto tile_of = [&](unsigned v) {
        const unsigned xcd = v & 7u;
        return (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (v >> 3);
    };
    const __bf16* w_lane = Wp + lane * 8;
    auto dma_w = [&](int n0, int stage, int kt) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int id = wave * 6 + u, plane = id >> 4, q = id & 15;
            const int64_t soff = (((int64_t)plane * (K / XBK) + kt) * N + n0 + q * 16) * XBK;
            __builtin_amdgcn_global_load_lds((gptr_t)(w_lane + soff),
                                             (lptr_t)(smem + stage * STAGE_BYTES + plane * PLANE_BYTES + q * 1024), 16, 0, 0);
        }
    };
    auto a_ptr = [&](int64_t m0, bool ok) { return A + (ok ? m0 + wave * 32 + r16 : m0) * lda + kg * 8; };
    const int boff = r16 * 64 + ((kg ^ ((r16 >> 2) & 3)) << 4);
    const int KT = K / XBK;
    unsigned v = blockIdx.x;
    unsigned tile = tile_of(v);
    int64_t m0 = (int64_t)(tile / n_tiles) * XBM;
    int n0 = (int)(tile % n_tiles) * XBN;
    bool rows_ok = m0 + wave * 32 < M;
    const float* ga = a_ptr(m0, rows_ok);
    const int64_t rt_stride = 16 * lda;
    f32x4 a0[4], a1[4], a2[4];  // [2*rt + h]: 8 floats of row-tile rt
    auto load_a = [&](f32x4 (&a)[4], int kt) {
        const float* p = ga + kt * XBK;
        const float* p2 = p + (rows_ok ? rt_stride : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[0]) : "v"(p));
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(a[1]) : "v"(p));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[2]) : "v"(p2));
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(a[3]) : "v"(p2));
    };
    // First requests of an output tile.  They go out before the previous tile's epilogue, but not into registers: the
    // compiler does not know that an inline-asm load is still in flight and would happily spill such a register under
    // the epilogue's pressure.  The W stage is LDS-DMA anyway; the first k-tile of A is PARKED in LDS by LDS-DMA as well
    // (each lane's own 4 x 16 bytes, in stage 1, which is idle until the first k-tile's mid-step) and picked up with
    // four ds_read_b128 after the epilogue.
    auto park_a0 = [&]() {
        const float* p = ga;
        const float* p2 = p + (rows_ok ? rt_stride : 0);
        char* slot = smem + STAGE_BYTES + wave * 4096;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)slot, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + 4), (lptr_t)(slot + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)p2, (lptr_t)(slot + 2048), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p2 + 4), (lptr_t)(slot + 3072), 16, 0, 0);
    };
    auto unpark_a0 = [&]() {  // after vmcnt(0) and a barrier: every lane reads back what its own loads brought.
        // Inline asm on purpose: a plain LDS load is free to be sunk towards its first use -- behind the next barrier,
        // where another wave's DMA of k-tile 1 may already be overwriting stage 1.
        const unsigned addr = (unsigned)(STAGE_BYTES + wave * 4096 + lane * 16);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                     "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0[0]), "=&v"(a0[1]), "=&v"(a0[2]), "=&v"(a0[3]) : "v"(addr) : "memory");
    };
    dma_w(n0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(a0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, 1);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 pa[2][3];
    const bool late = wave >= 4;
    for (;;) {
        f32x4 acc[2][16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 16; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rt][ct][e] = 0.f;
        auto split_all = [&](f32x4 (&ac)[4]) {
            asm volatile("" : "+v"(ac[0]));
            asm volatile("" : "+v"(ac[1]));
            asm volatile("" : "+v"(ac[2]));
            asm volatile("" : "+v"(ac[3]));
            split3(ac[0], ac[1], pa[0][0], pa[0][1], pa[0][2]);
            split3(ac[2], ac[3], pa[1][0], pa[1][1], pa[1][2]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto step = [&](auto tail, auto first, int kt, int stage, f32x4 (&ac)[4], f32x4 (&an1)[4], f32x4 (&an2)[4]) {
            constexpr int TAIL = decltype(tail)::value;
            constexpr bool FIRST = decltype(first)::value;
            if (TAIL == 2) ring_barrier<0>(); else ring_barrier<4>();
            if (FIRST || late) split_all(ac);
            const char* wb = smem + stage * STAGE_BYTES + boff;
            bf16x8 fb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[0][p] = *reinterpret_cast<const bf16x8*>(wb + p * PLANE_BYTES);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const int cur = ct & 1, nxt = cur ^ 1;
                if (ct + 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) fb[nxt][p] = *reinterpret_cast<const bf16x8*>(wb + p * PLANE_BYTES + (ct + 1) * 1024);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][2], fb[cur][0], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][1], fb[cur][1], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][0], fb[cur][2], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][1], fb[cur][0], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][0], fb[cur][1], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][0], fb[cur][0], acc[rt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ct == 7) {
                    if (TAIL <= 1) dma_w(n0, stage ^ 1, kt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (TAIL == 0) load_a(an2, kt + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (TAIL <= 1 && !late) {
                if (TAIL == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 10); else __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
                split_all(an1);
            }
        };
        constexpr std::integral_constant<int, 0> steady{};
        constexpr std::integral_constant<int, 1> tail1{};
        constexpr std::integral_constant<int, 2> tail2{};
        constexpr std::integral_constant<bool, true> first{};
        constexpr std::integral_constant<bool, false> later{};
        if (KT > 2) {
            step(steady, first, 0, 0, a0, a1, a2);
            step(steady, later, 1, 1, a1, a2, a0);
            step(steady, later, 2, 0, a2, a0, a1);
            for (int kt = 3; kt < KT - 2; kt += 3) {
                step(steady, later, kt, kt & 1, a0, a1, a2);
                step(steady, later, kt + 1, (kt + 1) & 1, a1, a2, a0);
                step(steady, later, kt + 2, kt & 1, a2, a0, a1);
            }
            step(tail1, later, KT - 2, 0, a0, a1, a2);
        } else {
            step(tail1, first, 0, 0, a0, a1, a2);
        }
        step(tail2, later, KT - 1, 1, a1, a2, a2);

        const unsigned v_next = v + gridDim.x;
        const bool has_next = v_next < total_tiles;
        const int64_t m0_cur = m0;
        const int n0_cur = n0;
        const bool rows_cur = rows_ok;
        if (has_next) {
            tile = tile_of(v_next);
            m0 = (int64_t)(tile / n_tiles) * XBM;
            n0 = (int)(tile % n_tiles) * XBN;
            rows_ok = m0 + wave * 32 < M;
            ga = a_ptr(m0, rows_ok);
        }
        if (has_next) {
            dma_w(n0, 0, 0);   // stage 0 was last read two barriers ago
            lds_barrier();     // stage 1 was read in the last k-tile: every wave must be done before A is parked there
#if X3_PARK == 1 || X3_PARK == 4 || X3_PARK == 5
            park_a0();
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (X3_ABLATE & 1) {
            float keep = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 16; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep += acc[rt][ct][e];
            if (keep == 123.456f) C[0] = keep;
        } else if (EPI == SCREAM_EPI_QKV && n0_cur >= ep.n_act) {
            kv_epilogue16(acc, slabs, wave, lane, tid, rows_cur, m0_cur, n0_cur, ep);
        } else {
            epilogue16<EPI>(acc, slabs, wave, lane, rows_cur, m0_cur, n0_cur, ep, C, ldc);
        }
        if (!has_next) break;
        v = v_next;
#if X3_PARK == 2  // experiment: park after the epilogue
        park_a0();
#endif
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the epilogue's stores and this wave's parked loads
#if X3_PARK == 4  // experiment: compare the parked bytes with a direct load
        __builtin_amdgcn_s_barrier();
        unpark_a0();
        {
            f32x4 t[4];
            const float* p = ga;
            const float* p2 = p + (rows_ok ? rt_stride : 0);
            t[0] = ld4(p); t[1] = ld4(p + 4); t[2] = ld4(p2); t[3] = ld4(p2 + 4);
            int nb = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) nb += t[j][c] != a0[j][c];
            if (nb) atomicAdd(reinterpret_cast<int*>(const_cast<float*>(ep.gamma)), nb);
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] = t[j];
        }
#elif X3_PARK == 5  // experiment: park (write LDS) but never read it back
        __builtin_amdgcn_s_barrier();
        load_a(a0, 0);
#elif X3_PARK == 3  // experiment: the extra barrier without parking
        __builtin_amdgcn_s_barrier();
        load_a(a0, 0);
#elif X3_PARK
        __builtin_amdgcn_s_barrier();  // LDS-DMA data is ordered for a ds_read only by the wait AND a barrier behind it,
        unpark_a0();                   // even for the wave that issued it (without: one wrong 32-row block in ~10^5 tiles)
#else
        load_a(a0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        load_a(a1, 1);  // the queue now holds nothing but these four loads: the counted first barrier is sound
        __builtin_amdgcn_sched_barrier(0);
    }
}

// W [N][K] fp32 -> packed planes [3][K/32][N][32] bf16; 16-byte chunk c of a row's 32-deep k-slice is stored at
// chunk c ^ ((n >> 2) & 3).  One thread per (n, k-tile, stored chunk).
__global__ void pack_w_x3_kernel(const float* __restrict__ W, int N, int K, __bf16* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KT = K / XBK;
    if (t >= (int64_t)N * KT * 4) return;
    const int cs = (int)(t & 3), n = (int)((t >> 2) % N), kt = (int)((t >> 2) / N);
    const int c = cs ^ ((n >> 2) & 3);
    const float* src = W + (int64_t)n * K + kt * XBK + c * 8;
    bf16x8 p0, p1, p2;
    split3(ld4(src), ld4(src + 4), p0, p1, p2);
    const int64_t plane = (int64_t)KT * N * XBK;
    __bf16* dst = out + ((int64_t)kt * N + n) * XBK + cs * 8;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + plane) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * plane) = p2;
}

template <int EPI>
int launch_x3(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, int64_t M, int N, int K,
              const EpiArgs& ep, hipStream_t st) {
    const int n_tiles = N / XBN;
    const int64_t total = ((M + XBM - 1) / XBM) * n_tiles;
    if (total == 0) return 0;
    SCREAM_REQUIRE(total < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = total < X_MAX_GRID ? (unsigned)total : (unsigned)X_MAX_GRID;
    gemm_x3_kernel<EPI><<<dim3(grid), dim3(XTHREADS), 0, st>>>(A, lda, reinterpret_cast<const __bf16*>(Wp), C, ldc, M,
                                                               n_tiles, (unsigned)total, N, K, ep);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int scream_pack_w_x3(const float* W, int32_t N, int32_t K, void* packed, void* stream) {
    SCREAM_REQUIRE(W && packed, SCREAM_EINVAL);
    SCREAM_REQUIRE(N > 0 && N % 4 == 0 && K > 0 && K % XBK == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, SCREAM_EINVAL);
    const int64_t threads = (int64_t)N * (K / XBK) * 4;
    pack_w_x3_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        W, N, K, reinterpret_cast<__bf16*>(packed));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_gemm_x3_f32(const float* A, int64_t lda, const void* W_planes, float* C, int64_t ldc, int64_t M,
                                  int32_t N, int32_t K, int32_t epilogue, int32_t n_act, const float* bias,
                                  const float* residual, int64_t ldr, const float* gamma, const float* beta,
                                  void* stream) {
    SCREAM_REQUIRE(A && W_planes && C, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0 && N > 0 && N % XBN == 0 && K >= 64 && K % 32 == 0 && (K / 32 - 2) % 3 == 0, SCREAM_EUNSUPPORTED);  // K = 64 + 96 j
    SCREAM_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W_planes) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(C) & 15) == 0, SCREAM_EINVAL);
    EpiArgs ep{n_act, bias, residual, ldr, gamma, beta, nullptr, nullptr, nullptr, nullptr, 0};
    hipStream_t st = as_stream(stream);
    switch (epilogue) {
        case SCREAM_EPI_NONE:
            return launch_x3<SCREAM_EPI_NONE>(A, lda, W_planes, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_ELU1:
            SCREAM_REQUIRE(n_act >= 0 && n_act % XBN == 0, SCREAM_EUNSUPPORTED);
            return launch_x3<SCREAM_EPI_ELU1>(A, lda, W_planes, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_RELU:
            return launch_x3<SCREAM_EPI_RELU>(A, lda, W_planes, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_BIAS_RELU:
            SCREAM_REQUIRE(bias, SCREAM_EINVAL);
            return launch_x3<SCREAM_EPI_BIAS_RELU>(A, lda, W_planes, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_RES_LN:
            SCREAM_REQUIRE(N == XBN, SCREAM_EUNSUPPORTED);
            SCREAM_REQUIRE(residual && gamma && beta && ldr >= N && ldr % 4 == 0, SCREAM_EINVAL);
            SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(residual) & 15) == 0, SCREAM_EINVAL);
            return launch_x3<SCREAM_EPI_RES_LN>(A, lda, W_planes, C, ldc, M, N, K, ep, st);
        default:
            return SCREAM_EINVAL;
    }
}

extern "C" int scream_gemm_qkv_x3_f32(const float* A, int64_t lda, const void* W_planes, float* Q, int64_t ldq, int64_t M,
                                      int32_t N, int32_t K, int32_t n_q, const int32_t* tile_cloud,
                                      const int32_t* cloud_row0, const int32_t* cloud_len, int64_t row_base,
                                      float* kv_partial, void* stream) {
    SCREAM_REQUIRE(A && W_planes && kv_partial && tile_cloud && cloud_row0 && cloud_len, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0 && N > 0 && N % XBN == 0 && K >= 64 && K % 32 == 0 && (K / 32 - 2) % 3 == 0, SCREAM_EUNSUPPORTED);  // K = 64 + 96 j
    SCREAM_REQUIRE((n_q == 0 || n_q == XBN) && N == n_q + 2 * XBN && row_base >= 0 && row_base % SCREAM_ROW_TILE == 0,
                   SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(n_q == 0 || (Q && ldq >= n_q && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0), SCREAM_EINVAL);
    SCREAM_REQUIRE(lda >= K && lda % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W_planes) & 15) == 0, SCREAM_EINVAL);
    EpiArgs ep{n_q, nullptr, nullptr, 0, nullptr, nullptr, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base};
    return launch_x3<SCREAM_EPI_QKV>(A, lda, W_planes, Q, ldq, M, N, K, ep, as_stream(stream));
}
