// fp32 GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  on the gfx950 fp32-input matrix cores.
//
// Replaces the nn.Linear / Conv1d(k=1) calls of the reference hot path
// (models/transformer.py:79-81,83,87; models/pointnet.py:60).  fp32 in, fp32 accumulate:
// v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain, so results are "exact fp32" in the
// same sense as the reference's sgemm (only the summation order differs).
//
// Geometry (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD for f32 MFMA, 160 KiB LDS/CU):
//   block tile 128 x 256, BK = 32, 256 threads = 4 waves stacked in M; wave tile 32 x 256
//   = 8 MFMA tiles of 32x32 -> 128 accumulator registers; 2 blocks per CU (64 KiB LDS each).
//   The contraction index is a dummy, so any k permutation shared by A and W is legal: within a
//   32-deep k-tile lane (r, half) owns k = 16*half + 4*j + i (j, i = 0..3) of row r for BOTH operands.
//   * A (activations): every wave only ever needs its own 32 rows, so A never touches LDS.  Lane (r, half)
//     loads its 64 contiguous bytes of row r per k-tile with four global_load_dwordx4 straight into
//     registers (the two half-waves together consume each 128-byte line exactly once), one tile ahead.
//   * W (weights, shared by the four waves): LDS-DMA (global_load_lds_dwordx4), double buffered, one tile
//     ahead, no VGPRs and no ds_write pass.  The DMA writes LDS lane-linearly (8 rows of 128 B per wave
//     instruction), so rows are unpadded and the bank-conflict fix is an XOR swizzle applied to the per-lane
//     SOURCE address and to the read: 16-byte chunk c of row n lives at chunk c ^ ((n >> 1) & 7).  The 16
//     lanes of a ds_read_b128 group then cover 16 distinct (n & 1, c') slots -> conflict free.
//   * one barrier per k-tile (the DMA of tile t+1 lands under the MFMAs of tile t); fragment reads are
//     software-pipelined half a k-chunk ahead of their MFMAs.
//   Epilogue: a wave owns whole 256-wide rows, so LayerNorm statistics never leave the wave, and
//   the accumulators go through a wave-private LDS slab (8 rows at a time) to turn the MFMA layout
//   (lane = column) into row-major float4 per lane: every residual load and output store is one
//   full 1 KiB row per wave instruction.
//   Persistent blocks: the grid is at most 2 blocks per CU and every block walks its tiles (stride = grid),
//   issuing the first k-tile of its NEXT output tile (DMA + A registers) before it starts the epilogue of
//   the current one, so neither the workgroup dispatch gap (~5k cycles) nor the first-load latency (~10k
//   cycles, s_memtime stamps in profiles/r01_gemm_stamps.txt) sits on the critical path any more.
//   Tiles are numbered so that the N-tiles of one M-tile run back to back on ONE XCD (shared A
//   rows in that XCD's L2) -- placement only changes speed, never results.
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 256;
constexpr int THREADS = 256;
constexpr int MAX_GRID = 512;   // 256 CUs x 2 resident blocks

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// BK is a template parameter only so that the staging geometry below is written once; BK = 32 is what ships
// (BK = 64 needs 128 KiB of LDS -> one block per CU: measured 117 vs 130 TFLOP/s, profiles/r01_gemm_ablation.txt).
template <int EPI, int BK>
__global__ __launch_bounds__(THREADS, 2) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ W,
                                                             float* __restrict__ C, int64_t ldc, int n_tiles,
                                                             unsigned total_tiles, int K, EpiArgs ep) {
    constexpr int WTILE = BN * BK;   // floats per W tile buffer (unpadded, swizzled)
    constexpr int NJ = BK / 8;       // 16-byte chunks per lane per k-tile
    constexpr int NC = BK / 4;       // 16-byte chunks per W row
    constexpr int RPI = 64 / NC;     // W rows per DMA wave-instruction (64 lanes x 16 B = 1 KiB)
    constexpr int NQ = BN / RPI / 4; // DMA instructions per wave per k-tile
    __shared__ __attribute__((aligned(16))) float smem[2 * WTILE + 256];  // two W tiles (+1 KiB); buffer 1 doubles as epilogue slabs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, half = lane >> 5;

    // Tile numbering (bijective for any tile count): virtual ids v, v+8, v+16, ... share an XCD (the grid is a
    // multiple of 8 whenever a block owns more than one tile), give them consecutive tiles so the n_tiles
    // tiles that read one block of A rows hit the same L2.
    const unsigned q8 = total_tiles >> 3, rem = total_tiles & 7u;
    auto tile_of = [&](unsigned v) {
        const unsigned xcd = v & 7u;
        return (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (v >> 3);
    };

    // W DMA: wave-instruction q of this wave fills LDS rows (wave*NQ+q)*RPI .. +RPI-1 (1 KiB, lane-linear):
    // lane -> row nq = (wave*NQ+q)*RPI + lane/NC, destination chunk c' = lane % NC, source chunk c = c' ^ swz(nq)
    // with swz(n) = (n >> 1) & 7 for 128-byte rows (BK 32) and n & 15 for 256-byte rows (BK 64): the 16 lanes of a
    // ds_read_b128 group (16 distinct rows mod 16, one chunk index) then cover 16 distinct 16-byte bank slots.
    auto swz = [](int n) { return BK == 32 ? ((n >> 1) & 7) : (n & 15); };
    const int64_t w_qstride = (int64_t)RPI * K;
    // B fragment read: row n = tn*32 + r, chunk (half*NJ + j) -> c' = (half*NJ + j) ^ swz(r)
    int boff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = r * BK + (((half * NJ + j) ^ swz(r)) << 2);

    auto dma_w = [&](const float* gw, int buf, int kt) {
        const float* src = gw + (int64_t)kt * BK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (lane % NC) ^ swz(q * RPI + lane / NC);  // (wave*NQ*RPI is a multiple of 16)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + q * w_qstride + c * 4),
                                             (lptr_t)(smem + buf * WTILE + (wave * NQ + q) * 256), 16, 0, 0);
        }
    };
    // A: lane (r, half) streams 64 contiguous bytes of its row per k-tile
    auto a_ptr = [&](int64_t m0) { return A + (m0 + wave * 32 + r) * lda + half * (BK / 2); };
    auto w_ptr = [&](int n0) { return W + (int64_t)(n0 + wave * 64 + lane / NC) * K; };

    const int KT = K / BK;  // even (host check): every output tile starts on LDS buffer 0 / register set a0
    unsigned v = blockIdx.x;
    unsigned tile = tile_of(v);
    int64_t m0 = (int64_t)(tile / n_tiles) * BM;
    int n0 = (int)(tile % n_tiles) * BN;
    const float* ga = a_ptr(m0);
    const float* gw = w_ptr(n0);

    f32x4 a0[NJ], a1[NJ];  // A fragments of the current / next k-tile (named, so indices stay static)
    dma_w(gw, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) a0[j] = ld4(ga + j * 4);
    __syncthreads();  // with an LDS-DMA in flight hipcc drains vmcnt(0) here: k-tile 0 has landed for every wave

    for (;;) {
        f32x16 acc[8];
#pragma unroll
        for (int tn = 0; tn < 8; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;

        // one k-tile: prefetch k-tile kt+1 (DMA + A registers), 128 MFMAs on k-tile kt, barrier
        auto step = [&](int kt, int buf, f32x4 (&ac)[NJ], f32x4 (&an)[NJ]) {
            // Retire the loads of this tile's A registers HERE, while nothing younger is in flight (they were
            // issued a whole tile ago and the barrier already drained them): with an LDS-DMA outstanding hipcc
            // would otherwise put s_waitcnt vmcnt(0) in front of the first MFMA and serialise tile t+1's
            // transfer with tile t's math.
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(ac[j]));
            if (kt + 1 < KT) {
                dma_w(gw, buf ^ 1, kt + 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) an[j] = ld4(ga + (kt + 1) * BK + j * 4);
            }
            const float* wb = smem + buf * WTILE;
            f32x4 fb[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[0][t] = ld4(wb + t * 32 * BK + boff[0]);
#pragma unroll
            for (int hc = 0; hc < 2 * NJ; ++hc) {  // half-chunk hc: j = hc >> 1, N-tiles (hc & 1)*4 .. +3
                const int cur = hc & 1, nxt = cur ^ 1;
                if (hc + 1 < 2 * NJ) {
                    const int j2 = (hc + 1) >> 1, th2 = (hc + 1) & 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) fb[nxt][t] = ld4(wb + (th2 * 4 + t) * 32 * BK + boff[j2]);
                }
                const int j = hc >> 1, th = hc & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[th * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j][i], fb[cur][t][i], acc[th * 4 + t], 0, 0, 0);
                // Pin the order inside this half-chunk: first MFMA (its lgkmcnt wait then only covers reads issued a
                // half-chunk ago), then the 4 prefetch reads, then the other 15 MFMAs.  With an LDS-DMA in flight hipcc
                // emits lgkmcnt(0) rather than a counted wait, so the prefetch must sit BEHIND that first MFMA.
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // all reads of `buf` done; k-tile kt+1 landed (vmcnt(0) drained by the barrier's fence)
        };
        for (int kt = 0; kt < KT; kt += 2) {
            step(kt, 0, a0, a1);
            step(kt + 1, 1, a1, a0);
        }

        // ---- next output tile: start its first k-tile now so that it lands under the epilogue ----------------
        const unsigned v_next = v + gridDim.x;
        const bool has_next = v_next < total_tiles;
        const int64_t m0_cur = m0;
        const int n0_cur = n0;
        if (has_next) {
            tile = tile_of(v_next);
            m0 = (int64_t)(tile / n_tiles) * BM;
            n0 = (int)(tile % n_tiles) * BN;
            ga = a_ptr(m0);
            gw = w_ptr(n0);
        }
        // RES_LN's residual loads would be waited for with vmcnt(0) behind the DMA (hipcc, LDS-DMA in flight), so that
        // epilogue prefetches afterwards; the others have no loads and take the prefetch first.
        if (EPI != SCREAM_EPI_RES_LN && has_next) {
            dma_w(gw, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) a0[j] = ld4(ga + j * 4);
        }

        __builtin_amdgcn_s_setprio(2);  // the epilogue's LDS/VALU/store issue is otherwise starved by the partner block's MFMA stream
        // buffer 1 (+ the 1 KiB tail) is free after the k-loop's last barrier: epilogue scratch
        gemm_epilogue<EPI, 4>(acc, smem + WTILE, wave, lane, tid, true, m0_cur, n0_cur, ep, C, ldc);
        __builtin_amdgcn_s_setprio(0);
        if (!has_next) break;
        v = v_next;
        if (EPI == SCREAM_EPI_RES_LN) {
            dma_w(gw, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) a0[j] = ld4(ga + j * 4);
        }
        __syncthreads();  // next tile's k-tile 0 landed (vmcnt(0)); every wave is done with its slab
    }
}

template <int EPI>
int launch(const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int64_t M, int N, int K,
           const EpiArgs& ep, hipStream_t st) {
    const int n_tiles = N / BN;
    const int64_t total = (M / BM) * n_tiles;
    if (total == 0) return 0;
    SCREAM_REQUIRE(total < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = total < MAX_GRID ? (unsigned)total : (unsigned)MAX_GRID;
    gemm_f32_kernel<EPI, 32><<<dim3(grid), dim3(THREADS), 0, st>>>(A, lda, W, C, ldc, n_tiles, (unsigned)total, K, ep);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int scream_gemm_f32(const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int64_t M,
                               int32_t N, int32_t K, int32_t epilogue, int32_t n_act, const float* bias,
                               const float* residual, int64_t ldr, const float* gamma, const float* beta,
                               void* stream) {
    SCREAM_REQUIRE(A && W && C, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % BM == 0 && N > 0 && N % BN == 0 && K > 0 && K % 64 == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(C) & 15) == 0, SCREAM_EINVAL);
    EpiArgs ep{n_act, bias, residual, ldr, gamma, beta, nullptr, nullptr, nullptr, nullptr, 0};
    hipStream_t st = as_stream(stream);
    switch (epilogue) {
        case SCREAM_EPI_NONE:
            return launch<SCREAM_EPI_NONE>(A, lda, W, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_ELU1:
            SCREAM_REQUIRE(n_act >= 0 && n_act % BN == 0, SCREAM_EUNSUPPORTED);
            return launch<SCREAM_EPI_ELU1>(A, lda, W, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_RELU:
            return launch<SCREAM_EPI_RELU>(A, lda, W, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_BIAS_RELU:
            SCREAM_REQUIRE(bias, SCREAM_EINVAL);
            return launch<SCREAM_EPI_BIAS_RELU>(A, lda, W, C, ldc, M, N, K, ep, st);
        case SCREAM_EPI_RES_LN:
            SCREAM_REQUIRE(N == BN, SCREAM_EUNSUPPORTED);
            SCREAM_REQUIRE(residual && gamma && beta && ldr >= N && ldr % 4 == 0, SCREAM_EINVAL);
            SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(residual) & 15) == 0, SCREAM_EINVAL);
            return launch<SCREAM_EPI_RES_LN>(A, lda, W, C, ldc, M, N, K, ep, st);
        default:
            return SCREAM_EINVAL;
    }
}

extern "C" int scream_gemm_qkv_f32(const float* A, int64_t lda, const float* W, float* Q, int64_t ldq, int64_t M,
                                   int32_t N, int32_t K, int32_t n_q, const int32_t* tile_cloud,
                                   const int32_t* cloud_row0, const int32_t* cloud_len, int64_t row_base,
                                   float* kv_partial, void* stream) {
    SCREAM_REQUIRE(A && W && kv_partial && tile_cloud && cloud_row0 && cloud_len, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % BM == 0 && N > 0 && N % BN == 0 && K > 0 && K % 64 == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE((n_q == 0 || n_q == BN) && N == n_q + 2 * BN && row_base >= 0 && row_base % BM == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(n_q == 0 || (Q && ldq >= n_q && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0), SCREAM_EINVAL);
    SCREAM_REQUIRE(lda >= K && lda % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0, SCREAM_EINVAL);
    EpiArgs ep{n_q, nullptr, nullptr, 0, nullptr, nullptr, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base};
    return launch<SCREAM_EPI_QKV>(A, lda, W, Q, ldq, M, N, K, ep, as_stream(stream));
}
