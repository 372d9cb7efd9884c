// fp32 GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  on the gfx950 fp32-input matrix cores.
//
// Replaces the nn.Linear / Conv1d(k=1) calls of the reference hot path
// (models/transformer.py:79-81,83,87; models/pointnet.py:60).  fp32 in, fp32 accumulate:
// v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain, so results are "exact fp32" in the
// same sense as the reference's sgemm (only the summation order differs).
//
// Geometry (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD for f32 MFMA, 160 KiB LDS/CU):
//   block tile 128 x 256, BK = 32, 256 threads = 4 waves stacked in M; wave tile 32 x 256
//   = 8 MFMA tiles of 32x32 -> 128 accumulator registers; 2 blocks per CU (54 KiB LDS each)
//   so the second block's MFMAs cover the first block's staging/barriers.
//   Both operands are K-contiguous.  A k-chunk of 8 is split so that lane (r, half) holds
//   k = 4*half .. 4*half+3 of row r for BOTH operands: the contraction index is a dummy, so any
//   k permutation shared by A and W is legal, and it lets every fragment be one ds_read_b128.
//   LDS rows are padded to 36 floats: the four 16-lane groups of a ds_read_b128 then hit 16
//   distinct 4-bank slots (36 r mod 64 is a bijection on r mod 16) -> conflict free (measured:
//   SQ_LDS_BANK_CONFLICT = 0, profiles/r01_v1_bench_pmc_mfma_lds.txt).
//   Staging is global -> registers (issued before the MFMAs of the current tile) -> LDS
//   (written after them); fp32 MFMA is slow enough (64 cycles each) that this is fully hidden.
//   Epilogue: a wave owns whole 256-wide rows, so LayerNorm statistics never leave the wave, and
//   the accumulators go through a wave-private LDS slab (8 rows at a time) to turn the MFMA layout
//   (lane = column) into row-major float4 per lane: every residual load and output store is one
//   full 1 KiB row per wave instruction (v1 issued 128 dword stores per lane and lost ~25 % of the
//   MFMA time to store issue).
//   Blocks are renumbered so that the N-tiles of one M-tile run back to back on ONE XCD (shared A
//   tile in that XCD's L2) -- placement only changes speed, never results.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 256;
constexpr int BK = 32;
constexpr int LDS_LD = 36;  // floats per LDS row (32 + 4 pad)
constexpr int THREADS = 256;

struct EpiArgs {
    int n_act;
    const float* bias;
    const float* residual;
    int64_t ldr;
    const float* gamma;
    const float* beta;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int EPI>
__global__ __launch_bounds__(THREADS, 2) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ W,
                                                             float* __restrict__ C, int64_t ldc, int n_tiles,
                                                             int K, EpiArgs ep) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;

    // XCD-aware renumbering (bijective for any grid size): blocks b, b+8, b+16, ... share an XCD, give them
    // consecutive tiles so the n_tiles blocks that read one A tile hit the same L2.
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned xcd = bid & 7u, q = nb >> 3, rem = nb & 7u;
    const unsigned tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int64_t mt = tile / n_tiles;
    const int nt = tile % n_tiles;
    const int64_t m0 = mt * BM;
    const int n0 = nt * BN;

    // staging map: float4 index f = tid + 256 i -> row f >> 3, 16-byte column f & 7
    const int srow = tid >> 3, sc4 = tid & 7;
    const float* ga = A + (m0 + srow) * lda + sc4 * 4;
    const float* gw = W + (int64_t)(n0 + srow) * K + sc4 * 4;
    const int lds_st = srow * LDS_LD + sc4 * 4;

    f32x4 ra[4], rb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = ld4(ga + (int64_t)(32 * i) * lda);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = ld4(gw + (int64_t)(32 * i) * K);

    f32x16 acc[8];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;

#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + lds_st + 32 * i * LDS_LD) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Bs + lds_st + 32 * i * LDS_LD) = rb[i];
    __syncthreads();

    const float* a_frag = As + (wave * 32 + r) * LDS_LD + half * 4;
    const float* b_frag = Bs + r * LDS_LD + half * 4;
    const int KT = K / BK;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) {
            ga += BK;
            gw += BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = ld4(ga + (int64_t)(32 * i) * lda);
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[i] = ld4(gw + (int64_t)(32 * i) * K);
        }
        // 8 half-chunks (4 k-chunks x 2 halves of the 8 N-tiles), software pipelined: the LDS reads of
        // half-chunk h+1 are issued BEFORE the 16 MFMAs of half-chunk h, into the other fragment set.
        f32x4 fa[2], fb[2][4];
        fa[0] = ld4(a_frag);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[0][t] = ld4(b_frag + t * 32 * LDS_LD);
#pragma unroll
        for (int hc = 0; hc < 8; ++hc) {
            const int cur = hc & 1, nxt = cur ^ 1;
            if (hc + 1 < 8) {
                const int kk = (hc + 1) >> 1, th = (hc + 1) & 1;
                fa[nxt] = ld4(a_frag + kk * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[nxt][t] = ld4(b_frag + (th * 4 + t) * 32 * LDS_LD + kk * 8);
            }
            __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks these reads to just before their use
            const int th0 = hc & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[th0 * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][t][j], acc[th0 * 4 + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads above, the next half-chunk's below
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + lds_st + 32 * i * LDS_LD) = ra[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Bs + lds_st + 32 * i * LDS_LD) = rb[i];
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue (wave-private)
    // acc[tn][e]: row = wave*32 + mfma32_row(e, half), col = tn*32 + r.  The loop above ended on a barrier, so
    // the staging LDS is free: each wave takes its own slab and never synchronises with the others again.
    constexpr int SLAB_LD = 260;  // 256 + 4 floats
    float* slab = smem + wave * (8 * SLAB_LD);  // 8 rows; 4 x 8320 B <= 55296 B
    const int col = n0 + lane * 4;
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};  // bias | gamma, beta
    if (EPI == SCREAM_EPI_BIAS_RELU) p0 = ld4(ep.bias + col);
    if (EPI == SCREAM_EPI_RES_LN) {
        p0 = ld4(ep.gamma + col);
        p1 = ld4(ep.beta + col);
    }
    const bool act = n0 < ep.n_act;  // n_act is a multiple of 256: uniform per block
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // rows 8g .. 8g+7 of the wave's 32
#pragma unroll
        for (int tn = 0; tn < 8; ++tn)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[(i + 4 * half) * SLAB_LD + tn * 32 + r] = acc[tn][4 * g + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ld4(slab + i * SLAB_LD + lane * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t row0 = m0 + wave * 32 + 8 * g;
        if (EPI == SCREAM_EPI_RES_LN) {
            f32x4 res[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) res[i] = ld4(ep.residual + (row0 + i) * ep.ldr + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] += res[i];
                const float mean = wave_sum((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) * (1.0f / 256.0f);
                const f32x4 d = v[i] - mean;
                const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                v[i] = d * rstd * p0 + p1;
            }
        } else if (EPI == SCREAM_EPI_ELU1) {
            if (act) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[i][c] = v[i][c] > 0.f ? v[i][c] + 1.0f : expf(v[i][c]);  // elu(x)+1 == exp(x) for x <= 0
            }
        } else if (EPI == SCREAM_EPI_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[i][c] = fmaxf(v[i][c], 0.f);
        } else if (EPI == SCREAM_EPI_BIAS_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[i][c] = fmaxf(v[i][c] + p0[c], 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(C + (row0 + i) * ldc + col) = v[i];
    }
}

template <int EPI>
int launch(const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int64_t M, int N, int K,
           const EpiArgs& ep, hipStream_t st) {
    const int n_tiles = N / BN;
    const int64_t blocks = (M / BM) * n_tiles;
    if (blocks == 0) return 0;
    SCREAM_REQUIRE(blocks < (1ll << 31), SCREAM_EUNSUPPORTED);
    gemm_f32_kernel<EPI><<<dim3((unsigned)blocks), dim3(THREADS), 0, st>>>(A, lda, W, C, ldc, n_tiles, K, ep);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int scream_gemm_f32(const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int64_t M,
                               int32_t N, int32_t K, int32_t epilogue, int32_t n_act, const float* bias,
                               const float* residual, int64_t ldr, const float* gamma, const float* beta,
                               void* stream) {
    SCREAM_REQUIRE(A && W && C, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % BM == 0 && N > 0 && N % BN == 0 && K > 0 && K % BK == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(C) & 15) == 0, SCREAM_EINVAL);
    EpiArgs ep{n_act, bias, residual, ldr, gamma, beta};
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
