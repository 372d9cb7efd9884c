// fp32 GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  on the gfx950 fp32-input matrix cores.
//
// Replaces the nn.Linear / Conv1d(k=1) calls of the reference hot path
// (models/transformer.py:79-81,83,87; models/pointnet.py:60).  fp32 in, fp32 accumulate:
// v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain, so results are "exact fp32" in the
// same sense as the reference's sgemm (only the summation order differs).
//
// Geometry (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD for f32 MFMA, 160 KiB LDS/CU):
//   block tile 128 x 256, BK = 32, 256 threads = 4 waves as 2(M) x 2(N); wave tile 64 x 128
//   = 2 x 4 MFMA tiles of 32x32 -> 128 accumulator registers; 2 blocks per CU (54 KiB LDS each)
//   so the second block's MFMAs cover the first block's staging/barriers.
//   Both operands are K-contiguous.  A k-chunk of 8 is split so that lane (r, half) holds
//   k = 4*half .. 4*half+3 of row r for BOTH operands: the contraction index is a dummy, so any
//   k permutation shared by A and W is legal, and it lets every fragment be one ds_read_b128.
//   LDS rows are padded to 36 floats: the four 16-lane groups of a ds_read_b128 then hit 16
//   distinct 4-bank slots (36 r mod 64 is a bijection on r mod 16) -> conflict free.
//   Staging is global -> registers (issued before the MFMAs of the current tile) -> LDS
//   (written after them); fp32 MFMA is slow enough (64 cycles each) that this is fully hidden.
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
    const int wm = wave & 1, wn = wave >> 1;
    const int r = lane & 31, half = lane >> 5;
    const int64_t mt = blockIdx.x / n_tiles;
    const int nt = blockIdx.x % n_tiles;
    const int64_t m0 = mt * BM;
    const int n0 = nt * BN;

    // staging map: float4 index f = tid + 256 i -> row f >> 3, 16-byte column f & 7
    const int srow = tid >> 3, sc4 = tid & 7;
    const float* ga = A + (m0 + srow) * lda + sc4 * 4;
    const float* gw = W + (int64_t)(n0 + srow) * K + sc4 * 4;
    const int lds_st = srow * LDS_LD + sc4 * 4;

    f32x4 ra[4], rb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ga + (int64_t)(32 * i) * lda);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const f32x4*>(gw + (int64_t)(32 * i) * K);

    f32x16 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + lds_st + 32 * i * LDS_LD) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Bs + lds_st + 32 * i * LDS_LD) = rb[i];
    __syncthreads();

    const float* a_frag = As + (wm * 64 + r) * LDS_LD + half * 4;
    const float* b_frag = Bs + (wn * 128 + r) * LDS_LD + half * 4;
    const int KT = K / BK;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) {
            ga += BK;
            gw += BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ga + (int64_t)(32 * i) * lda);
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const f32x4*>(gw + (int64_t)(32 * i) * K);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[2], bf[4];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) af[tm] = *reinterpret_cast<const f32x4*>(a_frag + tm * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j], acc[tm][tn], 0, 0, 0);
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

    // ------------------------------------------------------------------ epilogue
    // acc[tm][tn][e]: row = wm*64 + tm*32 + mfma32_row(e, half), col = wn*128 + tn*32 + r
    const int row_w = wm * 64;
    const int col_w = n0 + wn * 128 + r;

    if (EPI == SCREAM_EPI_RES_LN) {
        float* red1 = smem;        // [2][128] row sums per N-half (smem is free: loop ended on a barrier)
        float* red2 = smem + 256;  // [2][128] centred sums of squares
        float g[4], b[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            g[tn] = ep.gamma[col_w + tn * 32];
            b[tn] = ep.beta[col_w + tn * 32];
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = row_w + tm * 32 + mfma32_row(e, half);
                const float* rp = ep.residual + (m0 + rl) * ep.ldr + col_w;
                float s = 0.f;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    acc[tm][tn][e] += rp[tn * 32];
                    s += acc[tm][tn][e];
                }
                s = half_wave_sum(s);
                if (r == 0) red1[wn * 128 + rl] = s;
            }
        __syncthreads();
        float mean[2][16];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = row_w + tm * 32 + mfma32_row(e, half);
                mean[tm][e] = (red1[rl] + red1[128 + rl]) * (1.0f / 256.0f);
                float s = 0.f;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float d = acc[tm][tn][e] - mean[tm][e];
                    s += d * d;
                }
                s = half_wave_sum(s);
                if (r == 0) red2[wn * 128 + rl] = s;
            }
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = row_w + tm * 32 + mfma32_row(e, half);
                const float var = (red2[rl] + red2[128 + rl]) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                float* cp = C + (m0 + rl) * ldc + col_w;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    cp[tn * 32] = (acc[tm][tn][e] - mean[tm][e]) * rstd * g[tn] + b[tn];
            }
        return;
    }

    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == SCREAM_EPI_BIAS_RELU) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) bias[tn] = ep.bias[col_w + tn * 32];
    }
    const bool act = n0 < ep.n_act;  // n_act is a multiple of 256: uniform per block
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = row_w + tm * 32 + mfma32_row(e, half);
            float* cp = C + (m0 + rl) * ldc + col_w;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                float v = acc[tm][tn][e];
                if (EPI == SCREAM_EPI_ELU1) {
                    if (act) v = v > 0.f ? v + 1.0f : expm1f(v) + 1.0f;
                } else if (EPI == SCREAM_EPI_RELU) {
                    v = fmaxf(v, 0.f);
                } else if (EPI == SCREAM_EPI_BIAS_RELU) {
                    v = fmaxf(v + bias[tn], 0.f);
                }
                cp[tn * 32] = v;
            }
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
    SCREAM_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0, SCREAM_EINVAL);
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
            SCREAM_REQUIRE(residual && gamma && beta && ldr >= N, SCREAM_EINVAL);
            return launch<SCREAM_EPI_RES_LN>(A, lda, W, C, ldc, M, N, K, ep, st);
        default:
            return SCREAM_EINVAL;
    }
}
