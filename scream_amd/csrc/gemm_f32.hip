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
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 256;
constexpr int THREADS = 256;
constexpr int MAX_GRID = 512;   // 256 CUs x 2 resident blocks

struct EpiArgs {
    int n_act;
    const float* bias;
    const float* residual;
    int64_t ldr;
    const float* gamma;
    const float* beta;
    // SCREAM_EPI_QKV only
    float* kv_partial;          // [M/128][8][33*32]
    const int32_t* tile_cloud;  // cloud of each 128-row tile of the packed batch
    const int32_t* cloud_row0;
    const int32_t* cloud_len;
    int64_t row_base;           // packed row of A's row 0
};

constexpr int KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;

__device__ __forceinline__ void lds_barrier() {
    // workgroup barrier that waits for this wave's LDS traffic only: a __syncthreads() would also drain the
    // LDS-DMA prefetch of the next tile that is deliberately left in flight (vmcnt).
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

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
        // ---- fused K^T V epilogue (SCREAM_EPI_QKV, key/value tiles) ------------------------------------------------
        // A key/value tile holds, for four heads, K (columns 0-127) and V (columns 128-255) of the same 128 tokens.
        // In the 32x32 accumulator layout lane = column and the registers walk the rows, which is exactly the A / B
        // operand layout of v_mfma_f32_32x32x2_f32 with the TOKEN as the contraction index: KV_h += mfma(K'_h[e], V_h[e])
        // over the 16 accumulator registers is sum_tokens K'[t,d] V[t,v] -- straight from registers, K' and V never
        // reach HBM (models/transformer.py:38-41: the "nshd,nshv->nhdv" einsum and K.sum; the division by v_length, which
        // the reference applies to V "to prevent fp16 overflow", is linear and is applied to the fp32 sum in scream_kv_finalize).
        if (EPI == SCREAM_EPI_QKV && n0_cur >= ep.n_act) {
            const int mt_local = (int)(m0_cur / BM);
            const int cloud = ep.tile_cloud[(ep.row_base + m0_cur) / BM];
            const int clen = ep.cloud_len[cloud];
            const int valid_w = clen - (int)(ep.row_base + m0_cur - ep.cloud_row0[cloud]) - wave * 32;  // real tokens in this wave's rows
            const int hb = (n0_cur - ep.n_act) / BN * 4;
            float* part = ep.kv_partial + ((int64_t)mt_local * SCREAM_NHEAD + hb) * KV_ELEMS;
            float* slabs = smem + WTILE;  // 2 heads x 4 waves x 1056 floats: the free W buffer plus the 1 KiB tail
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {  // two heads per round: 4 workgroup barriers per tile instead of 8
#pragma unroll
                for (int hh2 = 0; hh2 < 2; ++hh2) {
                    const int hq = hp * 2 + hh2;
                    f32x16 kv;
#pragma unroll
                    for (int e = 0; e < 16; ++e) kv[e] = 0.f;
                    float ks = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float a = acc[hq][e];
                        a = a > 0.f ? a + 1.0f : expf(a);                  // elu(k) + 1
                        if (mfma32_row(e, half) >= valid_w) a = 0.f;       // padding rows do not exist
                        kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[4 + hq][e], kv, 0, 0, 0);  // (1 / v_length is applied once, in scream_kv_finalize)
                        ks += a;
                    }
                    ks += __shfl_xor(ks, 32);
                    float* sw = slabs + (hh2 * 4 + wave) * KV_ELEMS;
#pragma unroll
                    for (int e = 0; e < 16; ++e) sw[mfma32_row(e, half) * 32 + r] = kv[e];  // [d][v]
                    if (half == 0) sw[32 * 32 + r] = ks;
                }
                lds_barrier();
                for (int i = tid; i < 2 * KV_ELEMS; i += THREADS) {
                    const int hh2 = i >= KV_ELEMS ? 1 : 0, k = i - hh2 * KV_ELEMS;
                    const float* s4 = slabs + hh2 * 4 * KV_ELEMS + k;
                    part[(hp * 2 + hh2) * KV_ELEMS + k] = (s4[0] + s4[KV_ELEMS]) + (s4[2 * KV_ELEMS] + s4[3 * KV_ELEMS]);
                }
                lds_barrier();
            }
        } else {
        // ---- epilogue (wave-private) ---------------------------------------------------------------------------
        // acc[tn][e]: row = wave*32 + mfma32_row(e, half), col = tn*32 + r.  The k-loop ended on a barrier and its
        // last k-tile used buffer 0's partner, so buffer 1 is free: each wave takes an 8-row slab of it.
        constexpr int SLAB_LD = 256;
        float* slab = smem + WTILE + wave * (8 * SLAB_LD);
        const int col = n0_cur + lane * 4;
        f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};  // bias | gamma, beta
        if (EPI == SCREAM_EPI_BIAS_RELU) p0 = ld4(ep.bias + col);
        if (EPI == SCREAM_EPI_RES_LN) {
            p0 = ld4(ep.gamma + col);
            p1 = ld4(ep.beta + col);
        }
        const bool act = n0_cur < ep.n_act;  // n_act is a multiple of 256: uniform per tile
        // RES_LN: the residual rows of half-chunk h+1 are requested before half-chunk h is processed, so their
        // ~2 us first-touch latency is not paid eight times in a row
        f32x4 rsd[2][4];
        if (EPI == SCREAM_EPI_RES_LN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rsd[0][i] = ld4(ep.residual + (m0_cur + wave * 32 + i) * ep.ldr + col);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8g .. 8g+7 of the wave's 32
#pragma unroll
            for (int tn = 0; tn < 8; ++tn)
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[(i + 4 * half) * SLAB_LD + tn * 32 + r] = acc[tn][4 * g + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {  // two halves of 4 rows: keeps the live set inside 256 VGPRs
                f32x4 vv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[i] = ld4(slab + (hh * 4 + i) * SLAB_LD + lane * 4);
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
                            for (int c = 0; c < 4; ++c) vv[i][c] = vv[i][c] > 0.f ? vv[i][c] + 1.0f : expf(vv[i][c]);  // elu(x)+1 == exp(x), x <= 0
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
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(C + (row0 + i) * ldc + col) = vv[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        }  // standard epilogue
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
