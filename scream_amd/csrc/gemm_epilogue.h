// Epilogues shared by the fp32-MFMA GEMM (gemm_f32.hip, 4 waves, 128-row tiles) and the 3xbf16-split GEMM
// (gemm_x3.hip, 8 waves, 256-row tiles).  Both leave a 32 x 256 fp32 accumulator tile per wave in the
// 32x32 MFMA layout: acc[tn][e] = C[row = mfma32_row(e, half)][col = tn*32 + r].
#pragma once
#include "common.h"
#include "split.h"

namespace {

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
    // SCREAM_EPI_ELU1 / SCREAM_EPI_QKV, x3 kernel only: the activated tile (the 256 query columns) is written in the
    // FRAGMENT-major activation layout (SCREAM_ACT_FRAG, include/scream_hip.h) that tail_x3.hip reads; ldc must be 256
    int c_frag;
    // SCREAM_EPI_QKV with more than one key/value tile PAIR (the cross stage's six target-side projections as one GEMM,
    // N = n_act + 512 L): floats between the partial arrays of consecutive layers
    int64_t kv_layer_stride;
    // KV_H2 (the fp16 splits of gemm_split.hip): 2^k_exp, 2^v_exp and 2^-(k_exp + v_exp) -- the K^T V reduction runs on fp16 x 2
    // planes of K' = elu(k) + 1 and V (|K'| 2^k_exp, |V| 2^v_exp <= 2^15: bounds of the projections, scream_amd/scales.py)
    float kv_sk, kv_sv, kv_inv;
};

constexpr int KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;

__device__ __forceinline__ void lds_barrier() {
    // workgroup barrier that waits for this wave's LDS traffic only: a __syncthreads() would also drain the
    // LDS-DMA prefetch of the next tile that is deliberately left in flight (vmcnt).
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// NWAVES waves stacked in M (wave w owns rows w*32 .. w*32+31 of the block tile at packed-GEMM row m0);
// slabs: LDS scratch of max(NWAVES * 8 * 256, 8 * KV_ELEMS) floats that no other wave-group touches meanwhile.
template <int EPI, int NWAVES, bool KV_H2 = false>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[8], float* slabs, int wave, int lane, int tid,
                                              bool rows_exist, int64_t m0_cur, int n0_cur, const EpiArgs& ep,
                                              float* __restrict__ C, int64_t ldc) {
    constexpr int BN = 256;
    const int r = lane & 31, half = lane >> 5;
    // ---- fused K^T V epilogue (SCREAM_EPI_QKV, key/value tiles) ------------------------------------------------
    // A key/value tile holds, for four heads, K (columns 0-127) and V (columns 128-255) of the same 128 tokens.
    // In the 32x32 accumulator layout lane = column and the registers walk the rows, which is exactly the A / B
    // operand layout of v_mfma_f32_32x32x2_f32 with the TOKEN as the contraction index: KV_h += mfma(K'_h[e], V_h[e])
    // over the 16 accumulator registers is sum_tokens K'[t,d] V[t,v] -- straight from registers, K' and V never
    // reach HBM (models/transformer.py:38-41: the "nshd,nshv->nhdv" einsum and K.sum; the division by v_length, which
    // the reference applies to V "to prevent fp16 overflow", is linear and is applied to the fp32 sum in scream_kv_finalize).
    if (EPI == SCREAM_EPI_QKV && n0_cur >= ep.n_act) {
        // groups of four waves = one 128-row tile = one cloud (clouds start on 128-row boundaries)
        const int grp = wave >> 2, wg = wave & 3;
        const int64_t mrow = m0_cur + grp * SCREAM_ROW_TILE;
        int valid_w = 0;
        float* part = ep.kv_partial;
        if (rows_exist) {
            const int cloud = ep.tile_cloud[(ep.row_base + mrow) / SCREAM_ROW_TILE];
            valid_w = ep.cloud_len[cloud] - (int)(ep.row_base + mrow - ep.cloud_row0[cloud]) - wg * 32;  // real tokens in this wave's rows
            const int kvt = (n0_cur - ep.n_act) / BN;  // key/value tile: layer kvt / 2, heads 4 (kvt % 2) .. + 3
            part += (int64_t)(kvt >> 1) * ep.kv_layer_stride + ((int64_t)(mrow / SCREAM_ROW_TILE) * SCREAM_NHEAD + (kvt & 1) * 4) * KV_ELEMS;
        }
        constexpr int HPR = NWAVES == 4 ? 2 : 1;  // heads per LDS round: 8 x 1056 floats of scratch either way
#pragma unroll
        for (int hp = 0; hp < 4 / HPR; ++hp) {
#pragma unroll
            for (int hh2 = 0; hh2 < HPR; ++hh2) {
                const int hq = hp * HPR + hh2;
                f32x16 kv;
#pragma unroll
                for (int e = 0; e < 16; ++e) kv[e] = 0.f;
                float ks = 0.f;
                if (KV_H2) {
                    // Round 3: the same sum on the fp16 matrix cores.  Registers 8 s .. 8 s + 7 of the two tiles are the 16-token
                    // step s of the contraction (both operands walk the tokens the same way), K' and V go in as two fp16 planes
                    // each with their exponents, three products per step: 6 MFMAs of 32 cycles per head instead of 16 of 64
                    // (a third on top of a key/value tile's main product once that had halved).
                    f16x8 kp[2][2], vp[2][2];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float a = acc[hq][e];
                        a = elu1(a);                                       // elu(k) + 1
                        if (mfma32_row(e, half) >= valid_w) a = 0.f;       // padding rows do not exist
                        ks += a;
                        SplitH2::split1(a * ep.kv_sk, e & 7, kp[e >> 3]);
                        SplitH2::split1(acc[4 + hq][e] * ep.kv_sv, e & 7, vp[e >> 3]);
                    }
                    SplitH2::products(kv, kp[0], vp[0], kv);
                    SplitH2::products(kv, kp[1], vp[1], kv);
                    kv *= ep.kv_inv;  // exact: a power of two  (1 / v_length is applied once, in scream_kv_finalize)
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float a = acc[hq][e];
                        a = elu1(a);                                       // elu(k) + 1
                        if (mfma32_row(e, half) >= valid_w) a = 0.f;       // padding rows do not exist
                        kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[4 + hq][e], kv, 0, 0, 0);  // (1 / v_length is applied once, in scream_kv_finalize)
                        ks += a;
                    }
                }
                ks += __shfl_xor(ks, 32);
                float* sw = slabs + ((grp * HPR + hh2) * 4 + wg) * KV_ELEMS;
#pragma unroll
                for (int e = 0; e < 16; ++e) sw[mfma32_row(e, half) * 32 + r] = kv[e];  // [d][v]
                if (half == 0) sw[32 * 32 + r] = ks;
            }
            lds_barrier();
            if (rows_exist) {
                const int t4 = tid & 255;  // thread index inside the four-wave group
                for (int i = t4; i < HPR * KV_ELEMS; i += 256) {
                    const int hh2 = i >= KV_ELEMS ? 1 : 0, k = i - hh2 * KV_ELEMS;
                    const float* s4 = slabs + (grp * HPR + hh2) * 4 * KV_ELEMS + k;
                    part[(hp * HPR + hh2) * KV_ELEMS + k] = (s4[0] + s4[KV_ELEMS]) + (s4[2 * KV_ELEMS] + s4[3 * KV_ELEMS]);
                }
            }
            lds_barrier();
        }
    } else if ((EPI == SCREAM_EPI_ELU1 || EPI == SCREAM_EPI_QKV) && NWAVES == 8 && ep.c_frag && n0_cur < ep.n_act) {
    // ---- Q' = elu(q) + 1 in the fragment-major layout -------------------------------------------------------------
    // Element (row 32 t + rho, feature 32 blk + 8 a + 4 hf + b) lives at ((((t * 8 + blk) * 4 + a) * 64 + rho + 32 hf) * 4 + b:
    // per 32-row group, segment and a, one 1 KiB run in lane order -- what a wave of tail_x3.hip loads with one instruction.
    // Eight rows at a time go through the wave's slab (row-major, the 16-byte chunks of row i XOR-swizzled by i so that the
    // transposing read below is 2-way conflicted at worst); a store instruction then writes eight full 128-byte lines.
    if (rows_exist) {
        float* slab = slabs + wave * (8 * 256);
        float* cg = C + (m0_cur + wave * 32) * 256;  // this wave's 32-row group (8192 floats in either layout)
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8g .. 8g+7 of the wave's 32
#pragma unroll
            for (int tn = 0; tn < 8; ++tn)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int i8 = i + 4 * half;
                    slab[i8 * 256 + (((tn * 8 + (r >> 2)) ^ i8) << 2) + (r & 3)] = acc[tn][4 * g + i];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int a = lane >> 4, hf = (lane >> 3) & 1, rr = lane & 7;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                f32x4 v = ld4(slab + rr * 256 + (((8 * blk + 2 * a + hf) ^ rr) << 2));
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = elu1(v[c]);
                *reinterpret_cast<f32x4*>(cg + ((blk * 4 + a) * 64 + 8 * g + rr + 32 * hf) * 4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    } else if (rows_exist) {
    // ---- epilogue (wave-private) ---------------------------------------------------------------------------
    // acc[tn][e]: row = wave*32 + mfma32_row(e, half), col = tn*32 + r.  The k-loop ended on a barrier and its
    // last k-tile used buffer 0's partner, so buffer 1 is free: each wave takes an 8-row slab of it.
    constexpr int SLAB_LD = 256;
    float* slab = slabs + wave * (8 * SLAB_LD);
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
                        for (int c = 0; c < 4; ++c) vv[i][c] = elu1(vv[i][c]);
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
#ifdef X3_ABLATE
            if (X3_ABLATE & 64) {  // tuning aid: keep the arithmetic, drop the stores
                if (vv[0][0] + vv[1][1] + vv[2][2] + vv[3][3] != 123.456f) continue;
            }
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(C + (row0 + i) * ldc + col) = vv[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }  // standard epilogue
}

}  // namespace
