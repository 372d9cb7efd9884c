/*
 * scream_hip.h -- C ABI of libscream_hip.so: the MI355X (gfx950) kernels behind the
 * SCREAM registration hot path.
 *
 * The reference (xujiabo/SCREAM) has no FFI/operator layer: its hot path is stock ATen
 * calls made from Python.  Each entry point below therefore cites the reference
 * *call site* (file:line under the reference checkout) whose arithmetic it replaces;
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add at that site.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     unless the name ends in _host; kernels never allocate and keep no global state;
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - returns 0 on success, a negative SCREAM_E* code on a rejected argument (checked on
 *     the host before any launch), or the positive hipError_t of a failed launch;
 *   - every input, output and accumulation is IEEE fp32 (fp64 only inside the 3x3 Kabsch solve; the split GEMMs hold their
 *     operands as exact sums of 16-bit planes inside the kernels, see scream_gemm_split_f32);
 *   - "rows" are points/tokens.  A *packed batch* holds several clouds back to back, each
 *     cloud starting on a 128-row boundary (SCREAM_ROW_TILE) and zero-padded to one, so a
 *     128-row tile never spans two clouds.
 */
#ifndef SCREAM_HIP_H
#define SCREAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCREAM_ROW_TILE 128
#define SCREAM_D_MODEL 256
#define SCREAM_NHEAD 8
#define SCREAM_HEAD_DIM 32
#define SCREAM_KV_CHUNK 256 /* tokens per partial K^T V reduction */

#define SCREAM_EINVAL (-1) /* bad size / alignment / NULL pointer */
#define SCREAM_EUNSUPPORTED (-2) /* shape outside what the gfx950 kernels are built for */

/* GEMM epilogues (models/transformer.py:79-88, models/pointnet.py:27-33) */
enum {
    SCREAM_EPI_NONE = 0,      /* C = A W^T */
    SCREAM_EPI_ELU1 = 1,      /* columns < n_act: elu(x)+1 (transformer.py:7-8,28-29); rest plain */
    SCREAM_EPI_RELU = 2,      /* relu (transformer.py:66) */
    SCREAM_EPI_BIAS_RELU = 3, /* + bias, relu (pointnet.py:28-31) */
    SCREAM_EPI_RES_LN = 4,    /* LayerNorm(C + residual) * gamma + beta, N == 256 (transformer.py:84,88) */
    SCREAM_EPI_QKV = 5        /* internal to scream_gemm_qkv_f32 */
};

/* ---- Activation layouts of a [M, 256] fp32 matrix (M % 32 == 0).
 * Row-major is the default everywhere.  FRAGMENT-major (SCREAM_ACT_FRAG) is the layout the operand-split kernels pass
 * between each other inside scream_forward: element (row 32 t + r, feature 32 blk + 8 a + 4 h + b) -- t the 32-row group,
 * blk the 32-feature segment, a in 0..3, h in 0..1, b in 0..3 -- lives at float offset
 *     ((((t * 8 + blk) * 4 + a) * 64 + r + 32 h) * 4 + b.
 * A 32-row group occupies the same 32 KiB as in row-major, so slices of whole groups are the same pointers in both
 * layouts.  Why: lane r + 32 h of a wave owns exactly the 16-byte pieces 2a + h of every 128-byte segment of row r as an
 * MFMA operand; stored this way, each of its four loads per segment is one fully contiguous 1 KiB wave access instead of
 * 32 half-used cache lines (scream_amd/csrc/tail_split.hip).  scream_act_layout converts a matrix between the two. */
#define SCREAM_LAYOUT_A_FRAG 1 /* the GEMM's A operand is fragment-major */
#define SCREAM_LAYOUT_C_FRAG 2 /* the activated (elu + 1) query tile of the GEMM's output is fragment-major */
int scream_act_layout(const float* src, float* dst, int64_t M, int32_t to_fragment, void* stream);

/* Library / build identification: "scream_hip gfx950 <abi>"; abi bumps on any signature change. */
const char* scream_version(void);
int scream_abi_version(void);

/* ---- A2/A4/A6: C[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32-input MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces torch.nn.Linear / Conv1d(k=1) at models/transformer.py:79-81,83,87 and
 * models/pointnet.py:60.  M % 128 == 0, N % 256 == 0, K % 64 == 0; lda/ldc/ldr in floats, multiples of 4;
 * A, W, C and residual 16-byte aligned. */
int scream_gemm_f32(const float* A, int64_t lda, const float* W, float* C, int64_t ldc,
                    int64_t M, int32_t N, int32_t K, int32_t epilogue, int32_t n_act,
                    const float* bias, const float* residual, int64_t ldr,
                    const float* gamma, const float* beta, void* stream);

/* ---- A2 + A3 (reduce) fused: q/k/v projections whose key/value tiles never leave the chip.
 * Replaces models/transformer.py:79-81 together with :28-29 (elu+1) and :38-40 (values / v_length, K^T V, K.sum).
 * W [N,K] is laid out [ q (n_q = 256 rows, or 0 rows for the cross layers' key/value-only call) |
 * k heads 0-3 | v heads 0-3 | k heads 4-7 | v heads 4-7 ] (128 rows each), so one 256-wide tile holds K and V of
 * four heads for the same 128 tokens.  Query tiles are stored as elu(q)+1 into Q [M, ldq]; key/value tiles are
 * reduced in the epilogue straight from the MFMA accumulators into kv_partial [M/128][8][33*32] (per 128-row
 * tile and head: K'^T (V/S) as [d][v], then Ksum[d]); rows past the cloud's length are masked.  A's row 0 is
 * packed row `row_base`; tile_cloud / cloud_row0 / cloud_len describe the packed batch (absolute rows).
 * scream_kv_finalize then adds the tiles of every cloud in order. */
int scream_gemm_qkv_f32(const float* A, int64_t lda, const float* W, float* Q, int64_t ldq, int64_t M,
                        int32_t N, int32_t K, int32_t n_q, const int32_t* tile_cloud,
                        const int32_t* cloud_row0, const int32_t* cloud_len, int64_t row_base,
                        float* kv_partial, void* stream);
int scream_kv_finalize(const float* kv_partial, const int32_t* cloud_row0, const int32_t* cloud_len,
                       int64_t row_base, int32_t cloud_begin, int32_t n_kv, float* kv_out, void* stream);

/* ---- The same two GEMM entry points on the 16-bit matrix cores by OPERAND SPLITTING, fp32-level accuracy
 * (scream_amd/csrc/gemm_split.hip, split.h).  `split` selects the arithmetic:
 *   SCREAM_SPLIT_H2  (2): x 2^e = x0 + x1 in fp16, three fp16 MFMAs per 16-deep step (a1b0 + a0b1 + a0b0), fp32 accumulate --
 *                    the default of the forward since round 3.  fp16 has a 5-bit exponent, so each operand carries an exact
 *                    power-of-two scale: w_exp is baked into the packed weights, a_exp is applied to A as it is split, the
 *                    accumulators are multiplied by 2^-(a_exp + w_exp).  CONTRACT: |A| 2^a_exp <= 2^15 and |W| 2^w_exp <= 2^15 for
 *                    every element (fp16 overflows at 65 504; values below 2^-3 / 2^exp lose relative precision, so choose the
 *                    largest exponent the bound allows).  scream_amd/scales.py derives a_exp for every GEMM of the forward from
 *                    the weights alone (LayerNorm outputs are bounded by gamma / beta).
 *   SCREAM_SPLIT_BF3 (3): x = x0 + x1 + x2 in bf16 (exact), six bf16 MFMAs per step; bf16 keeps fp32's exponent range, so
 *                    a_exp / w_exp are ignored and the result is scale invariant bit for bit.  Rounds 1-2's arithmetic.
 * The weight matrix is pre-split and re-tiled ONCE by scream_pack_w_split (device to device): W [N,K] fp32 ->
 * W_packed, 2*split*N*K bytes: [split planes][K/32 k-tiles][N][32] 16-bit values, stored k-tile by k-tile in the order the
 * kernel stages it in LDS.  A, C, residual and bias stay fp32.  M % 128 == 0, N % 256 == 0, K = 64 + 192 j (64, 256, 448, ...,
 * 1024: the k-loop rotates three operand register sets over two LDS stages, so the k-tile count K/32 is even and 2 mod 3;
 * anything else is SCREAM_EUNSUPPORTED).  A row-block [n0, n0 + n) of W must be packed on its own to be used as a GEMM
 * operand on its own.  scream_gemm_qkv_split_f32 also takes N = 512 L with n_q == 0: the key/value projections of L layers
 * applied to the SAME rows as one GEMM (the cross stage's target side: the target features are frozen after the stem,
 * models/pointnet.py:53-57); W is then the L matrices [k heads 0-3 | v 0-3 | k 4-7 | v 4-7] stacked, and layer l's partials
 * are written at kv_partial + l * (M/128) * 8 * 1056 floats.  k_exp / v_exp (fp16 splits; ignored on SCREAM_SPLIT_BF3): the K^T V
 * reduction in the epilogue runs on fp16 x 2 planes of K' = elu(k) + 1 and V as well -- CONTRACT |K'| 2^k_exp, |V| 2^v_exp <= 2^15
 * (|k| and |v| are bounded like any projection of a LayerNorm output; K' <= 1 + max(k, 0)).  layout (SCREAM_LAYOUT_* bits): A fragment-major needs lda == K == 256; C fragment-major applies to the
 * activated tile of an ELU1 / QKV epilogue with n_act == ldc == 256 (the queries). */
#define SCREAM_SPLIT_H1 1 /* ONE fp16 plane, one product: NOT fp32-accurate (2^-11 per operand) -- the mirror of the reference's
                            * `with autocast()` around the KITTI forward (evaluate_kitti.py:37); exponents as for SCREAM_SPLIT_H2;
                            * explicit opt-in only (evaluate_kitti.evaluate(autocast=True)) */
#define SCREAM_SPLIT_H2 2
#define SCREAM_SPLIT_BF3 3
int scream_pack_w_split(const float* W, int32_t N, int32_t K, int32_t split, int32_t w_exp, void* W_packed,
                        void* stream);
int scream_gemm_split_f32(const float* A, int64_t lda, const void* W_packed, float* C, int64_t ldc, int64_t M,
                          int32_t N, int32_t K, int32_t epilogue, int32_t n_act, const float* bias,
                          const float* residual, int64_t ldr, const float* gamma, const float* beta,
                          int32_t layout, int32_t split, int32_t a_exp, int32_t w_exp, void* stream);
int scream_gemm_qkv_split_f32(const float* A, int64_t lda, const void* W_packed, float* Q, int64_t ldq,
                              int64_t M, int32_t N, int32_t K, int32_t n_q, const int32_t* tile_cloud,
                              const int32_t* cloud_row0, const int32_t* cloud_len, int64_t row_base,
                              float* kv_partial, int32_t layout, int32_t split, int32_t a_exp, int32_t w_exp,
                              int32_t k_exp, int32_t v_exp, void* stream);

/* ---- The same fused q/k/v projection (A2 + the A3 reduce) as a RING kernel (scream_amd/csrc/proj_ring.hip, round 4; fp16 splits,
 * fragment-major x in and Q' out): one wave per SIMD owns 64 rows and keeps their operand planes in registers for all of
 * q | k | v, the weights stream chunk by chunk (32 output columns over K = 256 = one 32 KiB stage) through a ring of LDS stages,
 * and every chunk's epilogue -- elu + 1 and the stores of a query chunk, or K' = elu(k) + 1, the operand split and K'^T V of a
 * head -- runs in the shadow of the NEXT chunk's matrix instructions.  Same arithmetic per product and the same outputs as
 * scream_gemm_qkv_split_f32 with SCREAM_LAYOUT_A_FRAG | SCREAM_LAYOUT_C_FRAG (Q' and the kv_partial array scream_kv_finalize_image
 * sums; the partial of a 128-row tile is added up from two 64-row halves instead of four 32-row quarters, so the two kernels
 * agree to fp32 rounding, not bit for bit).  The weight matrix W [N, 256] (row order of scream_gemm_qkv_f32 above: q | per
 * layer k 0-3 | v 0-3 | k 4-7 | v 4-7; n_q = 256 with N = 768, or n_q = 0 with N = 512 L) is packed once by scream_pack_proj
 * into scream_proj_image_bytes(N, split) bytes: [N / 32 stages][split planes][16 fragments][64 lanes][8] fp16, in the order the
 * kernel consumes them (query chunks 0 .. 7, then per layer and head K_h, V_h).  M % 128 == 0; exponents as for
 * scream_gemm_qkv_split_f32 (a_exp, w_exp of the product; k_exp, v_exp of K' and V in the reduction). */
int64_t scream_proj_image_bytes(int32_t N, int32_t split);
int scream_pack_proj(const float* W, int32_t N, int32_t n_q, int32_t split, int32_t w_exp, void* proj_image, void* stream);
int scream_proj_qkv_f32(const float* x, const void* proj_image, float* Q, int64_t M, int32_t N, int32_t n_q,
                        const int32_t* tile_cloud, const int32_t* cloud_row0, const int32_t* cloud_len,
                        int64_t row_base, float* kv_partial, int32_t split, int32_t a_exp, int32_t w_exp, int32_t k_exp,
                        int32_t v_exp, void* stream);

/* ---- A3 (apply) + A4 as ONE launch: everything of an MHAttention block that is local to a row,
 *     att = ((Q' . KV) * Z) * S;  m1 = LayerNorm1(att . Wm^T + x);  y = LayerNorm2(x + W2 . relu(W1 . m1))
 * Replaces models/transformer.py:41-42,83-88, i.e. scream_attn_apply + scream_gemm_split_f32(merge, EPI_RES_LN) +
 * scream_gemm_split_f32(mlp.0, EPI_RELU) + scream_gemm_split_f32(mlp.2, EPI_RES_LN), with the same split arithmetic; att, m1
 * and the 1024-wide hidden activations never reach memory (scream_amd/csrc/tail_split.hip).  Operands:
 * Q, x and y are FRAGMENT-major [M, 256] matrices (SCREAM_ACT_FRAG above; scream_act_layout converts):
 *   Q             the elu+1 mapped queries written by scream_gemm_qkv_split_f32 / scream_gemm_split_f32(EPI_ELU1) with
 *                 SCREAM_LAYOUT_C_FRAG;
 *   kv_image      scream_kv_image_bytes() per cloud, written by scream_kv_finalize_image from the K^T V partials of
 *                 scream_gemm_qkv_split_f32 (same arguments as scream_kv_finalize): KV^T / S as MFMA operand fragments + Ksum in fp32, for the SAME `split` as the tail that reads it: three bf16 planes
 *                 (SCREAM_SPLIT_BF3) or -- fp16 splits, round 4 -- two fp16 planes of KV_h^T / S * 2^e_h, e_h chosen on the device from the
 *                 head's largest |element| (a maximum: exact, order independent) with 2^-e_h stored beside them;  the key cloud of 128-row
 *                 tile t is tile_cloud[t] + kv_cloud_offset (tile_cloud points at the entry of the first row), cloud_len gives S;
 *   x             the block input (residual of BOTH norms), must not alias y;
 *   tail_image    scream_tail_image_bytes(split), built once by scream_pack_tail from merge [256,256], mlp.0 [1024,256]
 *                 and mlp.2 [256,1024] (fp32, device to device) for the same `split` and exponents;
 *   exps          SCREAM_SPLIT_H2 only (NULL otherwise): the power-of-two exponents of the operands.  CONTRACT, for every value
 *                 the operand can take: |att| 2^e_att, |m1| 2^e_m1, |relu(W1 m1)| 2^e_h <= 2^15 and |Wm| 2^e_wm, |W1| 2^e_w1,
 *                 |W2| 2^e_w2 <= 2^15.  |att| <= max |v| (a convex combination of value rows), |m1| <= 16 |gamma1| + |beta1|,
 *                 |W1 m1| <= 16 |W1_row * gamma1|_2 + |W1_row . beta1|: scream_amd/scales.py computes them from the weights.
 *                 Exponents in [-40, 40], e_wm + e_att and e_w2 + e_h in [-44, 44] (SCREAM_EINVAL otherwise).
 * M % 128 == 0, every pointer 16-byte aligned. */
typedef struct {
    int32_t e_att, e_wm, e_m1, e_w1, e_h, e_w2;
    int32_t e_y, e_wq; /* only with a next-layer query projection in the image (below): of the block output y and of that Wq */
    int32_t e_x;       /* q_first images only (scream_pack_tail): of the block INPUT x as the operand of this layer's own query projection (e_wq: of
                        * that Wq) */
    int32_t e_q;       /* of Q' = elu(q) + 1 as an operand of the attention apply (fp16 x 2 since round 4): |Q'| 2^e_q <= 2^15, Q' <= 1 + the
                        * bound of the query projection (scream_amd/scales.py: 16 |w * gamma|_2 + |w . beta| over the rows of Wq) */
} scream_tail_exps_t;
int64_t scream_tail_image_bytes(int32_t split, int32_t with_next_q);
int64_t scream_kv_image_bytes(void);
/* Wq_next (may be NULL; fp16 splits): q_proj.weight [256,256] of the NEXT layer when that layer takes its queries from this
 * block's output rows (a cross layer behind a self layer, models/transformer.py:130): eight more stages in the image, and
 * scream_layer_tail_f32 called with q_next != NULL ends every tile with Q'_next = elu(y . Wq_next^T) + 1 (fragment-major) --
 * the separate scream_gemm_split_f32(..., SCREAM_EPI_ELU1) launch of that layer is then not needed. */
/* EXPERIMENTAL (round 4; off in scream_amd/model.py -- results are not repeatable bit for bit beside other kernels, root cause open:
 * profiles/r04_qf_experiment.txt).  q_first != 0 (fp16 splits): Wq_next is THIS layer's q_proj.weight and its eight stages come FIRST in the image:
 * scream_layer_tail_f32 called with Q == NULL then begins every tile with Q' = elu(x . Wq^T) + 1 of its own rows, kept in registers
 * for the applies -- Q' is neither written by a projection launch nor read here (x is read anyway: the residual of both norms);
 * the projection of such a layer computes key/value chunks only (n_q = 0).  exps->e_x / e_wq: of x and of Wq as operands. */
int scream_pack_tail(const float* Wm, const float* W1, const float* W2, const float* Wq_next, int32_t q_first, int32_t split,
                     const scream_tail_exps_t* exps, void* tail_image, void* stream);
/* n_layers > 1: the partials of a batched key/value projection (scream_gemm_qkv_split_f32 with N = 512 n_layers): layer l
 * reads kv_partial + l * partial_layer_stride floats ((M/128) * 8 * 1056 of that GEMM) and writes its n_kv cloud images at
 * kv_image + l * image_layer_stride bytes.  n_layers == 1: strides ignored. */
int scream_kv_finalize_image(const float* kv_partial, const int32_t* cloud_row0, const int32_t* cloud_len,
                          int64_t row_base, int32_t cloud_begin, int32_t n_kv, void* kv_image, int32_t n_layers,
                          int64_t partial_layer_stride, int64_t image_layer_stride, int32_t split, void* stream);
int scream_layer_tail_f32(const float* Q, const void* kv_image, const int32_t* tile_cloud,
                          int32_t kv_cloud_offset, const int32_t* cloud_len, const float* x,
                          const void* tail_image, const float* g1, const float* b1, const float* g2,
                          const float* b2, float* y, float* q_next, int64_t M, int32_t split,
                          const scream_tail_exps_t* exps, void* stream);

/* ---- A1: feats = LayerNorm(PE_sine(xyz) + W_e (xyz - center[cloud]) + b_e)
 * Replaces models/pointnet.py:45-48 (+ models/transformer.py:157-179).  xyz [rows,3] packed;
 * tile_cloud[rows/128] gives the cloud of each 128-row tile; center [n_clouds,3] (zeros for
 * target clouds); dim_t [84] is the host-computed frequency table of transformer.py:168-170. */
int scream_pe_embed_ln(const float* xyz, const int32_t* tile_cloud, const float* center,
                       const float* dim_t, const float* emb_w, const float* emb_b,
                       const float* gamma, const float* beta, float* feats, int64_t rows,
                       void* stream);
/* the same values written FRAGMENT-major (SCREAM_ACT_FRAG below): what the fused forward feeds its first projection */
int scream_pe_embed_ln_frag(const float* xyz, const int32_t* tile_cloud, const float* center,
                            const float* dim_t, const float* emb_w, const float* emb_b,
                            const float* gamma, const float* beta, float* feats, int64_t rows,
                            void* stream);

/* ---- A3 (reduce): per key cloud and head, KV = sum_s K[s]^T (V[s]/S), Ksum = sum_s K[s]
 * Replaces models/transformer.py:38-41 (values / v_length, the "nshd,nshv->nhdv" einsum, K.sum).
 * Kf/Vf point at column 0 of the (elu+1)-mapped keys / raw values, row stride ld floats, row 0 =
 * packed row `row_base`.  cloud_row0/cloud_len [n_clouds] (int32, absolute packed rows / true lengths).
 * Processes clouds [cloud_begin, cloud_begin + n_kv).  partial is scratch of
 * n_kv * max_chunks * 8 * 33 * 32 floats; kv_out [n_clouds][8][33][32]: rows 0-31 of a head hold KV^T
 * ([v][d], d contiguous -- the layout scream_attn_apply stages into LDS), row 32 holds Ksum[d]. */
int scream_kv_reduce(const float* Kf, const float* Vf, int64_t ld, int64_t row_base,
                     const int32_t* cloud_row0, const int32_t* cloud_len, int32_t cloud_begin,
                     int32_t n_kv, int32_t max_chunks, float* partial, float* kv_out, void* stream);

/* ---- A3 (apply): out[l,h,:] = (Q[l,h,:] . KV[h]) * 1/(Q[l,h,:].Ksum[h] + 1e-6) * S
 * Replaces models/transformer.py:41-42.  Qf [rows, ldq] (elu+1 mapped), kv as written by
 * scream_kv_reduce; the key cloud of query tile t is tile_cloud[t] + kv_cloud_offset. */
int scream_attn_apply(const float* Qf, int64_t ldq, const float* kv, const int32_t* tile_cloud,
                      int32_t kv_cloud_offset, const int32_t* cloud_len, float* out, int64_t ldo,
                      int64_t rows, void* stream);

/* ---- A6 (head): out[rows,3] = X[rows,256] . W[3,256]^T + b.  models/pointnet.py:32,60 */
int scream_coor_head(const float* X, const float* W, const float* b, float* out, int64_t rows,
                     void* stream);

/* ---- Whole forward pass of PointTransformer over a packed batch (A1-A6).
 * Replaces models/pointnet.py:38-60 for B pairs at once (the reference asserts B == 1, :39-40). */
typedef struct {
    /* Weight matrices are fp32 [N,K] when scream_model_t.gemm_split == 0 and scream_pack_w_split images of them
     * (the pointers are then really const void*) when it is SCREAM_SPLIT_H2 / SCREAM_SPLIT_BF3. */
    const float* wqkv; /* [768,256]: q_proj | k_proj[0:128] | v_proj[0:128] | k_proj[128:256] | v_proj[128:256] */
    const float* wq;   /* rows [0,256) of wqkv as their own matrix (cross layers project q and k/v from different clouds) */
    const float* wkv;  /* rows [256,768) of wqkv as their own matrix */
    const float* wm;   /* merge [256,256] */
    const float* w1;   /* mlp.0 [1024,256] */
    const float* w2;   /* mlp.2 [256,1024] */
    const float* g1; const float* b1; /* norm1 */
    const float* g2; const float* b2; /* norm2 */
    /* gemm_split != 0 only; may be NULL.  scream_pack_tail image of (wm, w1, w2) for the same split: the forward then runs
     * attention apply, merge + norm1 and the FFN + norm2 as ONE launch per block (scream_layer_tail_f32) and ignores wm, w1, w2. */
    const void* tail;
    /* SCREAM_SPLIT_H2 only: power-of-two exponents (scream_amd/scales.py).  e_xq / e_xkv: the block input on the query side
     * and on the key/value side (different clouds in a cross layer), bounded by the LayerNorm that produced it; e_wqkv, e_wq,
     * e_wkv, e_wm_g, e_w1_g, e_w2_g: of the packed matrices above (the *_g ones only when wm / w1 / w2 run as separate GEMMs,
     * with e_att / e_m1 / e_h of tail_exps as their input exponents); tail_exps: of the fused tail and its image. */
    int32_t e_xq, e_xkv, e_wqkv, e_wq, e_wkv, e_wm_g, e_w1_g, e_w2_g;
    int32_t e_k, e_v; /* of K' = elu(k) + 1 and of V in the projection's K^T V epilogue */
    scream_tail_exps_t tail_exps;
    int32_t tail_next_q; /* the tail image carries the NEXT layer's query projection (scream_pack_tail, Wq_next) */
    /* fused-tail models on the fp16 splits only; may be NULL.  scream_pack_proj image of wqkv (exponent e_wqkv): the q/k/v
     * projection of a self layer then runs on scream_proj_qkv_f32 and ignores wqkv. */
    const void* proj;
    /* the tail image was packed with q_first (this layer's own Wq in front): the forward then launches no query projection for this
     * layer -- a self layer projects key/value chunks only (proj_kv: scream_pack_proj image of wkv with n_q = 0, or wkv on the GEMM),
     * a cross layer nothing on the query side -- and calls scream_layer_tail_f32 with Q == NULL */
    int32_t tail_q_first;
    const void* proj_kv;
} scream_layer_t;

typedef struct {
    int32_t n_self;  /* stem layers (models/pointnet.py:18-20) */
    int32_t n_cross; /* (self, cross) layer pairs (:22-25) */
    const float* dim_t; /* [84] */
    const float* emb_w; const float* emb_b; /* [256,3], [256] */
    const float* pre_g; const float* pre_b; /* pre_norm */
    const scream_layer_t* layers_host; /* HOST array: n_self stem layers, then cross.0, cross.1.layer, ... */
    const float* c0_w; const float* c0_b; /* coor_mlp.0 [256,256],[256] */
    const float* c2_w; const float* c2_b; /* coor_mlp.2 */
    const float* c4_w; const float* c4_b; /* coor_mlp.4 [3,256],[3] */
    /* NULL for PointTransformer (one stem for both clouds, models/pointnet.py:50-52).  DEMTransformer
     * (models/pointnet.py:113-118,143-145): HOST array of n_self layers applied to the SECOND clouds (stem_dem) while
     * layers_host[0..n_self) (stem_dsm) are applied to the first clouds only. */
    const scream_layer_t* stem_tgt_layers_host;
    /* 0: fp32 weights, fp32-input MFMA GEMMs (scream_gemm_f32).  SCREAM_SPLIT_H2 / SCREAM_SPLIT_BF3: every weight MATRIX
     * above (wqkv/wq/wkv/wm/w1/w2, c0_w, c2_w) is a scream_pack_w_split image and the GEMMs run on scream_gemm_split_f32 --
     * same fp32 results to rounding (tests/test_gpu_parity.py holds all three to the same tolerances). */
    int32_t gemm_split;
    /* SCREAM_SPLIT_H2 only: exponents of the coordinate MLP's two GEMMs (input: the last LayerNorm2 / relu(c0 . + b0)) */
    int32_t e_c0x, e_c0w, e_c2x, e_c2w;
    /* Fused-tail models only; may be NULL.  scream_pack_w_split image of the n_cross cross layers' wkv matrices stacked
     * ([512 n_cross, 256]; exponent e_wkv_cross): the forward then projects the frozen target features for ALL cross layers in
     * one launch right after the stem (and finalises their K^T V images in one) instead of once per layer, and ignores the
     * cross layers' wkv / e_wkv. */
    const float* wkv_cross;
    int32_t e_wkv_cross, e_k_cross, e_v_cross; /* e_k / e_v: the smallest over the cross layers */
    /* may be NULL: scream_pack_proj image of the same stacked matrix (n_q = 0, exponent e_wkv_cross): the batched target-side
     * projection then runs on scream_proj_qkv_f32 (wkv_cross may then be NULL) */
    const void* proj_cross;
} scream_model_t;

typedef struct {
    int32_t n_pairs;     /* B; clouds 0..B-1 are the sources, B..2B-1 the targets */
    int64_t rows_src;    /* packed rows of all source clouds (multiple of 128) */
    int64_t rows_total;  /* sources then targets (multiple of 128) */
    int32_t max_chunks;  /* max over clouds of ceil(len / SCREAM_KV_CHUNK) */
    const float* xyz;          /* [rows_total,3], zero in padding rows */
    const float* center;       /* [2B,3]: src_center per source cloud, zeros for targets */
    const int32_t* tile_cloud; /* [rows_total/128] */
    const int32_t* cloud_row0; /* [2B] */
    const int32_t* cloud_len;  /* [2B] */
} scream_batch_t;

/* Bytes of scratch scream_forward needs for this batch geometry.  fused_tail != 0: every layer carries a tail image (the
 * default of the split backends) -- 3 KB per row (two feature buffers and Q'); otherwise the attention output, LayerNorm1
 * output and FFN hidden buffers of the unfused chain are carved as well (9 KB per row).  n_cross_batched: scream_model_t.n_cross
 * when wkv_cross is set (partials and images of every cross layer's target-side K^T V live side by side), else 0. */
int64_t scream_forward_workspace_bytes(int64_t rows_src, int64_t rows_total, int32_t n_pairs,
                                       int32_t max_chunks, int32_t fused_tail, int32_t n_cross_batched);

/* src_pred [rows_src,3] (padding rows hold don't-care values).  If feats_out != NULL the final
 * source features [rows_src,256] are copied there (test hook).  trace: NULL, or a handle from
 * scream_trace_create -- then every kernel group of the forward is bracketed by HIP events on `stream`. */
int scream_forward(const scream_model_t* model, const scream_batch_t* batch, void* workspace,
                   int64_t workspace_bytes, float* src_pred, float* feats_out, void* trace,
                   void* stream);

/* Per-launch timing for bench.py's roofline leg.  A trace owns 2*capacity HIP events; records beyond
 * capacity are dropped.  scream_trace_read (after the stream has been synchronised) fills, per record,
 * the elapsed ms, the kind (a SCREAM_EPI_* value for a GEMM with its M, N, K; 100 = embed, 101 = K^T V
 * reduce, 102 = attention apply, 103 = 256->3 head, with M = rows) and returns the record count. */
void* scream_trace_create(int32_t capacity);
void scream_trace_destroy(void* trace);
int scream_trace_reset(void* trace);
int scream_trace_read(void* trace, int32_t max_records, float* ms, int32_t* kind, int64_t* m,
                      int32_t* n, int32_t* k);
/* Start of every record in ms after the start of record 0.  One trace may be handed to forwards running on
 * several streams at once (pairs are independent, so a step can run as concurrent lanes); their records then
 * overlap in time and the busy time of a kernel class is the union of its intervals, not the sum. */
int scream_trace_read_starts(void* trace, int32_t max_records, float* start_ms);

/* ---- A7: thresholded 1-NN of every query point in its pair's target cloud.
 * Replaces square_distance(src_pred / s, tgt / s)[0].min(dim=1) and the threshold compare at
 * evaluate_3d_match.py:94-95 (also models/pointnet.py:71-72, evaluate_kitti.py:53-54) without
 * materialising the N x M matrix.  Bit-exact fp32 arithmetic of utils.py:72-78 as torch-CPU
 * executes it: a = x / fp32(s) (IEEE divide); dot = fma(a2,b2,fma(a1,b1,a0*b0));
 * |a|^2 = (a0^2 + a1^2) + a2^2; d = ((-2 dot) + |a|^2) + |b|^2; ties -> lowest target index.
 * One launch set covers n_pairs pairs.  query/ref are packed [rows,3]; q_row0/q_len, r_row0/r_len
 * [n_pairs] int32 (device); s [n_pairs] fp32 (device).  Outputs are indexed by packed query row:
 * idx (int32, index inside the pair's target cloud; -1 in rows outside any cloud), dmin (fp32),
 * valid (uint8: dmin < thresh).  Scratch: ref_prep 4 floats per ref row, keys one uint64 per query row. */
int scream_nn_search(const float* query, const float* ref, const int32_t* q_row0,
                     const int32_t* q_len, const int32_t* r_row0, const int32_t* r_len,
                     const float* s, int32_t n_pairs, int32_t max_q_len, int32_t max_r_len,
                     int64_t q_rows_total, int64_t r_rows_total, float thresh, float* ref_prep,
                     uint64_t* keys, int32_t* idx, float* dmin, uint8_t* valid, void* stream);

/* Dense utils.square_distance (utils.py:72-78): out[B,N,M] = -2 src.dst^T + |src|^2 + |dst|^2 with the
 * rounding sequence above.  API compatibility only -- the hot path uses scream_nn_search. */
int scream_square_distance(const float* src, const float* dst, float* out, int32_t B, int32_t N,
                           int32_t M, void* stream);

/* ---- A8 + A9 (+ A10): correspondence gather fused into the weighted Kabsch solve.
 * Replaces evaluate_3d_match.py:96-101 and utils.py:138-178 for n_pairs pairs in one launch:
 *   A_n = src[n] / s + c, B_n = ref[idx[n]] / s + c for every n with valid[n] (idx == NULL: B row = n,
 *   the corr="src_pred" branch of evaluate_3d_match.py:99-101); centroids sum/(K + 1e-6);
 *   H = sum (A-cA)(B-cB)^T; 3x3 SVD by one-sided Jacobi in registers (fp64), R = V diag(1,1,det(V U^T)) U^T,
 *   t = cB - R cA.  K == 0 gives the identity, as torch.svd of a zero matrix does.
 * T_out [n_pairs,16] row-major 4x4; n_corr [n_pairs] (int32) = K.  c is [n_pairs,3]. */
int scream_kabsch_corr(const float* src, const float* ref, const int32_t* src_row0,
                       const int32_t* src_len, const int32_t* ref_row0, const int32_t* idx,
                       const uint8_t* valid, const float* s, const float* c, int32_t n_pairs,
                       float* T_out, int32_t* n_corr, void* stream);

/* Drop-in for utils.rigid_transform_3d (utils.py:138): A, B [bs,K,3] dense, w [bs,K] or NULL.
 * Weights below weight_threshold count as 0 (utils.py:151).  T_out [bs,16]. */
int scream_rigid_transform_3d(const float* A, const float* B, const float* w,
                              float weight_threshold, int32_t bs, int32_t K, float* T_out,
                              void* stream);

/* ---- A10: RE [deg] = acos(clamp((tr(Rp^T Rg) - 1)/2)) * 180/pi, TE = |tp - tg| (utils.py:181-189).
 * T_pred, T_gt [n,16]; re, te [n]. */
int scream_transformation_error(const float* T_pred, const float* T_gt, int32_t n, float* re,
                                float* te, void* stream);

/* ---- A11 (the per-pair L1 point loss the evaluators report): loss[p] = mean_n sum_xyz |src_pred_n - (R_p src_n + t_p)|
 * Replaces PointTransformer.loss (models/pointnet.py:93-99) called once per pair (evaluate_3d_match.py:86): src_pred / src packed
 * [rows,3] in the normalised frame, rot [n_pairs,9] / trans [n_pairs,3] the ground-truth pose of the same frame. */
int scream_point_loss(const float* src_pred, const float* src, const int32_t* src_row0, const int32_t* src_len,
                      const float* rot, const float* trans, int32_t n_pairs, float* loss, void* stream);

/* ---- next row: batched point-to-point ICP refinement on the GPU.
 * Replaces o3d.registration_icp(src_pc, tgt_pc, max_correspondence_distance, init) at
 * evaluate_3d_match.py:106-113 / evaluate_kitti.py:61-70 (open3d is absent here: parity with open3d is
 * unpinned; the loop is open3d's published RegistrationICP and is checked against oracle/icp_ref.py).
 * src/ref packed, normalised frame; metric points are x / s + c.  T [n_pairs,16] holds the initial
 * transforms on entry and the refined ones on return.  Per iteration (one launch): transform the source, thresholded
 * 1-NN (distance <= max_corr_dist), fitness = #corr / N and inlier_rmse = sqrt(mean d^2); stop a pair when
 * both change by less than rel_fitness / rel_rmse or after max_iter updates; otherwise compose the Kabsch
 * update of the correspondences.  fitness_rmse [n_pairs,2] and iters [n_pairs] may be NULL.  The clouds of a batch may be
 * packed in any order (the per-chunk partial sums are indexed by a prefix sum of the source lengths).
 * ASYNCHRONOUS, like every entry point: scream_icp_p2p enqueues the whole schedule (max_iter + 2 launches at most) and never
 * waits for the device; a pair that has stopped freezes there (its blocks return at their first instruction).  A caller with a
 * LONG schedule (KITTI: 1000) that wants to stop launching once every pair has stopped asks for the schedule in pieces:
 * scream_icp_p2p_range enqueues launches [it_begin, it_end) of the same run (launch `it` = evaluate search it - 1, stop or
 * update, search it; launch max_iter + 1 = the evaluation of the last search; it_begin == 0 also does the set-up, so a run
 * starts there; every call of a run gets the same arguments and workspace) and then copies the per-pair stopped flags to
 * done_flags [n_pairs] (device int32, may be NULL) -- the caller copies them to pinned memory behind an event and decides on the
 * host, between pieces, without ever blocking the stream (scream_amd/ops.py: IcpRun).  T is complete for pair p once its flag
 * is set.  Results do not depend on how the schedule was cut. */
int64_t scream_icp_workspace_bytes(int64_t src_rows_total, int64_t ref_rows_total, int32_t n_pairs);
int scream_icp_p2p_range(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                         const int32_t* ref_row0, const int32_t* ref_len, const float* s, const float* c,
                         int32_t n_pairs, int32_t max_src_len, int32_t max_ref_len, int64_t src_rows_total,
                         int64_t ref_rows_total, float max_corr_dist, int32_t max_iter, float rel_fitness,
                         float rel_rmse, float* T, float* fitness_rmse, int32_t* iters, int32_t it_begin, int32_t it_end,
                         int32_t* done_flags, void* workspace, int64_t workspace_bytes, void* stream);
int scream_icp_p2p(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                   const int32_t* ref_row0, const int32_t* ref_len, const float* s, const float* c,
                   int32_t n_pairs, int32_t max_src_len, int32_t max_ref_len, int64_t src_rows_total,
                   int64_t ref_rows_total, float max_corr_dist, int32_t max_iter, float rel_fitness,
                   float rel_rmse, float* T, float* fitness_rmse, int32_t* iters, void* workspace,
                   int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCREAM_HIP_H */
