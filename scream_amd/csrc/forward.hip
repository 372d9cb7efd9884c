// A1-A6: the PointTransformer layer schedule (models/pointnet.py:45-60) over a packed batch of pairs.
//
// The whole forward is one host call that enqueues every kernel on the caller's stream (no allocation,
// no synchronisation, graph-capturable).  Rows are packed [all source clouds | all target clouds], so
//   * the 6 stem layers, which the reference applies to tgt and then src with the SAME weights
//     (pointnet.py:50-52), run as one launch set over every row of every cloud of every pair;
//   * the cross stage (pointnet.py:53-57) runs over the source prefix only, the target features stay
//     frozen in the rows the stem left them in.
#include <vector>

#include "common.h"

namespace {

constexpr int D = SCREAM_D_MODEL;
constexpr int KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;

struct Workspace {
    float *x0, *x1, *q, *att, *m1, *hid, *kvp, *kv;
    char* kvimg;  // per-cloud operand images of the fused layer tail
    float* kvp_cross;   // batched target-side projection of the cross stage: K^T V partials of every cross layer ...
    char* kvimg_cross;  // ... and their images, [layer][target cloud]
    int64_t kvp_cross_stride, kvimg_cross_stride;  // floats / bytes per layer
    int64_t bytes;
};

// fused: every layer runs its tail as one launch -- the attention output, LayerNorm1 output and FFN hidden buffers of the
// unfused chain (6 KB per row) are then not carved (3 KB per row remain: two feature buffers and Q'); the coordinate MLP's
// two intermediates go into Q' and the idle feature buffer.
Workspace carve(void* base, int64_t rows_src, int64_t rows_total, int32_t n_pairs, int32_t max_chunks, bool fused, int32_t n_cross_batched) {
    Workspace w;
    float* p = reinterpret_cast<float*>(base);
    auto take = [&](int64_t n) {
        float* r = p;
        p += (n + 63) / 64 * 64;  // keep every buffer 256-byte aligned
        return r;
    };
    w.x0 = take(rows_total * D);
    w.x1 = take(rows_total * D);
    w.q = take(rows_total * D);
    w.att = fused ? nullptr : take(rows_total * D);
    w.m1 = fused ? nullptr : take(rows_total * D);
    w.hid = fused ? nullptr : take(rows_total * 4 * D);
    w.kvp = take(rows_total / SCREAM_ROW_TILE * SCREAM_NHEAD * KV_ELEMS);  // one K^T V partial per 128-row tile and head
    w.kv = take((int64_t)2 * n_pairs * SCREAM_NHEAD * KV_ELEMS);
    w.kvimg = reinterpret_cast<char*>(take((int64_t)2 * n_pairs * scream_kv_image_bytes() / 4));
    w.kvp_cross_stride = (rows_total - rows_src) / SCREAM_ROW_TILE * SCREAM_NHEAD * KV_ELEMS;
    w.kvimg_cross_stride = (int64_t)n_pairs * scream_kv_image_bytes();
    w.kvp_cross = take(n_cross_batched * w.kvp_cross_stride);
    w.kvimg_cross = reinterpret_cast<char*>(take(n_cross_batched * w.kvimg_cross_stride / 4));
    w.bytes = (p - reinterpret_cast<float*>(base)) * (int64_t)sizeof(float);
    return w;
}

// Optional per-launch timing (bench.py's roofline leg): HIP events recorded on the launch stream around
// every kernel group of the forward.  Owned by the caller through an opaque handle; no global state.
struct Trace {
    std::vector<hipEvent_t> ev;  // 2 per record
    std::vector<int32_t> kind, n, k;
    std::vector<int64_t> m;
    int count = 0;
    int capacity = 0;
};

struct Scope {  // records start on construction, stop on destruction
    Trace* t;
    int slot = -1;
    hipStream_t st;
    Scope(Trace* t_, int kind, int64_t m, int n, int k, void* stream) : t(t_), st(as_stream(stream)) {
        if (!t || t->count >= t->capacity) return;
        slot = t->count++;
        t->kind[slot] = kind;
        t->m[slot] = m;
        t->n[slot] = n;
        t->k[slot] = k;
        (void)hipEventRecord(t->ev[2 * slot], st);
    }
    ~Scope() {
        if (slot >= 0) (void)hipEventRecord(t->ev[2 * slot + 1], st);
    }
};

#define TRY(call)                 \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != 0) return rc_; \
    } while (0)

enum { TR_TAIL_FUSED = 7, TR_EMBED = 100, TR_KV_REDUCE = 101, TR_ATTN_APPLY = 102, TR_COOR_HEAD = 103 };

struct Ctx {
    void* st;
    Trace* tr;
    int split;    // scream_model_t.gemm_split: 0 = fp32 weights, else the weights are packed operand planes, GEMMs on the split kernel
    bool frag;    // every layer has a fused-tail image: the features travel FRAGMENT-major between the kernels (SCREAM_ACT_FRAG)
};

// a_exp / w_exp: SCREAM_SPLIT_H2's operand exponents (ignored otherwise)
int gemm(const Ctx& c, const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int64_t M, int N, int K,
         int epi, int n_act, const float* bias, const float* res, const float* g, const float* b, int a_exp, int w_exp,
         int layout = 0) {
    Scope sc(c.tr, epi, M, N, K, c.st);
    if (c.split)
        return scream_gemm_split_f32(A, lda, W, C, ldc, M, N, K, epi, n_act, bias, res, D, g, b, layout, c.split, a_exp, w_exp, c.st);
    return scream_gemm_f32(A, lda, W, C, ldc, M, N, K, epi, n_act, bias, res, D, g, b, c.st);
}

// proj: the ring kernel's image of the same matrix (fused-tail models on an fp16 split; NULL: the 8-wave GEMM)
int gemm_qkv(const Ctx& c, const float* A, const float* W, const void* proj, float* Q, int64_t M, int N, int n_q, const scream_batch_t& b,
             int64_t row_base, float* kvp, int a_exp, int w_exp, int k_exp, int v_exp) {
    Scope sc(c.tr, 5, M, N, D, c.st);
    if (proj && c.frag && (c.split == SCREAM_SPLIT_H2 || c.split == SCREAM_SPLIT_H1))
        return scream_proj_qkv_f32(A, proj, Q, M, N, n_q, b.tile_cloud, b.cloud_row0, b.cloud_len, row_base, kvp, c.split, a_exp, w_exp,
                                   k_exp, v_exp, c.st);
    if (c.split)
        return scream_gemm_qkv_split_f32(A, D, W, Q, D, M, N, D, n_q, b.tile_cloud, b.cloud_row0, b.cloud_len, row_base, kvp,
                                         c.frag ? (SCREAM_LAYOUT_A_FRAG | (n_q ? SCREAM_LAYOUT_C_FRAG : 0)) : 0, c.split, a_exp,
                                         w_exp, k_exp, v_exp, c.st);
    return scream_gemm_qkv_f32(A, D, W, Q, D, M, N, D, n_q, b.tile_cloud, b.cloud_row0, b.cloud_len, row_base, kvp, c.st);
}

// merge + norm1 + FFN + norm2 (models/transformer.py:83-88); x is the block input (residual of BOTH norms).
// x / y already point at packed row `row0`; the scratch buffers are indexed by packed row as well.
int mha_tail(const Ctx& c, const scream_layer_t& L, const Workspace& w, const float* x, float* y, int64_t row0,
             int64_t rows) {
    float* att = w.att + row0 * D;
    float* m1 = w.m1 + row0 * D;
    float* hid = w.hid + row0 * 4 * D;
    const scream_tail_exps_t& e = L.tail_exps;  // the operands of the three GEMMs are the fused tail's: att, m1, hidden
    TRY(gemm(c, att, D, L.wm, m1, D, rows, D, D, SCREAM_EPI_RES_LN, 0, nullptr, x, L.g1, L.b1, e.e_att, L.e_wm_g));
    TRY(gemm(c, m1, D, L.w1, hid, 4 * D, rows, 4 * D, D, SCREAM_EPI_RELU, 0, nullptr, nullptr, nullptr, nullptr, e.e_m1, L.e_w1_g));
    TRY(gemm(c, hid, 4 * D, L.w2, y, D, rows, D, 4 * D, SCREAM_EPI_RES_LN, 0, nullptr, x, L.g2, L.b2, e.e_h, L.e_w2_g));
    return 0;
}

// Self attention over packed rows [row0, row0 + rows) whose clouds are [cloud_begin, cloud_begin + n_clouds)
// (transformer.py:74-90 with q = k = v).  x / y are the full feature buffers (row 0 = packed row 0).
// The q/k/v projection reduces K^T V in its epilogue, so K' and V never reach HBM.
// next_q: the tail image carries the next (cross) layer's query projection -- Q' of that layer is written over this layer's (w.q)
int mha_self(const Ctx& c, const scream_layer_t& L, const scream_batch_t& b, const Workspace& w, const float* x,
             float* y, int64_t row0, int64_t rows, int32_t cloud_begin, int32_t n_clouds, bool next_q = false) {
    const float* xr = x + row0 * D;
    float* qr = w.q + row0 * D;
    float* kvp = w.kvp + row0 / SCREAM_ROW_TILE * SCREAM_NHEAD * KV_ELEMS;
    const bool qf = c.frag && L.tail_q_first;  // the layer tail projects its own queries: key/value chunks only here
    if (qf) TRY(gemm_qkv(c, xr, L.wkv, L.proj_kv, nullptr, rows, 2 * D, 0, b, row0, kvp, L.e_xkv, L.e_wkv, L.e_k, L.e_v));
    else TRY(gemm_qkv(c, xr, L.wqkv, L.proj, qr, rows, 3 * D, D, b, row0, kvp, L.e_xq, L.e_wqkv, L.e_k, L.e_v));
    if (c.frag) {  // apply + merge + norm1 + FFN + norm2 in one launch
        {
            Scope sc(c.tr, TR_KV_REDUCE, rows, 0, 0, c.st);
            TRY(scream_kv_finalize_image(kvp, b.cloud_row0, b.cloud_len, row0, cloud_begin, n_clouds, w.kvimg, 1, 0, 0, c.split, c.st));
        }
        // merge (256) + FFN up and down (2 x 1024) (+ the next layer's query projection, 256) output columns per row
        Scope sc(c.tr, TR_TAIL_FUSED, rows, ((next_q || qf) ? 10 : 9) * D, D, c.st);
        return scream_layer_tail_f32(qf ? nullptr : qr, w.kvimg, b.tile_cloud + row0 / SCREAM_ROW_TILE, 0, b.cloud_len, xr, L.tail, L.g1,
                                     L.b1, L.g2, L.b2, y + row0 * D, (next_q && !qf) ? qr : nullptr, rows, c.split, &L.tail_exps, c.st);
    }
    {
        Scope sc(c.tr, TR_KV_REDUCE, rows, 0, 0, c.st);
        TRY(scream_kv_finalize(kvp, b.cloud_row0, b.cloud_len, row0, cloud_begin, n_clouds, w.kv, c.st));
    }
    {
        Scope sc(c.tr, TR_ATTN_APPLY, rows, 0, 0, c.st);
        TRY(scream_attn_apply(qr, D, w.kv, b.tile_cloud + row0 / SCREAM_ROW_TILE, 0, b.cloud_len, w.att + row0 * D, D, rows,
                              c.st));
    }
    return mha_tail(c, L, w, xr, y + row0 * D, row0, rows);
}

// The target side of EVERY cross layer at once (fused-tail models with scream_model_t.wkv_cross): the target features are
// frozen after the stem (models/pointnet.py:53-57), so the n_cross key/value projections read the same rows -- one GEMM with
// N = 512 n_cross (12 n_cross column tiles per 256-row tile instead of n_cross launches of two: the persistent grid's partial
// last round shrinks from 15 % to under 2 %) and one finalize launch for all n_cross x n_pairs K^T V images.
int cross_kv_all(const Ctx& c, const scream_model_t& m, const scream_batch_t& b, const Workspace& w, const float* x_tgt) {
    const int64_t rs = b.rows_src, rt = b.rows_total - b.rows_src;
    const scream_layer_t& L0 = m.layers_host[m.n_self + 1];  // every cross layer sees the same target features: one e_xkv
    TRY(gemm_qkv(c, x_tgt, m.wkv_cross, m.proj_cross, nullptr, rt, 2 * D * m.n_cross, 0, b, rs, w.kvp_cross, L0.e_xkv, m.e_wkv_cross, m.e_k_cross,
                 m.e_v_cross));
    Scope sc(c.tr, TR_KV_REDUCE, rt, 0, 0, c.st);
    // images are indexed by ABSOLUTE cloud (targets are clouds n_pairs .. 2 n_pairs - 1): layer l's block starts n_pairs images early
    return scream_kv_finalize_image(w.kvp_cross, b.cloud_row0, b.cloud_len, rs, b.n_pairs, b.n_pairs,
                                 w.kvimg_cross - (int64_t)b.n_pairs * scream_kv_image_bytes(), m.n_cross, w.kvp_cross_stride,
                                 w.kvimg_cross_stride, c.split, c.st);
}

// Cross attention: queries from the source rows, keys/values from the frozen target rows (transformer.py:130).
// kvimg_layer != NULL: this layer's target-side K^T V images were already built by cross_kv_all (image of target cloud j at
// kvimg_layer + j * scream_kv_image_bytes()).
// q_ready: Q' (w.q) was written by the layer tail of the self layer in front (scream_layer_t.tail_next_q)
int mha_cross(const Ctx& c, const scream_layer_t& L, const scream_batch_t& b, const Workspace& w, const float* x_src,
              const float* x_tgt, float* y, const char* kvimg_layer, bool q_ready = false) {
    const int64_t rs = b.rows_src, rt = b.rows_total - b.rows_src;
    const bool qf = c.frag && L.tail_q_first;  // the layer tail projects its own queries from x_src
    if (!q_ready && !qf)
        TRY(gemm(c, x_src, D, L.wq, w.q, D, rs, D, D, SCREAM_EPI_ELU1, D, nullptr, nullptr, nullptr, nullptr, L.e_xq, L.e_wq,
                 c.frag ? (SCREAM_LAYOUT_A_FRAG | SCREAM_LAYOUT_C_FRAG) : 0));
    if (c.frag && kvimg_layer) {
        Scope sc(c.tr, TR_TAIL_FUSED, rs, (qf ? 10 : 9) * D, D, c.st);
        // tile_cloud holds source cloud i for the source tiles; its target cloud's image is entry i of this layer's block
        return scream_layer_tail_f32(qf ? nullptr : w.q, kvimg_layer, b.tile_cloud, 0, b.cloud_len + b.n_pairs, x_src, L.tail, L.g1, L.b1, L.g2,
                                     L.b2, y, nullptr, rs, c.split, &L.tail_exps, c.st);
    }
    TRY(gemm_qkv(c, x_tgt, L.wkv, nullptr, nullptr, rt, 2 * D, 0, b, rs, w.kvp, L.e_xkv, L.e_wkv, L.e_k, L.e_v));
    if (c.frag) {
        {
            Scope sc(c.tr, TR_KV_REDUCE, rt, 0, 0, c.st);
            TRY(scream_kv_finalize_image(w.kvp, b.cloud_row0, b.cloud_len, rs, b.n_pairs, b.n_pairs, w.kvimg, 1, 0, 0, c.split, c.st));
        }
        Scope sc(c.tr, TR_TAIL_FUSED, rs, (qf ? 10 : 9) * D, D, c.st);
        return scream_layer_tail_f32(qf ? nullptr : w.q, w.kvimg, b.tile_cloud, b.n_pairs, b.cloud_len, x_src, L.tail, L.g1, L.b1, L.g2, L.b2,
                                     y, nullptr, rs, c.split, &L.tail_exps, c.st);
    }
    {
        Scope sc(c.tr, TR_KV_REDUCE, rt, 0, 0, c.st);
        TRY(scream_kv_finalize(w.kvp, b.cloud_row0, b.cloud_len, rs, b.n_pairs, b.n_pairs, w.kv, c.st));
    }
    {
        Scope sc(c.tr, TR_ATTN_APPLY, rs, 0, 0, c.st);
        TRY(scream_attn_apply(w.q, D, w.kv, b.tile_cloud, b.n_pairs, b.cloud_len, w.att, D, rs, c.st));
    }
    return mha_tail(c, L, w, x_src, y, 0, rs);
}

}  // namespace

extern "C" const char* scream_version(void) { return "scream_hip gfx950 abi18"; }
extern "C" int scream_abi_version(void) { return 18; }

extern "C" void* scream_trace_create(int32_t capacity) {
    if (capacity <= 0) return nullptr;
    Trace* t = new Trace();
    t->capacity = capacity;
    t->ev.resize(2 * (size_t)capacity);
    t->kind.resize(capacity);
    t->n.resize(capacity);
    t->k.resize(capacity);
    t->m.resize(capacity);
    for (auto& e : t->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            delete t;
            return nullptr;
        }
    }
    return t;
}

extern "C" void scream_trace_destroy(void* trace) {
    Trace* t = reinterpret_cast<Trace*>(trace);
    if (!t) return;
    for (auto& e : t->ev) (void)hipEventDestroy(e);
    delete t;
}

extern "C" int scream_trace_reset(void* trace) {
    SCREAM_REQUIRE(trace, SCREAM_EINVAL);
    reinterpret_cast<Trace*>(trace)->count = 0;
    return 0;
}

extern "C" int scream_trace_read(void* trace, int32_t max_records, float* ms, int32_t* kind, int64_t* m, int32_t* n,
                                 int32_t* k) {
    SCREAM_REQUIRE(trace && ms && kind && m && n && k && max_records >= 0, SCREAM_EINVAL);
    Trace* t = reinterpret_cast<Trace*>(trace);
    const int cnt = t->count < max_records ? t->count : max_records;
    for (int i = 0; i < cnt; ++i) {
        if (hipEventElapsedTime(&ms[i], t->ev[2 * i], t->ev[2 * i + 1]) != hipSuccess) return SCREAM_EINVAL;
        kind[i] = t->kind[i];
        m[i] = t->m[i];
        n[i] = t->n[i];
        k[i] = t->k[i];
    }
    return cnt;
}

extern "C" int scream_trace_read_starts(void* trace, int32_t max_records, float* start_ms) {
    SCREAM_REQUIRE(trace && start_ms && max_records >= 0, SCREAM_EINVAL);
    Trace* t = reinterpret_cast<Trace*>(trace);
    const int cnt = t->count < max_records ? t->count : max_records;
    for (int i = 0; i < cnt; ++i)
        if (hipEventElapsedTime(&start_ms[i], t->ev[0], t->ev[2 * i]) != hipSuccess) return SCREAM_EINVAL;
    return cnt;
}

extern "C" int64_t scream_forward_workspace_bytes(int64_t rows_src, int64_t rows_total, int32_t n_pairs,
                                                  int32_t max_chunks, int32_t fused_tail, int32_t n_cross_batched) {
    if (rows_src < 0 || rows_total < rows_src || n_pairs < 0 || max_chunks < 0 || n_cross_batched < 0) return SCREAM_EINVAL;
    return carve(nullptr, rows_src, rows_total, n_pairs, max_chunks, fused_tail != 0, n_cross_batched).bytes + 256;
}

extern "C" int scream_forward(const scream_model_t* model, const scream_batch_t* batch, void* workspace,
                              int64_t workspace_bytes, float* src_pred, float* feats_out, void* trace, void* stream) {
    SCREAM_REQUIRE(model && batch && workspace && src_pred, SCREAM_EINVAL);
    const scream_model_t& m = *model;
    const scream_batch_t& b = *batch;
    SCREAM_REQUIRE(m.layers_host && m.n_self >= 0 && m.n_cross >= 0 &&
                       (m.gemm_split == 0 || m.gemm_split == SCREAM_SPLIT_H1 || m.gemm_split == SCREAM_SPLIT_H2 || m.gemm_split == SCREAM_SPLIT_BF3),
                   SCREAM_EINVAL);
    SCREAM_REQUIRE(b.n_pairs > 0 && b.rows_src > 0 && b.rows_total > b.rows_src && b.max_chunks > 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(b.rows_src % SCREAM_ROW_TILE == 0 && b.rows_total % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(b.xyz && b.center && b.tile_cloud && b.cloud_row0 && b.cloud_len, SCREAM_EINVAL);
    // fragment-major features between the kernels iff EVERY layer carries a fused-tail image (all or none)
    int n_layers = m.n_self + 2 * m.n_cross, n_tail = 0;
    for (int i = 0; i < n_layers; ++i) n_tail += m.layers_host[i].tail != nullptr;
    if (m.stem_tgt_layers_host)
        for (int i = 0; i < m.n_self; ++i, ++n_layers) n_tail += m.stem_tgt_layers_host[i].tail != nullptr;
    SCREAM_REQUIRE(n_tail == 0 || (n_tail == n_layers && m.gemm_split != 0), SCREAM_EINVAL);
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    const bool batched_kv = n_tail > 0 && m.wkv_cross != nullptr && m.n_cross > 0;
    const Workspace w = carve(reinterpret_cast<void*>(base), b.rows_src, b.rows_total, b.n_pairs, b.max_chunks, n_tail > 0,
                              batched_kv ? m.n_cross : 0);
    SCREAM_REQUIRE((int64_t)(base - reinterpret_cast<uintptr_t>(workspace)) + w.bytes <= workspace_bytes, SCREAM_EINVAL);
    const Ctx c{stream, reinterpret_cast<Trace*>(trace), m.gemm_split, n_tail > 0};

    const int64_t rs = b.rows_src, ra = b.rows_total;
    {
        Scope sc(c.tr, TR_EMBED, ra, 0, 0, stream);
        if (c.frag)  // straight into the layout the projections and the layer tail read
            TRY(scream_pe_embed_ln_frag(b.xyz, b.tile_cloud, b.center, m.dim_t, m.emb_w, m.emb_b, m.pre_g, m.pre_b, w.x0, ra, stream));
        else
            TRY(scream_pe_embed_ln(b.xyz, b.tile_cloud, b.center, m.dim_t, m.emb_w, m.emb_b, m.pre_g, m.pre_b, w.x0, ra, stream));
    }
    float* cur = w.x0;
    float* nxt = w.x1;
    for (int i = 0; i < m.n_self; ++i) {
        if (!m.stem_tgt_layers_host) {  // pointnet.py:50-52: one set of stem weights for both clouds
            TRY(mha_self(c, m.layers_host[i], b, w, cur, nxt, 0, ra, 0, 2 * b.n_pairs));
        } else {  // DEMTransformer (pointnet.py:113-118,143-145): stem_dsm on the first clouds, stem_dem on the second
            TRY(mha_self(c, m.layers_host[i], b, w, cur, nxt, 0, rs, 0, b.n_pairs));
            TRY(mha_self(c, m.stem_tgt_layers_host[i], b, w, cur, nxt, rs, ra - rs, b.n_pairs, b.n_pairs));
        }
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
    const float* x_tgt = cur + rs * D;  // frozen from here on: the cross stage only writes rows [0, rs)
    if (batched_kv) TRY(cross_kv_all(c, m, b, w, x_tgt));
    for (int i = 0; i < 2 * m.n_cross; ++i) {  // pointnet.py:53-57
        const scream_layer_t& L = m.layers_host[m.n_self + i];
        if (i % 2 == 0) {
            TRY(mha_self(c, L, b, w, cur, nxt, 0, rs, 0, b.n_pairs, c.frag && L.tail_next_q));
        } else {
            const bool q_ready = c.frag && m.layers_host[m.n_self + i - 1].tail_next_q;
            TRY(mha_cross(c, L, b, w, cur, x_tgt, nxt, batched_kv ? w.kvimg_cross + (i / 2) * w.kvimg_cross_stride : nullptr, q_ready));
        }
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
    // coor_mlp, pointnet.py:27-33,60
    float* c_mid = c.frag ? w.q : w.m1;   // (fused: Q' and the idle feature buffer are dead by now)
    float* c_out = c.frag ? nxt : w.att;
    TRY(gemm(c, cur, D, m.c0_w, c_mid, D, rs, D, D, SCREAM_EPI_BIAS_RELU, 0, m.c0_b, nullptr, nullptr, nullptr, m.e_c0x, m.e_c0w,
             c.frag ? SCREAM_LAYOUT_A_FRAG : 0));  // the output (and everything behind it) is row-major
    TRY(gemm(c, c_mid, D, m.c2_w, c_out, D, rs, D, D, SCREAM_EPI_BIAS_RELU, 0, m.c2_b, nullptr, nullptr, nullptr, m.e_c2x, m.e_c2w));
    {
        Scope sc(c.tr, TR_COOR_HEAD, rs, 0, 0, stream);
        TRY(scream_coor_head(c_out, m.c4_w, m.c4_b, src_pred, rs, stream));
    }
    if (feats_out && c.frag) {
        TRY(scream_act_layout(cur, feats_out, rs, 0, stream));
    } else if (feats_out) {
        hipError_t e = hipMemcpyAsync(feats_out, cur, (size_t)rs * D * sizeof(float), hipMemcpyDeviceToDevice,
                                      as_stream(stream));
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
