// A1: sine position embedding + 1x1-conv embedding + pre_norm, and the 256->3 head of coor_mlp (A6).
//
// Both kernels are row-streaming (HBM-bound): one wavefront owns one 256-wide feature row, so every
// global access of a wave is a full, contiguous 256 B / 1 KiB segment.
#include "common.h"

namespace {

constexpr int D = SCREAM_D_MODEL;
constexpr int NPF = 84;  // num_pos_feats = 256 // 3 // 2 * 2 (models/transformer.py:148)

// feats[row] = LN(pe(xyz[row]) + W_e (xyz[row] - center[cloud]) + b_e); models/pointnet.py:45-48.
// Feature f < 252: axis a = f / 84, i = f % 84, value sin(p) for even i, cos(p) for odd i with
// p = (x_a * 2 pi) / dim_t[i] (models/transformer.py:172-176); features 252..255 are the zero pad (:179).
__global__ __launch_bounds__(256) void pe_embed_ln_kernel(const float* __restrict__ xyz,
                                                         const int32_t* __restrict__ tile_cloud,
                                                         const float* __restrict__ center,
                                                         const float* __restrict__ dim_t,
                                                         const float* __restrict__ emb_w,
                                                         const float* __restrict__ emb_b,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ feats, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float x[3] = {xyz[row * 3 + 0], xyz[row * 3 + 1], xyz[row * 3 + 2]};
    const int cloud = tile_cloud[row / SCREAM_ROW_TILE];
    const float xe[3] = {x[0] - center[cloud * 3 + 0], x[1] - center[cloud * 3 + 1], x[2] - center[cloud * 3 + 2]};
    const float two_pi = 6.283185307179586f;  // fp32(1.0 * 2 * math.pi), transformer.py:155,171

    float v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = lane + 64 * k;
        float pe = 0.f;
        if (f < 3 * NPF) {
            const int a = f / NPF, i = f - a * NPF;
            const float p = (x[a] * two_pi) / dim_t[i];
            pe = (i & 1) ? cosf(p) : sinf(p);
        }
        const float e = emb_w[f * 3 + 0] * xe[0] + emb_w[f * 3 + 1] * xe[1] + emb_w[f * 3 + 2] * xe[2] + emb_b[f];
        v[k] = pe + e;
        s += v[k];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = v[k] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = lane + 64 * k;
        feats[row * D + f] = (v[k] - mean) * rstd * gamma[f] + beta[f];
    }
}

// out[row, j] = X[row, :] . W[j, :] + b[j], j < 3 (models/pointnet.py:32).
__global__ __launch_bounds__(256) void coor_head_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ b, float* __restrict__ out,
                                                       int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + 0 * D + lane * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(W + 1 * D + lane * 4);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(W + 2 * D + lane * 4);
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    for (int64_t row = wave; row < rows; row += n_waves) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + row * D + lane * 4);
        float d0 = x[0] * w0[0] + x[1] * w0[1] + x[2] * w0[2] + x[3] * w0[3];
        float d1 = x[0] * w1[0] + x[1] * w1[1] + x[2] * w1[2] + x[3] * w1[3];
        float d2 = x[0] * w2[0] + x[1] * w2[1] + x[2] * w2[2] + x[3] * w2[3];
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        d2 = wave_sum(d2);
        if (lane == 0) {
            out[row * 3 + 0] = d0 + b0;
            out[row * 3 + 1] = d1 + b1;
            out[row * 3 + 2] = d2 + b2;
        }
    }
}

}  // namespace

extern "C" int scream_pe_embed_ln(const float* xyz, const int32_t* tile_cloud, const float* center,
                                  const float* dim_t, const float* emb_w, const float* emb_b, const float* gamma,
                                  const float* beta, float* feats, int64_t rows, void* stream) {
    SCREAM_REQUIRE(xyz && tile_cloud && center && dim_t && emb_w && emb_b && gamma && beta && feats, SCREAM_EINVAL);
    SCREAM_REQUIRE(rows >= 0 && rows % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    if (rows == 0) return 0;
    const int64_t blocks = rows / 4;
    SCREAM_REQUIRE(blocks < (1ll << 31), SCREAM_EUNSUPPORTED);
    pe_embed_ln_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(
        xyz, tile_cloud, center, dim_t, emb_w, emb_b, gamma, beta, feats, rows);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_coor_head(const float* X, const float* W, const float* b, float* out, int64_t rows,
                                void* stream) {
    SCREAM_REQUIRE(X && W && b && out, SCREAM_EINVAL);
    SCREAM_REQUIRE(rows >= 0, SCREAM_EINVAL);
    if (rows == 0) return 0;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;  // grid-stride: 256 CUs x 8 blocks
    coor_head_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(X, W, b, out, rows);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
