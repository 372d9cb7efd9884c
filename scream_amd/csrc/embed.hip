// A1: sine position embedding + 1x1-conv embedding + pre_norm, and the 256->3 head of coor_mlp (A6).
//
// Both kernels are row-streaming (HBM-bound): one wavefront owns one 256-wide feature row, so every
// global access of a wave is a full, contiguous 256 B / 1 KiB segment.
#include "common.h"

namespace {

constexpr int D = SCREAM_D_MODEL;
constexpr int NPF = 84;  // num_pos_feats = 256 // 3 // 2 * 2 (models/transformer.py:148)

// sin and cos of one argument for the position embedding: three-term Cody-Waite reduction by pi / 2 (the first two constants are
// short enough for q times them to be exact for |q| < 2^13) and the Cephes single-precision kernels on [-pi/4, pi/4]: absolute
// error <= 1e-7 for |p| <= 8192 and <= 1.1 ulp for |p| <= 100 (normalised coordinates give |p| <= 2 pi; checked on the host
// against float64 over 6 x 10^6 arguments and on the device by test_pe_sine_embedding_against_float64) in ~26 vector
// instructions, where sincosf spends three times that on a reduction that must also cover |p| ~ 1e38.  Larger arguments,
// inf and NaN take sincosf.
__device__ __forceinline__ void sincos_pe(float p, float* sn, float* cs) {
    if (!(fabsf(p) <= 8192.0f)) {  // (also NaN / inf)
        sincosf(p, sn, cs);
        return;
    }
    const float q = rintf(p * 0.636619772367581343f);                       // nearest multiple of pi / 2
    float r = __builtin_fmaf(q, -1.5703125f, p);                             // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8 + ...
    r = __builtin_fmaf(q, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(q, -7.54978995489188e-8f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                                    __builtin_fmaf(-0.5f, z, 1.0f));
    const int n = (int)q;
    const float s0 = (n & 1) ? pc : ps, c0 = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}

// feats[row] = LN(pe(xyz[row]) + W_e (xyz[row] - center[cloud]) + b_e); models/pointnet.py:45-48.
// Feature f < 252: axis a = f / 84, i = f % 84, value sin(p) for even i, cos(p) for odd i with
// p = (x_a * 2 pi) / dim_t[i] (models/transformer.py:172-176); features 252..255 are the zero pad (:179).
// One block = one 32-row group, eight rows per wave.  Lane l owns the FOUR consecutive features 4 l .. 4 l + 3 (round 4): dim_t[2 j] ==
// dim_t[2 j + 1] (transformer.py:168-170), so the features 2 j, 2 j + 1 are the sine and the cosine of the SAME argument -- two
// sincosf per lane and row where a lane that owned features 64 apart evaluated four sinf AND four cosf (a per-lane choice between
// two calls runs both) behind four divisions: the kernel was bound by its ~440 vector instructions per row, 0.26 of the HBM rate
// for 1 KB written per row; 84 = 4 x 21 puts the axis boundaries on lane boundaries (lanes 0-20 | 21-41 | 42-62 | lane 63 = the pad).
// FRAG: the group goes out FRAGMENT-major (SCREAM_ACT_FRAG, include/scream_hip.h) through an LDS tile -- the layout the projection
// and the layer tail read; the values are the row-major kernel's, bit for bit.
template <bool FRAG>
__global__ __launch_bounds__(256) void pe_embed_ln_kernel(const float* __restrict__ xyz,
                                                         const int32_t* __restrict__ tile_cloud,
                                                         const float* __restrict__ center,
                                                         const float* __restrict__ dim_t,
                                                         const float* __restrict__ emb_w,
                                                         const float* __restrict__ emb_b,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ feats) {
    constexpr int LDT = D + 4;  // row stride of the tile: 16-byte aligned, rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float tile[FRAG ? 32 * LDT : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int cloud = tile_cloud[row0 / SCREAM_ROW_TILE];
    const float c[3] = {center[cloud * 3 + 0], center[cloud * 3 + 1], center[cloud * 3 + 2]};
    const float two_pi = 6.283185307179586f;  // fp32(1.0 * 2 * math.pi), transformer.py:155,171
    const int f0 = 4 * lane;
    const int ax = lane < 63 ? lane / 21 : -1;       // axis of the lane's four features (-1: the zero pad)
    const int i0 = ax >= 0 ? f0 - ax * NPF : 0;      // even: (i0, i0 + 1) and (i0 + 2, i0 + 3) are (sin, cos) pairs
    const float dt0 = dim_t[i0], dt1 = dim_t[i0 + 2];
    float ew[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) ew[k][a] = emb_w[(f0 + k) * 3 + a];
    const f32x4 eb = *reinterpret_cast<const f32x4*>(emb_b + f0), g = *reinterpret_cast<const f32x4*>(gamma + f0),
                be = *reinterpret_cast<const f32x4*>(beta + f0);
    for (int rr = 0; rr < 8; ++rr) {
        const int rl = wave * 8 + rr;
        const int64_t row = row0 + rl;
        const float x[3] = {xyz[row * 3 + 0], xyz[row * 3 + 1], xyz[row * 3 + 2]};
        const float xe[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
        float pe[4] = {0.f, 0.f, 0.f, 0.f};
        if (ax >= 0) {
            const float xa = (ax == 0 ? x[0] : ax == 1 ? x[1] : x[2]) * two_pi;
            sincos_pe(xa / dt0, &pe[0], &pe[1]);
            sincos_pe(xa / dt1, &pe[2], &pe[3]);
        }
        float v[4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = ew[k][0] * xe[0] + ew[k][1] * xe[1] + ew[k][2] * xe[2] + eb[k];
            v[k] = pe[k] + e;
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
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (v[k] - mean) * rstd * g[k] + be[k];
        if (FRAG) *reinterpret_cast<f32x4*>(&tile[rl * LDT + f0]) = o;
        else *reinterpret_cast<f32x4*>(feats + row * D + f0) = o;  // 1 KiB per wave instruction
    }
    if (FRAG) {
        __syncthreads();
        // float4 q of the group = [segment of 8 features q >> 6][lane (r, half) = q & 63]: features 8 seg + 4 half .. + 3 of row r
        float* out = feats + row0 * D;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = threadIdx.x + 256 * i;
            const int seg = q >> 6, l = q & 63, r = l & 31, half = l >> 5;
            *reinterpret_cast<f32x4*>(out + q * 4) = *reinterpret_cast<const f32x4*>(&tile[r * LDT + seg * 8 + 4 * half]);
        }
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

namespace {
int pe_embed_ln_launch(const float* xyz, const int32_t* tile_cloud, const float* center, const float* dim_t, const float* emb_w,
                       const float* emb_b, const float* gamma, const float* beta, float* feats, int64_t rows, bool frag,
                       void* stream) {
    SCREAM_REQUIRE(xyz && tile_cloud && center && dim_t && emb_w && emb_b && gamma && beta && feats, SCREAM_EINVAL);
    SCREAM_REQUIRE(rows >= 0 && rows % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(feats) & 15) == 0, SCREAM_EINVAL);
    if (rows == 0) return 0;
    const int64_t blocks = rows / 32;
    SCREAM_REQUIRE(blocks < (1ll << 31), SCREAM_EUNSUPPORTED);
    if (frag)
        pe_embed_ln_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(xyz, tile_cloud, center, dim_t, emb_w, emb_b,
                                                                                           gamma, beta, feats);
    else
        pe_embed_ln_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(xyz, tile_cloud, center, dim_t, emb_w, emb_b,
                                                                                            gamma, beta, feats);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int scream_pe_embed_ln(const float* xyz, const int32_t* tile_cloud, const float* center,
                                  const float* dim_t, const float* emb_w, const float* emb_b, const float* gamma,
                                  const float* beta, float* feats, int64_t rows, void* stream) {
    return pe_embed_ln_launch(xyz, tile_cloud, center, dim_t, emb_w, emb_b, gamma, beta, feats, rows, false, stream);
}

extern "C" int scream_pe_embed_ln_frag(const float* xyz, const int32_t* tile_cloud, const float* center,
                                       const float* dim_t, const float* emb_w, const float* emb_b, const float* gamma,
                                       const float* beta, float* feats, int64_t rows, void* stream) {
    return pe_embed_ln_launch(xyz, tile_cloud, center, dim_t, emb_w, emb_b, gamma, beta, feats, rows, true, stream);
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
