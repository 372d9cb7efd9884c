// A8 + A9 + A10: correspondence gather fused into the weighted Kabsch / Procrustes solve, and RE/TE.
//
// Reference: evaluate_3d_match.py:96-101 (gather), utils.py:138-178 (rigid_transform_3d; the reference
// builds a dense K x K diag_embed and runs the 3x3 SVD on the HOST via H.cpu()), utils.py:181-189.
// Here one workgroup owns one pair: two streaming passes over the correspondences (centroids, then
// the 3x3 covariance H), sums carried in fp64 and reduced wave -> workgroup deterministically, then
// the 3x3 SVD as a one-sided (Hestenes) Jacobi sweep held entirely in registers, evaluated redundantly by
// every lane of wave 0 (the data is wave-uniform; no LDS, no divergence).  The rotation
// R = V diag(1,1,det(V U^T)) U^T is invariant to the sign/order ambiguities of the SVD whenever the
// reference's own result is well defined, so parity is checked on R|t, not on U, S, V.
// Compiled with -ffp-contract=off: the fp32 steps (x / s + c, x - centroid, sum / (K + 1e-6)) are the
// reference's individually rounded operations.
#include <vector>

#include <stdlib.h>

#include "common.h"
#include "icp_grid.h"

namespace {

struct D3 {
    double x, y, z;
};

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Columns g[c][0..2] of G = H V and v[c][0..2] of V.  After convergence G's columns are orthogonal:
// H = U S V^T with s_c = |g_c|, u_c = g_c / s_c.
__device__ void jacobi_svd3(const double H[3][3], double U[3][3], double V[3][3], double sig[3]) {
    double g[3][3], v[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            g[c][i] = H[i][c];
            v[c][i] = (i == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            const double alpha = dot3(g[p], g[p]), beta = dot3(g[q], g[q]), gamma = dot3(g[p], g[q]);
            const double lim = 1e-30 + 1e-16 * sqrt(alpha * beta);
            if (fabs(gamma) > lim) {
                off += fabs(gamma);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double gp = g[p][i], gq = g[q][i];
                    g[p][i] = cs * gp - sn * gq;
                    g[q][i] = sn * gp + cs * gq;
                    const double vp = v[p][i], vq = v[q][i];
                    v[p][i] = cs * vp - sn * vq;
                    v[q][i] = sn * vp + cs * vq;
                }
            }
        }
        if (off == 0.0) break;
    }
    double s[3] = {sqrt(dot3(g[0], g[0])), sqrt(dot3(g[1], g[1])), sqrt(dot3(g[2], g[2]))};
    // sort columns by descending singular value (LAPACK order; the det fix applies to the smallest)
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int a = (pass == 1) ? 1 : 0, b = a + 1;
        if (s[a] < s[b]) {
            const double ts = s[a];
            s[a] = s[b];
            s[b] = ts;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double tg = g[a][i];
                g[a][i] = g[b][i];
                g[b][i] = tg;
                tg = v[a][i];
                v[a][i] = v[b][i];
                v[b][i] = tg;
            }
        }
    }
    double u[3][3];
    if (!(s[0] > 1e-300)) {
        // H == 0 (no correspondences): torch.svd returns U = V = I -> identity transform (utils.py:155-175)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                u[c][i] = (i == c) ? 1.0 : 0.0;
                v[c][i] = (i == c) ? 1.0 : 0.0;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) u[0][i] = g[0][i] / s[0];
        if (s[1] > 1e-14 * s[0]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) u[1][i] = g[1][i] / s[1];
        } else {
            // rank 1: any unit vector orthogonal to u0 (rotation about u0 is undetermined in the reference too)
            const int k = (fabs(u[0][0]) <= fabs(u[0][1]) && fabs(u[0][0]) <= fabs(u[0][2])) ? 0
                          : (fabs(u[0][1]) <= fabs(u[0][2]) ? 1 : 2);
            double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
            const double pr = dot3(e, u[0]);
            double w[3] = {e[0] - pr * u[0][0], e[1] - pr * u[0][1], e[2] - pr * u[0][2]};
            const double nw = sqrt(dot3(w, w));
#pragma unroll
            for (int i = 0; i < 3; ++i) u[1][i] = w[i] / nw;
        }
        if (s[2] > 1e-14 * s[0]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) u[2][i] = g[2][i] / s[2];
        } else {  // rank <= 2: complete the basis; the sign cancels against det(V U^T) below
            u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
            u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
            u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sig[c] = s[c];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            U[i][c] = u[c][i];
            V[i][c] = v[c][i];
        }
    }
}

__device__ __forceinline__ double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// R = V diag(1,1,det(V U^T)) U^T; t = cB - R cA (utils.py:169-175).  Writes a row-major 4x4.
__device__ void solve_pose(const double H[3][3], const float cA[3], const float cB[3], float* T) {
    double U[3][3], V[3][3], sig[3];
    jacobi_svd3(H, U, V, sig);
    const double delta = det3(V) * det3(U);  // det(V U^T)
    float R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[i][j] = (float)(V[i][0] * U[j][0] + V[i][1] * U[j][1] + delta * V[i][2] * U[j][2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double rc = (double)R[i][0] * cA[0] + (double)R[i][1] * cA[1] + (double)R[i][2] * cA[2];
        T[i * 4 + 0] = R[i][0];
        T[i * 4 + 1] = R[i][1];
        T[i * 4 + 2] = R[i][2];
        T[i * 4 + 3] = (float)((double)cB[i] - rc);
    }
    T[12] = 0.f;
    T[13] = 0.f;
    T[14] = 0.f;
    T[15] = 1.f;
}

// Workgroup-wide sum of NV doubles per thread; result broadcast to every thread.  NT threads (256: the A9 solve; 1024:
// the ICP update), combined in a fixed order -- waves pairwise, then groups of four -- so the sum depends on NT and on
// nothing else.
template <int NV, int NT = 256>
__device__ void block_sum(double (&v)[NV], double* red /* [NT / 64][NV] */) {
    constexpr int NW = NT / 64;
    static_assert(NW == 4 || NW == 8 || NW == 16, "");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_f64(v[k]);
    __syncthreads();  // red may still be read from a previous call
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double q4[NW / 4];
#pragma unroll
        for (int g = 0; g < NW / 4; ++g)
            q4[g] = (red[(4 * g + 0) * NV + k] + red[(4 * g + 1) * NV + k]) + (red[(4 * g + 2) * NV + k] + red[(4 * g + 3) * NV + k]);
        v[k] = NW == 4 ? q4[0] : NW == 8 ? q4[0] + q4[1 % (NW / 4)] : (q4[0] + q4[1 % (NW / 4)]) + (q4[2 % (NW / 4)] + q4[3 % (NW / 4)]);
    }
}

struct CorrFetch {  // evaluate_3d_match.py:96-101
    const float* src;
    const float* ref;
    const int32_t* idx;
    const uint8_t* valid;
    int64_t src_row0, ref_row0;
    float s, c0, c1, c2;
    int n;
    __device__ __forceinline__ bool get(int i, float a[3], float b[3], float& w) const {
        const int64_t row = src_row0 + i;
        if (!valid[row]) return false;
        const int64_t rrow = ref_row0 + (idx ? (int64_t)idx[row] : (int64_t)i);
        a[0] = src[row * 3 + 0] / s + c0;
        a[1] = src[row * 3 + 1] / s + c1;
        a[2] = src[row * 3 + 2] / s + c2;
        b[0] = ref[rrow * 3 + 0] / s + c0;
        b[1] = ref[rrow * 3 + 1] / s + c1;
        b[2] = ref[rrow * 3 + 2] / s + c2;
        w = 1.0f;
        return true;
    }
};

struct DenseFetch {  // utils.py:138-151
    const float* A;
    const float* B;
    const float* w;
    float thr;
    int n;
    __device__ __forceinline__ bool get(int i, float a[3], float b[3], float& wt) const {
        wt = w ? w[i] : 1.0f;
        if (wt < thr) wt = 0.f;  // weights[weights < weight_threshold] = 0
        if (wt == 0.f) return false;
        a[0] = A[i * 3 + 0];
        a[1] = A[i * 3 + 1];
        a[2] = A[i * 3 + 2];
        b[0] = B[i * 3 + 0];
        b[1] = B[i * 3 + 1];
        b[2] = B[i * 3 + 2];
        return true;
    }
};

template <class Fetch, int NT = 256>
__device__ void kabsch_block(const Fetch& f, float* T_out, int32_t* n_corr_out) {
    __shared__ double red[NT / 64 * 9];
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < f.n; i += NT) {
        float a[3], b[3], w;
        if (f.get(i, a, b, w)) {
            acc[0] += (double)(a[0] * w);
            acc[1] += (double)(a[1] * w);
            acc[2] += (double)(a[2] * w);
            acc[3] += (double)(b[0] * w);
            acc[4] += (double)(b[1] * w);
            acc[5] += (double)(b[2] * w);
            acc[6] += (double)w;
            acc[7] += 1.0;
        }
    }
    block_sum<9, NT>(acc, red);
    const float denom = (float)acc[6] + 1e-6f;  // utils.py:155-158
    const float cA[3] = {(float)acc[0] / denom, (float)acc[1] / denom, (float)acc[2] / denom};
    const float cB[3] = {(float)acc[3] / denom, (float)acc[4] / denom, (float)acc[5] / denom};
    const int count = (int)acc[7];

    double h[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = 0.0;
    for (int i = threadIdx.x; i < f.n; i += NT) {
        float a[3], b[3], w;
        if (f.get(i, a, b, w)) {
            const float am[3] = {a[0] - cA[0], a[1] - cA[1], a[2] - cA[2]};
            const float bm[3] = {(b[0] - cB[0]) * w, (b[1] - cB[1]) * w, (b[2] - cB[2]) * w};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) h[r * 3 + c] += (double)am[r] * (double)bm[c];
        }
    }
    block_sum<9, NT>(h, red);
    if (threadIdx.x < 64) {  // wave 0, every lane redundantly (wave-uniform data)
        double H[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[r][c] = (double)(float)h[r * 3 + c];  // the reference holds H in fp32
        float T[16];
        solve_pose(H, cA, cB, T);
        if (threadIdx.x < 16) T_out[threadIdx.x] = T[threadIdx.x];
        if (threadIdx.x == 0 && n_corr_out) *n_corr_out = count;
    }
}

__global__ __launch_bounds__(256) void kabsch_corr_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                         const int32_t* __restrict__ src_row0,
                                                         const int32_t* __restrict__ src_len,
                                                         const int32_t* __restrict__ ref_row0,
                                                         const int32_t* __restrict__ idx,
                                                         const uint8_t* __restrict__ valid,
                                                         const float* __restrict__ s, const float* __restrict__ c,
                                                         float* __restrict__ T_out, int32_t* __restrict__ n_corr) {
    const int p = blockIdx.x;
    CorrFetch f{src, ref, idx, valid, src_row0[p], ref_row0[p], s[p], c[p * 3 + 0], c[p * 3 + 1], c[p * 3 + 2],
                src_len[p]};
    kabsch_block(f, T_out + p * 16, n_corr ? n_corr + p : nullptr);
}

__global__ __launch_bounds__(256) void kabsch_dense_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ w, float thr, int K,
                                                          float* __restrict__ T_out) {
    const int p = blockIdx.x;
    DenseFetch f{A + (int64_t)p * K * 3, B + (int64_t)p * K * 3, w ? w + (int64_t)p * K : nullptr, thr, K};
    kabsch_block(f, T_out + p * 16, nullptr);
}

// utils.py:181-189 with the fp32 operation order of torch-CPU (3-term fma chains for the 3x3 product).
__global__ __launch_bounds__(64) void transformation_error_kernel(const float* __restrict__ Tp,
                                                                 const float* __restrict__ Tg, int n,
                                                                 float* __restrict__ re, float* __restrict__ te) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* P = Tp + i * 16;
    const float* G = Tg + i * 16;
    float diag[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float d = P[0 * 4 + j] * G[0 * 4 + j];
        d = __fmaf_rn(P[1 * 4 + j], G[1 * 4 + j], d);
        d = __fmaf_rn(P[2 * 4 + j], G[2 * 4 + j], d);
        diag[j] = d;
    }
    const float tr = (diag[0] + diag[1]) + diag[2];
    float x = (tr - 1.0f) / 2.0f;
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    re[i] = (acosf(x) * 180.0f) / 3.14159265358979323846f;
    const float dx = P[3] - G[3], dy = P[7] - G[7], dz = P[11] - G[11];
    te[i] = sqrtf((dx * dx + dy * dy) + dz * dz);
}

// ------------------------------------------------------------------------------------------------
// Point-to-point ICP refinement (the "next" row of SURVEY.md 8f: evaluate_3d_match.py:106-119 calls
// open3d.registration_icp, which is not in this image -> parity with open3d itself is UNPINNED; the loop
// below follows open3d's published RegistrationICP: evaluate correspondences of the transformed source
// (nearest target within max_dist), estimate the rigid update from them (Kabsch, no scaling),
// compose, re-evaluate, stop when |d fitness| and |d inlier_rmse| both fall below their thresholds
// or after max_iter updates).  It is exactly loop{ A7 with a radius, A8, A9 } in metric space, so it reuses
// the search and solve kernels; all pairs of a batch iterate together and converged pairs freeze on the
// device (no host synchronisation inside the loop).

// loss[p] = mean_n sum_xyz |pred_n - (R_p a_n + t_p)| (models/pointnet.py:93-99, the L1 point loss the evaluators report per pair).
// One workgroup per pair; the terms in fp32 like the reference's, their sum in fp64 in a fixed order (the torch expression costs
// seven small launches per pair).
__global__ __launch_bounds__(256) void point_loss_kernel(const float* __restrict__ pred, const float* __restrict__ src,
                                                        const int32_t* __restrict__ row0, const int32_t* __restrict__ len,
                                                        const float* __restrict__ R, const float* __restrict__ t,
                                                        float* __restrict__ out) {
    __shared__ double red[4];
    const int p = blockIdx.x, n = len[p];
    const int64_t r0 = row0[p];
    const float* Rp = R + p * 9;
    const float* tp = t + p * 3;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = src[(r0 + i) * 3 + 0], y = src[(r0 + i) * 3 + 1], z = src[(r0 + i) * 3 + 2];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float reg = ((Rp[k * 3 + 0] * x + Rp[k * 3 + 1] * y) + Rp[k * 3 + 2] * z) + tp[k];
            s += fabsf(pred[(r0 + i) * 3 + k] - reg);
        }
        acc[0] += (double)s;
    }
    block_sum<1, 256>(acc, red);
    if (threadIdx.x == 0) out[p] = n > 0 ? (float)(acc[0] / n) : 0.f;
}

// metric frame: x / s + c (evaluate_3d_match.py:106-107)
__global__ __launch_bounds__(256) void icp_to_metric_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ row0,
                                                           const int32_t* __restrict__ len, const float* __restrict__ s,
                                                           const float* __restrict__ c, float* __restrict__ out) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len[p]) return;
    const int64_t row = (int64_t)row0[p] + i;
    out[row * 3 + 0] = xyz[row * 3 + 0] / s[p] + c[p * 3 + 0];
    out[row * 3 + 1] = xyz[row * 3 + 1] / s[p] + c[p * 3 + 1];
    out[row * 3 + 2] = xyz[row * 3 + 2] / s[p] + c[p * 3 + 2];
}

// q = R x + t with the pair's current transform
__global__ __launch_bounds__(256) void icp_transform_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ row0,
                                                           const int32_t* __restrict__ len, const float* __restrict__ T,
                                                           float* __restrict__ out) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len[p]) return;
    const int64_t row = (int64_t)row0[p] + i;
    const float* t = T + p * 16;
    const float x = xyz[row * 3 + 0], y = xyz[row * 3 + 1], z = xyz[row * 3 + 2];
    out[row * 3 + 0] = t[0] * x + t[1] * y + t[2] * z + t[3];
    out[row * 3 + 1] = t[4] * x + t[5] * y + t[6] * z + t[7];
    out[row * 3 + 2] = t[8] * x + t[9] * y + t[10] * z + t[11];
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, float v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- the ICP iteration (round 3, second version): ONE launch per iteration --------------------------------------------------
// A search launch leaves, per 256-point chunk of a pair, a PARTIAL of its correspondences: count, sum d^2, sum a, sum b, sum a b^T
// (17 doubles; a = the transformed source point, b = its target).  The next launch starts, in EVERY block of the pair, by summing
// the pair's partials in a fixed order and deriving from them what rounds 1-3's separate update launch derived from a second
// gathering pass: fitness / inlier RMSE, the convergence test, and the Kabsch update composed into T -- then searches under the
// new T.  One dependent launch per iteration instead of two (rounds 1-2: three), no gather of idx -> target rows at all.
// The covariance about the fp32 centroids cA, cB (utils.py:155-166) is expanded from the one-pass sums in fp64:
//   sum (a - cA)(b - cB)^T = sum a b^T - cA (sum b)^T - (sum a) cB^T + n cA cB^T
// (the two-pass form rounds the differences to fp32 first: 1e-7 relative apart, both far inside the 1e-4 of the ICP tests; the
// A9 solve of the parity path, kabsch_block above, is untouched).  Buffers indexed by the parity of the launch are written by the
// pair's block 0 and read by every block of the NEXT launch, so no block ever reads what another block of its own launch writes.
struct IcpState {  // per pair: fitness and inlier RMSE of the last evaluated search
    float fitness, rmse;
};

constexpr int ICP_NP = 17;  // doubles per chunk partial
constexpr int ICP_NG = 15;  // strided groups the partials of a pair are summed in (15 x 17 = 255 threads)

struct IcpArgs {
    const int32_t* src_row0;
    const int32_t* src_len;
    float* Tbuf;         // [2][n_pairs][16]: T of search it lives at parity it & 1
    IcpState* state;     // [2][n_pairs]: evaluation of search e at parity e & 1
    double* part;        // [2][n_chunks_total][ICP_NP]: partials of search it at parity it & 1; chunk c of pair p = chunk0[p] + c
    const int32_t* chunk0;  // [n_pairs] exclusive prefix sum of icp_chunks(src_len[p]) (icp_chunk0_kernel): collision-free for ANY order / overlap of the clouds
    int32_t* done;       // [n_pairs] 0 -> 1, once
    int32_t* act_len;    // [n_pairs] src_len, 0 once done (the brute-force yardstick's kernels take their work from it)
    int64_t part_stride; // n_chunks_total * ICP_NP
    int32_t n_pairs, max_iter;
    float rel_fitness, rel_rmse;
    float* T_out;        // [n_pairs][16] the caller's transforms
    float* fit_rmse_out; // may be NULL
    int32_t* iters_out;  // may be NULL
};

__device__ __forceinline__ int icp_chunks(int n) { return n > 0 ? (n + 255) / 256 : 1; }  // (an empty source still has its block 0)
__device__ __forceinline__ int64_t icp_chunk0(const IcpArgs& a, int p) { return a.chunk0[p]; }

// chunk0[p] = sum_{q < p} icp_chunks(src_len[q]); one thread (n_pairs is a batch size).  Until round 3 the slots were derived from
// src_row0[p] / 256 + p, which is collision-free only for clouds packed in ascending, non-overlapping order -- true for PackedBatch,
// not promised by the C ABI.
__global__ void icp_chunk0_kernel(const int32_t* __restrict__ src_len, int32_t n_pairs, int32_t* __restrict__ chunk0) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int32_t acc = 0;
    for (int p = 0; p < n_pairs; ++p) {
        chunk0[p] = acc;
        acc += icp_chunks(src_len[p]);
    }
}

// Whole block (256 threads).  Evaluates search e = it - 1 of pair p from its partials; returns (block-uniform) whether the pair
// stops.  If it goes on, T_sh (LDS) holds T_it = dT . T_e.  `writer`: this block publishes state / done / outputs / T_it.
__device__ bool icp_pose_step(const IcpArgs& a, int p, int it, bool writer, float* T_sh) {
    __shared__ double grp[ICP_NG][ICP_NP];
    __shared__ double tot[ICP_NP];
    __shared__ float dT_sh[16];
    __shared__ int stop_sh;
    const int e = it - 1, n = a.src_len[p], nb = icp_chunks(n);
    const double* part = a.part + (int64_t)(e & 1) * a.part_stride + icp_chunk0(a, p) * ICP_NP;
    const float* T_e = a.Tbuf + ((int64_t)(e & 1) * a.n_pairs + p) * 16;
    const int tid = threadIdx.x;
    if (tid < ICP_NG * ICP_NP) {
        const int k = tid % ICP_NP, g = tid / ICP_NP;
        double sum = 0.0;
        for (int c = g; c < nb; c += ICP_NG) sum += part[(int64_t)c * ICP_NP + k];
        grp[g][k] = sum;
    }
    __syncthreads();
    if (tid < ICP_NP) {
        double sum = grp[0][tid];
#pragma unroll
        for (int g = 1; g < ICP_NG; ++g) sum += grp[g][tid];
        tot[tid] = sum;
    }
    __syncthreads();
    if (tid < 64) {  // wave 0, every lane redundantly (wave-uniform data)
        const double cnt = tot[0];
        const float fitness = n > 0 ? (float)(cnt / n) : 0.f;
        const float rmse = cnt > 0 ? (float)sqrt(tot[1] / cnt) : 0.f;
        const IcpState prev = a.state[(int64_t)((e + 1) & 1) * a.n_pairs + p];  // evaluation e - 1 (unused at e == 0)
        const bool converged = e > 0 && fabsf(prev.fitness - fitness) < a.rel_fitness && fabsf(prev.rmse - rmse) < a.rel_rmse;
        const bool stop = converged || e >= a.max_iter;
        if (tid == 0) {
            stop_sh = stop ? 1 : 0;
            if (writer) {
                const IcpState o = {fitness, rmse};
                a.state[(int64_t)(e & 1) * a.n_pairs + p] = o;
                if (a.fit_rmse_out) {
                    a.fit_rmse_out[2 * p + 0] = fitness;
                    a.fit_rmse_out[2 * p + 1] = rmse;
                }
                if (a.iters_out) a.iters_out[p] = e;  // number of updates applied
                if (stop) {
                    a.done[p] = 1;
                    a.act_len[p] = 0;
                }
            }
        }
        if (stop) {
            if (writer && tid < 16) a.T_out[p * 16 + tid] = T_e[tid];
        } else {
            const float denom = (float)cnt + 1e-6f;  // utils.py:155-158 with unit weights
            const float cA[3] = {(float)tot[2] / denom, (float)tot[3] / denom, (float)tot[4] / denom};
            const float cB[3] = {(float)tot[5] / denom, (float)tot[6] / denom, (float)tot[7] / denom};
            double H[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double h = tot[8 + r * 3 + c] - (double)cA[r] * tot[5 + c] - tot[2 + r] * (double)cB[c] + cnt * (double)cA[r] * (double)cB[c];
                    H[r][c] = (double)(float)h;  // the reference holds H in fp32
                }
            float dT[16];
            solve_pose(H, cA, cB, dT);
            if (tid < 16) dT_sh[tid] = dT[tid];
        }
    }
    __syncthreads();
    const bool stop = stop_sh != 0;
    if (!stop) {
        if (tid < 16) {  // T_it = dT . T_e
            const int i = tid >> 2, j = tid & 3;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) v += dT_sh[i * 4 + k] * T_e[k * 4 + j];
            T_sh[tid] = v;
            if (writer) a.Tbuf[((int64_t)(it & 1) * a.n_pairs + p) * 16 + tid] = v;
        }
        __syncthreads();
    }
    return stop;
}

// the block's partial of search `it`: one correspondence (or none) per thread, summed over the 256 threads in a fixed order
__device__ void icp_store_partial(const IcpArgs& a, int p, int c, int it, bool ok, float ax, float ay, float az, const float (&b)[3],
                                  float d) {
    __shared__ double red[4 * ICP_NP];
    double v[ICP_NP];
#pragma unroll
    for (int k = 0; k < ICP_NP; ++k) v[k] = 0.0;
    if (ok) {
        const float av[3] = {ax, ay, az};
        v[0] = 1.0;
        v[1] = (double)d;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[2 + k] = (double)av[k];
            v[5 + k] = (double)b[k];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) v[8 + r * 3 + cc] = (double)av[r] * (double)b[cc];
    }
    block_sum<ICP_NP, 256>(v, red);
    if (threadIdx.x < ICP_NP) {
        double out = v[0];
#pragma unroll
        for (int k = 1; k < ICP_NP; ++k)
            if ((int)threadIdx.x == k) out = v[k];
        a.part[(int64_t)(it & 1) * a.part_stride + (icp_chunk0(a, p) + c) * ICP_NP + threadIdx.x] = out;
    }
}

// grid (ceil(max_src_len / 256), n_pairs): [evaluate search it - 1, update T] + search it on the target grid + partial
__global__ __launch_bounds__(256) void icp_iter_kernel(IcpArgs a, const float* __restrict__ src_m, const int32_t* __restrict__ r_row0,
                                                      const scream_internal::GridParam* __restrict__ gp,
                                                      const int32_t* __restrict__ start, const float* __restrict__ sorted_prep,
                                                      const int32_t* __restrict__ sorted_idx, float thresh, int it) {
    __shared__ float T_sh[16];
    const int p = blockIdx.y, c = blockIdx.x;
    // (block 0 of this very launch may be setting the flag: a block that sees it early returns where it would have returned
    // after deriving the same stop -- the derivation is deterministic)
    if (a.done[p]) return;
    const int n = a.src_len[p];
    if (c >= icp_chunks(n)) return;
    if (it > 0) {
        if (icp_pose_step(a, p, it, c == 0, T_sh)) return;
    } else {
        if (threadIdx.x < 16) T_sh[threadIdx.x] = a.Tbuf[(int64_t)p * 16 + threadIdx.x];
        __syncthreads();
    }
    const int i = c * 256 + threadIdx.x;
    const bool in = i < n;
    float ax = 0.f, ay = 0.f, az = 0.f, d = 0.f, b[3] = {0.f, 0.f, 0.f};
    uint8_t ok = 0;
    if (in) {
        const int64_t row = (int64_t)a.src_row0[p] + i;
        const float x = src_m[row * 3 + 0], y = src_m[row * 3 + 1], z = src_m[row * 3 + 2];
        const float* t = T_sh;  // (the arithmetic of icp_transform_kernel)
        ax = t[0] * x + t[1] * y + t[2] * z + t[3];
        ay = t[4] * x + t[5] * y + t[6] * z + t[7];
        az = t[8] * x + t[9] * y + t[10] * z + t[11];
        int32_t bi;
        scream_internal::grid_search_point(gp[p], start + (int64_t)p * (ICP_GRID_CELLS + 1), sorted_prep + (int64_t)r_row0[p] * 4,
                                           sorted_idx + r_row0[p], ax, ay, az, thresh, bi, d, ok, b);
    }
    icp_store_partial(a, p, c, it, ok != 0, ax, ay, az, b, d);
}

// one block per pair: the evaluation (+ update) alone -- the last search's, and every iteration's on the brute-force yardstick
__global__ __launch_bounds__(256) void icp_pose_kernel(IcpArgs a, int it) {
    __shared__ float T_sh[16];
    const int p = blockIdx.x;
    if (a.done[p]) return;
    icp_pose_step(a, p, it, true, T_sh);
}

// SCREAM_ICP_BRUTE=1: the partials of a search done by icp_transform_kernel + scream_nn_search (same per-thread values, same sums)
__global__ __launch_bounds__(256) void icp_partials_kernel(IcpArgs a, const float* __restrict__ q, const float* __restrict__ ref,
                                                          const int32_t* __restrict__ ref_row0, const int32_t* __restrict__ idx,
                                                          const uint8_t* __restrict__ valid, const float* __restrict__ dmin, int it) {
    const int p = blockIdx.y, c = blockIdx.x;
    if (a.done[p]) return;
    const int n = a.src_len[p];
    if (c >= icp_chunks(n)) return;
    const int i = c * 256 + threadIdx.x;
    float ax = 0.f, ay = 0.f, az = 0.f, d = 0.f, b[3] = {0.f, 0.f, 0.f};
    bool ok = false;
    if (i < n) {
        const int64_t row = (int64_t)a.src_row0[p] + i;
        ok = valid[row] != 0;
        if (ok) {
            const int64_t rrow = (int64_t)ref_row0[p] + idx[row];
            ax = q[row * 3 + 0];
            ay = q[row * 3 + 1];
            az = q[row * 3 + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k) b[k] = ref[rrow * 3 + k];
            d = dmin[row];
        }
    }
    icp_store_partial(a, p, c, it, ok, ax, ay, az, b, d);
}

}  // namespace

extern "C" int scream_kabsch_corr(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                                  const int32_t* ref_row0, const int32_t* idx, const uint8_t* valid, const float* s,
                                  const float* c, int32_t n_pairs, float* T_out, int32_t* n_corr, void* stream) {
    SCREAM_REQUIRE(src && ref && src_row0 && src_len && ref_row0 && valid && s && c && T_out, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_pairs >= 0, SCREAM_EINVAL);
    if (n_pairs == 0) return 0;
    kabsch_corr_kernel<<<dim3(n_pairs), dim3(256), 0, as_stream(stream)>>>(src, ref, src_row0, src_len, ref_row0, idx,
                                                                           valid, s, c, T_out, n_corr);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_rigid_transform_3d(const float* A, const float* B, const float* w, float weight_threshold,
                                         int32_t bs, int32_t K, float* T_out, void* stream) {
    SCREAM_REQUIRE(T_out && bs >= 0 && K >= 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(K == 0 || (A && B), SCREAM_EINVAL);
    if (bs == 0) return 0;
    kabsch_dense_kernel<<<dim3(bs), dim3(256), 0, as_stream(stream)>>>(A, B, w, weight_threshold, K, T_out);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_point_loss(const float* src_pred, const float* src, const int32_t* src_row0, const int32_t* src_len,
                                 const float* rot, const float* trans, int32_t n_pairs, float* loss, void* stream) {
    SCREAM_REQUIRE(src_pred && src && src_row0 && src_len && rot && trans && loss && n_pairs >= 0, SCREAM_EINVAL);
    if (n_pairs == 0) return 0;
    point_loss_kernel<<<dim3(n_pairs), dim3(256), 0, as_stream(stream)>>>(src_pred, src, src_row0, src_len, rot, trans, loss);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_transformation_error(const float* T_pred, const float* T_gt, int32_t n, float* re, float* te,
                                           void* stream) {
    SCREAM_REQUIRE(T_pred && T_gt && re && te && n >= 0, SCREAM_EINVAL);
    if (n == 0) return 0;
    transformation_error_kernel<<<dim3((n + 63) / 64), dim3(64), 0, as_stream(stream)>>>(T_pred, T_gt, n, re, te);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

namespace {
int64_t icp_chunks_total(int64_t src_rows_total, int32_t n_pairs) { return src_rows_total / 256 + n_pairs + 1; }
}  // namespace

extern "C" int64_t scream_icp_workspace_bytes(int64_t src_rows_total, int64_t ref_rows_total, int32_t n_pairs) {
    if (src_rows_total < 0 || ref_rows_total < 0 || n_pairs < 0) return SCREAM_EINVAL;
    // src metric + transformed src (3 floats each), ref metric (3) + nn ref_prep (4), keys (2), idx, dmin, valid, per pair: ones,
    // T x 2 (32), state x 2 (4), done, act_len; the chunk partials (2 parities x 17 doubles) + the target grid of icp_grid.hip
    return (src_rows_total * (3 + 3 + 2 + 1 + 1 + 1) + ref_rows_total * (3 + 4) + (int64_t)n_pairs * (1 + 32 + 4 + 1 + 1 + 1) + 64 +
            icp_chunks_total(src_rows_total, n_pairs) * (2 * ICP_NP * 2) +
            scream_internal::icp_grid_workspace_floats(ref_rows_total, n_pairs)) * 4 + 16384;
}

extern "C" int scream_icp_p2p_range(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                                    const int32_t* ref_row0, const int32_t* ref_len, const float* s, const float* c,
                                    int32_t n_pairs, int32_t max_src_len, int32_t max_ref_len, int64_t src_rows_total,
                                    int64_t ref_rows_total, float max_corr_dist, int32_t max_iter, float rel_fitness,
                                    float rel_rmse, float* T, float* fitness_rmse, int32_t* iters, int32_t it_begin,
                                    int32_t it_end, int32_t* done_flags, void* workspace, int64_t workspace_bytes, void* stream) {
    SCREAM_REQUIRE(src && ref && src_row0 && src_len && ref_row0 && ref_len && s && c && T && workspace, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_pairs >= 0 && max_iter >= 0 && max_corr_dist > 0.f && it_begin >= 0 && it_end >= it_begin, SCREAM_EINVAL);
    SCREAM_REQUIRE(workspace_bytes >= scream_icp_workspace_bytes(src_rows_total, ref_rows_total, n_pairs), SCREAM_EINVAL);
    if (n_pairs == 0) return 0;
    hipStream_t st = as_stream(stream);
    float* w = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    auto take = [&](int64_t n) { float* r = w; w += (n + 63) / 64 * 64; return r; };
    float* src_m = take(src_rows_total * 3);
    float* q = take(src_rows_total * 3);
    float* ref_m = take(ref_rows_total * 3);
    float* ref_prep = take(ref_rows_total * 4);
    uint64_t* keys = reinterpret_cast<uint64_t*>(take(src_rows_total * 2));
    int32_t* idx = reinterpret_cast<int32_t*>(take(src_rows_total));
    float* dmin = take(src_rows_total);
    uint8_t* valid = reinterpret_cast<uint8_t*>(take((src_rows_total + 3) / 4));
    float* ones = take(n_pairs);
    IcpArgs a{};
    a.src_row0 = src_row0;
    a.src_len = src_len;
    a.Tbuf = take((int64_t)n_pairs * 32);
    a.state = reinterpret_cast<IcpState*>(take((int64_t)n_pairs * 4));
    a.done = reinterpret_cast<int32_t*>(take(n_pairs));
    a.act_len = reinterpret_cast<int32_t*>(take(n_pairs));
    a.part_stride = icp_chunks_total(src_rows_total, n_pairs) * ICP_NP;
    a.part = reinterpret_cast<double*>(take(a.part_stride * 2 * 2));
    int32_t* chunk0 = reinterpret_cast<int32_t*>(take(n_pairs));
    a.chunk0 = chunk0;
    a.n_pairs = n_pairs;
    a.max_iter = max_iter;
    a.rel_fitness = rel_fitness;
    a.rel_rmse = rel_rmse;
    a.T_out = T;
    a.fit_rmse_out = fitness_rmse;
    a.iters_out = iters;
    float* grid_work = take(scream_internal::icp_grid_workspace_floats(ref_rows_total, n_pairs));
    SCREAM_REQUIRE(reinterpret_cast<char*>(w) <= reinterpret_cast<char*>(workspace) + workspace_bytes, SCREAM_EINVAL);
    // SCREAM_ICP_BRUTE=1 (tests): every iteration on the brute-force search of nn_search.hip instead of the target grid
    const char* brute_env = getenv("SCREAM_ICP_BRUTE");
    const bool brute = brute_env && brute_env[0] == '1';

    hipError_t e = hipSuccess;
    scream_internal::IcpGrid grid{};
    if (!brute)  // pointers into grid_work: a pure function of the arguments (every call of a run sees the same ones)
        scream_internal::icp_grid_carve(ref_rows_total, n_pairs, grid_work, &grid);
    if (it_begin == 0) {  // set-up: flags, T_0, metric clouds, the target grid
        e = hipMemsetAsync(a.done, 0, sizeof(int32_t) * n_pairs, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(a.state, 0, sizeof(IcpState) * 2 * n_pairs, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemcpyAsync(a.act_len, src_len, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemcpyAsync(a.Tbuf, T, sizeof(float) * 16 * n_pairs, hipMemcpyDeviceToDevice, st);  // T_0: parity 0
        if (e != hipSuccess) return (int)e;
        icp_chunk0_kernel<<<dim3(1), dim3(64), 0, st>>>(src_len, n_pairs, chunk0);
        // ones[p] = 1.0f: the search's "scale" (it divides by it), since these clouds are already metric
        fill_f32_kernel<<<dim3((n_pairs + 255) / 256), dim3(256), 0, st>>>(ones, 1.0f, n_pairs);
        if (max_src_len > 0)
            icp_to_metric_kernel<<<dim3((max_src_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(src, src_row0, src_len, s, c, src_m);
        if (max_ref_len > 0)
            icp_to_metric_kernel<<<dim3((max_ref_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(ref, ref_row0, ref_len, s, c, ref_m);
        SCREAM_LAUNCH_CHECK();
        if (!brute) {  // the targets do not move: prepare and bin them once (icp_grid.hip)
            int rc = scream_internal::nn_prepare_targets(ref_m, ref_row0, ref_len, ones, n_pairs, max_ref_len, ref_prep, st);
            if (rc != 0) return rc;
            rc = scream_internal::icp_grid_build(ref_m, ref_prep, ref_row0, ref_len, n_pairs, max_ref_len, ref_rows_total, max_corr_dist,
                                                 grid_work, &grid, st);
            if (rc != 0) return rc;
        }
    }
    const dim3 chunks((max_src_len > 0 ? max_src_len + 255 : 256) / 256, n_pairs);
    // launch `it` = [evaluate search it - 1, stop or update T] + search it; search max_iter is evaluated by a last pose launch
    // (launch index max_iter + 1).  NOTHING here waits for the device: a pair that has stopped freezes on the device (its
    // blocks return at their first instruction), and a caller that wants to stop LAUNCHING early asks for the schedule in
    // pieces and reads done_flags between them (scream_hip.h).
    const int last = it_end < max_iter + 1 ? it_end : max_iter + 1;
    for (int it = it_begin; it < last; ++it) {
        if (brute) {  // SCREAM_ICP_BRUTE=1, the yardstick of the tests: the same steps as separate launches around scream_nn_search
            if (it > 0) icp_pose_kernel<<<dim3(n_pairs), dim3(256), 0, st>>>(a, it);
            if (max_src_len > 0)
                icp_transform_kernel<<<dim3((max_src_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(
                    src_m, src_row0, a.act_len, a.Tbuf + (int64_t)(it & 1) * n_pairs * 16, q);
            SCREAM_LAUNCH_CHECK();
            const int rc = scream_nn_search(q, ref_m, src_row0, a.act_len, ref_row0, ref_len, ones, n_pairs, max_src_len, max_ref_len,
                                            src_rows_total, ref_rows_total, max_corr_dist * max_corr_dist, ref_prep, keys, idx, dmin, valid, stream);
            if (rc != 0) return rc;
            icp_partials_kernel<<<chunks, dim3(256), 0, st>>>(a, q, ref_m, ref_row0, idx, valid, dmin, it);
        } else {
            icp_iter_kernel<<<chunks, dim3(256), 0, st>>>(a, src_m, ref_row0, reinterpret_cast<const scream_internal::GridParam*>(grid.params),
                                                          grid.start, grid.sorted_prep, grid.sorted_idx, max_corr_dist * max_corr_dist, it);
        }
        SCREAM_LAUNCH_CHECK();
    }
    if (it_end > max_iter + 1 && it_begin <= max_iter + 1) {
        icp_pose_kernel<<<dim3(n_pairs), dim3(256), 0, st>>>(a, max_iter + 1);
        SCREAM_LAUNCH_CHECK();
    }
    if (done_flags) {
        e = hipMemcpyAsync(done_flags, a.done, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// the whole schedule in one call: max_iter + 2 launches at most, none of them waited for
extern "C" int scream_icp_p2p(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                              const int32_t* ref_row0, const int32_t* ref_len, const float* s, const float* c,
                              int32_t n_pairs, int32_t max_src_len, int32_t max_ref_len, int64_t src_rows_total,
                              int64_t ref_rows_total, float max_corr_dist, int32_t max_iter, float rel_fitness,
                              float rel_rmse, float* T, float* fitness_rmse, int32_t* iters, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    SCREAM_REQUIRE(max_iter >= 0 && max_iter < (1 << 30), SCREAM_EINVAL);
    return scream_icp_p2p_range(src, ref, src_row0, src_len, ref_row0, ref_len, s, c, n_pairs, max_src_len, max_ref_len, src_rows_total,
                                ref_rows_total, max_corr_dist, max_iter, rel_fitness, rel_rmse, T, fitness_rmse, iters, 0, max_iter + 2,
                                nullptr, workspace, workspace_bytes, stream);
}
