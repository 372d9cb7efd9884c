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

struct IcpState {  // per pair
    float fitness, rmse;
    int32_t done, iters;
};

constexpr int ICP_NT = 512;   // threads of an ICP update workgroup (one workgroup per pair)
constexpr int ICP_KC = 16;    // correspondences a thread keeps in registers between the passes (pairs up to 8 192 points)

// One workgroup (ICP_NT threads) per pair: fitness / inlier RMSE of the current correspondences, convergence test, and (if
// the pair goes on) the Kabsch update composed into T.  q = transformed source, idx/valid/dmin from the search.  Returns
// (block-uniform) whether the pair has stopped.  Round 3: ONE gathering pass -- a thread's correspondences (a = q_i, b = its
// target, the distance) stay in registers for the centroid-relative covariance pass, where rounds 1-2 walked the dependent
// chain valid -> idx -> target row three times with 256 threads (43 us per launch, more than the search before it).  Same
// formulas as kabsch_block (centroids sum / (K + 1e-6) in fp32, covariance of the fp32 differences summed in fp64); the
// summation order is fixed by ICP_NT, so the grid and the brute-force search paths, which share this kernel, stay bit-identical.
__device__ bool icp_update_block(int p, const float* q, const float* __restrict__ ref,
                                 const int32_t* __restrict__ src_row0, const int32_t* __restrict__ src_len,
                                 const int32_t* __restrict__ ref_row0, const int32_t* idx,
                                 const uint8_t* valid, const float* dmin, int iter, int max_iter,
                                 float rel_fitness, float rel_rmse, float* T, IcpState* state,
                                 int32_t* act_len, float* fit_rmse_out, int32_t* iters_out) {
    __shared__ double red[ICP_NT / 64 * 9];
    __shared__ float dT_sh[16];
    const IcpState st = state[p];
    const int n = src_len[p];
    const int64_t r0 = src_row0[p], rr0 = ref_row0[p];
    float ca[ICP_KC][3], cb[ICP_KC][3];
    bool ok[ICP_KC];
    auto fetch = [&](int i, float (&a)[3], float (&b)[3], float& d) {
        const int64_t row = r0 + i;
        if (!valid[row]) return false;
        const int64_t rrow = rr0 + idx[row];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a[k] = q[row * 3 + k];
            b[k] = ref[rrow * 3 + k];
        }
        d = dmin[row];
        return true;
    };
    double acc[8];  // count, sum d, sum a, sum b
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    auto add1 = [&](const float (&a)[3], const float (&b)[3], float d) {
        acc[0] += 1.0;
        acc[1] += (double)d;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            acc[2 + k] += (double)a[k];
            acc[5 + k] += (double)b[k];
        }
    };
#pragma unroll
    for (int c = 0; c < ICP_KC; ++c) {
        const int i = threadIdx.x + c * ICP_NT;
        float d = 0.f;
        ok[c] = i < n && fetch(i, ca[c], cb[c], d);
        if (ok[c]) add1(ca[c], cb[c], d);
    }
    for (int i = threadIdx.x + ICP_KC * ICP_NT; i < n; i += ICP_NT) {  // (clouds beyond 8 192 points: the rest is re-fetched below)
        float a[3], b[3], d;
        if (fetch(i, a, b, d)) add1(a, b, d);
    }
    block_sum<8, ICP_NT>(acc, red);
    const float fitness = n > 0 ? (float)(acc[0] / n) : 0.f;
    const float rmse = acc[0] > 0 ? (float)sqrt(acc[1] / acc[0]) : 0.f;
    const bool converged = iter > 0 && fabsf(st.fitness - fitness) < rel_fitness && fabsf(st.rmse - rmse) < rel_rmse;
    const bool stop = converged || iter >= max_iter;
    __syncthreads();  // everyone has read state[p] before thread 0 rewrites it
    if (threadIdx.x == 0) {
        IcpState o = {fitness, rmse, stop ? 1 : 0, iter};
        state[p] = o;
        if (fit_rmse_out) {
            fit_rmse_out[2 * p + 0] = fitness;
            fit_rmse_out[2 * p + 1] = rmse;
        }
        if (iters_out) iters_out[p] = iter;  // number of updates applied
        if (stop) act_len[p] = 0;            // a finished pair costs no further transform / search work
    }
    if (stop) return true;
    const float denom = (float)acc[0] + 1e-6f;  // utils.py:155-158 with unit weights
    const float cA[3] = {(float)acc[2] / denom, (float)acc[3] / denom, (float)acc[4] / denom};
    const float cB[3] = {(float)acc[5] / denom, (float)acc[6] / denom, (float)acc[7] / denom};
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = 0.0;
    auto add2 = [&](const float (&a)[3], const float (&b)[3]) {
        const float am[3] = {a[0] - cA[0], a[1] - cA[1], a[2] - cA[2]};
        const float bm[3] = {b[0] - cB[0], b[1] - cB[1], b[2] - cB[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[r * 3 + c] += (double)am[r] * (double)bm[c];
    };
#pragma unroll
    for (int c = 0; c < ICP_KC; ++c)
        if (ok[c]) add2(ca[c], cb[c]);
    for (int i = threadIdx.x + ICP_KC * ICP_NT; i < n; i += ICP_NT) {
        float a[3], b[3], d;
        if (fetch(i, a, b, d)) add2(a, b);
    }
    block_sum<9, ICP_NT>(h, red);
    if (threadIdx.x < 64) {  // wave 0, every lane redundantly (wave-uniform data)
        double H[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[r][c] = (double)(float)h[r * 3 + c];
        float dT[16];
        solve_pose(H, cA, cB, dT);
        if (threadIdx.x < 16) dT_sh[threadIdx.x] = dT[threadIdx.x];
    }
    __syncthreads();
    float v = 0.f;
    if (threadIdx.x < 16) {  // T <- dT . T
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) v += dT_sh[i * 4 + k] * T[p * 16 + k * 4 + j];
    }
    __syncthreads();
    if (threadIdx.x < 16) T[p * 16 + threadIdx.x] = v;
    return false;
}

__global__ __launch_bounds__(ICP_NT) void icp_update_kernel(const float* __restrict__ q, const float* __restrict__ ref,
                                                           const int32_t* __restrict__ src_row0,
                                                           const int32_t* __restrict__ src_len,
                                                           const int32_t* __restrict__ ref_row0,
                                                           const int32_t* __restrict__ idx,
                                                           const uint8_t* __restrict__ valid,
                                                           const float* __restrict__ dmin, int iter, int max_iter,
                                                           float rel_fitness, float rel_rmse, float* __restrict__ T,
                                                           IcpState* __restrict__ state, int32_t* __restrict__ act_len,
                                                           float* __restrict__ fit_rmse_out,
                                                           int32_t* __restrict__ iters_out) {
    const int p = blockIdx.x;
    if (state[p].done) return;  // block-uniform
    icp_update_block(p, q, ref, src_row0, src_len, ref_row0, idx, valid, dmin, iter, max_iter, rel_fitness, rel_rmse, T, state,
                     act_len, fit_rmse_out, iters_out);
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

extern "C" int scream_transformation_error(const float* T_pred, const float* T_gt, int32_t n, float* re, float* te,
                                           void* stream) {
    SCREAM_REQUIRE(T_pred && T_gt && re && te && n >= 0, SCREAM_EINVAL);
    if (n == 0) return 0;
    transformation_error_kernel<<<dim3((n + 63) / 64), dim3(64), 0, as_stream(stream)>>>(T_pred, T_gt, n, re, te);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t scream_icp_workspace_bytes(int64_t src_rows_total, int64_t ref_rows_total, int32_t n_pairs) {
    if (src_rows_total < 0 || ref_rows_total < 0 || n_pairs < 0) return SCREAM_EINVAL;
    // src metric + transformed src (3 floats each), ref metric (3) + nn ref_prep (4), keys (2), idx, dmin, valid, ones, state
    // + the target grid of icp_grid.hip
    return (src_rows_total * (3 + 3 + 2 + 1 + 1 + 1) + ref_rows_total * (3 + 4) + (int64_t)n_pairs * 16 +
            scream_internal::icp_grid_workspace_floats(ref_rows_total, n_pairs)) * 4 + 8192;
}

extern "C" int scream_icp_p2p(const float* src, const float* ref, const int32_t* src_row0, const int32_t* src_len,
                              const int32_t* ref_row0, const int32_t* ref_len, const float* s, const float* c,
                              int32_t n_pairs, int32_t max_src_len, int32_t max_ref_len, int64_t src_rows_total,
                              int64_t ref_rows_total, float max_corr_dist, int32_t max_iter, float rel_fitness,
                              float rel_rmse, float* T, float* fitness_rmse, int32_t* iters, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    SCREAM_REQUIRE(src && ref && src_row0 && src_len && ref_row0 && ref_len && s && c && T && workspace, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_pairs >= 0 && max_iter >= 0 && max_corr_dist > 0.f, SCREAM_EINVAL);
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
    IcpState* state = reinterpret_cast<IcpState*>(take((int64_t)n_pairs * 4));
    int32_t* act_len = reinterpret_cast<int32_t*>(take(n_pairs));
    float* grid_work = take(scream_internal::icp_grid_workspace_floats(ref_rows_total, n_pairs));
    SCREAM_REQUIRE(reinterpret_cast<char*>(w) <= reinterpret_cast<char*>(workspace) + workspace_bytes, SCREAM_EINVAL);
    // SCREAM_ICP_BRUTE=1 (tests): every iteration on the brute-force search of nn_search.hip instead of the target grid
    const char* brute_env = getenv("SCREAM_ICP_BRUTE");
    const bool brute = brute_env && brute_env[0] == '1';

    hipError_t e = hipMemsetAsync(state, 0, sizeof(IcpState) * n_pairs, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(act_len, src_len, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    // ones[p] = 1.0f: the search's "scale" (it divides by it), since these clouds are already metric
    fill_f32_kernel<<<dim3((n_pairs + 255) / 256), dim3(256), 0, st>>>(ones, 1.0f, n_pairs);
    if (max_src_len > 0)
        icp_to_metric_kernel<<<dim3((max_src_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(src, src_row0, src_len, s, c, src_m);
    if (max_ref_len > 0)
        icp_to_metric_kernel<<<dim3((max_ref_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(ref, ref_row0, ref_len, s, c, ref_m);
    SCREAM_LAUNCH_CHECK();
    scream_internal::IcpGrid grid{};
    if (!brute) {  // the targets do not move: prepare and bin them once (icp_grid.hip)
        int rc = scream_internal::nn_prepare_targets(ref_m, ref_row0, ref_len, ones, n_pairs, max_ref_len, ref_prep, st);
        if (rc != 0) return rc;
        rc = scream_internal::icp_grid_build(ref_m, ref_prep, ref_row0, ref_len, n_pairs, max_ref_len, ref_rows_total, max_corr_dist,
                                             grid_work, &grid, st);
        if (rc != 0) return rc;
        rc = scream_internal::nn_fill_padding(idx, dmin, valid, src_rows_total, st);
        if (rc != 0) return rc;
    }
    std::vector<IcpState> host_state;
    for (int it = 0; it <= max_iter; ++it) {
        int rc;
        if (brute) {  // SCREAM_ICP_BRUTE=1, the yardstick of the tests: transform, then the brute-force search of nn_search.hip
            if (max_src_len > 0)
                icp_transform_kernel<<<dim3((max_src_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(src_m, src_row0, act_len, T, q);
            SCREAM_LAUNCH_CHECK();
            rc = scream_nn_search(q, ref_m, src_row0, act_len, ref_row0, ref_len, ones, n_pairs, max_src_len, max_ref_len,
                                  src_rows_total, ref_rows_total, max_corr_dist * max_corr_dist, ref_prep, keys, idx, dmin, valid, stream);
        } else {      // two launches per iteration: the transform rides in the grid search (same arithmetic, one launch fewer)
            rc = scream_internal::icp_grid_search(grid, src_m, T, q, src_row0, act_len, ref_row0, n_pairs, max_src_len,
                                                  max_corr_dist * max_corr_dist, idx, dmin, valid, st);
        }
        if (rc != 0) return rc;
        icp_update_kernel<<<dim3(n_pairs), dim3(ICP_NT), 0, st>>>(q, ref_m, src_row0, src_len, ref_row0, idx, valid, dmin, it,
                                                               max_iter, rel_fitness, rel_rmse, T, state, act_len,
                                                               fitness_rmse, iters);
        SCREAM_LAUNCH_CHECK();
        // Long schedules (KITTI asks for up to 1000 iterations, evaluate_kitti.py:69) usually converge in tens:
        // look at the flags every 32 iterations and stop launching once every pair is done.  Short schedules (the
        // 30-iteration default) never synchronise: converged pairs have frozen on the device and cost no search work,
        // and a host that does not block here can keep the other lane and the next batch queued.
        if (max_iter > 64 && (it & 31) == 31 && it < max_iter) {
            host_state.resize(n_pairs);
            e = hipMemcpyAsync(host_state.data(), state, sizeof(IcpState) * n_pairs, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return (int)e;
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) return (int)e;
            bool all_done = true;
            for (const IcpState& hs : host_state) all_done = all_done && hs.done;
            if (all_done) break;
        }
    }
    return 0;
}
