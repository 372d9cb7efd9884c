// A7: thresholded 1-nearest-neighbour search (evaluate_3d_match.py:94-95 / utils.py:72-78), brute force,
// without the N x M distance matrix of the reference.
//
// Bit-exactness contract (see include/scream_hip.h): every operation below is an explicitly rounded
// fp32 intrinsic in the order torch-CPU executes utils.py:75-77, so distances are bit-identical to
// the reference and the arg-min (strict '<' while scanning targets in ascending order; 64-bit
// (distance, index) keys when the target range is split over blocks) resolves ties to the lowest index.
// This file is compiled with -ffp-contract=off as a second line of defence.
//
// Layout: targets are pre-divided once into float4 {bx, by, bz, |b|^2} (coalesced 16-byte reads);
// a block stages 1024 targets (16 KiB) in LDS and every lane scans them with wave-uniform
// (broadcast) ds_read_b128; each thread carries QPT query points, so one LDS read feeds QPT x 8 VALU
// ops.  The kernel is VALU-bound (8 flop per pair over 12 bytes per POINT), not HBM-bound.
#include <stdint.h>

#include "common.h"
#include "icp_grid.h"

namespace {

constexpr int QPT = 4;              // queries per thread
constexpr int QB = 256 * QPT;       // queries per block
constexpr int RT = 1024;            // targets per LDS tile

__device__ __forceinline__ uint32_t f32_orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// grid (ceil(max_r_len/256), n_pairs): ref_prep[row] = {b/s, |b/s|^2}
__global__ __launch_bounds__(256) void nn_prep_kernel(const float* __restrict__ ref, const int32_t* __restrict__ r_row0,
                                                     const int32_t* __restrict__ r_len, const float* __restrict__ s,
                                                     float* __restrict__ ref_prep) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r_len[p]) return;
    const int64_t row = (int64_t)r_row0[p] + i;
    const float sp = s[p];
    const float bx = __fdiv_rn(ref[row * 3 + 0], sp);
    const float by = __fdiv_rn(ref[row * 3 + 1], sp);
    const float bz = __fdiv_rn(ref[row * 3 + 2], sp);
    const float sb = __fadd_rn(__fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by)), __fmul_rn(bz, bz));
    f32x4 o = {bx, by, bz, sb};
    *reinterpret_cast<f32x4*>(ref_prep + row * 4) = o;
}

// everything the search needs set up, as ONE launch (round 4; three before: the stage is launch-bound at 5 k points): blocks
// [0, q_blocks) initialise the keys and the outputs of every packed query row (rows outside any cloud keep idx -1, dmin inf,
// valid 0), the other p_blocks x n_pairs blocks prepare the targets like nn_prep_kernel
__global__ __launch_bounds__(256) void nn_setup_kernel(uint64_t* __restrict__ keys, int32_t* __restrict__ idx, float* __restrict__ dmin,
                                                      uint8_t* __restrict__ valid, int64_t n_q, unsigned q_blocks, unsigned p_blocks,
                                                      const float* __restrict__ ref, const int32_t* __restrict__ r_row0,
                                                      const int32_t* __restrict__ r_len, const float* __restrict__ s,
                                                      float* __restrict__ ref_prep) {
    if (blockIdx.x < q_blocks) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i < n_q) {
            keys[i] = ~0ull;
            idx[i] = -1;
            dmin[i] = __builtin_inff();
            valid[i] = 0;
        }
        return;
    }
    const unsigned b = blockIdx.x - q_blocks;
    const int p = b / p_blocks;
    const int i = (b % p_blocks) * 256 + threadIdx.x;
    if (i >= r_len[p]) return;
    const int64_t row = (int64_t)r_row0[p] + i;
    const float sp = s[p];
    const float bx = __fdiv_rn(ref[row * 3 + 0], sp);
    const float by = __fdiv_rn(ref[row * 3 + 1], sp);
    const float bz = __fdiv_rn(ref[row * 3 + 2], sp);
    const float sb = __fadd_rn(__fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by)), __fmul_rn(bz, bz));
    f32x4 o = {bx, by, bz, sb};
    *reinterpret_cast<f32x4*>(ref_prep + row * 4) = o;
}

// grid (ceil(max_q_len/QB), r_splits, n_pairs)
__global__ __launch_bounds__(256) void nn_search_kernel(const float* __restrict__ query,
                                                       const float* __restrict__ ref_prep,
                                                       const int32_t* __restrict__ q_row0,
                                                       const int32_t* __restrict__ q_len,
                                                       const int32_t* __restrict__ r_row0,
                                                       const int32_t* __restrict__ r_len,
                                                       const float* __restrict__ s, int r_per_split,
                                                       uint64_t* __restrict__ keys) {
    __shared__ __attribute__((aligned(16))) float tile[RT * 4];
    const int p = blockIdx.z;
    const int nq = q_len[p], nr = r_len[p];
    const int qb0 = blockIdx.x * QB;
    const int j_begin = blockIdx.y * r_per_split;
    const int j_end = min(nr, j_begin + r_per_split);
    if (qb0 >= nq || j_begin >= j_end) return;  // block-uniform
    const int tid = threadIdx.x;
    const float sp = s[p];
    const int64_t qrow0 = q_row0[p];
    const float* rp = ref_prep + (int64_t)r_row0[p] * 4;

    const bool wave_has_queries = qb0 + (tid & ~63) < nq;  // wave-uniform: the wave's lowest query index (u = 0, lane 0) exists
    float ax[QPT], ay[QPT], az[QPT], sa[QPT], best[QPT];
    int bi[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int qi = qb0 + tid + 256 * u;
        const int64_t row = qrow0 + min(qi, nq - 1);  // clamp: out-of-range slots recompute the last point, never stored
        ax[u] = __fdiv_rn(query[row * 3 + 0], sp);
        ay[u] = __fdiv_rn(query[row * 3 + 1], sp);
        az[u] = __fdiv_rn(query[row * 3 + 2], sp);
        sa[u] = __fadd_rn(__fadd_rn(__fmul_rn(ax[u], ax[u]), __fmul_rn(ay[u], ay[u])), __fmul_rn(az[u], az[u]));
        best[u] = __builtin_inff();
        bi[u] = 0x7fffffff;
    }

    // Round 4: the scan keeps only the running MINIMUM per chunk of NC targets (five arithmetic instructions + v_min per pair where
    // compare + two selects made it eight); a chunk whose minimum beats the best so far (strict <, so the earliest chunk wins a
    // tie) is remembered, and at the end of the tile the remembered chunk of each query is rescanned from LDS for the FIRST target
    // that attains the minimum -- the same instruction sequence, hence the same bits, and the lowest index among equal distances,
    // exactly what the scalar scan gave.  (A distance is never -0: it ends in an addition of |b|^2 >= +0.)
    constexpr int NC = 32;
    auto dist = [&](int u, const f32x4& b) __attribute__((always_inline)) {
        float dot = __fmul_rn(ax[u], b[0]);
        dot = __fmaf_rn(ay[u], b[1], dot);
        dot = __fmaf_rn(az[u], b[2], dot);
        // -2*dot is exact, so fma(-2, dot, |a|^2) rounds exactly like (-2*dot) + |a|^2
        return __fadd_rn(__fmaf_rn(-2.0f, dot, sa[u]), b[3]);
    };
    for (int jt = j_begin; jt < j_end; jt += RT) {
        const int cnt = min(RT, j_end - jt);
        __syncthreads();
        for (int i = tid; i < cnt; i += 256)
            *reinterpret_cast<f32x4*>(tile + i * 4) = *reinterpret_cast<const f32x4*>(rp + (int64_t)(jt + i) * 4);
        __syncthreads();
        if (!wave_has_queries) continue;  // (the last block of a cloud: 5 135 queries leave waves 1-3 of the sixth block without any)
        int won[QPT];  // first target (tile-relative) of the chunk that lowered this query's best in this tile, or -1
#pragma unroll
        for (int u = 0; u < QPT; ++u) won[u] = -1;
        for (int c = 0; c < cnt; c += NC) {
            const int ce = min(NC, cnt - c);
            float cm[QPT];
#pragma unroll
            for (int u = 0; u < QPT; ++u) cm[u] = __builtin_inff();
            if (ce == NC) {
#pragma unroll 8
                for (int j = 0; j < NC; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + (c + j) * 4);  // wave-uniform address: broadcast
#pragma unroll
                    for (int u = 0; u < QPT; ++u) cm[u] = fminf(cm[u], dist(u, b));
                }
            } else {
                for (int j = 0; j < ce; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + (c + j) * 4);
#pragma unroll
                    for (int u = 0; u < QPT; ++u) cm[u] = fminf(cm[u], dist(u, b));
                }
            }
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                if (cm[u] < best[u]) {
                    best[u] = cm[u];
                    won[u] = c;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            if (won[u] >= 0) {  // (per lane: the rescan reads the lane's own chunk)
                const int ce = min(NC, cnt - won[u]);
                int first = ce;
                for (int j = ce - 1; j >= 0; --j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + (won[u] + j) * 4);
                    if (dist(u, b) == best[u]) first = j;
                }
                bi[u] = jt + won[u] + first;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int qi = qb0 + tid + 256 * u;
        if (qi < nq && bi[u] != 0x7fffffff) {
            const uint64_t key = ((uint64_t)f32_orderable(best[u]) << 32) | (uint32_t)bi[u];
            atomicMin(reinterpret_cast<unsigned long long*>(keys + qrow0 + qi), (unsigned long long)key);
        }
    }
}

// grid (ceil(max_q_len/256), n_pairs)
__global__ __launch_bounds__(256) void nn_finalize_kernel(const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ q_row0,
                                                         const int32_t* __restrict__ q_len, float thresh,
                                                         int32_t* __restrict__ idx, float* __restrict__ dmin,
                                                         uint8_t* __restrict__ valid) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= q_len[p]) return;
    const int64_t row = (int64_t)q_row0[p] + i;
    const uint64_t key = keys[row];
    if (key == ~0ull) {  // empty target cloud
        idx[row] = -1;
        dmin[row] = __builtin_inff();
        valid[row] = 0;
        return;
    }
    const float d = f32_from_orderable((uint32_t)(key >> 32));
    idx[row] = (int32_t)(uint32_t)key;
    dmin[row] = d;
    valid[row] = d < thresh ? 1 : 0;
}

__global__ __launch_bounds__(256) void nn_fill_padding_kernel(int32_t* __restrict__ idx, float* __restrict__ dmin,
                                                             uint8_t* __restrict__ valid, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        idx[i] = -1;
        dmin[i] = __builtin_inff();
        valid[i] = 0;
    }
}

// Dense utils.square_distance (utils.py:72-78) for API compatibility only: the hot path never
// materialises N x M.  grid (ceil(M/256), min(N,65535), B); same rounding sequence as nn_search_kernel.
__global__ __launch_bounds__(256) void square_distance_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ dst, float* __restrict__ out,
                                                             int N, int M) {
    const int b = blockIdx.z;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* bp = dst + ((int64_t)b * M + m) * 3;
    const float bx = bp[0], by = bp[1], bz = bp[2];
    const float sb = __fadd_rn(__fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by)), __fmul_rn(bz, bz));
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float* ap = src + ((int64_t)b * N + n) * 3;
        const float ax = ap[0], ay = ap[1], az = ap[2];
        const float sa = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
        float dot = __fmul_rn(ax, bx);
        dot = __fmaf_rn(ay, by, dot);
        dot = __fmaf_rn(az, bz, dot);
        out[((int64_t)b * N + n) * M + m] = __fadd_rn(__fmaf_rn(-2.0f, dot, sa), sb);
    }
}

}  // namespace

namespace scream_internal {  // icp_grid.h: shared with the ICP loop

int nn_prepare_targets(const float* ref, const int32_t* r_row0, const int32_t* r_len, const float* s, int32_t n_pairs,
                       int32_t max_r_len, float* ref_prep, hipStream_t st) {
    if (max_r_len <= 0 || n_pairs <= 0) return 0;
    nn_prep_kernel<<<dim3((max_r_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(ref, r_row0, r_len, s, ref_prep);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

int nn_fill_padding(int32_t* idx, float* dmin, uint8_t* valid, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    nn_fill_padding_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(idx, dmin, valid, n);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

}  // namespace scream_internal

extern "C" int scream_square_distance(const float* src, const float* dst, float* out, int32_t B, int32_t N, int32_t M,
                                      void* stream) {
    SCREAM_REQUIRE(B >= 0 && N >= 0 && M >= 0, SCREAM_EINVAL);
    if (B == 0 || N == 0 || M == 0) return 0;
    SCREAM_REQUIRE(src && dst && out, SCREAM_EINVAL);
    SCREAM_REQUIRE(B <= 65535, SCREAM_EUNSUPPORTED);
    square_distance_kernel<<<dim3((M + 255) / 256, N < 65535 ? N : 65535, B), dim3(256), 0, as_stream(stream)>>>(
        src, dst, out, N, M);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_nn_search(const float* query, const float* ref, const int32_t* q_row0, const int32_t* q_len,
                                const int32_t* r_row0, const int32_t* r_len, const float* s, int32_t n_pairs,
                                int32_t max_q_len, int32_t max_r_len, int64_t q_rows_total, int64_t r_rows_total,
                                float thresh, float* ref_prep, uint64_t* keys, int32_t* idx, float* dmin,
                                uint8_t* valid, void* stream) {
    SCREAM_REQUIRE(query && ref && q_row0 && q_len && r_row0 && r_len && s && ref_prep && keys && idx && dmin && valid,
                   SCREAM_EINVAL);
    SCREAM_REQUIRE(n_pairs >= 0 && max_q_len >= 0 && max_r_len >= 0 && q_rows_total >= 0 && r_rows_total >= 0,
                   SCREAM_EINVAL);
    SCREAM_REQUIRE(n_pairs <= 65535, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(ref_prep) & 15) == 0, SCREAM_EINVAL);
    if (n_pairs == 0 || q_rows_total == 0) return 0;
    hipStream_t st = as_stream(stream);
    const unsigned qblk = (unsigned)((q_rows_total + 255) / 256);
    const unsigned pblk = max_q_len > 0 ? (unsigned)((max_r_len + 255) / 256) : 0;  // (no queries: nothing reads the prepared targets)
    SCREAM_REQUIRE((uint64_t)qblk + (uint64_t)pblk * n_pairs < (1ull << 31), SCREAM_EUNSUPPORTED);
    nn_setup_kernel<<<dim3(qblk + pblk * n_pairs), dim3(256), 0, st>>>(keys, idx, dmin, valid, q_rows_total, qblk, pblk, ref, r_row0, r_len, s, ref_prep);
    SCREAM_LAUNCH_CHECK();
    if (max_q_len == 0) return 0;
    if (max_r_len > 0) {
        const int qblocks = (max_q_len + QB - 1) / QB;
        // split the target range so that the launch is a whole number of rounds of equal blocks on the 256 CUs when it can be (5 k
        // points: 6 query blocks x 4 splits x 32 pairs = 768 blocks = 3 per CU, where six splits of 1 024 left 4 full blocks on some
        // CUs and 3 on others): the split count in [1, 64] that minimises rounds x targets per split; results do not depend on it
        // (the 64-bit (distance, index) keys of the splits are merged with atomicMin)
        const int max_splits = (max_r_len + 255) / 256 < 64 ? (max_r_len + 255) / 256 : 64;
        int splits = 1;
        int64_t best_cost = INT64_MAX;
        for (int sp = 1; sp <= (max_splits > 0 ? max_splits : 1); ++sp) {
            const int64_t blocks = (int64_t)qblocks * sp * n_pairs, per = (max_r_len + sp - 1) / sp;
            const int64_t cost = ((blocks + 255) / 256) * (per + 64);  // (+ 64: a block's fixed cost in units of targets)
            if (cost < best_cost) {
                best_cost = cost;
                splits = sp;
            }
        }
        int r_per_split = (max_r_len + splits - 1) / splits;
        splits = (max_r_len + r_per_split - 1) / r_per_split;
        nn_search_kernel<<<dim3(qblocks, splits, n_pairs), dim3(256), 0, st>>>(query, ref_prep, q_row0, q_len, r_row0,
                                                                              r_len, s, r_per_split, keys);
        SCREAM_LAUNCH_CHECK();
    }
    nn_finalize_kernel<<<dim3((max_q_len + 255) / 256, n_pairs), dim3(256), 0, st>>>(keys, q_row0, q_len, thresh, idx,
                                                                                     dmin, valid);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
