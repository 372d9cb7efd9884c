// Internal interface between kabsch.hip (the ICP loop) and icp_grid.hip / nn_search.hip; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int ICP_GRID_CELLS = 1 << 17;  // cells per pair (50^3 fits: a 5 m cloud at the 0.1 m radius of evaluate_3d_match.py)

namespace scream_internal {

struct GridParam {  // per pair
    float ox, oy, oz, inv_h;
    int32_t nx, ny, nz, ncell;
};

#ifdef __HIPCC__
typedef float icp_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h, int n) {
    const int c = (int)floorf((x - o) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// Nearest target of ONE query point within the radius (thresh = radius^2), over the nine x-runs of the 27 cells around it:
// the same explicitly rounded distance as nn_search_kernel with s = 1, ties to the lowest ORIGINAL target index.
// st / sp / si: the pair's cell starts, sorted records {b, |b|^2} and original indices; b_out: the coordinates of the winner
// (the record's own, i.e. what ref[idx] holds).  Used by icp_iter_kernel (kabsch.hip), one launch per iteration.
__device__ __forceinline__ void grid_search_point(const GridParam& g, const int32_t* __restrict__ st, const float* __restrict__ sp,
                                                  const int32_t* __restrict__ si, float ax, float ay, float az, float thresh,
                                                  int32_t& idx_out, float& d_out, uint8_t& valid_out, float (&b_out)[3]) {
    const float sa = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
    const int cx = cell_coord(ax, g.ox, g.inv_h, g.nx), cy = cell_coord(ay, g.oy, g.inv_h, g.ny), cz = cell_coord(az, g.oz, g.inv_h, g.nz);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    float best = __builtin_inff();
    int bi = 0x7fffffff;
    float bx = 0.f, by = 0.f, bz = 0.f;
    // The nine x-runs: all eighteen cell-start loads go out together (a thread's time here is memory latency: one dependent
    // round per run cost 32 us per search launch), then the candidates of each run, four at a time.
    int jb[9], je[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = z >= 0 && z < g.nz && y >= 0 && y < g.ny;
        const int c0 = in ? (z * g.ny + y) * g.nx : 0;
        jb[r] = st[c0 + x0];
        je[r] = in ? st[c0 + x1 + 1] : jb[r];  // (an outside run is empty)
    }
    auto take = [&](const icp_f32x4& b, int o) {  // same explicitly rounded sequence as nn_search_kernel; ties -> lowest index
        float dot = __fmul_rn(ax, b[0]);
        dot = __fmaf_rn(ay, b[1], dot);
        dot = __fmaf_rn(az, b[2], dot);
        const float d = __fadd_rn(__fmaf_rn(-2.0f, dot, sa), b[3]);
        if (d < best || (d == best && o < bi)) {
            best = d;
            bi = o;
            bx = b[0];
            by = b[1];
            bz = b[2];
        }
    };
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        int j = jb[r];
        for (; j + 4 <= je[r]; j += 4) {
            icp_f32x4 b[4];
            int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b[u] = *reinterpret_cast<const icp_f32x4*>(sp + (int64_t)(j + u) * 4);
                o[u] = si[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) take(b[u], o[u]);
        }
        for (; j < je[r]; ++j) take(*reinterpret_cast<const icp_f32x4*>(sp + (int64_t)j * 4), si[j]);
    }
    const bool ok = best < thresh;
    idx_out = ok ? bi : -1;
    d_out = ok ? best : __builtin_inff();
    valid_out = ok ? 1 : 0;
    b_out[0] = bx;
    b_out[1] = by;
    b_out[2] = bz;
}
#endif

struct IcpGrid {
    const void* params;        // per pair: origin, 1 / cell edge, dimensions
    const int32_t* start;      // [n_pairs][ICP_GRID_CELLS + 1] first sorted position of every cell
    const float* sorted_prep;  // [ref_rows_total][4] {b, |b|^2} in cell order (per pair, from the pair's first target row)
    const int32_t* sorted_idx; // [ref_rows_total] original target index of every sorted record
};

int64_t icp_grid_workspace_floats(int64_t ref_rows_total, int32_t n_pairs);
void icp_grid_carve(int64_t ref_rows_total, int32_t n_pairs, float* work, IcpGrid* out);  // the pointers icp_grid_build fills
int icp_grid_build(const float* ref_m, const float* ref_prep, const int32_t* r_row0, const int32_t* r_len, int32_t n_pairs,
                   int32_t max_r_len, int64_t ref_rows_total, float radius, float* work, IcpGrid* out, hipStream_t st);
// nn_search.hip: ref_prep[row] = {b / s, |b / s|^2} (the brute-force search's own preparation) and the padding fill
int nn_prepare_targets(const float* ref, const int32_t* r_row0, const int32_t* r_len, const float* s, int32_t n_pairs,
                       int32_t max_r_len, float* ref_prep, hipStream_t st);
int nn_fill_padding(int32_t* idx, float* dmin, uint8_t* valid, int64_t n, hipStream_t st);

}  // namespace scream_internal
