// Radius-limited nearest neighbour for the ICP loop (scream_icp_p2p, kabsch.hip) on a uniform grid over the TARGET cloud.
//
// open3d's registration_icp, which the reference calls after every prediction (evaluate_3d_match.py:106-119,
// evaluate_kitti.py:63-76), only ever uses correspondences closer than max_correspondence_distance, and the target cloud does
// not move during the loop.  The brute-force search of nn_search.hip evaluates every source x target distance in every
// iteration (25 M per 5 k-point pair, 250 M at KITTI size, times 30 .. 1000 iterations); here the targets are binned once per
// call into cells of edge h >= 1.01 x the radius, and a query looks at the 27 cells around its own.
//
// Same results, bit for bit, wherever the brute-force search reports a valid correspondence: a candidate's distance is
// computed by the same explicitly rounded sequence (|a|^2 - 2 a.b + |b|^2 in nn_search_kernel's order, on the same
// pre-divided target records), ties go to the lowest ORIGINAL target index, and every target whose computed distance is
// below the threshold lies in the 27 cells (the 1 % margin on h is four orders of magnitude above the rounding of the
// cell arithmetic and of the distance).  Where no target is inside the radius the brute-force search still returns the
// global nearest neighbour with valid = 0; this one returns idx = -1, dmin = inf, valid = 0 -- the ICP update reads
// neither for such a point (icp_store_partial, kabsch.hip: valid correspondences only).
// The search itself (grid_search_point, icp_grid.h) is called from icp_iter_kernel (kabsch.hip), which does a whole iteration.
//
// Build, once per scream_icp_p2p call: bounding box per pair -> grid parameters (the cell edge grows by 2^(1/3) until the
// grid fits ICP_GRID_CELLS cells: correct for any h >= the radius, only more candidates) -> count -> exclusive scan ->
// scatter of the prepared records {b, |b|^2} and their original indices in cell order.  Cell index = (z ny + y) nx + x, so
// the three x-neighbours of a cell are one contiguous run and a query scans nine runs.
#include "common.h"
#include "icp_grid.h"

namespace {

using scream_internal::GridParam;
using scream_internal::cell_coord;

// grid n_pairs, block 256: bounding box of the pair's targets (metric coordinates) and the grid over it
__global__ __launch_bounds__(256) void grid_params_kernel(const float* __restrict__ ref, const int32_t* __restrict__ r_row0,
                                                         const int32_t* __restrict__ r_len, float h0,
                                                         GridParam* __restrict__ gp) {
    __shared__ float red[6][256];
    const int p = blockIdx.x, n = r_len[p];
    const int64_t r0 = r_row0[p];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = ref[(r0 + i) * 3 + k];
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        red[k][threadIdx.x] = lo[k];
        red[3 + k][threadIdx.x] = hi[k];
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
                red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        GridParam g;
        if (n <= 0) {
            g = GridParam{0.f, 0.f, 0.f, 1.f, 1, 1, 1, 1};
        } else {
            float h = h0;
            int nx, ny, nz;
            for (;;) {  // one spare cell on either side: a query within the radius of the box still maps next to its targets
                nx = (int)floorf((red[3][0] - red[0][0]) / h) + 3;
                ny = (int)floorf((red[4][0] - red[1][0]) / h) + 3;
                nz = (int)floorf((red[5][0] - red[2][0]) / h) + 3;
                if ((int64_t)nx * ny * nz <= ICP_GRID_CELLS) break;
                h *= 1.26f;
            }
            g = GridParam{red[0][0] - h, red[1][0] - h, red[2][0] - h, 1.0f / h, nx, ny, nz, nx * ny * nz};
        }
        gp[p] = g;
    }
}

// grid (ceil(max_r_len / 256), n_pairs)
__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ ref, const int32_t* __restrict__ r_row0,
                                                        const int32_t* __restrict__ r_len, const GridParam* __restrict__ gp,
                                                        int32_t* __restrict__ count, int32_t* __restrict__ cell_of) {
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r_len[p]) return;
    const GridParam g = gp[p];
    const int64_t row = (int64_t)r_row0[p] + i;
    const int cx = cell_coord(ref[row * 3 + 0], g.ox, g.inv_h, g.nx), cy = cell_coord(ref[row * 3 + 1], g.oy, g.inv_h, g.ny),
              cz = cell_coord(ref[row * 3 + 2], g.oz, g.inv_h, g.nz);
    const int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[row] = c;
    atomicAdd(count + (int64_t)p * (ICP_GRID_CELLS + 1) + c, 1);
}

// grid n_pairs, block 1024: in-place exclusive scan of count[p][0 .. ncell] (entry ncell = total), and a copy as the
// scatter cursors
__global__ __launch_bounds__(1024) void grid_scan_kernel(const GridParam* __restrict__ gp, int32_t* __restrict__ count,
                                                        int32_t* __restrict__ cursor) {
    __shared__ int32_t part[1024];
    const int p = blockIdx.x, t = threadIdx.x;
    const int ncell = gp[p].ncell;
    int32_t* c = count + (int64_t)p * (ICP_GRID_CELLS + 1);
    int32_t* cur = cursor + (int64_t)p * ICP_GRID_CELLS;
    const int chunk = (ncell + 1023) / 1024;
    const int b = t * chunk, e = min(ncell, b + chunk);
    int32_t s = 0;
    for (int i = b; i < e; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of the per-thread sums
        const int32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - s;
    for (int i = b; i < e; ++i) {
        const int32_t v = c[i];
        c[i] = run;
        cur[i] = run;
        run += v;
    }
    if (t == 1023) c[ncell] = part[1023];
}

// grid (ceil(max_r_len / 256), n_pairs): the prepared records and original indices in cell order
__global__ __launch_bounds__(256) void grid_scatter_kernel(const float* __restrict__ ref_prep, const int32_t* __restrict__ r_row0,
                                                          const int32_t* __restrict__ r_len, const int32_t* __restrict__ cell_of,
                                                          int32_t* __restrict__ cursor, float* __restrict__ sorted_prep,
                                                          int32_t* __restrict__ sorted_idx) {
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r_len[p]) return;
    const int64_t r0 = r_row0[p], row = r0 + i;
    const int pos = atomicAdd(cursor + (int64_t)p * ICP_GRID_CELLS + cell_of[row], 1);
    *reinterpret_cast<f32x4*>(sorted_prep + (r0 + pos) * 4) = *reinterpret_cast<const f32x4*>(ref_prep + row * 4);
    sorted_idx[r0 + pos] = i;
}

}  // namespace

namespace scream_internal {

int64_t icp_grid_workspace_floats(int64_t ref_rows_total, int32_t n_pairs) {
    // params (8) + count/start (CELLS + 1) + cursor (CELLS) per pair; cell_of (1) + sorted record (4) + sorted index (1) per target
    return (int64_t)n_pairs * (8 + 2 * (int64_t)ICP_GRID_CELLS + 1) + ref_rows_total * 6 + 5 * 64;
}

// where the grid of a workspace lives: a pure function of (sizes, work), so a later call of the same run finds the grid an
// earlier one built (scream_icp_p2p_range)
struct GridCarve {
    GridParam* gp;
    int32_t *count, *cursor, *cell_of, *sorted_idx;
    float* sorted_prep;
};
static GridCarve carve_grid(int64_t ref_rows_total, int32_t n_pairs, float* work) {
    float* w = work;
    auto take = [&](int64_t n) { float* r = w; w += (n + 63) / 64 * 64; return r; };
    GridCarve c;
    c.gp = reinterpret_cast<GridParam*>(take((int64_t)n_pairs * 8));
    c.count = reinterpret_cast<int32_t*>(take((int64_t)n_pairs * (ICP_GRID_CELLS + 1)));
    c.cursor = reinterpret_cast<int32_t*>(take((int64_t)n_pairs * ICP_GRID_CELLS));
    c.cell_of = reinterpret_cast<int32_t*>(take(ref_rows_total));
    c.sorted_prep = take(ref_rows_total * 4);
    c.sorted_idx = reinterpret_cast<int32_t*>(take(ref_rows_total));
    return c;
}

void icp_grid_carve(int64_t ref_rows_total, int32_t n_pairs, float* work, IcpGrid* out) {
    const GridCarve c = carve_grid(ref_rows_total, n_pairs, work);
    out->params = c.gp;
    out->start = c.count;
    out->sorted_prep = c.sorted_prep;
    out->sorted_idx = c.sorted_idx;
}

int icp_grid_build(const float* ref_m, const float* ref_prep, const int32_t* r_row0, const int32_t* r_len, int32_t n_pairs,
                   int32_t max_r_len, int64_t ref_rows_total, float radius, float* work, IcpGrid* out, hipStream_t st) {
    const GridCarve cv = carve_grid(ref_rows_total, n_pairs, work);
    GridParam* gp = cv.gp;
    int32_t *count = cv.count, *cursor = cv.cursor, *cell_of = cv.cell_of, *sorted_idx = cv.sorted_idx;
    float* sorted_prep = cv.sorted_prep;
    icp_grid_carve(ref_rows_total, n_pairs, work, out);
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)n_pairs * (ICP_GRID_CELLS + 1), st);
    if (e != hipSuccess) return (int)e;
    grid_params_kernel<<<dim3(n_pairs), dim3(256), 0, st>>>(ref_m, r_row0, r_len, radius * 1.01f, gp);
    SCREAM_LAUNCH_CHECK();
    if (max_r_len > 0) {
        const dim3 grid((max_r_len + 255) / 256, n_pairs);
        grid_count_kernel<<<grid, dim3(256), 0, st>>>(ref_m, r_row0, r_len, gp, count, cell_of);
        SCREAM_LAUNCH_CHECK();
        grid_scan_kernel<<<dim3(n_pairs), dim3(1024), 0, st>>>(gp, count, cursor);
        SCREAM_LAUNCH_CHECK();
        grid_scatter_kernel<<<grid, dim3(256), 0, st>>>(ref_prep, r_row0, r_len, cell_of, cursor, sorted_prep, sorted_idx);
        SCREAM_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace scream_internal
