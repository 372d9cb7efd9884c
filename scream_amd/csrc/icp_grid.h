// Internal interface between kabsch.hip (the ICP loop) and icp_grid.hip / nn_search.hip; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int ICP_GRID_CELLS = 1 << 17;  // cells per pair (50^3 fits: a 5 m cloud at the 0.1 m radius of evaluate_3d_match.py)

namespace scream_internal {

struct IcpGrid {
    const void* params;        // per pair: origin, 1 / cell edge, dimensions
    const int32_t* start;      // [n_pairs][ICP_GRID_CELLS + 1] first sorted position of every cell
    const float* sorted_prep;  // [ref_rows_total][4] {b, |b|^2} in cell order (per pair, from the pair's first target row)
    const int32_t* sorted_idx; // [ref_rows_total] original target index of every sorted record
};

int64_t icp_grid_workspace_floats(int64_t ref_rows_total, int32_t n_pairs);
int icp_grid_build(const float* ref_m, const float* ref_prep, const int32_t* r_row0, const int32_t* r_len, int32_t n_pairs,
                   int32_t max_r_len, int64_t ref_rows_total, float radius, float* work, IcpGrid* out, hipStream_t st);
int icp_grid_search(const IcpGrid& g, const float* query, const int32_t* q_row0, const int32_t* q_len, const int32_t* r_row0,
                    int32_t n_pairs, int32_t max_q_len, float thresh, int32_t* idx, float* dmin, uint8_t* valid, hipStream_t st);
// nn_search.hip: ref_prep[row] = {b / s, |b / s|^2} (the brute-force search's own preparation) and the padding fill
int nn_prepare_targets(const float* ref, const int32_t* r_row0, const int32_t* r_len, const float* s, int32_t n_pairs,
                       int32_t max_r_len, float* ref_prep, hipStream_t st);
int nn_fill_padding(int32_t* idx, float* dmin, uint8_t* valid, int64_t n, hipStream_t st);

}  // namespace scream_internal
