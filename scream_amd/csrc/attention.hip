// A3: linear attention ("Transformers are RNNs", models/transformer.py:17-44) for packed var-len clouds.
//
//   reduce : KV[h] = sum_s K'[s,h,:]^T (v[s,h,:] / S),  Ksum[h] = sum_s K'[s,h,:]      (:38-41)
//   apply  : out[l,h,:] = ((Q'[l,h,:] . KV[h]) * Z[l,h]) * S,  Z = 1 / (Q'[l,h,:].Ksum[h] + 1e-6)  (:41-42)
// K' = elu(k)+1 and Q' = elu(q)+1 come out of the projection GEMM's epilogue.
//
// Both are ~8 flop/byte -> HBM-bound.  The 32x32 per-head products still go through the fp32 MFMA
// (head_dim 32 == the 32x32x2 tile): in the reduce the A/B fragments are read straight from global
// memory (lane = feature, two tokens per MFMA: each half-wave reads one 128-byte row segment), so the
// reduce needs no LDS at all; the token sum is chunked (256 tokens per partial, partials added in
// chunk order by a second tiny kernel) which keeps it deterministic and bounds the fma-chain length.
#include "common.h"

namespace {

constexpr int HD = SCREAM_HEAD_DIM;       // 32
constexpr int NH = SCREAM_NHEAD;          // 8
constexpr int KV_ELEMS = (HD + 1) * HD;   // 32x32 KV^T + 32 Ksum = 1056 floats per head
constexpr int CHUNK = SCREAM_KV_CHUNK;    // 256 tokens

// grid (max_chunks, n_kv); block 512 = one wave per head.
__global__ __launch_bounds__(512) void kv_partial_kernel(const float* __restrict__ Kf, const float* __restrict__ Vf,
                                                        int64_t ld, int64_t row_base,
                                                        const int32_t* __restrict__ cloud_row0,
                                                        const int32_t* __restrict__ cloud_len, int cloud_begin,
                                                        int max_chunks, float* __restrict__ partial) {
    const int cloud = cloud_begin + blockIdx.y;
    const int len = cloud_len[cloud];
    const int t0 = blockIdx.x * CHUNK;
    if (t0 >= len) return;  // block-uniform
    const int t1 = min(len, t0 + CHUNK);
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int d = lane & 31, half = lane >> 5;
    const float S = (float)len;  // values / v_length, transformer.py:38-39
    const int64_t r0 = (int64_t)cloud_row0[cloud] - row_base;
    const float* kp = Kf + r0 * ld + h * HD + d;
    const float* vp = Vf + r0 * ld + h * HD + d;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float ks = 0.f;
    for (int t = t0; t < t1; t += 16) {  // 16 tokens = 8 MFMAs per trip, loads issued together
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tok = t + 2 * u + half;
            const bool ok = tok < t1;
            a[u] = ok ? kp[(int64_t)tok * ld] : 0.f;
            b[u] = ok ? vp[(int64_t)tok * ld] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // A[i=d][k=half] = K'[tok][d], B[k=half][j=v] = V[tok][v]/S  ->  D[d][v] += sum_k
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u] / S, acc, 0, 0, 0);
            ks += a[u];
        }
    }
    float* out = partial + (((int64_t)blockIdx.y * max_chunks + blockIdx.x) * NH + h) * KV_ELEMS;
#pragma unroll
    for (int e = 0; e < 16; ++e) out[mfma32_row(e, half) * HD + d] = acc[e];  // [d][v] with v = lane & 31
    ks += __shfl_xor(ks, 32);
    if (half == 0) out[HD * HD + d] = ks;
}

// grid n_kv * 8; block 256.  kv_out[cloud][h] = { KV^T as [v][d] (32x32), Ksum[d] (32) }.
__global__ __launch_bounds__(256) void kv_final_kernel(const float* __restrict__ partial,
                                                      const int32_t* __restrict__ cloud_len, int cloud_begin,
                                                      int max_chunks, float* __restrict__ kv_out) {
    const int kvi = blockIdx.x / NH, h = blockIdx.x % NH;
    const int cloud = cloud_begin + kvi;
    const int n_chunks = (cloud_len[cloud] + CHUNK - 1) / CHUNK;
    const float* p = partial + ((int64_t)kvi * max_chunks * NH + h) * KV_ELEMS;
    float* o = kv_out + ((int64_t)cloud * NH + h) * KV_ELEMS;
    for (int i = threadIdx.x; i < KV_ELEMS; i += 256) {
        float s = 0.f;
        for (int c = 0; c < n_chunks; ++c) s += p[(int64_t)c * NH * KV_ELEMS + i];
        if (i < HD * HD) {
            const int dd = i / HD, v = i % HD;
            o[v * HD + dd] = s;  // transpose: the apply kernel wants d contiguous per output column v
        } else {
            o[i] = s;
        }
    }
}

// Final sum of the per-128-row-tile partials written by the fused q/k/v GEMM epilogue (scream_gemm_qkv_f32).
// grid n_kv * 8; block 1024 (one element per thread: the sum is a chain of dependent loads, so parallelism across
// elements is what hides the latency).  Same output layout as kv_final_kernel.
__global__ __launch_bounds__(1024) void kv_finalize_tiles_kernel(const float* __restrict__ partial,
                                                               const int32_t* __restrict__ cloud_row0,
                                                               const int32_t* __restrict__ cloud_len, int64_t row_base,
                                                               int cloud_begin, float* __restrict__ kv_out) {
    const int kvi = blockIdx.x / NH, h = blockIdx.x % NH;
    const int cloud = cloud_begin + kvi;
    const int t0 = (int)((cloud_row0[cloud] - row_base) / SCREAM_ROW_TILE);
    const int nt = (cloud_len[cloud] + SCREAM_ROW_TILE - 1) / SCREAM_ROW_TILE;
    const float* p = partial + ((int64_t)t0 * NH + h) * KV_ELEMS;
    float* o = kv_out + ((int64_t)cloud * NH + h) * KV_ELEMS;
    const float S = (float)cloud_len[cloud];
    for (int i = threadIdx.x; i < KV_ELEMS; i += 1024) {
        // sixteen independent chains (tile c goes to chain c % 16) keep sixteen loads in flight; the combination
        // order is fixed, so the result is deterministic
        float s8[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) s8[u] = 0.f;
        for (int c = 0; c < nt; c += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c + u < nt) s8[u] += p[(int64_t)(c + u) * NH * KV_ELEMS + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += s8[u + 8];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        if (i < HD * HD) {
            const int dd = i / HD, v = i % HD;
            o[v * HD + dd] = s / S;  // values / v_length (models/transformer.py:38-39), applied to the sum
        } else {
            o[i] = s;
        }
    }
}

// grid rows/128; block 256 = 4 waves, wave w owns heads 2w and 2w+1.
constexpr int QS_LD = 260;  // 256 + 4: ds_read_b128 over 16 rows hits 16 distinct 4-bank slots
constexpr int KT_LD = 36;

__global__ __launch_bounds__(256, 2) void attn_apply_kernel(const float* __restrict__ Qf, int64_t ldq,
                                                           const float* __restrict__ kv,
                                                           const int32_t* __restrict__ tile_cloud,
                                                           int kv_cloud_offset,
                                                           const int32_t* __restrict__ cloud_len,
                                                           float* __restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float smem[NH * HD * KT_LD + NH * HD + 32 * QS_LD];
    float* KVt = smem;                 // [8][32 v][36]
    float* Ksum = smem + NH * HD * KT_LD;  // [8][32]
    float* Qs = Ksum + NH * HD;        // [32 rows][260]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * SCREAM_ROW_TILE;
    const int kvc = tile_cloud[blockIdx.x] + kv_cloud_offset;
    const float S = (float)cloud_len[kvc];
    const float* kvp = kv + (int64_t)kvc * NH * KV_ELEMS;

    for (int i = tid; i < NH * HD * (HD / 4); i += 256) {  // 2048 float4: [h][v][d/4]
        const int h = i >> 8, v = (i >> 3) & 31, d4 = i & 7;
        *reinterpret_cast<f32x4*>(KVt + (h * HD + v) * KT_LD + d4 * 4) =
            *reinterpret_cast<const f32x4*>(kvp + h * KV_ELEMS + v * HD + d4 * 4);
    }
    {
        const int h = tid >> 5, dd = tid & 31;
        Ksum[tid] = kvp[h * KV_ELEMS + HD * HD + dd];
    }

    for (int sub = 0; sub < 4; ++sub) {
        __syncthreads();  // previous sub-tile's Qs reads done (and KVt/Ksum visible on the first pass)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;  // 32 rows x 64 float4
            const int rr = f >> 6, c4 = f & 63;
            *reinterpret_cast<f32x4*>(Qs + rr * QS_LD + c4 * 4) =
                *reinterpret_cast<const f32x4*>(Qf + (row0 + sub * 32 + rr) * ldq + c4 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wave * 2 + hh;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            float zp = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k0 = kk * 8 + half * 4;
                const f32x4 af = *reinterpret_cast<const f32x4*>(Qs + r * QS_LD + h * HD + k0);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(KVt + (h * HD + r) * KT_LD + k0);
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ksum + h * HD + k0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
                    zp += af[j] * kf[j];
                }
            }
            zp += __shfl_xor(zp, 32);
            const float Z = 1.0f / (zp + 1e-6f);  // row r of this sub-tile, transformer.py:41
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rho = mfma32_row(e, half);
                const float Zr = __shfl(Z, rho);
                out[(row0 + sub * 32 + rho) * ldo + h * HD + r] = (acc[e] * Zr) * S;
            }
        }
    }
}

}  // namespace

extern "C" int scream_kv_reduce(const float* Kf, const float* Vf, int64_t ld, int64_t row_base,
                                const int32_t* cloud_row0, const int32_t* cloud_len, int32_t cloud_begin,
                                int32_t n_kv, int32_t max_chunks, float* partial, float* kv_out, void* stream) {
    SCREAM_REQUIRE(Kf && Vf && cloud_row0 && cloud_len && partial && kv_out, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_kv >= 0 && max_chunks > 0 && cloud_begin >= 0 && ld >= SCREAM_D_MODEL, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_kv <= 65535, SCREAM_EUNSUPPORTED);
    if (n_kv == 0) return 0;
    hipStream_t st = as_stream(stream);
    kv_partial_kernel<<<dim3(max_chunks, n_kv), dim3(512), 0, st>>>(Kf, Vf, ld, row_base, cloud_row0, cloud_len,
                                                                    cloud_begin, max_chunks, partial);
    SCREAM_LAUNCH_CHECK();
    kv_final_kernel<<<dim3(n_kv * NH), dim3(256), 0, st>>>(partial, cloud_len, cloud_begin, max_chunks, kv_out);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_kv_finalize(const float* kv_partial, const int32_t* cloud_row0, const int32_t* cloud_len,
                                  int64_t row_base, int32_t cloud_begin, int32_t n_kv, float* kv_out, void* stream) {
    SCREAM_REQUIRE(kv_partial && cloud_row0 && cloud_len && kv_out, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_kv >= 0 && cloud_begin >= 0 && row_base >= 0, SCREAM_EINVAL);
    if (n_kv == 0) return 0;
    kv_finalize_tiles_kernel<<<dim3(n_kv * NH), dim3(1024), 0, as_stream(stream)>>>(kv_partial, cloud_row0, cloud_len,
                                                                                   row_base, cloud_begin, kv_out);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_attn_apply(const float* Qf, int64_t ldq, const float* kv, const int32_t* tile_cloud,
                                 int32_t kv_cloud_offset, const int32_t* cloud_len, float* out, int64_t ldo,
                                 int64_t rows, void* stream) {
    SCREAM_REQUIRE(Qf && kv && tile_cloud && cloud_len && out, SCREAM_EINVAL);
    SCREAM_REQUIRE(rows >= 0 && rows % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(ldq >= SCREAM_D_MODEL && ldq % 4 == 0 && ldo >= SCREAM_D_MODEL, SCREAM_EINVAL);
    if (rows == 0) return 0;
    const int64_t blocks = rows / SCREAM_ROW_TILE;
    SCREAM_REQUIRE(blocks < (1ll << 31), SCREAM_EUNSUPPORTED);
    attn_apply_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(Qf, ldq, kv, tile_cloud,
                                                                                    kv_cloud_offset, cloud_len, out, ldo);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
