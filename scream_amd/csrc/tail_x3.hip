// The row-local tail of an MHAttention block (models/transformer.py:83-88) as ONE kernel on the gfx950 bf16 matrix
// cores, fp32-accurate by the same 3-way operand split as gemm_x3.hip:
//
//     m1 = LayerNorm1(merge(att) + x)        [phase 2, not in this file yet: m1 is an input for now]
//     y  = LayerNorm2(x + W2 . relu(W1 . m1))
//
// Why one kernel.  At the 1400 W socket cap a GEMM launch costs what its joules cost, and the
// unfused FFN spends a third of them moving the 1024-wide hidden activations: 4 KB per row written by the FFN-up
// epilogue (through LDS slabs, to be row-major), 4 KB per row read back and re-split by FFN-down.  Here the hidden
// activations never leave the register file.
//
// How: everything is computed TRANSPOSED.  The weights are the MFMA A operand (M = output feature), a wave's 32
// activation rows are the B operand (N = row), so an accumulator holds C^T: lane = activation row, registers = output
// features.  That IS the B-operand layout of the next GEMM (lane = row, registers = contraction index), so
//     h^T[32 hidden x 32 rows]  = W1[chunk] . m1^T          (96 MFMAs, one accumulator tile)
//     relu, split into 3 bf16 planes in registers
//     y^T[256 x 32 rows]       += W2[:, chunk] . h^T         (96 MFMAs, eight accumulator tiles)
// for 32 chunks of 32 hidden units, with no transposition anywhere: the contraction index of an MFMA is a dummy, so the
// fixed permutation between "register i of lane (r, half)" and "hidden unit" is baked into the packed weight images
// (pack_ffn_kernel below).  LayerNorm is a sum over a lane's 128 registers plus ONE cross-lane add (lanes r and r + 32
// share a row) instead of a slab transpose and 2 x 32 DPP wave reductions per wave.
//
// Geometry: 256 threads = 4 waves, one per SIMD (404 of the 512 registers: 128 accumulators, 192 for the three bf16
// planes of the wave's m1 rows, the rest operands), one persistent block per CU, 128 rows per block tile.  The weights
// stream through a ring of three 48 KiB LDS stages filled by LDS-DMA (global_load_lds_dwordx4), one stage per 96 MFMAs;
// the stage sequence (W1 chunk 0, W2 chunk 0, W1 chunk 1, ...) is the same for every row tile, so the ring never
// drains at a tile boundary.  A stage image is stored exactly as the fragments are read: [plane][fragment][lane][16 B],
// i.e. every ds_read_b128 and every DMA piece is 1 KiB of contiguous memory, conflict-free without any swizzle.
//
// HBM traffic per row: m1 1 KB + x 1 KB in, y 1 KB out (FFN-up + FFN-down launches: 11 KB).  The loads and stores are
// row-per-lane (512 contiguous bytes per lane) -- eight times the address-processing work of a coalesced access, and
// irrelevant here: a row tile is 24 576 MFMAs per wave.
// Tuning aid (tools/tail_ablate.py builds variants): bit 0 no weight DMA after the first two stages, 1 no MFMAs,
// 2 no LDS fragment reads, 3 no relu/split between the two GEMMs, 4 no row loads/stores (m1, x, y).  Always 0 in
// libscream_hip.so.
#ifndef T_ABLATE
#define T_ABLATE 0
#endif
#include <type_traits>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TT = 256;               // threads
constexpr int T_STAGE = 48 * 1024;    // one ring stage: 48 fragments of 1 KiB
constexpr int T_SLOTS = 3;
constexpr int FFN_STAGES = 64;        // W1 chunk c -> stage 2c, W2 chunk c -> stage 2c + 1
constexpr int T_MAX_GRID = 256;

__device__ __forceinline__ void split3(const f32x4 lo, const f32x4 hi, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? lo[i] : hi[i - 4];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[i] = a;
        p1[i] = b;
        p2[i] = (__bf16)(r1 - (float)b);
    }
}

// s_waitcnt vmcnt(N) lgkmcnt(0) + workgroup barrier (see gemm_x3.hip): the N youngest vector-memory operations of
// this wave -- the DMA pieces of the stage after the one about to be read -- stay in flight across the barrier.
template <int N>
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
}

// Layout convention of every transposed tile in this file: accumulator tile blk, register i of lane (r, half) holds
// feature 32 blk + mfma32_row(i, half) = 32 blk + 8 (i >> 2) + 4 half + (i & 3) of activation row r.  When such a tile
// is the B operand of the next GEMM, registers 8 s2 .. 8 s2 + 7 are 16-deep step s2, so lane-half `half` supplies, as
// element j of step s2, contraction index chunk_k(s2, half, j) of its 32-wide chunk.
__host__ __device__ __forceinline__ int chunk_k(int s2, int half, int j) { return 8 * (2 * s2 + (j >> 2)) + 4 * half + (j & 3); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// one 1 KiB weight fragment (16 bytes per lane) from the current stage
__device__ __forceinline__ bf16x8 ld_frag(const char* p) {
    if (T_ABLATE & 4) {
        bf16x8 v;
        asm volatile("" : "=v"(v));  // opaque, undefined: keeps the consumers alive without the LDS read
        return v;
    }
    return *reinterpret_cast<const bf16x8*>(p);
}

// acc += W . act with both operands split in three bf16 planes: six exact products, smallest first
// (W plane + activation plane <= 2), fp32 accumulate.  w = A operand (weights), a = B operand (activations).
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3]) {
    if (T_ABLATE & 2) {
        acc[0] += (float)w[0][0] + (float)w[1][1] + (float)w[2][2] + (float)a[0][0] + (float)a[1][1] + (float)a[2][2];
        return;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], a[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], a[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[0], acc, 0, 0, 0);
    // first MFMA, then the three prefetch reads of the next fragment group, then the other five MFMAs
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TT, 1) void ffn_x3_kernel(const float* __restrict__ m1, int64_t ldm,
                                                       const __bf16* __restrict__ Wimg,  // [64 stages][48 KiB]
                                                       const float* __restrict__ xres, int64_t ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int64_t ldy, int n_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[T_SLOTS * T_STAGE];  // 144 KiB, the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, half = lane >> 5;

    // DMA: a stage is 48 pieces of 1 KiB, twelve per wave; the image is stored in read order, so the per-lane part of
    // the source address is lane * 16 bytes and the destination is the piece's base (the hardware adds lane * 16).
    const char* w_lane = reinterpret_cast<const char*>(Wimg) + lane * 16;
    auto dma_stage = [&](unsigned q) {  // q = running stage number of this block
        if ((T_ABLATE & 1) && q >= 2) return;
        const unsigned src = q % (unsigned)FFN_STAGES, slot = q % (unsigned)T_SLOTS;
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int piece = wave * 12 + u;
            __builtin_amdgcn_global_load_lds((gptr_t)(w_lane + (size_t)src * T_STAGE + piece * 1024),
                                             (lptr_t)(smem + slot * T_STAGE + piece * 1024), 16, 0, 0);
        }
    };

    // one DMA piece of stage q (12 per wave and stage): issued BETWEEN the MFMA groups of the stage two before it, one
    // per group -- a piece costs the issuing wave ~60 cycles, and twelve of them back to back behind the barrier were
    // 13 % of the kernel (tools/tail_ablate.py)
    auto dma_piece = [&](unsigned q, int u) {
        if ((T_ABLATE & 1) && q >= 2) return;
        const unsigned src = q % (unsigned)FFN_STAGES, slot = q % (unsigned)T_SLOTS;
        const int piece = wave * 12 + u;
        __builtin_amdgcn_global_load_lds((gptr_t)(w_lane + (size_t)src * T_STAGE + piece * 1024),
                                         (lptr_t)(smem + slot * T_STAGE + piece * 1024), 16, 0, 0);
    };

    unsigned q = 0;  // next stage to be consumed
    dma_stage(0);
    dma_stage(1);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row = (int64_t)tile * 128 + wave * 32 + r;
        // ---- this lane's share of its m1 row -> three bf16 planes: step s = 2 blk + s2 holds features 32 blk + chunk_k(s2, half, .)
        bf16x8 mp[16][3];
        {
            const float* p = m1 + row * ldm + 4 * half;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (T_ABLATE & 16) {  // no loads: derive operands from the lane id
                    const f32x4 f = {(float)lane, (float)s, 1.0f, (float)tile};
                    split3(f, f, mp[s][0], mp[s][1], mp[s][2]);
                } else {
                    split3(ld4(p + 16 * s), ld4(p + 16 * s + 8), mp[s][0], mp[s][1], mp[s][2]);
                }
            }
        }
        f32x16 acc[8];
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        f32x16 hT;
        bf16x8 hpA[2][3], hpB[2][3];

        // relu + split of elements 2k, 2k+1 of the finished h^T tile into the B-operand planes of the next GEMM
        auto split_pair = [&](int k, bf16x8 (&hout)[2][3]) {
            const int s2 = k >> 2, j = (2 * k) & 7;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x = fmaxf(hT[2 * k + e], 0.f);
                if (T_ABLATE & 8) {
                    hout[s2][0][j + e] = (__bf16)x;
                    hout[s2][1][j + e] = hout[s2][0][j + e];
                    hout[s2][2][j + e] = hout[s2][0][j + e];
                } else {
                    const __bf16 a = (__bf16)x;
                    const float r1 = x - (float)a;
                    const __bf16 b = (__bf16)r1;
                    hout[s2][0][j + e] = a;
                    hout[s2][1][j + e] = b;
                    hout[s2][2][j + e] = (__bf16)(r1 - (float)b);
                }
            }
        };
        // Stage kinds.  Fragments are prefetched TWO groups ahead (three register sets); DMA piece g of the stage after
        // next goes out in group g.  FIRST: the first stage of a row tile has this tile's m1 loads and the previous
        // tile's y stores in the queue, so its barrier drains it (a counted wait is only sound among loads, gemm_x3.hip).
        auto stage_up = [&](auto first) {  // h^T = W1[chunk] . m1^T
            if (decltype(first)::value) ring_barrier<0>(); else ring_barrier<12>();
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[3][3];
#pragma unroll
            for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
#pragma unroll
            for (int e = 0; e < 16; ++e) hT[e] = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) wf[(g + 2) % 3][p] = ld_frag(wb + (p * 16 + g + 2) * 1024);
                }
                if (g < 12) dma_piece(q + 2, g);
                mfma6(hT, wf[g % 3], mp[g]);
            }
            ++q;
        };
        auto stage_down = [&](bf16x8 (&hin)[2][3], bf16x8 (&hout)[2][3], auto with_split) {  // y^T += W2[:, chunk] . h^T
            ring_barrier<12>();
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[3][3];
#pragma unroll
            for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
#pragma unroll
            for (int g = 0; g < 16; ++g) {  // g = blk * 2 + s2
                if (g + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) wf[(g + 2) % 3][p] = ld_frag(wb + (p * 16 + g + 2) * 1024);
                }
                if (g < 12) dma_piece(q + 2, g);
                // the relu/split of the h^T tile the previous stage finished rides under this stage's MFMAs
                if (decltype(with_split)::value && (g & 1) == 0) split_pair(g >> 1, hout);
                mfma6(acc[g >> 1], wf[g % 3], hin[g & 1]);
            }
            ++q;
        };
        constexpr std::integral_constant<bool, true> yes{};
        constexpr std::integral_constant<bool, false> no{};

        // stage order (= image order): W1_0 | W1_c, W2_{c-1} for c = 1 .. 31 | W2_31
        stage_up(yes);
#pragma unroll
        for (int k = 0; k < 8; ++k) split_pair(k, hpA);  // chunk 0: nothing to hide under yet
        for (int c = 1; c < 31; c += 2) {
            stage_up(no);               // chunk c
            stage_down(hpA, hpB, yes);  // chunk c - 1, splitting chunk c
            stage_up(no);               // chunk c + 1
            stage_down(hpB, hpA, yes);  // chunk c, splitting chunk c + 1
        }
        stage_up(no);                // chunk 31
        stage_down(hpA, hpB, yes);   // chunk 30, splitting chunk 31
        stage_down(hpB, hpA, no);    // chunk 31

        // ---- y = LayerNorm2(x + ffn): acc[blk][i] is feature 32 blk + 8 (i >> 2) + 4 half + (i & 3) of row r
        const float* xp = xres + row * ldx + 4 * half;
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 xv = (T_ABLATE & 16) ? f32x4{1.f, 2.f, 3.f, 4.f} : ld4(xp + 32 * b + 8 * i4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[b][4 * i4 + k] += xv[k];
                    sum += acc[b][4 * i4 + k];
                }
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 256.0f);
        float var = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[b][e] -= mean;
                var += acc[b][e] * acc[b][e];
            }
        var += __shfl_xor(var, 32);
        const float rstd = 1.0f / sqrtf(var * (1.0f / 256.0f) + 1e-5f);
        float* yp = y + row * ldy + 4 * half;
        const float* gp = gamma + 4 * half;
        const float* bp = beta + 4 * half;
        // opaque on purpose: 256 loop-invariant floats per lane would otherwise be hoisted out of the tile loop and
        // parked in scratch for the whole kernel (they are L1/L2 hits once per 24 576 MFMAs here)
        asm volatile("" : "+v"(gp), "+v"(bp));
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 g4 = ld4(gp + 32 * b + 8 * i4), b4 = ld4(bp + 32 * b + 8 * i4);
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = acc[b][4 * i4 + k] * rstd * g4[k] + b4[k];
                if (!(T_ABLATE & 16) || o[0] + o[1] + o[2] + o[3] == 123.456f) *reinterpret_cast<f32x4*>(yp + 32 * b + 8 * i4) = o;
            }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // the two stages requested past the end must have landed before the LDS is released
}

// W1 [1024][256], W2 [256][1024] fp32 -> the stage images, [64][3 planes][16 fragments][64 lanes][8] bf16, in the
// order the kernel consumes them: W1_0 | W1_c, W2_{c-1} (c = 1 .. 31) | W2_31.  One thread per (stage, fragment, lane).
__global__ void pack_ffn_kernel(const float* __restrict__ W1, const float* __restrict__ W2, __bf16* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FFN_STAGES * 16 * 64) return;
    const int lane = t & 63, frag = (t >> 6) & 15, stage = t >> 10;
    const int m = lane & 31, half = lane >> 5;
    const bool up = stage == 0 || (stage < 63 && (stage & 1));
    const int c = stage == 0 ? 0 : stage == 63 ? 31 : up ? (stage + 1) / 2 : stage / 2 - 1;
    float v[8];
    if (up) {  // W1 chunk c: A-operand row m = hidden unit 32 c + m; fragment s = 2 blk + s2 covers features 32 blk + chunk_k
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W1[(int64_t)(32 * c + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
    } else {  // W2 chunk c: fragment = 2 blk + s2, A-operand row m = output feature 32 blk + m, k = hidden unit 32 c + chunk_k
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W2[(int64_t)(32 * (frag >> 1) + m) * 1024 + 32 * c + chunk_k(frag & 1, half, j)];
    }
    bf16x8 p0, p1, p2;
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    split3(lo, hi, p0, p1, p2);
    __bf16* dst = out + ((int64_t)stage * 3 * 16 + frag) * 64 * 8 + lane * 8;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + 16 * 64 * 8) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * 16 * 64 * 8) = p2;
}

}  // namespace

extern "C" int64_t scream_ffn_image_bytes(void) { return (int64_t)FFN_STAGES * T_STAGE; }

extern "C" int scream_pack_ffn_x3(const float* W1, const float* W2, void* image, void* stream) {
    SCREAM_REQUIRE(W1 && W2 && image, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, SCREAM_EINVAL);
    pack_ffn_kernel<<<dim3(FFN_STAGES * 16 * 64 / 256), dim3(256), 0, as_stream(stream)>>>(W1, W2, reinterpret_cast<__bf16*>(image));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_ffn_x3_f32(const float* m1, int64_t ldm, const void* ffn_image, const float* residual, int64_t ldr,
                                 const float* gamma, const float* beta, float* y, int64_t ldy, int64_t M, void* stream) {
    SCREAM_REQUIRE(m1 && ffn_image && residual && gamma && beta && y, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(ldm >= SCREAM_D_MODEL && ldr >= SCREAM_D_MODEL && ldy >= SCREAM_D_MODEL && ldm % 4 == 0 && ldr % 4 == 0 && ldy % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(ffn_image) | reinterpret_cast<uintptr_t>(residual) |
                     reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, SCREAM_EINVAL);
    const int64_t tiles = M / SCREAM_ROW_TILE;
    if (tiles == 0) return 0;
    SCREAM_REQUIRE(tiles < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = tiles < T_MAX_GRID ? (unsigned)tiles : (unsigned)T_MAX_GRID;
    ffn_x3_kernel<<<dim3(grid), dim3(TT), 0, as_stream(stream)>>>(m1, ldm, reinterpret_cast<const __bf16*>(ffn_image), residual, ldr, gamma,
                                                                 beta, y, ldy, (int)tiles);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
