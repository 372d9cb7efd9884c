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
#ifndef T_PF
#define T_PF 3  // register sets of weight fragments in tail_x3_kernel: fragments are read T_PF - 1 MFMA groups ahead
#endif
#include <type_traits>

#include "ring_x3.h"

// -DT_STAMPS (tools/tail_stamps.py): s_memtime stamps of the phases of the SECOND tile of every block, lane 0 of each wave.
// Diagnostic build only -- run tools/asm_inflight_check.py on it first: the extra registers can push hipcc into spilling
// a pending load destination (it did, with one stamp per stage).
#ifdef T_STAMPS
#define T_STAMP_SLOTS 24  // 0-5: phase boundaries (64-bit s_memtime); 8-23: low words of the stamps taken at stage tops (TMARK)
__device__ long long t_stamps[256 * 4 * T_STAMP_SLOTS];
extern "C" int scream_tail_stamps_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(t_stamps), sizeof(long long) * 256 * 4 * T_STAMP_SLOTS);
}
#define TSTAMP(slot)                                                                                 \
    do {                                                                                             \
        if (stamp_on && lane == 0) t_stamps[((int)blockIdx.x * 4 + wave) * T_STAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// a stamp that stays in a scalar register until the tile's end: no memory instruction inside the stages
#define TMARK(i) marks[i] = (unsigned)__builtin_amdgcn_s_memtime()
#define TMARK2(i, prev) do { marks[prev] = marks[i]; TMARK(i); } while (0)
#define TMARKS_FLUSH()                                                                               \
    do {                                                                                             \
        if (stamp_on && lane == 0)                                                                   \
            for (int i_ = 0; i_ < 16; ++i_) t_stamps[((int)blockIdx.x * 4 + wave) * T_STAMP_SLOTS + 8 + i_] = marks[i_]; \
    } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#define TMARK(i) do {} while (0)
#define TMARK2(i, prev) do {} while (0)
#define TMARKS_FLUSH() do {} while (0)
#endif

namespace {

constexpr int FFN_STAGES = 64;        // W1 chunk c -> stage 2c, W2 chunk c -> stage 2c + 1

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
        dma_1k(w_lane + (size_t)src * T_STAGE + (wave * 12 + (u & ~3)) * 1024, smem + slot * T_STAGE + (wave * 12 + (u & ~3)) * 1024, u & 3);
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


// =================================================================================================================
// The whole row-local tail of the block: attention apply (models/transformer.py:41-42), merge + norm1 (:83-84) and the
// FFN + norm2 above, one launch.  Per 128-row tile the weight ring carries 72 stages: Wm head 0..7, then the 64 FFN
// stages.  Merge stage h:  att_h^T = (KV_h^T . Q'_h^T) * Z * S  is a 32 x 32 x 32 product whose accumulator tile is,
// again, exactly the B operand of  m^T += Wm[:, head h] . att_h^T;  the apply of head h + 1 (12 MFMAs, Z, split) rides
// under the 96 MFMAs of head h.  LayerNorm1 turns the merge accumulators (+ x) into the bf16 planes of m1 in place.
// Neither att nor m1 nor the hidden activations exist in memory: per row and layer the kernel reads Q' and x (twice)
// and writes y -- 4 KB against the 15 KB of apply + merge + FFN-up + FFN-down.
//
// Row operands.  Lane (r, half) needs, of every 128-byte segment of row r, the 16-byte pieces 2a + half (the layout
// convention above).  From a row-major matrix that is a row-per-lane access -- 32 to 64 distinct lines per wave
// instruction, ~400 cycles of the CU's texture unit each, a quarter of the kernel however the requests were spread
// (tools/tail_ablate.py, profiles/r02_tail_ablation_*.txt); staged through a wave-private LDS slab by LDS-DMA the lines
// were full but, with room for one 4 KiB slab per wave, every request had half a stage of lead and the HBM latency showed
// instead (+3 % only).  So the kernels agree on the layout in HBM: Q', x and y are FRAGMENT-major (SCREAM_ACT_FRAG,
// include/scream_hip.h) -- per 32-row group and 32-feature segment the pieces are stored [a][lane], i.e. each of a lane's
// four loads or stores per segment is one contiguous 1 KiB wave access that lands in exactly the registers the MFMA
// wants, with no LDS in between.  All row operands are inline-asm register loads (hipcc would otherwise wait vmcnt(0)
// at their first use and drain the weight ring), requested one stage ahead and BEFORE the stage's weight pieces so that
// the ring's counted wait covers them as well.
constexpr int TAIL_STAGES = 72;
constexpr int KV_PLANES_BYTES = 8 * 3 * 2 * 1024;            // per cloud: [head][plane][step][lane][8] bf16
constexpr int KV_IMAGE_BYTES = KV_PLANES_BYTES + 8 * 32 * 4;  // + Ksum [head][32] fp32

struct HeadOps {   // the per-cloud operands of one head's apply, as loaded (Q' travels separately: f32x4 q[4], pieces a = 0 .. 3)
    f32x4 kv[6];   // KV_h^T fragments [plane][step], 16 bytes per lane
    f32x4 ks[4];   // Ksum[h][8 a + 4 half .. + 4]
};

// merge stage h (h >= 1): which quarter of the x-segment add rides in group g (-1: none) -- the last four groups of
// 14 .. 8 that do not accumulate into tile h - 1
__device__ __forceinline__ constexpr int xadd_slot(int h, int g) {
    int n = 0;
    for (int c = 14; c >= 8; --c) {
        if ((c >> 1) == h - 1) continue;
        if (c == g) return n < 4 ? n : -1;
        ++n;
    }
    return -1;
}

// ---- T_MFMA16 (build option, SCREAM_TAIL_MFMA16=1; off by default): the tail kernel on v_mfma_f32_16x16x32_bf16 -----------
// A pure MFMA stream of the 16x16x32 shape sustains 12 % more flops at the socket power cap than the 32x32x16 one
// (tools/ubench/mfma_energy.py).  In this kernel it does not pay: same results to rounding (the whole GPU suite passes on
// it), the clock goes from 1.86 to 2.19 GHz, but the stages take 17 % more cycles (two accumulator tiles per step instead of
// one: half the distance between dependent MFMAs, 16-cycle MFMAs hide half as many ride instructions) and a launch costs the
// same joules -- 1.729 vs 1.742 ms per 333 k-row launch, 2.28 J both (profiles/r02_tail_mfma16_vs_32.txt).
// The kernel keeps its data layouts -- fragment-major activations in HBM, "lane = row r of 32, lane half = features + 4" in
// the VALU code (E form) -- and changes form only around the MFMAs:
//   * B operand of a 16x16x32: lane (n = l & 15, kgroup = l >> 4) holds 8 contraction values of row n of a 16-row sub-tile.
//     Two E-form plane registers R0, R1 (the two 16-deep steps of a 32-wide block) become the operands of the two sub-tiles
//     with ONE v_permlane16_swap per dword: X0 = [R0.q0 R1.q0 R0.q2 R1.q2] (rows 0-15), X1 = [R0.q1 R1.q1 R0.q3 R1.q3].
//   * its result (lane (n, q): 4 registers = output rows 4q + i of a 16 x 16 tile) comes back to the E form by the same swap
//     between the two sub-tiles' registers; which output feature sits where is a fixed permutation baked into the weight
//     images (pack_tail_kernel, kv_finalize_x3_kernel), as the contraction order is (tools/ubench/mfma16_layout.hip).
// A step g of a stage (3 fragment reads, 192 cycles of MFMAs) is unchanged in shape: 12 MFMAs (2 sub-tiles x 6 products) on
// the fragment of (32-deep block, 16-feature half) instead of 6 on (16-deep step, 32 features).
#ifndef T_MFMA16
#define T_MFMA16 0
#endif
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// feature (0 .. 15) of a 16-feature half that row m of the A operand must carry so that the swapped-back accumulator is in
// the layout convention (register 4a + b of lane half h = feature 8a + 4h + b of the 32-block)
__host__ __device__ __forceinline__ int perm16(int m) { return (m & 3) | ((m & 4) << 1) | ((m & 8) >> 1); }
// contraction index (0 .. 31) that element j of lane group kg of a swapped B operand holds
__host__ __device__ __forceinline__ int k_of16(int kg, int j) { return chunk_k(kg & 1, kg >> 1, j); }

// v_permlane16_swap_b32 as an asm statement: through __builtin_amdgcn_permlane16_swap hipcc 7.2 folded sixteen swaps of
// lane-affine values into three and copied the results around (tools/ubench/swap_rt.hip: every component became component 0
// of the partner row).  Two swaps per statement; two wait states in front (a VALU result feeding a permlane swap) and
// behind (its results feeding a VALU or matrix instruction) -- hipcc pads neither side of an asm statement.
__device__ __forceinline__ void swap16x2(unsigned& a0, unsigned& a1, unsigned& b0, unsigned& b1) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %2\n\t"
                 "v_permlane16_swap_b32 %1, %3\n\t"
                 "s_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
}
__device__ __forceinline__ void swap16x4(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned& b0, unsigned& b1,
                                         unsigned& b2, unsigned& b3) {  // (two statements of two: eight tied registers at once made hipcc
    swap16x2(a0, a1, b0, b1);                                            //  park pending load destinations elsewhere)
    swap16x2(a2, a3, b2, b3);
}
// (R0, R1) <-> (X0, X1) for the three planes of a 32-deep block (an involution)
__device__ __forceinline__ void swap_planes(bf16x8 (&a)[3], bf16x8 (&b)[3]) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4_t ua = __builtin_bit_cast(u32x4_t, a[p]), ub = __builtin_bit_cast(u32x4_t, b[p]);
        unsigned x0 = ua[0], x1 = ua[1], x2 = ua[2], x3 = ua[3], y0 = ub[0], y1 = ub[1], y2 = ub[2], y3 = ub[3];
        swap16x4(x0, x1, x2, x3, y0, y1, y2, y3);
        ua = u32x4_t{x0, x1, x2, x3};
        ub = u32x4_t{y0, y1, y2, y3};
        a[p] = __builtin_bit_cast(bf16x8, ua);
        b[p] = __builtin_bit_cast(bf16x8, ub);
    }
}
__device__ __forceinline__ void swap_f4(float& a0, float& a1, float& a2, float& a3, float& b0, float& b1, float& b2, float& b3) {
    unsigned x0 = __builtin_bit_cast(unsigned, a0), x1 = __builtin_bit_cast(unsigned, a1), x2 = __builtin_bit_cast(unsigned, a2),
             x3 = __builtin_bit_cast(unsigned, a3), y0 = __builtin_bit_cast(unsigned, b0), y1 = __builtin_bit_cast(unsigned, b1),
             y2 = __builtin_bit_cast(unsigned, b2), y3 = __builtin_bit_cast(unsigned, b3);
    swap16x4(x0, x1, x2, x3, y0, y1, y2, y3);
    a0 = __builtin_bit_cast(float, x0); a1 = __builtin_bit_cast(float, x1); a2 = __builtin_bit_cast(float, x2); a3 = __builtin_bit_cast(float, x3);
    b0 = __builtin_bit_cast(float, y0); b1 = __builtin_bit_cast(float, y1); b2 = __builtin_bit_cast(float, y2); b3 = __builtin_bit_cast(float, y3);
}
// a 32 x 32 accumulator tile between the E form and the four 16 x 16 tiles (sub-tile, feature half) at registers
// 4 (2 fb + sub) .. + 3; also a 4 x f32x4 segment of x or Q' (same register numbering)
__device__ __forceinline__ void swap_tile(f32x16& t) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float a0 = t[8 * fb], a1 = t[8 * fb + 1], a2 = t[8 * fb + 2], a3 = t[8 * fb + 3];
        float b0 = t[8 * fb + 4], b1 = t[8 * fb + 5], b2 = t[8 * fb + 6], b3 = t[8 * fb + 7];
        swap_f4(a0, a1, a2, a3, b0, b1, b2, b3);
        t[8 * fb] = a0; t[8 * fb + 1] = a1; t[8 * fb + 2] = a2; t[8 * fb + 3] = a3;
        t[8 * fb + 4] = b0; t[8 * fb + 5] = b1; t[8 * fb + 6] = b2; t[8 * fb + 7] = b3;
    }
}
__device__ __forceinline__ void swap_seg(f32x4 (&x)[4]) {
    if (!T_MFMA16) return;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float a0 = x[2 * fb][0], a1 = x[2 * fb][1], a2 = x[2 * fb][2], a3 = x[2 * fb][3];
        float b0 = x[2 * fb + 1][0], b1 = x[2 * fb + 1][1], b2 = x[2 * fb + 1][2], b3 = x[2 * fb + 1][3];
        swap_f4(a0, a1, a2, a3, b0, b1, b2, b3);
        x[2 * fb] = f32x4{a0, a1, a2, a3};
        x[2 * fb + 1] = f32x4{b0, b1, b2, b3};
    }
}

// One step of a stage: acc (32 features x 32 rows) += W fragment . operand, six products.  32x32x16: w = the fragment of 16-deep
// step `sel` of the block, operand = (sel ? b1 : b0).  16x16x32: w = the fragment of feature half `sel` over the whole 32-deep
// block, operands b0 / b1 = the two row sub-tiles (swap_planes), results into the tiles (sub, sel).
template <int NV = 0>
__device__ __forceinline__ void mfma_step(f32x16& acc, const bf16x8 (&w)[3], const bf16x8 (&b0)[3], const bf16x8 (&b1)[3], int sel,
                                          bool zero = false) {
#if T_MFMA16
    f32x4 d0, d1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d0[i] = zero ? 0.f : acc[8 * sel + i];
        d1[i] = zero ? 0.f : acc[8 * sel + 4 + i];
    }
#define M16(D, A, B) D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, D, 0, 0, 0)
    M16(d0, w[0], b0[2]); M16(d1, w[0], b1[2]);
    M16(d0, w[1], b0[1]); M16(d1, w[1], b1[1]);
    M16(d0, w[2], b0[0]); M16(d1, w[2], b1[0]);
    M16(d0, w[0], b0[1]); M16(d1, w[0], b1[1]);
    M16(d0, w[1], b0[0]); M16(d1, w[1], b1[0]);
    M16(d0, w[0], b0[0]); M16(d1, w[0], b1[0]);
#undef M16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[8 * sel + i] = d0[i];
        acc[8 * sel + 4 + i] = d1[i];
    }
    if (NV >= 0) {  // first MFMA, the three prefetch reads of the next fragment group, the other eleven MFMAs (with ride slots)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (NV == 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, (NV + 1) / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x002, (NV + 1) / 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    if (NV >= 0) mfma6<(NV > 0 ? NV : 0)>(acc, w, sel ? b1 : b0, zero);
    else mfma6_free(acc, w, sel ? b1 : b0, zero);
#endif
}

__global__ __launch_bounds__(TT, 1) void tail_x3_kernel(const float* __restrict__ Q,      // fragment-major [M, 256]
                                                        const char* __restrict__ kvimg,   // [n_clouds][KV_IMAGE_BYTES]
                                                        const int32_t* __restrict__ tile_cloud, int kv_cloud_offset,
                                                        const int32_t* __restrict__ cloud_len,
                                                        const float* __restrict__ xres,   // fragment-major [M, 256]
                                                        const __bf16* __restrict__ Wimg,  // [72 stages][48 KiB]
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        const float* __restrict__ g2, const float* __restrict__ b2,
                                                        float* __restrict__ y,            // fragment-major [M, 256]
                                                        int n_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[T_SLOTS * T_STAGE + 4096];  // 144 KiB ring + the norm parameters, the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // gamma1 | beta1 | gamma2 | beta2 live in LDS for the whole launch.  Read from global memory inside the norm blocks they
    // cost more than the arithmetic: on gfx950 loads and stores share vmcnt, hipcc cannot order a load against the y stores
    // issued before it and waits with vmcnt(0) -- a full store round trip per group of rows (tools/tail_stamps.py: 11.8 k
    // cycles for norm2 + stores).  From LDS the y stores are fire-and-forget.
    float* lnp = reinterpret_cast<float*>(smem + T_SLOTS * T_STAGE);
    lnp[tid] = g1[tid];
    lnp[256 + tid] = b1[tid];
    lnp[512 + tid] = g2[tid];
    lnp[768 + tid] = b2[tid];
    const unsigned v_lane16 = lane * 16, v_half16 = half * 16;  // the only per-lane address parts of the kernel
    // weight pieces: uniform (scalar) source address + the 32-bit lane offset.  A per-lane 64-bit pointer kept across the
    // kernel was spilled by hipcc and reloaded from scratch in EVERY stage -- behind a vmcnt(0) that drained the ring.
    auto dma_piece = [&](unsigned q, int u) {
        if ((T_ABLATE & 1) && q >= 2) return;
        const unsigned src = q % (unsigned)TAIL_STAGES, slot = q % (unsigned)T_SLOTS;
        const char* sbase = reinterpret_cast<const char*>(Wimg) + (size_t)src * T_STAGE + (wave * 12 + (u & ~3)) * 1024;
        dma_1k(sbase + v_lane16, smem + slot * T_STAGE + (wave * 12 + (u & ~3)) * 1024, u & 3);
    };
    unsigned q = 0;  // next stage to be consumed
#pragma unroll
    for (int u = 0; u < 12; ++u) dma_piece(0, u);
#pragma unroll
    for (int u = 0; u < 12; ++u) dma_piece(1, u);

    // ---- row operand requests (inline asm; every consumer sits behind a counted wait + pin) -------------------------
    // RULE (tools/asm_inflight_check.py enforces it on the generated code): a requested register is consumed at the top
    // of the NEXT stage, never kept pending across a LayerNorm block -- hipcc, which believes the value present, otherwise
    // spills it to scratch right behind the asm statement when registers are short there.
    // grp: first float of the wave's 32-row group (8192 floats in either layout); segment seg, piece a, this lane:
    // the uniform part of the address of segment seg in the 32-row group starting at float `grp`
    auto seg_base = [&](const float* base, int64_t grp, int seg) { return base + grp + seg * 1024; };
    auto req_q = [&](f32x4 (&qb)[4], int64_t grp, int h) {  // Q' of head h: the one operand that comes from HBM
        if (T_ABLATE & (16 | 32)) return;
        if (T_ABLATE & 2048) grp = (int64_t)wave * 32 * SCREAM_D_MODEL;  // tuning aid: always the first tile's rows (cache hits)
        ld_asm4<1024>(qb, seg_base(Q, grp, h), v_lane16);
    };
    auto req_head = [&](HeadOps& o, const char* kvc, int h) {  // KV^T fragments and Ksum of head h: L2-hot per-cloud data
        if (T_ABLATE & (16 | 256)) return;
        const char* kp = kvc + h * (3 * 2 * 1024);
        f32x4 (&kv4)[4] = reinterpret_cast<f32x4 (&)[4]>(o.kv[0]);
        ld_asm4<1024>(kv4, kp, v_lane16);
        ld_asm2k(o.kv[4], o.kv[5], kp + 4 * 1024, v_lane16);
        ld_asm4<32>(o.ks, kvc + KV_PLANES_BYTES + 128 * h, v_half16);
    };
    auto req_x = [&](f32x4 (&xs)[4], int64_t grp, int blk) {
        if (T_ABLATE & (16 | 64)) return;
        if (T_ABLATE & 2048) grp = (int64_t)wave * 32 * SCREAM_D_MODEL;
        ld_asm4<1024>(xs, seg_base(xres, grp, blk), v_lane16);
    };
    auto pin_head = [&](HeadOps& o) {
#pragma unroll
        for (int a = 0; a < 4; ++a) pin(o.ks[a]);
#pragma unroll
        for (int f = 0; f < 6; ++f) pin(o.kv[f]);
    };
    auto add_x4 = [&](f32x16& t, const f32x4 (&xs)[4], int a) {  // one quarter (registers 4a .. 4a + 3)
#pragma unroll
        for (int k = 0; k < 4; ++k) t[4 * a + k] += (T_ABLATE & 16) ? 1.0f : xs[a][k];
    };
    auto pin_x = [&](f32x4 (&xs)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a) pin(xs[a]);
    };
    auto add_x = [&](f32x16& t, const f32x4 (&xs)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[4 * a + k] += (T_ABLATE & 16) ? 1.0f : xs[a][k];
    };

    // KV^T / Ksum: ONE buffer, consumed in a stage's first groups and re-requested right after (a third of a stage of
    // lead is plenty for L2-hot data).  Q' comes from HBM and gets TWO buffers, alternating by head, requested at the
    // top of the stage BEFORE the one that consumes it -- with a single buffer the merge stages took 6.6 k cycles against
    // 3.8 k for an FFN stage (tools/tail_stamps.py).
    HeadOps op;
    f32x4 qA[4], qB[4];     // Q' of even / odd heads
    f32x4 xs[4], xs2[4];    // x segments (xs2: only segment 7 of the norm1 residual)
    bf16x8 apA[2][3], apB[2][3];  // planes of att_h^T, the B operand of the merge GEMM: heads of even / odd index
    f32x16 aT;              // att_h^T tile of the head being applied
    bf16x8 qp[2][3];
    float Zs = 0.f;

    // ---- apply of one head (models/transformer.py:41-42), in pieces that ride inside the groups of another stage ----
    auto apply_qsplit = [&](f32x4 (&qb)[4], int tile_tag, int s2) {  // 16-deep step s2 of Q' into its three planes
        if (T_ABLATE & 16) {
            const f32x4 f = {(float)lane, 1.0f, 0.5f, (float)tile_tag};
            split3(f, f, qp[s2][0], qp[s2][1], qp[s2][2]);
            if (s2 == 1) swap_planes(qp[0], qp[1]);
            return;
        }
        split3(qb[2 * s2], qb[2 * s2 + 1], qp[s2][0], qp[s2][1], qp[s2][2]);
        if (s2 == 1) swap_planes(qp[0], qp[1]);
    };
    auto apply_mfma = [&](int s2) {
        bf16x8 w[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) w[p] = __builtin_bit_cast(bf16x8, op.kv[p * 2 + s2]);
        mfma_step<-1>(aT, w, qp[0], qp[1], s2, T_MFMA16 ? true : s2 == 0);
        if (s2 == 1) swap_tile(aT);  // back to the E form for the scaling and the split
    };
    auto apply_z = [&](f32x4 (&qb)[4]) {  // Z = 1 / (Q'.Ksum + 1e-6); lanes r and r + 32 share row r
        float zp = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) zp += qb[a][k] * op.ks[a][k];
        zp += __shfl_xor(zp, 32);
        Zs = 1.0f / (zp + 1e-6f);
    };
    auto apply_split_pair = [&](int k, bf16x8 (&ap)[2][3], float S) {  // elements 2k, 2k+1: (aT * Z) * S, then the 3-way split
        const int s2 = k >> 2, j = (2 * k) & 7;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float x = (aT[2 * k + e] * Zs) * S;
            const __bf16 a = (__bf16)x;
            const float r1 = x - (float)a;
            const __bf16 b = (__bf16)r1;
            ap[s2][0][j + e] = a;
            ap[s2][1][j + e] = b;
            ap[s2][2][j + e] = (__bf16)(r1 - (float)b);
        }
        if (k == 7) swap_planes(ap[0], ap[1]);
    };
    // the pieces of one apply as they ride in group g of a 16-group stage: operands consumed in groups 0-3
    auto apply_ride = [&](f32x4 (&qb)[4], int g, bf16x8 (&ap)[2][3], float S, int tile_tag) {
        if (g == 0) apply_qsplit(qb, tile_tag, 0);
        if (g == 1) apply_qsplit(qb, tile_tag, 1);
        if (g == 1) apply_mfma(0);
        if (g == 2) apply_mfma(1);
        if (g == 3) apply_z(qb);
        if (g >= 4 && g < 12) apply_split_pair(g - 4, ap, S);
    };

    int tile = blockIdx.x;
    int64_t grp = ((int64_t)tile * 128 + wave * 32) * SCREAM_D_MODEL;  // first float of this wave's 32-row group
    const char* kvc = nullptr;
    float S = 1.f;
    if (tile < n_tiles) {  // the block's first tile: heads 0 and 1 are applied in the open (once per block)
        const int cl = tile_cloud[tile] + kv_cloud_offset;
        kvc = kvimg + (size_t)cl * KV_IMAGE_BYTES;
        S = (float)cloud_len[cl];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            req_q(qA, grp, h);
            req_head(op, kvc, h);
            VM_WAIT(0);
            pin_x(qA);
            pin_head(op);
#pragma unroll
            for (int g = 0; g < 12; ++g) apply_ride(qA, g, h == 0 ? apA : apB, S, tile);
        }
    }

#ifdef T_STAMPS
    int tile_no = 0;
#endif
    while (tile < n_tiles) {
#ifdef T_STAMPS
        const bool stamp_on = tile_no == 1;
        ++tile_no;
#endif
#ifdef T_STAMPS
        unsigned marks[16] = {};
#endif
        TSTAMP(0);  // tile start
        // The last MFMA group of every stage is DEFERRED across the barrier: its weight fragments are read into wfd, and
        // the next stage issues it right after the first fragment reads of its own -- 6 MFMAs (192 cycles) of work with
        // register operands exactly where a lone in-order wave otherwise waits for the LDS (tools/tail_stamps.py: 3.8 k
        // cycles per 3.07 k-cycle stage).  `flush` arguments below name the deferred group of the preceding stage.
        bf16x8 wfd[3];
        f32x16 acc[8];  // (started by the first product of merge stage 0 / of the first down stage: no zeroing moves)
        // the block's next tile (its heads 0 and 1 are applied under / right after this tile's last stage)
        const int tile_next = tile + (int)gridDim.x;
        const bool has_next = tile_next < n_tiles;
        // (the last tile of a block "requests" its own operands again instead of branching around the requests: an asm load
        // that is skipped on one path leaves the buffer's OLD contents live across the whole FFN phase -- 40 spilled registers)
        const int64_t grp_next = ((int64_t)(has_next ? tile_next : tile) * 128 + wave * 32) * SCREAM_D_MODEL;
        const char* kvc_next = kvc;
        float S_next = 1.f;
        if (has_next) {
            const int cl = tile_cloud[tile_next] + kv_cloud_offset;
            kvc_next = kvimg + (size_t)cl * KV_IMAGE_BYTES;
            S_next = (float)cloud_len[cl];
        }

        // merge stage of head h: m^T += Wm[:, head h] . att_h^T with att_h's planes in `ap` (heads 0, 1: applied during the
        // previous tile).
        //   top:         x segment h - 1 (the norm1 residual, requested by stage h - 1) is added into accumulator tile h - 1,
        //                then segment h is requested (stage 6 also requests segment 7, stage 7 adds both);
        //   groups 0-3:  stages 1 .. 6 consume the operands of head h + 1 (requested by stage h - 1) -- its apply rides here;
        //   group 4:     the operand buffer is re-requested for head h + 2 (stage 0: requested at the top);
        //   groups 4-15: the stage's weight pieces, AFTER every row request, so that the next barrier's vmcnt(12) covers them.
        auto stage_merge = [&](auto hh, bf16x8 (&ap)[2][3], bf16x8 (&ap_next)[2][3], f32x4 (&q_cons)[4], f32x4 (&q_req)[4], auto flush) {
            constexpr int h = decltype(hh)::value;
            constexpr bool RIDE = h >= 1 && h <= 6;
            // stage 0 of a tile: everything older was drained at the end of the previous tile (only its y stores may
            // still be in flight, and nothing of this stage depends on them)
            if (h == 0) lds_only_barrier(); else ring_barrier<12>();
            TMARK(h);  // T_STAMPS builds: marks 0-7 = the tops of the merge stages
            __builtin_amdgcn_sched_barrier(0);
            f32x4 (&x_prev)[4] = (h & 1) ? xs : xs2;   // segment h - 1 (landed: requested by stage h - 1)
            f32x4 (&x_req)[4] = (h & 1) ? xs2 : xs;    // segment h
            if (h > 0) {
                pin_x(x_prev);
                swap_seg(x_prev);  // (16x16x32: the accumulators are in the MFMA's form while they accumulate)
                if (RIDE) {
                    pin_head(op);
                    pin_x(q_cons);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            req_x(x_req, grp, h);
            if (h + 2 < 8) req_q(q_req, grp, h + 2);  // consumed by stage h + 1
            if (h == 0) req_head(op, kvc, 2);
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[T_PF][3];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < 15; ++g) {  // g = blk * 2 + s2; group 15 is deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        (g + T_PF - 1 == 15 ? wfd[p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (g == 4 && RIDE && h + 2 < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    req_head(op, kvc, h + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g >= 4) dma_piece(q + 2, g - 4);
                if (g == 14) dma_piece(q + 2, 11);  // (group 15 is deferred: the twelfth piece goes out with the eleventh)
                if (RIDE && !(T_ABLATE & 512)) apply_ride(q_cons, g, ap_next, S, tile);
                // the norm1 residual: segment h - 1 joins accumulator tile h - 1 in quarters, in late groups that do not
                // accumulate into that tile (its MFMAs are groups 2h - 2 and 2h - 1)
                if (h > 0) {
                    const int pc = xadd_slot(h, g);
                    if (pc >= 0 && !(T_ABLATE & 1024)) add_x4(acc[h > 0 ? h - 1 : 0], x_prev, pc);
                }
                mfma_step<(h > 0 ? 6 : 0)>(acc[g >> 1], wf[g % T_PF], ap[0], ap[1], g & 1, h == 0 && (T_MFMA16 || (g & 1) == 0));
            }
            ++q;
        };
        constexpr std::integral_constant<bool, true> yes{};
        constexpr std::integral_constant<bool, false> no{};
#define HEAD(n) std::integral_constant<int, n>{}
        auto flush_none = [&]() {};
        auto flush_mergeA = [&]() { mfma_step<-1>(acc[7], wfd, apA[0], apA[1], 1); };  // deferred group of a merge stage of an even head
        auto flush_mergeB = [&]() { mfma_step<-1>(acc[7], wfd, apB[0], apB[1], 1); };
        auto flush_mergeA0 = [&]() { mfma_step<-1>(acc[7], wfd, apA[0], apA[1], 1, T_MFMA16 != 0); };  // ... of stage 0 (16x16x32: its tiles start there)
        //            head   planes  planes of head + 1   Q' consumed (head + 1)   Q' requested (head + 2)   deferred group of
        stage_merge(HEAD(0), apA, apB, qB, qA, flush_none);   // (the previous tile ended flushed)
        stage_merge(HEAD(1), apB, apA, qA, qB, flush_mergeA0);
        stage_merge(HEAD(2), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(3), apB, apA, qA, qB, flush_mergeA);
        stage_merge(HEAD(4), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(5), apB, apA, qA, qB, flush_mergeA);
        stage_merge(HEAD(6), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(7), apB, apA, qA, qB, flush_mergeA);
        flush_mergeB();  // norm1 needs the finished accumulators
        VM_WAIT(12);     // x segment 7 (requested at the top of stage 7, older than that stage's twelve weight pieces)
        pin_x(xs2);
        swap_seg(xs2);
        add_x(acc[7], xs2);
#pragma unroll
        for (int b = 0; b < 8; ++b) swap_tile(acc[b]);  // the norm works on the E form

        TSTAMP(1);  // end of the merge phase
        // ---- m1 = LayerNorm1(merge + x) (models/transformer.py:84), straight into the B-operand planes of FFN-up ---
        bf16x8 mp[16][3];
        {
            float sum = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[b][e];
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
            const float* gp = lnp + 4 * half;
            const float* bp = lnp + 256 + 4 * half;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f32x4 v[2];
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const int a = 2 * s2 + a2;
                        const f32x4 g4 = ld4(gp + 32 * b + 8 * a), b4 = ld4(bp + 32 * b + 8 * a);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[a2][k] = acc[b][4 * a + k] * rstd * g4[k] + b4[k];
                    }
                    split3(v[0], v[1], mp[2 * b + s2][0], mp[2 * b + s2][1], mp[2 * b + s2][2]);
                    if (s2 == 1) swap_planes(mp[2 * b], mp[2 * b + 1]);
                }
        }

        TSTAMP(2);  // end of norm1
        // ---- FFN (as ffn_x3_kernel); x segments 0 .. 7 (the norm2 residual) are added under the first eight down stages
        f32x16 hT;
        bf16x8 hpA[2][3], hpB[2][3];
        auto split_pair = [&](int k, bf16x8 (&hout)[2][3]) {
            const int s2 = k >> 2, j = (2 * k) & 7;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x = fmaxf(hT[2 * k + e], 0.f);
                const __bf16 a = (__bf16)x;
                const float r1 = x - (float)a;
                const __bf16 b = (__bf16)r1;
                hout[s2][0][j + e] = a;
                hout[s2][1][j + e] = b;
                hout[s2][2][j + e] = (__bf16)(r1 - (float)b);
            }
            if (k == 7) swap_planes(hout[0], hout[1]);
        };
        // XLOAD: x segment to request in this stage (-1: none)
        auto stage_up = [&](auto first, auto xload, auto flush) {
            constexpr int XLOAD = decltype(xload)::value;
            // FIRST: the norm1 block above used ordinary loads (gamma, beta), which hipcc waits for with vmcnt(0)
            if (decltype(first)::value) ring_barrier<0>(); else ring_barrier<12>();
            TMARK2(8, 10);  // T_STAMPS builds: tops of the last two up stages
            __builtin_amdgcn_sched_barrier(0);
            if (XLOAD >= 0) req_x(xs, grp, XLOAD);
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[T_PF][3];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < 15; ++g) {  // group 15 (hT += wfd . mp[15]) is deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        (g + T_PF - 1 == 15 ? wfd[p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (g < 12) dma_piece(q + 2, g);
                mfma_step(hT, wf[g % T_PF], mp[g & ~1], mp[g | 1], g & 1, T_MFMA16 ? g < 2 : g == 0);
            }
            ++q;
        };
        // XADD: x segment (requested by the previous stage) to add into its accumulator tile (-1: none).
        // RIDE (the tile's last two stages, when the FFN's operand planes are dead and registers are available again):
        // 1 = requests the operands of the next tile's head 0; 2 = the apply of that head rides here, and head 1 is
        // requested behind it (applied in the open right after the stage).
        auto stage_down = [&](bf16x8 (&hin)[2][3], bf16x8 (&hout)[2][3], auto with_split, auto xadd, auto ride, auto flush) {
            constexpr int XADD = decltype(xadd)::value;
            constexpr int RIDE = decltype(ride)::value;
            ring_barrier<12>();
            TMARK2(9, 11);  // ... and of the last two down stages
            __builtin_amdgcn_sched_barrier(0);
            if (XADD == 0) {  // the tile's first down stage starts the accumulators: tile 0 from its x segment, the others from 0
                pin_x(xs);
                swap_seg(xs);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[0][4 * a + k] = (T_ABLATE & 16) ? 1.0f : xs[a][k];
            } else if (XADD > 0) {
                pin_x(xs);
                swap_seg(xs);
                add_x(acc[XADD > 0 ? XADD : 0], xs);
            }
            if (RIDE == 2) {  // (untouched registers when the block has no next tile: the results are never used)
                pin_head(op);
                pin_x(qA);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (RIDE == 1) {
                req_q(qA, grp_next, 0);
                req_head(op, kvc_next, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[T_PF][3];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
            flush();
            if (decltype(with_split)::value) swap_tile(hT);  // (16x16x32: the relu / split below reads the E form)
#pragma unroll
            for (int g = 0; g < 15; ++g) {  // g = blk * 2 + s2; group 15 (acc[7] += wfd . hin[1]) is deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        (g + T_PF - 1 == 15 ? wfd[p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (RIDE == 2 && g == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    req_q(qA, grp_next, 1);  // head 1 into the buffers head 0 has just left (a second Q' buffer here, at the
                    req_head(op, kvc_next, 1);  // top of the stage, was spilled by hipcc right behind its loads)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (RIDE == 2) {
                    if (g >= 4) dma_piece(q + 2, g - 4);
                    if (g == 14) dma_piece(q + 2, 11);  // (group 15 is deferred: the twelfth piece goes out with the eleventh)
                } else {
                    if (g < 12) dma_piece(q + 2, g);
                }
                if (RIDE == 2) apply_ride(qA, g, apA, S_next, tile_next);
                if (decltype(with_split)::value && (g & 1) == 0) split_pair(g >> 1, hout);
                mfma_step(acc[g >> 1], wf[g % T_PF], hin[0], hin[1], g & 1, XADD == 0 && g >= 2 && (T_MFMA16 || (g & 1) == 0));
            }
            ++q;
        };
        constexpr std::integral_constant<int, -1> none{};
        constexpr std::integral_constant<int, 0> ride0{};
        // stage order (= image order): W1_0 | W1_c, W2_{c-1} for c = 1 .. 31 | W2_31.  The norm2 residual: the up stage of
        // chunk c requests x segment c - 1, the down stage of chunk c - 1 that follows adds it (c - 1 < 8) -- the first
        // four pair iterations are peeled so that every accumulator index is a compile-time constant.
        auto flush_up = [&]() { mfma_step<-1>(hT, wfd, mp[14], mp[15], 1); };            // deferred group of an up stage
        auto flush_downA = [&]() { mfma_step<-1>(acc[7], wfd, hpA[0], hpA[1], 1); };     // ... of a down stage whose operand was hpA
        auto flush_downB = [&]() { mfma_step<-1>(acc[7], wfd, hpB[0], hpB[1], 1); };
        auto flush_downA0 = [&]() { mfma_step<-1>(acc[7], wfd, hpA[0], hpA[1], 1, T_MFMA16 != 0); };  // ... of the tile's first down stage
        stage_up(yes, none, flush_none);
        flush_up();
        swap_tile(hT);
#pragma unroll
        for (int k = 0; k < 8; ++k) split_pair(k, hpA);
#define PAIR(c, fl, flA)                                                                    \
        stage_up(no, HEAD((c) - 1), fl);                           /* chunk c */             \
        stage_down(hpA, hpB, yes, HEAD((c) - 1), ride0, flush_up); /* chunk c - 1, splits c */ \
        stage_up(no, HEAD(c), flA);                                /* chunk c + 1 */         \
        stage_down(hpB, hpA, yes, HEAD(c), ride0, flush_up);       /* chunk c, splits c + 1 */
        PAIR(1, flush_none, flush_downA0) PAIR(3, flush_downB, flush_downA) PAIR(5, flush_downB, flush_downA) PAIR(7, flush_downB, flush_downA)
#undef PAIR
        for (int c = 9; c < 31; c += 2) {
            stage_up(no, none, flush_downB);
            stage_down(hpA, hpB, yes, none, ride0, flush_up);
            stage_up(no, none, flush_downA);
            stage_down(hpB, hpA, yes, none, ride0, flush_up);
        }
        stage_up(no, none, flush_downB);                           // chunk 31
        stage_down(hpA, hpB, yes, none, HEAD(1), flush_up);         // chunk 30, splitting chunk 31; requests head 0 of the block's next tile
        stage_down(hpB, hpA, no, none, HEAD(2), flush_downA);       // chunk 31; applies that head, requests head 1
        flush_downB();                                              // norm2 needs the finished accumulators
#undef HEAD
        // Drain: the ring's two stages in flight and head 1 of the next tile, requested two thirds of a stage ago.  That
        // head is applied here in the open (12 MFMAs), so that nothing pending lives across the norm2 block, and the y
        // stores below are YOUNGER than every load a later counted wait is meant to cover (stores retire out of order with
        // respect to loads, gemm_x3.hip; stage 0 of the next tile starts without a vector-memory wait).
        TSTAMP(3);  // end of the last stage
        VM_WAIT(0);
        if (has_next) {
            pin_head(op);
            pin_x(qA);
#pragma unroll
            for (int g = 0; g < 12; ++g) apply_ride(qA, g, apB, S_next, tile_next);
        }

        TSTAMP(4);  // after the open apply of the next tile's head 1
        // ---- y = LayerNorm2(x + ffn) (the residual is already in the accumulators), stored fragment-major ------------
#pragma unroll
        for (int b = 0; b < 8; ++b) swap_tile(acc[b]);  // (16x16x32: back to the E form)
        {
            float sum = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[b][e];
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
            const float* gp = lnp + 512 + 4 * half;
            const float* bp = lnp + 768 + 4 * half;
            float* yg = y + grp + lane * 4;  // (uniform base + lane: the stores below differ by immediates and scalar adds)
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 g4 = ld4(gp + 32 * b + 8 * a), b4 = ld4(bp + 32 * b + 8 * a);
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = acc[b][4 * a + k] * rstd * g4[k] + b4[k];
                    // one contiguous 1 KiB per wave instruction
                    if (!(T_ABLATE & (16 | 128)) || o[0] + o[1] + o[2] + o[3] == 123.456f) *reinterpret_cast<f32x4*>(yg + (b * 4 + a) * 256) = o;
                }
        }
        TSTAMP(5);  // tile end
        TMARKS_FLUSH();
        tile = tile_next;
        grp = ((int64_t)tile_next * 128 + wave * 32) * SCREAM_D_MODEL;
        kvc = kvc_next;
        S = S_next;
    }
    VM_WAIT(0);  // the two stages requested past the end must have landed before the LDS is released
}

// [M, 256] fp32 row-major <-> fragment-major (SCREAM_ACT_FRAG, include/scream_hip.h).  One block per 32-row group: the
// group is read in full lines, turned around in LDS and written in full lines.  Boundary use only (after the embedding,
// for tests and for callers that hold row-major data).
__global__ __launch_bounds__(256) void act_layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int to_frag) {
    __shared__ float tile[32][260];  // +4: the transposing accesses below touch rows 1040 bytes apart
    const int t = threadIdx.x;
    const float* s = src + (int64_t)blockIdx.x * 8192;
    float* d = dst + (int64_t)blockIdx.x * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = t + 256 * i;  // float4 index inside the group
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + e * 4);
        int row, col;
        if (to_frag) {
            row = e >> 6, col = (e & 63) * 4;  // source is row-major
        } else {
            const int ln = e & 63, a = (e >> 6) & 3, blk = e >> 8;  // source is fragment-major
            row = ln & 31, col = 32 * blk + 8 * a + 4 * (ln >> 5);
        }
        *reinterpret_cast<f32x4*>(&tile[row][col]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = t + 256 * i;
        int row, col;
        if (to_frag) {
            const int ln = e & 63, a = (e >> 6) & 3, blk = e >> 8;
            row = ln & 31, col = 32 * blk + 8 * a + 4 * (ln >> 5);
        } else {
            row = e >> 6, col = (e & 63) * 4;
        }
        *reinterpret_cast<f32x4*>(d + e * 4) = *reinterpret_cast<const f32x4*>(&tile[row][col]);
    }
}

// Wm [256][256], W1 [1024][256], W2 [256][1024] fp32 -> the 72 stage images of tail_x3_kernel: merge head h (stage h) is
// "W2 chunk h" of a 256-deep matrix, stages 8 .. 71 are the FFN image.
__global__ void pack_tail_kernel(const float* __restrict__ Wm, const float* __restrict__ W1, const float* __restrict__ W2,
                                 __bf16* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= TAIL_STAGES * 16 * 64) return;
    const int lane = t & 63, frag = (t >> 6) & 15, stage = t >> 10;
    const int m = lane & 31, half = lane >> 5;
    // T_MFMA16: fragment (block frag >> 1, feature half frag & 1): lane (m16, kg) carries output feature 16 fb + perm16(m16)
    // and contraction indices k_of16(kg, 0 .. 7) of the 32-deep block
    const int fb = frag & 1, f16 = 16 * fb + perm16(lane & 15), kg = lane >> 4;
    float v[8];
    if (stage < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = T_MFMA16 ? Wm[(int64_t)(32 * (frag >> 1) + f16) * 256 + 32 * stage + k_of16(kg, j)]
                            : Wm[(int64_t)(32 * (frag >> 1) + m) * 256 + 32 * stage + chunk_k(frag & 1, half, j)];
    } else {
        const int st = stage - 8;
        const bool up = st == 0 || (st < 63 && (st & 1));
        const int c = st == 0 ? 0 : st == 63 ? 31 : up ? (st + 1) / 2 : st / 2 - 1;
        if (up) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = T_MFMA16 ? W1[(int64_t)(32 * c + f16) * 256 + 32 * (frag >> 1) + k_of16(kg, j)]
                                : W1[(int64_t)(32 * c + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = T_MFMA16 ? W2[(int64_t)(32 * (frag >> 1) + f16) * 1024 + 32 * c + k_of16(kg, j)]
                                : W2[(int64_t)(32 * (frag >> 1) + m) * 1024 + 32 * c + chunk_k(frag & 1, half, j)];
        }
    }
    bf16x8 p0, p1, p2;
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    split3(lo, hi, p0, p1, p2);
    __bf16* dst = out + ((int64_t)stage * 3 * 16 + frag) * 64 * 8 + lane * 8;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + 16 * 64 * 8) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * 16 * 64 * 8) = p2;
}

// Sum of the per-128-row-tile K^T V partials of the fused q/k/v GEMM (as kv_finalize_tiles_kernel, attention.hip) written
// as the operand image of tail_x3_kernel: per cloud and head the A-operand fragments of KV_h^T / S (row m = value index
// v, contraction index d = chunk_k(step, half, j)) in three bf16 planes, then Ksum as fp32.  grid n_kv * 8, block 1024.
__global__ __launch_bounds__(1024) void kv_finalize_x3_kernel(const float* __restrict__ partial,
                                                            const int32_t* __restrict__ cloud_row0,
                                                            const int32_t* __restrict__ cloud_len, int64_t row_base,
                                                            int cloud_begin, char* __restrict__ kvimg) {
    constexpr int KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;
    const int kvi = blockIdx.x / SCREAM_NHEAD, h = blockIdx.x % SCREAM_NHEAD;
    const int cloud = cloud_begin + kvi;
    const int t0 = (int)((cloud_row0[cloud] - row_base) / SCREAM_ROW_TILE);
    const int nt = (cloud_len[cloud] + SCREAM_ROW_TILE - 1) / SCREAM_ROW_TILE;
    const float* p = partial + ((int64_t)t0 * SCREAM_NHEAD + h) * KV_ELEMS;
    char* img = kvimg + (size_t)cloud * KV_IMAGE_BYTES;
    const float S = (float)cloud_len[cloud];
    for (int i = threadIdx.x; i < KV_ELEMS; i += 1024) {
        float s8[16];  // sixteen chains in a fixed combination order: deterministic (same order as kv_finalize_tiles_kernel)
#pragma unroll
        for (int u = 0; u < 16; ++u) s8[u] = 0.f;
        for (int c = 0; c < nt; c += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c + u < nt) s8[u] += p[(int64_t)(c + u) * SCREAM_NHEAD * KV_ELEMS + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += s8[u + 8];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        if (i < 32 * 32) {
            const int d = i >> 5, v = i & 31;             // partial layout [d][v]
            const float x = s / S;                         // values / v_length (models/transformer.py:38-39), applied to the sum
            const int s2 = d >> 4, hf = (d >> 2) & 1, j = 4 * ((d >> 3) & 1) + (d & 3);  // d = chunk_k(s2, hf, j)
            const __bf16 a = (__bf16)x;
            const float r1 = x - (float)a;
            const __bf16 b = (__bf16)r1;
            const __bf16 cc = (__bf16)(r1 - (float)b);
            // T_MFMA16: fragment = value half v >> 4, lane = (kg = s2 + 2 hf, m16 = perm16(v & 15)) -- d = k_of16(kg, j)
            __bf16* base = reinterpret_cast<__bf16*>(img) +
                           (T_MFMA16 ? ((size_t)(h * 3) * 2 + (v >> 4)) * 512 + (16 * (s2 + 2 * hf) + perm16(v & 15)) * 8 + j
                                     : ((size_t)(h * 3) * 2 + s2) * 512 + (v + 32 * hf) * 8 + j);
            base[0] = a;
            base[2 * 512] = b;
            base[4 * 512] = cc;
        } else {
            reinterpret_cast<float*>(img + KV_PLANES_BYTES)[h * 32 + (i - 32 * 32)] = s;
        }
    }
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

extern "C" int64_t scream_tail_image_bytes(void) { return (int64_t)TAIL_STAGES * T_STAGE; }
extern "C" int64_t scream_kv_image_bytes(void) { return KV_IMAGE_BYTES; }

extern "C" int scream_pack_tail_x3(const float* Wm, const float* W1, const float* W2, void* image, void* stream) {
    SCREAM_REQUIRE(Wm && W1 && W2 && image, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, SCREAM_EINVAL);
    pack_tail_kernel<<<dim3(TAIL_STAGES * 16 * 64 / 256), dim3(256), 0, as_stream(stream)>>>(Wm, W1, W2, reinterpret_cast<__bf16*>(image));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_kv_finalize_x3(const float* kv_partial, const int32_t* cloud_row0, const int32_t* cloud_len,
                                     int64_t row_base, int32_t cloud_begin, int32_t n_kv, void* kv_image, void* stream) {
    SCREAM_REQUIRE(kv_partial && cloud_row0 && cloud_len && kv_image, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_kv >= 0 && cloud_begin >= 0 && row_base >= 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(kv_image) & 15) == 0, SCREAM_EINVAL);
    if (n_kv == 0) return 0;
    kv_finalize_x3_kernel<<<dim3(n_kv * SCREAM_NHEAD), dim3(1024), 0, as_stream(stream)>>>(kv_partial, cloud_row0, cloud_len, row_base,
                                                                                         cloud_begin, reinterpret_cast<char*>(kv_image));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_layer_tail_x3_f32(const float* Q, const void* kv_image, const int32_t* tile_cloud,
                                        int32_t kv_cloud_offset, const int32_t* cloud_len, const float* x,
                                        const void* tail_image, const float* g1, const float* b1, const float* g2,
                                        const float* b2, float* y, int64_t M, void* stream) {
    SCREAM_REQUIRE(Q && kv_image && tile_cloud && cloud_len && x && tail_image && g1 && b1 && g2 && b2 && y, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(kv_image) | reinterpret_cast<uintptr_t>(x) |
                     reinterpret_cast<uintptr_t>(tail_image) | reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(b1) |
                     reinterpret_cast<uintptr_t>(g2) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(x != y, SCREAM_EINVAL);  // the residual of a row is read twice, long after its neighbours were written
    const int64_t tiles = M / SCREAM_ROW_TILE;
    if (tiles == 0) return 0;
    SCREAM_REQUIRE(tiles < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = tiles < T_MAX_GRID ? (unsigned)tiles : (unsigned)T_MAX_GRID;
    tail_x3_kernel<<<dim3(grid), dim3(TT), 0, as_stream(stream)>>>(Q, reinterpret_cast<const char*>(kv_image), tile_cloud,
                                                                  kv_cloud_offset, cloud_len, x,
                                                                  reinterpret_cast<const __bf16*>(tail_image), g1, b1, g2, b2, y,
                                                                  (int)tiles);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_act_layout(const float* src, float* dst, int64_t M, int32_t to_fragment, void* stream) {
    SCREAM_REQUIRE(src && dst && src != dst, SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % 32 == 0 && M / 32 < (1ll << 31), SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, SCREAM_EINVAL);
    if (M == 0) return 0;
    act_layout_kernel<<<dim3((unsigned)(M / 32)), dim3(256), 0, as_stream(stream)>>>(src, dst, to_fragment ? 1 : 0);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
