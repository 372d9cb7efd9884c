// The row-local tail of an MHAttention block as ONE kernel on the 16-bit matrix cores of gfx950, fp32-accurate by an operand
// split (split.h: SplitH2, two fp16 planes and three products -- the default since round 3 -- or SplitBf3, three bf16 planes and
// six products):
//
//     att = ((Q' . KV) * Z) * S                 attention apply            (models/transformer.py:41-42)
//     m1  = LayerNorm1(att . Wm^T + x)          merge + norm1              (:83-84)
//     y   = LayerNorm2(x + W2 . relu(W1 . m1))  mlp + norm2                (:85-88; the residual is the block INPUT x)
//
// Why one kernel.  At the 1400 W socket cap a launch costs what its joules cost; the unfused chain moved
// 15 KB per row through HBM (att, m1 and the 1024-wide hidden activations twice), each time through an LDS-slab epilogue
// and an operand re-split.  Here neither att nor m1 nor the hidden activations exist in memory: per row and layer the
// kernel reads Q' and x (twice) and writes y -- 4 KB.
//
// How: everything is computed TRANSPOSED.  The weights are the MFMA A operand (M = output feature), a wave's 32
// activation rows are the B operand (N = row), so an accumulator holds C^T: lane = activation row, registers = output
// features.  That IS the B-operand layout of the next GEMM (lane = row, registers = contraction index), so the chain
//     att_h^T = KV_h^T . Q'_h^T -> (* Z * S, split) -> m^T += Wm[:, h] . att_h^T -> LN1 -> (split)
//             -> h^T_c = W1[c] . m1^T -> (relu, split) -> y^T += W2[:, c] . h^T_c -> LN2
// runs with no transposition, LDS round trip or HBM write in between: the contraction index of an MFMA is a dummy, so the
// fixed permutation between "register i of lane (r, half)" and "feature" is baked into the packed weight images
// (pack_tail_kernel below).  A LayerNorm is a sum over a lane's 128 registers plus ONE cross-lane add (lanes r and r + 32
// share a row) instead of a slab transpose and 2 x 32 DPP wave reductions per wave.
//
// Geometry: 256 threads = 4 waves, one per SIMD (128 accumulator registers, the operand planes of the wave's m1 rows -- 192
// registers as bf16 x 3, 128 as fp16 x 2 --, operand buffers), one persistent block per CU, 128 rows per block tile.  The
// weights stream through a ring of three LDS stages (16 KiB per operand plane) filled by LDS-DMA (global_load_lds_dwordx4);
// per 128-row tile the ring carries 72 stages: Wm head 0..7, then  W1_0 | W1_c, W2_{c-1} (c = 1 .. 31) | W2_31  -- the same
// sequence for every row tile, so the ring never drains at a tile boundary.  A stage image is stored exactly as the
// fragments are read: [plane][fragment][lane][16 B], i.e. every ds_read_b128 and every DMA piece is 1 KiB of contiguous
// memory, conflict-free without any swizzle.
//
// Row operands.  Lane (r, half) needs, of every 128-byte segment of row r, the 16-byte pieces 2a + half.  From a row-major
// matrix that is a row-per-lane access -- 32 to 64 distinct lines per wave instruction, ~400 cycles of the CU's texture unit
// each, a quarter of the kernel however the requests were spread (profiles/r02_tail_ablation_*.txt); staged through a
// wave-private LDS slab by LDS-DMA the lines were full but, with room for one 4 KiB slab per wave, every request had half a
// stage of lead and the HBM latency showed instead.  So the kernels agree on the layout in HBM: Q', x and y are
// FRAGMENT-major (SCREAM_ACT_FRAG, include/scream_hip.h) -- per 32-row group and 32-feature segment the pieces are stored
// [a][lane], i.e. each of a lane's four loads or stores per segment is one contiguous 1 KiB wave access that lands in
// exactly the registers the MFMA wants, with no LDS in between.  All row operands are inline-asm register loads (hipcc
// would otherwise wait vmcnt(0) at their first use and drain the weight ring), requested one stage ahead and BEFORE the
// stage's weight pieces so that the ring's counted wait covers them as well.
//
// SplitH2 scales (scream_tail_exps_t, chosen by scream_amd/scales.py so that no operand can leave fp16's range): the att, m1
// and hidden operands are multiplied by exact powers of two before their split (att: folded into Z; m1: folded into gamma1 /
// beta1; hidden: one multiply next to the relu), the weight images carry theirs.  The merge and FFN-down accumulators are
// then c = 2^(e_w + e_a) times the true sums; the residual x joins them as fma(x, c, acc) and the LayerNorm runs on the
// scaled values with eps c^2 -- bit for bit the LayerNorm of the unscaled values, so no power of two ever costs a rounding.
// The attention apply itself (12 matrix instructions per head, 3 % of a tile's) stays on the bf16 x 3 split in both
// instantiations: its KV operand is a data-dependent sum with no useful static bound.
//
// Tuning aids (tools/tail_ablate.py builds variants; always 0 in libscream_hip.so): T_ABLATE bit 0 no weight DMA after the
// first two stages, 1 no MFMAs, 2 no LDS fragment reads, 4 no row loads/stores, 5 no Q', 6 no x, 7 no y stores, 8 no KV
// operands, 9 no apply rides, 10 no residual adds, 11 every row request goes to the first tile (cache hits).
#include <type_traits>

#include "ring.h"

#ifndef T_MIX
// fp16 splits: operand planes by v_fma_mix (split.h: split2s), bit 0 in the FFN's relu / split ride (the default: 103 -> 52 vector
// instructions per down stage, bit-identical, 1.043 vs 1.046 ms per 333 k-row launch), bit 1 in norm1, the apply and the y planes
// as well (measured SLOWER, 1.057 ms: the asm statements keep hipcc from packing the norm arithmetic into v_pk_* -- profiles/r04_tail_mix_ab.txt)
#define T_MIX 1
#endif
#ifndef T_RIDE0
#define T_RIDE0 3  // SplitH2: first MFMA group of a down stage that carries a relu / split pair of the ride (eight groups from there); 0 / 3 / 6: 1.044 / 1.040 / 1.041 ms per 333 k-row launch (profiles/r04_tail_ride0_ab.txt)
#endif
#ifndef T_STASH
#define T_STASH 3  // x segments kept in LDS between their two reads (fp16 kernels; 0: every segment is read twice from memory)
#endif
#ifndef T_NT
#define T_NT 7  // non-temporal hint on: 1 the Q' loads, 2 the y (and next-layer Q') stores, 4 the second (last) read of the x rows, 8 their first read
#endif
#ifndef T_QF_DUMP
#define T_QF_DUMP 0
#endif
#ifndef T_QF_DBG
#define T_QF_DBG 0  // debugging aid for the QF kernel: 1 full drain + barrier at the first query stage, 2 the applies of heads 0 / 1 in the open, 4 no deferral across query stages
#endif
#ifndef T_DEFER_H2
#define T_DEFER_H2 2  // SplitH2: MFMA groups of a stage deferred across the barrier into the next stage (1 or 2)
#endif

// -DT_STAMPS (tools/tail_stamps.py): s_memtime stamps of the phases of the SECOND tile of every block, lane 0 of each wave.
// Diagnostic build only -- tools/tail_stamps.py runs tools/asm_inflight_check.py on it first: the extra registers can push
// hipcc into spilling a pending load destination (it did, with one stamp per stage).
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

constexpr int TAIL_STAGES = 72;
constexpr int NEXT_Q_STAGES = 8;  // tail_kernel<SP, true>: the NEXT layer's 256 -> 256 query projection rides behind norm2 (below)
constexpr int KV_PLANES_BYTES = 8 * 3 * 2 * 1024;            // per cloud: [head][plane][step][lane][8] bf16
constexpr int KV_IMAGE_BYTES = KV_PLANES_BYTES + 8 * 32 * 4;  // + Ksum [head][32] fp32

// fp16 splits (round 4, T_APPLY_H2): the apply runs on fp16 x 2 as well.  The image then holds, per head, KV_h^T / S * 2^e_h in TWO fp16
// planes ([head][plane][step][lane][8], 4 KiB per head) with e_h the largest exponent that keeps the head's largest |element| at or
// below 2^15 -- computed on the device by kv_finalize_image_kernel from the reduced sum itself (a maximum: exact and independent of
// any order, so batched == single pair stays bitwise) -- and 2^-e_h eight times over in the 32 bytes of head h at KV_H2_SCALE_OFF.
constexpr int KV_H2_HEAD_BYTES = 2 * 2 * 1024;
constexpr int KV_H2_SCALE_OFF = 8 * KV_H2_HEAD_BYTES;  // 32 KiB: inside the plane area the bf16 layout fills, unused by this one
#ifndef T_APPLY_H2
#define T_APPLY_H2 1  // 0: the attention apply of the fp16 kernels stays on bf16 x 3 (rounds 2-3)
#endif

struct HeadOps {   // the per-cloud operands of one head's apply, as loaded (Q' travels separately: f32x4 q[4], pieces a = 0 .. 3)
    f32x4 kv[6];   // KV_h^T fragments [plane][step], 16 bytes per lane (fp16 x 2: four of them)
    f32x4 ks[4];   // Ksum[h][8 a + 4 half .. + 4]
    f32x4 sc;      // fp16 x 2: 2^-e_h in every element
};

// SplitH2: the exact power-of-two factors of the kernel (all 1 / unused for SplitBf3)
struct TailScales {
    float s_att;  // 2^e_att, folded into Z
    float s_q;    // fp16 x 2 apply: 2^e_q, the scale of Q' = elu(q) + 1 <= 1 + the bound of q as an operand
    float s_attq; // 2^(e_att - e_q): with the head's 2^-e_h what takes the apply's accumulator to the scaled attention output
    float c1;     // 2^(e_wm + e_att): unit of the merge accumulators
    float eps1;   // 1e-5 c1^2
    float s_m1;   // 2^e_m1, folded into gamma1 / beta1
    float ch;     // 2^(e_h - e_w1 - e_m1): FFN-up accumulator -> scaled hidden activation
    float c2;     // 2^(e_w2 + e_h): unit of the FFN-down accumulators
    float eps2;   // 1e-5 c2^2
    float s_y;    // NQ: 2^e_y, the block OUTPUT as the operand of the next layer's query projection
    float cq;     // NQ: 2^-(e_y + e_wq): accumulator of that projection -> q
    float s_x;    // QF: 2^e_x, the block INPUT as the operand of this layer's query projection
    float cqf;    // QF: 2^-(e_x + e_wq)
};

// merge stage h (h >= 1): which quarter of the x-segment add rides in group g (-1: none) -- the last four groups of
// 15 - nd .. 8 (nd = deferred groups of the stage) that do not accumulate into tile h - 1
__device__ __forceinline__ constexpr int xadd_slot(int h, int g, int nd) {
    int n = 0;
    for (int c = 15 - nd; c >= 8; --c) {
        if ((c >> 1) == h - 1) continue;
        if (c == g) return n < 4 ? n : -1;
        ++n;
    }
    return -1;
}

// NQ (round 3, fp16 splits): the image carries eight more stages, Wq of the NEXT layer (a cross layer: its queries are this
// block's output rows, models/transformer.py:130), and the kernel ends every tile with  Q'_next = elu(y . Wq^T) + 1  -- y is in
// registers in operand layout the moment norm2 is done, so the separate projection launch (x re-read, re-split, one more launch
// per layer with its partial last round) disappears.  q_next may alias Q: a tile reads its own rows of Q long before it writes them.
// QF (round 4, fp16 splits): the image carries eight more stages IN FRONT -- Wq of THIS layer -- and every tile begins with
// Q' = elu(x . Wq^T) + 1 of its own rows, head by head, kept in registers for the applies: Q' is neither written by the projection
// (1 KB per row: 11 % of a projection launch by the ablation, profiles/r04_ring_proj_ride_ablation.txt) nor read back here, and the
// projection kernel shrinks to the key/value chunks.  x is in the tile's hands anyway (the residual of both norms).  Q is unused.
template <class SP, bool NQ, bool QF>
__global__ __launch_bounds__(TT, 1) void tail_kernel(const float* __restrict__ Q,      // fragment-major [M, 256] (QF: unused)
                                                     const char* __restrict__ kvimg,   // [n_clouds][KV_IMAGE_BYTES]
                                                     const int32_t* __restrict__ tile_cloud, int kv_cloud_offset,
                                                     const int32_t* __restrict__ cloud_len,
                                                     const float* __restrict__ xres,   // fragment-major [M, 256]
                                                     const char* __restrict__ Wimg,    // [72 stages][NP x 16 KiB]
                                                     const float* __restrict__ g1, const float* __restrict__ b1,
                                                     const float* __restrict__ g2, const float* __restrict__ b2,
                                                     float* __restrict__ y,            // fragment-major [M, 256]
                                                     float* q_next,                    // NQ: fragment-major [M, 256] (may alias Q)
                                                     int n_tiles, TailScales sc) {
    static_assert(!NQ || SP::SCALED, "the next-layer query projection is built for the fp16 splits");
    static_assert(!QF || (SP::SCALED && !NQ), "the in-kernel query projection is built for the fp16 splits and replaces the next-layer one");
    constexpr int N_STAGES = TAIL_STAGES + ((NQ || QF) ? NEXT_Q_STAGES : 0);
    typedef typename SP::vec V;
    constexpr int NP = SP::NP;
    constexpr int STAGE = stage_bytes<SP>();
    constexpr int PIECES = wave_pieces<SP>();  // LDS-DMA pieces per wave and stage = what a counted ring wait leaves in flight
    constexpr bool APPLY_H2 = SP::SCALED && T_APPLY_H2;  // the attention apply on fp16 x 2 planes (image written by kv_finalize_image in that form)
#ifdef T_NV_MERGE  // tuning aid (tools/tail_stamps.py, T_EXTRA): another ride-slot count
    constexpr int NV_MERGE = T_NV_MERGE;
#else
    constexpr int NV_MERGE = SP::NPROD >= 6 ? 6 : 8;  // ride slots behind every MFMA of a merge stage (32 cycles / 4 per VALU issue)
#endif
    // The last ND MFMA groups of every stage are DEFERRED across the barrier (below): 192 cycles of work on register operands
    // must cover the barrier skew and the first fragment reads of the next stage -- one group of six bf16 products, two groups
    // of three fp16 products.
    constexpr int ND = SP::NPROD >= 6 ? 1 : T_DEFER_H2;
    constexpr int NG = 16 - ND;  // groups issued inside their own stage
    // x segments 0 .. NSTASH - 1 of the tile stay in LDS between their first read (the norm1 residual, merge stages) and their second
    // (the norm2 residual, FFN): 16 KiB per segment and block, per-wave private (no synchronisation), in the LDS the two-plane rings
    // leave free -- 3/8 of the second read of x never reaches the L2 / HBM (round 4: the review's traffic item)
    constexpr int NSTASH = (NP == 2 && !QF) ? T_STASH : 0;
    // (ONE object on purpose: with a second __shared__ variable in the kernel hipcc can no longer tell the ring's LDS-DMA writes from other
    // LDS traffic and puts a vmcnt(0) in front of every fragment read -- 267 of them, every stage 3.4 x slower; met in round 4 with two ints)
    __shared__ __attribute__((aligned(16))) char smem[T_SLOTS * STAGE + 4096 + NSTASH * 16384];  // the ring + the norm parameters + the stash, the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // gamma1 | beta1 | gamma2 | beta2 live in LDS for the whole launch.  Read from global memory inside the norm blocks they
    // cost more than the arithmetic: on gfx950 loads and stores share vmcnt, hipcc cannot order a load against the y stores
    // issued before it and waits with vmcnt(0) -- a full store round trip per group of rows (tools/tail_stamps.py: 11.8 k
    // cycles for norm2 + stores).  From LDS the y stores are fire-and-forget.  (SplitH2: gamma1 / beta1 carry m1's 2^e.)
    float* lnp = reinterpret_cast<float*>(smem + T_SLOTS * STAGE);
    f32x4* stash = reinterpret_cast<f32x4*>(smem + T_SLOTS * STAGE + 4096 + wave * (NSTASH * 4096)) + lane;  // [segment][piece a][lane]
    lnp[tid] = SP::SCALED ? g1[tid] * sc.s_m1 : g1[tid];
    lnp[256 + tid] = SP::SCALED ? b1[tid] * sc.s_m1 : b1[tid];
    lnp[512 + tid] = g2[tid];
    lnp[768 + tid] = b2[tid];
    const float eps1 = SP::SCALED ? sc.eps1 : 1e-5f, eps2 = SP::SCALED ? sc.eps2 : 1e-5f;
    const unsigned v_lane16 = lane * 16, v_half16 = half * 16;  // the only per-lane address parts of the kernel
    // weight pieces: uniform (scalar) source address + the 32-bit lane offset.  A per-lane 64-bit pointer kept across the
    // kernel was spilled by hipcc and reloaded from scratch in EVERY stage -- behind a vmcnt(0) that drained the ring.
    auto dma_piece = [&](unsigned q, int u) {
        if ((T_ABLATE & 1) && q >= 2) return;
        const unsigned src = q % (unsigned)N_STAGES, slot = q % (unsigned)T_SLOTS;
        const char* sbase = Wimg + (size_t)src * STAGE + (wave * PIECES + (u & ~3)) * 1024;
        dma_1k(sbase + v_lane16, smem + slot * STAGE + (wave * PIECES + (u & ~3)) * 1024, u & 3);
    };
    unsigned q = 0;  // next stage to be consumed
#pragma unroll
    for (int u = 0; u < PIECES; ++u) dma_piece(0, u);
#pragma unroll
    for (int u = 0; u < PIECES; ++u) dma_piece(1, u);

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
        ld_asm4<1024, (T_NT & 1) != 0>(qb, seg_base(Q, grp, h), v_lane16);
    };
    auto req_head = [&](HeadOps& o, const char* kvc, int h) {  // KV^T fragments and Ksum of head h: L2-hot per-cloud data
        if (T_ABLATE & (16 | 256)) return;
        f32x4 (&kv4)[4] = reinterpret_cast<f32x4 (&)[4]>(o.kv[0]);
        if (APPLY_H2) {
            ld_asm4<1024>(kv4, kvc + h * KV_H2_HEAD_BYTES, v_lane16);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(o.sc) : "v"(v_half16), "s"(kvc + KV_H2_SCALE_OFF + 32 * h));
        } else {
            const char* kp = kvc + h * (3 * 2 * 1024);
            ld_asm4<1024>(kv4, kp, v_lane16);
            ld_asm2k(o.kv[4], o.kv[5], kp + 4 * 1024, v_lane16);
        }
        ld_asm4<32>(o.ks, kvc + KV_PLANES_BYTES + 128 * h, v_half16);
    };
    auto req_x = [&](f32x4 (&xs)[4], int64_t grp, int blk, auto second) {  // second: the FFN's read (norm2 residual), the last use of the rows
        if (T_ABLATE & (16 | 64)) return;
        if (T_ABLATE & 2048) grp = (int64_t)wave * 32 * SCREAM_D_MODEL;
        ld_asm4<1024, (T_NT & (decltype(second)::value ? 4 : 8)) != 0>(xs, seg_base(xres, grp, blk), v_lane16);
    };
    auto pin_head = [&](HeadOps& o) {
#pragma unroll
        for (int a = 0; a < 4; ++a) pin(o.ks[a]);
#pragma unroll
        for (int f = 0; f < (APPLY_H2 ? 4 : 6); ++f) pin(o.kv[f]);
        if (APPLY_H2) pin(o.sc);
    };
    // the residual x joins an accumulator tile: plain add, or (SplitH2) fma with the tile's unit c
    auto add_x4 = [&](f32x16& t, const f32x4 (&xs)[4], int a, float c) {  // one quarter (registers 4a .. 4a + 3)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xv = (T_ABLATE & 16) ? 1.0f : xs[a][k];
            t[4 * a + k] = SP::SCALED ? __builtin_fmaf(xv, c, t[4 * a + k]) : t[4 * a + k] + xv;
        }
    };
    auto pin_x = [&](f32x4 (&xs)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a) pin(xs[a]);
    };
    auto add_x = [&](f32x16& t, const f32x4 (&xs)[4], float c) {
#pragma unroll
        for (int a = 0; a < 4; ++a) add_x4(t, xs, a, c);
    };

    // KV^T / Ksum: ONE buffer, consumed in a stage's first groups and re-requested right after (a third of a stage of
    // lead is plenty for L2-hot data).  Q' comes from HBM and gets TWO buffers, alternating by head, requested at the
    // top of the stage BEFORE the one that consumes it -- with a single buffer the merge stages took 6.6 k cycles against
    // 3.8 k for an FFN stage (tools/tail_stamps.py).
    HeadOps op;
    f32x4 qA[4], qB[4];     // Q' of even / odd heads
    f32x4 xs[4], xs2[4];    // x segments (xs2: only segment 7 of the norm1 residual)
    V apA[2][NP], apB[2][NP];  // planes of att_h^T, the B operand of the merge GEMM: heads of even / odd index
    f32x16 aT;              // att_h^T tile of the head being applied
    bf16x8 qp[2][3];        // bf16 x 3 apply (SplitBf3 kernels; the fp16 kernels with T_APPLY_H2 = 0)
    f16x8 qph[2][2];        // fp16 x 2 apply: the planes of Q' 2^e_q
    float Zs = 0.f;

    // ---- apply of one head (models/transformer.py:41-42), in pieces that ride inside the groups of another stage ----
    auto apply_qsplit = [&](f32x4 (&qb)[4], int tile_tag, int s2) {  // 16-deep step s2 of Q' into its planes
        f32x4 lo = qb[2 * s2], hi = qb[2 * s2 + 1];
        if (T_ABLATE & 16) lo = hi = f32x4{(float)lane, 1.0f, 0.5f, (float)tile_tag};
        if (APPLY_H2) split8s<SplitH2>(lo, hi, sc.s_q, qph[s2]);
        else split8<SplitBf3>(lo, hi, qp[s2]);
    };
    auto apply_mfma = [&](int s2) {
        if (APPLY_H2) {
            f16x8 w[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) w[p] = __builtin_bit_cast(f16x8, op.kv[p * 2 + s2]);
            mfma_group<SplitH2, -1>(aT, w, qph[s2], s2 == 0);
        } else {
            bf16x8 w[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p] = __builtin_bit_cast(bf16x8, op.kv[p * 2 + s2]);
            mfma_group<SplitBf3, -1>(aT, w, qp[s2], s2 == 0);
        }
    };
    auto apply_z = [&](f32x4 (&qb)[4]) {  // Z = 1 / (Q'.Ksum + 1e-6); lanes r and r + 32 share row r
        float zp = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) zp += qb[a][k] * op.ks[a][k];
        zp += __shfl_xor(zp, 32);
        Zs = 1.0f / (zp + 1e-6f);
        // exact powers of two: (aT * (Z 2^e)) * S == ((aT * Z) * S) 2^e bit for bit; the fp16 x 2 apply's accumulator is in units of
        // 2^(e_h + e_q), so its Z also carries 2^-(e_h + e_q)
        if (APPLY_H2) Zs *= sc.s_attq * op.sc[0];
        else if (SP::SCALED) Zs *= sc.s_att;
    };
    auto apply_split_pair = [&](int k, V (&ap)[2][NP], float S) {  // elements 2k, 2k+1: (aT * Z) * S, then the operand split
        const int s2 = k >> 2, j = (2 * k) & 7;
        if constexpr (SP::SCALED && (T_MIX & 2)) {
            SP::split2s((aT[2 * k] * Zs) * S, (aT[2 * k + 1] * Zs) * S, 1.0f, j, ap[s2]);
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) SP::split1((aT[2 * k + e] * Zs) * S, j + e, ap[s2]);
        }
    };
    // the pieces of one apply as they ride in group g of a 16-group stage: operands consumed in groups 0-3
    auto apply_ride = [&](f32x4 (&qb)[4], int g, V (&ap)[2][NP], float S, int tile_tag) {
        if (g == 0) apply_qsplit(qb, tile_tag, 0);
        if (g == 1) apply_qsplit(qb, tile_tag, 1);
        if (g == 1) apply_mfma(0);
        if (g == 2) apply_mfma(1);
        if (g == 3) apply_z(qb);
        if (g >= 4 && g < 12) apply_split_pair(g - 4, ap, S);
    };

    // ---- QF: this layer's query projection in front of the merge stages --------------------------------------------------------
    V xp[QF ? 16 : 1][NP];        // operand planes of the wave's x rows (B operand of the eight query stages)
    f32x4 qall[QF ? 8 : 1][4];    // Q' of every head, fp32, in the piece layout the applies consume (what req_q loads otherwise)
    f32x16 hqf[2];                // query chunk accumulators, alternating
    auto elu_q = [&](int k, const f32x16& h, f32x4 (&dst)[4]) {  // elements 2k, 2k + 1 of a finished chunk
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * k + e;
            dst[i >> 2][i & 3] = elu1(h[i] * sc.cqf);
        }
    };

    float dbg6[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // T_QF_DUMP == 6: checksums of the first two applies, stored at the tile's end
    auto xsum = [&](const V (&ap)[2][NP]) {  // xor of every word of a head's operand planes
        unsigned c = 0;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const f32x4 w = __builtin_bit_cast(f32x4, ap[s2][pl]);
#pragma unroll
                for (int k = 0; k < 4; ++k) c ^= __builtin_bit_cast(unsigned, w[k]);
            }
        return __builtin_bit_cast(float, (c & 0x007fffffu) | 0x3f800000u);  // (a float in [1, 2): comparable bit for bit, never a NaN)
    };
    auto asum = [&](const f32x16 (&a)[8]) {
        unsigned c = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) c ^= __builtin_bit_cast(unsigned, a[b][e]);
        return __builtin_bit_cast(float, (c & 0x007fffffu) | 0x3f800000u);
    };
    int tile = blockIdx.x;
    int64_t grp = ((int64_t)tile * 128 + wave * 32) * SCREAM_D_MODEL;  // first float of this wave's 32-row group
    const char* kvc = nullptr;
    float S = 1.f;
    if (tile < n_tiles) {
        const int cl = tile_cloud[tile] + kv_cloud_offset;
        kvc = kvimg + (size_t)cl * KV_IMAGE_BYTES;
        S = (float)cloud_len[cl];
    }
    if (!QF && tile < n_tiles) {  // the block's first tile: heads 0 and 1 are applied in the open (once per block)
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
        unsigned marks[16] = {};
#endif
        TSTAMP(0);  // tile start
        // The last MFMA group(s) of every stage are DEFERRED across the barrier: their weight fragments are read into wfd, and
        // the next stage issues them right after the first fragment reads of its own -- work with register
        // operands exactly where a lone in-order wave otherwise waits for the LDS (tools/tail_stamps.py: 3.8 k cycles per
        // 3.07 k-cycle stage of the bf16 kernel).  `flush` arguments below name the deferred groups of the preceding stage.
        V wfd[ND][NP];
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
        //   groups 4-15: the stage's weight pieces, AFTER every row request, so that the next barrier's counted wait covers them.
        auto stage_merge = [&](auto hh, V (&ap)[2][NP], V (&ap_next)[2][NP], f32x4 (&q_cons)[4], f32x4 (&q_req)[4], auto flush) {
            constexpr int h = decltype(hh)::value;
            constexpr bool RIDE = h >= 1 && h <= 6;
            // stage 0 of a tile: everything older was drained at the end of the previous tile (only its y stores may
            // still be in flight, and nothing of this stage depends on them)
            // (NQ: the previous tile ended in ring stages whose last chunk's STORES are the youngest operations in the queue -- a counted
            // wait is only sound among loads, so this one barrier per tile drains)
            // (QF: the query stages sit in front, so merge stage 0 is an ordinary ring stage)
            if (h == 0 && !QF) { if (NQ) ring_barrier<0>(); else lds_only_barrier(); }
            else if (QF && (T_QF_DBG & 32) && h <= 1) ring_barrier<0>();
            else ring_barrier<PIECES>();
            TMARK(h);  // T_STAMPS builds: marks 0-7 = the tops of the merge stages
            __builtin_amdgcn_sched_barrier(0);
            f32x4 (&x_prev)[4] = (h & 1) ? xs : xs2;   // segment h - 1 (landed: requested by stage h - 1)
            f32x4 (&x_req)[4] = (h & 1) ? xs2 : xs;    // segment h
            if (h > 0) {
                pin_x(x_prev);
                if (h - 1 < NSTASH) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) stash[((h - 1) * 4 + a) * 64] = x_prev[a];
                }
                if (RIDE) {
                    pin_head(op);
                    pin_x(q_cons);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            req_x(x_req, grp, h, std::false_type{});
            if (!QF && h + 2 < 8) req_q(q_req, grp, h + 2);  // consumed by stage h + 1 (QF: Q' of every head is in registers already)
            if (h == 0) req_head(op, kvc, 2);
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
            V wf[T_PF][NP];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < NG; ++g) {  // g = blk * 2 + s2; the last ND groups are deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        (g + T_PF - 1 >= NG ? wfd[g + T_PF - 1 - NG][p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag<V>(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (g == 4 && RIDE && h + 2 < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    req_head(op, kvc, h + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g >= 4 && g - 4 < PIECES) dma_piece(q + 2, g - 4);
                if (PIECES == 12 && g == 14) dma_piece(q + 2, 11);  // (group 15 is deferred: the twelfth piece goes out with the eleventh)
                if (RIDE && !(T_ABLATE & 512)) apply_ride(q_cons, g, ap_next, S, tile);
                if (QF && h == 0 && g < 8) elu_q(g, hqf[1], qall[QF ? 7 : 0]);  // the last query chunk (its deferred groups were flushed above)
                // the norm1 residual: segment h - 1 joins accumulator tile h - 1 in quarters, in late groups that do not
                // accumulate into that tile (its MFMAs are groups 2h - 2 and 2h - 1)
                if (h > 0) {
                    const int pc = xadd_slot(h, g, ND);
                    if (pc >= 0 && !(T_ABLATE & 1024)) add_x4(acc[h > 0 ? h - 1 : 0], x_prev, pc, sc.c1);
                }
                mfma_group<SP, (h > 0 ? NV_MERGE : 0)>(acc[g >> 1], wf[g % T_PF], ap[g & 1], h == 0 && (g & 1) == 0);
            }
            ++q;
        };
        constexpr std::integral_constant<bool, true> yes{};
        constexpr std::integral_constant<bool, false> no{};
#define HEAD(n) std::integral_constant<int, n>{}
        auto flush_none = [&]() {};
        // the deferred groups 16 - ND .. 15 of a merge or down stage all accumulate into tile 7; first: the tile starts here
        auto flush_acc7 = [&](V (&b)[2][NP], bool first) {
#pragma unroll
            for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(acc[7], wfd[i], b[(NG + i) & 1], first && ((NG + i) & 1) == 0);
        };
        auto flush_mergeA = [&]() { flush_acc7(apA, false); };  // deferred groups of a merge stage of an even head
        auto flush_mergeB = [&]() { flush_acc7(apB, false); };
        auto flush_mergeA0 = [&]() { flush_acc7(apA, true); };  // ... of stage 0, which starts the accumulator tiles
        if constexpr (QF) {
            // ---- Q' = elu(x . Wq^T) + 1 of this tile, head j in ring stage j (an FFN-up stage with x's planes as the operand); the elu of
            // chunk j - 1 rides in stage j, the applies of heads 0 and 1 ride in stages 6 and 7 (their KV operands are requested one stage
            // ahead, before the stage's weight pieces, as in the merge stages), everything else of the merge phase is unchanged.
            {
                f32x4 raw[8][4];  // the wave's 32 rows: piece a of segment blk = one contiguous 1 KiB per wave instruction
                const float* xg = xres + grp + lane * 4;
                if (T_QF_DBG & 256) VM_WAIT(0);  // (the previous tile's 32 y stores: never more than 32 operations in the queue)
#pragma unroll
                for (int blk = 0; blk < 8; ++blk)
#pragma unroll
                    for (int a = 0; a < 4; ++a) raw[blk][a] = ld4(xg + (blk * 4 + a) * 256);
                VM_WAIT(0);  // (also: this wave's pieces of the tile's first two ring stages have landed -- stage 0 starts with an LDS-only barrier)
#pragma unroll
                for (int blk = 0; blk < 8; ++blk)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) split8s<SP>(raw[blk][2 * s2], raw[blk][2 * s2 + 1], sc.s_x, xp[QF ? 2 * blk + s2 : 0]);
            }
            auto stage_qf = [&](auto jj) {
                constexpr int j = decltype(jj)::value;
                constexpr int RIDE = j >= 6 ? j - 6 : -1;  // the head whose apply rides here
                constexpr bool RIDE6 = !(T_QF_DBG & (2 | 64)), RIDE7 = !(T_QF_DBG & (2 | 128));
                constexpr bool RIDE_ON = j == 6 ? RIDE6 : RIDE7;
                if (j == 0) { if (T_QF_DBG & 1) ring_barrier<0>(); else lds_only_barrier(); }
                else if ((T_QF_DBG & 8) && j >= 6) ring_barrier<0>();
                else if ((T_QF_DBG & 16) && j >= 1 && j <= 5) ring_barrier<0>();
                else ring_barrier<PIECES>();
                __builtin_amdgcn_sched_barrier(0);
                if (RIDE >= 0 && RIDE_ON) pin_head(op);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 5 && RIDE6) req_head(op, kvc, 0);
                if (j == 6 && !RIDE6 && RIDE7) req_head(op, kvc, 1);
                __builtin_amdgcn_sched_barrier(0);
                const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
                V wf[T_PF][NP];
#pragma unroll
                for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                    for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
                if (j > 0) {
#pragma unroll
                    for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(hqf[(j + 1) & 1], wfd[i], xp[QF ? NG + i : 0]);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + T_PF - 1 < 16) {
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            (g + T_PF - 1 >= NG ? wfd[g + T_PF - 1 - NG][p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag<V>(wb + (p * 16 + g + T_PF - 1) * 1024);
                    }
                    if (g == 4 && RIDE == 0 && RIDE6 && RIDE7) {
                        __builtin_amdgcn_sched_barrier(0);
                        req_head(op, kvc, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (g >= 4 && g - 4 < PIECES) dma_piece(q + 2, g - 4);
                    if (j > 0 && g < 8) elu_q(g, hqf[(j + 1) & 1], qall[QF && j > 0 ? j - 1 : 0]);
                    if (RIDE >= 0 && RIDE_ON && !(T_ABLATE & 512)) apply_ride(qall[QF && RIDE >= 0 ? RIDE : 0], g, RIDE == 0 ? apA : apB, S, tile);
                    mfma_group<SP, 8>(hqf[j & 1], wf[g % T_PF], xp[QF ? g : 0], g == 0);
                }
                ++q;
            };
            stage_qf(HEAD(0)); stage_qf(HEAD(1)); stage_qf(HEAD(2)); stage_qf(HEAD(3));
            stage_qf(HEAD(4)); stage_qf(HEAD(5)); stage_qf(HEAD(6));
            stage_qf(HEAD(7));
            if (T_QF_DUMP == 6) {
                unsigned c = 0;
#pragma unroll
                for (int h = 0; h < 7; ++h)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int k = 0; k < 4; ++k) c ^= __builtin_bit_cast(unsigned, qall[QF ? h : 0][a][k]);
                dbg6[4] = __builtin_bit_cast(float, (c & 0x007fffffu) | 0x3f800000u);  // Q' of heads 0 .. 6
            }
            if (T_QF_DBG & (2 | 64 | 128)) {  // debugging: heads 0 and / or 1 applied in the open behind the query stages (as the first tile's prologue does)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!(T_QF_DBG & 2) && !(T_QF_DBG & (h == 0 ? 64 : 128))) continue;
                    req_head(op, kvc, h);
                    VM_WAIT(0);
                    pin_head(op);
#pragma unroll
                    for (int g = 0; g < 12; ++g) apply_ride(qall[QF ? h : 0], g, h == 0 ? apA : apB, S, tile);
                }
            }
        }
        auto flush_q7 = [&]() {  // QF: the deferred groups of the last query stage
#pragma unroll
            for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(hqf[1], wfd[i], xp[QF ? NG + i : 0]);
        };
        // QF: Q' of head h + 1 comes from qall, nothing is requested
#define QC(h1, q_) qall[QF ? (h1) : 0]
        if constexpr (QF) {
            stage_merge(HEAD(0), apA, apB, QC(1, qB), QC(1, qA), flush_q7);
            stage_merge(HEAD(1), apB, apA, QC(2, qA), QC(2, qB), flush_mergeA0);
            stage_merge(HEAD(2), apA, apB, QC(3, qB), QC(3, qA), flush_mergeB);
            stage_merge(HEAD(3), apB, apA, QC(4, qA), QC(4, qB), flush_mergeA);
            stage_merge(HEAD(4), apA, apB, QC(5, qB), QC(5, qA), flush_mergeB);
            stage_merge(HEAD(5), apB, apA, QC(6, qA), QC(6, qB), flush_mergeA);
            stage_merge(HEAD(6), apA, apB, QC(7, qB), QC(7, qA), flush_mergeB);
            stage_merge(HEAD(7), apB, apA, QC(7, qA), QC(7, qB), flush_mergeA);
        } else {
        //            head   planes  planes of head + 1   Q' consumed (head + 1)   Q' requested (head + 2)   deferred groups of
        stage_merge(HEAD(0), apA, apB, qB, qA, flush_none);   // (the previous tile ended flushed)
        stage_merge(HEAD(1), apB, apA, qA, qB, flush_mergeA0);
        stage_merge(HEAD(2), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(3), apB, apA, qA, qB, flush_mergeA);
        stage_merge(HEAD(4), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(5), apB, apA, qA, qB, flush_mergeA);
        stage_merge(HEAD(6), apA, apB, qB, qA, flush_mergeB);
        stage_merge(HEAD(7), apB, apA, qA, qB, flush_mergeA);
        }
#undef QC
        flush_mergeB();  // norm1 needs the finished accumulators
        vm_wait<PIECES>();  // x segment 7 (requested at the top of stage 7, older than that stage's weight pieces)
        pin_x(xs2);
        add_x(acc[7], xs2, sc.c1);

        TSTAMP(1);  // end of the merge phase
#if T_QF_DUMP  // debugging: y := Q' of the tile (1) / the merge accumulators in front of norm1 (3); the first up stage below drains
        if (QF && T_QF_DUMP != 6 && (T_QF_DUMP < 4 || q_next)) {  // (4 / 5: the same two quantities into q_next, y stored as usual)
            float* yd = (T_QF_DUMP < 4 ? y : q_next) + grp + lane * 4;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (T_QF_DUMP & 1) ? qall[QF ? b : 0][a][k] : acc[b][4 * a + k];
                    *reinterpret_cast<f32x4*>(yd + (b * 4 + a) * 256) = o;
                }
        }
#endif
        // ---- m1 = LayerNorm1(merge + x) (models/transformer.py:84), straight into the B-operand planes of FFN-up ---
        V mp[16][NP];
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
            const float rstd = 1.0f / sqrtf(var * (1.0f / 256.0f) + eps1);
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
                    if (T_MIX & 2) split8s<SP>(v[0], v[1], 1.0f, mp[2 * b + s2]); else split8<SP>(v[0], v[1], mp[2 * b + s2]);
                }
        }

        TSTAMP(2);  // end of norm1
        if (QF && T_QF_DUMP == 6) {
            unsigned c = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const f32x4 w = __builtin_bit_cast(f32x4, mp[i][pl]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) c ^= __builtin_bit_cast(unsigned, w[k]);
                }
            (void)c;
        }
        // ---- FFN; x segments 0 .. 7 (the norm2 residual) are added under the first eight down stages
        f32x16 hT;
        V hpA[2][NP], hpB[2][NP];
        // relu + split of elements 2k, 2k+1 of the finished h^T tile into the B-operand planes of the down GEMM
        auto split_pair = [&](int k, V (&hout)[2][NP]) {
            const int s2 = k >> 2, j = (2 * k) & 7;
            // relu as ONE v_max_f32: fmaxf() (and every builtin that folds to it) costs a second instruction that quiets a NaN the
            // accumulator cannot hold
            float x0, x1;
            if (T_MIX & 1) {
                asm("v_max_f32 %0, 0, %1" : "=v"(x0) : "v"(hT[2 * k]));
                asm("v_max_f32 %0, 0, %1" : "=v"(x1) : "v"(hT[2 * k + 1]));
            } else {
                x0 = fmaxf(hT[2 * k], 0.f);
                x1 = fmaxf(hT[2 * k + 1], 0.f);
            }
            if constexpr (SP::SCALED && (T_MIX & 1)) {
                SP::split2s(x0, x1, sc.ch, j, hout[s2]);  // the planes of x 2^e, one v_fma_mix each (split.h)
            } else if constexpr (SP::SCALED) {
                SP::split1(x0 * sc.ch, j, hout[s2]);
                SP::split1(x1 * sc.ch, j + 1, hout[s2]);
            } else {
                SP::split1(x0, j, hout[s2]);
                SP::split1(x1, j + 1, hout[s2]);
            }
        };
        // XLOAD: x segment to request in this stage (-1: none)
        auto stage_up = [&](auto first, auto xload, auto flush) {
            constexpr int XLOAD = decltype(xload)::value;
            // FIRST: the norm1 block above used ordinary loads (gamma, beta), which hipcc waits for with vmcnt(0)
            if (decltype(first)::value) ring_barrier<0>(); else ring_barrier<PIECES>();
            TMARK2(8, 10);  // T_STAMPS builds: tops of the last two up stages
            __builtin_amdgcn_sched_barrier(0);
            if (XLOAD >= 0 && XLOAD < NSTASH) {
#pragma unroll
                for (int a = 0; a < 4; ++a) xs[a] = stash[(XLOAD * 4 + a) * 64];
            } else if (XLOAD >= 0) req_x(xs, grp, XLOAD, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
            V wf[T_PF][NP];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < NG; ++g) {  // the last ND groups (hT += wfd . mp[g]) are deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        (g + T_PF - 1 >= NG ? wfd[g + T_PF - 1 - NG][p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag<V>(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (g < PIECES) dma_piece(q + 2, g);
                mfma_group<SP>(hT, wf[g % T_PF], mp[g], g == 0);
            }
            ++q;
        };
        // XADD: x segment (requested by the previous stage) to add into its accumulator tile (-1: none).
        // RIDE (the tile's last two stages, when the FFN's operand planes are dead and registers are available again):
        // 1 = requests the operands of the next tile's head 0; 2 = the apply of that head rides here, and head 1 is
        // requested behind it (applied in the open right after the stage).
        auto stage_down = [&](V (&hin)[2][NP], V (&hout)[2][NP], auto with_split, auto xadd, auto ride, auto flush) {
            constexpr int XADD = decltype(xadd)::value;
            constexpr int RIDE = decltype(ride)::value;
#ifdef T_PROBE_NOBAR  // TIMING PROBE ONLY (wrong results): the down stages wait for their pieces but skip the workgroup barrier
            __builtin_amdgcn_s_waitcnt(0x0070 | (PIECES & 15) | ((PIECES >> 4) << 14));
#else
            ring_barrier<PIECES>();
#endif
            TMARK2(9, 11);  // ... and of the last two down stages
            __builtin_amdgcn_sched_barrier(0);
            if (XADD == 0) {  // the tile's first down stage starts the accumulators: tile 0 from its x segment, the others from 0
                pin_x(xs);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xv = (T_ABLATE & 16) ? 1.0f : xs[a][k];
                        acc[0][4 * a + k] = SP::SCALED ? xv * sc.c2 : xv;
                    }
            } else if (XADD > 0) {
                pin_x(xs);
                add_x(acc[XADD > 0 ? XADD : 0], xs, sc.c2);
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
            const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
            V wf[T_PF][NP];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < NG; ++g) {  // g = blk * 2 + s2; the last ND groups (acc[7] += wfd . hin[g & 1]) are deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        (g + T_PF - 1 >= NG ? wfd[g + T_PF - 1 - NG][p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag<V>(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (RIDE == 2 && g == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    req_q(qA, grp_next, 1);  // head 1 into the buffers head 0 has just left (a second Q' buffer here, at the
                    req_head(op, kvc_next, 1);  // top of the stage, was spilled by hipcc right behind its loads)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (RIDE == 2) {
                    if (g >= 4 && g - 4 < PIECES) dma_piece(q + 2, g - 4);
                    if (PIECES == 12 && g == 14) dma_piece(q + 2, 11);  // (group 15 is deferred: the twelfth piece goes out with the eleventh)
                } else {
                    if (g < PIECES) dma_piece(q + 2, g);
                }
                if (RIDE == 2) apply_ride(qA, g, apA, S_next, tile_next);
                // the relu / split of the h^T tile the previous stage finished: eight pairs, in every other group (in the first
                // eight groups when group 14 is deferred)
                // (T_RIDE0: the first group that carries a pair -- the h^T tile was completed by the deferred groups flushed at this stage's
                // top, and a vector instruction that reads it stalls the whole in-order wave, MFMAs included, until those have drained)
                if (decltype(with_split)::value && (ND == 1 ? (g & 1) == 0 : (g >= T_RIDE0 && g < T_RIDE0 + 8))) split_pair(ND == 1 ? g >> 1 : g - T_RIDE0, hout);
                mfma_group<SP>(acc[g >> 1], wf[g % T_PF], hin[g & 1], XADD == 0 && g >= 2 && (g & 1) == 0);
            }
            ++q;
        };
        constexpr std::integral_constant<int, -1> none{};
        constexpr std::integral_constant<int, 0> ride0{};
        // stage order (= image order): W1_0 | W1_c, W2_{c-1} for c = 1 .. 31 | W2_31.  The norm2 residual: the up stage of
        // chunk c requests x segment c - 1, the down stage of chunk c - 1 that follows adds it (c - 1 < 8) -- the first
        // four pair iterations are peeled so that every accumulator index is a compile-time constant.
        auto flush_up = [&]() {                                   // deferred groups of an up stage
#pragma unroll
            for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(hT, wfd[i], mp[NG + i]);
        };
        auto flush_downA = [&]() { flush_acc7(hpA, false); };     // ... of a down stage whose operand was hpA
        auto flush_downB = [&]() { flush_acc7(hpB, false); };
        auto flush_downA0 = [&]() { flush_acc7(hpA, true); };     // ... of the tile's first down stage, which starts the accumulator tiles
        stage_up(yes, none, flush_none);
        flush_up();
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
        if constexpr (QF) {  // (the next tile's Q' does not exist yet: its first applies ride in its own query stages)
            stage_down(hpA, hpB, yes, none, ride0, flush_up);
            stage_down(hpB, hpA, no, none, ride0, flush_downA);
        } else {
        stage_down(hpA, hpB, yes, none, HEAD(1), flush_up);         // chunk 30, splitting chunk 31; requests head 0 of the block's next tile
        stage_down(hpB, hpA, no, none, HEAD(2), flush_downA);       // chunk 31; applies that head, requests head 1
        }
        flush_downB();                                              // norm2 needs the finished accumulators
#undef HEAD
        // Drain: the ring's two stages in flight and head 1 of the next tile, requested two thirds of a stage ago.  That
        // head is applied here in the open (12 MFMAs), so that nothing pending lives across the norm2 block, and the y
        // stores below are YOUNGER than every load a later counted wait is meant to cover (stores retire out of order with
        // respect to loads, gemm_split.hip; stage 0 of the next tile starts without a vector-memory wait).
        TSTAMP(3);  // end of the last stage
        VM_WAIT(0);
        if (QF && T_QF_DUMP == 6) dbg6[3] = asum(acc);  // x + ffn, in front of norm2
        if (!QF && has_next) {
            pin_head(op);
            pin_x(qA);
#pragma unroll
            for (int g = 0; g < 12; ++g) apply_ride(qA, g, apB, S_next, tile_next);
        }

        TSTAMP(4);  // after the open apply of the next tile's head 1
        // ---- y = LayerNorm2(x + ffn) (the residual is already in the accumulators), stored fragment-major ------------
        V yp[NQ ? 16 : 1][NP];  // NQ: the planes of y
        unsigned osum_keep = 0, gsum_keep = 0;
        {
            float sum = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[b][e];
            if (QF && T_QF_DUMP == 6) {  // the lane's own partial sum, the partner's as exchanged, and a second exchange of the same value
                const float p1 = __shfl_xor(sum, 32), p2 = __shfl_xor(sum, 32);
                dbg6[5] = sum; dbg6[6] = p1; dbg6[7] = (p1 == p2) ? 1.0f : 2.0f;
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
            const float rstd = 1.0f / sqrtf(var * (1.0f / 256.0f) + eps2);
            const float* gp = lnp + 512 + 4 * half;
            const float* bp = lnp + 768 + 4 * half;
            float* yg = y + grp + lane * 4;  // (uniform base + lane: the stores below differ by immediates and scalar adds)
            unsigned osum = 0, gsum = 0;
            if (QF && T_QF_DUMP == 6) { dbg6[0] = mean; dbg6[1] = rstd; dbg6[2] = var; }
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f32x4 o2[2];
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const int a = 2 * s2 + a2;
                        const f32x4 g4 = ld4(gp + 32 * b + 8 * a), b4 = ld4(bp + 32 * b + 8 * a);
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = acc[b][4 * a + k] * rstd * g4[k] + b4[k];
                        if (QF && T_QF_DUMP == 6) {  // checksum of the values as computed (slot 4) and of the norm parameters as read (slot 3)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                osum ^= __builtin_bit_cast(unsigned, o[k]);
                                gsum ^= __builtin_bit_cast(unsigned, g4[k]) ^ (__builtin_bit_cast(unsigned, b4[k]) * 3u);
                            }
                        }
                        // one contiguous 1 KiB per wave instruction
                        if ((!(T_ABLATE & (16 | 128)) && !(QF && T_QF_DUMP && T_QF_DUMP < 4)) || o[0] + o[1] + o[2] + o[3] == 123.456f) {
                            if (T_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yg + (b * 4 + a) * 256));
                            else *reinterpret_cast<f32x4*>(yg + (b * 4 + a) * 256) = o;
                        }
                        o2[a2] = o;
                    }
                    if (NQ) {  // B operand of the query stages below
                        if (T_MIX & 2) split8s<SP>(o2[0], o2[1], sc.s_y, yp[NQ ? 2 * b + s2 : 0]);
                        else split8<SP>(o2[0] * sc.s_y, o2[1] * sc.s_y, yp[NQ ? 2 * b + s2 : 0]);
                    }
                }
            osum_keep = osum;
            gsum_keep = gsum;
        }
        if constexpr (NQ) {
            // ---- Q'_next = elu(y . Wq^T) + 1: eight ring stages, chunk j = output features 32 j .. 32 j + 31 (an FFN-up stage with y's
            // planes as the operand), accumulated alternately in hq[0 / 1].  Chunk j - 1 -- complete once stage j has flushed its deferred
            // groups -- gets its elu in stage j's first eight groups and is stored behind them, BEFORE the stage issues its weight
            // pieces (second half of the groups): the next barrier's counted wait then leaves exactly those pieces in flight, and
            // every store is older than every load a counted wait has to cover (stores retire out of order against loads).
            f32x16 hq[2];
            f32x4 oq[4];
            float* qg = q_next + grp + lane * 4;
            auto elu_pair = [&](int k, const f32x16& h) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * k + e;
                    const float x = h[i] * sc.cq;
                    oq[i >> 2][i & 3] = elu1(x);
                }
            };
            auto store_chunk = [&](int j) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (!(T_ABLATE & (16 | 128)) || oq[a][0] == 123.456f) {
                        if (T_NT & 2) __builtin_nontemporal_store(oq[a], reinterpret_cast<f32x4*>(qg + (j * 4 + a) * 256));
                        else *reinterpret_cast<f32x4*>(qg + (j * 4 + a) * 256) = oq[a];
                    }
            };
            constexpr int PP = (PIECES + NG - 9) / (NG - 8);  // weight pieces per group from group 8 on
            auto stage_q = [&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (j == 0) lds_only_barrier(); else ring_barrier<PIECES>();  // (stage 0: the queue was drained in front of norm2)
                __builtin_amdgcn_sched_barrier(0);
                const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
                V wf[T_PF][NP];
#pragma unroll
                for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                    for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
                if (j > 0) {
#pragma unroll
                    for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(hq[(j + 1) & 1], wfd[i], yp[NG + i]);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + T_PF - 1 < 16) {
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            (g + T_PF - 1 >= NG ? wfd[g + T_PF - 1 - NG][p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag<V>(wb + (p * 16 + g + T_PF - 1) * 1024);
                    }
                    if (j > 0 && g < 8) elu_pair(g, hq[(j + 1) & 1]);
                    if (g == 8) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (j > 0) store_chunk(j - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (g >= 8) {
#pragma unroll
                        for (int u = (g - 8) * PP; u < (g - 7) * PP; ++u)
                            if (u < PIECES) dma_piece(q + 2, u);
                    }
                    mfma_group<SP, 8>(hq[j & 1], wf[g % T_PF], yp[g], g == 0);
                }
                ++q;
            };
            stage_q(std::integral_constant<int, 0>{});
            stage_q(std::integral_constant<int, 1>{});
            stage_q(std::integral_constant<int, 2>{});
            stage_q(std::integral_constant<int, 3>{});
            stage_q(std::integral_constant<int, 4>{});
            stage_q(std::integral_constant<int, 5>{});
            stage_q(std::integral_constant<int, 6>{});
            stage_q(std::integral_constant<int, 7>{});
#pragma unroll
            for (int i = 0; i < ND; ++i) mfma_group<SP, -1>(hq[1], wfd[i], yp[NG + i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) elu_pair(k, hq[1]);
            store_chunk(7);
        }
        if (QF && T_QF_DUMP == 6 && q_next) {
            float* qd = q_next + grp + lane * 8;
            dbg6[4] = __builtin_bit_cast(float, (osum_keep & 0x007fffffu) | 0x3f800000u);
            *reinterpret_cast<f32x4*>(qd) = f32x4{dbg6[0], dbg6[1], dbg6[2], dbg6[3]};
            *reinterpret_cast<f32x4*>(qd + 4) = f32x4{dbg6[4], dbg6[5], dbg6[6], dbg6[7]};
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

// Wm [256][256], W1 [1024][256], W2 [256][1024] fp32 -> the 72 stage images of tail_kernel<SP>,
// [72][SP::NP planes][16 fragments][64 lanes][8] 16-bit values, in the order the kernel consumes them: merge head h (stage h)
// is "W2 chunk h" of a 256-deep matrix; stages 8 .. 71 are  W1_0 | W1_c, W2_{c-1} (c = 1 .. 31) | W2_31.  A-operand row
// m = lane & 31 of fragment frag = 2 blk + s2 is output feature 32 blk + m (down / merge) or hidden unit 32 c + m (up); its
// eight values are contraction indices chunk_k(s2, half, 0 .. 7) of the stage's 32-wide chunk.  SplitH2: every matrix is
// multiplied by its exact 2^e first.  One thread per (stage, fragment, lane).
// Wq_next != NULL: eight more stages, chunk j of the next layer's query projection [256][256] laid out like an up stage.
template <class SP>
__global__ void pack_tail_kernel(const float* __restrict__ Wm, const float* __restrict__ W1, const float* __restrict__ W2,
                                 const float* __restrict__ Wq_next, int q_first, float s_wm, float s_w1, float s_w2, float s_wq,
                                 typename SP::vec* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (TAIL_STAGES + (Wq_next ? NEXT_Q_STAGES : 0)) * 16 * 64) return;
    const int lane = t & 63, frag = (t >> 6) & 15, stage_out = t >> 10;
    // q_first: the eight query stages (THIS layer's Wq) come first in the image, the 72 stages of the tail behind them
    const int stage = !(Wq_next && q_first) ? stage_out : stage_out < NEXT_Q_STAGES ? TAIL_STAGES + stage_out : stage_out - NEXT_Q_STAGES;
    const int m = lane & 31, half = lane >> 5;
    float v[8];
    float s;
    if (stage < 8) {
        s = s_wm;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wm[(int64_t)(32 * (frag >> 1) + m) * 256 + 32 * stage + chunk_k(frag & 1, half, j)];
    } else if (stage >= TAIL_STAGES) {
        s = s_wq;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wq_next[(int64_t)(32 * (stage - TAIL_STAGES) + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
    } else {
        const int st = stage - 8;
        const bool up = st == 0 || (st < 63 && (st & 1));
        const int c = st == 0 ? 0 : st == 63 ? 31 : up ? (st + 1) / 2 : st / 2 - 1;
        s = up ? s_w1 : s_w2;
        if (up) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W1[(int64_t)(32 * c + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W2[(int64_t)(32 * (frag >> 1) + m) * 1024 + 32 * c + chunk_k(frag & 1, half, j)];
        }
    }
    typename SP::vec p[SP::NP];
#pragma unroll
    for (int j = 0; j < 8; ++j) SP::split1(SP::SCALED ? v[j] * s : v[j], j, p);
#pragma unroll
    for (int pl = 0; pl < SP::NP; ++pl) out[(((int64_t)stage_out * SP::NP + pl) * 16 + frag) * 64 + lane] = p[pl];
}

constexpr int KVF_THREADS = 320;  // kv_finalize_image_kernel: 264 threads x 4 consecutive elements = the 1 056 of a head

// Sum of the per-128-row-tile K^T V partials of the fused q/k/v projection (as kv_finalize_tiles_kernel, attention.hip) written
// as the operand image of tail_kernel: per cloud and head the A-operand fragments of KV_h^T / S (row m = value index
// v, contraction index d = chunk_k(step, half, j)), then Ksum as fp32.  H2 = false: three bf16 planes (exact split, no scale).
// H2 = true (round 4): two fp16 planes of KV_h^T / S * 2^e_h, e_h from the head's largest |element| (see KV_H2_SCALE_OFF above).
// grid (n_kv * 8, n_layers), block KVF_THREADS; layer l reads partial + l * partial_layer_stride floats and writes
// kvimg + l * image_layer_stride bytes.
template <bool H2>
__global__ __launch_bounds__(KVF_THREADS) void kv_finalize_image_kernel(const float* __restrict__ partial,
                                                                      const int32_t* __restrict__ cloud_row0,
                                                                      const int32_t* __restrict__ cloud_len, int64_t row_base,
                                                                      int cloud_begin, char* __restrict__ kvimg,
                                                                      int64_t partial_layer_stride, int64_t image_layer_stride) {
    constexpr int KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;
    static_assert(KV_ELEMS % 4 == 0 && KV_ELEMS / 4 <= KVF_THREADS, "");
    __shared__ float wmax[KVF_THREADS / 64];
    // ONE pass, four consecutive elements per thread (264 of the 320 threads): with 1024 threads and one element each, the 32
    // Ksum elements cost a second trip through the whole reduction for half a wave -- the launch is latency, not bandwidth.
    const int i4 = threadIdx.x;
    const bool active = i4 < KV_ELEMS / 4;
    const int kvi = blockIdx.x / SCREAM_NHEAD, h = blockIdx.x % SCREAM_NHEAD;
    const int cloud = cloud_begin + kvi;
    partial += (int64_t)blockIdx.y * partial_layer_stride;
    kvimg += (int64_t)blockIdx.y * image_layer_stride;
    const int t0 = (int)((cloud_row0[cloud] - row_base) / SCREAM_ROW_TILE);
    const int nt = (cloud_len[cloud] + SCREAM_ROW_TILE - 1) / SCREAM_ROW_TILE;
    const float* p = partial + ((int64_t)t0 * SCREAM_NHEAD + h) * KV_ELEMS + 4 * (active ? i4 : 0);
    char* img = kvimg + (size_t)cloud * KV_IMAGE_BYTES;
    const float S = (float)cloud_len[cloud];
    f32x4 s8[16];  // sixteen chains in a fixed combination order: deterministic (same order as kv_finalize_tiles_kernel)
#pragma unroll
    for (int u = 0; u < 16; ++u) s8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (int c = 0; c < nt; c += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c + u < nt) s8[u] += *reinterpret_cast<const f32x4*>(p + (int64_t)(c + u) * SCREAM_NHEAD * KV_ELEMS);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += s8[u + 8];
    const f32x4 s4 = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    const bool is_kv = 4 * i4 < 32 * 32;  // (the four elements of a thread are all K^T V or all Ksum: 1024 % 4 == 0)
    f32x4 x4;                             // values / v_length (models/transformer.py:38-39), applied to the sum
#pragma unroll
    for (int e = 0; e < 4; ++e) x4[e] = s4[e] / S;
    float scale = 1.f;
    if (H2) {
        // the head's largest |KV / S| -> e_h = the largest exponent with max 2^e_h <= 2^15, clamped to [-30, 40] (an all-zero head: 40)
        float m = (active && is_kv) ? fmaxf(fmaxf(fabsf(x4[0]), fabsf(x4[1])), fmaxf(fabsf(x4[2]), fabsf(x4[3]))) : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        m = wmax[0];
#pragma unroll
        for (int w = 1; w < KVF_THREADS / 64; ++w) m = fmaxf(m, wmax[w]);
        const uint32_t mb = __float_as_uint(m);
        int eh = 15 - ((int)(mb >> 23) - 127) - ((mb & 0x7fffffu) ? 1 : 0);  // m = 1.f x 2^E: f == 0 -> 15 - E, else 14 - E
        if (!(m >= 1.17549435e-38f) || eh > 40) eh = 40;                     // zero / subnormal / NaN maxima: nothing to protect
        if (eh < -30) eh = -30;                                               // (|KV / S| > 2^45: outside anything the bounds allow)
        scale = __uint_as_float((uint32_t)(eh + 127) << 23);
        if (threadIdx.x < 8) reinterpret_cast<float*>(img + KV_H2_SCALE_OFF)[h * 8 + threadIdx.x] = __uint_as_float((uint32_t)(127 - eh) << 23);
    }
    if (!active) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * i4 + e;
        if (is_kv) {
            const int d = i >> 5, v = i & 31;             // partial layout [d][v]
            const float x = x4[e];
            const int s2 = d >> 4, hf = (d >> 2) & 1, j = 4 * ((d >> 3) & 1) + (d & 3);  // d = chunk_k(s2, hf, j)
            if (H2) {
                const _Float16 a = (_Float16)(x * scale);
                const _Float16 b = (_Float16)__builtin_fmaf(x, scale, -(float)a);
                _Float16* base = reinterpret_cast<_Float16*>(img) + ((size_t)(h * 2) * 2 + s2) * 512 + (v + 32 * hf) * 8 + j;
                base[0] = a;
                base[2 * 512] = b;
            } else {
                const __bf16 a = (__bf16)x;
                const float r1 = x - (float)a;
                const __bf16 b = (__bf16)r1;
                const __bf16 cc = (__bf16)(r1 - (float)b);
                __bf16* base = reinterpret_cast<__bf16*>(img) + ((size_t)(h * 3) * 2 + s2) * 512 + (v + 32 * hf) * 8 + j;
                base[0] = a;
                base[2 * 512] = b;
                base[4 * 512] = cc;
            }
        } else {
            reinterpret_cast<float*>(img + KV_PLANES_BYTES)[h * 32 + (i - 32 * 32)] = s4[e];
        }
    }
}

bool split_ok(int32_t split) { return split == SCREAM_SPLIT_BF3 || split == SCREAM_SPLIT_H2 || split == SCREAM_SPLIT_H1; }

// the kernel's factors from the six exponents; false if one leaves the range the arithmetic was checked for
bool tail_scales(const scream_tail_exps_t* ex, TailScales* sc) {
    if (!ex) return false;
    const int es[6] = {ex->e_att, ex->e_wm, ex->e_m1, ex->e_w1, ex->e_h, ex->e_w2};
    for (int e : es)
        if (e < -40 || e > 40) return false;
    const int e1 = ex->e_wm + ex->e_att, e2 = ex->e_w2 + ex->e_h, eh = ex->e_h - ex->e_w1 - ex->e_m1;
    if (e1 < -44 || e1 > 44 || e2 < -44 || e2 > 44 || eh < -100 || eh > 100) return false;  // c^2 and c^2 * var stay finite in fp32
    if (ex->e_q < -40 || ex->e_q > 40) return false;
    sc->s_att = exp2i(ex->e_att);
    sc->s_q = exp2i(ex->e_q);
    sc->s_attq = exp2i(ex->e_att - ex->e_q);
    sc->c1 = exp2i(e1);
    sc->eps1 = 1e-5f * exp2i(2 * e1);
    sc->s_m1 = exp2i(ex->e_m1);
    sc->ch = exp2i(eh);
    sc->c2 = exp2i(e2);
    sc->eps2 = 1e-5f * exp2i(2 * e2);
    if (ex->e_y < -40 || ex->e_y > 40 || ex->e_wq < -40 || ex->e_wq > 40) return false;
    sc->s_y = exp2i(ex->e_y);
    sc->cq = exp2i(-ex->e_y - ex->e_wq);
    if (ex->e_x < -40 || ex->e_x > 40) return false;
    sc->s_x = exp2i(ex->e_x);
    sc->cqf = exp2i(-ex->e_x - ex->e_wq);
    return true;
}

}  // namespace

extern "C" int64_t scream_tail_image_bytes(int32_t split, int32_t with_next_q) {
    if (!split_ok(split) || (with_next_q && split == SCREAM_SPLIT_BF3)) return SCREAM_EINVAL;
    return (int64_t)(TAIL_STAGES + (with_next_q ? NEXT_Q_STAGES : 0)) * split * 16 * 1024;
}
extern "C" int64_t scream_kv_image_bytes(void) { return KV_IMAGE_BYTES; }

extern "C" int scream_pack_tail(const float* Wm, const float* W1, const float* W2, const float* Wq_next, int32_t q_first, int32_t split,
                                const scream_tail_exps_t* exps, void* image, void* stream) {
    SCREAM_REQUIRE(Wm && W1 && W2 && image && split_ok(split), SCREAM_EINVAL);
    SCREAM_REQUIRE(!Wq_next || split != SCREAM_SPLIT_BF3, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(!q_first || Wq_next, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, SCREAM_EINVAL);
    const dim3 grid((TAIL_STAGES + (Wq_next ? NEXT_Q_STAGES : 0)) * 16 * 64 / 256), block(256);
    if (split != SCREAM_SPLIT_BF3) {
        TailScales sc;
        SCREAM_REQUIRE(tail_scales(exps, &sc), SCREAM_EINVAL);
        if (split == SCREAM_SPLIT_H2)
            pack_tail_kernel<SplitH2><<<grid, block, 0, as_stream(stream)>>>(Wm, W1, W2, Wq_next, q_first, exp2i(exps->e_wm), exp2i(exps->e_w1),
                                                                             exp2i(exps->e_w2), exp2i(exps->e_wq), reinterpret_cast<f16x8*>(image));
        else
            pack_tail_kernel<SplitH1><<<grid, block, 0, as_stream(stream)>>>(Wm, W1, W2, Wq_next, q_first, exp2i(exps->e_wm), exp2i(exps->e_w1),
                                                                             exp2i(exps->e_w2), exp2i(exps->e_wq), reinterpret_cast<f16x8*>(image));
    } else {
        pack_tail_kernel<SplitBf3><<<grid, block, 0, as_stream(stream)>>>(Wm, W1, W2, nullptr, 0, 1.f, 1.f, 1.f, 1.f, reinterpret_cast<bf16x8*>(image));
    }
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_kv_finalize_image(const float* kv_partial, const int32_t* cloud_row0, const int32_t* cloud_len,
                                        int64_t row_base, int32_t cloud_begin, int32_t n_kv, void* kv_image, int32_t n_layers,
                                        int64_t partial_layer_stride, int64_t image_layer_stride, int32_t split, void* stream) {
    SCREAM_REQUIRE(kv_partial && cloud_row0 && cloud_len && kv_image && split_ok(split), SCREAM_EINVAL);
    SCREAM_REQUIRE(n_kv >= 0 && cloud_begin >= 0 && row_base >= 0 && n_layers >= 1 && n_layers <= 65535, SCREAM_EINVAL);
    SCREAM_REQUIRE(n_layers == 1 || (partial_layer_stride > 0 && image_layer_stride > 0 && image_layer_stride % 16 == 0), SCREAM_EINVAL);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(kv_image) | reinterpret_cast<uintptr_t>(kv_partial)) & 15) == 0 && partial_layer_stride % 4 == 0,
                   SCREAM_EINVAL);
    if (n_kv == 0) return 0;
    const dim3 grid(n_kv * SCREAM_NHEAD, n_layers), block(KVF_THREADS);
    if (split != SCREAM_SPLIT_BF3 && T_APPLY_H2)
        kv_finalize_image_kernel<true><<<grid, block, 0, as_stream(stream)>>>(kv_partial, cloud_row0, cloud_len, row_base, cloud_begin,
                                                                              reinterpret_cast<char*>(kv_image), partial_layer_stride, image_layer_stride);
    else
        kv_finalize_image_kernel<false><<<grid, block, 0, as_stream(stream)>>>(kv_partial, cloud_row0, cloud_len, row_base, cloud_begin,
                                                                               reinterpret_cast<char*>(kv_image), partial_layer_stride, image_layer_stride);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

namespace {
template <class SP, bool NQ, bool QF = false>
void launch_tail(unsigned grid, hipStream_t st, const float* Q, const void* kv_image, const int32_t* tile_cloud, int32_t kv_cloud_offset,
                 const int32_t* cloud_len, const float* x, const void* tail_image, const float* g1, const float* b1, const float* g2,
                 const float* b2, float* y, float* q_next, int tiles, const TailScales& sc) {
    tail_kernel<SP, NQ, QF><<<dim3(grid), dim3(TT), 0, st>>>(Q, reinterpret_cast<const char*>(kv_image), tile_cloud, kv_cloud_offset, cloud_len, x,
                                                         reinterpret_cast<const char*>(tail_image), g1, b1, g2, b2, y, q_next, tiles, sc);
}
}  // namespace

extern "C" int scream_layer_tail_f32(const float* Q, const void* kv_image, const int32_t* tile_cloud,
                                     int32_t kv_cloud_offset, const int32_t* cloud_len, const float* x,
                                     const void* tail_image, const float* g1, const float* b1, const float* g2,
                                     const float* b2, float* y, float* q_next, int64_t M, int32_t split,
                                     const scream_tail_exps_t* exps, void* stream) {
    // Q == NULL: the image was packed with q_first (scream_pack_tail) -- the kernel computes Q' = elu(x . Wq^T) + 1 itself (fp16 splits)
    SCREAM_REQUIRE(kv_image && tile_cloud && cloud_len && x && tail_image && g1 && b1 && g2 && b2 && y && split_ok(split), SCREAM_EINVAL);
    SCREAM_REQUIRE(Q || (split != SCREAM_SPLIT_BF3 && (!q_next || T_QF_DUMP >= 4)), SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(kv_image) | reinterpret_cast<uintptr_t>(x) |
                     reinterpret_cast<uintptr_t>(tail_image) | reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(b1) |
                     reinterpret_cast<uintptr_t>(g2) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(y) |
                     reinterpret_cast<uintptr_t>(q_next)) & 15) == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(x != y, SCREAM_EINVAL);  // the residual of a row is read twice, long after its neighbours were written
    // q_next (the image then has its eight query stages): fp16 splits only; it may be Q itself, never x or y
    SCREAM_REQUIRE(!q_next || (split != SCREAM_SPLIT_BF3 && q_next != x && q_next != y), SCREAM_EINVAL);
    TailScales sc{1.f, 1.f, 1.f, 1.f, 1e-5f, 1.f, 1.f, 1.f, 1e-5f, 1.f, 1.f, 1.f, 1.f};
    if (split != SCREAM_SPLIT_BF3) SCREAM_REQUIRE(tail_scales(exps, &sc), SCREAM_EINVAL);
    const int64_t tiles = M / SCREAM_ROW_TILE;
    if (tiles == 0) return 0;
    SCREAM_REQUIRE(tiles < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = tiles < T_MAX_GRID ? (unsigned)tiles : (unsigned)T_MAX_GRID;
    hipStream_t st = as_stream(stream);
#define TAIL_ARGS grid, st, Q, kv_image, tile_cloud, kv_cloud_offset, cloud_len, x, tail_image, g1, b1, g2, b2, y, q_next, (int)tiles, sc
    if (split == SCREAM_SPLIT_H2) {
        if (!Q) launch_tail<SplitH2, false, true>(TAIL_ARGS);
        else if (q_next) launch_tail<SplitH2, true>(TAIL_ARGS);
        else launch_tail<SplitH2, false>(TAIL_ARGS);
    } else if (split == SCREAM_SPLIT_H1) {
        if (!Q) launch_tail<SplitH1, false, true>(TAIL_ARGS);
        else if (q_next) launch_tail<SplitH1, true>(TAIL_ARGS);
        else launch_tail<SplitH1, false>(TAIL_ARGS);
    } else {
        launch_tail<SplitBf3, false>(TAIL_ARGS);
    }
#undef TAIL_ARGS
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
