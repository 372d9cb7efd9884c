// The q / k / v projections of an MHAttention block (models/transformer.py:79-81 with the elu(.) + 1 of :28-29 on q and k) and the
// fused K^T V reduction (:38-41) as a RING kernel on the fp16 matrix cores of gfx950 (round 4) -- the geometry of the layer tail
// (tail_split.hip: one wave per SIMD, the weights streamed through a ring of LDS stages by LDS-DMA, one barrier per stage) with
// SIXTY-FOUR rows per wave:
//
//   * a wave owns two 32-row groups of x and keeps BOTH groups' operand planes (split.h: x 2^e = x0 + x1 in fp16) in registers
//     for the whole 256-row tile: x is read and split ONCE per tile, where gemm_split_kernel re-reads and re-splits its A rows for
//     every 256-column tile (three times for q | k | v; PMC: 1.9 x the input fetched);
//   * every weight fragment read from LDS feeds SIX matrix instructions (three per row group) instead of three: half the LDS
//     fragment reads per product of both older kernels, and a stage -- one 32-column chunk of W over the whole K = 256, 32 KiB
//     -- carries 96 MFMAs per wave between two barriers instead of 48;
//   * the epilogue of a chunk RIDES inside the MFMA groups of the next stage (the accumulators are double buffered):
//       - a query chunk is computed transposed (A = weights, B = x rows): lane = row, registers = features, which IS the
//         fragment-major layout of Q' (include/scream_hip.h) -- elu + 1 and four 1 KiB stores per row group, no LDS;
//       - a key or value chunk is computed the other way round (A = x rows, B = weights): lane = feature, registers = rows --
//         the operand layout of a product whose contraction index is the ROW.  K' = elu(k) + 1 (padding rows zeroed) and V are
//         split into fp16 planes straight from the accumulators and K'^T V of the wave's 64 rows is 12 MFMAs; the two waves
//         of a 128-row tile add their tiles through 8 KiB of LDS in a fixed order and write the per-tile partial
//         [head][33][32] that scream_kv_finalize_image sums per cloud.  K' and V never exist in memory.
//     In the 8-wave GEMM a query tile's epilogue (a third of its time) ran with the matrix pipe idle -- its block-wide k-loop
//     barriers keep both waves of a SIMD in lockstep.
//     (Built first with LDS-DMA "touches" -- one dword per 128-byte line of the NEXT tile's rows, one instruction per stage -- to have
//     the rows in the L2 before the tile boundary reads them: 3-5 % SLOWER and 3.7 x the HBM fetch traffic by the counters, the
//     narrow loads are fetched line by line and do not stay; removed -- profiles/r04_ring_proj_prefetch_experiment.txt.)
//   * work is cut into UNITS of two stages (two query chunks, or the K and V chunk of one head) and every block takes a
//     CONTIGUOUS range of the launch's units, tile-major: no partial last round (1 302 row tiles on 256 CUs were 5.09 -> 6 rounds
//     of the old persistent grid), at the price of one extra x tile load per block.  A row's results do not depend on how its
//     tile was cut: chunks and heads are independent outputs.
//
// Stage sequence per 256-row tile (= image order): Q chunk 0 .. 7 (absent for a key/value-only call), then per layer of the
// call and head h: K_h, V_h.  Vector-memory discipline: the ring's barrier waits with a COUNTED vmcnt that leaves the pieces
// issued during the previous stage in flight; every store of a stage is issued before that stage's pieces (a counted wait is
// only sound if what it leaves in flight are loads: stores retire out of order against loads); the only register loads are the
// x rows at a tile boundary, plain loads that hipcc waits for with vmcnt(0).
#include <type_traits>

#include "ring.h"

#ifndef P_PF
#define P_PF 2  // register sets of weight fragments: fragments are read P_PF - 1 MFMA groups ahead (a third set does not fit: 512 registers)
#endif
#ifndef P_LB
#define P_LB 2  // x segments per load batch at a tile boundary (two batches in flight)
#endif
#ifndef P_NT
#define P_NT 3  // non-temporal hint on: 1 the x row loads, 2 the Q' stores, 4 the K^T V partial stores
#endif
#ifndef P_ABLATE
#define P_ABLATE 0  // tuning aid (SCREAM_HIPCC_EXTRA builds): 1 no rides (epilogues dropped), 2 no MFMA, 8 no W DMA after the first two stages (-DT_ABLATE=4: no LDS fragment reads);
                    // inside the rides: 32 elu + 1 without its exponential, 64 no Q' stores, 128 no K'^T V products, 256 no operand splits of K' and V, 512 no slab / partial traffic
#endif

namespace {

constexpr int P_KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;  // 1056
constexpr int P_SLAB_BYTES = 4 * P_KV_ELEMS * 4;                      // one K^T V tile + Ksum per wave
constexpr int P_MAX_GRID = SCREAM_MAX_GRID;

struct ProjArgs {
    const float* x;        // fragment-major [M, 256]
    const char* Wimg;      // [S][NP][16][64][16 B]
    float* Q;              // fragment-major [M, 256] (n_q == 8) or NULL
    int64_t M;
    int32_t n_q;           // query stages per tile: 0 or 8
    int32_t S;             // stages per tile: n_q + 16 L
    float* kv_partial;     // [L][M / 128][8][1056]
    int64_t kv_layer_stride;
    const int32_t* tile_cloud;
    const int32_t* cloud_row0;
    const int32_t* cloud_len;
    int64_t row_base;
    float a_scale, kv_sk, kv_cv, kv_inv;  // 2^a_exp; 2^k_exp; 2^(v_exp - a_exp - w_exp); 2^-(k_exp + v_exp)
    Elu1Consts ec;                         // of c = 2^-(a_exp + w_exp): accumulator -> q, k (common.h: elu1s)
};

template <class SP>
__global__ __launch_bounds__(TT, 1) void proj_ring_kernel(ProjArgs pa) {
    typedef typename SP::vec V;
    constexpr int NP = SP::NP;
    constexpr int STAGE = stage_bytes<SP>();
    constexpr int PIECES = wave_pieces<SP>();   // LDS-DMA weight pieces per wave and stage
    constexpr int INFLIGHT = PIECES;            // what a ring wait leaves in flight: the pieces issued during the previous stage
    constexpr int NV = 6;                       // ride slots behind every MFMA
    __shared__ __attribute__((aligned(16))) char smem[T_SLOTS * STAGE + P_SLAB_BYTES];  // the ONLY LDS object
    float* slabs = reinterpret_cast<float*>(smem + T_SLOTS * STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, r = lane & 31;
    const unsigned v_lane16 = lane * 16;

    // ---- this block's units -----------------------------------------------------------------------------------------------
    const int S = pa.S, UPT = S >> 1, nq2 = pa.n_q >> 1;
    const int64_t n_tiles = (pa.M + 255) >> 8;
    const int64_t U = n_tiles * UPT;
    const int64_t u_lo = U * blockIdx.x / gridDim.x, u_hi = U * (blockIdx.x + 1) / gridDim.x;
    if (u_lo >= u_hi) return;
    const int s0 = (int)(u_lo % UPT) * 2;  // image stage of the block's first ring stage

    unsigned q = 0;  // ring stages consumed so far
    auto dma_piece = [&](unsigned qq, int u) __attribute__((always_inline)) {
        if ((P_ABLATE & 8) && qq >= 2) return;
        const unsigned img = (unsigned)(s0 + qq) % (unsigned)S, slot = qq % (unsigned)T_SLOTS;
        const char* sbase = pa.Wimg + (size_t)img * STAGE + (wave * PIECES + (u & ~3)) * 1024;
        dma_1k(sbase + v_lane16, smem + slot * STAGE + (wave * PIECES + (u & ~3)) * 1024, u & 3);
    };
#pragma unroll
    for (int u = 0; u < PIECES; ++u) dma_piece(0, u);
#pragma unroll
    for (int u = 0; u < PIECES; ++u) dma_piece(1, u);

    // ---- per-tile state ---------------------------------------------------------------------------------------------------
    V xp[2][16][NP];     // operand planes of the wave's two row groups: xp[rg][2 blk + s2] = 16-deep step s2 of feature segment blk
    int64_t tile = -1;
    bool wave_ok = false;      // the wave's rows exist (the last tile of an odd number of 128-row tiles has two idle waves)
    int valid0 = 0;            // real tokens among the wave's 64 rows and behind (padding rows do not exist for K^T V)
    float* part_tile = nullptr;  // kv_partial of the wave's 128-row tile, layer 0, head 0
    float* qg = nullptr;         // Q + first float of the wave's 64 rows + this lane's 4 floats

    auto enter_tile = [&](int64_t t) __attribute__((always_inline)) {
        tile = t;
        const int64_t t128 = t * 2 + (wave >> 1);
        wave_ok = t128 * 128 < pa.M;
        const int64_t row0 = wave_ok ? t * 256 + wave * 64 : t * 256;  // (idle waves recompute the tile's first rows; nothing of it is stored)
        const float* g = pa.x + row0 * SCREAM_D_MODEL + lane * 4;
        valid0 = 0;
        if (wave_ok && pa.S > pa.n_q) {
            const int64_t arow = pa.row_base + t128 * 128;
            const int cloud = pa.tile_cloud[arow / SCREAM_ROW_TILE];
            valid0 = pa.cloud_len[cloud] - (int)(arow - pa.cloud_row0[cloud]) - (wave & 1) * 64;
        }
        part_tile = pa.kv_partial + t128 * SCREAM_NHEAD * P_KV_ELEMS;
        qg = pa.Q + row0 * SCREAM_D_MODEL + lane * 4;
        // the wave's 64 rows, P_LB segments of one row group (4 P_LB loads, 16 P_LB registers) at a time, two batches in flight:
        // the next batch is requested before the previous one is split (the whole tile at once would not fit beside the
        // pending accumulators and the K' planes)
        constexpr int LB = P_LB, NB = 16 / LB;  // batch b: row group b / (NB / 2), segments LB (b % (NB / 2)) .. + LB - 1
        f32x4 raw[2][LB][4];
        auto request = [&](int b) __attribute__((always_inline)) {
            const int rg = b / (NB / 2), blk0 = LB * (b % (NB / 2));
#pragma unroll
            for (int k = 0; k < LB; ++k)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    raw[b & 1][k][a] = (P_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + rg * 8192 + ((blk0 + k) * 4 + a) * 256))
                                                  : *reinterpret_cast<const f32x4*>(g + rg * 8192 + ((blk0 + k) * 4 + a) * 256);
        };
        request(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int rg = b / (NB / 2), blk0 = LB * (b % (NB / 2));
            __builtin_amdgcn_sched_barrier(0);
            if (b + 1 < NB) request(b + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < LB; ++k)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                {
                    V tmp[NP];
#pragma unroll
                    for (int j = 0; j < 8; ++j) SP::split1s(raw[b & 1][k][2 * s2 + (j >> 2)][j & 3], pa.a_scale, j, tmp);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        // the planes LIVE in accumulation registers (the MFMA reads its A / B operand from either file): 256 of them,
                        // the whole AGPR file -- left to itself hipcc keeps them VGPR-class and moves each one back before every use
                        asm volatile("" : "+a"(tmp[p]));
                        xp[rg][2 * (blk0 + k) + s2][p] = tmp[p];
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- pipeline state: the accumulators of a stage are consumed by the ride of the next one -----------------------------------
    f32x16 acc[2][2];       // [stage parity][row group]
    f16x8 kp[2][2][2];      // K' planes [row group][16-row step][plane]: A operand of K'^T V (the reduction runs on fp16 x 2 for every SP)
    f32x16 kv;              // K'^T V of the wave's 64 rows: lane = v, registers = d
    float ks = 0.f;         // Ksum[d = r] over the wave's rows
    float* q_pend = nullptr;     // where the pending query chunk goes (its tile's qg + chunk offset)
    bool q_pend_ok = false;
    float* part_pend = nullptr;  // where the pending head's partial goes
    bool part_pend_ok = false;

    // one MFMA group: 16-deep step g of both row groups against the same weight fragment
    // KIND 0: acc^T += W . x^T (lane = row, registers = features); KIND 1: acc += x . W^T (lane = feature, registers = rows)
    auto group = [&](auto kind, f32x16 (&a)[2], const V (&w)[NP], int g, bool zero) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind)::value;
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            if (P_ABLATE & 2) {
                if (zero) a[rg] = z;
                a[rg][0] += (float)w[0][0] + (float)w[NP - 1][1] + (float)xp[rg][g][0][0] + (float)xp[rg][g][NP - 1][1];
                continue;
            }
            if (KIND == 0) SP::products(a[rg], w, xp[rg][g], zero ? z : a[rg]);
            else SP::products(a[rg], xp[rg][g], w, zero ? z : a[rg]);
        }
        if (P_ABLATE & 2) return;
        // first MFMA, then the prefetch reads of the next fragments (one per plane), then the other MFMAs with NV ride slots each
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
#pragma unroll
        for (int i = 0; i < 2 * SP::NPROD - 1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // One ring stage: 16 groups = one 32-column chunk of W over K = 256 against the wave's 64 rows.  ride(g): the previous
    // chunk's epilogue, cut into per-group pieces; its stores only in groups <= 8, this stage's weight pieces from group 9 on.
    auto stage = [&](auto kind, f32x16 (&a)[2], auto ride) __attribute__((always_inline)) {
        ring_barrier<INFLIGHT>();
        __builtin_amdgcn_sched_barrier(0);
        const char* wb = smem + (q % T_SLOTS) * STAGE + lane * 16;
        V wf[P_PF][NP];
#pragma unroll
        for (int g0 = 0; g0 < P_PF - 1; ++g0)
#pragma unroll
            for (int p = 0; p < NP; ++p) wf[g0][p] = ld_frag<V>(wb + (p * 16 + g0) * 1024);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + P_PF - 1 < 16) {
#pragma unroll
                for (int p = 0; p < NP; ++p) wf[(g + P_PF - 1) % P_PF][p] = ld_frag<V>(wb + (p * 16 + g + P_PF - 1) * 1024);
            }
            if (g >= 9 && g < 13) {  // the PIECES weight pieces of the stage after next, two per group, behind every store of this stage's ride
#pragma unroll
                for (int u = (g - 9) * 2; u < (g - 8) * 2; ++u)
                    if (u < PIECES) dma_piece(q + 2, u);
            }
            if (!(P_ABLATE & 1)) ride(g);
            group(kind, a, wf[g % P_PF], g, g == 0);
        }
        ++q;
    };
    static_assert(wave_pieces<SP>() <= 8, "the request schedule of a stage holds eight pieces");
    constexpr std::integral_constant<int, 0> kindQ{};
    constexpr std::integral_constant<int, 1> kindKV{};

    // ---- rides --------------------------------------------------------------------------------------------------------------
    // Q' chunk = elu(q) + 1 in place, fragment-major: piece a of an accumulator tile is one 1 KiB wave store (tail_split.hip: store_chunk)
    auto ride_q = [&](f32x16 (&t)[2], int g) __attribute__((always_inline)) {
        if (g < 8) {
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int e = 0; e < 2; ++e) t[rg][2 * g + e] = (P_ABLATE & 32) ? fmaxf(__builtin_fmaf(t[rg][2 * g + e], pa.ec.c, 1.0f), 0.25f) : elu1s(t[rg][2 * g + e], pa.ec);
        }
        if (g == 8) {
            __builtin_amdgcn_sched_barrier(0);
            if (q_pend_ok && (!(P_ABLATE & 64) || t[0][0] == 123.456f)) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x4 o = {t[rg][4 * a], t[rg][4 * a + 1], t[rg][4 * a + 2], t[rg][4 * a + 3]};
                        if (P_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(q_pend + rg * 8192 + a * 256));
                        else *reinterpret_cast<f32x4*>(q_pend + rg * 8192 + a * 256) = o;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // K' = elu(k) + 1 with the padding rows zeroed, its row sum and its two fp16 planes: one register of both row groups per group.
    // MASK (compile time; wave-uniform per tile): only a cloud's last tile has padding rows -- everywhere else the compare and
    // select per element (two of a dozen vector instructions; at the socket power cap every one of them is paid in time) are left out
    auto ride_k = [&](auto mask, const f32x16 (&t)[2], int valid, int g) __attribute__((always_inline)) {
        if (g == 0) ks = 0.f;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            float a = (P_ABLATE & 32) ? fmaxf(__builtin_fmaf(t[rg][g], pa.ec.c, 1.0f), 0.25f) : elu1s(t[rg][g], pa.ec);
            if (decltype(mask)::value && (g & 3) + 8 * (g >> 2) + 32 * rg >= valid) a = 0.f;  // row mfma32_row(g, half) + 32 rg of the wave's 64 (valid carries the half)
            ks += a;
            if (P_ABLATE & 256) kp[rg][g >> 3][0][g & 7] = (_Float16)a; else SplitH2::split1s(a, pa.kv_sk, g & 7, kp[rg][g >> 3]);
        }
        if (g == 15) ks += __shfl_xor(ks, 32);
    };
    // V's planes, the 12 MFMAs of K'^T V over the wave's 64 rows, the wave's tile into its LDS slab
    f16x8 vp[2][2];  // [row group][plane] of the 16-row step being made
    auto ride_v = [&](const f32x16 (&t)[2], int g) __attribute__((always_inline)) {
        if (g < 8) {
            const int s2 = g >> 2;  // groups 0-3: rows' step 0, groups 4-7: step 1
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 8 * s2 + 2 * (g & 3) + e;
                    if (P_ABLATE & 256) vp[rg][0][i & 7] = (_Float16)t[rg][i]; else SplitH2::split1s(t[rg][i], pa.kv_cv, i & 7, vp[rg]);
                }
            if ((g & 3) == 3) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.f;
                if (P_ABLATE & 128) {
                    kv = s2 == 0 ? z : kv;
                    kv[0] += (float)kp[0][s2][0][0] + (float)vp[0][0][0] + (float)kp[1][s2][1][1] + (float)vp[1][1][1];
                } else {
                    SplitH2::products(kv, kp[0][s2], vp[0], s2 == 0 ? z : kv);
                    SplitH2::products(kv, kp[1][s2], vp[1], kv);
                }
            }
        }
        if (g == 9) kv *= pa.kv_inv;  // exact: a power of two (1 / v_length is applied once, in scream_kv_finalize_image)
        if (g >= 10 && g < 14 && (!(P_ABLATE & 512) || kv[0] == 123.456f)) {
            float* sw = slabs + wave * P_KV_ELEMS;
#pragma unroll
            for (int e = 4 * (g - 10); e < 4 * (g - 9); ++e) sw[mfma32_row(e, half) * 32 + r] = kv[e];  // [d][v]
            if (g == 13 && half == 0) sw[32 * 32 + r] = ks;
        }
    };
    // the two waves of a 128-row tile add their K'^T V tiles (even wave's first) and write the partial: 1 056 floats, half each
    auto ride_kvstore = [&](int g) __attribute__((always_inline)) {
        if (g > 8) return;
        const int i = (wave & 1) * 64 + lane + 128 * g;
        if (i < P_KV_ELEMS && part_pend_ok && !(P_ABLATE & 512)) {
            const float* s2 = slabs + (wave & ~1) * P_KV_ELEMS + i;
            if (P_NT & 4) __builtin_nontemporal_store(s2[0] + s2[P_KV_ELEMS], part_pend + i);
            else part_pend[i] = s2[0] + s2[P_KV_ELEMS];
        }
    };

    // ---- the block's units ----------------------------------------------------------------------------------------------------
    int prev = 0;  // what the previous unit left pending in acc[1]: 0 nothing, 1 a query chunk, 2 a value chunk (+ its head's K' planes)
    auto no_ride = [&](int) __attribute__((always_inline)) {};
#define LAMBDA(...) [&](__VA_ARGS__) __attribute__((always_inline))
    for (int64_t u = u_lo; u < u_hi; ++u) {
        const int64_t t = u / UPT;
        const int uu = (int)(u - t * UPT);
        if (t != tile) enter_tile(t);
        if (uu < nq2) {
            // ---- two query chunks 2 uu, 2 uu + 1
            float* q_even = qg + (2 * uu) * 1024;
            if (prev == 0) stage(kindQ, acc[0], no_ride);
            else if (prev == 1) stage(kindQ, acc[0], LAMBDA(int g) { ride_q(acc[1], g); });
            else stage(kindQ, acc[0], LAMBDA(int g) { ride_v(acc[1], g); });
            const bool store_kv = prev == 2;
            q_pend = q_even;
            q_pend_ok = wave_ok;
            if (store_kv) stage(kindQ, acc[1], LAMBDA(int g) { ride_kvstore(g); ride_q(acc[0], g); });
            else stage(kindQ, acc[1], LAMBDA(int g) { ride_q(acc[0], g); });
            q_pend = q_even + 1024;
            prev = 1;
        } else {
            // ---- head h of layer l: K chunk, then V chunk
            const int p = uu - nq2, l = p >> 3, h = p & 7;
            if (prev == 0) stage(kindKV, acc[0], no_ride);
            else if (prev == 1) stage(kindKV, acc[0], LAMBDA(int g) { ride_q(acc[1], g); });
            else stage(kindKV, acc[0], LAMBDA(int g) { ride_v(acc[1], g); });
            const bool store_kv = prev == 2;
            const int valid = valid0 - 4 * half;  // per lane: the compare below is then against a literal
            constexpr std::integral_constant<bool, true> masked{};
            constexpr std::integral_constant<bool, false> whole{};
            if (valid0 < 64) {  // (idle waves: valid0 == 0)
                if (store_kv) stage(kindKV, acc[1], LAMBDA(int g) { ride_kvstore(g); ride_k(masked, acc[0], valid, g); });
                else stage(kindKV, acc[1], LAMBDA(int g) { ride_k(masked, acc[0], valid, g); });
            } else {
                if (store_kv) stage(kindKV, acc[1], LAMBDA(int g) { ride_kvstore(g); ride_k(whole, acc[0], valid, g); });
                else stage(kindKV, acc[1], LAMBDA(int g) { ride_k(whole, acc[0], valid, g); });
            }
            part_pend = part_tile + (int64_t)l * pa.kv_layer_stride + h * P_KV_ELEMS;
            part_pend_ok = wave_ok;
            prev = 2;
        }
    }
    // ---- the pipeline's end, in the open
    __builtin_amdgcn_sched_barrier(0);
    if (!(P_ABLATE & 1)) {
        if (prev == 1) {
#pragma unroll
            for (int g = 0; g < 9; ++g) ride_q(acc[1], g);
        } else if (prev == 2) {
            lds_only_barrier();  // every wave has read the previous head's slabs (the ride of the stage just finished)
#pragma unroll
            for (int g = 0; g < 14; ++g) ride_v(acc[1], g);
            lds_only_barrier();
#pragma unroll
            for (int g = 0; g < 9; ++g) ride_kvstore(g);
        }
    }
#undef LAMBDA
    if (P_ABLATE & 1) {  // keep the accumulators alive
        float keep = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[b][rg][e];
        if (keep == 123.456f) pa.kv_partial[0] = keep;
    }
    VM_WAIT(0);  // the ring's last two stages (requested past the end) must have landed before the LDS is released
}

// W [N][256] fp32 in the row order of the q/k/v GEMM (include/scream_hip.h: q (n_q rows) | per layer: k heads 0-3 | v heads 0-3 |
// k heads 4-7 | v heads 4-7) -> the stage images of proj_ring_kernel, [S][NP][16 fragments][64 lanes][8] 16-bit values, S = N / 32,
// in the order the kernel consumes them: query chunk 0 .. 7, then per layer and head K_h, V_h.  Lane (m, half) of fragment f holds
// row 32 c + m of the chunk, contraction indices 32 (f >> 1) + chunk_k(f & 1, half, 0 .. 7) -- the same for either operand side.
template <class SP>
__global__ void pack_proj_kernel(const float* __restrict__ W, int n_q, int S, float w_scale, typename SP::vec* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * 16 * 64) return;
    const int lane = t & 63, frag = (t >> 6) & 15, stage = t >> 10;
    const int m = lane & 31, half = lane >> 5;
    int row0;
    if (stage < n_q / 32) {
        row0 = 32 * stage;
    } else {
        const int st = stage - n_q / 32, p = st >> 1, l = p >> 3, h = p & 7, is_v = st & 1;
        row0 = n_q + l * 512 + (h >> 2) * 256 + is_v * 128 + (h & 3) * 32;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)(row0 + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
    typename SP::vec p[SP::NP];
#pragma unroll
    for (int j = 0; j < 8; ++j) SP::split1(v[j] * w_scale, j, p);
#pragma unroll
    for (int pl = 0; pl < SP::NP; ++pl) out[(((int64_t)stage * SP::NP + pl) * 16 + frag) * 64 + lane] = p[pl];
}

bool proj_split_ok(int32_t split) { return split == SCREAM_SPLIT_H2 || split == SCREAM_SPLIT_H1; }

}  // namespace

extern "C" int64_t scream_proj_image_bytes(int32_t N, int32_t split) {
    if (!proj_split_ok(split) || N <= 0 || N % 64 != 0) return SCREAM_EINVAL;
    return (int64_t)(N / 32) * split * 16 * 1024;
}

extern "C" int scream_pack_proj(const float* W, int32_t N, int32_t n_q, int32_t split, int32_t w_exp, void* image, void* stream) {
    SCREAM_REQUIRE(W && image && proj_split_ok(split), SCREAM_EINVAL);
    SCREAM_REQUIRE((n_q == 0 || n_q == SCREAM_D_MODEL) && N > n_q && (N - n_q) % 512 == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(w_exp >= -60 && w_exp <= 60 && (reinterpret_cast<uintptr_t>(image) & 15) == 0, SCREAM_EINVAL);
    const int S = N / 32;
    const dim3 grid(S * 16 * 64 / 256), block(256);
    if (split == SCREAM_SPLIT_H2)
        pack_proj_kernel<SplitH2><<<grid, block, 0, as_stream(stream)>>>(W, n_q, S, exp2i(w_exp), reinterpret_cast<f16x8*>(image));
    else
        pack_proj_kernel<SplitH1><<<grid, block, 0, as_stream(stream)>>>(W, n_q, S, exp2i(w_exp), reinterpret_cast<f16x8*>(image));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_proj_qkv_f32(const float* x, const void* proj_image, float* Q, int64_t M, int32_t N, int32_t n_q,
                                   const int32_t* tile_cloud, const int32_t* cloud_row0, const int32_t* cloud_len,
                                   int64_t row_base, float* kv_partial, int32_t split, int32_t a_exp, int32_t w_exp, int32_t k_exp,
                                   int32_t v_exp, void* stream) {
    SCREAM_REQUIRE(x && proj_image && kv_partial && tile_cloud && cloud_row0 && cloud_len && proj_split_ok(split), SCREAM_EINVAL);
    SCREAM_REQUIRE((n_q == 0 || n_q == SCREAM_D_MODEL) && N > n_q && (N - n_q) % 512 == 0 && (n_q == 0 || N == n_q + 512), SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0 && row_base >= 0 && row_base % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(n_q == 0 || Q, SCREAM_EINVAL);
    SCREAM_REQUIRE(a_exp >= -60 && a_exp <= 60 && w_exp >= -60 && w_exp <= 60 && k_exp >= -40 && k_exp <= 40 && v_exp >= -40 && v_exp <= 40, SCREAM_EINVAL);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(proj_image) | reinterpret_cast<uintptr_t>(Q) |
                     reinterpret_cast<uintptr_t>(kv_partial)) & 15) == 0, SCREAM_EINVAL);
    if (M == 0) return 0;
    ProjArgs pa;
    pa.x = x;
    pa.Wimg = reinterpret_cast<const char*>(proj_image);
    pa.Q = Q;
    pa.M = M;
    pa.n_q = n_q / 32;
    pa.S = N / 32;
    pa.kv_partial = kv_partial;
    pa.kv_layer_stride = (M / SCREAM_ROW_TILE) * SCREAM_NHEAD * (int64_t)P_KV_ELEMS;
    pa.tile_cloud = tile_cloud;
    pa.cloud_row0 = cloud_row0;
    pa.cloud_len = cloud_len;
    pa.row_base = row_base;
    pa.a_scale = exp2i(a_exp);
    pa.ec = elu1_consts(exp2i(-a_exp - w_exp));
    pa.kv_sk = exp2i(k_exp);
    pa.kv_cv = exp2i(-a_exp - w_exp + v_exp);
    pa.kv_inv = exp2i(-k_exp - v_exp);
    const int64_t units = ((M + 255) / 256) * (pa.S / 2);
    const unsigned grid = units < P_MAX_GRID ? (unsigned)units : (unsigned)P_MAX_GRID;
    if (split == SCREAM_SPLIT_H2) proj_ring_kernel<SplitH2><<<dim3(grid), dim3(TT), 0, as_stream(stream)>>>(pa);
    else proj_ring_kernel<SplitH1><<<dim3(grid), dim3(TT), 0, as_stream(stream)>>>(pa);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
