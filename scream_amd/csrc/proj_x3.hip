// The q / k / v projections of an MHAttention block (models/transformer.py:27-36: q_proj, k_proj, v_proj, elu(.) + 1 on
// q and k) with the fused K^T V reduction (models/transformer.py:38-41), in the geometry of tail_x3.hip: one wave per
// SIMD, 128-row tiles, all weights streamed through the three-stage LDS ring, fp32 accuracy by the 3-way bf16 split.
//
// Why a second projection kernel beside gemm_x3.hip's: the launch is bounded by the socket power cap, so what counts is
// joules per row.  The 8-wave GEMM splits its A rows into bf16 planes once per 256-column tile (three times for q|k|v),
// turns the query tile around in an LDS slab to store it and reduces K^T V with sixteen fp32 MFMAs per head; here
//   * the three bf16 planes of a wave's 32 rows of x are made ONCE per tile and stay in registers for all 24 chunks;
//   * a query chunk is computed transposed (A = weights, B = x rows), so its accumulator tile is the fragment-major
//     layout of Q' as it stands: elu + 1 and four 1 KiB stores, no LDS;
//   * a key / value chunk is computed the other way round (A = x rows, B = weights: the same registers, swapped), so
//     lane = feature and the registers walk the rows -- which IS the operand layout of the next product with the ROW as
//     contraction index: K'^T V of a head is twelve bf16 MFMAs on the split accumulators, straight from registers.
// K' and V never exist in memory.  The four waves' 32-row K^T V tiles are added through 12 KiB of LDS in a fixed order
// and wave 0 writes the per-tile partial [head][33][32] that scream_kv_finalize_x3 sums per cloud (same format as
// gemm_epilogue.h's).
//
// Stage sequence per tile (image order): Q chunk 0 .. 7 | K head 0, V head 0, K head 1, ... V head 7 (either part may be
// absent: the cross layers project q from the source rows and k, v from the target rows).  The epilogue of a stage rides
// inside the MFMA groups of the next one.  The only register loads are the x rows of the block's next tile, requested
// at the tile boundary and followed by a full drain; every store is older than the weight pieces that a later counted
// wait leaves in flight or harmless to it (loads retire in order among themselves: while a piece of stage s - 1 is
// outstanding all twelve of stage s are, so vmcnt(12) cannot return early whatever the stores do).
#include <type_traits>

#define RING_ASM_PADDED 1
#include "ring_x3.h"

namespace {

constexpr int P_XCH_BYTES = 12 * 1024 + 3 * 32 * 4;  // 4 destination waves x 3 sources x 1 KiB quarter tiles + 3 x Ksum[32]
constexpr int P_KV_ELEMS = (SCREAM_HEAD_DIM + 1) * SCREAM_HEAD_DIM;

// The x rows of the next tile are loaded into ACCUMULATION registers (gfx950 loads can target them): they stay pending
// across the tile-end section, whose arithmetic wants the ordinary registers -- with "=v" destinations hipcc parked the
// still-pending values in AGPRs right behind the loads (tools/asm_inflight_check.py).
__device__ __forceinline__ void ld_asm4_acc(f32x4 (&d)[4], const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\t"  // (hazard: ring_x3.h)
                 "global_load_dwordx4 %0, %4, %5\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                 "global_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&a"(d[0]), "=&a"(d[1]), "=&a"(d[2]), "=&a"(d[3]) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void pin_acc(f32x4& v) { asm volatile("" : "+a"(v)); }
// Stores, like the loads, as (uniform base in SGPRs) + (32-bit lane offset): hipcc's own stores take a 64-bit VGPR address per
// store site, keeps dozens of them across the stages and spills some -- and a spilled value that is reloaded inside a stage
// costs a vmcnt(0) in front of its use, i.e. a drain of the weight ring (tail_x3.hip).
__device__ __forceinline__ void st_asm(const void* sbase, unsigned voff, const f32x4& d) {
    // second hazard hipcc cannot see through an asm statement: a store of more than 64 bits of data reads its data registers
    // over several cycles, and a VALU write to one of them in the next cycle corrupts what is stored (the query-only variant
    // stored one wrong dword of four in a quarter of the lanes) -- two wait states behind the store.
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void st_asm1(const void* sbase, unsigned voff, float d) {
    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(voff), "v"(d), "s"(sbase) : "memory");
}

template <bool HAS_Q, bool HAS_KV>
__global__ __launch_bounds__(TT, 1) void proj_x3_kernel(const float* __restrict__ x,        // fragment-major [M, 256]
                                                        const __bf16* __restrict__ Wimg,    // [NS stages][48 KiB]
                                                        float* __restrict__ Qout,           // fragment-major [M, 256]
                                                        float* __restrict__ kv_partial,     // [M / 128][8][33 * 32]
                                                        const int32_t* __restrict__ tile_cloud,
                                                        const int32_t* __restrict__ cloud_row0,
                                                        const int32_t* __restrict__ cloud_len, int64_t row_base, int n_tiles) {
    constexpr int NS = (HAS_Q ? 8 : 0) + (HAS_KV ? 16 : 0);
    __shared__ __attribute__((aligned(16))) char smem[T_SLOTS * T_STAGE + P_XCH_BYTES];  // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, r = lane & 31;
    const unsigned v_lane16 = lane * 16;
    char* xch = smem + T_SLOTS * T_STAGE;
    // ---- work items -----------------------------------------------------------------------------------------------------
    // A block walks whole tiles (tile = block + k * grid) for the n_full rounds every block takes part in; the tiles left
    // over are cut into PARTS items by stage range -- the query chunks and the heads are independent outputs, so the parts of
    // a tile need no reduction and a row's result does not depend on how its tile was cut -- and dealt out the same way.
    // With 1300 tiles on 256 CUs the last round then costs a third of a tile instead of a whole one.
    constexpr int NQ8 = HAS_Q ? 8 : 0;
    constexpr int PARTS = (HAS_Q && HAS_KV) ? 3 : 2;
    struct Item {
        int tile, c0, c1, h0, h1;  // query chunks [c0, c1) (even bounds), heads [h0, h1); tile < 0: none
    };
    const int G = (int)gridDim.x;
    const int n_full = n_tiles >= G && G == T_MAX_GRID ? n_tiles / G : 0;  // (a launch with fewer items than CUs is all parts)
    const int n_left = n_tiles - n_full * G;
    auto item_of = [&](int k) __attribute__((always_inline)) {
        Item it{-1, 0, 0, 0, 0};
        if (k < n_full) {
            it = Item{(int)blockIdx.x + k * G, 0, NQ8, 0, HAS_KV ? 8 : 0};
        } else {
            const int u = (int)blockIdx.x + (k - n_full) * G;
            if (u < n_left * PARTS) {
                const int part = u % PARTS;
                it.tile = n_full * G + u / PARTS;
                if (HAS_Q && HAS_KV) {  // 12 + 6 + 6 stages; every part ends with heads (one kind of item end in this kernel)
                    if (part == 0) { it.c1 = 8; it.h1 = 2; }
                    else { it.h0 = 3 * part - 1; it.h1 = it.h0 + 3; }
                } else if (HAS_Q) {
                    it.c0 = 4 * part; it.c1 = it.c0 + 4;
                } else {
                    it.h0 = 4 * part; it.h1 = it.h0 + 4;
                }
            }
        }
        return it;
    };
    auto item_len = [&](const Item& it) __attribute__((always_inline)) { return (it.c1 - it.c0) + 2 * (it.h1 - it.h0); };
    auto item_img = [&](const Item& it, int pos) __attribute__((always_inline)) {  // image stage of the item's stage number pos
        const int nq = it.c1 - it.c0;
        const int k = pos - nq;
        return pos < nq ? it.c0 + pos : NQ8 + 2 * (it.h0 + (k >> 1)) + (k & 1);
    };
    int item_no = 0;
    Item cur = item_of(0), nxt = item_of(1);
    int pos = 0;       // stage of `cur` about to be consumed
    unsigned q = 0;    // stages consumed so far (ring slot = q % 3)
    auto img_ahead = [&](int d) __attribute__((always_inline)) {  // image stage of the stage d steps ahead in the block's sequence
        const int p = pos + d, len = item_len(cur);
        if (p < len) return item_img(cur, p);
        return nxt.tile >= 0 ? item_img(nxt, p - len) : 0;  // (past the end: any stage, never read)
    };
    auto dma_piece = [&](int img, unsigned slot_q, int u) __attribute__((always_inline)) {
        const unsigned slot = slot_q % (unsigned)T_SLOTS;
        const char* sbase = reinterpret_cast<const char*>(Wimg) + (size_t)img * T_STAGE + (wave * 12 + (u & ~3)) * 1024;
        dma_1k(sbase + v_lane16, smem + slot * T_STAGE + (wave * 12 + (u & ~3)) * 1024, u & 3);
    };
    if (cur.tile < 0) return;
    {
        const int i0 = img_ahead(0), i1 = img_ahead(1);
#pragma unroll
        for (int u = 0; u < 12; ++u) dma_piece(i0, 0, u);
#pragma unroll
        for (int u = 0; u < 12; ++u) dma_piece(i1, 1, u);
    }
    // (every lambda of this kernel is always_inline: one that hipcc leaves out of line takes xp / raw by address and both
    // arrays then live in scratch)
    // the three bf16 planes of the wave's 32 rows of x: xp[2 b + s2] = 16-deep step s2 of the 32-feature segment b
    bf16x8 xp[16][3];
    f32x4 raw[8][4];
    auto request_x = [&](int t) __attribute__((always_inline)) {
        const float* g = x + ((int64_t)t * 128 + wave * 32) * SCREAM_D_MODEL;
#pragma unroll
        for (int b = 0; b < 8; ++b) ld_asm4_acc(raw[b], g + b * 1024, v_lane16);
    };
    auto split_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int a = 0; a < 4; ++a) pin_acc(raw[b][a]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) split3(raw[b][2 * s2], raw[b][2 * s2 + 1], xp[2 * b + s2][0], xp[2 * b + s2][1], xp[2 * b + s2][2]);
        }
    };
    auto pin_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) pin_acc(raw[b][a]);
    };
    auto split_one = [&](int i) __attribute__((always_inline)) {  // xp[i] of the requested rows: segment i / 2, step i % 2
        const int b = i >> 1, s2 = i & 1;
        split3(raw[b][2 * s2], raw[b][2 * s2 + 1], xp[i][0], xp[i][1], xp[i][2]);
    };
    request_x(cur.tile);
    VM_WAIT(0);
    split_x();

    while (true) {
        asm volatile("" : "+s"(q));  // (opaque once per item: hipcc otherwise hoists per-stage addresses out of the loop and spills them)
        const int tile = cur.tile;
        const bool has_next = nxt.tile >= 0;
        const int tile_next = nxt.tile;
        const bool has_q = cur.c1 > cur.c0, has_kv = cur.h1 > cur.h0;  // what this item computes
        const int64_t grp = ((int64_t)tile * 128 + wave * 32) * SCREAM_D_MODEL;  // first float of the wave's 32-row group
        int valid_w = 32;  // real tokens among the wave's 32 rows (padding rows do not exist for K^T V)
        if (HAS_KV) {
            const int cloud = tile_cloud[(row_base + (int64_t)tile * 128) / SCREAM_ROW_TILE];
            valid_w = cloud_len[cloud] - (int)(row_base + (int64_t)tile * 128 - cloud_row0[cloud]) - wave * 32;
        }
        float* part = kv_partial + (int64_t)tile * SCREAM_NHEAD * P_KV_ELEMS;

        f32x16 tA, tB;       // accumulator tiles, alternating by stage; a finished one is consumed by the next stage's ride
        bf16x8 wfd[3];       // weight fragments of a stage's last MFMA group, issued behind the next barrier (tail_x3.hip)
        bf16x8 kp[2][3], vp[2][3];  // planes of K'_h (A operand) and V_h (B operand), contraction index = row
        f32x16 kv;           // K'^T V of the wave's 32 rows: lane = v, registers = d
        float ks = 0.f;      // Ksum[d = r] over the wave's rows of the head whose K' is being made
        float ks_out = 0.f;  // ... of the head whose K'^T V is on its way out (ks is reused one stage earlier than kv)

        // KIND 0: acc^T += W . x^T (lane = row, registers = features); KIND 1: acc += x . W^T (lane = feature, registers = rows)
        // XMODE 1: the tile's second-to-last stage requests the x rows of the block's next tile (behind its barrier and before
        //          its weight pieces: the last stage's vmcnt(12) then covers them);
        // XMODE 2: the tile's last stage.  Its group g is the last user of xp[g], so the planes of the next tile's rows are
        //          written in place one group behind the MFMAs: neither the load latency nor the split is ever in the open.
        auto stage = [&](auto kind, auto xmode, f32x16& acc, auto flush, auto ride) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind)::value;
            constexpr int XMODE = decltype(xmode)::value;
            ring_barrier<12>();
            __builtin_amdgcn_sched_barrier(0);
            const int img2 = img_ahead(2);  // the stage whose weights this one requests
            if (XMODE == 1) request_x(has_next ? tile_next : tile);
            if (XMODE == 2) pin_x();
            __builtin_amdgcn_sched_barrier(0);
            const char* wb = smem + (q % T_SLOTS) * T_STAGE + lane * 16;
            bf16x8 wf[T_PF][3];
#pragma unroll
            for (int g0 = 0; g0 < T_PF - 1; ++g0)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[g0][p] = ld_frag(wb + (p * 16 + g0) * 1024);
            flush();
#pragma unroll
            for (int g = 0; g < 15; ++g) {  // group 15 is deferred to the next stage
                if (g + T_PF - 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        (g + T_PF - 1 == 15 ? wfd[p] : wf[(g + T_PF - 1) % T_PF][p]) = ld_frag(wb + (p * 16 + g + T_PF - 1) * 1024);
                }
                if (g < 12) dma_piece(img2, q + 2, g);
                ride(g);
                if (XMODE == 2 && g >= 1) split_one(g - 1);
                if (KIND == 0) mfma6<4>(acc, wf[g % T_PF], xp[g], g == 0);
                else mfma6<4>(acc, xp[g], wf[g % T_PF], g == 0);
            }
            ++q;
            ++pos;
        };
        constexpr std::integral_constant<int, 0> kindQ{};
        constexpr std::integral_constant<int, 1> kindKV{};
        constexpr std::integral_constant<int, 0> x0{};
        constexpr std::integral_constant<int, 1> x1{};
        constexpr std::integral_constant<int, 2> x2{};
        auto flush_q = [&](f32x16& acc) __attribute__((always_inline)) { mfma6_free(acc, wfd, xp[15]); };
        auto flush_kv = [&](f32x16& acc) __attribute__((always_inline)) { mfma6_free(acc, xp[15], wfd); };

        // ---- rides ------------------------------------------------------------------------------------------------------
        // Q' chunk c = elu(q) + 1, fragment-major: piece a of the accumulator tile is one 1 KiB wave store
        // (stores address memory as uniform base + 32-bit lane offset, like the loads: no per-lane 64-bit pointer to keep)
        auto q_epilogue = [&](const f32x16& t, int c, int a) __attribute__((always_inline)) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = t[4 * a + k];
                o[k] = v > 0.f ? v + 1.0f : expf(v);  // elu(x) + 1 == exp(x), x <= 0
            }
            st_asm(Qout + grp + (c * 4 + a) * 256, v_lane16, o);
        };
        // K' = elu(k) + 1 with the padding rows zeroed, its row sum, and its three planes: elements 2k, 2k + 1
        auto k_pair = [&](const f32x16& t, int k) __attribute__((always_inline)) {
            const int s2 = k >> 2, j = (2 * k) & 7;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float a = t[2 * k + e];
                a = a > 0.f ? a + 1.0f : expf(a);
                if (mfma32_row(2 * k + e, half) >= valid_w) a = 0.f;
                ks += a;
                const __bf16 p0 = (__bf16)a;
                const float r1 = a - (float)p0;
                const __bf16 p1 = (__bf16)r1;
                kp[s2][0][j + e] = p0;
                kp[s2][1][j + e] = p1;
                kp[s2][2][j + e] = (__bf16)(r1 - (float)p1);
            }
        };
        auto v_pair = [&](const f32x16& t, int k) __attribute__((always_inline)) {
            const int s2 = k >> 2, j = (2 * k) & 7;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a = t[2 * k + e];
                const __bf16 p0 = (__bf16)a;
                const float r1 = a - (float)p0;
                const __bf16 p1 = (__bf16)r1;
                vp[s2][0][j + e] = p0;
                vp[s2][1][j + e] = p1;
                vp[s2][2][j + e] = (__bf16)(r1 - (float)p1);
            }
        };
        // The four waves' 32-row K'^T V tiles are added in a fixed order, a quarter of the registers (8 values of d) per wave:
        // wave w sends quarter a != w to wave a through LDS (slot [a][source index among the other three], 1 KiB each) and
        // its Ksum to wave 0; after the next barrier wave a adds (w0 + w1) + (w2 + w3) for its quarter and stores 4 x 128 bytes
        // per lane half.  Every wave has 3 LDS writes, 3 reads and 4 stores per head (one wave doing all of it issued 17
        // stores at the end of a stage, right in front of the counted wait).
        auto xch_write = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a == wave) continue;
                const int sidx = wave < a ? wave : wave - 1;
                const f32x4 v4 = {kv[4 * a], kv[4 * a + 1], kv[4 * a + 2], kv[4 * a + 3]};
                *reinterpret_cast<f32x4*>(xch + ((a * 3 + sidx) * 64 + lane) * 16) = v4;
            }
            if (wave > 0 && half == 0) *reinterpret_cast<float*>(xch + 12 * 1024 + ((wave - 1) * 32 + r) * 4) = ks;
            ks_out = ks;
        };
        auto xch_sum_store_a = [&](int h, auto aa) __attribute__((always_inline)) {  // executed by wave A
            constexpr int A = decltype(aa)::value;
            float* ph = part + h * P_KV_ELEMS;
            f32x4 w[4];
#pragma unroll
            for (int src = 0; src < 4; ++src) {
                if (src == A) w[src] = f32x4{kv[4 * A], kv[4 * A + 1], kv[4 * A + 2], kv[4 * A + 3]};
                else w[src] = *reinterpret_cast<const f32x4*>(xch + ((A * 3 + (src < A ? src : src - 1)) * 64 + lane) * 16);
            }
            const unsigned v_dv = (4 * half * 32 + r) * 4;  // the lane's part of [d][v]: d = 8 A + 4 half + k, v = r
#pragma unroll
            for (int k = 0; k < 4; ++k)
                st_asm1(ph + (8 * A + k) * 32, v_dv, (w[0][k] + w[1][k]) + (w[2][k] + w[3][k]));
            if (A == 0 && half == 0) {
                const float s1 = *reinterpret_cast<const float*>(xch + 12 * 1024 + r * 4);
                const float s2 = *reinterpret_cast<const float*>(xch + 12 * 1024 + (32 + r) * 4);
                const float s3 = *reinterpret_cast<const float*>(xch + 12 * 1024 + (64 + r) * 4);
                st_asm1(ph + 32 * 32, (unsigned)(r * 4), (ks_out + s1) + (s2 + s3));
            }
        };
        auto xch_sum_store = [&](int h) __attribute__((always_inline)) {
            if (wave == 0) xch_sum_store_a(h, std::integral_constant<int, 0>{});
            else if (wave == 1) xch_sum_store_a(h, std::integral_constant<int, 1>{});
            else if (wave == 2) xch_sum_store_a(h, std::integral_constant<int, 2>{});
            else xch_sum_store_a(h, std::integral_constant<int, 3>{});
        };
        // ride of a V stage: K' of the same head (finished accumulator tk); the PREVIOUS head's sum goes out first, so that
        // its stores are the oldest operations of the stage
        auto ride_after_k = [&](const f32x16& tk, int h_prev, bool sum_prev, int g) __attribute__((always_inline)) {
            if (g == 0 && sum_prev) xch_sum_store(h_prev);
            if (g == 0) ks = 0.f;
            if (g >= 1 && g <= 8) k_pair(tk, g - 1);
            if (g == 9) ks += __shfl_xor(ks, 32);
        };
        // ride of the stage behind a V stage: V's planes, the 12 MFMAs of K'^T V, the exchange write (kp is dead after group 10)
        auto ride_after_v = [&](const f32x16& tv, int g) __attribute__((always_inline)) {
            if (g >= 1 && g <= 8) v_pair(tv, g - 1);
            if (g == 9) mfma6_free(kv, kp[0], vp[0], true);
            if (g == 10) mfma6_free(kv, kp[1], vp[1]);
            if (g == 12) xch_write();
        };

        // ---- the stages of this item --------------------------------------------------------------------------------------
        // Its last two stages request the next item's rows and split them in place (x1 / x2).  Every item of a kernel ends the
        // same way (query chunks in the q-only kernel, heads in the other two): with request sites on different paths hipcc
        // assumes a path on which none runs, keeps the 128 row registers of the previous item alive through the whole body
        // and spills.
#define LAMBDA(...) [&](__VA_ARGS__) __attribute__((always_inline))
#define QPAIR(xa, xb)                                                                                              \
        {                                                                                                          \
            const bool first = c == cur.c0;                                                                        \
            stage(kindQ, xa, tA, LAMBDA() { if (!first) flush_q(tB); },                                            \
                  LAMBDA(int g) { if (!first && g >= 1 && g <= 4) q_epilogue(tB, c - 1, g - 1); });                \
            stage(kindQ, xb, tB, LAMBDA() { flush_q(tA); }, LAMBDA(int g) { if (g >= 1 && g <= 4) q_epilogue(tA, c, g - 1); }); \
        }
        if (HAS_Q && !HAS_KV) {
            // ---- query chunks only (the q-only kernel): two chunks per iteration
            int c = cur.c0;
            for (; c + 2 < cur.c1; c += 2) QPAIR(x0, x0)
            QPAIR(x1, x2)
            flush_q(tB);
            __builtin_amdgcn_sched_barrier(0);
            split_one(14);
            split_one(15);
#pragma unroll
            for (int a = 0; a < 4; ++a) q_epilogue(tB, cur.c1 - 1, a);
        } else if (HAS_KV) {
            // ---- (query chunks, then) heads
            if (HAS_Q && has_q)
                for (int c = cur.c0; c < cur.c1; c += 2) QPAIR(x0, x0)
            // first head: its K stage carries the epilogue of the last query chunk (if the item has any), its V stage the K' ride
            stage(kindKV, x0, tA, LAMBDA() { if (HAS_Q && has_q) flush_q(tB); },
                  LAMBDA(int g) { if (HAS_Q && has_q && g >= 1 && g <= 4) q_epilogue(tB, cur.c1 - 1, g - 1); });
            stage(kindKV, x0, tB, LAMBDA() { flush_kv(tA); }, LAMBDA(int g) { ride_after_k(tA, 0, false, g); });
            for (int h = cur.h0 + 1; h < cur.h1 - 1; ++h) {
                stage(kindKV, x0, tA, LAMBDA() { flush_kv(tB); }, LAMBDA(int g) { ride_after_v(tB, g); });               // K_h | V_{h-1}: planes, K'^T V
                stage(kindKV, x0, tB, LAMBDA() { flush_kv(tA); }, LAMBDA(int g) { ride_after_k(tA, h - 1, true, g); });  // V_h | K'_h; head h - 1 out
            }
            stage(kindKV, x1, tA, LAMBDA() { flush_kv(tB); }, LAMBDA(int g) { ride_after_v(tB, g); });                   // the item's last head
            stage(kindKV, x2, tB, LAMBDA() { flush_kv(tA); }, LAMBDA(int g) { ride_after_k(tA, cur.h1 - 2, true, g); });
            flush_kv(tB);
            // item end, in the open: the two x steps the last MFMA groups were still using, and the last head's K'^T V
            __builtin_amdgcn_sched_barrier(0);
            split_one(14);
            split_one(15);
#pragma unroll
            for (int g = 0; g < 11; ++g) ride_after_v(tB, g);
            lds_only_barrier();  // every wave has read the previous head's tiles (top of its last stage)
            xch_write();
            lds_only_barrier();
            xch_sum_store(cur.h1 - 1);
        }
#undef QPAIR
#undef LAMBDA
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        ++item_no;
        cur = nxt;
        nxt = item_of(item_no + 1);
        pos = 0;
    }
    VM_WAIT(0);  // the ring's last two stages (requested past the end) must have landed before the LDS is released
}

// Wq, Wk, Wv [256][256] fp32 (rows = output features; either Wq or the pair Wk, Wv may be null) -> the stage images of
// proj_x3_kernel: chunk c of a matrix is rows 32c .. 32c + 31, fragment f its 16-deep step f, stored [plane][fragment][lane][8].
__global__ void pack_proj_kernel(const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                 __bf16* __restrict__ out, int n_stages) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_stages * 16 * 64) return;
    const int lane = t & 63, frag = (t >> 6) & 15, stage = t >> 10;
    const int m = lane & 31, half = lane >> 5;
    const int nq = Wq ? 8 : 0;
    const float* W;
    int c;
    if (stage < nq) {
        W = Wq;
        c = stage;
    } else {
        const int st = stage - nq;
        W = (st & 1) ? Wv : Wk;
        c = st >> 1;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)(32 * c + m) * 256 + 32 * (frag >> 1) + chunk_k(frag & 1, half, j)];
    bf16x8 p0, p1, p2;
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    split3(lo, hi, p0, p1, p2);
    __bf16* dst = out + ((int64_t)stage * 3 * 16 + frag) * 64 * 8 + lane * 8;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + 16 * 64 * 8) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * 16 * 64 * 8) = p2;
}

}  // namespace

extern "C" int64_t scream_proj_image_bytes(int32_t has_q, int32_t has_kv) {
    return (int64_t)((has_q ? 8 : 0) + (has_kv ? 16 : 0)) * T_STAGE;
}

extern "C" int scream_pack_proj_x3(const float* Wq, const float* Wk, const float* Wv, void* image, void* stream) {
    SCREAM_REQUIRE(image && (Wq || (Wk && Wv)) && (!Wk == !Wv), SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, SCREAM_EINVAL);
    const int ns = (Wq ? 8 : 0) + (Wk ? 16 : 0);
    pack_proj_kernel<<<dim3(ns * 16 * 64 / 256), dim3(256), 0, as_stream(stream)>>>(Wq, Wk, Wv, reinterpret_cast<__bf16*>(image), ns);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int scream_proj_x3_f32(const float* x, const void* proj_image, int32_t has_q, int32_t has_kv, float* Q,
                                  float* kv_partial, const int32_t* tile_cloud, const int32_t* cloud_row0,
                                  const int32_t* cloud_len, int64_t row_base, int64_t M, void* stream) {
    SCREAM_REQUIRE(x && proj_image && (has_q || has_kv), SCREAM_EINVAL);
    SCREAM_REQUIRE(!has_q || Q, SCREAM_EINVAL);
    SCREAM_REQUIRE(!has_kv || (kv_partial && tile_cloud && cloud_row0 && cloud_len && row_base >= 0 && row_base % SCREAM_ROW_TILE == 0), SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(proj_image) | reinterpret_cast<uintptr_t>(Q) |
                     reinterpret_cast<uintptr_t>(kv_partial)) & 15) == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(x != Q, SCREAM_EINVAL);
    const int64_t tiles = M / SCREAM_ROW_TILE;
    if (tiles == 0) return 0;
    SCREAM_REQUIRE(tiles < (1ll << 31), SCREAM_EUNSUPPORTED);
    // fewer tiles than CUs: every tile goes out in parts (3 for q|k|v, 2 otherwise), see the kernel
    const int64_t items = tiles * ((has_q && has_kv) ? 3 : 2);
    const dim3 grid(tiles >= T_MAX_GRID ? (unsigned)T_MAX_GRID : (unsigned)(items < T_MAX_GRID ? items : T_MAX_GRID)), block(TT);
    const __bf16* img = reinterpret_cast<const __bf16*>(proj_image);
    hipStream_t st = as_stream(stream);
    if (has_q && has_kv) proj_x3_kernel<true, true><<<grid, block, 0, st>>>(x, img, Q, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base, (int)tiles);
    else if (has_q) proj_x3_kernel<true, false><<<grid, block, 0, st>>>(x, img, Q, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base, (int)tiles);
    else proj_x3_kernel<false, true><<<grid, block, 0, st>>>(x, img, Q, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base, (int)tiles);
    SCREAM_LAUNCH_CHECK();
    return 0;
}
