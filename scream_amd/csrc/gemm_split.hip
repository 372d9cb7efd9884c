// fp32-accurate GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  on the gfx950 16-bit matrix cores by operand splitting
// (split.h); the kernel is a template over the split:
//
//   SplitBf3:  x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   (exact: 3 x 8 significand bits)
//              a.b ~= a2b0 + a1b1 + a0b2 + a1b0 + a0b1 + a0b0                                (dropped terms <= 2^-24 relative)
//   SplitH2:   x 2^e = x0 + x1,  x0 = fp16(x 2^e), x1 = fp16(x 2^e - x0)                    (2 x 11 significand bits)
//              a.b ~= (a1b0 + a0b1 + a0b0) 2^-(ea + ew)                                      (half the matrix instructions)
//
// Every 16-bit x 16-bit product is exact in fp32 and the MFMAs of a 16-deep step accumulate in fp32, so the result carries
// fp32-level error -- measured against fp64 with the hardware's accumulation both are at or below the fp32-input MFMA path
// (max |err| / sum|a||b| on rows of mixed scale: 2.6-3.7e-7 for SplitBf3, 1.9-2.1e-7 for SplitH2, tools/ubench/split_acc.py,
// profiles/r03_split_acc.txt) -- while the matrix pipe spends 6 x 32 = 192 (3 x 32 = 96) cycles per 32x32x16 block instead of
// 8 x 64 = 512.  Same inputs, same outputs, same epilogues and tolerances as gemm_f32.hip; the dtype of the path stays fp32.
// SplitH2's power-of-two scales are arguments (a_exp, w_exp): the CALLER guarantees |A| 2^a_exp <= 2^15 (scream_amd/scales.py
// derives it from the weights for every GEMM of the forward); the accumulators are multiplied by 2^-(a_exp + w_exp), exactly,
// in front of the epilogue.
//
// Geometry: block tile 256 x 256, 512 threads = 8 waves stacked in M (wave tile 32 x 256, 128 accumulator VGPRs), one
// persistent block per CU.
//   * W is split and re-tiled once (scream_pack_w_split) into an image that is stored k-tile by k-tile exactly as it sits
//     in LDS: [NP planes][K/32][N][32] 16-bit values, rows of 64 B whose 16-byte chunk c lives at c ^ ((n >> 2) & 3) so that the 16
//     lanes of a ds_read_b128 group cover 16 distinct bank slots.  A k-tile stage is 16 NP one-KiB LDS-DMA pieces of
//     CONTIGUOUS memory (with a plain [N][K] plane every piece touched 16 lines for half their bytes); double buffered.
//   * A stays fp32 in HBM: lane (r, half) streams its 16 floats of row r per k-tile straight into registers (three
//     register sets, requested TWO k-tiles ahead) and splits them there (v_cvt_pk_bf16_f32 + subtract, twice).
//     Lane-half h owns k = 8 (2 s + (j >> 2)) + 4 h + (j & 3) of step s for both operands (the contraction index is a
//     dummy; this is the ownership of a transposed accumulator tile in tail_x3.hip, so the two kernels share one
//     activation layout).  Two layouts of A: row-major (the four 16-byte pieces 2a + h of the lane's 128-byte segment),
//     or FRAGMENT-major (SCREAM_ACT_FRAG, include/scream_hip.h): the same pieces stored [32-row group][segment][a][lane],
//     so that every wave instruction reads 1 KiB of contiguous memory instead of 32 half-used lines.
//   * One barrier per k-tile with COUNTED waits: requests are issued in a fixed order (D(kt+1), then A(kt+2), pinned
//     with sched_barrier) so that s_waitcnt vmcnt(4) at the barrier covers the W stage and leaves the A loads of the
//     k-tile after next in flight; the barrier is followed directly by MFMAs (the first step's operand split was done
//     at the end of the previous k-tile, the requests go out between the two steps).
//   * The A loads are plain inline asm: hipcc's own vmcnt bookkeeping cannot count across the epilogue's stores and
//     the loop back edge and would wait for vmcnt(0) everywhere.  Counted waits are used only where nothing but loads
//     is in flight; the first barrier of an output tile, behind the previous epilogue's stores, drains the queue.
//   * The two waves of a SIMD are not symmetric: the older one gets the matrix pipe first (s_memtime stamps,
//     tools/gemm_stamps.py: 1.8 k cycles for its first 48 MFMAs against 3.8 k for the younger wave's), so the younger
//     waves (4-7) do their operand split AFTER the barrier, where they would be starved anyway, the older ones before.
//   * A dedicated 64 KiB LDS region holds the epilogue slabs, so the first k-tile of the next output tile is already
//     in flight during the epilogue (for every epilogue kind).
// What was measured and rejected on the way (tools/gemm_ablate.py, tools/ubench/, profiles/r01_x3_ablation.txt): two
// independent 128-row blocks per CU (doubles the W traffic; same speed), one wave per SIMD with 64-row wave tiles and
// 512 registers (slower: a lone in-order wave does not keep the pipe full), spreading the fragment reads between the
// MFMAs, starting the CUs out of phase, non-temporal stores.  On this chip a wave that issues MFMAs back to back
// starves the LDS and vector-memory instructions of the other wave on its SIMD (not its VALU), which is why the
// k-tile time is close to the SUM of the MFMA, LDS, VMEM and VALU issue times rather than their maximum.
// Tuning aid (tools/gemm_ablate.py builds variants): bit 0 no epilogue, 1 no W DMA after the first k-tile, 2 no A loads
// after the first, 3 no MFMAs, 4 no LDS fragment reads, 5 no operand split, 6 no epilogue stores, 7 no key/value-tile epilogue
// (the fused K^T V reduce), 8 no query-tile epilogue (elu + 1 and the fragment-major stores).  Always 0 in libscream_hip.so.
#ifndef X3_ABLATE
#define X3_ABLATE 0
#endif
#include <type_traits>

#include "gemm_epilogue.h"
#include "split.h"

#ifdef X3_STAMPS  // tuning aid: s_memtime stamps of one output tile per block (tools/gemm_stamps.py)
__device__ long long gemm_stamps[256 * 8 * 160];
extern "C" int scream_gemm_stamps_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gemm_stamps), sizeof(long long) * 256 * 8 * 160);
}
#define STAMP(slot)                                                                                    \
    do {                                                                                               \
        if (stamp_on && lane == 0 && (slot) < 160)                                                      \
            gemm_stamps[((int)blockIdx.x * 8 + wave) * 160 + (slot)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int XBM = 256, XBN = 256, XBK = 32, XTHREADS = 512, XWAVES = 8;
constexpr int PLANE_BYTES = XBN * XBK * 2;     // 16 KiB; a k-tile stage is SP::NP of them
constexpr int SLAB_BYTES = XWAVES * 8 * 256 * 4;  // 64 KiB
constexpr int X_MAX_GRID = SCREAM_MAX_GRID;

// s_waitcnt vmcnt(N) lgkmcnt(0) + workgroup barrier: the N youngest vector-memory operations of this wave (the A
// loads of the k-tile after next) stay in flight across the barrier.
template <int N>
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
}

template <class SP, int EPI, bool AFRAG>
__global__ __launch_bounds__(XTHREADS, 2) void gemm_split_kernel(const float* __restrict__ A, int64_t lda,
                                                                const char* __restrict__ Wp, float* __restrict__ C,
                                                                int64_t ldc, int64_t M, int n_tiles, unsigned total_tiles,
                                                                int N, int K, float a_scale, float c_scale, EpiArgs ep) {
    typedef typename SP::vec V;
    constexpr int NP = SP::NP;
    constexpr int STAGE_BYTES = NP * PLANE_BYTES;  // 48 / 32 KiB
    constexpr int WPW = 2 * NP;                    // LDS-DMA pieces per wave and k-tile
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + SLAB_BYTES];  // 160 / 128 KiB
    float* slabs = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, half = lane >> 5;

    const unsigned q8 = total_tiles >> 3, rem = total_tiles & 7u;
    auto tile_of = [&](unsigned v) {  // XCD-aware numbering, as in gemm_f32.hip
        const unsigned xcd = v & 7u;
        return (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (v >> 3);
    };
    // DMA: 16 NP one-KiB pieces per k-tile (NP planes x 16 pieces of 16 rows x 64 B), 2 NP per wave.  The packed weight
    // image (scream_pack_w_split) is stored k-tile by k-tile exactly as it sits in LDS, chunk swizzle included, so
    // every piece is 1 KiB of CONTIGUOUS memory (eight full 128-byte lines, no over-fetch) and the per-lane part
    // of the address is just lane * 16 bytes.
    const char* w_lane = Wp + lane * 16;
    auto dma_w = [&](int n0, int stage, int kt) {
#pragma unroll
        for (int u = 0; u < WPW; ++u) {
            const int id = wave * WPW + u, plane = id >> 4, q = id & 15;
            const int64_t soff = (((int64_t)plane * (K / XBK) + kt) * N + n0 + q * 16) * XBK;  // scalar, in 16-bit elements
            __builtin_amdgcn_global_load_lds((gptr_t)(w_lane + 2 * soff),
                                             (lptr_t)(smem + stage * STAGE_BYTES + plane * PLANE_BYTES + q * 1024), 16, 0, 0);
        }
    };
    // A: rows past M (a trailing 128-row half tile) are clamped to the tile's first row and never stored
    auto a_ptr = [&](int64_t m0, bool ok) {
        if (AFRAG) return A + (ok ? m0 + wave * 32 : m0) * lda + lane * 4;  // the wave's 32-row group, piece a = 0 of this lane
        return A + (ok ? m0 + wave * 32 + r : m0) * lda + half * 4;
    };
    int boff[2];  // byte offset of this lane's 16-byte chunk of row r for step s (swizzled)
#pragma unroll
    for (int s = 0; s < 2; ++s) boff[s] = r * 64 + (((2 * half + s) ^ ((r >> 2) & 3)) << 4);

    const int KT = K / XBK;  // even (host check)
    unsigned v = blockIdx.x;
    unsigned tile = tile_of(v);
    int64_t m0 = (int64_t)(tile / n_tiles) * XBM;
    int n0 = (int)(tile % n_tiles) * XBN;
    bool rows_ok = m0 + wave * 32 < M;
    const float* ga = a_ptr(m0, rows_ok);

    // A registers: three sets, requested TWO k-tiles ahead (activations that only one output tile reads come from
    // HBM: one k-tile of lead did not cover that latency, profiles/r01_x3_ablation.txt)
    f32x4 a0[4], a1[4], a2[4];
    // Plain inline-asm loads, on purpose: hipcc's own vmcnt bookkeeping cannot count across the epilogue's stores and
    // the loop back edge and falls back to vmcnt(0) in front of the first use of these registers.  Every use below
    // sits behind an explicit counted s_waitcnt followed by an empty asm on the register (which pins the order).
    auto load_a = [&](f32x4 (&a)[4], int kt) {
        if (AFRAG) {  // segment kt of the group: [a][lane][4 floats], 1 KiB per wave instruction
            const float* p = ga + kt * (4 * 64 * 4);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[0]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(a[1]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(a[2]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(a[3]) : "v"(p));
        } else {      // pieces 2a + half of the lane's 128-byte row segment
            const float* p = ga + kt * XBK;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[0]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(a[1]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(a[2]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(a[3]) : "v"(p));
        }
    };
    // first requests of an output tile, D(0), A(0), A(1) (its first barrier drains the queue; the order is kept pinned)
    auto request_first = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        dma_w(n0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    request_first();

    V pa_s0[NP];  // split planes of the first 16-deep step of the k-tile about to be computed
    const bool late = wave >= 4;  // the second wave of each SIMD
#ifdef X3_STAMPS
    int tile_no = 0;
#endif
    for (;;) {
#ifdef X3_STAMPS
        const bool stamp_on = tile_no == 2 && blockIdx.x < 256;
        ++tile_no;
        STAMP(0);
#endif
        f32x16 acc[8];
#pragma unroll
        for (int tn = 0; tn < 8; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;

        // One k-tile = two 16-deep MFMA steps over the eight N-tiles.  The barrier at the top is followed directly by
        // MFMAs: the operand split of step 0 was done at the end of the previous k-tile (pa_s0), and the requests for
        // the next k-tile -- D(kt+1), then A(kt+2), in that order so that vmcnt(4) at the next barrier leaves the A
        // loads in flight -- are issued between the two steps instead of in the post-barrier bubble.
        // tail: 0 steady, 1 = only D(kt+1) left to request, 2 = nothing.  Compile-time on purpose: with a conditional
        // load inside the loop hipcc stops counting and falls back to s_waitcnt vmcnt(0).
        // tr (compile time): the tile is computed TRANSPOSED -- the weight fragments are the MFMA's first operand, so lane =
        // activation row, registers = output features: the fragment-major layout of a query tile, stored straight from the
        // accumulators (below).  Same products in the same order either way.
        auto groups = [&](auto tr, const char* wb, int s, const V (&pa)[NP], V (&fb)[2][NP]) __attribute__((always_inline)) {
#pragma unroll
            for (int tn = 0; tn < 8; ++tn) {
                const int cur = tn & 1, nxt = cur ^ 1;
                if (s * 8 + tn + 1 < 16 && !(X3_ABLATE & 16)) {  // fragments of the next (step, N-tile), one group ahead
                    const int s2 = (s * 8 + tn + 1) >> 3, t2 = (s * 8 + tn + 1) & 7;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        fb[nxt][p] = *reinterpret_cast<const V*>(wb + p * PLANE_BYTES + t2 * 32 * 64 + boff[s2]);
                }
                if (X3_ABLATE & 8) {
                    acc[tn][0] += (float)pa[0][0] + (float)pa[NP - 1][1] + (float)fb[cur][0][0] + (float)fb[cur][NP - 1][0];
                    continue;
                }
                if (decltype(tr)::value) SP::template products<false>(acc[tn], fb[cur], pa, acc[tn]);
                else SP::template products<true>(acc[tn], pa, fb[cur], acc[tn]);
                // first MFMA, then the prefetch reads (one per plane), then the other MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
                if (SP::NPROD > 1) __builtin_amdgcn_sched_group_barrier(0x008, SP::NPROD - 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto split_of = [&](const f32x4& lo, const f32x4& hi, V (&pa)[NP]) {
            if (X3_ABLATE & 32) {
                pa[0] = __builtin_bit_cast(V, lo);
                pa[1] = __builtin_bit_cast(V, hi);
                pa[NP - 1] = pa[0];
            } else if (SP::SCALED) {
                split8<SP>(lo * a_scale, hi * a_scale, pa);  // exact: a power of two
            } else {
                split8<SP>(lo, hi, pa);
            }
        };
        // (always_inline: with two copies of the k-loop in the kernel hipcc otherwise leaves `step` out of line in some
        // instantiations -- operands passed through memory, behind the back of the hand-counted waits; the build's checker refused it)
        auto step = [&](auto tr, auto tail, auto first, int kt, int stage, f32x4 (&ac)[4], f32x4 (&an1)[4], f32x4 (&an2)[4]) __attribute__((always_inline)) {
            constexpr int TAIL = decltype(tail)::value;
            constexpr bool FIRST = decltype(first)::value;
            // Everything but the four youngest operations (A(kt+1)) has landed: W k-tile kt is in LDS for every wave
            // and every wave is done reading the other stage.  The first k-tile of an output tile drains everything:
            // the previous epilogue's STORES are in the queue there, and a counted wait is only sound among loads --
            // stores complete out of order with respect to older loads (tools/gemm_soak.py caught vmcnt(4 + #stores)
            // returning with a load still in flight, once in ~10^8 tile boundaries).
            STAMP(4 + kt * 4 + 0);
            if (FIRST || TAIL == 2) ring_barrier<0>();
            else ring_barrier<4>();
            STAMP(4 + kt * 4 + 1);
            if (FIRST || late) {  // the younger waves of each SIMD split their first half here: they would only be
                asm volatile("" : "+v"(ac[0]));  // starved by the older wave's MFMAs for that long anyway, and the
                asm volatile("" : "+v"(ac[1]));  // barrier is released that much earlier
                split_of(ac[0], ac[1], pa_s0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const char* wb = smem + stage * STAGE_BYTES;
            V fb[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) fb[0][p] = *reinterpret_cast<const V*>(wb + p * PLANE_BYTES + boff[0]);
            groups(tr, wb, 0, pa_s0, fb);
            STAMP(4 + kt * 4 + 2);
            if (TAIL <= 1 && !(X3_ABLATE & 2)) dma_w(n0, stage ^ 1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);  // the counted waits rely on this issue order: D(kt+1), then A(kt+2)
            if (TAIL == 0 && !(X3_ABLATE & 4)) load_a(an2, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            V pa_s1[NP];
            asm volatile("" : "+v"(ac[2]));
            asm volatile("" : "+v"(ac[3]));
            split_of(ac[2], ac[3], pa_s1);
            __builtin_amdgcn_sched_barrier(0);
            groups(tr, wb, 1, pa_s1, fb);
            STAMP(4 + kt * 4 + 3);
            if (TAIL <= 1 && !late) {  // A(kt+1) is older than what was requested above: wait for it alone, split its first half
                if (TAIL == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (WPW + 4)); else __builtin_amdgcn_s_waitcnt(0x0F70 | WPW);  // vmcnt only
                asm volatile("" : "+v"(an1[0]));
                asm volatile("" : "+v"(an1[1]));
                split_of(an1[0], an1[1], pa_s0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr std::integral_constant<int, 0> steady{};
        constexpr std::integral_constant<int, 1> tail1{};
        constexpr std::integral_constant<int, 2> tail2{};
        constexpr std::integral_constant<bool, true> first{};
        constexpr std::integral_constant<bool, false> later{};
        auto k_loop = [&](auto tr) __attribute__((always_inline)) {
            if (KT > 2) {  // (KT - 2) % 3 == 0 (host check); KT is even, so stage = kt & 1
                step(tr, steady, first, 0, 0, a0, a1, a2);
                step(tr, steady, later, 1, 1, a1, a2, a0);
                step(tr, steady, later, 2, 0, a2, a0, a1);
                for (int kt = 3; kt < KT - 2; kt += 3) {
                    step(tr, steady, later, kt, kt & 1, a0, a1, a2);
                    step(tr, steady, later, kt + 1, (kt + 1) & 1, a1, a2, a0);
                    step(tr, steady, later, kt + 2, kt & 1, a2, a0, a1);
                }
                step(tr, tail1, later, KT - 2, 0, a0, a1, a2);
            } else {
                step(tr, tail1, first, 0, 0, a0, a1, a2);
            }
            step(tr, tail2, later, KT - 1, 1, a1, a2, a2);
        };
        // a query tile whose result goes out fragment-major (SCREAM_LAYOUT_C_FRAG: n_act == ldc == 256, so it is the tile at
        // column 0) is computed transposed; uniform per tile.  SplitH2 only: the bf16 x 3 instantiations stay the round-2 code
        // (a second copy of their k-loop costs them spill stores inside the tile loop, which tests/test_host_cpu.py forbids)
        constexpr bool CAN_TR = SP::SCALED && (EPI == SCREAM_EPI_ELU1 || EPI == SCREAM_EPI_QKV);
        const bool tr_cur = CAN_TR && ep.c_frag && n0 < ep.n_act;
        if constexpr (CAN_TR) {
            if (tr_cur) k_loop(std::integral_constant<bool, true>{});
            else k_loop(std::integral_constant<bool, false>{});
        } else {
            k_loop(std::integral_constant<bool, false>{});
        }

        STAMP(1);
        // next output tile: its first k-tile lands under the epilogue (the slabs have their own LDS region)
        const unsigned v_next = v + gridDim.x;
        const bool has_next = v_next < total_tiles;
        const int64_t m0_cur = m0;
        const int n0_cur = n0;
        const bool rows_cur = rows_ok;
        if (has_next) {
            tile = tile_of(v_next);
            m0 = (int64_t)(tile / n_tiles) * XBM;
            n0 = (int)(tile % n_tiles) * XBN;
            rows_ok = m0 + wave * 32 < M;
            ga = a_ptr(m0, rows_ok);
        }
        if (has_next) request_first();
        const bool skip_epi = (X3_ABLATE & 1) || ((X3_ABLATE & 128) && EPI == SCREAM_EPI_QKV && n0_cur >= ep.n_act) ||
                              ((X3_ABLATE & 256) && CAN_TR && tr_cur);
        if (skip_epi) {
            float keep = 0.f;
#pragma unroll
            for (int tn = 0; tn < 8; ++tn)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[tn][e];
            if (keep == 123.456f) C[0] = keep;
        } else {
            if (SP::SCALED) {  // back to true units: 2^-(a_exp + w_exp), exact
#pragma unroll
                for (int tn = 0; tn < 8; ++tn) acc[tn] *= c_scale;
            }
            if (CAN_TR && tr_cur) {
                // Q' = elu(q) + 1, fragment-major (SCREAM_ACT_FRAG), straight from the transposed accumulators: register
                // 4a + b of lane (r, half) in tile tn is feature 32 tn + 8 a + 4 half + b of row r, i.e. float
                // ((tn * 4 + a) * 64 + lane) * 4 + b of the wave's 32-row group -- one contiguous 1 KiB per store instruction,
                // no LDS slab, no transposition (the slab epilogue was a third of this kernel's time on query tiles,
                // profiles/r03_gemm_ablation_h2.txt)
                if (rows_cur) {
                    float* cg = C + (m0_cur + wave * 32) * 256 + lane * 4;
#pragma unroll
                    for (int tn = 0; tn < 8; ++tn)
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            f32x4 o;
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float x = acc[tn][4 * a + b];
                                o[b] = elu1(x);
                            }
                            *reinterpret_cast<f32x4*>(cg + (tn * 4 + a) * 256) = o;
                        }
                }
            } else {
                gemm_epilogue<EPI, XWAVES, SP::SCALED>(acc, slabs, wave, lane, tid, rows_cur, m0_cur, n0_cur, ep, C, ldc);
            }
        }
        STAMP(2);
        if (!has_next) break;
        v = v_next;
    }
}

// W [N][K] fp32 (x 2^w_exp for SplitH2) -> packed planes [NP][K/32][N][32] of 16-bit values.  Logical chunk c = 2 half + s of a row's 32-deep k-slice holds
// the eight contraction indices lane-half `half` owns in step s, k = 8 (2 s + (j >> 2)) + 4 half + (j & 3), and is stored
// at chunk c ^ ((n >> 2) & 3).  One thread per (n, k-tile, stored chunk).
template <class SP>
__global__ void pack_w_kernel(const float* __restrict__ W, int N, int K, float w_scale, typename SP::vec* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KT = K / XBK;
    if (t >= (int64_t)N * KT * 4) return;
    const int cs = (int)(t & 3), n = (int)((t >> 2) % N), kt = (int)((t >> 2) / N);
    const int c = cs ^ ((n >> 2) & 3);
    const int hf = c >> 1, st = c & 1;
    const float* src = W + (int64_t)n * K + kt * XBK + 16 * st + 4 * hf;  // j = 0..3 here, j = 4..7 eight floats on
    typename SP::vec p[SP::NP];
    if (SP::SCALED) split8<SP>(ld4(src) * w_scale, ld4(src + 8) * w_scale, p);
    else split8<SP>(ld4(src), ld4(src + 8), p);
    const int64_t plane = (int64_t)KT * N * XBK / 8;  // in 16-byte vectors
#pragma unroll
    for (int pl = 0; pl < SP::NP; ++pl) out[((int64_t)kt * N + n) * (XBK / 8) + cs + pl * plane] = p[pl];
}

struct SplitArgs {
    int32_t split, a_exp, w_exp;
};

bool split_args_ok(const SplitArgs& sa) {
    if (sa.split == SCREAM_SPLIT_BF3) return true;
    return (sa.split == SCREAM_SPLIT_H2 || sa.split == SCREAM_SPLIT_H1) && sa.a_exp >= -60 && sa.a_exp <= 60 && sa.w_exp >= -60 && sa.w_exp <= 60;
}

template <class SP, int EPI>
int launch_sp(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, int64_t M, int N, int K, const EpiArgs& ep,
              hipStream_t st, bool a_frag, const SplitArgs& sa) {
    const int n_tiles = N / XBN;
    const int64_t total = ((M + XBM - 1) / XBM) * n_tiles;
    if (total == 0) return 0;
    SCREAM_REQUIRE(total < (1ll << 31), SCREAM_EUNSUPPORTED);
    const unsigned grid = total < X_MAX_GRID ? (unsigned)total : (unsigned)X_MAX_GRID;
    const float a_scale = SP::SCALED ? exp2i(sa.a_exp) : 1.f, c_scale = SP::SCALED ? exp2i(-sa.a_exp - sa.w_exp) : 1.f;
    if (a_frag)
        gemm_split_kernel<SP, EPI, true><<<dim3(grid), dim3(XTHREADS), 0, st>>>(A, lda, reinterpret_cast<const char*>(Wp), C, ldc, M, n_tiles,
                                                                                (unsigned)total, N, K, a_scale, c_scale, ep);
    else
        gemm_split_kernel<SP, EPI, false><<<dim3(grid), dim3(XTHREADS), 0, st>>>(A, lda, reinterpret_cast<const char*>(Wp), C, ldc, M, n_tiles,
                                                                                 (unsigned)total, N, K, a_scale, c_scale, ep);
    SCREAM_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_split(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, int64_t M, int N, int K, const EpiArgs& ep,
                 hipStream_t st, bool a_frag, const SplitArgs& sa) {
    if (sa.split == SCREAM_SPLIT_H2) return launch_sp<SplitH2, EPI>(A, lda, Wp, C, ldc, M, N, K, ep, st, a_frag, sa);
    if (sa.split == SCREAM_SPLIT_H1) return launch_sp<SplitH1, EPI>(A, lda, Wp, C, ldc, M, N, K, ep, st, a_frag, sa);
    return launch_sp<SplitBf3, EPI>(A, lda, Wp, C, ldc, M, N, K, ep, st, a_frag, sa);
}

}  // namespace

extern "C" int scream_pack_w_split(const float* W, int32_t N, int32_t K, int32_t split, int32_t w_exp, void* packed, void* stream) {
    SCREAM_REQUIRE(W && packed, SCREAM_EINVAL);
    SCREAM_REQUIRE(split_args_ok(SplitArgs{split, 0, w_exp}), SCREAM_EINVAL);
    SCREAM_REQUIRE(N > 0 && N % 4 == 0 && K > 0 && K % XBK == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, SCREAM_EINVAL);
    const int64_t threads = (int64_t)N * (K / XBK) * 4;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (split == SCREAM_SPLIT_H2)
        pack_w_kernel<SplitH2><<<grid, block, 0, as_stream(stream)>>>(W, N, K, exp2i(w_exp), reinterpret_cast<f16x8*>(packed));
    else if (split == SCREAM_SPLIT_H1)
        pack_w_kernel<SplitH1><<<grid, block, 0, as_stream(stream)>>>(W, N, K, exp2i(w_exp), reinterpret_cast<f16x8*>(packed));
    else
        pack_w_kernel<SplitBf3><<<grid, block, 0, as_stream(stream)>>>(W, N, K, 1.f, reinterpret_cast<bf16x8*>(packed));
    SCREAM_LAUNCH_CHECK();
    return 0;
}

// layout bit 0 (SCREAM_LAYOUT_A_FRAG): A is fragment-major (K == 256 == lda); bit 1 (SCREAM_LAYOUT_C_FRAG): the activated
// tile of an ELU1 / QKV epilogue (the 256 query columns) is written fragment-major (ldc == 256 == n_act)
static int check_layout(int32_t layout, int64_t lda, int64_t ldc, int32_t K, int32_t epilogue, int32_t n_act) {
    SCREAM_REQUIRE((layout & ~3) == 0, SCREAM_EINVAL);
    if (layout & SCREAM_LAYOUT_A_FRAG) SCREAM_REQUIRE(lda == SCREAM_D_MODEL && K == SCREAM_D_MODEL, SCREAM_EUNSUPPORTED);
    if (layout & SCREAM_LAYOUT_C_FRAG)
        SCREAM_REQUIRE(ldc == SCREAM_D_MODEL && n_act == SCREAM_D_MODEL && (epilogue == SCREAM_EPI_ELU1 || epilogue == SCREAM_EPI_QKV), SCREAM_EUNSUPPORTED);
    return 0;
}

extern "C" int scream_gemm_split_f32(const float* A, int64_t lda, const void* W_packed, float* C, int64_t ldc, int64_t M,
                                     int32_t N, int32_t K, int32_t epilogue, int32_t n_act, const float* bias,
                                     const float* residual, int64_t ldr, const float* gamma, const float* beta,
                                     int32_t layout, int32_t split, int32_t a_exp, int32_t w_exp, void* stream) {
    SCREAM_REQUIRE(A && W_packed && C, SCREAM_EINVAL);
    const SplitArgs sa{split, a_exp, w_exp};
    SCREAM_REQUIRE(split_args_ok(sa), SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0 && N > 0 && N % XBN == 0 && K >= 64 && K % 64 == 0 && (K / 32 - 2) % 3 == 0, SCREAM_EUNSUPPORTED);  // K = 64 + 192 j: an EVEN number of k-tiles (stage = kt & 1), in groups of three after the first two
    SCREAM_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W_packed) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(C) & 15) == 0, SCREAM_EINVAL);
    if (int rc = check_layout(layout, lda, ldc, K, epilogue, n_act)) return rc;
    const bool af = layout & SCREAM_LAYOUT_A_FRAG;
    EpiArgs ep{n_act, bias, residual, ldr, gamma, beta, nullptr, nullptr, nullptr, nullptr, 0, (layout & SCREAM_LAYOUT_C_FRAG) ? 1 : 0};
    hipStream_t st = as_stream(stream);
    switch (epilogue) {
        case SCREAM_EPI_NONE:
            return launch_split<SCREAM_EPI_NONE>(A, lda, W_packed, C, ldc, M, N, K, ep, st, af, sa);
        case SCREAM_EPI_ELU1:
            SCREAM_REQUIRE(n_act >= 0 && n_act % XBN == 0, SCREAM_EUNSUPPORTED);
            return launch_split<SCREAM_EPI_ELU1>(A, lda, W_packed, C, ldc, M, N, K, ep, st, af, sa);
        case SCREAM_EPI_RELU:
            return launch_split<SCREAM_EPI_RELU>(A, lda, W_packed, C, ldc, M, N, K, ep, st, af, sa);
        case SCREAM_EPI_BIAS_RELU:
            SCREAM_REQUIRE(bias, SCREAM_EINVAL);
            return launch_split<SCREAM_EPI_BIAS_RELU>(A, lda, W_packed, C, ldc, M, N, K, ep, st, af, sa);
        case SCREAM_EPI_RES_LN:
            SCREAM_REQUIRE(N == XBN, SCREAM_EUNSUPPORTED);
            SCREAM_REQUIRE(residual && gamma && beta && ldr >= N && ldr % 4 == 0, SCREAM_EINVAL);
            SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(residual) & 15) == 0, SCREAM_EINVAL);
            return launch_split<SCREAM_EPI_RES_LN>(A, lda, W_packed, C, ldc, M, N, K, ep, st, af, sa);
        default:
            return SCREAM_EINVAL;
    }
}

extern "C" int scream_gemm_qkv_split_f32(const float* A, int64_t lda, const void* W_packed, float* Q, int64_t ldq, int64_t M,
                                         int32_t N, int32_t K, int32_t n_q, const int32_t* tile_cloud,
                                         const int32_t* cloud_row0, const int32_t* cloud_len, int64_t row_base,
                                         float* kv_partial, int32_t layout, int32_t split, int32_t a_exp, int32_t w_exp,
                                         int32_t k_exp, int32_t v_exp, void* stream) {
    SCREAM_REQUIRE(A && W_packed && kv_partial && tile_cloud && cloud_row0 && cloud_len, SCREAM_EINVAL);
    const SplitArgs sa{split, a_exp, w_exp};
    SCREAM_REQUIRE(split_args_ok(sa), SCREAM_EINVAL);
    SCREAM_REQUIRE(split == SCREAM_SPLIT_BF3 || (k_exp >= -40 && k_exp <= 40 && v_exp >= -40 && v_exp <= 40), SCREAM_EINVAL);
    SCREAM_REQUIRE(M >= 0 && M % SCREAM_ROW_TILE == 0 && N > 0 && N % XBN == 0 && K >= 64 && K % 64 == 0 && (K / 32 - 2) % 3 == 0, SCREAM_EUNSUPPORTED);  // K = 64 + 192 j
    // N = n_q + 512 L: L key/value tile pairs; L > 1 (several layers' key/value projections of the same rows) only without queries
    SCREAM_REQUIRE((n_q == 0 || n_q == XBN) && N > n_q && (N - n_q) % (2 * XBN) == 0 && (n_q == 0 || N == n_q + 2 * XBN) && row_base >= 0 &&
                       row_base % SCREAM_ROW_TILE == 0, SCREAM_EUNSUPPORTED);
    SCREAM_REQUIRE(n_q == 0 || (Q && ldq >= n_q && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0), SCREAM_EINVAL);
    SCREAM_REQUIRE(lda >= K && lda % 4 == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W_packed) & 15) == 0, SCREAM_EINVAL);
    SCREAM_REQUIRE(!(layout & SCREAM_LAYOUT_C_FRAG) || n_q == XBN, SCREAM_EUNSUPPORTED);
    if (int rc = check_layout(layout, lda, n_q ? ldq : SCREAM_D_MODEL, K, SCREAM_EPI_QKV, n_q ? n_q : SCREAM_D_MODEL)) return rc;
    EpiArgs ep{n_q, nullptr, nullptr, 0, nullptr, nullptr, kv_partial, tile_cloud, cloud_row0, cloud_len, row_base,
               (layout & SCREAM_LAYOUT_C_FRAG) ? 1 : 0, (M / SCREAM_ROW_TILE) * SCREAM_NHEAD * (int64_t)KV_ELEMS,
               exp2i(k_exp), exp2i(v_exp), exp2i(-k_exp - v_exp)};
    return launch_split<SCREAM_EPI_QKV>(A, lda, W_packed, Q, ldq, M, N, K, ep, as_stream(stream), layout & SCREAM_LAYOUT_A_FRAG, sa);
}
