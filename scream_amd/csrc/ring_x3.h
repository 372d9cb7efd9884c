// Shared device helpers of the one-wave-per-SIMD kernels that stream their weights through a ring of LDS stages
// (tail_x3.hip: the row-local tail of a block; proj_x3.hip: the q/k/v projections): the 3-way bf16 operand split, the ring's
// counted barrier, LDS-DMA pieces with shared address registers, fragment reads, the six-product MFMA group with its
// scheduling pattern, and the inline-asm register loads (scalar base + 32-bit lane offset) with their hand-counted waits.
#pragma once
#ifndef T_ABLATE
#define T_ABLATE 0  // tuning aid (tools/tail_ablate.py); always 0 in libscream_hip.so
#endif
#ifndef T_PF
#define T_PF 3  // register sets of weight fragments: fragments are read T_PF - 1 MFMA groups ahead
#endif
#include "common.h"

// The six products of one split step, smallest terms first, the exact leading product last.  (Orders that keep one operand
// in place across consecutive instructions -- runs of the same weight plane, or of the same activation plane -- draw the
// same 2.37 J per 333 k-row tail launch: the energy is not in the operand switching.)
#define RING_MFMA(acc, A, B, C) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#define RING_SIX_PRODUCTS(acc, w, a, c0) \
    RING_MFMA(acc, w[0], a[2], c0); RING_MFMA(acc, w[1], a[1], acc); RING_MFMA(acc, w[2], a[0], acc); \
    RING_MFMA(acc, w[0], a[1], acc); RING_MFMA(acc, w[1], a[0], acc); RING_MFMA(acc, w[0], a[0], acc)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TT = 256;               // threads
constexpr int T_STAGE = 48 * 1024;    // one ring stage: 48 fragments of 1 KiB
constexpr int T_SLOTS = 3;
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
    constexpr int W = (T_ABLATE & 1) ? 0 : N;  // the "no DMA" tuning build has no pieces to leave in flight: it must drain,
    __builtin_amdgcn_s_waitcnt(0x0070 | (W & 15) | ((W >> 4) << 14));  // or row operands would still be pending at their use
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void lds_only_barrier() {  // lgkmcnt(0) + workgroup barrier, vector-memory queue untouched
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

// Layout convention of every transposed tile in this file: accumulator tile blk, register i of lane (r, half) holds
// feature 32 blk + mfma32_row(i, half) = 32 blk + 8 (i >> 2) + 4 half + (i & 3) of activation row r.  When such a tile
// is the B operand of the next GEMM, registers 8 s2 .. 8 s2 + 7 are 16-deep step s2, so lane-half `half` supplies, as
// element j of step s2, contraction index chunk_k(s2, half, j) of its 32-wide chunk.
__host__ __device__ __forceinline__ int chunk_k(int s2, int half, int j) { return 8 * (2 * s2 + (j >> 2)) + 4 * half + (j & 3); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// One 1 KiB LDS-DMA piece, number k (0 .. 3) of a group of four that share their address registers: the instruction's
// immediate offset moves the global source AND the LDS destination, so four pieces cost one address computation and
// one M0 write (k folds to a constant after unrolling; the builtin wants a literal).
__device__ __forceinline__ void dma_1k(const char* src_lane, char* dst, int k) {
    switch (k) {
        case 0: __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst, 16, 3072, 0); break;
    }
}

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
// NV > 0: the stage carries VALU work of another computation (a "ride"); the scheduler is told to place up to NV of
// those instructions behind every MFMA instead of leaving them in one run between two groups -- an MFMA occupies the
// matrix pipe for 32 cycles, a VALU instruction issues in 4, so up to seven ride for free behind each.
// zero: the accumulator tile starts here (first product takes the constant 0 as its C operand: no zeroing moves).
template <int NV = 0>
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3], bool zero = false) {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    if (T_ABLATE & 2) {
        if (zero) acc = z;
        acc[0] += (float)w[0][0] + (float)w[1][1] + (float)w[2][2] + (float)a[0][0] + (float)a[1][1] + (float)a[2][2];
        return;
    }
    RING_SIX_PRODUCTS(acc, w, a, zero ? z : acc);
    // first MFMA, then the three prefetch reads of the next fragment group, then the other five MFMAs
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    if (NV == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}


// Register loads address memory as (wave-uniform base in SGPRs) + (32-bit per-lane offset in ONE VGPR) + immediate: the
// bases are scalar arithmetic, and no load needs a 64-bit VGPR address of its own (dozens of those, precomputed per tile
// by hipcc, were what spilled at the tile boundaries).
// HAZARD: gfx950 needs 5 wait states between a VALU instruction that writes an SGPR (v_readlane / v_readfirstlane -- which
// is how hipcc restores a spilled scalar) and a vector-memory instruction that reads it as its address.  hipcc inserts them
// in front of its own memory instructions but does not look inside an asm statement: a scalar base restored right in front
// of one of these loads or stores was read stale, and the access went to a wild address (proj_x3.hip, query-only variant:
// memory fault).  tools/asm_inflight_check.py verifies the wait states on the generated code of every kernel that uses
// these helpers, at every build.
// four loads STEP bytes apart: the pieces a = 0 .. 3 of a fragment-major segment (1 KiB) or of a lane's 128-byte segment (32 B)
// RING_ASM_PADDED (proj_x3.hip defines it): every load group is ONE asm statement that starts with the five wait states, so
// nothing hipcc emits can break the rule.  tail_x3.hip keeps the unpadded form -- four separate statements, which allocate
// better (280 k vs 286 k cycles per tile) -- and relies on the build: scream_amd/build.py and the CPU suite run
// tools/asm_inflight_check.py on the generated code and fail on any scalar base written fewer than five wait states
// before its use.
template <int STEP>
__device__ __forceinline__ void ld_asm4(f32x4 (&d)[4], const void* sbase, unsigned voff) {
    static_assert(STEP == 1024 || STEP == 32, "");
#ifdef RING_ASM_PADDED
    if (STEP == 1024) {
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %4, %5\n\t"
                     "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                     "global_load_dwordx4 %3, %4, %5 offset:3072"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(voff), "s"(sbase));
    } else {
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %4, %5\n\t"
                     "global_load_dwordx4 %1, %4, %5 offset:32\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:64\n\t"
                     "global_load_dwordx4 %3, %4, %5 offset:96"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(voff), "s"(sbase));
    }
#else
    if (STEP == 1024) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d[0]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(d[1]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(d[2]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(d[3]) : "v"(voff), "s"(sbase));
    } else {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d[0]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(d[1]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(d[2]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=v"(d[3]) : "v"(voff), "s"(sbase));
    }
#endif
}
__device__ __forceinline__ void ld_asm2k(f32x4& d0, f32x4& d1, const void* sbase, unsigned voff) {  // two loads 1 KiB apart
#ifdef RING_ASM_PADDED
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx4 %0, %2, %3\n\t"
                 "global_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(d0), "=&v"(d1) : "v"(voff), "s"(sbase));
#else
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d0) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(d1) : "v"(voff), "s"(sbase));
#endif
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }


// s_waitcnt vmcnt(N) alone, as an asm statement: ordered against the other asm statements (slab reads, register loads)
#define VM_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// acc += W . act, plain (no scheduling directives): for the short products that ride inside another stage's groups
__device__ __forceinline__ void mfma6_free(f32x16& acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3], bool zero = false) {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    if (T_ABLATE & 2) {
        if (zero) acc = z;
        acc[0] += (float)w[0][0] + (float)a[0][0];
        return;
    }
    RING_SIX_PRODUCTS(acc, w, a, zero ? z : acc);
}

}  // namespace
