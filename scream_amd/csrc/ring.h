// Shared device helpers of the one-wave-per-SIMD kernel that streams its weights through a ring of LDS stages
// (tail_split.hip: the row-local tail of a block), templated over the operand split (split.h): the ring's counted barrier,
// LDS-DMA pieces with shared address registers, fragment reads, the MFMA group of one 16-deep step with its scheduling
// pattern, and the inline-asm register loads (scalar base + 32-bit lane offset) with their hand-counted waits.
#pragma once
#ifndef T_ABLATE
#define T_ABLATE 0  // tuning aid (tools/tail_ablate.py); always 0 in libscream_hip.so
#endif
#ifndef T_PF
#define T_PF 3  // register sets of weight fragments: fragments are read T_PF - 1 MFMA groups ahead
#endif
#include "split.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TT = 256;               // threads
constexpr int T_SLOTS = 3;
constexpr int T_MAX_GRID = SCREAM_MAX_GRID;

// one ring stage: 16 fragments of 1 KiB per operand plane (SplitBf3: 48 KiB, SplitH2: 32 KiB, SplitH1: 16 KiB); a wave issues a quarter of
// its LDS-DMA pieces
template <class SP> constexpr int stage_bytes() { return SP::NP * 16 * 1024; }
template <class SP> constexpr int wave_pieces() { return SP::NP * 4; }

// s_waitcnt vmcnt(N) lgkmcnt(0) + workgroup barrier (see gemm_split.hip): the N youngest vector-memory operations of
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
template <class V>
__device__ __forceinline__ V ld_frag(const char* p) {
    if (T_ABLATE & 4) {
        V v;
        asm volatile("" : "=v"(v));  // opaque, undefined: keeps the consumers alive without the LDS read
        return v;
    }
    return *reinterpret_cast<const V*>(p);
}

// acc += W . act for one 16-deep step with both operands split (SP::NPROD exact products, smallest first, fp32 accumulate).
// w = A operand (weights), a = B operand (activations).
// NV > 0: the stage carries VALU work of another computation (a "ride"); the scheduler is told to place up to NV of
// those instructions behind every MFMA instead of leaving them in one run between two groups -- an MFMA occupies the
// matrix pipe for 32 cycles, a VALU instruction issues in 4, so up to seven ride for free behind each.
// NV < 0: no scheduling directives at all (the short products that ride inside another stage's groups).
// zero: the accumulator tile starts here (first product takes the constant 0 as its C operand: no zeroing moves).
template <class SP, int NV = 0>
__device__ __forceinline__ void mfma_group(f32x16& acc, const typename SP::vec (&w)[SP::NP], const typename SP::vec (&a)[SP::NP],
                                           bool zero = false) {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    if (T_ABLATE & 2) {
        if (zero) acc = z;
        acc[0] += (float)w[0][0] + (float)w[SP::NP - 1][1] + (float)a[0][0] + (float)a[SP::NP - 1][1];
        return;
    }
    SP::products(acc, w, a, zero ? z : acc);
    if (NV < 0) return;
    // first MFMA, then the prefetch reads of the next fragment group (one per plane), then the other MFMAs
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, SP::NP, 0);
    if (NV == 0) {
        if (SP::NPROD > 1) __builtin_amdgcn_sched_group_barrier(0x008, SP::NPROD - 1, 0);
    } else {
#pragma unroll
        for (int i = 0; i < SP::NPROD - 1; ++i) {
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
// of one of these loads or stores was read stale, and the access went to a wild address (round 2, the ring-design projection
// kernel's query-only variant: memory fault).  tools/asm_inflight_check.py verifies the wait states on the generated code of
// every kernel that uses these helpers, at every build (scream_amd/build.py refuses to link otherwise; the CPU suite asserts).
// four loads STEP bytes apart: the pieces a = 0 .. 3 of a fragment-major segment (1 KiB) or of a lane's 128-byte segment (32 B)
// NT: the non-temporal hint (rows that stream through once: the L2 then keeps the weight image instead of them)
template <int STEP, bool NT = false>
__device__ __forceinline__ void ld_asm4(f32x4 (&d)[4], const void* sbase, unsigned voff) {
    static_assert(STEP == 1024 || STEP == 32, "");
    if (STEP == 1024 && NT) {
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d[0]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(d[1]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 nt" : "=v"(d[2]) : "v"(voff), "s"(sbase));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 nt" : "=v"(d[3]) : "v"(voff), "s"(sbase));
    } else if (STEP == 1024) {
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
}
__device__ __forceinline__ void ld_asm2k(f32x4& d0, f32x4& d1, const void* sbase, unsigned voff) {  // two loads 1 KiB apart
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d0) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(d1) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// s_waitcnt vmcnt(N) alone, as an asm statement: ordered against the other asm statements (slab reads, register loads)
#define VM_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
template <int N>
__device__ __forceinline__ void vm_wait() {  // N = the weight pieces that may stay in flight ("no DMA" tuning build: none exist, drain)
    static_assert(N == 0 || N == 4 || N == 8 || N == 12, "");
    if (N == 0 || (T_ABLATE & 1)) VM_WAIT(0);
    else if (N == 4) VM_WAIT(4);
    else if (N == 8) VM_WAIT(8);
    else VM_WAIT(12);
}

}  // namespace
