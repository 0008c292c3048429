// Split-fp16 ("x3") GEMM main loop, WIDE tiles: 256 x 256 (or 128 x 256) outputs per 512-thread workgroup.
//
// Same arithmetic as gemm_core_x3.h (three v_mfma_f32_32x32x16_f16 per product, the same k order and the same order of the
// three terms per accumulator: results are bit-identical whichever tile ran).  What changes is the traffic per MFMA.  The
// 128 x 128 x3 kernels sit at 26-35 % MfmaUtil with NO unit saturated (PMC: VALU ~50 % of a SIMD, LDS ~40 %, waves 45 %
// issue-stalled, 11-20 % parked on waitcnt/barrier) and go no faster when the splitting arithmetic is removed
// (pram_linear_x3p_f32: +5 %): per 24-MFMA chunk a workgroup moves 32 KB global -> LDS and reads 64 KB back, i.e.
// ~42 B/clk/CU of L2 traffic and ~0.9 LDS-pipe cycles per matrix-pipe cycle at full MFMA rate — both pipes would have to run
// near their peaks at once.  At 5x the f32-MFMA rate the tile has to grow instead:
//   workgroup tile 256 x 256 : global -> LDS bytes and LDS writes per flop halve (64 KB per 12.6 MFLOP vs 32 KB per 3.1);
//   wave tile      128 x 64  : 12 ds_read_b128 per 24 MFMAs instead of 8 per 12.
// 8 waves (WM x WN = 2 x 4, or 2 x 4 waves of 64 x 64 for the 128-row variant), 128 KB of LDS (double buffered), one workgroup
// per CU = two waves per SIMD.  A arrives either as fp32 (split while staged, like gemm_core_x3.h) or as pre-split planes.
#pragma once
#include "common.h"
#include "gemm_core_x3.h"
#include <type_traits>

namespace gemmx3w {

using gemmx3::half4;
using gemmx3::half8;
using gemmx3::split4;
using gemmx3::swz;

constexpr int BK = 32;
#ifndef PRAM_GEMM_SHADOW
#define PRAM_GEMM_SHADOW 1      // 1: the 256-row tile loop stages the next chunk inside its MFMA stream (mainloop, "Staging in the shadow")
#endif
constexpr bool SHADOW = PRAM_GEMM_SHADOW != 0;

// PRAM_GEMM_ABLATE=4 (profiling only): wave 0 of every workgroup adds the shader-clock cycles it spent per main-loop phase
// [0] issue + MFMA k-steps  [1] waiting for the chunk's loads (vmcnt)  [2] commit (split + ds_write)  [3] barrier
// [4] whole main loop  [5] workgroups  [6] epilogue (added by the kernel);  [8 + 8 w + i]: workgroup 0's wave w, chunk 3, time stamp i
// (0 loop top, 1 loads issued, 2 first k-step issued, 3 second k-step issued, 4 loads landed, 5 commit done, 6 past the
// barrier).  Read with pram_debug_gemm_phases().
static __device__ unsigned long long prof[72];

template <int MI, int WM, int WN>
struct Cfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int BM = WM * 32 * MI;
    static constexpr int BN = WN * 64;
    static constexpr int PA = BM * 8 / NT;     // float4 (fp32) staging loads per thread for A
    static constexpr int QA = BM * 4 / NT;     // 16-byte loads per thread and plane for pre-split A
    static constexpr int QB = BN * 4 / NT;     // ... for B
    static constexpr int RA = NT / 8;          // rows per pass of the fp32 A loader
    static constexpr int RQ = NT / 4;          // rows per pass of the plane loaders
};

template <int MI, int WM, int WN>
struct alignas(16) Smem {
    using C = Cfg<MI, WM, WN>;
    _Float16 ah[2][C::BM * BK];
    _Float16 al[2][C::BM * BK];
    _Float16 bh[2][C::BN * BK];
    _Float16 bl[2][C::BN * BK];
};  // <4,2,4>: 128 KiB; <2,2,4>: 96 KiB

// LDS-DMA staging of a pre-split operand tile (rows x 32 halves per plane): one global_load_lds_dwordx4 moves 64 x 16 B from
// per-lane global addresses to 1 KiB of LDS at (wave-uniform base) + 16 * lane, i.e. 16 consecutive 64-B tile rows.  The LDS
// image is the swizzled one the fragment reads expect, so the swizzle is applied on the SOURCE side: lane (row_local = l / 4,
// physical slot = l % 4) fetches logical slot (l % 4) ^ ((row >> 2) & 3).  No VGPR staging, no ds_write, no VALU.
//   rowptr(row, plane) -> const _Float16* of tile row `row` (clamped to a legal row by the caller) at the chunk's k offset
template <int ROWS, int NWAVES, class RowPtr>
__device__ __forceinline__ void dma_tile(_Float16* lds_hi, _Float16* lds_lo, RowPtr& rowptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PIECES = ROWS / 16;                 // 1-KiB pieces per plane
#pragma unroll
    for (int i = 0; i < (2 * PIECES + NWAVES - 1) / NWAVES; ++i) {
        const int piece = wave + NWAVES * i;          // wave-uniform
        if (piece < 2 * PIECES) {
            const int plane = piece / PIECES, pc = piece % PIECES;
            const int row = pc * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3);
            const _Float16* src = rowptr(row, plane) + slot * 8;
            _Float16* dst = (plane ? lds_lo : lds_hi) + pc * 16 * BK;      // wave-uniform
            __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
        }
    }
}

// APLANES = false: ALoad(p, kt) -> raw float4 A[row = tid/8 + RA p][kt*32 + (tid%8)*4 ..+3] (fp32, split here)
// APLANES = true : ALoad(p, kt, plane) -> raw uint4 plane[row = tid/4 + RQ p][kt*32 + (tid%4)*8 ..+7]
// BLoad(p, kt, plane) -> raw uint4 W_plane[col = tid/4 + RQ p][kt*32 + (tid%4)*8 ..+7];  *Ok: predicates;  Adv as gemm_core_x3.h
// ABL (profiling only, PRAM_GEMM_ABLATE): bit 0 = no staging after the first chunk (LDS content stale), bit 1 = fragments read
// once per chunk instead of per k-step, 16 = no A staging (loads, split, ds_write), 32 = no B DMA.  Results are garbage; the
// remaining work keeps its shape.
// DMA: 0 = register staging for both operands; 1 = B (weights) by LDS-DMA (bptr(row, plane, kt) -> row pointer at the chunk's k);
//      2 = A planes by LDS-DMA as well (aptr likewise; APLANES only).  Rows are clamped by the pointer functors; an out-of-range
//      row is a duplicate whose outputs the epilogue never stores.
template <int MI, int WM, int WN, bool APLANES, int ABL, int DMA, class Adv, class ALoad, class AOk, class BLoad, class BOk, class APtr, class BPtr, class AXf>
__device__ __forceinline__ void mainloop(Smem<MI, WM, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, APtr& aptr, BPtr& bptr,
                                         int nk, float a_scale, f32x16 (&acc)[MI][2], float& amax, AXf& axf);

template <int MI, int WM, int WN, bool APLANES, int ABL, int DMA, class Adv, class ALoad, class AOk, class BLoad, class BOk, class APtr, class BPtr>
__device__ __forceinline__ void mainloop(Smem<MI, WM, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, APtr& aptr, BPtr& bptr,
                                         int nk, float a_scale, f32x16 (&acc)[MI][2], float& amax) {
    gemmx3::NoXform none;
    mainloop<MI, WM, WN, APLANES, ABL, DMA>(s, adv, la, oka, lb, okb, aptr, bptr, nk, a_scale, acc, amax, none);
}

// AXf: see gemm_core_x3.h (fp32 A only)
template <int MI, int WM, int WN, bool APLANES, int ABL, int DMA, class Adv, class ALoad, class AOk, class BLoad, class BOk, class APtr, class BPtr, class AXf>
__device__ __forceinline__ void mainloop(Smem<MI, WM, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, APtr& aptr, BPtr& bptr,
                                         int nk, float a_scale, f32x16 (&acc)[MI][2], float& amax, AXf& axf) {
    static_assert(DMA < 2 || APLANES, "A can only travel by DMA when it is already split");
    using C = Cfg<MI, WM, WN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int arow = tid >> 3, akq = tid & 7;
    const int qrow = tid >> 2, qsl = tid & 3;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    constexpr int NA = APLANES ? C::QA : C::PA;
    struct Regs {
        float4 a[APLANES ? 1 : C::PA];
        uint4 ah[APLANES ? C::QA : 1], al[APLANES ? C::QA : 1];
        uint4 bh[C::QB], bl[C::QB];
        unsigned ok;
        int kt;
    };
    auto issue = [&](int kt, Regs& g) {
        g.ok = 0u;
        g.kt = kt;
        if constexpr (DMA < 2 && !(ABL & 16)) {
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                if constexpr (APLANES) { g.ah[p] = la(p, kt, 0); g.al[p] = la(p, kt, 1); }
                else g.a[p] = la(p, kt);
                g.ok |= (oka(p, kt) ? 1u : 0u) << p;
            }
        }
        if constexpr (DMA < 1) {
#pragma unroll
            for (int p = 0; p < C::QB; ++p) { g.bh[p] = lb(p, kt, 0); g.bl[p] = lb(p, kt, 1); g.ok |= (okb(p, kt) ? 1u : 0u) << (8 + p); }
        }
    };
    // the DMA part of a chunk's staging: issued with the register loads, lands in LDS buffer `buf` on its own
    auto dma = [&](int buf, int kt) {
        if constexpr (DMA >= 1 && !(ABL & 32)) {
            auto bp = [&](int row, int plane) { return bptr(row, plane, kt); };
            dma_tile<C::BN, C::NT / 64>(s.bh[buf], s.bl[buf], bp);
        }
        if constexpr (DMA >= 2) {
            auto ap = [&](int row, int plane) { return aptr(row, plane, kt); };
            dma_tile<C::BM, C::NT / 64>(s.ah[buf], s.al[buf], ap);
        }
    };
    auto dma_wait = [&]() {
        if constexpr (DMA >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto commit = [&](int buf, const Regs& g) {
#pragma unroll
        for (int p = 0; p < ((DMA < 2 && !(ABL & 16)) ? NA : 0); ++p) {
            if constexpr (APLANES) {
                const int row = qrow + C::RQ * p;
                uint4 vh = g.ah[p], vl = g.al[p];
                if (!((g.ok >> p) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
                const int off = row * BK + swz(qsl, row) * 8;
                *reinterpret_cast<uint4*>(&s.ah[buf][off]) = vh;
                *reinterpret_cast<uint4*>(&s.al[buf][off]) = vl;
            } else {
                const int row = arow + C::RA * p;
                float4 v = g.a[p];
                if (!((g.ok >> p) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                else axf(v, p, g.kt);
                half4 hi, lo;
                split4(v, a_scale, hi, lo, amax);      // amax: range guard (common.h), reported by the caller
                const int off = row * BK + swz(akq >> 1, row) * 8 + (akq & 1) * 4;
                *reinterpret_cast<half4*>(&s.ah[buf][off]) = hi;
                *reinterpret_cast<half4*>(&s.al[buf][off]) = lo;
            }
        }
#pragma unroll
        for (int p = 0; p < (DMA < 1 ? C::QB : 0); ++p) {
            const int row = qrow + C::RQ * p;
            uint4 vh = g.bh[p], vl = g.bl[p];
            if (!((g.ok >> (8 + p)) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
            const int off = row * BK + swz(qsl, row) * 8;
            *reinterpret_cast<uint4*>(&s.bh[buf][off]) = vh;
            *reinterpret_cast<uint4*>(&s.bl[buf][off]) = vl;
        }
    };
    // one 16-deep k-step: B fragments once, A fragments per 32-row block one block ahead of their MFMAs
    auto kstep = [&](int cur, int ks) {
        const int arow0 = (wm * 32 * MI + r) * BK, brow0 = (wn * 64 + r) * BK;
        const int slot = swz(2 * ks + h, r) * 8;      // rows differ from r by multiples of 32: same swizzle
        half8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bh[ni] = *reinterpret_cast<const half8*>(&s.bh[cur][brow0 + ni * 32 * BK + slot]);
            bl[ni] = *reinterpret_cast<const half8*>(&s.bl[cur][brow0 + ni * 32 * BK + slot]);
        }
        ah[0] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + slot]);
        al[0] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + slot]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (mi + 1 < MI) {
                ah[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + (mi + 1) * 32 * BK + slot]);
                al[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + (mi + 1) * 32 * BK + slot]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // small terms first, the dominant hi.hi last (the order of gemm_core_x3.h: lo.hi, hi.lo, hi.hi per accumulator)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    if constexpr (MI <= 2) {
        // 128-row tiles have the registers for two chunks of loads in flight (chunk kt + 2 requested before chunk kt is
        // multiplied): with one workgroup per CU all eight waves otherwise wait on the same load round trip at the same time
        Regs g0, g1;
        adv(0);
        issue(0, g0);
        dma(0, 0);
        if (nk > 1) { adv(1); issue(1, g1); }
        commit(0, g0);
        dma_wait();
        __syncthreads();
        auto step = [&](int kt, Regs& fresh, Regs& next) {
            if (kt + 1 < nk && !(ABL & 1)) dma((kt + 1) & 1, kt + 1);      // the other LDS buffer is free since the last barrier
            if (kt + 2 < nk && !(ABL & 1)) { adv(kt + 2); issue(kt + 2, fresh); }
            __builtin_amdgcn_sched_barrier(0);
            kstep(kt & 1, 0);
            kstep(kt & 1, (ABL & 2) ? 0 : 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk && !(ABL & 1)) commit((kt + 1) & 1, next);
            dma_wait();
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, g0, g1);
            if (kt + 1 < nk) step(kt + 1, g1, g0);
        }
    } else if constexpr ((ABL == 0 || ABL >= 64) && SHADOW && DMA >= 1) {      // (register-staged B, DMA == 0: the plain loop below)      // (64 / 128: whole-kernel ablations of the callers, the loop itself is the shipped one)
        // "Staging in the shadow" (round 5).  A SIMD runs its two waves almost one at a time (the older wave wins every issue
        // arbitration: profiles/r05_attn_phases.txt, r02_x3_gemm_phases.txt), so what a wave does outside its MFMA stream is paid
        // in full: the plain loop below spends ~700 clocks issuing a chunk's eight memory instructions in FRONT of its 48 MFMAs and
        // ~600 splitting and writing the staged A operand BEHIND them, per wave.  Here both ride inside the stream: the loads of
        // chunk kt + 1 are issued between the MFMAs of k-step 0 of chunk kt (one A load and one B DMA piece per 32-row block), and
        // their split / LDS writes between the MFMAs of k-step 1 (one staged quad per block: ~25 vector instructions behind 6 MFMAs,
        // 1 : 4 by sched_group_barrier — a VOP2 costs a wave 4 clocks, an MFMA 33.5, profiles/r05_mfma_valu_overlap.txt).  Same
        // MFMA order, same products: bit-identical to the plain loop.
        static_assert(MI == 4, "one staging piece per 32-row block of the 256-row tile");
        // a tile's DMA leaves in MI rounds of one 1-KiB piece per wave: both planes of B (and of A when it travels that way) must fit
        static_assert(2 * (C::BN / 16) <= (C::NT / 64) * MI && 2 * (C::BM / 16) <= (C::NT / 64) * MI, "MI DMA rounds cover the tile");
        Regs g;
        adv(0);
        issue(0, g);
        dma(0, 0);
        commit(0, g);
        dma_wait();
        __syncthreads();
        auto issue_piece = [&](int kt, int p) {
            if (p == 0) { g.ok = 0u; g.kt = kt; }
            if constexpr (DMA < 2) {
                if (p < NA) {
                    if constexpr (APLANES) { g.ah[p] = la(p, kt, 0); g.al[p] = la(p, kt, 1); }
                    else g.a[p] = la(p, kt);
                    g.ok |= (oka(p, kt) ? 1u : 0u) << p;
                }
            }
        };
        // piece i of the B tile's DMA (dma_tile's i-th round: wave w moves 1-KiB piece w + 8 i), and of A's when it travels that way
        auto dma_piece = [&](int buf, int kt, int i) {
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
            constexpr int NWV = C::NT / 64;
            {
                constexpr int PIECES = C::BN / 16;
                const int piece = wv + NWV * i;
                if (piece < 2 * PIECES) {
                    const int plane = piece / PIECES, pc = piece % PIECES;
                    const int row = pc * 16 + (ln >> 2);
                    const int slot = (ln & 3) ^ ((row >> 2) & 3);
                    __builtin_amdgcn_global_load_lds(bptr(row, plane, kt) + slot * 8, (plane ? s.bl[buf] : s.bh[buf]) + pc * 16 * BK, 16, 0, 0);
                }
            }
            if constexpr (DMA >= 2) {
                constexpr int PIECES = C::BM / 16;
                const int piece = wv + NWV * i;
                if (piece < 2 * PIECES) {
                    const int plane = piece / PIECES, pc = piece % PIECES;
                    const int row = pc * 16 + (ln >> 2);
                    const int slot = (ln & 3) ^ ((row >> 2) & 3);
                    __builtin_amdgcn_global_load_lds(aptr(row, plane, kt) + slot * 8, (plane ? s.al[buf] : s.ah[buf]) + pc * 16 * BK, 16, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int buf, int p) {
            if (p >= NA) return;
            if constexpr (DMA < 2) {
                if constexpr (APLANES) {
                    const int row = qrow + C::RQ * p;
                    uint4 vh = g.ah[p], vl = g.al[p];
                    if (!((g.ok >> p) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
                    const int off = row * BK + swz(qsl, row) * 8;
                    *reinterpret_cast<uint4*>(&s.ah[buf][off]) = vh;
                    *reinterpret_cast<uint4*>(&s.al[buf][off]) = vl;
                } else {
                    const int row = arow + C::RA * p;
                    float4 v = g.a[p];
                    if (!((g.ok >> p) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    else axf(v, p, g.kt);
                    half4 hi, lo;
                    split4(v, a_scale, hi, lo, amax);
                    const int off = row * BK + swz(akq >> 1, row) * 8 + (akq & 1) * 4;
                    *reinterpret_cast<half4*>(&s.ah[buf][off]) = hi;
                    *reinterpret_cast<half4*>(&s.al[buf][off]) = lo;
                }
            }
        };
        // one k-step with a piece of staging work behind each 32-row block's six MFMAs
        auto kstep_with = [&](int cur, int ks, auto&& piece) {
            const int arow0 = (wm * 32 * MI + r) * BK, brow0 = (wn * 64 + r) * BK;
            const int slot = swz(2 * ks + h, r) * 8;
            half8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                bh[ni] = *reinterpret_cast<const half8*>(&s.bh[cur][brow0 + ni * 32 * BK + slot]);
                bl[ni] = *reinterpret_cast<const half8*>(&s.bl[cur][brow0 + ni * 32 * BK + slot]);
            }
            ah[0] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + slot]);
            al[0] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + slot]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi + 1 < MI) {
                    ah[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + (mi + 1) * 32 * BK + slot]);
                    al[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + (mi + 1) * 32 * BK + slot]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
                piece(mi);
            }
        };
        auto chunk = [&](int kt, auto more_t) {
            constexpr bool more = decltype(more_t)::value;
            if constexpr (more) adv(kt + 1);
            kstep_with(kt & 1, 0, [&](int mi) {
                if constexpr (more) {
                    issue_piece(kt + 1, mi);
                    dma_piece((kt + 1) & 1, kt + 1, mi);
                    // six MFMAs, the address arithmetic and the two memory instructions of the piece spread behind them
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
                        if (i == 2 || i == 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            });
            kstep_with(kt & 1, 1, [&](int mi) {
                if constexpr (more) {
                    commit_piece((kt + 1) & 1, mi);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        if (i == 5) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    }
                }
            });
            dma_wait();
            __syncthreads();
        };
        for (int kt = 0; kt + 1 < nk; ++kt) chunk(kt, std::true_type{});
        chunk(nk - 1, std::false_type{});
    } else {
        Regs g;
        adv(0);
        issue(0, g);
        dma(0, 0);
        commit(0, g);
        dma_wait();
        __syncthreads();
        unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull}, tl0 = 0ull;
        if constexpr ((ABL & 4) != 0) tl0 = __builtin_readcyclecounter();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            unsigned long long t0 = 0ull, t1 = 0ull, t2 = 0ull, t3 = 0ull, s1 = 0ull, s2 = 0ull;
            if constexpr ((ABL & 4) != 0) t0 = __builtin_readcyclecounter();
            if (more && !(ABL & 1)) { adv(kt + 1); issue(kt + 1, g); dma((kt + 1) & 1, kt + 1); }
            __builtin_amdgcn_sched_barrier(0);       // the loads go out first; nothing of commit() (its waits) moves above the MFMAs
            if constexpr ((ABL & 4) != 0) s1 = __builtin_readcyclecounter();
            kstep(kt & 1, 0);
            if constexpr ((ABL & 4) != 0) { __builtin_amdgcn_sched_barrier(0); s2 = __builtin_readcyclecounter(); }
            kstep(kt & 1, (ABL & 2) ? 0 : 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 4) != 0) {
                t1 = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t2 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more && !(ABL & 1)) commit((kt + 1) & 1, g);
            dma_wait();
            if constexpr ((ABL & 4) != 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t3 = __builtin_readcyclecounter(); }
            __syncthreads();
            if constexpr ((ABL & 4) != 0) {
                const unsigned long long t4 = __builtin_readcyclecounter();
                ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
                if (blockIdx.x == 0 && kt == 3 && lane == 0) {
                    unsigned long long* q = &prof[8 + 8 * wave];
                    q[0] = t0; q[1] = s1; q[2] = s2; q[3] = t1; q[4] = t2; q[5] = t3; q[6] = t4; q[7] = 0ull;
                }
            }
        }
        if constexpr ((ABL & 4) != 0) {
            if (tid == 0) {
                for (int i = 0; i < 4; ++i) atomicAdd(&prof[i], ph[i]);
                atomicAdd(&prof[4], (unsigned long long)__builtin_readcyclecounter() - tl0);
                atomicAdd(&prof[5], 1ull);
            }
        }
    }
}

}  // namespace gemmx3w
