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

namespace gemmx3w {

using gemmx3::half4;
using gemmx3::half8;
using gemmx3::split4;
using gemmx3::swz;

constexpr int BK = 32;

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

// APLANES = false: ALoad(p, kt) -> raw float4 A[row = tid/8 + RA p][kt*32 + (tid%8)*4 ..+3] (fp32, split here)
// APLANES = true : ALoad(p, kt, plane) -> raw uint4 plane[row = tid/4 + RQ p][kt*32 + (tid%4)*8 ..+7]
// BLoad(p, kt, plane) -> raw uint4 W_plane[col = tid/4 + RQ p][kt*32 + (tid%4)*8 ..+7];  *Ok: predicates;  Adv as gemm_core_x3.h
template <int MI, int WM, int WN, bool APLANES, class Adv, class ALoad, class AOk, class BLoad, class BOk>
__device__ __forceinline__ void mainloop(Smem<MI, WM, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                         float a_scale, f32x16 (&acc)[MI][2]) {
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
    float4 ra[APLANES ? 1 : C::PA];
    uint4 rah[APLANES ? C::QA : 1], ral[APLANES ? C::QA : 1];
    uint4 rbh[C::QB], rbl[C::QB];
    unsigned ok = 0u;
    auto issue = [&](int kt) {
        ok = 0u;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            if constexpr (APLANES) { rah[p] = la(p, kt, 0); ral[p] = la(p, kt, 1); }
            else ra[p] = la(p, kt);
            ok |= (oka(p, kt) ? 1u : 0u) << p;
        }
#pragma unroll
        for (int p = 0; p < C::QB; ++p) { rbh[p] = lb(p, kt, 0); rbl[p] = lb(p, kt, 1); ok |= (okb(p, kt) ? 1u : 0u) << (8 + p); }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            if constexpr (APLANES) {
                const int row = qrow + C::RQ * p;
                uint4 vh = rah[p], vl = ral[p];
                if (!((ok >> p) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
                const int off = row * BK + swz(qsl, row) * 8;
                *reinterpret_cast<uint4*>(&s.ah[buf][off]) = vh;
                *reinterpret_cast<uint4*>(&s.al[buf][off]) = vl;
            } else {
                const int row = arow + C::RA * p;
                float4 v = ra[p];
                if (!((ok >> p) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                half4 hi, lo;
                split4(v, a_scale, hi, lo);
                const int off = row * BK + swz(akq >> 1, row) * 8 + (akq & 1) * 4;
                *reinterpret_cast<half4*>(&s.ah[buf][off]) = hi;
                *reinterpret_cast<half4*>(&s.al[buf][off]) = lo;
            }
        }
#pragma unroll
        for (int p = 0; p < C::QB; ++p) {
            const int row = qrow + C::RQ * p;
            uint4 vh = rbh[p], vl = rbl[p];
            if (!((ok >> (8 + p)) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
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

    adv(0);
    issue(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) { adv(kt + 1); issue(kt + 1); }
        __builtin_amdgcn_sched_barrier(0);       // the loads go out first; nothing of commit() (its waits) moves above the MFMAs
        kstep(kt & 1, 0);
        kstep(kt & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) commit((kt + 1) & 1);
        __syncthreads();
    }
}

}  // namespace gemmx3w
