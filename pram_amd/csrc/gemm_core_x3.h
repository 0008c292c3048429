// Split-fp16 ("x3") GEMM main loop for gfx950: fp32-class products on the fp16 matrix pipe.
//
// f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 VECTOR rate, 1/16 of the fp16 rate.  Here every fp32 operand x is
// carried as two fp16 numbers,
//     x * s = hi + lo,   hi = fp16(x * s),  lo = fp16(x * s - hi)        (s = a power of two: exact)
// and a product is formed from three fp16 MFMAs accumulated in fp32,
//     a . b  ~=  (a_hi . b_hi + a_hi . b_lo + a_lo . b_hi) / (s_a s_b)       (a_lo . b_lo ~ 2^-22 |a b| is dropped).
// Products of fp16 values are exact in the fp32 accumulator, so the only errors are the 2^-22-relative truncation of
// hi + lo and the dropped term: the same class as the fp32 rounding of the f32-MFMA path (measured end to end against
// the fp32 oracle: SegNetViT logits 1.4e-5, GML / AdaGML indices identical at 2048 x 2048 —
// profiles/tools/split_emulation.py; single-product fp16 gives 1.1e-2).  3 x 32 cycles per 32x32x16 block against
// 8 x 64 cycles for the same block in f32 MFMAs: 5.3x less matrix time.
//
// The scale keeps the parts inside fp16's range: activations use ACT_SCALE = 16 (hi overflows only beyond |x| = 4094;
// lo is a normal fp16 number down to |x| = 2^-7 and loses at most 2^-29 absolute below that); weights are scaled per
// tensor to the top of the range when they are split once on the host.
//
// Tiling as gemm_core.h (so the epilogues are shared): 4 waves as WM x WN, MI x 2 accumulators of 32 x 32 per wave,
// K chunks of 32 through a double-buffered LDS stage.  A (activations / im2col) arrives as fp32 and is split while it
// is staged: 8 threads read one 128-B row segment as float4 and store 2 x 8 bytes; B (weights) is pre-split: 4 threads
// read one 64-B row segment of each plane.  LDS rows are 64 B (32 fp16): four rows share the 64 banks, and the 16-B
// slot is XOR-swizzled with (row >> 2) & 3, which makes every 16-lane group of a ds_read_b128 touch 16 distinct
// (bank-row quarter, slot) pairs.
#pragma once
#include "common.h"

namespace gemmx3 {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32, NT = 256;
constexpr float ACT_SCALE = 16.0f;       // power of two: scaling is exact

template <int MI, int WN>
struct Cfg {
    static constexpr int WM = 4 / WN;
    static constexpr int BM = WM * 32 * MI;
    static constexpr int BN = WN * 64;
    static constexpr int PA = BM / 32;   // float4 (fp32) staging loads per thread for A
    static constexpr int PB = BN / 64;   // 16-byte (8 x fp16) staging loads per thread and plane for B
};

template <int MI, int WN>
struct alignas(16) Smem {
    _Float16 ah[2][Cfg<MI, WN>::BM * BK];
    _Float16 al[2][Cfg<MI, WN>::BM * BK];
    _Float16 bh[2][Cfg<MI, WN>::BN * BK];
    _Float16 bl[2][Cfg<MI, WN>::BN * BK];
};  // <2,2>: 64 KiB -> two workgroups per CU

__device__ __forceinline__ int swz(int slot, int row) { return slot ^ ((row >> 2) & 3); }

// lo part of a split, fp16(x - hi), as ONE instruction: fma(x, 1, -hi) with the 1 hidden from the optimiser selects v_fma_mixlo /
// mixhi_f16 (hi read as fp16); written as a subtraction it is convert, subtract, convert.  The difference is exact in fp32: same bits.
__device__ __forceinline__ _Float16 lo_part(float x, _Float16 hi) {
    float one = 1.0f;
    asm("" : "+s"(one));
    return (_Float16)__builtin_fmaf(x, one, -(float)hi);
}

// x * s -> (hi, lo) for four values
__device__ __forceinline__ void split4(const float4& v, float s, half4& hi, half4& lo) {
    const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 h = (_Float16)x[i];
        hi[i] = h;
        lo[i] = lo_part(x[i], h);
    }
}

// the same, tracking amax = max |x * s| for the range guard (common.h): products of an fmul are canonical, so the maxima are
// two v_max3_f32 with |.| modifiers per four values
__device__ __forceinline__ void split4(const float4& v, float s, half4& hi, half4& lo, float& amax) {
    const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#ifndef PRAM_NO_RANGE_GUARD      // A/B builds only (profiles/): what the tracking costs
    amax = fmaxf(fmaxf(amax, fabsf(x[0])), fabsf(x[1]));
    amax = fmaxf(fmaxf(amax, fabsf(x[2])), fabsf(x[3]));
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 h = (_Float16)x[i];
        hi[i] = h;
        lo[i] = lo_part(x[i], h);
    }
}

// ALoad(p, kt) -> raw float4 A[row = tid/8 + 32p][kt*32 + (tid%8)*4 ..+3];  AOk(p, kt) its predicate
// BLoad(p, kt, plane) -> raw uint4 W_plane[col = tid/4 + 64p][kt*32 + (tid%4)*8 ..+7] (fp16); BOk(p, kt) its predicate
// Adv(kt): wave-uniform loader state, advanced once per chunk (the convolution's tap / channel walk).
// AXf(v, p, kt): optional transform of a staged A quad (row slot p, chunk kt) before it is split — the LayerNorm + GELU of an MLP's
// hidden layer applied on the way into the second GEMM (linear.hip); NoXform compiles to nothing.
struct NoXform {
    __device__ __forceinline__ void operator()(float4&, int, int) const {}
};

template <int MI, int WN, class Adv, class ALoad, class AOk, class BLoad, class BOk, class AXf>
__device__ __forceinline__ void mainloop(Smem<MI, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                         float a_scale, f32x16 (&acc)[MI][2], float& amax, AXf& axf);

template <int MI, int WN, class Adv, class ALoad, class AOk, class BLoad, class BOk>
__device__ __forceinline__ void mainloop(Smem<MI, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                         float a_scale, f32x16 (&acc)[MI][2], float& amax) {
    NoXform none;
    mainloop<MI, WN>(s, adv, la, oka, lb, okb, nk, a_scale, acc, amax, none);
}

template <int MI, int WN, class Adv, class ALoad, class AOk, class BLoad, class BOk, class AXf>
__device__ __forceinline__ void mainloop(Smem<MI, WN>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                         float a_scale, f32x16 (&acc)[MI][2], float& amax, AXf& axf) {
    using C = Cfg<MI, WN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int arow = tid >> 3, akq = tid & 7;
    const int brow = tid >> 2, bsl = tid & 3;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // Two chunks of global loads are in flight: a chunk is only 24 MFMAs = 768 matrix cycles per wave, far less than a
    // global-load round trip under load, so chunk kt + 2 is requested (into the register set chunk kt has just left)
    // before chunk kt is multiplied, and chunk kt + 1 — requested one iteration earlier — is split and written to LDS after.
    struct Regs {
        float4 a[C::PA];
        uint4 bh[C::PB], bl[C::PB];
        unsigned ok;
        int kt;
    };
    auto issue = [&](int kt, Regs& g) {
        g.ok = 0u;
        g.kt = kt;
#pragma unroll
        for (int p = 0; p < C::PA; ++p) { g.a[p] = la(p, kt); g.ok |= (oka(p, kt) ? 1u : 0u) << p; }
#pragma unroll
        for (int p = 0; p < C::PB; ++p) { g.bh[p] = lb(p, kt, 0); g.bl[p] = lb(p, kt, 1); g.ok |= (okb(p, kt) ? 1u : 0u) << (8 + p); }
    };
    auto commit = [&](int buf, const Regs& g) {
#pragma unroll
        for (int p = 0; p < C::PA; ++p) {
            const int row = arow + 32 * p;
            float4 v = g.a[p];
            if (!((g.ok >> p) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            else axf(v, p, g.kt);
            half4 hi, lo;
            split4(v, a_scale, hi, lo, amax);      // amax: range guard (common.h), reported by the caller
            const int off = row * BK + swz(akq >> 1, row) * 8 + (akq & 1) * 4;
            *reinterpret_cast<half4*>(&s.ah[buf][off]) = hi;
            *reinterpret_cast<half4*>(&s.al[buf][off]) = lo;
        }
#pragma unroll
        for (int p = 0; p < C::PB; ++p) {
            const int row = brow + 64 * p;
            uint4 vh = g.bh[p], vl = g.bl[p];
            if (!((g.ok >> (8 + p)) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
            const int off = row * BK + swz(bsl, row) * 8;
            *reinterpret_cast<uint4*>(&s.bh[buf][off]) = vh;
            *reinterpret_cast<uint4*>(&s.bl[buf][off]) = vl;
        }
    };
    // Fragments of k-step 1 are requested before the MFMAs of k-step 0 (pinned with sched_barrier: hipcc otherwise sinks
    // every ds_read next to its first use and each MFMA group starts with an exposed LDS round trip).
    struct Frag {
        half8 ah[MI], al[MI], bh[2], bl[2];
    };
    auto fload = [&](int cur, int ks, Frag& f) {
        const int arow0 = (wm * 32 * MI + r) * BK, brow0 = (wn * 64 + r) * BK;
        const int slot = swz(2 * ks + h, r) * 8;      // tile rows differ from r by multiples of 32: same swizzle
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            f.ah[mi] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + mi * 32 * BK + slot]);
            f.al[mi] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + mi * 32 * BK + slot]);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            f.bh[ni] = *reinterpret_cast<const half8*>(&s.bh[cur][brow0 + ni * 32 * BK + slot]);
            f.bl[ni] = *reinterpret_cast<const half8*>(&s.bl[cur][brow0 + ni * 32 * BK + slot]);
        }
    };
    auto fmma = [&](const Frag& f) {
        // small terms first, the dominant hi.hi last: each accumulator sees lo.hi, hi.lo, hi.hi
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
    };
    auto compute = [&](int cur) {
        Frag f0, f1;
        fload(cur, 0, f0);
        fload(cur, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f0);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f1);
    };

    if constexpr (C::PA <= 4) {
        Regs g0, g1;
        adv(0);
        issue(0, g0);
        if (nk > 1) { adv(1); issue(1, g1); }
        commit(0, g0);
        __syncthreads();
        // one step: chunk kt (in LDS buffer kt & 1) is multiplied; `fresh` = the set chunk kt came from (free again: receives
        // chunk kt + 2), `next` = the set holding chunk kt + 1 (goes to LDS after the MFMAs)
        auto step = [&](int kt, Regs& fresh, Regs& next) {
            if (kt + 2 < nk) { adv(kt + 2); issue(kt + 2, fresh); }
            __builtin_amdgcn_sched_barrier(0);   // the loads go out first; nothing of commit() (its waits!) moves above the MFMAs
            compute(kt & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) commit((kt + 1) & 1, next);
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, g0, g1);
            if (kt + 1 < nk) step(kt + 1, g1, g0);
        }
    } else {
        // 256-row tiles (64-channel outputs): one register set — a second one does not fit beside 8 staging loads per thread
        Regs g;
        adv(0);
        issue(0, g);
        commit(0, g);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) { adv(kt + 1); issue(kt + 1, g); }
            __builtin_amdgcn_sched_barrier(0);
            compute(kt & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) commit((kt + 1) & 1, g);
            __syncthreads();
        }
    }
}

// ---- both operands pre-split in HBM ("planes"): A = two fp16 planes [rows][K] written by a producer's epilogue
// (linear / attention / LayerNorm kernels on the x3 path), B = the host-split weights.  Staging is then a pure copy — 16-byte
// loads and ds_write_b128, no conversion arithmetic: the split main loop above is INSTRUCTION-ISSUE bound (150 VALU per 24
// MFMAs per chunk, most of it splitting A again in every column tile), this one is left with ~40 non-MFMA instructions.
// PLoad(p, kt, plane) -> raw uint4 of rows tid/4 + 64p, halves kt*32 + (tid%4)*8 ..+7 of the plane; POk(p, kt) its predicate.
template <int MI, int WN, class ALoad, class AOk, class BLoad, class BOk>
__device__ __forceinline__ void mainloop_planes(Smem<MI, WN>& s, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                                f32x16 (&acc)[MI][2]) {
    using C = Cfg<MI, WN>;
    constexpr int QA = C::BM / 64, QB = C::BN / 64;      // 16-byte loads per thread and plane
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 2, ssl = tid & 3;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    struct Regs {
        uint4 ah[QA], al[QA], bh[QB], bl[QB];
        unsigned ok;
    };
    auto issue = [&](int kt, Regs& g) {
        g.ok = 0u;
#pragma unroll
        for (int p = 0; p < QA; ++p) { g.ah[p] = la(p, kt, 0); g.al[p] = la(p, kt, 1); g.ok |= (oka(p, kt) ? 1u : 0u) << p; }
#pragma unroll
        for (int p = 0; p < QB; ++p) { g.bh[p] = lb(p, kt, 0); g.bl[p] = lb(p, kt, 1); g.ok |= (okb(p, kt) ? 1u : 0u) << (8 + p); }
    };
    auto commit = [&](int buf, const Regs& g) {
#pragma unroll
        for (int p = 0; p < QA; ++p) {
            const int row = srow + 64 * p;
            uint4 vh = g.ah[p], vl = g.al[p];
            if (!((g.ok >> p) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
            const int off = row * BK + swz(ssl, row) * 8;
            *reinterpret_cast<uint4*>(&s.ah[buf][off]) = vh;
            *reinterpret_cast<uint4*>(&s.al[buf][off]) = vl;
        }
#pragma unroll
        for (int p = 0; p < QB; ++p) {
            const int row = srow + 64 * p;
            uint4 vh = g.bh[p], vl = g.bl[p];
            if (!((g.ok >> (8 + p)) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
            const int off = row * BK + swz(ssl, row) * 8;
            *reinterpret_cast<uint4*>(&s.bh[buf][off]) = vh;
            *reinterpret_cast<uint4*>(&s.bl[buf][off]) = vl;
        }
    };
    struct Frag {
        half8 ah[MI], al[MI], bh[2], bl[2];
    };
    auto fload = [&](int cur, int ks, Frag& f) {
        const int arow0 = (wm * 32 * MI + r) * BK, brow0 = (wn * 64 + r) * BK;
        const int slot = swz(2 * ks + h, r) * 8;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            f.ah[mi] = *reinterpret_cast<const half8*>(&s.ah[cur][arow0 + mi * 32 * BK + slot]);
            f.al[mi] = *reinterpret_cast<const half8*>(&s.al[cur][arow0 + mi * 32 * BK + slot]);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            f.bh[ni] = *reinterpret_cast<const half8*>(&s.bh[cur][brow0 + ni * 32 * BK + slot]);
            f.bl[ni] = *reinterpret_cast<const half8*>(&s.bl[cur][brow0 + ni * 32 * BK + slot]);
        }
    };
    auto fmma = [&](const Frag& f) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
    };
    auto body = [&](int kt) {
        Frag f0, f1;
        fload(kt & 1, 0, f0);
        fload(kt & 1, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f0);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f1);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (QA <= 2) {
        Regs g0, g1;
        issue(0, g0);
        if (nk > 1) issue(1, g1);
        commit(0, g0);
        __syncthreads();
        auto step = [&](int kt, Regs& fresh, Regs& next) {
            if (kt + 2 < nk) issue(kt + 2, fresh);
            body(kt);
            if (kt + 1 < nk) commit((kt + 1) & 1, next);
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, g0, g1);
            if (kt + 1 < nk) step(kt + 1, g1, g0);
        }
    } else {
        Regs g;
        issue(0, g);
        commit(0, g);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(kt + 1, g);
            body(kt);
            if (kt + 1 < nk) commit((kt + 1) & 1, g);
            __syncthreads();
        }
    }
}

}  // namespace gemmx3
