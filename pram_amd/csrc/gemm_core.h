// Exact-fp32 MFMA GEMM main loop for gfx950, shared by the token linear layers and the
// implicit-GEMM convolutions.
//
//   C[128 x 128] per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave, 2 x 2 tiles of
//   v_mfma_f32_32x32x2_f32), K consumed in chunks of 32 or 16 through a double-buffered LDS stage.
//
// Both operands are "K-contiguous rows" (activations [row][k], weights [col][k]), so A and B
// fragments use the same LDS image and the same read: lane l (r = l & 31, h = l >> 5) reads one
// b128 = 4 consecutive k at k-offset 8*kk + 4*h of row r; MFMA step j then contracts
// k = 8*kk + j (lower half-wave) and k = 8*kk + 4 + j (upper half-wave).  The operand maps
// (A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]) only require that A and B agree on which k a
// half-wave supplies, so one ds_read_b128 feeds four MFMAs.
//
// LDS rows are padded to BK + 4 floats: the 16-B slot of row r is 9r (BK = 32) or 5r (BK = 16) + const
// (mod 16), a bijection over any 16 rows distinct mod 16, so every 16-lane group of ds_read_b128 is
// conflict-free; the staging ds_write_b128 (KQ lanes = one contiguous row) is too for BK = 32 (for BK = 16 the
// PMC pass shows write conflicts; an unpadded XOR-swizzled image that removes them measured slower).
//
// f32 MFMA runs at the f32 vector rate (64 cycles per 32x32x2 per SIMD): per 32-deep chunk a wave issues
// 64 MFMA = 4096 matrix cycles against 16 ds_read_b128 and 8 global float4 loads, so plain register staging
// with one barrier per chunk is enough; measured MfmaUtil is 64-72 % (profiles/r01_pmc_summary.md), the rest
// being barrier skew and the per-tile prologue / epilogue of the short-K token GEMMs.
#pragma once
#include "common.h"

namespace gemm {

constexpr int NT = 256;

// Tile geometry: 4 waves as WM x WN (WN = 2 or 1), each wave MI x 2 accumulators of 32 x 32:
//   BM = WM * 32 * MI,  BN = WN * 64.   <2,2> = 128 x 128 (default), <1,2> = 64 x 128 (more, smaller
//   workgroups when a 128-row grid cannot fill 2 x 256 CU slots), <2,1> = 256 x 64 and <1,1> = 128 x 64
//   (64-channel outputs: conv1a / conv1b).
// K chunk BK = 32 or 16.  BK = 16 halves the LDS stage and the staging registers (40 KB, <= 168 VGPRs for
// <2,2>), which buys a THIRD workgroup per CU (3 waves per SIMD) at the price of twice the barriers: a win
// where the per-tile overhead counts — the token GEMMs with K = 256..512 (+2..6 %) and the 64-channel
// convolutions (conv1a: K = 36 pads to 48 instead of 64, 744 -> 447 us) — a small loss (1..8 %) on the deep-K
// 3x3 convolutions, which keep BK = 32.  The k order of the MFMA sequence is the same either way, so the
// choice never changes a result bit.
template <int MI, int WN, int BK_>
struct Cfg {
    static_assert(BK_ == 16 || BK_ == 32, "BK");
    static constexpr int BK = BK_;
    static constexpr int LDT = BK + 4;        // padded LDS row (floats)
    static constexpr int KQ = BK / 4;         // float4 slots per row per chunk
    static constexpr int RPP = NT / KQ;       // rows staged per pass of the 256 threads
    static constexpr int SUB = BK / 8;        // 8-deep sub-chunks (one ds_read_b128 per operand tile each)
    static constexpr int WAVES = BK == 16 ? 3 : 2;   // waves per SIMD the kernel is compiled for
    static constexpr bool INTERLEAVE = BK == 32;     // staging issue interleaved with the first MFMA group (see mainloop)
    static constexpr int WM = 4 / WN;
    static constexpr int BM = WM * 32 * MI;
    static constexpr int BN = WN * 64;
    static constexpr int PA = BM / RPP;       // float4 staging loads per thread for A
    static constexpr int PB = BN / RPP;       // ... for B
    static __device__ __forceinline__ int stage_row(int tid) { return tid / KQ; }
    static __device__ __forceinline__ int stage_kq(int tid) { return tid % KQ; }
};

template <class C>
struct alignas(16) Smem {
    float a[2][C::BM * C::LDT];
    float b[2][C::BN * C::LDT];
};  // <2,2,32>: 73,728 B -> two workgroups per CU; <2,2,16>: 40,960 B -> three

// Guarded staging loads are branch-free and split in two: the loader returns the RAW float4 from an
// always-legal (clamped) address, and a separate predicate says whether the element is in range.  The
// zeroing select is applied only when the registers are written to LDS (after the MFMAs of the current
// chunk), so nothing touches the loaded value — and no s_waitcnt vmcnt lands — before the matrix work.
__device__ __forceinline__ float4 zero_unless(float4 v, bool ok) {
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// ALoad: float4 operator()(int p, int kt) -> raw A[row = tid/KQ + RPP*p][kt*BK + (tid%KQ)*4 ..+3]; AOk: its predicate.
// BLoad / BOk: same for the weight rows.
// Adv(kt) is called once per chunk before its loads: wave-uniform loader state (e.g. the convolution's current
// tap / channel offset) advances incrementally there instead of being re-derived with integer divisions per load.
template <class C, int MI, class Adv, class ALoad, class AOk, class BLoad, class BOk>
__device__ __forceinline__ void mainloop(Smem<C>& s, Adv& adv, ALoad& la, AOk& oka, BLoad& lb, BOk& okb, int nk,
                                         f32x16 (&acc)[MI][2]) {
    constexpr int WN = C::BN / 64, LDT = C::LDT, RPP = C::RPP, SUB = C::SUB;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int srow = C::stage_row(tid), skq = C::stage_kq(tid);

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // one chunk of matrix work on LDS buffer `cur`
    auto compute = [&](int cur, auto&& stage) {
        const float* sa = &s.a[cur][(wm * 32 * MI + r) * LDT + h * 4];
        const float* sb = &s.b[cur][(wn * 64 + r) * LDT + h * 4];
        // fragments of sub-chunk kk+1 are read from LDS before the MFMAs of kk (pinned with sched_barrier:
        // hipcc otherwise sinks every ds_read next to its first use and the wave eats the LDS latency)
        float4 af[2][MI], bf[2][2];
        auto fload = [&](int kk, float4 (&a)[MI], float4 (&b)[2]) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const float4*>(sa + mi * 32 * LDT + kk * 8);
            b[0] = *reinterpret_cast<const float4*>(sb + kk * 8);
            b[1] = *reinterpret_cast<const float4*>(sb + 32 * LDT + kk * 8);
        };
        auto fmma = [&](const float4 (&a)[MI], const float4 (&b)[2]) {
            float av[MI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { av[mi][0] = a[mi].x; av[mi][1] = a[mi].y; av[mi][2] = a[mi].z; av[mi][3] = a[mi].w; }
            const float b0[4] = {b[0].x, b[0].y, b[0].z, b[0].w};
            const float b1[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], b0[j], acc[mi][0], 0, 0, 0);
                    acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], b1[j], acc[mi][1], 0, 0, 0);
                }
            }
        };
        fload(0, af[0], bf[0]);
        if (SUB > 1) fload(1, af[1], bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        // INTERLEAVE (BK = 32, two waves per SIMD): `stage` holds the next chunk's address arithmetic and global loads,
        // issued in the shadow of the first MFMA group — one load and a few VALU per two MFMAs — instead of in one
        // burst before the matrix work (-3..5 % on the deep-K convolutions).  With three waves per SIMD (BK = 16) the
        // burst is already covered by the other waves and the interleaved order measured 3-10 % slower: there the
        // caller issues the loads before compute() and passes an empty `stage`.
        stage();
        fmma(af[0], bf[0]);
        if (2 < SUB) fload(2, af[0], bf[0]);
        if constexpr (C::INTERLEAVE) {
#pragma unroll
            for (int g = 0; g < 2 * MI; ++g) {            // pattern found by measurement; what does not fit it follows
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU (address arithmetic)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 global load
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 1; kk < SUB; ++kk) {
            fmma(af[kk & 1], bf[kk & 1]);
            if (kk + 2 < SUB) fload(kk + 2, af[kk & 1], bf[kk & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // staging: issue() starts the global loads of a chunk into a register set and records its predicates (from the
    // addresses only: nothing waits on the data); commit() zero-fills by predicate and writes the set to an LDS buffer
    auto issue = [&](int kt, float4 (&ra)[C::PA], float4 (&rb)[C::PB], unsigned& ok) {
        ok = 0u;
#pragma unroll
        for (int p = 0; p < C::PA; ++p) { ra[p] = la(p, kt); ok |= (oka(p, kt) ? 1u : 0u) << p; }
#pragma unroll
        for (int p = 0; p < C::PB; ++p) { rb[p] = lb(p, kt); ok |= (okb(p, kt) ? 1u : 0u) << (8 + p); }
    };
    auto commit = [&](int buf, const float4 (&ra)[C::PA], const float4 (&rb)[C::PB], unsigned ok) {
#pragma unroll
        for (int p = 0; p < C::PA; ++p)
            *reinterpret_cast<float4*>(&s.a[buf][(srow + RPP * p) * LDT + skq * 4]) = zero_unless(ra[p], (ok >> p) & 1u);
#pragma unroll
        for (int p = 0; p < C::PB; ++p)
            *reinterpret_cast<float4*>(&s.b[buf][(srow + RPP * p) * LDT + skq * 4]) = zero_unless(rb[p], (ok >> (8 + p)) & 1u);
    };

    // chunk kt+1 is loaded under the MFMAs of chunk kt.  (Issuing chunk kt+2 there instead — two register sets, same
    // LDS double buffer — was measured for the 16-deep chunks: within +-1 % everywhere, +1.5 % only at K = 4096.)
    float4 ra[C::PA], rb[C::PB];
    unsigned ok;
    adv(0);
    issue(0, ra, rb, ok);
    commit(0, ra, rb, ok);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) adv(kt + 1);
        if constexpr (C::INTERLEAVE) {
            const int lk = more ? kt + 1 : kt;   // the last iteration re-issues its own (legal) addresses: no branch in the region
            compute(kt & 1, [&] { issue(lk, ra, rb, ok); });
        } else {
            if (more) issue(kt + 1, ra, rb, ok);
            compute(kt & 1, [] {});
        }
        if (more) commit((kt + 1) & 1, ra, rb, ok);
        __syncthreads();
    }
}

// Accumulator element e of tile (mi, ni) of this lane lives at
//   row = 32*MI*wm + 32*mi + (e & 3) + 8*(e >> 2) + 4*h ,  col = 64*wn + 32*ni + (lane & 31)
__device__ __forceinline__ int acc_row(int mi, int e, int h) { return 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h; }

// Tile choice: 64-wide N tiles for <= 64 output channels; halve BM when the 128-row grid cannot put two
// workgroups on every CU (the second co-resident workgroup is what hides barrier / epilogue time).
inline void choose_tile(int m, int n, int* mi, int* wn) {
    *wn = (n <= 64) ? 1 : 2;
    const int bn = *wn * 64, bm2 = (4 / *wn) * 64;
    const long blocks = (long)cdiv(m, bm2) * cdiv(n, bn);
    *mi = (blocks < 2 * 256) ? 1 : 2;
}

}  // namespace gemm
