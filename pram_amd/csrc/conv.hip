// SFD2 convolution stack (ResNet4x, nets/sfd2.py:135-170,281-293,331-333) as NHWC implicit GEMM
// on the exact-fp32 matrix cores, plus the grouped 3x3 of the ResBlock on the vector ALU.
//
// Implicit GEMM: M = batch*Ho*Wo output pixels (row-major, so 128 consecutive rows of a tile are
// a run of one image row), N = Cout, K = ks*ks*Cin with K ordered (ky, kx, ci) — the NHWC input
// makes every K-chunk of 32 a contiguous 128-B read per pixel, and the weights are repacked once
// on the host to [Cout][ky][kx][Cin].  The A loader does the im2col on the fly (bounds-checked
// zero padding); everything after that is gemm_core.h.  Eval-mode BatchNorm is applied as a
// per-channel scale/shift in the epilogue (same op order as conv -> BN), then residual, then ReLU.
#include <stdlib.h>
#include "gemm_core.h"
#include "gemm_core_f16.h"
#include "gemm_core_x3.h"
#include "gemm_core_x3w.h"

namespace {

struct ConvArgs {
    const float* in; const float* w; const float* bias; const float* scale; const float* shift;
    const float* residual; float* out;
    int batch, h, wd, cin, cout, ks, stride, relu;
    int ho, wo, m, k;
    int tiles_m, tiles_n;
    unsigned int* status;      // range guard of the split-fp16 path (common.h), or nullptr
    _Float16* out_hi; _Float16* out_lo;      // PLANES epilogue: the output * act_scale as two fp16 planes (the next layer's split operand)
    float act_scale = gemmx3::ACT_SCALE;     // split-fp16 path: scale of the activation planes (pram_act_scale() at launch)
};

// Shared epilogue of the fp32 and fp16 main loops: bias -> BN scale/shift -> residual -> ReLU.
template <int MI, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[MI][2], int row0, int col0, int BM, int BN) {
    using gemm::acc_row;
    const int tid = threadIdx.x;
    const int nlast = p.cout - 1;
    // loads first, predicated stores last
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int mlast = p.m - 1;
    const int rbase = row0 + wm * 32 * MI;
    const bool full = (row0 + BM <= p.m) && (col0 + BN <= p.cout);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = col0 + wn * 64 + ni * 32 + r;
        const int cc = min(col, nlast);
        const float bi = p.bias ? p.bias[cc] : 0.f;
        const float sc = p.scale ? p.scale[cc] : 1.f;
        const float sh = p.scale ? p.shift[cc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float q[16];
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    q[e] = p.residual[(size_t)min(rbase + acc_row(mi, e, h), mlast) * p.cout + cc];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mi][ni][e] + bi;
                if (p.scale) v = v * sc + sh;
                if (p.residual) v += q[e];
                if (p.relu) v = fmaxf(v, 0.f);
                q[e] = v;
            }
            if (full) {   // block-uniform fast path
#pragma unroll
                for (int e = 0; e < 16; ++e) p.out[(size_t)(rbase + acc_row(mi, e, h)) * p.cout + col] = q[e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rbase + acc_row(mi, e, h);
                    if (row < p.m && col < p.cout) p.out[(size_t)row * p.cout + col] = q[e];
                }
            }
        }
    }
}

// conv_epilogue followed by F.normalize over the channels of every output pixel (x / max(||x||, 1e-12): the descriptor head,
// reference nets/sfd2.py:333): the workgroup's BN columns are the pixel's whole channel vector (cout <= BN, one column tile), so
// the squared sums are reduced across the 32 lanes of a half-wave (DPP within the 16-lane rows, one swizzle across them) and
// across the WN waves of a tile row through `scratch` ([WN][BM] floats of LDS the main loop is done with) — instead of a second
// kernel that reads the map back and writes it again.
template <int MI, int WN>
__device__ __forceinline__ void conv_epilogue_l2norm(const ConvArgs& p, f32x16 (&acc)[MI][2], int row0, int col0, int BM, float* scratch) {
    using gemm::acc_row;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int nlast = p.cout - 1, mlast = p.m - 1;
    const int rloc = wm * 32 * MI;                      // first tile row of this wave
    const int rbase = row0 + rloc;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = col0 + wn * 64 + ni * 32 + r;
        const int cc = min(col, nlast);
        const float bi = p.bias ? p.bias[cc] : 0.f;
        const float sc = p.scale ? p.scale[cc] : 1.f;
        const float sh = p.scale ? p.shift[cc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mi][ni][e] + bi;
                if (p.scale) v = v * sc + sh;
                if (p.residual) v += p.residual[(size_t)min(rbase + acc_row(mi, e, h), mlast) * p.cout + cc];
                if (p.relu) v = fmaxf(v, 0.f);
                acc[mi][ni][e] = col < p.cout ? v : 0.f;
            }
    }
    __syncthreads();                                    // every wave is out of the main loop: its LDS is free
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float s = acc[mi][0][e] * acc[mi][0][e] + acc[mi][1][e] * acc[mi][1][e];
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, false));     // row_half_mirror
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, false));     // row_mirror
            s += __shfl_xor(s, 16, 64);                                                                                          // the other 16-lane row of the half
            if (r == 0) scratch[wn * BM + rloc + acc_row(mi, e, h)] = s;
        }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tr = rloc + acc_row(mi, e, h);
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) tot += scratch[w * BM + tr];
            const float d = fmaxf(sqrtf(tot), 1e-12f);
            const int row = row0 + tr;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = col0 + wn * 64 + ni * 32 + r;
                if (row < p.m && col < p.cout) p.out[(size_t)row * p.cout + col] = acc[mi][ni][e] / d;
            }
        }
}

// The epilogue of a layer whose result is the split operand of the next split-fp16 layer: instead of fp32 it leaves as fp16 planes
// hi = fp16(16 v), lo = fp16(16 v - hi), [m][cout] each — what that layer's staging would compute, done once here, and the same
// four bytes per value.  bias -> BN scale/shift -> residual -> ReLU as conv_epilogue, with the 16 folded into the constants (a
// power of two: the same bits).  Lane pairs (two neighbouring channels of one pixel) trade halves through a DPP move so that
// every lane still writes four bytes: even lanes two channels of the hi plane, odd lanes the same two channels of the lo plane.
// Needs an even cout.  The largest |16 v| goes to the range guard.
template <int MI, int WN>
__device__ __forceinline__ void conv_epilogue_planes(const ConvArgs& p, f32x16 (&acc)[MI][2], int row0, int col0, int BM, int BN) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int nlast = p.cout - 1, mlast = p.m - 1;
    const int rbase = row0 + wm * 32 * MI;
    const bool full = (row0 + BM <= p.m) && (col0 + BN <= p.cout);
    const bool odd = lane & 1;
    unsigned int* plane = reinterpret_cast<unsigned int*>(odd ? p.out_lo : p.out_hi);
    // bytes of (other : mine): even lanes keep their own low half (hi_r) and take the partner's (hi_r+1); odd lanes take the
    // partner's high half (lo_r-1) and keep their own (lo_r)
    const unsigned int sel = odd ? 0x03020706u : 0x05040100u;
    const unsigned int hc = (unsigned int)p.cout >> 1;      // a row of a plane in 4-byte words
    const bool relu = p.relu;
    float amax = 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = col0 + wn * 64 + ni * 32 + r;
        const int cc = min(col, nlast);
        const float bi = p.bias ? p.bias[cc] : 0.f;
        const float sc = (p.scale ? p.scale[cc] : 1.f) * p.act_scale;
        const float sh = (p.scale ? p.shift[cc] : 0.f) * p.act_scale;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rb = rbase + 32 * mi + 4 * h;
            float q[16];
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    q[e] = p.residual[(size_t)min(rb + (e & 3) + 8 * (e >> 2), mlast) * p.cout + cc] * p.act_scale;
            }
            unsigned int* dst = plane + (size_t)rb * hc + ((unsigned int)(col & ~1) >> 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = (acc[mi][ni][e] + bi) * sc + sh;
                if (p.residual) x += q[e];
                if (relu) x = fmaxf(x, 0.f);
                amax = fmaxf(amax, fabsf(x));
                const _Float16 hi = (_Float16)x, lo = gemmx3::lo_part(x, hi);
                const unsigned int mine = (unsigned int)__builtin_bit_cast(unsigned short, hi) | ((unsigned int)__builtin_bit_cast(unsigned short, lo) << 16);
                const unsigned int other = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
                const unsigned int pair = __builtin_amdgcn_perm(other, mine, sel);
                const int ro = (e & 3) + 8 * (e >> 2);
                if (full || (rb + ro < p.m && col < p.cout)) dst[(size_t)ro * hc] = pair;
            }
        }
    }
    x3_range_flag(p.status, amax);
}

template <bool CIN4, int MI, int WN, int BKT>
__global__ __launch_bounds__(gemm::NT, (gemm::Cfg<MI, WN, BKT>::WAVES)) void conv_kernel(ConvArgs p) {
    using namespace gemm;
    using C = Cfg<MI, WN, BKT>;
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, RPP = C::RPP, KQ = C::KQ;
    __shared__ Smem<C> smem;
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int tid = threadIdx.x;
    const int srow = C::stage_row(tid), skq = C::stage_kq(tid);
    const int row0 = tm * BM, col0 = tn * BN;
    const int pad = p.ks >> 1;

    // per-thread staging rows: decompose the output pixel once
    int iy0[C::PA], ix0[C::PA], roff[C::PA];
    const float* base[C::PA];
    bool ok[C::PA];
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int row = row0 + srow + RPP * pp;
        ok[pp] = row < p.m;
        const int rr = ok[pp] ? row : 0;
        const int ox = rr % p.wo;
        const int t = rr / p.wo;
        const int oy = t % p.ho;
        const int b = t / p.ho;
        iy0[pp] = oy * p.stride - pad;
        ix0[pp] = ox * p.stride - pad;
        base[pp] = p.in + (size_t)b * p.h * p.wd * p.cin;
    }

    // loaders: clamped (always legal) addresses + select, no branches around the loads
    const int nlast = p.cout - 1, klast = p.k - 4;
    // Wave-uniform walk over K = (ky, kx, ci): advanced incrementally once per chunk (no integer divisions in the loop).
    // CIN4 (conv1a): a chunk spans KQ taps, one per staging column, so the tap is per-thread (tap = KQ kt + skq <= 15).
    int cky = 0, ckx = 0, cci = 0;
    auto adv = [&](int kt) {
        if (CIN4) {
            const int tap = kt * KQ + skq;
            cky = (tap * 11) >> 5;            // tap / 3 for tap < 16
            ckx = tap - cky * 3;
            cci = 0;
        } else if (kt == 0) {
            cky = ckx = cci = 0;
        } else {
            cci += BK;
            if (cci >= p.cin) { cci = 0; if (++ckx == p.ks) { ckx = 0; ++cky; } }
        }
    };
    const float* brow[C::PB];
#pragma unroll
    for (int pp = 0; pp < C::PB; ++pp) brow[pp] = p.w + (size_t)min(col0 + srow + RPP * pp, nlast) * p.k;
    auto la = [&](int pp, int kt) -> float4 {
        const int ky = min(cky, p.ks - 1);
        const int iyc = min(max(iy0[pp] + ky, 0), p.h - 1), ixc = min(max(ix0[pp] + ckx, 0), p.wd - 1);
        return *reinterpret_cast<const float4*>(base[pp] + ((size_t)iyc * p.wd + ixc) * p.cin + cci + (CIN4 ? 0 : skq * 4));
    };
    auto oka = [&](int pp, int kt) -> bool {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        return ok[pp] && cky < p.ks && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
    };
    auto lb = [&](int pp, int kt) -> float4 { return *reinterpret_cast<const float4*>(brow[pp] + min(kt * BK + skq * 4, klast)); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + srow + RPP * pp) < p.cout && (kt * BK + skq * 4) < p.k; };

    f32x16 acc[MI][2];
    mainloop<C, MI>(smem, adv, la, oka, lb, okb, (p.k + BK - 1) / BK, acc);

    conv_epilogue<MI, WN>(p, acc, row0, col0, BM, BN);
}

// fp16-operand variant (BASELINE C5): w16 = [cout][ks][ks][cin] in fp16, cin % 64 == 0 (a 64-deep chunk never
// straddles taps).
template <int MI, int WN>
__global__ __launch_bounds__(gemm16::NT, 2) void conv_f16_kernel(ConvArgs p, const _Float16* __restrict__ w16) {
    using namespace gemm16;
    using C = Cfg<MI, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    __shared__ Smem<MI, WN> smem;
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int tid = threadIdx.x;
    const int arow = tid >> 4, akq = tid & 15, brow = tid >> 3, bsl = tid & 7;
    const int row0 = tm * BM, col0 = tn * BN;
    const int pad = p.ks >> 1;
    int iy0[C::PA], ix0[C::PA], roff[C::PA];
    const float* base[C::PA];
    bool ok[C::PA];
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int row = row0 + arow + 16 * pp;
        ok[pp] = row < p.m;
        const int rr = ok[pp] ? row : 0;
        const int ox = rr % p.wo;
        const int t = rr / p.wo;
        const int oy = t % p.ho;
        const int b = t / p.ho;
        iy0[pp] = oy * p.stride - pad;
        ix0[pp] = ox * p.stride - pad;
        base[pp] = p.in + (size_t)b * p.h * p.wd * p.cin;
        roff[pp] = (iy0[pp] * p.wd + ix0[pp]) * p.cin + akq * 4;      // as conv_x3_kernel: 32-bit element offsets inside an image
    }
    const int nlast = p.cout - 1;
    // taps innermost, like conv_x3_kernel below (L2 re-use of the input window); koff = the chunk's offset in a weight row
    int cky = 0, ckx = 0, cci = 0, koff = 0, tap = 0;
    auto adv = [&](int kt) {
        if (kt == 0) { cky = ckx = cci = koff = tap = 0; return; }
        if (++ckx == p.ks) { ckx = 0; if (++cky == p.ks) { cky = 0; cci += BK; } }
        koff = (cky * p.ks + ckx) * p.cin + cci;
        tap = (cky * p.wd + ckx) * p.cin + cci;
    };
    const _Float16* brow16[C::PB];
#pragma unroll
    for (int pp = 0; pp < C::PB; ++pp) brow16[pp] = w16 + (size_t)min(col0 + brow + 32 * pp, nlast) * p.k + bsl * 8;
    auto la = [&](int pp, int kt) -> float4 {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        const bool in = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
        const unsigned off = in ? (unsigned)(roff[pp] + tap) : 0u;      // a padding pixel reads the image's first elements: never used (oka)
        return *reinterpret_cast<const float4*>(base[pp] + off);
    };
    auto oka = [&](int pp, int kt) -> bool {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        return ok[pp] && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
    };
    auto lb = [&](int pp, int kt) -> uint4 { return *reinterpret_cast<const uint4*>(brow16[pp] + koff); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + brow + 32 * pp) < p.cout; };
    f32x16 acc[MI][2];
    mainloop<MI, WN>(smem, adv, la, oka, lb, okb, p.k / BK, acc);
    conv_epilogue<MI, WN>(p, acc, row0, col0, BM, BN);
}

// split-fp16 variant (gemm_core_x3.h): wh / wl = [cout][ks][ks][cin] * w_scale split into two fp16 planes on the host,
// cin % 32 == 0 (a 32-deep chunk never straddles taps); the im2col rows are split while they are staged.
template <int MI, int WN, bool L2N = false>
__global__ __launch_bounds__(gemmx3::NT, 2) void conv_x3_kernel(ConvArgs p, const _Float16* __restrict__ wh,
                                                                 const _Float16* __restrict__ wl, float inv) {
    using namespace gemmx3;
    using C = Cfg<MI, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    __shared__ Smem<MI, WN> smem;
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int tid = threadIdx.x;
    const int arow = tid >> 3, akq = tid & 7, brow = tid >> 2, bsl = tid & 3;
    const int row0 = tm * BM, col0 = tn * BN;
    const int pad = p.ks >> 1;
    int iy0[C::PA], ix0[C::PA], roff[C::PA];
    const float* base[C::PA];
    bool ok[C::PA];
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int row = row0 + arow + 32 * pp;
        ok[pp] = row < p.m;
        const int rr = ok[pp] ? row : 0;
        const int ox = rr % p.wo;
        const int t = rr / p.wo;
        const int oy = t % p.ho;
        const int b = t / p.ho;
        iy0[pp] = oy * p.stride - pad;
        ix0[pp] = ox * p.stride - pad;
        base[pp] = p.in + (size_t)b * p.h * p.wd * p.cin;
        // the thread's element offset inside its image at tap (0, 0), chunk 0 (may be negative: the padding); a tap / chunk adds the
        // workgroup-uniform `tap` — one v_add per load instead of clamps and a 64-bit multiply-add chain (an image holds < 2^31 elements:
        // checked by the launcher)
        roff[pp] = (iy0[pp] * p.wd + ix0[pp]) * p.cin + akq * 4;
    }
    const int nlast = p.cout - 1;
    // K is walked channel chunk by channel chunk with the ks x ks taps innermost: the nine shifted reads of a 32-channel slab
    // of the tile's input window (one 128-byte line per pixel) follow each other, so eight of them hit the L2 — with the taps
    // outermost a workgroup came back to a line only after a whole tap (256 KB per workgroup, 8 MB per XCD against 4 MB of L2)
    // and the 3x3 layers fetched 3.4x their input from HBM (profiles/r02b_pmc_summary.md).  koff = the chunk's offset in the
    // [ky][kx][ci] rows of the weight matrix.
    int cky = 0, ckx = 0, cci = 0, koff = 0, tap = 0;
    auto adv = [&](int kt) {
        if (kt == 0) { cky = ckx = cci = koff = tap = 0; return; }
        if (++ckx == p.ks) { ckx = 0; if (++cky == p.ks) { cky = 0; cci += BK; } }
        koff = (cky * p.ks + ckx) * p.cin + cci;
        tap = (cky * p.wd + ckx) * p.cin + cci;
    };
    size_t boff[C::PB];
#pragma unroll
    for (int pp = 0; pp < C::PB; ++pp) boff[pp] = (size_t)min(col0 + brow + 64 * pp, nlast) * p.k + bsl * 8;
    auto la = [&](int pp, int kt) -> float4 {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        const bool in = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
        const unsigned off = in ? (unsigned)(roff[pp] + tap) : 0u;      // a padding pixel reads the image's first elements: never used (oka)
        return *reinterpret_cast<const float4*>(base[pp] + off);
    };
    auto oka = [&](int pp, int kt) -> bool {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        return ok[pp] && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
    };
    auto lb = [&](int pp, int kt, int plane) -> uint4 { return *reinterpret_cast<const uint4*>((plane ? wl : wh) + boff[pp] + koff); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + brow + 64 * pp) < p.cout; };
    f32x16 acc[MI][2];
    float amax = 0.f;
    mainloop<MI, WN>(smem, adv, la, oka, lb, okb, p.k / BK, p.act_scale, acc, amax);
    x3_range_flag(p.status, amax);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= inv;
    if constexpr (L2N) conv_epilogue_l2norm<MI, WN>(p, acc, row0, col0, BM, reinterpret_cast<float*>(&smem));
    else conv_epilogue<MI, WN>(p, acc, row0, col0, BM, BN);
}

// wide-tile split-fp16 convolution (gemm_core_x3w.h): 256 output pixels x 256 channels per 512-thread workgroup — the 256-channel
// layers (conv3a / conv3b / convDa.* / convPa.* / conv4 1x1: 80 % of the stack's MACs).  Same arithmetic and accumulation order
// as conv_x3_kernel: bit-identical results.
template <int MI, int WM, int WN, bool PLANES = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_x3w_kernel(ConvArgs p, const _Float16* __restrict__ wh,
                                                                    const _Float16* __restrict__ wl, float inv) {
    using namespace gemmx3w;
    using C = Cfg<MI, WM, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<MI, WM, WN>& smem = *reinterpret_cast<Smem<MI, WM, WN>*>(smem_raw);
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int tid = threadIdx.x;
    const int arow = tid >> 3, akq = tid & 7, qrow = tid >> 2, qsl = tid & 3;
    const int row0 = tm * BM, col0 = tn * BN;
    const int pad = p.ks >> 1;
    int iy0[C::PA], ix0[C::PA], roff[C::PA];
    const float* base[C::PA];
    bool ok[C::PA];
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int row = row0 + arow + C::RA * pp;
        ok[pp] = row < p.m;
        const int rr = ok[pp] ? row : 0;
        const int ox = rr % p.wo;
        const int t = rr / p.wo;
        const int oy = t % p.ho;
        const int b = t / p.ho;
        iy0[pp] = oy * p.stride - pad;
        ix0[pp] = ox * p.stride - pad;
        base[pp] = p.in + (size_t)b * p.h * p.wd * p.cin;
        // the thread's element offset inside its image at tap (0, 0), chunk 0 (may be negative: the padding); a tap / chunk adds the
        // workgroup-uniform `tap` — one v_add per load instead of clamps and a 64-bit multiply-add chain (an image holds < 2^31 elements:
        // checked by the launcher)
        roff[pp] = (iy0[pp] * p.wd + ix0[pp]) * p.cin + akq * 4;
    }
    const int nlast = p.cout - 1;
    // K is walked channel chunk by channel chunk with the ks x ks taps innermost: the nine shifted reads of a 32-channel slab
    // of the tile's input window (one 128-byte line per pixel) follow each other, so eight of them hit the L2 — with the taps
    // outermost a workgroup came back to a line only after a whole tap (256 KB per workgroup, 8 MB per XCD against 4 MB of L2)
    // and the 3x3 layers fetched 3.4x their input from HBM (profiles/r02b_pmc_summary.md).  koff = the chunk's offset in the
    // [ky][kx][ci] rows of the weight matrix.
    int cky = 0, ckx = 0, cci = 0, koff = 0, tap = 0;
    auto adv = [&](int kt) {
        if (kt == 0) { cky = ckx = cci = koff = tap = 0; return; }
        if (++ckx == p.ks) { ckx = 0; if (++cky == p.ks) { cky = 0; cci += BK; } }
        koff = (cky * p.ks + ckx) * p.cin + cci;
        tap = (cky * p.wd + ckx) * p.cin + cci;
    };
    size_t boff[C::QB];
#pragma unroll
    for (int pp = 0; pp < C::QB; ++pp) boff[pp] = (size_t)min(col0 + qrow + C::RQ * pp, nlast) * p.k + qsl * 8;
    auto la = [&](int pp, int kt) -> float4 {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        const bool in = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
        const unsigned off = in ? (unsigned)(roff[pp] + tap) : 0u;      // a padding pixel reads the image's first elements: never used (oka)
        return *reinterpret_cast<const float4*>(base[pp] + off);
    };
    auto oka = [&](int pp, int kt) -> bool {
        const int iy = iy0[pp] + cky, ix = ix0[pp] + ckx;
        return ok[pp] && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
    };
    auto lb = [&](int pp, int kt, int plane) -> uint4 { return *reinterpret_cast<const uint4*>((plane ? wl : wh) + boff[pp] + koff); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + qrow + C::RQ * pp) < p.cout; };
    f32x16 acc[MI][2];
    auto aptr = [](int, int, int) -> const _Float16* { return nullptr; };
    auto bptr = [&](int row, int plane, int kt) -> const _Float16* {
        return (plane ? wl : wh) + (size_t)min(col0 + row, nlast) * p.k + koff;
    };
    float amax = 0.f;
    mainloop<MI, WM, WN, false, 0, 1>(smem, adv, la, oka, lb, okb, aptr, bptr, p.k / BK, p.act_scale, acc, amax);
    x3_range_flag(p.status, amax);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= inv;
    if constexpr (PLANES) conv_epilogue_planes<MI, WN>(p, acc, row0, col0, BM, BN);
    else conv_epilogue<MI, WN>(p, acc, row0, col0, BM, BN);
}

// ---------------------------------------------------------------- 3x3 / stride 1 with the input window resident in LDS
// conv_x3w_kernel stages (and splits) the A operand once per K chunk, i.e. once per TAP: every input element of a 3x3 layer goes
// L2 -> registers -> two fp16 parts -> LDS nine times.  Here a workgroup owns 8 x 32 output pixels x 256 channels, and per
// 32-channel slab of the input its (8 + 2) x (32 + 2) pixel window is staged and split ONCE (zero padding written into the tile);
// the nine taps read their A fragments from that window at shifted pixel rows.  Only the weights still move per chunk (LDS-DMA).
// A: 43.5 KB per stage x 2 (the next slab's window is requested at the slab's first tap and written at its last), B: 32 KB x 2.
// K is walked slab by slab with the taps innermost and every accumulator sees lo.hi, hi.lo, hi.hi per 16-deep step — the order
// of conv_x3w_kernel: the two kernels agree bit for bit (tests/test_gpu_guard_chunks_mlp.py::test_conv3x3_halo_equals_chunked).
namespace halo {
constexpr int TH = 8, TW = 32, HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;      // 340 window pixels
constexpr int BK = 32, NT = 512;
constexpr int NL = (HP * 8 + NT - 1) / NT;                                      // float4 loads per thread and slab (6)
// WN = 4: 256 output channels per workgroup, waves as 2 x 4, four tile rows each (the 256-channel layers); WN = 2: 128 channels,
// waves as 4 x 2, two tile rows each (conv2a, 64 -> 128 at 240 x 320: 880 -> 800 us).  Same arithmetic and order either way.
// (For the 128-channel form two more shapes were measured and dropped — 4 x 32-pixel tiles with one window stage so that two
// 4-wave workgroups share a CU, and that with a ring of three weight stages: 816 and 822 us.)
template <int WN_>
struct alignas(16) Smem {
    _Float16 ah[2][HP * BK];
    _Float16 al[2][HP * BK];
    _Float16 bh[2][WN_ * 64 * BK];
    _Float16 bl[2][WN_ * 64 * BK];
};      // 152 576 bytes (WN_ = 4), 119 808 (WN_ = 2)
}  // namespace halo

template <int WN>
__global__ __launch_bounds__(halo::NT, 1) void conv3x3_x3h_kernel(ConvArgs p, const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                                  float inv, int tiles_x, int tiles_y) {
    using namespace halo;
    using gemmx3::half4;
    using gemmx3::half8;
    constexpr int BN = WN * 64, MI = TH / (NT / 64 / WN);      // 256 channels x 4 rows per wave, or 128 x 2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<WN>& s = *reinterpret_cast<Smem<WN>*>(smem_raw);
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n;
    int t = id / p.tiles_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int nlast = p.cout - 1;

    // window staging: element e = tid + NT j is float4 q = e % 8 of window pixel hp = e / 8.  The geometry is recomputed where it is
    // used (once per slab and element: a dozen integer operations) instead of living in twelve registers across the tap loop.
    struct Geo { int hp, q; bool valid, inimg; unsigned goff; };
    auto geo = [&](int j) {
        Geo g;
        const int e = tid + NT * j;
        g.valid = (e >> 3) < HP;
        g.hp = min(e >> 3, HP - 1);
        g.q = e & 7;
        const int hy = g.hp / HWD, hx = g.hp - hy * HWD;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        g.inimg = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
        const int iyc = min(max(iy, 0), p.h - 1), ixc = min(max(ix, 0), p.wd - 1);
        g.goff = (unsigned)((iyc * p.wd + ixc) * p.cin + g.q * 4);      // a batch element is < 2^32 floats
        return g;
    };
    float amax = 0.f;
    float4 hv[NL];
    const float* img = p.in + (size_t)b * p.h * p.wd * p.cin;
    auto hload = [&](int slab) {
#pragma unroll
        for (int j = 0; j < NL; ++j) hv[j] = *reinterpret_cast<const float4*>(img + geo(j).goff + slab * BK);
    };
    auto hcommit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const Geo g = geo(j);
            float4 v = hv[j];
            if (!g.inimg) v = make_float4(0.f, 0.f, 0.f, 0.f);
            half4 hi, lo;
            gemmx3::split4(v, p.act_scale, hi, lo, amax);
            if (g.valid) {
                const int off = g.hp * BK + gemmx3::swz(g.q >> 1, g.hp) * 8 + (g.q & 1) * 4;
                *reinterpret_cast<half4*>(&s.ah[buf][off]) = hi;
                *reinterpret_cast<half4*>(&s.al[buf][off]) = lo;
            }
        }
    };
    auto bdma = [&](int buf, int koff) {
        auto bp = [&](int row, int plane) { return (plane ? wl : wh) + (size_t)min(col0 + row, nlast) * p.k + koff; };
        gemmx3w::dma_tile<BN, NT / 64>(s.bh[buf], s.bl[buf], bp);
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    // one 16-deep k-step of tap (ky, kx): B fragments once, A fragments per 32-pixel row one block ahead of their MFMAs
    auto kstep = [&](int abuf, int bbuf, int ky, int kx, int ks) {
        const int brow0 = (wn * 64 + r) * BK;
        const int bslot = gemmx3::swz(2 * ks + h, r) * 8;
        half8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bh[ni] = *reinterpret_cast<const half8*>(&s.bh[bbuf][brow0 + ni * 32 * BK + bslot]);
            bl[ni] = *reinterpret_cast<const half8*>(&s.bl[bbuf][brow0 + ni * 32 * BK + bslot]);
        }
        auto aoff = [&](int mi) {
            const int hp = (wm * MI + mi + ky) * HWD + r + kx;
            return hp * BK + gemmx3::swz(2 * ks + h, hp) * 8;
        };
        int o = aoff(0);
        ah[0] = *reinterpret_cast<const half8*>(&s.ah[abuf][o]);
        al[0] = *reinterpret_cast<const half8*>(&s.al[abuf][o]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (mi + 1 < MI) {
                o = aoff(mi + 1);
                ah[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.ah[abuf][o]);
                al[(mi + 1) & 1] = *reinterpret_cast<const half8*>(&s.al[abuf][o]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi & 1], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    };
    const int nslab = p.cin / BK;
    hload(0);
    bdma(0, 0);
    hcommit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int c = 0;
    for (int slab = 0; slab < nslab; ++slab) {
        const bool more_slab = slab + 1 < nslab;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++c) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool more = tap < 8 || more_slab;
            if (more) {      // the next chunk's weights: (slab, tap + 1) or (slab + 1, 0)
                const int nt = tap < 8 ? tap + 1 : 0, ns = tap < 8 ? slab : slab + 1;
                bdma((c + 1) & 1, nt * p.cin + ns * BK);
            }
            const bool fetch = tap == 0 && more_slab;
            if (fetch) hload(slab + 1);      // behind the DMA: its wait below leaves these outstanding
            __builtin_amdgcn_sched_barrier(0);
            kstep(slab & 1, c & 1, ky, kx, 0);
            kstep(slab & 1, c & 1, ky, kx, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (tap == 8 && more_slab) hcommit((slab + 1) & 1);
            if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    x3_range_flag(p.status, amax);
    // epilogue: tile row tr = 32 y + x  ->  output pixel (b, oy0 + y, ox0 + x); bias -> BN scale / shift -> residual -> ReLU
    const int rbase = wm * 32 * MI;
    float* out_b = p.out + (size_t)b * p.ho * p.wo * p.cout;
    const float* res_b = p.residual ? p.residual + (size_t)b * p.ho * p.wo * p.cout : nullptr;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = col0 + wn * 64 + ni * 32 + r;
        const int cc = min(col, nlast);
        const float bi = p.bias ? p.bias[cc] : 0.f;
        const float sc = p.scale ? p.scale[cc] : 1.f;
        const float sh = p.scale ? p.shift[cc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tr = rbase + gemm::acc_row(mi, e, h);
                const int oy = oy0 + (tr >> 5), ox = ox0 + (tr & 31);
                if (oy < p.ho && ox < p.wo && col < p.cout) {
                    // a batch element's map holds < 2^31 floats (checked by the entry point): 32-bit offsets from its (scalar) base
                    const unsigned o = (unsigned)(oy * p.wo + ox) * (unsigned)p.cout + (unsigned)col;
                    float v = acc[mi][ni][e] * inv + bi;
                    if (p.scale) v = v * sc + sh;
                    if (p.residual) v += res_b[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    out_b[o] = v;
                }
            }
    }
}

// ---------------------------------------------------------------- grouped 3x3 (VALU)
// groups = 32, 8 in / 8 out channels per group (72-deep dot products: too thin for MFMA tiles).
// Workgroup = 64 consecutive x-pixels x 4 output rows x 4 groups (one group per wave, so the 576
// weights of a group are wave-uniform LDS broadcasts).  A lane owns 4 vertically adjacent pixels of
// one group: each tap's 64 weights are read from LDS once (16 x ds_read_b128) and reused for the
// 4 rows, so LDS issues 1 read per 16 FMAs; the 3x6 input window comes through L1.
struct GConvArgs {
    const float* in; const float* w; const float* scale; const float* shift; float* out;
    int batch, h, wd, c, groups, relu, xtiles, ytiles;
};

__global__ __launch_bounds__(256) void gconv3x3_kernel(GConvArgs p) {
    // Two FMAs per instruction (v_pk_fma_f32) with both operands naturally paired: adjacent INPUT channels (ci, ci + 1) are
    // adjacent in the NHWC pixel and in the [co][tap][ci] weight row, so every output accumulates an even-ci and an odd-ci
    // partial sum in one 64-bit register pair (added at the end) and nothing has to be broadcast or shuffled.
    //
    // The (4 + 2) x (64 + 2) pixel window of the workgroup's 32 channels is staged through LDS once: read from global memory
    // with 8 lanes per pixel (128 contiguous bytes), where the direct form had every lane fetch its own 32 bytes out of a
    // different cache line for each of the 9 taps (64 lines per load instruction: the kernel was bound by L1 requests, 545 us
    // against 72 us of FMA issue and 105 us of HBM time).  Pixel stride 36 floats: the 16 lanes of a ds_read_b128 group land
    // on 16 distinct bank quads.
    constexpr int PXS = 36, TW = 66, TH = 6;
    __shared__ __attribute__((aligned(16))) float sw[4][8 * 72];
    __shared__ __attribute__((aligned(16))) float sx[TH * TW * PXS];
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y * 4 + wave;
    for (int i = lane; i < 576; i += 64) sw[wave][i] = p.w[(size_t)g * 576 + i];
    int t = blockIdx.x;
    const int xt = t % p.xtiles; t /= p.xtiles;
    const int yt = t % p.ytiles;
    const int b = t / p.ytiles;
    const int x0 = xt * 64, y0 = yt * 4;
    {
        // all 13 loads of a thread are issued before the first one is consumed (one round trip, not thirteen)
        const float* ibq = p.in + (size_t)b * p.h * p.wd * p.c + blockIdx.y * 32;
        constexpr int NLD = (TH * TW * 8 + 255) / 256;
        float4 v[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int q = i & 7, pix = min(i >> 3, TH * TW - 1);
            const int ty = pix / TW, tx = pix - ty * TW;
            const int iy = min(max(y0 + ty - 1, 0), p.h - 1), ix = min(max(x0 + tx - 1, 0), p.wd - 1);
            v[j] = *reinterpret_cast<const float4*>(ibq + ((size_t)iy * p.wd + ix) * p.c + q * 4);
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int q = i & 7, pix = i >> 3;
            const int ty = pix / TW, tx = pix - ty * TW;
            const int iy = y0 + ty - 1, ix = x0 + tx - 1;
            if (pix < TH * TW) {
                const bool in = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
                *reinterpret_cast<float4*>(&sx[pix * PXS + q * 4]) = in ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __syncthreads();
    f2 acc2[4][8];
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc2[ro][o] = (f2){0.f, 0.f};
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {   // one tap's 64 weights live at a time (full unroll spills)
        const int ky = tap / 3, kx = tap - ky * 3;
        f2 wt[8][4];                       // [co][pair of ci]
        const float* ws = &sw[wave][tap * 8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float4 w0 = *reinterpret_cast<const float4*>(ws + o * 72);
            const float4 w1 = *reinterpret_cast<const float4*>(ws + o * 72 + 4);
            wt[o][0] = (f2){w0.x, w0.y}; wt[o][1] = (f2){w0.z, w0.w};
            wt[o][2] = (f2){w1.x, w1.y}; wt[o][3] = (f2){w1.z, w1.w};
        }
#pragma unroll
        for (int ro = 0; ro < 4; ++ro) {
            const float* src = &sx[((ro + ky) * TW + lane + kx) * PXS + wave * 8];      // zero padding is in the tile
            const float4 a0 = *reinterpret_cast<const float4*>(src);
            const float4 a1 = *reinterpret_cast<const float4*>(src + 4);
            const f2 xp[4] = {(f2){a0.x, a0.y}, (f2){a0.z, a0.w}, (f2){a1.x, a1.y}, (f2){a1.z, a1.w}};
#pragma unroll
            for (int o = 0; o < 8; ++o)
#pragma unroll
                for (int ip = 0; ip < 4; ++ip) acc2[ro][o] = __builtin_elementwise_fma(xp[ip], wt[o][ip], acc2[ro][o]);
        }
    }
    float acc[4][8];
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[ro][o] = acc2[ro][o][0] + acc2[ro][o][1];
    float sc[8], sh[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        sc[o] = p.scale ? p.scale[g * 8 + o] : 1.f;
        sh[o] = p.scale ? p.shift[g * 8 + o] : 0.f;
    }
    // results go back through the (now free) LDS tile so that the stores are 128 contiguous bytes per pixel as well
    __syncthreads();
#pragma unroll
    for (int ro = 0; ro < 4; ++ro) {
        float res[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float v = acc[ro][o];
            if (p.scale) v = v * sc[o] + sh[o];
            if (p.relu) v = fmaxf(v, 0.f);
            res[o] = v;
        }
        float* dst = &sx[(ro * 64 + lane) * PXS + wave * 8];
        *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(res[4], res[5], res[6], res[7]);
    }
    __syncthreads();
    float* ob = p.out + (size_t)b * p.h * p.wd * p.c + blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j;
        const int q = i & 7, pix = i >> 3;          // pix = ro * 64 + px
        const int y = y0 + (pix >> 6), xx = x0 + (pix & 63);
        if (y < p.h && xx < p.wd)
            *reinterpret_cast<float4*>(ob + ((size_t)y * p.wd + xx) * p.c + q * 4) = *reinterpret_cast<const float4*>(&sx[pix * PXS + q * 4]);
    }
}

__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ img, float* __restrict__ out, int hw, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / hw, px = i - b * hw;
    const float* s = img + (size_t)b * 3 * hw + px;
    *reinterpret_cast<float4*>(out + (size_t)i * 4) = make_float4(s[0], s[hw], s[2 * hw], 0.f);
}

// NHWC -> NCHW through a 32x33 LDS tile (coalesced both ways)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int hw, int c) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int px = p0 + j, ch = c0 + tx;
        tile[j][tx] = (px < hw && ch < c) ? in[((size_t)b * hw + px) * c + ch] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int ch = c0 + j, px = p0 + tx;
        if (px < hw && ch < c) out[((size_t)b * c + ch) * hw + px] = tile[tx][j];
    }
}

}  // namespace

extern "C" int pram_conv2d_nhwc_f32(const float* in, int batch, int h, int w, int cin, const float* wgt,
                                    const float* bias, const float* scale, const float* shift, const float* residual,
                                    float* out, int cout, int ks, int stride, int relu, void* stream) {
    PRAM_REQUIRE(in && wgt && out, "pram_conv2d_nhwc_f32: null pointer");
    PRAM_REQUIRE(ks == 1 || ks == 3, "pram_conv2d_nhwc_f32: ks must be 1 or 3");
    PRAM_REQUIRE(stride == 1 || stride == 2, "pram_conv2d_nhwc_f32: stride must be 1 or 2");
    PRAM_REQUIRE(cin == 4 || cin % 32 == 0, "pram_conv2d_nhwc_f32: cin=%d must be 4 or a multiple of 32", cin);
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv2d_nhwc_f32: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    const int pad = ks / 2;
    ConvArgs p{in, wgt, bias, scale, shift, residual, out, batch, h, w, cin, cout, ks, stride, relu};
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.m = batch * p.ho * p.wo;
    p.k = ks * ks * cin;
    int mi, wn;
    gemm::choose_tile(p.m, cout, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    // 64-channel outputs (conv1a, conv1b: short K, write-heavy) take the 16-deep chunk, the deep-K layers the 32-deep one
#define LAUNCH(C4, MI_, WN_, BK_)                                                                                   \
    do {                                                                                                            \
        p.tiles_m = cdiv(p.m, gemm::Cfg<MI_, WN_, BK_>::BM);                                                        \
        p.tiles_n = cdiv(cout, gemm::Cfg<MI_, WN_, BK_>::BN);                                                       \
        hipLaunchKernelGGL((conv_kernel<C4, MI_, WN_, BK_>), dim3(p.tiles_m * p.tiles_n), dim3(gemm::NT), 0, st, p); \
    } while (0)
    if (cin == 4) { if (wn == 1) LAUNCH(true, 2, 1, 16); else LAUNCH(true, 2, 2, 16); }
    else if (wn == 1) { if (mi == 2) LAUNCH(false, 2, 1, 16); else LAUNCH(false, 1, 1, 16); }
    else { if (mi == 2) LAUNCH(false, 2, 2, 32); else LAUNCH(false, 1, 2, 32); }
#undef LAUNCH
    return pram_launch_status("pram_conv2d_nhwc_f32");
}

extern "C" int pram_conv2d_nhwc_f16_f32(const float* in, int batch, int h, int w, int cin, const void* wgt16,
                                        const float* bias, const float* scale, const float* shift, const float* residual,
                                        float* out, int cout, int ks, int stride, int relu, void* stream) {
    PRAM_REQUIRE(in && wgt16 && out, "pram_conv2d_nhwc_f16_f32: null pointer");
    PRAM_REQUIRE(ks == 1 || ks == 3, "pram_conv2d_nhwc_f16_f32: ks must be 1 or 3");
    PRAM_REQUIRE((long long)h * w * cin < (1ll << 31), "pram_conv2d_nhwc_f16_f32: an image of %d x %d x %d elements does not fit the 32-bit offsets of the im2col loader", h, w, cin);
    PRAM_REQUIRE(stride == 1 || stride == 2, "pram_conv2d_nhwc_f16_f32: stride must be 1 or 2");
    PRAM_REQUIRE(cin % 64 == 0, "pram_conv2d_nhwc_f16_f32: cin=%d must be a multiple of 64", cin);
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv2d_nhwc_f16_f32: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    const int pad = ks / 2;
    ConvArgs p{in, nullptr, bias, scale, shift, residual, out, batch, h, w, cin, cout, ks, stride, relu};
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.m = batch * p.ho * p.wo;
    p.k = ks * ks * cin;
    int mi, wn;
    gemm::choose_tile(p.m, cout, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* w16 = (const _Float16*)wgt16;
#define LAUNCH16(MI_, WN_)                                                                                          \
    do {                                                                                                            \
        p.tiles_m = cdiv(p.m, gemm16::Cfg<MI_, WN_>::BM);                                                           \
        p.tiles_n = cdiv(cout, gemm16::Cfg<MI_, WN_>::BN);                                                          \
        hipLaunchKernelGGL((conv_f16_kernel<MI_, WN_>), dim3(p.tiles_m * p.tiles_n), dim3(gemm16::NT), 0, st, p, w16); \
    } while (0)
    if (wn == 1) { if (mi == 2) LAUNCH16(2, 1); else LAUNCH16(1, 1); }
    else { if (mi == 2) LAUNCH16(2, 2); else LAUNCH16(1, 2); }
#undef LAUNCH16
    return pram_launch_status("pram_conv2d_nhwc_f16_f32");
}

extern "C" int pram_conv2d_nhwc_x3_f32(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                                       float w_scale, const float* bias, const float* scale, const float* shift,
                                       const float* residual, float* out, int cout, int ks, int stride, int relu, void* stream) {
    PRAM_REQUIRE(in && wgt_hi && wgt_lo && out, "pram_conv2d_nhwc_x3_f32: null pointer");
    PRAM_REQUIRE(ks == 1 || ks == 3, "pram_conv2d_nhwc_x3_f32: ks must be 1 or 3");
    PRAM_REQUIRE((long long)h * w * cin < (1ll << 31), "pram_conv2d_nhwc_x3_f32: an image of %d x %d x %d elements does not fit the 32-bit offsets of the im2col loader", h, w, cin);
    PRAM_REQUIRE(stride == 1 || stride == 2, "pram_conv2d_nhwc_x3_f32: stride must be 1 or 2");
    PRAM_REQUIRE(cin % 32 == 0 && w_scale > 0.f, "pram_conv2d_nhwc_x3_f32: cin=%d must be a multiple of 32", cin);
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv2d_nhwc_x3_f32: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    const int pad = ks / 2;
    ConvArgs p{in, nullptr, bias, scale, shift, residual, out, batch, h, w, cin, cout, ks, stride, relu};
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.m = batch * p.ho * p.wo;
    p.k = ks * ks * cin;
    p.status = pram_status_ptr();
    p.act_scale = pram_act_scale();
    int mi, wn;
    gemm::choose_tile(p.m, cout, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* wh = (const _Float16*)wgt_hi;
    const _Float16* wl = (const _Float16*)wgt_lo;
    const float inv = 1.0f / (pram_act_scale() * w_scale);
#define LAUNCHX3(MI_, WN_)                                                                                              \
    do {                                                                                                                \
        p.tiles_m = cdiv(p.m, gemmx3::Cfg<MI_, WN_>::BM);                                                               \
        p.tiles_n = cdiv(cout, gemmx3::Cfg<MI_, WN_>::BN);                                                              \
        hipLaunchKernelGGL((conv_x3_kernel<MI_, WN_>), dim3(p.tiles_m * p.tiles_n), dim3(gemmx3::NT), 0, st, p, wh, wl, inv); \
    } while (0)
    static const char* force = prof_env("PRAM_X3_TILE");
    const char* halo_env = getenv("PRAM_CONV_HALO");      // "0": the per-tap staging kernel for every layer (profiling / the equality test; read per call)
    // PRAM_CONV_HALO: "0" = never, "w" = the 256-channel form only (what the 128-channel form buys is measured with it)
    if (ks == 3 && stride == 1 && (cout >= 256 || (cout == 128 && !(halo_env && halo_env[0] == 'w'))) && !(force && force[0] == 'n') &&
        !(halo_env && halo_env[0] == '0') && (long long)p.ho * p.wo * cout < (1ll << 31)) {      // (its epilogue: 32-bit offsets per map)
        const int tiles_x = cdiv(p.wo, halo::TW), tiles_y = cdiv(p.ho, halo::TH);
        p.tiles_m = batch * tiles_x * tiles_y;
        const bool narrow = cout <= 128;      // one 128-channel column tile
        p.tiles_n = cdiv(cout, narrow ? 128 : 256);
        if ((long)p.tiles_m * p.tiles_n >= 224) {
            static bool hattr = false;
            if (!hattr) {
                (void)hipFuncSetAttribute((const void*)conv3x3_x3h_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(halo::Smem<4>));
                (void)hipFuncSetAttribute((const void*)conv3x3_x3h_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(halo::Smem<2>));
                hattr = true;
            }
            if (narrow)
                hipLaunchKernelGGL(conv3x3_x3h_kernel<2>, dim3(p.tiles_m * p.tiles_n), dim3(halo::NT), sizeof(halo::Smem<2>), st, p, wh, wl, inv, tiles_x, tiles_y);
            else
                hipLaunchKernelGGL(conv3x3_x3h_kernel<4>, dim3(p.tiles_m * p.tiles_n), dim3(halo::NT), sizeof(halo::Smem<4>), st, p, wh, wl, inv, tiles_x, tiles_y);
            return pram_launch_status("pram_conv2d_nhwc_x3_f32");
        }
    }
    if (cout >= 256 && (long)cdiv(p.m, 256) * cdiv(cout, 256) >= 224 && !(force && force[0] == 'n')) {
        using CW = gemmx3w::Cfg<4, 2, 4>;
        const size_t shm = sizeof(gemmx3w::Smem<4, 2, 4>);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)conv_x3w_kernel<4, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            attr_set = true;
        }
        p.tiles_m = cdiv(p.m, CW::BM);
        p.tiles_n = cdiv(cout, CW::BN);
        hipLaunchKernelGGL((conv_x3w_kernel<4, 2, 4>), dim3(p.tiles_m * p.tiles_n), dim3(CW::NT), shm, st, p, wh, wl, inv);
        return pram_launch_status("pram_conv2d_nhwc_x3_f32");
    }
    if (wn == 1) { if (mi == 2) LAUNCHX3(2, 1); else LAUNCHX3(1, 1); }
    else { if (mi == 2) LAUNCHX3(2, 2); else LAUNCHX3(1, 2); }
#undef LAUNCHX3
    return pram_launch_status("pram_conv2d_nhwc_x3_f32");
}

/* pram_conv2d_nhwc_x3_f32 followed by F.normalize over the channels of every output pixel (x / max(||x||, 1e-12)) in the same
   kernel: SFD2's descriptor head (convDb -> normalize, reference nets/sfd2.py:331-333).  cout <= 128 (a workgroup holds a pixel's
   whole channel vector), cin % 32 == 0. */
extern "C" int pram_conv2d_nhwc_x3_l2norm_f32(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                                              float w_scale, const float* bias, const float* scale, const float* shift,
                                              const float* residual, float* out, int cout, int ks, int stride, int relu, void* stream) {
    PRAM_REQUIRE(in && wgt_hi && wgt_lo && out, "pram_conv2d_nhwc_x3_l2norm_f32: null pointer");
    PRAM_REQUIRE(ks == 1 || ks == 3, "pram_conv2d_nhwc_x3_l2norm_f32: ks must be 1 or 3");
    PRAM_REQUIRE((long long)h * w * cin < (1ll << 31), "pram_conv2d_nhwc_x3_l2norm_f32: an image of %d x %d x %d elements does not fit the 32-bit offsets of the im2col loader", h, w, cin);
    PRAM_REQUIRE(stride == 1 || stride == 2, "pram_conv2d_nhwc_x3_l2norm_f32: stride must be 1 or 2");
    PRAM_REQUIRE(cin % 32 == 0 && w_scale > 0.f, "pram_conv2d_nhwc_x3_l2norm_f32: cin=%d must be a multiple of 32", cin);
    PRAM_REQUIRE(cout > 0 && cout <= 128, "pram_conv2d_nhwc_x3_l2norm_f32: cout=%d must be at most 128", cout);
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv2d_nhwc_x3_l2norm_f32: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    const int pad = ks / 2;
    ConvArgs p{in, nullptr, bias, scale, shift, residual, out, batch, h, w, cin, cout, ks, stride, relu};
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.m = batch * p.ho * p.wo;
    p.k = ks * ks * cin;
    p.status = pram_status_ptr();
    p.act_scale = pram_act_scale();
    p.tiles_m = cdiv(p.m, gemmx3::Cfg<2, 2>::BM);
    p.tiles_n = 1;
    hipLaunchKernelGGL((conv_x3_kernel<2, 2, true>), dim3(p.tiles_m), dim3(gemmx3::NT), 0, (hipStream_t)stream, p, (const _Float16*)wgt_hi,
                       (const _Float16*)wgt_lo, 1.0f / (pram_act_scale() * w_scale));
    return pram_launch_status("pram_conv2d_nhwc_x3_l2norm_f32");
}

/* pram_conv2d_nhwc_x3_f32 with the result as the split operand of the next split-fp16 layer: out_hi = fp16(16 y),
   out_lo = fp16(16 y - out_hi), [batch][ho][wo][cout] each (pram_conv3x3_grouped_planes_x3_f32 takes them).  cin % 32 == 0, even cout;
   |y| >= 4095 is reported through the range guard. */
extern "C" int pram_conv2d_nhwc_x3_planes(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                                          float w_scale, const float* bias, const float* scale, const float* shift,
                                          const float* residual, void* out_hi, void* out_lo, int cout, int ks, int stride, int relu,
                                          void* stream) {
    PRAM_REQUIRE(in && wgt_hi && wgt_lo && out_hi && out_lo, "pram_conv2d_nhwc_x3_planes: null pointer");
    PRAM_REQUIRE(ks == 1 || ks == 3, "pram_conv2d_nhwc_x3_planes: ks must be 1 or 3");
    PRAM_REQUIRE((long long)h * w * cin < (1ll << 31), "pram_conv2d_nhwc_x3_planes: an image of %d x %d x %d elements does not fit the 32-bit offsets of the im2col loader", h, w, cin);
    PRAM_REQUIRE(stride == 1 || stride == 2, "pram_conv2d_nhwc_x3_planes: stride must be 1 or 2");
    PRAM_REQUIRE(cin % 32 == 0 && w_scale > 0.f, "pram_conv2d_nhwc_x3_planes: cin=%d must be a multiple of 32", cin);
    PRAM_REQUIRE(cout > 0 && cout % 2 == 0, "pram_conv2d_nhwc_x3_planes: cout=%d must be even", cout);
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv2d_nhwc_x3_planes: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    const int pad = ks / 2;
    ConvArgs p{in, nullptr, bias, scale, shift, residual, nullptr, batch, h, w, cin, cout, ks, stride, relu};
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.m = batch * p.ho * p.wo;
    p.k = ks * ks * cin;
    p.status = pram_status_ptr();
    p.act_scale = pram_act_scale();
    p.out_hi = (_Float16*)out_hi;
    p.out_lo = (_Float16*)out_lo;
    using CW = gemmx3w::Cfg<4, 2, 4>;
    const size_t shm = sizeof(gemmx3w::Smem<4, 2, 4>);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_x3w_kernel<4, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    p.tiles_m = cdiv(p.m, CW::BM);
    p.tiles_n = cdiv(cout, CW::BN);
    hipLaunchKernelGGL((conv_x3w_kernel<4, 2, 4, true>), dim3(p.tiles_m * p.tiles_n), dim3(CW::NT), shm, (hipStream_t)stream, p,
                       (const _Float16*)wgt_hi, (const _Float16*)wgt_lo, 1.0f / (pram_act_scale() * w_scale));
    return pram_launch_status("pram_conv2d_nhwc_x3_planes");
}

extern "C" int pram_conv3x3_grouped_nhwc_f32(const float* in, int batch, int h, int w, int c, const float* wgt,
                                             const float* scale, const float* shift, float* out, int groups, int relu,
                                             void* stream) {
    PRAM_REQUIRE(in && wgt && out, "pram_conv3x3_grouped_nhwc_f32: null pointer");
    PRAM_REQUIRE(groups > 0 && c == groups * 8 && groups % 4 == 0, "pram_conv3x3_grouped_nhwc_f32: needs 8 channels per group");
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv3x3_grouped_nhwc_f32: scale and shift go together");
    if (batch == 0) return PRAM_OK;
    GConvArgs p{in, wgt, scale, shift, out, batch, h, w, c, groups, relu, cdiv(w, 64), cdiv(h, 4)};
    hipLaunchKernelGGL(gconv3x3_kernel, dim3(p.xtiles * p.ytiles * batch, groups / 4), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_conv3x3_grouped_nhwc_f32");
}

extern "C" int pram_image_to_nhwc4_f32(const float* img, float* out, int batch, int h, int w, void* stream) {
    PRAM_REQUIRE(img && out, "pram_image_to_nhwc4_f32: null pointer");
    const int total = batch * h * w;
    if (total == 0) return PRAM_OK;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, h * w, total);
    return pram_launch_status("pram_image_to_nhwc4_f32");
}

extern "C" int pram_nhwc_to_nchw_f32(const float* in, float* out, int batch, int h, int w, int c, void* stream) {
    PRAM_REQUIRE(in && out, "pram_nhwc_to_nchw_f32: null pointer");
    if (batch == 0) return PRAM_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(h * w, 32), cdiv(c, 32), batch), dim3(256), 0, (hipStream_t)stream, in,
                       out, h * w, c);
    return pram_launch_status("pram_nhwc_to_nchw_f32");
}

// ---------------------------------------------------------------- grouped 3x3 on the matrix pipe (split-fp16 path)
// gconv3x3_kernel above is bound by the LDS broadcasts of its wave-uniform weights (16 ds_read_b128 per tap for 64 packed FMAs):
// 237-268 us for 16 frames of 120 x 160 x 256 where its 630 MB need 140 us.  Here the grouped convolution is a block-diagonal
// split-fp16 product: a 32-channel block is four groups, a 16-deep k-step is one tap of two groups (8 input channels each), so a
// lane's weight operand is its group's eight weights when the step carries its group and zeros otherwise — three quarters of the
// multiplied operand are zeros, and it is still cheap: 54 MFMAs per wave for 32 pixels x 32 channels.
//
// The input arrives already split (the fp16 planes pram_conv2d_nhwc_x3_planes writes: same bytes as fp32), so a window goes from
// HBM to LDS by DMA, no register and no vector instruction in between, and a persistent workgroup (one per CU, 8 waves, a
// channel quarter = 64 channels = 8 groups, walking 8 x 16-pixel tiles) keeps the next tile's window in flight under the
// current tile's MFMAs: three window buffers of 2 x 184 pixels x 128 bytes (the 16-byte slot swizzled with the window column through the
// source addresses; a pixel outside the image is addressed past the end of the buffer, which reads as zeros).  The weights never touch LDS: a lane's column is one output
// channel, its 72 weights (two planes) stay in registers for the kernel's life and a select zeroes them for the k-steps of the
// other groups.  out[pixel][channel]: lanes are channels, a store instruction writes 128 contiguous bytes per pixel.
// fp32-class (three fp16 products, fp32 accumulation), not the fp32 FMA chain of gconv3x3_kernel: the two agree to ~3e-7
// relative (tests/test_gpu_precision_conv_fusions.py::test_grouped_conv_on_the_matrix_pipe).
namespace {
namespace gx {
constexpr int TH = 8, TW = 16, HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;      // 180 window pixels
constexpr int NT = 512, CQ = 64;                                                 // channels per workgroup (8 groups)
constexpr int ROWBLKS = (HP + 7) / 8;                                            // 1-KiB DMA pieces (8 pixels x 128 B) per plane: 23
constexpr int PLANE_B = ROWBLKS * 1024;                                          // 23 552 bytes
constexpr int BUF_B = 2 * PLANE_B;
constexpr int SMEM = 3 * BUF_B;                                                  // 141 312: one workgroup per CU
constexpr int NDMA = (2 * ROWBLKS + NT / 64 - 1) / (NT / 64);                    // DMA instructions per wave and window (6)
struct Args {
    const _Float16* in_hi; const _Float16* in_lo; float* out; const _Float16* wh; const _Float16* wl; float inv;
    const float* scale; const float* shift;
    int batch, h, wd, c, relu, tiles_x, tiles_y;
};
}  // namespace gx

__global__ __launch_bounds__(gx::NT, 1) void gconv3x3_x3_kernel(gx::Args p) {
    using namespace gx;
    using gemmx3::half8;
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_smem[];
    // workgroup L keeps channel quarter q = L % nq and walks tiles L / nq, + stride, ...; the quarters of a tile are neighbours
    // on one XCD and share the window's cache lines
    const int nq = p.c / CQ;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int q = L % nq;
    const int ntiles = p.batch * p.tiles_x * p.tiles_y, tstride = gridDim.x / nq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // Addresses.  Everything a lane adds to a tile's origin is fixed for the kernel's life, so the planes and the output are
    // addressed as raw buffers: 32-bit offset = (tile origin, uniform) + (lane constant), one vector add per window piece and
    // one per tile for the stores, no 64-bit arithmetic in the loop; and a window pixel outside the image gets an offset past
    // the buffer, which the hardware reads as zeros — the zero padding of the convolution.
    const unsigned int plane_bytes = (unsigned int)p.batch * p.h * p.wd * p.c * 2u;
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in_hi), 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in_lo), 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(2u * plane_bytes), 0x00020000);
    constexpr unsigned int PAST = 0x80000000u;      // (the host keeps the planes under 2 GiB)

    // the tile walk: (b, oy0, ox0) stepped by tstride tiles without a division
    struct Tile { int b, oy0, ox0; };
    const int adv_x = (tstride % p.tiles_x) * TW, adv_rows = tstride / p.tiles_x;
    const int adv_y = (adv_rows % p.tiles_y) * TH, adv_b = adv_rows / p.tiles_y;
    auto next = [&](Tile t) {
        t.ox0 += adv_x;
        if (t.ox0 >= p.tiles_x * TW) { t.ox0 -= p.tiles_x * TW; t.oy0 += TH; }
        t.oy0 += adv_y;
        t.b += adv_b;
        if (t.oy0 >= p.tiles_y * TH) { t.oy0 -= p.tiles_y * TH; ++t.b; }
        return t;
    };
    auto elems = [&](const Tile& t, int dy, int dx) {      // element index of pixel (oy0 + dy, ox0 + dx), channel 0; wraps for -1
        return (unsigned int)(((t.b * p.h + t.oy0 + dy) * p.wd + t.ox0 + dx) * p.c);
    };
    auto inside = [&](const Tile& t) { return t.oy0 >= 1 && t.ox0 >= 1 && t.oy0 + TH < p.h && t.ox0 + TW < p.wd; };

    // this lane's part of a window: piece i = wave + 8 k of the 2 x 23 (plane, 8-pixel row block) pieces; the lane brings
    // physical slot lane % 8 of window pixel 8 rowblk + lane / 8 = (hy, hx), i.e. logical slot (lane % 8) ^ ((hx >> 1) & 7): with the
    // window column in the swizzle the sixteen lanes of each ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}:
    // two tile rows) land on sixteen different slots of the 256-byte bank row for every tap (the pixel index did not: 2-way).  The last
    // row block's pixels 180 .. 183 do not exist: they repeat pixel 179 into LDS rows nobody reads.
    unsigned int woff[NDMA];
    int hyx[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int i = wave + (NT / 64) * k;
        const int rowblk = i % ROWBLKS;
        const int hp = min(8 * rowblk + (lane >> 3), HP - 1);
        const int hy = hp / HWD, hx = hp - hy * HWD;
        hyx[k] = (hy << 8) | hx;
        woff[k] = (unsigned int)(((hy * p.wd + hx) * p.c + CQ * q) * 2 + (((lane & 7) ^ ((hx >> 1) & 7)) << 4));
    }
    auto fetch = [&](const Tile& t, int buf) {
        const unsigned int org = elems(t, -1, -1) * 2u;
        const bool in = inside(t);                   // workgroup-uniform: no pixel of the window needs the zero padding
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int i = wave + (NT / 64) * k;      // wave-uniform
            if (i < 2 * ROWBLKS) {
                const int plane = i / ROWBLKS, rowblk = i - plane * ROWBLKS;
                unsigned int off = org + woff[k];
                if (!in) {
                    const int iy = t.oy0 - 1 + (hyx[k] >> 8), ix = t.ox0 - 1 + (hyx[k] & 0xff);
                    if (!((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd)) off = PAST;
                }
                auto* dst = (__attribute__((address_space(3))) void*)(gx_smem + buf * BUF_B + plane * PLANE_B + rowblk * 1024);
                if (plane) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lo, dst, 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hi, dst, 16, off, 0, 0, 0);
            }
        }
    };

    // wave = (32-pixel block pbk, 32-channel block nb); out[pixel][channel] += window[pixel + tap][group's 8 channels] . w
    const int nb = wave & 1, pbk = wave >> 1;
    const int tr = 32 * pbk + r, py = tr >> 4, px = tr & 15;             // this lane's pixel row of the A operand
    const int gn = r >> 3;                                               // this lane's channel column of the B operand: group 4 nb + gn
    const int ch = CQ * q + 32 * nb + r;
    const float sci = (p.scale ? p.scale[ch] : 1.f) * p.inv, sh = p.scale ? p.shift[ch] : 0.f;
    const half8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    // the column's weights [tap][ci 8], both planes: registers for the kernel's life.  k-step (tap, s) carries groups 2 s and
    // 2 s + 1 of the block in the lane halves h = 0 / 1: the lane's operand is its weights when gn == 2 s + h, zeros otherwise
    half8 wh[9], wl[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        wh[tap] = *reinterpret_cast<const half8*>(p.wh + (size_t)ch * 72 + tap * 8);
        wl[tap] = *reinterpret_cast<const half8*>(p.wl + (size_t)ch * 72 + tap * 8);
    }
    const bool sel0 = gn == h, sel1 = gn == 2 + h;

    // Three window buffers: tile i's MFMAs read buffer i % 3 while the windows of tiles i + 1 and i + 2 are in flight or landed.
    // Tile i - 1's results stay in the accumulator registers until the barrier of round i is behind, and are stored in front of
    // the DMA issue of window i + 2 — so the wait on top of a round has, in issue order, window i + 1, stores, window i + 2 in
    // flight, and s_waitcnt vmcnt(pieces of i + 2) is enough for window i + 1 however stores and loads overtake each other:
    // loads return in order, so at most that many operations left means no load of window i + 1 is left.
    f32x16 g;
    // accumulator element e is pixel (2 pbk + (e >> 3), 4 h + (e & 3) + 8 ((e >> 2) & 1)) of the tile, channel ch
    const unsigned int soff = (unsigned int)(((2 * pbk * p.wd + 4 * h) * p.c + ch) * 4);
    auto store = [&](const Tile& t) {      // BN -> ReLU -> store
        const unsigned int off = elems(t, 0, 0) * 4u + soff;
        const bool whole = t.oy0 + TH <= p.h && t.ox0 + TW <= p.wd;      // workgroup-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dy = e >> 3, dx = (e & 3) + 8 * ((e >> 2) & 1);
            float o = g[e] * sci + sh;
            if (p.relu) o = fmaxf(o, 0.f);
            const unsigned int eoff = (unsigned int)((dy * p.wd + dx) * p.c * 4);      // uniform: a scalar offset of the store
            if (whole || (t.oy0 + 2 * pbk + dy < p.h && t.ox0 + 4 * h + dx < p.wd))
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, o), rs_out, off, eoff, 0);
        }
    };
    Tile c0, c1, c2, prev{0, 0, 0};
    {
        int t = L / nq;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        c0 = Tile{t / p.tiles_y, (t % p.tiles_y) * TH, tx * TW};
    }
    c1 = next(c0);
    c2 = next(c1);
    const int t0 = L / nq;
    if (t0 < ntiles) fetch(c0, 0);
    if (t0 + tstride < ntiles) fetch(c1, 1);
    int buf = 0;
    bool have = false;
    for (int t = t0; t < ntiles; t += tstride) {
        if (t + tstride < ntiles) {      // window t + tstride may stay in flight: the pieces of it this wave issued
            if (wave < 2 * ROWBLKS - (NT / 64) * (NDMA - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                          // everyone's pieces have landed; everyone is done with buffer (buf + 2) % 3
        if (have) store(prev);
        if (t + 2 * tstride < ntiles) fetch(c2, buf == 0 ? 2 : buf - 1);
        const unsigned char* yh = gx_smem + buf * BUF_B;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
        // a tap's four fragments (two k-steps x two planes) are read one tap ahead of the MFMAs that take them; in step s lane
        // half h reads the eight input channels of group 2 s + h of its block: logical slot 4 nb + 2 s + h
        half8 a[2][4];
        auto rd = [&](int tap, half8 (&f)[4]) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hp = (py + ky) * HWD + px + kx;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = hp * 128 + (((4 * nb + 2 * s + h) ^ (((px + kx) >> 1) & 7)) << 4);
                f[2 * s] = *reinterpret_cast<const half8*>(yh + off);
                f[2 * s + 1] = *reinterpret_cast<const half8*>(yh + PLANE_B + off);
            }
        };
        rd(0, a[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) rd(tap + 1, a[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler would sink the reads back to their MFMAs)
            const half8 (&f)[4] = a[tap & 1];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool mine = s ? sel1 : sel0;
                const half8 bh = mine ? wh[tap] : zero8, bl = mine ? wl[tap] : zero8;
                g = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s + 1], bh, g, 0, 0, 0);
                g = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s], bl, g, 0, 0, 0);
                g = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * s], bh, g, 0, 0, 0);
            }
        }
        prev = c0;
        have = true;
        c0 = c1;
        c1 = c2;
        c2 = next(c2);
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (have) store(prev);
}
}  // namespace

/* The grouped 3x3 of the ResBlock (8 channels per group) on the split-fp16 path, as a block-diagonal product on the matrix pipe.
   in_hi / in_lo: the input * 16 as fp16 planes [batch][h][w][c] (pram_conv2d_nhwc_x3_planes writes them); w_hi / w_lo: the
   [c][3][3][8] weights * w_scale as fp16 planes; c % 64 == 0.  fp32-class results. */
extern "C" int pram_conv3x3_grouped_planes_x3_f32(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, const void* w_hi,
                                                  const void* w_lo, float w_scale, const float* scale, const float* shift, float* out,
                                                  int groups, int relu, void* stream) {
    PRAM_REQUIRE(in_hi && in_lo && w_hi && w_lo && out, "pram_conv3x3_grouped_planes_x3_f32: null pointer");
    PRAM_REQUIRE(groups > 0 && c == groups * 8 && c % gx::CQ == 0 && w_scale > 0.f, "pram_conv3x3_grouped_planes_x3_f32: needs 8 channels per group, c %% 64 == 0");
    PRAM_REQUIRE((scale == nullptr) == (shift == nullptr), "pram_conv3x3_grouped_planes_x3_f32: scale and shift go together");
    PRAM_REQUIRE((size_t)batch * h * w * c < ((size_t)1 << 29), "pram_conv3x3_grouped_planes_x3_f32: at most 2^29 elements (32-bit buffer offsets)");
    if (batch == 0) return PRAM_OK;
    gx::Args p{(const _Float16*)in_hi, (const _Float16*)in_lo, out, (const _Float16*)w_hi, (const _Float16*)w_lo,
               1.0f / (pram_act_scale() * w_scale), scale, shift, batch, h, w, c, relu, cdiv(w, gx::TW), cdiv(h, gx::TH)};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gconv3x3_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gx::SMEM);
        attr = true;
    }
    const int nq = c / gx::CQ, ntiles = batch * p.tiles_x * p.tiles_y;
    const int per_q = min(ntiles, max(1, pram_cu_count() / nq));      // one resident workgroup per CU, each a (quarter, tile walk)
    hipLaunchKernelGGL(gconv3x3_x3_kernel, dim3(per_q * nq), dim3(gx::NT), gx::SMEM, (hipStream_t)stream, p);
    return pram_launch_status("pram_conv3x3_grouped_planes_x3_f32");
}
