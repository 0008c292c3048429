// SFD2 ResBlock (nets/sfd2.py:107-124: 1x1 -> BN -> ReLU -> 3x3 groups = 32 -> BN -> ReLU -> 1x1 -> BN -> + identity -> ReLU, 256
// channels) as ONE kernel on the split-fp16 path.
//
// As three kernels (conv_x3w_kernel, gconv3x3_kernel, conv_x3w_kernel) a block reads or writes its 315 MB feature map seven times
// (16 frames of 120 x 160 x 256 fp32): every one of them is bound by that traffic, not by its arithmetic — the 1x1 convolutions run
// at 2.6 TB/s with the matrix pipe 27 % busy, the grouped 3x3 at 2.7 TB/s.  Here a workgroup owns 8 x 16 output pixels and all 256
// channels and the two intermediate maps never leave the CU: x is read once (plus the one-pixel halo and the residual, L2 hits),
// the result written once.
//
//   phase 1   y1 = ReLU(BN1(x W1^T)) on the (8 + 2) x (16 + 2) pixel window (180 rows padded to 192), 128 output channels at a time
//             ("half"): K = 256 in eight 32-deep chunks, x split while it is staged (as conv3x3_x3h_kernel stages its window), W1 by
//             LDS-DMA; 48 accumulator registers per lane.  The window costs 1.5x the rows of the tile: 1.25x the MFMAs of the two
//             1x1 convolutions overall.
//   phase 2   per 64 channels ("quarter" = 8 groups): y1 -> LDS as fp32 (zeros outside the image: the 3x3's padding), grouped 3x3 on
//             the vector ALU (one group per wave, two vertically adjacent pixels per lane, the arithmetic of gconv3x3_kernel in its
//             order), y2 = ReLU(BN2(.)) split into the A planes of the second 1x1, whose 64-deep K slice is multiplied at once
//             (W3 slice by LDS-DMA): out[128 x 256] accumulates in 64 registers per lane across the four quarters.
//   epilogue  BN3 -> + x -> ReLU -> store.
// Every accumulator sees its products in the order the three kernels use (k ascending, lo.hi, hi.lo, hi.hi per 16-deep step; taps
// and input-channel pairs ascending in the grouped convolution) and every intermediate is rounded to fp32 where they round it:
// the result is bit-identical to the three-kernel path (tests/test_gpu_round4.py::test_fused_resblock_equals_three_kernels).
#include <stdlib.h>
#include "gemm_core.h"
#include "gemm_core_x3.h"
#include "gemm_core_x3w.h"

namespace {
namespace rb {

using gemmx3::half4;
using gemmx3::half8;
using gemmx3::swz;

constexpr int TH = 8, TW = 16, HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;      // 180 window pixels
constexpr int AR = 192;                                                          // window rows of the MFMA tile (six 32-row blocks)
constexpr int C = 256, BK = 32, NT = 512;
constexpr int NL = (HP * 8 + NT - 1) / NT;                                       // float4 loads per thread and chunk of the window (3)
constexpr int PXS = 68;                                                          // floats per y1 pixel in LDS (64 + 4: lanes on distinct banks)

// LDS map (bytes).  Region A: phase 1 = two A stages (hi | lo planes of 192 x 32); phase 2 = y1 of the quarter (180 x 68 fp32), then
// the second 32-deep W3 chunk of the quarter.  Region B: phase 1 = two stages of W1 (128 x 32, hi | lo); phase 2 = the first W3
// chunk (256 x 32, hi | lo) and the y2 planes (two chunks of 128 x 32, hi | lo).  Region C: the quarter's grouped-3x3 weights.
constexpr int A_STAGE = AR * BK * 2;                      // bytes of one plane of one A stage (12 288)
constexpr int OFF_A = 0, SZ_A = 4 * A_STAGE;              // 49 152  (>= HP * PXS * 4 = 48 960, >= 32 768)
constexpr int OFF_B = OFF_A + SZ_A, SZ_B = 65536;
constexpr int OFF_C = OFF_B + SZ_B, SZ_C = 8 * 576 * 4;   // 18 432
constexpr int SMEM_BYTES = OFF_C + SZ_C;                  // 133 120
static_assert(HP * PXS * 4 <= SZ_A, "y1 of a quarter must fit region A");

struct Args {
    const float* in; float* out;
    const _Float16* w1h; const _Float16* w1l; float inv1;
    const float* s1; const float* t1;
    const float* w2; const float* s2; const float* t2;
    const _Float16* w3h; const _Float16* w3l; float inv3;
    const float* s3; const float* t3;
    int batch, h, wd, tiles_x, tiles_y;
    unsigned int* status;
    int abl;      // profiling only (PRAM_RB_ABLATE): 1 = no grouped-3x3 taps, 2 = one K chunk in phase 1, 4 = no second-1x1 k-steps
};

}  // namespace rb

__global__ __launch_bounds__(rb::NT, 1) void resblock_x3_kernel(rb::Args p) {
    using namespace rb;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int nblk = p.batch * p.tiles_x * p.tiles_y;
    int t = xcd_remap(blockIdx.x, nblk);
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const float* img = p.in + (size_t)b * p.h * p.wd * C;

    // ---- LDS views
    auto H16 = [&](int off) { return reinterpret_cast<_Float16*>(smem_raw + off); };
    auto a_h = [&](int s) { return H16(OFF_A + 2 * A_STAGE * s); };
    auto a_l = [&](int s) { return H16(OFF_A + 2 * A_STAGE * s + A_STAGE); };
    float* y1q = reinterpret_cast<float*>(smem_raw + OFF_A);
    _Float16* w3c1_h = H16(OFF_A);                                                   // second W3 chunk of a quarter (after its grouped 3x3)
    _Float16* w3c1_l = H16(OFF_A + 16384);
    auto b1_h = [&](int s) { return H16(OFF_B + 16384 * s); };
    auto b1_l = [&](int s) { return H16(OFF_B + 16384 * s + 8192); };
    _Float16* w3c0_h = H16(OFF_B);
    _Float16* w3c0_l = H16(OFF_B + 16384);
    auto y2_h = [&](int c) { return H16(OFF_B + 32768 + 8192 * c); };
    auto y2_l = [&](int c) { return H16(OFF_B + 49152 + 8192 * c); };
    float* w2q = reinterpret_cast<float*>(smem_raw + OFF_C);

    float amax = 0.f;      // range guard: largest |value * 16| this lane split

    // ---- window staging (phase 1): element e = tid + NT j is float4 q = e % 8 of window pixel hp = e / 8
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
        g.goff = (unsigned)((iyc * p.wd + ixc) * C + g.q * 4);
        return g;
    };
    float4 hv[NL];
    auto hload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NL; ++j) hv[j] = *reinterpret_cast<const float4*>(img + geo(j).goff + kt * BK);
    };
    auto hcommit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const Geo g = geo(j);
            float4 v = hv[j];
            if (!g.inimg) v = make_float4(0.f, 0.f, 0.f, 0.f);
            half4 hi, lo;
            gemmx3::split4(v, gemmx3::ACT_SCALE, hi, lo, amax);
            if (g.valid) {
                const int off = g.hp * BK + swz(g.q >> 1, g.hp) * 8 + (g.q & 1) * 4;
                *reinterpret_cast<half4*>(&a_h(buf)[off]) = hi;
                *reinterpret_cast<half4*>(&a_l(buf)[off]) = lo;
            }
        }
    };
    // rows HP .. AR - 1 of both A stages are never written: zeros (their outputs are never used; NaN-free for the tidy mind)
    auto zero_pad_rows = [&]() {
        for (int i = tid; i < (AR - HP) * BK / 8; i += NT) {
            const int off = HP * BK + i * 8;
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                *reinterpret_cast<uint4*>(&a_h(s)[off]) = z;
                *reinterpret_cast<uint4*>(&a_l(s)[off]) = z;
            }
        }
    };

    // ---- output accumulators of the second 1x1: rows 64 wm + 32 mi + .., columns 64 wn + 32 ni + r
    f32x16 acc3[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc3[mi][ni][e] = 0.f;

#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
        // ================================================================ phase 1: y1 (this half's 128 channels) on the window
        // accumulators: rows 96 wm + 32 mi + .., column 128 hh + 32 wn + r
        f32x16 acc1[3];
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[mi][e] = 0.f;
        auto b1dma = [&](int buf, int kt) {
            auto bp = [&](int row, int plane) { return (plane ? p.w1l : p.w1h) + (size_t)(128 * hh + row) * C + kt * BK; };
            gemmx3w::dma_tile<128, NT / 64>(b1_h(buf), b1_l(buf), bp);
        };
        auto kstep1 = [&](int cur, int ks) {
            const int slot = swz(2 * ks + h, r) * 8;      // rows differ from r by multiples of 32: same swizzle
            const int brow0 = (wn * 32 + r) * BK, arow0 = (wm * 96 + r) * BK;
            const half8 bh = *reinterpret_cast<const half8*>(&b1_h(cur)[brow0 + slot]);
            const half8 bl = *reinterpret_cast<const half8*>(&b1_l(cur)[brow0 + slot]);
            half8 ah[3], al[3];
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                ah[mi] = *reinterpret_cast<const half8*>(&a_h(cur)[arow0 + mi * 32 * BK + slot]);
                al[mi] = *reinterpret_cast<const half8*>(&a_l(cur)[arow0 + mi * 32 * BK + slot]);
            }
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh, acc1[mi], 0, 0, 0);
                acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl, acc1[mi], 0, 0, 0);
                acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh, acc1[mi], 0, 0, 0);
            }
        };
        __syncthreads();                    // everybody is done with regions A and B (the previous half's last quarter)
        zero_pad_rows();
        hload(0);
        b1dma(0, 0);
        hcommit(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nk1 = (p.abl & 2) ? 1 : C / BK;
#pragma unroll 1
        for (int kt = 0; kt < nk1; ++kt) {
            const bool more = kt + 1 < nk1;
            if (more) { b1dma((kt + 1) & 1, kt + 1); hload(kt + 1); }
            __builtin_amdgcn_sched_barrier(0);
            kstep1(kt & 1, 0);
            kstep1(kt & 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) hcommit((kt + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // BN1 -> ReLU in place (the values the first 1x1 convolution would have written)
        {
            const int c1 = 128 * hh + 32 * wn + r;
            const float sc = p.s1[c1], sh = p.t1[c1];
#pragma unroll
            for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc1[mi][e] * p.inv1;
                    v = v + 0.f;                      // conv_epilogue adds its (absent) bias
                    v = v * sc + sh;
                    acc1[mi][e] = fmaxf(v, 0.f);
                }
        }

        // ================================================================ phase 2: the half's two quarters
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {
            const int q = 2 * hh + j;                 // channels 64 q .. 64 q + 63 = groups 8 q .. 8 q + 7
            // (the barrier that ended phase 1 / the previous quarter's last k-steps is behind us: regions A and B are free)
            auto w3dma = [&](_Float16* dh, _Float16* dl, int c) {
                auto bp = [&](int row, int plane) { return (plane ? p.w3l : p.w3h) + (size_t)row * C + 64 * q + 32 * c; };
                gemmx3w::dma_tile<C, NT / 64>(dh, dl, bp);
            };
            w3dma(w3c0_h, w3c0_l, 0);
            // grouped-3x3 weights of the quarter's eight groups: 8 x 576 floats, contiguous in [group][co][tap][ci]
            {
                const float* src = p.w2 + (size_t)(8 * q) * 576;
                float4 wv[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) wv[i] = *reinterpret_cast<const float4*>(src + min((tid + NT * i) * 4, 8 * 576 - 4));
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if ((tid + NT * i) * 4 < 8 * 576) *reinterpret_cast<float4*>(w2q + (tid + NT * i) * 4) = wv[i];
            }
            // y1 of the quarter -> LDS, zeros outside the image (the padding of the 3x3); written by the waves that hold its columns
            if ((wn >> 1) == j) {
                const int ch = 32 * (wn & 1) + r;
#pragma unroll
                for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int hp = 96 * wm + gemm::acc_row(mi, e, h);
                        if (hp < HP) {
                            const int hy = hp / HWD, hx = hp - hy * HWD;
                            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                            const bool in = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
                            y1q[hp * PXS + ch] = in ? acc1[mi][e] : 0.f;
                        }
                    }
            }
            __syncthreads();
            // ---- grouped 3x3: wave = group 8 q + wave, lane = column x and row pair (2 yp, 2 yp + 1); gconv3x3_kernel's arithmetic
            {
                const int x = lane & 15, yp = lane >> 4;
                f2 g2[2][8];
#pragma unroll
                for (int ro = 0; ro < 2; ++ro)
#pragma unroll
                    for (int o = 0; o < 8; ++o) g2[ro][o] = (f2){0.f, 0.f};
#pragma unroll 1
                for (int tap = (p.abl & 1) ? 9 : 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    // four output channels' weights at a time (32 registers: the block's 112 accumulator registers leave no room for
                    // all eight); every accumulator still sees its taps and channel pairs in gconv3x3_kernel's order
                    const float* ws = w2q + wave * 576 + tap * 8;
#pragma unroll
                    for (int oh = 0; oh < 2; ++oh) {
                        f2 wt[4][4];
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const float4 w0 = *reinterpret_cast<const float4*>(ws + (4 * oh + o) * 72);
                            const float4 w1 = *reinterpret_cast<const float4*>(ws + (4 * oh + o) * 72 + 4);
                            wt[o][0] = (f2){w0.x, w0.y}; wt[o][1] = (f2){w0.z, w0.w};
                            wt[o][2] = (f2){w1.x, w1.y}; wt[o][3] = (f2){w1.z, w1.w};
                        }
#pragma unroll
                        for (int ro = 0; ro < 2; ++ro) {
                            const float* src = y1q + ((2 * yp + ro + ky) * HWD + x + kx) * PXS + wave * 8;
                            const float4 a0 = *reinterpret_cast<const float4*>(src);
                            const float4 a1 = *reinterpret_cast<const float4*>(src + 4);
                            const f2 xp[4] = {(f2){a0.x, a0.y}, (f2){a0.z, a0.w}, (f2){a1.x, a1.y}, (f2){a1.z, a1.w}};
#pragma unroll
                            for (int o = 0; o < 4; ++o)
#pragma unroll
                                for (int ip = 0; ip < 4; ++ip)
                                    g2[ro][4 * oh + o] = __builtin_elementwise_fma(xp[ip], wt[o][ip], g2[ro][4 * oh + o]);
                        }
                    }
                }
                // BN2 -> ReLU -> split planes of the second 1x1's A operand: row = tile pixel, k = 8 wave + o inside the quarter
                float sc[8], sh[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) { sc[o] = p.s2[64 * q + 8 * wave + o]; sh[o] = p.t2[64 * q + 8 * wave + o]; }
                const int kc = wave >> 2, slot = wave & 3;
#pragma unroll
                for (int ro = 0; ro < 2; ++ro) {
                    float res[8];
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        float v = g2[ro][o][0] + g2[ro][o][1];
                        v = v * sc[o] + sh[o];
                        res[o] = fmaxf(v, 0.f);
                    }
                    half4 h0, l0, h1, l1;
                    gemmx3::split4(make_float4(res[0], res[1], res[2], res[3]), gemmx3::ACT_SCALE, h0, l0, amax);
                    gemmx3::split4(make_float4(res[4], res[5], res[6], res[7]), gemmx3::ACT_SCALE, h1, l1, amax);
                    const int tr = (2 * yp + ro) * TW + x;
                    const int off = tr * BK + swz(slot, tr) * 8;
                    const half8 hv8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const half8 lv8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    *reinterpret_cast<half8*>(&y2_h(kc)[off]) = hv8;
                    *reinterpret_cast<half8*>(&y2_l(kc)[off]) = lv8;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first W3 chunk has landed
            __syncthreads();                                      // y2 planes and W3 chunk 0 visible; y1 of the quarter is dead
            w3dma(w3c1_h, w3c1_l, 1);                             // second chunk into region A, under the first chunk's MFMAs
            auto kstep3 = [&](const _Float16* ya_h, const _Float16* ya_l, const _Float16* wb_h, const _Float16* wb_l, int ks) {
                const int slot2 = swz(2 * ks + h, r) * 8;
                const int brow0 = (wn * 64 + r) * BK, arow0 = (wm * 64 + r) * BK;
                half8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    bh[ni] = *reinterpret_cast<const half8*>(&wb_h[brow0 + ni * 32 * BK + slot2]);
                    bl[ni] = *reinterpret_cast<const half8*>(&wb_l[brow0 + ni * 32 * BK + slot2]);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    ah[mi] = *reinterpret_cast<const half8*>(&ya_h[arow0 + mi * 32 * BK + slot2]);
                    al[mi] = *reinterpret_cast<const half8*>(&ya_l[arow0 + mi * 32 * BK + slot2]);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc3[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc3[mi][ni], 0, 0, 0);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc3[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc3[mi][ni], 0, 0, 0);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc3[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc3[mi][ni], 0, 0, 0);
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.abl & 4)) {
            kstep3(y2_h(0), y2_l(0), w3c0_h, w3c0_l, 0);
            kstep3(y2_h(0), y2_l(0), w3c0_h, w3c0_l, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!(p.abl & 4)) {
            kstep3(y2_h(1), y2_l(1), w3c1_h, w3c1_l, 0);
            kstep3(y2_h(1), y2_l(1), w3c1_h, w3c1_l, 1);
            }
            __syncthreads();                                      // regions A and B free for the next quarter / half
        }
    }
    x3_range_flag(p.status, amax);

    // ================================================================ epilogue: BN3 -> + x -> ReLU
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wn + 32 * ni + r;
        const float sc = p.s3[col], sh = p.t3[col];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float res[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tr = 64 * wm + gemm::acc_row(mi, e, h);
                const int oy = min(oy0 + (tr >> 4), p.h - 1), ox = min(ox0 + (tr & 15), p.wd - 1);
                res[e] = img[((size_t)oy * p.wd + ox) * C + col];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tr = 64 * wm + gemm::acc_row(mi, e, h);
                const int oy = oy0 + (tr >> 4), ox = ox0 + (tr & 15);
                float v = acc3[mi][ni][e] * p.inv3;
                v = v + 0.f;
                v = v * sc + sh;
                v += res[e];
                v = fmaxf(v, 0.f);
                if (oy < p.h && ox < p.wd) p.out[(((size_t)b * p.h + oy) * p.wd + ox) * C + col] = v;
            }
        }
    }
}

}  // namespace

/* One ResBlock of SFD2's conv4 (nets/sfd2.py:107-124) on the split-fp16 path, fused: out = ReLU(BN3(conv1x1(ReLU(BN2(gconv3x3(
   ReLU(BN1(conv1x1(in)))))))) + in), 256 channels, 32 groups of 8, NHWC fp32.  w1 / w3: [256][256] weight planes * w?_scale (hi, lo
   as pram_conv2d_nhwc_x3_f32 takes them), w2: fp32 [32][8][3][3][8] (pram_conv3x3_grouped_nhwc_f32's layout); s? / t?: eval-mode
   BatchNorm as per-channel scale / shift.  Bit-identical to pram_conv2d_nhwc_x3_f32 (ks = 1) -> pram_conv3x3_grouped_nhwc_f32 ->
   pram_conv2d_nhwc_x3_f32 (ks = 1, residual).  `out` must not alias `in`. */
extern "C" int pram_resblock_nhwc_x3_f32(const float* in, int batch, int h, int w, const void* w1_hi, const void* w1_lo, float w1_scale,
                                         const float* s1, const float* t1, const float* w2, const float* s2, const float* t2,
                                         const void* w3_hi, const void* w3_lo, float w3_scale, const float* s3, const float* t3,
                                         float* out, void* stream) {
    PRAM_REQUIRE(in && out && w1_hi && w1_lo && w2 && w3_hi && w3_lo && s1 && t1 && s2 && t2 && s3 && t3, "pram_resblock_nhwc_x3_f32: null pointer");
    PRAM_REQUIRE(in != out, "pram_resblock_nhwc_x3_f32: out must not alias in (the halo and the residual are read while tiles are written)");
    PRAM_REQUIRE(batch >= 0 && h > 0 && w > 0 && w1_scale > 0.f && w3_scale > 0.f, "pram_resblock_nhwc_x3_f32: bad sizes");
    PRAM_REQUIRE((long long)h * w * rb::C < (1ll << 32), "pram_resblock_nhwc_x3_f32: a frame must hold fewer than 2^32 floats");
    if (batch == 0) return PRAM_OK;
    rb::Args p{in, out, (const _Float16*)w1_hi, (const _Float16*)w1_lo, 1.0f / (gemmx3::ACT_SCALE * w1_scale), s1, t1, w2, s2, t2,
               (const _Float16*)w3_hi, (const _Float16*)w3_lo, 1.0f / (gemmx3::ACT_SCALE * w3_scale), s3, t3,
               batch, h, w, cdiv(w, rb::TW), cdiv(h, rb::TH), pram_status_ptr(), 0};
    { const char* e = getenv("PRAM_RB_ABLATE"); p.abl = e ? atoi(e) : 0; }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)resblock_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rb::SMEM_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(resblock_x3_kernel, dim3(batch * p.tiles_x * p.tiles_y), dim3(rb::NT), rb::SMEM_BYTES, (hipStream_t)stream, p);
    return pram_launch_status("pram_resblock_nhwc_x3_f32");
}
