// SFD2's first two convolutions (nets/sfd2.py:135-139,281-282: conv1a 3 -> 64, 3x3, stride 1; conv1b 64 -> 64, 3x3, stride 2; each
// bias -> BN -> ReLU) as ONE kernel on the split-fp16 path.
//
// As two kernels conv1a writes a 1.26 GB map (16 frames of 480 x 640 x 64 fp32) at 2.9 TB/s and conv1b fetches 2.1 GB to read it
// back: the largest round trip of the step (5.4 % of its HBM traffic) for 3 % of its arithmetic.  Here a workgroup owns 8 x 16
// conv1b outputs; the (2 * 8 + 1) x (2 * 16 + 1) window of conv1a outputs it needs is computed from the image window in LDS, 32
// channels at a time ("half"), written to LDS as split planes and consumed by conv1b's nine taps in place; only the image is read
// (20 MB per step) and the 240 x 320 x 64 result written.
//   conv1a of a half: out^T[channel][pixel] on the matrix pipe, K = 9 taps x 4 channels (the image is NHWC4) padded to 48; the
//       window's 561 pixels are 18 blocks of 32 lanes, three or two per wave; bias -> BN -> ReLU and conv1b's zero padding (window
//       pixels outside the 480 x 640 map) per lane, then 8-byte plane writes.  1.16x the pixels of the tile's footprint.
//   conv1b of a half: the half's nine [64 x 32] weight tiles (72 KB as planes, LDS-DMA, requested before the half's conv1a) and the
//       window planes (70 KB) are both resident: 54 MFMAs per wave between two barriers, accumulators kept across the halves.
// Arithmetic: conv1a runs here as split-fp16 products like every other layer (the stand-alone conv1a is the exact-fp32 MFMA kernel
// only because its 4-channel input does not fit the generic split-fp16 tiles), conv1b's K order is (half, tap, channel) — the order
// of conv_x3_kernel.  fp32-class results, not bit-identical to the two-kernel path (tests/test_gpu_precision_conv_fusions.py::test_fused_conv1).
#include <stdlib.h>
#include "gemm_core.h"
#include "gemm_core_x3.h"
#include "gemm_core_x3w.h"

namespace {
namespace c1 {

using gemmx3::half4;
using gemmx3::half8;
using gemmx3::swz;

constexpr int TH = 8, TW = 16;                                   // conv1b outputs per workgroup
constexpr int WH = 2 * TH + 1, WW = 2 * TW + 1, WP = WH * WW;    // conv1a window: 17 x 33 = 561 pixels
constexpr int IH = WH + 2, IW = WW + 2, IP = IH * IW;            // image window: 19 x 35 = 665 pixels
constexpr int NT = 512, NBLK = (WP + 31) / 32;                   // 18 pixel blocks of conv1a
constexpr int KA = 48;                                           // conv1a's K: 9 taps x 4 channels = 36, padded to three 16-deep steps
constexpr int Y_PLANE = WH * 36 * 32 * 2;                         // 39 168 bytes: 17 window rows of 36 pixel rows (c1::RW; columns permuted in eights)
constexpr int W_PLANE = 9 * 64 * 32 * 2;                         // 36 864
constexpr int I_PLANE = IP * 4 * 2;                              // 5 320
constexpr int OFF_Y = 0, OFF_W = OFF_Y + 2 * Y_PLANE, OFF_I = OFF_W + 2 * W_PLANE;
constexpr int SMEM_BYTES = OFF_I + 2 * I_PLANE;                  // 162 704 of the CU's 163 840
static_assert(Y_PLANE % 16 == 0 && W_PLANE % 16 == 0, "16-byte aligned regions");

struct Args {
    const float* img; float* out;                  // image: NHWC4 [b][h][w][4], or (nchw3) the reference's NCHW [b][3][h][w]; out [b][ho][wo][64]
    const _Float16* wah; const _Float16* wal; float inva;      // conv1a weight planes [64][KA] * scale
    const float* ba; const float* sa; const float* ta;
    const _Float16* wbh; const _Float16* wbl; float invb;      // conv1b weight planes [64][3][3][64] * scale
    const float* bb; const float* sb; const float* tb;
    int batch, h, wd, ho, wo, tiles_x, tiles_y;
    unsigned int* status;
    int abl;      // profiling only (PRAM_C1_ABLATE): 1 = no conv1a blocks, 2 = no conv1b taps, 4 = no weight DMA after the first
    int nchw3;
    float act_scale = gemmx3::ACT_SCALE;      // scale of the activation planes (pram_act_scale() at launch)
};

__device__ __forceinline__ int rowoff(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Window planes: 64-byte rows (32 channels of a pixel), window rows RW pixel rows apart.  conv1b reads them with a pixel stride of
// 2, a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} — the sixteen output columns of the tile,
// split over two of its rows — and a group is conflict-free when its lanes hit sixteen different (row mod 4, slot) positions of the
// 256-byte bank row.  Both are taken from the window COLUMN x = 2 ox + kx: the physical row is x with its low three bits rotated
// (bit 0 -> bit 2: row mod 4 = (x >> 1) & 3, RW % 4 == 0 keeps that through the row pitch) and the slot is ^ (x >> 3) & 3, so the
// sixteen consecutive x >> 1 of a group give the sixteen positions whatever rows they sit in.  (Round 4's first layout swizzled with
// the pixel index y * 33 + x: 2-way, SQ_LDS_BANK_CONFLICT 2.56 per LDS instruction, profiles/r04_pmc_summary.md.)
constexpr int RW = 36;
__device__ __forceinline__ int yoff(int y, int x, int slot) {
    const int prow = y * RW + ((x & ~7) | ((x >> 1) & 3) | ((x & 1) << 2));
    return prow * 32 + ((slot ^ ((x >> 3) & 3)) << 3);
}

}  // namespace c1

__global__ __launch_bounds__(c1::NT, 1) void conv1ab_x3_kernel(c1::Args p) {
    using namespace c1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    auto H16 = [&](int off) { return reinterpret_cast<_Float16*>(smem_raw + off); };
    _Float16* y_h = H16(OFF_Y);
    _Float16* y_l = H16(OFF_Y + Y_PLANE);
    _Float16* w_h = H16(OFF_W);
    _Float16* w_l = H16(OFF_W + W_PLANE);
    _Float16* i_h = H16(OFF_I);
    _Float16* i_l = H16(OFF_I + I_PLANE);

    const int nblk = p.batch * p.tiles_x * p.tiles_y;
    int t = xcd_remap(blockIdx.x, nblk);
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int ay0 = 2 * oy0 - 1, ax0 = 2 * ox0 - 1;          // conv1a coordinates of window pixel (0, 0)   (conv1b: stride 2, pad 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    float amax = 0.f;

    // conv1b's weight tiles of a half: nine [64 cout x 32 k] tiles per plane in the swizzled 64-byte rows the fragment reads expect
    auto wdma = [&](int half) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            auto bp = [&](int row, int plane) { return (plane ? p.wbl : p.wbh) + (size_t)row * 576 + tap * 64 + 32 * half; };
            gemmx3w::dma_tile<64, NT / 64>(w_h + tap * 2048, w_l + tap * 2048, bp);
        }
    };
    wdma(0);

    // ---- the image window as split planes [pixel][4 channels] (zeros outside the image: conv1a's padding)
    {
        const float4* img4 = reinterpret_cast<const float4*>(p.img) + (size_t)b * p.h * p.wd;
#pragma unroll
        for (int j = 0; j < (IP + NT - 1) / NT; ++j) {
            const int ip = tid + NT * j;
            if (ip < IP) {
                const int wy = ip / IW, wx = ip - wy * IW;
                const int gy = ay0 - 1 + wy, gx = ax0 - 1 + wx;
                const bool in = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.wd;
                const size_t px = (size_t)min(max(gy, 0), p.h - 1) * p.wd + min(max(gx, 0), p.wd - 1);
                float4 v;
                if (p.nchw3) {      // three planes of the image as the caller holds it: the repack kernel and its 16-byte pixels are skipped
                    const float* pl = p.img + (size_t)b * 3 * p.h * p.wd + px;
                    v = make_float4(pl[0], pl[(size_t)p.h * p.wd], pl[2 * (size_t)p.h * p.wd], 0.f);
                } else {
                    v = img4[px];
                }
                if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
                half4 hi, lo;
                gemmx3::split4(v, p.act_scale, hi, lo, amax);
                *reinterpret_cast<half4*>(&i_h[ip * 4]) = hi;
                *reinterpret_cast<half4*>(&i_l[ip * 4]) = lo;
            }
        }
    }
    __syncthreads();

    // conv1b accumulators: rows (pixels) 32 wm + .., columns (channels) 32 wn + r
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accb = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // ================================================================ conv1a, channels 32 half .. 32 half + 31, on the window
        {
            half8 wa_h[3], wa_l[3];
            const size_t wrow = (size_t)(32 * half + r) * KA;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                wa_h[ks] = *reinterpret_cast<const half8*>(p.wah + wrow + 16 * ks + 8 * h);
                wa_l[ks] = *reinterpret_cast<const half8*>(p.wal + wrow + 16 * ks + 8 * h);
            }
            // bias -> BN folded with the accumulator's scale and the planes' 16: y * 16 = acc * ca + cc, one multiply-add per value
            float ca[16], cc[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = 32 * half + rowoff(e, h);
                const float sc = p.sa[ch];
                ca[e] = p.inva * sc * p.act_scale;
                cc[e] = (p.ba[ch] * sc + p.ta[ch]) * p.act_scale;
            }
#pragma unroll 1
            for (int bi = (p.abl & 1) ? NBLK : wave; bi < NBLK; bi += NT / 64) {
                const int hpr = 32 * bi + r;
                const int hp = min(hpr, WP - 1);
                const int wy = hp / WW, wx = hp - wy * WW;
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    // this lane's eight k of the step: taps 4 ks + 2 h and the next one, four channels each (taps >= 9 are the padding
                    // of K: their weights are zeros, any finite pixel will do)
                    const int t0 = min(4 * ks + 2 * h, 8), t1 = min(4 * ks + 2 * h + 1, 8);
                    const int ip0 = (wy + t0 / 3) * IW + wx + (t0 - (t0 / 3) * 3);
                    const int ip1 = (wy + t1 / 3) * IW + wx + (t1 - (t1 / 3) * 3);
                    const half4 h0 = *reinterpret_cast<const half4*>(&i_h[ip0 * 4]), h1 = *reinterpret_cast<const half4*>(&i_h[ip1 * 4]);
                    const half4 l0 = *reinterpret_cast<const half4*>(&i_l[ip0 * 4]), l1 = *reinterpret_cast<const half4*>(&i_l[ip1 * 4]);
                    const half8 bh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const half8 bl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa_h[ks], bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa_l[ks], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa_h[ks], bh, acc, 0, 0, 0);
                }
                const int gy = ay0 + wy, gx = ax0 + wx;
                const bool in = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.wd;      // outside: conv1b's zero padding
                const float keep = in ? 1.0f : 0.0f;
                half4 hi[4], lo[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * g4 + i;
                        const float x = fmaxf(fmaf(acc[e], ca[e], cc[e]), 0.f) * keep;      // 16 * ReLU(BN(conv + bias)), 0 outside the map
                        amax = fmaxf(amax, x);
                        const _Float16 hx = (_Float16)x;
                        hi[g4][i] = hx;
                        lo[g4][i] = gemmx3::lo_part(x, hx);
                    }
                }
                if (hpr < WP) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int off = yoff(wy, wx, g4) + 4 * h;      // channels 8 g4 + 4 h + 0..3 of the half
                        *reinterpret_cast<half4*>(&y_h[off]) = hi[g4];
                        *reinterpret_cast<half4*>(&y_l[off]) = lo[g4];
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the half's conv1b weights have landed
        __syncthreads();
        // ================================================================ conv1b over the half's 32 input channels: nine taps in place
        {
            const int tr = 32 * wm + r, oy = tr >> 4, ox = tr & 15;
            const int brow = (32 * wn + r) * 32;
#pragma unroll 1
            for (int tap = (p.abl & 2) ? 9 : 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = 2 * ks + h;
                    const int aoff = yoff(2 * oy + ky, 2 * ox + kx, slot);
                    const int boff = tap * 2048 + brow + swz(slot, r) * 8;      // row 32 wn + r: same swizzle as r
                    const half8 ah = *reinterpret_cast<const half8*>(&y_h[aoff]);
                    const half8 al = *reinterpret_cast<const half8*>(&y_l[aoff]);
                    const half8 bh = *reinterpret_cast<const half8*>(&w_h[boff]);
                    const half8 bl = *reinterpret_cast<const half8*>(&w_l[boff]);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accb, 0, 0, 0);
                }
            }
        }
        __syncthreads();                      // everybody is done with the window planes and the weight tiles of this half
        if (half == 0 && !(p.abl & 4)) wdma(1);               // the other half's weights travel under its conv1a
    }
    x3_range_flag(p.status, amax);

    // ================================================================ epilogue: bias -> BN -> ReLU -> store
    {
        const int col = 32 * wn + r;
        const float bi = p.bb[col], sc = p.sb[col], sh = p.tb[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tr = 32 * wm + rowoff(e, h);
            const int gy = oy0 + (tr >> 4), gx = ox0 + (tr & 15);
            float v = accb[e] * p.invb + bi;
            v = v * sc + sh;
            v = fmaxf(v, 0.f);
            if (gy < p.ho && gx < p.wo) p.out[(((size_t)b * p.ho + gy) * p.wo + gx) * 64 + col] = v;
        }
    }
}

}  // namespace

/* SFD2's conv1a -> conv1b (nets/sfd2.py:135-139,281-282) in one launch on the split-fp16 path: img NHWC4 fp32 [batch][h][w][4]
   (pram_image_to_nhwc4_f32), out NHWC fp32 [batch][ho][wo][64] with ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1.  wa: conv1a's weights
   [64][3][3][4] flattened to [64][36], zero-padded to [64][48] and split into (hi, lo) planes * wa_scale; wb: conv1b's weights
   [64][3][3][64] as (hi, lo) planes * wb_scale (pram_conv2d_nhwc_x3_f32's operand); b? / s? / t?: bias and the eval-mode BatchNorm
   as per-channel scale / shift of each layer. */
extern "C" int pram_sfd2_conv1_x3_f32(const float* img, int batch, int h, int w, const void* wa_hi, const void* wa_lo, float wa_scale,
                                      const float* ba, const float* sa, const float* ta, const void* wb_hi, const void* wb_lo,
                                      float wb_scale, const float* bb, const float* sb, const float* tb, float* out, int img_nchw3,
                                      void* stream) {
    PRAM_REQUIRE(img && out && wa_hi && wa_lo && wb_hi && wb_lo && ba && sa && ta && bb && sb && tb, "pram_sfd2_conv1_x3_f32: null pointer");
    PRAM_REQUIRE(batch >= 0 && h > 0 && w > 0 && wa_scale > 0.f && wb_scale > 0.f, "pram_sfd2_conv1_x3_f32: bad sizes");
    if (batch == 0) return PRAM_OK;
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    c1::Args p{img, out, (const _Float16*)wa_hi, (const _Float16*)wa_lo, 1.0f / (pram_act_scale() * wa_scale), ba, sa, ta,
               (const _Float16*)wb_hi, (const _Float16*)wb_lo, 1.0f / (pram_act_scale() * wb_scale), bb, sb, tb,
               batch, h, w, ho, wo, cdiv(wo, c1::TW), cdiv(ho, c1::TH), pram_status_ptr(), 0, img_nchw3 != 0};
#ifdef PRAM_PROFILING      // ablations return garbage: compiled into profiling builds only (profiles/tools/build_variants.py ...:-DPRAM_PROFILING)
    { const char* e = getenv("PRAM_C1_ABLATE"); p.abl = e ? atoi(e) : 0; }
#else
    p.abl = 0;
#endif
    p.act_scale = pram_act_scale();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv1ab_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c1::SMEM_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(conv1ab_x3_kernel, dim3(batch * p.tiles_x * p.tiles_y), dim3(c1::NT), c1::SMEM_BYTES, (hipStream_t)stream, p);
    return pram_launch_status("pram_sfd2_conv1_x3_f32");
}
