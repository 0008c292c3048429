// SFD2 detector post-processing: score map, NMS, keypoint selection, descriptor sampling.
// Reference: nets/sfd2.py:20-64 (simple_nms, remove_borders, top_k_keypoints, sample_descriptors),
// :294-329 (softmax / depth-to-space / threshold / fallback), :348-369 (ResNet4x.sample).
// All HBM/latency-bound integer-and-compare work; results are bit-exact given the same score map.
#include "common.h"
#include <math.h>

namespace {

// ---------------------------------------------------------------- K2: softmax65 + depth-to-space
// one wave per 8x8 cell: lane c holds channel c, lane 0 also the dustbin channel 64.
__global__ __launch_bounds__(256) void score_map_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                        int hc, int wc, int ncell) {
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= ncell) return;
    const float* src = logits + (size_t)cell * 65;
    const float x = src[lane];
    const float xd = src[64];
    const float mx = fmaxf(wave_max(x), xd);
    const float e = expf(x - mx);
    const float sum = wave_sum(e) + expf(xd - mx);
    const int cx = cell % wc;
    const int t = cell / wc;
    const int cy = t % hc;
    const int b = t / hc;
    const int W = wc * 8;
    score[((size_t)b * hc * 8 + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)] = e / sum;
}

// ---------------------------------------------------------------- K3: simple_nms
// Five chained (2r+1)^2 max-pools (1 on the scores, then 2 x {mask, suppressed scores}).  Each pool is
// one launch: a 32x32 output tile + r halo staged in LDS, separable row/column max, and the
// stage's elementwise rule fused behind it.  The intermediate maps (mask, suppressed scores) are
// fp32 images that stay L2-resident (1.2 MB per frame).  Out-of-image pixels are -inf for score
// pools and 0 for mask pools (max_pool2d's implicit padding); all comparisons are exact fp32.
constexpr int NMS_T = 32;
constexpr int NMS_RMAX = 4;

// STAGE 0: mask  = (s == pool(s))
// STAGE 1: in = mask;  supp = pool(mask) > 0;  x = supp ? 0 : s           -> outputs x, supp
// STAGE 2: in = x;     new = (x == pool(x)) & !supp;  mask |= new         -> output mask (or final scores)
template <int R, int STAGE>
__global__ __launch_bounds__(256) void nms_stage_kernel(const float* __restrict__ in, const float* __restrict__ score,
                                                        const float* __restrict__ supp_in, const float* __restrict__ mask_in,
                                                        float* __restrict__ out0, float* __restrict__ out1, int h, int w,
                                                        int final_scores) {
    constexpr int S = NMS_T + 2 * R;
    __shared__ float tile[S][S + 1];
    __shared__ float rowm[S][NMS_T + 1];
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * NMS_T, x0 = blockIdx.x * NMS_T;
    const size_t off = (size_t)b * h * w;
    const float pad = (STAGE == 1) ? 0.f : -INFINITY;
    for (int i = threadIdx.x; i < S * S; i += 256) {
        const int ly = i / S, lx = i - ly * S;
        const int gy = y0 + ly - R, gx = x0 + lx - R;
        const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
        tile[ly][lx] = ok ? in[off + (size_t)gy * w + gx] : pad;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S * NMS_T; i += 256) {      // row pass: S rows x 32 output columns
        const int ly = i / NMS_T, lx = i - ly * NMS_T;
        float m = tile[ly][lx];
#pragma unroll
        for (int d = 1; d <= 2 * R; ++d) m = fmaxf(m, tile[ly][lx + d]);
        rowm[ly][lx] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NMS_T * NMS_T; i += 256) {  // column pass + stage rule
        const int ly = i / NMS_T, lx = i - ly * NMS_T;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= h || gx >= w) continue;
        float m = rowm[ly][lx];
#pragma unroll
        for (int d = 1; d <= 2 * R; ++d) m = fmaxf(m, rowm[ly + d][lx]);
        const size_t g = off + (size_t)gy * w + gx;
        const float c = tile[ly + R][lx + R];
        if (STAGE == 0) {
            out0[g] = (c == m) ? 1.f : 0.f;
        } else if (STAGE == 1) {
            const bool sp = m > 0.f;
            out0[g] = sp ? 0.f : score[g];
            out1[g] = sp ? 1.f : 0.f;
        } else {
            const bool keep = (mask_in[g] != 0.f) || ((c == m) && supp_in[g] == 0.f);
            out0[g] = final_scores ? (keep ? score[g] : 0.f) : (keep ? 1.f : 0.f);
        }
    }
}

template <int R>
static void nms_launch(const float* score, float* out, float* ws, int batch, int h, int w, hipStream_t st) {
    const size_t n = (size_t)batch * h * w;
    float* mask = ws;           // [n]
    float* x = ws + n;          // [n] suppressed scores
    float* supp = ws + 2 * n;   // [n]
    float* mask2 = ws + 3 * n;  // [n]
    dim3 grid(cdiv(w, NMS_T), cdiv(h, NMS_T), batch), blk(256);
    hipLaunchKernelGGL((nms_stage_kernel<R, 0>), grid, blk, 0, st, score, score, nullptr, nullptr, mask, nullptr, h, w, 0);
    hipLaunchKernelGGL((nms_stage_kernel<R, 1>), grid, blk, 0, st, mask, score, nullptr, nullptr, x, supp, h, w, 0);
    hipLaunchKernelGGL((nms_stage_kernel<R, 2>), grid, blk, 0, st, x, score, supp, mask, mask2, nullptr, h, w, 0);
    hipLaunchKernelGGL((nms_stage_kernel<R, 1>), grid, blk, 0, st, mask2, score, nullptr, nullptr, x, supp, h, w, 0);
    hipLaunchKernelGGL((nms_stage_kernel<R, 2>), grid, blk, 0, st, x, score, supp, mask2, out, nullptr, h, w, 1);
}

// ---------------------------------------------------------------- K4: keypoint selection
constexpr int SEL_T = 1024;       // threads per image
constexpr int SEL_KMAX = 8192;    // max_keypoints supported by the in-LDS sort

constexpr int SEL_SLABS = 64;     // pixel slabs per frame of the count / compaction kernels (one workgroup each)

struct SelWs {
    int* cnt_hi;            // [batch]   #(nms >= conf_th), no border test  (fallback decision)
    int* slab;              // [batch][SEL_SLABS][2]  candidates per slab inside the border: >= conf_th | >= conf_th / 2
    unsigned* cand;         // [batch][h*w]  flat indices of candidates, row-major order
    unsigned* cbits;        // [batch][h*w]  their score bit patterns (same order), so later passes stream one array
    unsigned long long* keys;   // [batch][p2]  sort keys in global memory when max_keypoints > SEL_KMAX (p2 = next power of two); else unused
    int p2;
};

// slab s of a frame = pixels [s * slab_px, (s + 1) * slab_px) (slab_px a multiple of 8).  One workgroup per (slab, frame):
// the frame's count above conf_th without the border test (the min_keypoints fallback decision, nets/sfd2.py:311) and, inside
// the border, the slab's candidates at conf_th and at conf_th / 2 — whichever threshold the frame ends up with, the compaction
// kernel below knows every slab's offset without a second counting pass.
__global__ __launch_bounds__(256) void sel_count_kernel(const float* __restrict__ nms, int h, int w, int slab_px, float th, int border,
                                                        SelWs ws) {
    __shared__ int red[3][4];
    const int b = blockIdx.y, sl = blockIdx.x;
    const int hw = h * w;
    const float* img = nms + (size_t)b * hw;
    const int p0 = sl * slab_px, p1 = min(hw, p0 + slab_px);
    const float th2 = th * 0.5f;
    int c_hi = 0, c1 = 0, c2 = 0;
    for (int i = p0 + threadIdx.x; i < p1; i += 256) {
        const float v = img[i];
        const int y = i / w, x = i - y * w;
        const bool inb = y >= border && y < h - border && x >= border && x < w - border;
        c_hi += v >= th;
        c1 += inb && v >= th;
        c2 += inb && v >= th2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c_hi += __shfl_xor(c_hi, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = c_hi; red[1][wave] = c1; red[2][wave] = c2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t_hi = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        if (t_hi) atomicAdd(&ws.cnt_hi[b], t_hi);      // integer: the order of the additions does not matter
        ws.slab[((size_t)b * SEL_SLABS + sl) * 2 + 0] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        ws.slab[((size_t)b * SEL_SLABS + sl) * 2 + 1] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

// Ordered compaction of the candidates (>= th, inside the border) by SEL_SLABS workgroups per frame: a slab's offset is the sum
// of the counts of the slabs before it, inside the slab a block scan keeps the row-major order — the list (and everything
// downstream) is the one the single-workgroup sweep produced, 307 200 pixels are no longer one workgroup's 38 dependent scans.
__global__ __launch_bounds__(256) void sel_compact_kernel(const float* __restrict__ nms, int h, int w, int slab_px, float conf_th,
                                                          int min_kp, int border, int fallback_ref, SelWs ws) {
    __shared__ int wsum[4];
    __shared__ int s_base;
    const int b = blockIdx.y, sl = blockIdx.x;
    const int hw = h * w;
    const float* img = nms + (size_t)b * hw;
    const int ref = fallback_ref < 0 ? b : fallback_ref;
    const bool low = ws.cnt_hi[ref] <= min_kp;
    const float th = low ? conf_th * 0.5f : conf_th;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int q = 0; q < sl; ++q) off += ws.slab[((size_t)b * SEL_SLABS + q) * 2 + (low ? 1 : 0)];
        s_base = off;
    }
    __syncthreads();
    int base = s_base;
    unsigned* cand = ws.cand + (size_t)b * hw;
    unsigned* cbits = ws.cbits + (size_t)b * hw;
    const int p0 = sl * slab_px, p1 = min(hw, p0 + slab_px);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = 8;
    for (int sweep = p0; sweep < p1; sweep += 256 * PER) {
        const int i0 = sweep + threadIdx.x * PER;
        float v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = (i0 + j < p1) ? img[i0 + j] : -1.f;
        int y = i0 / w, x = i0 - y * w;
        unsigned keepmask = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool keep = (i0 + j < p1) && v[j] >= th && y >= border && y < h - border && x >= border && x < w - border;
            keepmask |= (keep ? 1u : 0u) << j;
            if (++x == w) { x = 0; ++y; }
        }
        const int n = __popc(keepmask);
        int inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pre = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (q < wave) pre += wsum[q]; tot += wsum[q]; }
        int pos = base + pre + inc - n;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (keepmask & (1u << j)) {
                cand[pos] = (unsigned)(i0 + j);
                cbits[pos] = __float_as_uint(v[j]);
                ++pos;
            }
        base += tot;
        __syncthreads();
    }
}

// block-wide exclusive scan of one int per thread (1024 threads = 16 waves); returns the exclusive
// prefix and writes the block total to *total.  Uses sbuf[17].
__device__ __forceinline__ int block_excl_scan(int v, int* sbuf, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sbuf[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < SEL_T / 64; ++i) { const int t = sbuf[i]; sbuf[i] = run; run += t; }
        sbuf[16] = run;
    }
    __syncthreads();
    const int res = sbuf[wave] + inc - v;
    *total = sbuf[16];
    __syncthreads();
    return res;
}

// BIG: max_keypoints beyond the in-LDS sort (nets/sfd2.py:38-50 has no bound): the same selection with its sort keys in global
// memory — a rare configuration (a frame needs more than 8192 candidates after the NMS), built for completeness, not for speed.
template <bool BIG>
__global__ __launch_bounds__(SEL_T) void sel_select_kernel(const float* __restrict__ nms, int h, int w, float conf_th,
                                                           int min_kp, int border, int kmax, int fallback_ref, SelWs ws,
                                                           float* __restrict__ kpts, float* __restrict__ scores,
                                                           int* __restrict__ counts) {
    __shared__ unsigned long long lds_keys[BIG ? 1 : SEL_KMAX];
    unsigned long long* keys = BIG ? ws.keys + (size_t)blockIdx.x * ws.p2 : lds_keys;
    __shared__ int hist[256];
    __shared__ int sbuf[17];
    __shared__ unsigned s_prefix;
    __shared__ int s_remaining;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int hw = h * w;
    unsigned* cand = ws.cand + (size_t)b * hw;
    unsigned* cbits = ws.cbits + (size_t)b * hw;
    const int ref = fallback_ref < 0 ? b : fallback_ref;

    // ---- pass 1 ran as sel_compact_kernel: cand / cbits hold the frame's candidates in row-major order, c of them
    int c = 0;
    {
        const bool low = ws.cnt_hi[ref] <= min_kp;
        for (int q = 0; q < SEL_SLABS; ++q) c += ws.slab[((size_t)b * SEL_SLABS + q) * 2 + (low ? 1 : 0)];
    }

    float* ko = kpts + (size_t)b * kmax * 2;
    float* so = scores + (size_t)b * kmax;
    if (c <= kmax) {   // fewer than k: keep the row-major nonzero() order (top_k_keypoints, sfd2.py:47-48)
        for (int i = tid; i < c; i += SEL_T) {
            const unsigned idx = cand[i];
            ko[2 * i] = (float)(idx % w);
            ko[2 * i + 1] = (float)(idx / w);
            so[i] = __uint_as_float(cbits[i]);
        }
        if (tid == 0) counts[b] = c;
        return;
    }

    // ---- radix select of the k-th largest score (positive floats: bit pattern is monotone)
    if (tid == 0) { s_prefix = 0u; s_remaining = kmax; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i0 = tid; i0 < c; i0 += 4 * SEL_T) {      // four independent loads in flight per thread
            unsigned bits[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bits[u] = (i0 + u * SEL_T < c) ? cbits[i0 + u * SEL_T] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * SEL_T < c && (bits[u] & himask) == prefix) atomicAdd(&hist[(bits[u] >> shift) & 255], 1);
        }
        __syncthreads();
        // the digit of the k-th largest: the largest d whose suffix count S(d) = #(digit >= d) reaches the remaining rank.  The
        // 256 suffix counts come from one parallel scan (thread t < 256 owns digit 255 - t: a prefix scan in descending digit
        // order) — the serial walk over the bins by one thread was 256 dependent LDS reads per pass, four passes per frame.
        {
            const int rem = s_remaining;
            __syncthreads();                                   // everyone has read s_remaining / s_prefix before they change
            int own = 0, inc = 0;
            if (tid < 256) {
                own = hist[255 - tid];
                inc = own;
                const int lane = tid & 63;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t2 = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += t2;
                }
                if (lane == 63) sbuf[tid >> 6] = inc;
            }
            __syncthreads();
            if (tid < 256) {
                int pre = 0;
                for (int q = 0; q < (tid >> 6); ++q) pre += sbuf[q];
                const int s_incl = pre + inc, s_excl = s_incl - own;      // #(digit >= d), #(digit > d) for d = 255 - tid
                // exactly one digit satisfies S(d + 1) < rem <= S(d); if the candidates run out first (cannot happen: c > kmax
                // here) digit 0 takes what is left, as the serial walk did
                const bool hit = (s_excl < rem && rem <= s_incl) || (tid == 255 && s_incl < rem);
                if (hit) {
                    s_remaining = rem - s_excl;
                    s_prefix = prefix | ((unsigned)(255 - tid) << shift);
                }
            }
        }
        __syncthreads();
    }
    const unsigned tbits = s_prefix;      // bit pattern of the k-th largest score
    const int take_eq = s_remaining;      // how many candidates equal to it are taken (lowest index first)

    // ---- pass 2: ordered gather of the selected k into LDS keys
    int ngt = 0, neq = 0;
    for (int base = 0; base < c; base += SEL_T) {
        const int i = base + tid;
        unsigned idx = 0, bits = 0;
        bool gt = false, eq = false;
        if (i < c) {
            idx = cand[i];
            bits = cbits[i];
            gt = bits > tbits;
            eq = bits == tbits;
        }
        int tot_eq, tot_gt;
        const int peq = block_excl_scan(eq ? 1 : 0, sbuf, &tot_eq);
        const int pgt = block_excl_scan(gt ? 1 : 0, sbuf, &tot_gt);
        const bool sel = gt || (eq && (neq + peq) < take_eq);
        // slot: order within the selected set does not matter before the sort; use (#gt so far) + (#eq taken so far)
        if (sel) {
            const int eq_before = min(neq + peq, take_eq);
            const int slot = ngt + pgt + eq_before;
            keys[slot] = ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
        }
        ngt += tot_gt;
        neq += tot_eq;
    }
    int p2 = 1;
    while (p2 < kmax) p2 <<= 1;
    for (int i = kmax + tid; i < p2; i += SEL_T) keys[i] = 0ull;
    if (BIG) __threadfence_block();      // the keys live in global memory: make them visible to the workgroup at every barrier
    __syncthreads();
    // ---- bitonic sort, descending: (score desc, flat index asc)
    for (int k2 = 2; k2 <= p2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += SEL_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool desc = (i & k2) == 0;
                    if (desc ? (a < bb) : (a > bb)) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            if (BIG) __threadfence_block();
            __syncthreads();
        }
    }
    for (int i = tid; i < kmax; i += SEL_T) {
        const unsigned long long kk = keys[i];
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull);
        ko[2 * i] = (float)(idx % w);
        ko[2 * i + 1] = (float)(idx / w);
        so[i] = __uint_as_float((unsigned)(kk >> 32));
    }
    if (tid == 0) counts[b] = kmax;
}

// ---------------------------------------------------------------- K5: bilinear sampling of an NHWC map
// one wave per keypoint; grid_sample(bilinear, align_corners=True, zeros padding) arithmetic.
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ fmap, int fh, int fw, int c,
                                                     const float* __restrict__ kpts, const int* __restrict__ lens,
                                                     int n_max, float s, int l2norm, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int len = lens ? lens[b] : n_max;
    if (n >= len) return;
    const float kx = kpts[((size_t)b * n_max + n) * 2], ky = kpts[((size_t)b * n_max + n) * 2 + 1];
    // s > 0: sample_descriptors map  k = k - s/2 + 0.5 ; k /= (w*s - s/2 - 0.5, h*s - s/2 - 0.5) ; k = k*2 - 1
    // s <= 0: kpts are already normalised grid coordinates in [-1, 1] (plain F.grid_sample)
    float gx = kx, gy = ky;
    if (s > 0.f) {
        const float half = s * 0.5f;
        const float dx = (float)fw * s - half - 0.5f, dy = (float)fh * s - half - 0.5f;
        gx = (kx - half) + 0.5f;
        gy = (ky - half) + 0.5f;
        gx = gx / dx;
        gy = gy / dy;
        gx = gx * 2.f - 1.f;
        gy = gy * 2.f - 1.f;
    }
    // grid_sampler_compute_source_index, align_corners=True: ((g + 1) / 2) * (size - 1)
    const float ix = ((gx + 1.f) / 2.f) * (float)(fw - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(fh - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((fx0 + 1.f) - ix) * ((fy0 + 1.f) - iy);
    const float wne = (ix - fx0) * ((fy0 + 1.f) - iy);
    const float wsw = ((fx0 + 1.f) - ix) * (iy - fy0);
    const float wse = (ix - fx0) * (iy - fy0);
    const bool vx0 = (unsigned)x0 < (unsigned)fw, vx1 = (unsigned)x1 < (unsigned)fw;
    const bool vy0 = (unsigned)y0 < (unsigned)fh, vy1 = (unsigned)y1 < (unsigned)fh;
    const float* base = fmap + (size_t)b * fh * fw * c;
    float* dst = out + ((size_t)b * n_max + n) * c;
    // up to two float4 per lane (c <= 512), named registers (no runtime-indexed arrays)
    auto gather = [&](int c4) -> float4 {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 * 4 < c) {
            auto tap = [&](bool ok, int yy, int xx, float wgt) {
                if (ok) {
                    const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * fw + xx) * c + c4 * 4);
                    a.x += v.x * wgt; a.y += v.y * wgt; a.z += v.z * wgt; a.w += v.w * wgt;
                }
            };
            tap(vx0 && vy0, y0, x0, wnw);
            tap(vx1 && vy0, y0, x1, wne);
            tap(vx0 && vy1, y1, x0, wsw);
            tap(vx1 && vy1, y1, x1, wse);
        }
        return a;
    };
    float4 a0 = gather(lane), a1 = gather(lane + 64);
    if (l2norm) {
        const float sq = ((a0.x * a0.x + a0.y * a0.y) + (a0.z * a0.z + a0.w * a0.w)) +
                         ((a1.x * a1.x + a1.y * a1.y) + (a1.z * a1.z + a1.w * a1.w));
        const float d = fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
        a0.x /= d; a0.y /= d; a0.z /= d; a0.w /= d;
        a1.x /= d; a1.y /= d; a1.z /= d; a1.w /= d;
    }
    if (lane * 4 < c) *reinterpret_cast<float4*>(dst + lane * 4) = a0;
    if ((lane + 64) * 4 < c) *reinterpret_cast<float4*>(dst + (lane + 64) * 4) = a1;
}

// F.normalize over the channel (last NHWC) dim, in place: x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + (size_t)row * cols;
    float sq = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float d = fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
    for (int c = lane * 4; c < cols; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(p + c);
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        *reinterpret_cast<float4*>(p + c) = v;
    }
}

__global__ void score_lookup_kernel(const float* __restrict__ sm, long long map_stride, int h, int w,
                                    const float* __restrict__ kpts, const int* __restrict__ lens, int n_max,
                                    float* __restrict__ out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens ? lens[b] : n_max;
    if (n >= len) return;
    const long long x = (long long)kpts[((size_t)b * n_max + n) * 2];       // .long() truncation
    const long long y = (long long)kpts[((size_t)b * n_max + n) * 2 + 1];
    out[(size_t)b * n_max + n] = sm[b * map_stride + y * w + x];
}

}  // namespace

extern "C" int pram_score_map_f32(const float* logits, float* score, int batch, int hc, int wc, void* stream) {
    PRAM_REQUIRE(logits && score, "pram_score_map_f32: null pointer");
    const int ncell = batch * hc * wc;
    if (ncell == 0) return PRAM_OK;
    hipLaunchKernelGGL(score_map_kernel, dim3(cdiv(ncell, 4)), dim3(256), 0, (hipStream_t)stream, logits, score, hc, wc, ncell);
    return pram_launch_status("pram_score_map_f32");
}

extern "C" size_t pram_simple_nms_workspace_bytes(int batch, int h, int w) { return (size_t)batch * h * w * 4 * sizeof(float); }

extern "C" int pram_simple_nms_f32(const float* score, float* nms, int batch, int h, int w, int radius, void* workspace,
                                   void* stream) {
    PRAM_REQUIRE(score && nms && workspace, "pram_simple_nms_f32: null pointer");
    PRAM_REQUIRE(radius >= 0 && radius <= NMS_RMAX, "pram_simple_nms_f32: radius %d > %d", radius, NMS_RMAX);
    PRAM_REQUIRE(score != nms, "pram_simple_nms_f32: in-place is not supported");
    if (batch == 0) return PRAM_OK;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    switch (radius) {
        case 0: nms_launch<0>(score, nms, ws, batch, h, w, st); break;
        case 1: nms_launch<1>(score, nms, ws, batch, h, w, st); break;
        case 2: nms_launch<2>(score, nms, ws, batch, h, w, st); break;
        case 3: nms_launch<3>(score, nms, ws, batch, h, w, st); break;
        default: nms_launch<4>(score, nms, ws, batch, h, w, st); break;
    }
    return pram_launch_status("pram_simple_nms_f32");
}

static SelWs sel_carve(void* ws, int batch, int h, int w, int max_keypoints, size_t* total) {
    SelWs s;
    char* base = (char*)ws;
    size_t off = 0;
    s.cnt_hi = (int*)(base + off);
    off += ((size_t)batch * 4 + 255) & ~(size_t)255;
    s.slab = (int*)(base + off);
    off += ((size_t)batch * SEL_SLABS * 2 * 4 + 255) & ~(size_t)255;
    s.cand = (unsigned*)(base + off);
    off += ((size_t)batch * h * w * 4 + 255) & ~(size_t)255;
    s.cbits = (unsigned*)(base + off);
    off += ((size_t)batch * h * w * 4 + 255) & ~(size_t)255;
    s.keys = nullptr;
    s.p2 = 0;
    if (max_keypoints > SEL_KMAX && max_keypoints < h * w) {      // top-k beyond the in-LDS sort: keys in global memory
        int p2 = 1;
        while (p2 < max_keypoints) p2 <<= 1;
        s.keys = (unsigned long long*)(base + off);
        s.p2 = p2;
        off += ((size_t)batch * p2 * 8 + 255) & ~(size_t)255;
    }
    if (total) *total = off;
    return s;
}

extern "C" size_t pram_select_keypoints_workspace_bytes(int batch, int h, int w, int max_keypoints) {
    size_t total = 0;
    sel_carve(nullptr, batch, h, w, max_keypoints, &total);
    return total;
}

extern "C" int pram_select_keypoints_f32(const float* nms, int batch, int h, int w, float conf_th, int min_keypoints,
                                         int border, int max_keypoints, int fallback_ref, float* kpts, float* scores,
                                         int* counts, void* workspace, void* stream) {
    PRAM_REQUIRE(nms && kpts && scores && counts && workspace, "pram_select_keypoints_f32: null pointer");
    // a bound >= h*w can never be exceeded ("keep all", nets/sfd2.py:324 with max_keypoints < 0): the top-k sort is then
    // unreachable; bounds up to SEL_KMAX sort in LDS, larger ones in the workspace (pram_select_keypoints_workspace_bytes grows)
    PRAM_REQUIRE(max_keypoints > 0, "pram_select_keypoints_f32: max_keypoints=%d must be positive (>= h*w = keep all)", max_keypoints);
    PRAM_REQUIRE(fallback_ref < batch, "pram_select_keypoints_f32: fallback_ref out of range");
    PRAM_REQUIRE(conf_th > 0.f, "pram_select_keypoints_f32: conf_th must be positive");
    if (batch == 0) return PRAM_OK;
    hipStream_t st = (hipStream_t)stream;
    SelWs ws = sel_carve(workspace, batch, h, w, max_keypoints, nullptr);
    if (hipMemsetAsync(ws.cnt_hi, 0, (size_t)batch * 4, st) != hipSuccess) {
        pram_set_error("pram_select_keypoints_f32: memset failed");
        return PRAM_E_LAUNCH;
    }
    const int slab_px = (cdiv(h * w, SEL_SLABS) + 7) & ~7;
    hipLaunchKernelGGL(sel_count_kernel, dim3(SEL_SLABS, batch), dim3(256), 0, st, nms, h, w, slab_px, conf_th, border, ws);
    hipLaunchKernelGGL(sel_compact_kernel, dim3(SEL_SLABS, batch), dim3(256), 0, st, nms, h, w, slab_px, conf_th, min_keypoints, border,
                       fallback_ref, ws);
    if (ws.keys) hipLaunchKernelGGL(sel_select_kernel<true>, dim3(batch), dim3(SEL_T), 0, st, nms, h, w, conf_th, min_keypoints, border,
                                    max_keypoints, fallback_ref, ws, kpts, scores, counts);
    else hipLaunchKernelGGL(sel_select_kernel<false>, dim3(batch), dim3(SEL_T), 0, st, nms, h, w, conf_th, min_keypoints, border,
                            max_keypoints, fallback_ref, ws, kpts, scores, counts);
    return pram_launch_status("pram_select_keypoints_f32");
}

extern "C" int pram_sample_nhwc_f32(const float* fmap, int batch, int fh, int fw, int c, const float* kpts, const int* lens,
                                    int n_max, int s, int l2norm, float* out, void* stream) {
    PRAM_REQUIRE(fmap && kpts && out, "pram_sample_nhwc_f32: null pointer");
    PRAM_REQUIRE(c % 4 == 0 && c <= 512, "pram_sample_nhwc_f32: c=%d must be a multiple of 4 and <= 512", c);
    if (batch == 0 || n_max == 0) return PRAM_OK;
    hipLaunchKernelGGL(sample_kernel, dim3(cdiv(n_max, 4), batch), dim3(256), 0, (hipStream_t)stream, fmap, fh, fw, c, kpts,
                       lens, n_max, (float)s, l2norm, out);
    return pram_launch_status("pram_sample_nhwc_f32");
}

extern "C" int pram_l2norm_rows_f32(float* x, int rows, int cols, void* stream) {
    PRAM_REQUIRE(x && cols % 4 == 0, "pram_l2norm_rows_f32: cols must be a multiple of 4");
    if (rows == 0) return PRAM_OK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, cols);
    return pram_launch_status("pram_l2norm_rows_f32");
}

extern "C" int pram_score_lookup_f32(const float* score_map, long long map_stride, int h, int w, const float* kpts,
                                     const int* lens, int batch, int n_max, float* out, void* stream) {
    PRAM_REQUIRE(score_map && kpts && out, "pram_score_lookup_f32: null pointer");
    if (batch == 0 || n_max == 0) return PRAM_OK;
    hipLaunchKernelGGL(score_lookup_kernel, dim3(cdiv(n_max, 256), batch), dim3(256), 0, (hipStream_t)stream, score_map,
                       map_stride, h, w, kpts, lens, n_max, out);
    return pram_launch_status("pram_score_lookup_f32");
}
