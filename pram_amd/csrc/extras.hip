// Kernels at the edges of the hot path: bilinear resize (score map of non-multiple-of-8 frames,
// multi-scale extraction), generic grid_sample, the recogniser epilogue (softmax / background filter /
// argmax, localization/frame.py:96-121), full descending row sort (torch.topk(k=C),
// localization/multimap3d.py:348-350) and row top-2 (nearest-neighbour matching,
// localization/matchers/nearest_neighbor.py:5-17; projection refinement singlemap3d.py:428-433).
#include "common.h"
#include <math.h>

namespace {

// F.interpolate(mode='bilinear', align_corners=True) on planar maps [planes][h][w] -> [planes][oh][ow]
// (ATen upsample_bilinear2d arithmetic: src = dst * (in-1)/(out-1), lambda = src - floor(src)).
__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w,
                                       int oh, int ow, float rh, float rw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)planes * oh * ow;
    if (i >= total) return;
    const int ox = (int)(i % ow);
    const long long t = i / ow;
    const int oy = (int)(t % oh);
    const int pl = (int)(t / oh);
    const float h1r = rh * (float)oy;
    const int h1 = (int)h1r;
    const int h1p = (h1 < h - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float w1r = rw * (float)ox;
    const int w1 = (int)w1r;
    const int w1p = (w1 < w - 1) ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float* p = in + (size_t)pl * h * w + (size_t)h1 * w + w1;
    out[i] = h0l * (w0l * p[0] + w1l * p[w1p]) + h1l * (w0l * p[(size_t)h1p * w] + w1l * p[(size_t)h1p * w + w1p]);
}

// ---- recogniser epilogue: one wave per token -------------------------------------------------------
// seg_scores = softmax(logits); non_bg = scores[0] < thr; seg_id = argmax(logits) - 1 (first occurrence)
__global__ __launch_bounds__(256) void seg_epilogue_kernel(const float* __restrict__ logits, const int* __restrict__ lens,
                                                           int n_max, int c, float thr, float* __restrict__ scores_out,
                                                           int* __restrict__ seg_id, int* __restrict__ non_bg,
                                                           int* __restrict__ n_non_bg) {
    __shared__ int s_cnt[4];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = lens ? lens[b] : n_max;
    int kept = 0;
    // 16 tokens per workgroup (4 per wave): one atomic per workgroup instead of one per token
    for (int it = 0; it < 4; ++it) {
        const int n = blockIdx.x * 16 + it * 4 + wave;
        if (n >= len) continue;           // wave-uniform
        const float* x = logits + ((size_t)b * n_max + n) * c;
        float mx = -INFINITY;
        int mi = 0x7fffffff;
        for (int j = lane; j < c; j += 64) {
            const float v = x[j];
            if (v > mx) { mx = v; mi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mx, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
        }
        float sum = 0.f;
        for (int j = lane; j < c; j += 64) sum += expf(x[j] - mx);
        sum = wave_sum(sum);
        if (scores_out) {
            float* so = scores_out + ((size_t)b * n_max + n) * c;
            for (int j = lane; j < c; j += 64) so[j] = expf(x[j] - mx) / sum;
        }
        const float bg = expf(x[0] - mx) / sum;
        const int keep = bg < thr;
        kept += keep;
        if (lane == 0) {
            seg_id[(size_t)b * n_max + n] = mi - 1;
            non_bg[(size_t)b * n_max + n] = keep;
        }
    }
    if (lane == 0) s_cnt[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (tot) atomicAdd(&n_non_bg[b], tot);
    }
}

// ---- full descending sort of each row (value desc, index asc): bitonic in LDS, one workgroup per row
constexpr int SORT_MAX = 1024;
__global__ __launch_bounds__(256) void row_sort_kernel(const float* __restrict__ x, int ld, int c, float* __restrict__ vals,
                                                       long long* __restrict__ idx) {
    __shared__ unsigned long long keys[SORT_MAX];
    const int row = blockIdx.x;
    int p2 = 1;
    while (p2 < c) p2 <<= 1;
    for (int j = threadIdx.x; j < p2; j += 256) {
        unsigned long long k = 0ull;     // pads sort last
        if (j < c) {
            unsigned u = __float_as_uint(x[(size_t)row * ld + j]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving float -> uint
            k = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j);
            if (k == 0ull) k = 1ull;
        }
        keys[j] = k;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= p2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < p2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool desc = (i & k2) == 0;
                    if (desc ? (a < bb) : (a > bb)) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int j = threadIdx.x; j < c; j += 256) {
        const unsigned long long k = keys[j];
        unsigned u = (unsigned)(k >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        vals[(size_t)row * c + j] = __uint_as_float(u);
        idx[(size_t)row * c + j] = (long long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
    }
}

// ---- row top-2 (largest or smallest), one wave per row; ties -> lowest index ---------------------------
__global__ __launch_bounds__(256) void row_top2_kernel(const float* __restrict__ x, int ld, long long stride,
                                                       const int* __restrict__ row_lens, const int* __restrict__ col_lens,
                                                       int m_max, int n_max, int largest, float* __restrict__ v0,
                                                       float* __restrict__ v1, long long* __restrict__ i0) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = row_lens ? row_lens[b] : m_max, n = col_lens ? col_lens[b] : n_max;
    if (row >= m) return;
    const float* p = x + b * stride + (size_t)row * ld;
    const float sgn = largest ? 1.f : -1.f;
    float a0 = -INFINITY, a1 = -INFINITY;     // top-1 / top-2 of sgn * x
    int j0 = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float v = sgn * p[j];
        if (v > a0) { a1 = a0; a0 = v; j0 = j; }
        else if (v > a1) a1 = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float b0 = __shfl_xor(a0, o, 64), b1 = __shfl_xor(a1, o, 64);
        const int k0 = __shfl_xor(j0, o, 64);
        if (b0 > a0 || (b0 == a0 && k0 < j0)) { a1 = fmaxf(a0, b1); a0 = b0; j0 = k0; }
        else a1 = fmaxf(a1, b0);
    }
    if (lane == 0) {
        v0[(size_t)b * m_max + row] = sgn * a0;
        if (v1) v1[(size_t)b * m_max + row] = sgn * a1;
        i0[(size_t)b * m_max + row] = (n > 0) ? j0 : -1;
    }
}

// ---- projection refinement matching (SingleMap3D.refine_pose_by_projection, singlemap3d.py:416-433) ----
// dist[i][j] = sqrt(2 - 2 sim[i][j] + 1e-6) + (||kpt_i - uv_j|| >= 2*thr ? 100 : 0); row top-2 smallest.
__global__ __launch_bounds__(256) void proj_top2_kernel(const float* __restrict__ sim, int ld, const float* __restrict__ kpts,
                                                        const float* __restrict__ uv, int m, int n, float range,
                                                        float* __restrict__ d0, float* __restrict__ d1,
                                                        long long* __restrict__ i0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float kx = kpts[row * 2], ky = kpts[row * 2 + 1];
    const float* p = sim + (size_t)row * ld;
    float a0 = INFINITY, a1 = INFINITY;
    int j0 = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float ex = kx - uv[j], ey = ky - uv[n + j];
        const float pe = sqrtf(ex * ex + ey * ey);
        float v = sqrtf((2.f - 2.f * p[j]) + 1e-6f);
        if (pe >= range) v = v + 100.f;
        if (v < a0) { a1 = a0; a0 = v; j0 = j; }
        else if (v < a1) a1 = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float b0 = __shfl_xor(a0, o, 64), b1 = __shfl_xor(a1, o, 64);
        const int k0 = __shfl_xor(j0, o, 64);
        if (b0 < a0 || (b0 == a0 && k0 < j0)) { a1 = fminf(a0, b1); a0 = b0; j0 = k0; }
        else a1 = fminf(a1, b0);
    }
    if (lane == 0) { d0[row] = a0; d1[row] = a1; i0[row] = (n > 0) ? j0 : -1; }
}

// ---- landmark vote of MultiMap3D.process_segmentations (multimap3d.py:348-379) on the device ---------------------------
// idx / vals [n][c]: every token's classes sorted by score (row_sort_kernel).  Rank k = 0, 1, ...: the landmarks that tokens
// put at sorted position k, skipping background (0) and landmarks already seen at an earlier rank, ordered by the number of
// such tokens (descending; equal counts in ascending landmark id — Python's stable sort over np.unique's ascending ids),
// until `topk` landmarks are collected.  One workgroup: class histograms in LDS, winners by repeated block arg-max.
constexpr int VOTE_MAX_C = 1024;

__global__ __launch_bounds__(1024) void seg_vote_kernel(const long long* __restrict__ idx, int n, int c, int topk,
                                                        int* __restrict__ win_sid, int* __restrict__ win_rank,
                                                        int* __restrict__ win_cnt, int* __restrict__ n_win) {
    __shared__ int cnt[VOTE_MAX_C];
    __shared__ unsigned char used[VOTE_MAX_C];
    __shared__ long long wbest[16];
    __shared__ int nsel, left;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < VOTE_MAX_C) used[tid] = 0;
    if (tid == 0) nsel = 0;
    __syncthreads();
    for (int k = 0; k < c; ++k) {
        if (nsel >= topk) break;
        if (tid < VOTE_MAX_C) cnt[tid] = 0;
        __syncthreads();
        for (int t = tid; t < n; t += 1024) atomicAdd(&cnt[(int)idx[(size_t)t * c + k]], 1);
        __syncthreads();
        // candidates of this rank: seen here, not background, not used before; all of them count as used from now on
        const bool cand0 = tid < c && tid != 0 && cnt[tid] > 0 && !used[tid];
        if (tid < c && cnt[tid] > 0) used[tid] = 1;
        bool cand = cand0;
        for (;;) {
            // block arg-max of (count, lowest id): key = count * 2048 + (2047 - id)
            long long key = cand ? ((long long)cnt[tid] * 2048 + (2047 - tid)) : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 64));
            if (lane == 0) wbest[wave] = key;
            __syncthreads();
            long long best = -1;
            for (int w = 0; w < 16; ++w) best = max(best, wbest[w]);
            if (tid == 0) left = (best >= 0);
            if (best >= 0) {
                const int sid = 2047 - (int)(best % 2048);
                if (tid == sid) {
                    cand = false;
                    const int o = nsel;
                    win_sid[o] = sid; win_rank[o] = k; win_cnt[o] = cnt[sid];
                    nsel = o + 1;
                }
            }
            __syncthreads();
            if (!left || nsel >= topk) break;
        }
        __syncthreads();
    }
    if (tid == 0) *n_win = nsel;
}

// per winner (one wave each): its tokens in ascending order and the mean of their rank-k scores (fixed summation order)
__global__ __launch_bounds__(64) void seg_vote_gather_kernel(const long long* __restrict__ idx, const float* __restrict__ vals,
                                                             int n, int c, const int* __restrict__ win_sid,
                                                             const int* __restrict__ win_rank, const int* __restrict__ n_win,
                                                             int* __restrict__ tokens, float* __restrict__ mean) {
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= *n_win) return;
    const int sid = win_sid[w], k = win_rank[w];
    int base = 0;
    double part = 0.0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const bool f = t < n && (int)idx[(size_t)t * c + k] == sid;
        const unsigned long long bal = __ballot(f);
        if (f) {
            tokens[(size_t)w * n + base + __popcll(bal & ((1ull << lane) - 1ull))] = t;
            part += (double)vals[(size_t)t * c + k];
        }
        base += __popcll(bal);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) mean[w] = (float)(part / (double)base);
}

// Same with the projected points in double precision: the reference projects in float64 (numpy poses / intrinsics) and
// float32 keypoint - float64 projection promotes the pixel error and its `>= 2 * threshold` test to float64.
__global__ __launch_bounds__(256) void proj_top2_f64_kernel(const float* __restrict__ sim, int ld, const float* __restrict__ kpts,
                                                            const double* __restrict__ uv, int ldu, int m, int n, double range,
                                                            float* __restrict__ d0, float* __restrict__ d1,
                                                            long long* __restrict__ i0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const double kx = (double)kpts[row * 2], ky = (double)kpts[row * 2 + 1];
    const float* p = sim + (size_t)row * ld;
    float a0 = INFINITY, a1 = INFINITY;
    int j0 = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const double ex = kx - uv[j], ey = ky - uv[ldu + j];
        const double pe = sqrt(ex * ex + ey * ey);
        float v = sqrtf((2.f - 2.f * p[j]) + 1e-6f);
        if (pe >= range) v = v + 100.f;
        if (v < a0) { a1 = a0; a0 = v; j0 = j; }
        else if (v < a1) a1 = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float b0 = __shfl_xor(a0, o, 64), b1 = __shfl_xor(a1, o, 64);
        const int k0 = __shfl_xor(j0, o, 64);
        if (b0 < a0 || (b0 == a0 && k0 < j0)) { a1 = fminf(a0, b1); a0 = b0; j0 = k0; }
        else a1 = fminf(a1, b0);
    }
    if (lane == 0) { d0[row] = a0; d1[row] = a1; i0[row] = (n > 0) ? j0 : -1; }
}

// ---- projection of the map points (SingleMap3D.refine_pose_by_projection, singlemap3d.py:405-415), float64 like the reference:
// p = K (Tcw [X 1])[:3];  u = p0 / p2, v = p1 / p2;  keep = 0 < p2 < 100 and 0 <= u < w and 0 <= v < h
__global__ void proj_points_kernel(const double* __restrict__ xyz, const double* __restrict__ K, const double* __restrict__ T,
                                   int n, double imw, double imh, double* __restrict__ uvd, int* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    double c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) c[r] = ((T[r * 4] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3];
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) p[r] = (K[r * 3] * c[0] + K[r * 3 + 1] * c[1]) + K[r * 3 + 2] * c[2];
    const double u = p[0] / p[2], v = p[1] / p[2];
    uvd[i] = u;
    uvd[n + i] = v;
    uvd[2 * n + i] = p[2];
    mask[i] = (p[2] > 0.0) && (p[2] < 100.0) && (u >= 0.0) && (u < imw) && (v >= 0.0) && (v < imh);
}

// ordered stream compaction of the survivors (the reference's boolean indexing keeps the original order): one workgroup walks the
// array in chunks of 1024 with a block scan; writes the surviving original indices, their (u, v) and the count
__global__ __launch_bounds__(1024) void proj_compact_kernel(const int* __restrict__ mask, const double* __restrict__ uvd, int n,
                                                            int* __restrict__ keep_idx, double* __restrict__ uv_keep,
                                                            int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        const int f = (i < n) ? (mask[i] != 0) : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        const int b = base;
        if (f) {
            const int o = b + woff + before;
            keep_idx[o] = i;
            uv_keep[o] = uvd[i];
            uv_keep[n + o] = uvd[n + i];
        }
        __syncthreads();
        if (tid == 0) base = b + tot;
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

}  // namespace

extern "C" int pram_seg_vote(const long long* sorted_ids, const float* sorted_vals, int n, int c, int topk, int* win_sid,
                             int* win_rank, int* win_count, int* n_win, int* tokens, float* mean_score, void* stream) {
    PRAM_REQUIRE(sorted_ids && sorted_vals && win_sid && win_rank && win_count && n_win && tokens && mean_score, "pram_seg_vote: null pointer");
    PRAM_REQUIRE(c > 0 && c <= VOTE_MAX_C && topk > 0 && n >= 0, "pram_seg_vote: needs 0 < classes <= %d, topk > 0", VOTE_MAX_C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_vote_kernel, dim3(1), dim3(1024), 0, st, sorted_ids, n, c, topk, win_sid, win_rank, win_count, n_win);
    hipLaunchKernelGGL(seg_vote_gather_kernel, dim3(topk), dim3(64), 0, st, sorted_ids, sorted_vals, n, c, win_sid, win_rank, n_win,
                       tokens, mean_score);
    return pram_launch_status("pram_seg_vote");
}

extern "C" int pram_project_points_f64(const double* xyz, const double* K, const double* Tcw, int n, double im_w, double im_h,
                                       double* uvd, int* mask, int* keep_idx, double* uv_keep, int* count, void* stream) {
    PRAM_REQUIRE(xyz && K && Tcw && uvd && mask && keep_idx && uv_keep && count, "pram_project_points_f64: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n > 0) hipLaunchKernelGGL(proj_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, xyz, K, Tcw, n, im_w, im_h, uvd, mask);
    hipLaunchKernelGGL(proj_compact_kernel, dim3(1), dim3(1024), 0, st, mask, uvd, n, keep_idx, uv_keep, count);
    return pram_launch_status("pram_project_points_f64");
}

extern "C" int pram_proj_dist_top2_f64uv(const float* sim, int ld, const float* kpts, const double* proj_uv, int ldu, int m, int n,
                                         double range, float* d0, float* d1, long long* i0, void* stream) {
    PRAM_REQUIRE(sim && kpts && proj_uv && d0 && d1 && i0, "pram_proj_dist_top2_f64uv: null pointer");
    if (m == 0) return PRAM_OK;
    hipLaunchKernelGGL(proj_top2_f64_kernel, dim3(cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, sim, ld, kpts, proj_uv, ldu, m, n,
                       range, d0, d1, i0);
    return pram_launch_status("pram_proj_dist_top2_f64uv");
}

extern "C" int pram_proj_dist_top2_f32(const float* sim, int ld, const float* kpts, const float* proj_uv, int m, int n,
                                       float range, float* d0, float* d1, long long* i0, void* stream) {
    PRAM_REQUIRE(sim && kpts && proj_uv && d0 && d1 && i0, "pram_proj_dist_top2_f32: null pointer");
    if (m == 0) return PRAM_OK;
    hipLaunchKernelGGL(proj_top2_kernel, dim3(cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, sim, ld, kpts, proj_uv, m, n, range,
                       d0, d1, i0);
    return pram_launch_status("pram_proj_dist_top2_f32");
}

extern "C" int pram_resize_bilinear_f32(const float* in, float* out, int planes, int h, int w, int oh, int ow, void* stream) {
    PRAM_REQUIRE(in && out && h > 0 && w > 0 && oh > 0 && ow > 0, "pram_resize_bilinear_f32: bad arguments");
    if (planes == 0) return PRAM_OK;
    const float rh = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float rw = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const long long total = (long long)planes * oh * ow;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       planes, h, w, oh, ow, rh, rw);
    return pram_launch_status("pram_resize_bilinear_f32");
}

extern "C" int pram_seg_epilogue_f32(const float* logits, const int* lens, int batch, int n_max, int n_class, float bg_threshold,
                                     float* seg_scores, int* seg_ids, int* non_bg_mask, int* n_non_bg, void* stream) {
    PRAM_REQUIRE(logits && seg_ids && non_bg_mask && n_non_bg && n_class > 0, "pram_seg_epilogue_f32: bad arguments");
    if (batch == 0 || n_max == 0) return PRAM_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(n_non_bg, 0, (size_t)batch * 4, st) != hipSuccess) {
        pram_set_error("pram_seg_epilogue_f32: memset failed");
        return PRAM_E_LAUNCH;
    }
    hipLaunchKernelGGL(seg_epilogue_kernel, dim3(cdiv(n_max, 16), batch), dim3(256), 0, st, logits, lens, n_max, n_class,
                       bg_threshold, seg_scores, seg_ids, non_bg_mask, n_non_bg);
    return pram_launch_status("pram_seg_epilogue_f32");
}

extern "C" int pram_row_sort_desc_f32(const float* x, int ld, int rows, int cols, float* vals, long long* idx, void* stream) {
    PRAM_REQUIRE(x && vals && idx, "pram_row_sort_desc_f32: null pointer");
    PRAM_REQUIRE(cols > 0 && cols <= SORT_MAX, "pram_row_sort_desc_f32: cols=%d not in (0, %d]", cols, SORT_MAX);
    if (rows == 0) return PRAM_OK;
    hipLaunchKernelGGL(row_sort_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, cols, vals, idx);
    return pram_launch_status("pram_row_sort_desc_f32");
}

extern "C" int pram_row_top2_f32(const float* x, int ld, long long stride, const int* row_lens, const int* col_lens, int batch,
                                 int m_max, int n_max, int largest, float* v0, float* v1, long long* i0, void* stream) {
    PRAM_REQUIRE(x && v0 && i0, "pram_row_top2_f32: null pointer");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    hipLaunchKernelGGL(row_top2_kernel, dim3(cdiv(m_max, 4), batch), dim3(256), 0, (hipStream_t)stream, x, ld, stride, row_lens,
                       col_lens, m_max, n_max, largest, v0, v1, i0);
    return pram_launch_status("pram_row_top2_f32");
}

// ---------------------------------------------------------------- result record + fills (host-side glue of the query pipeline)
// Fixed-size per-query result record [B][k][6] fp32: x, y, score, landmark id, match index, match score — what the one
// all-gather of SURVEY.md §8(e) carries.  landmark / matches may be absent (NULL: 0 / -1 and 0); only the first km keypoints
// of a query went through the matcher (km <= k: the secondary shape of SURVEY.md §8(d) matches the 512 best of 2048).
__global__ __launch_bounds__(256) void pack_record_kernel(const float* __restrict__ kpts, const float* __restrict__ scores,
                                                          const int* __restrict__ landmark, const long long* __restrict__ matches0,
                                                          const float* __restrict__ mscores0, int k, int km, long long total,
                                                          float* __restrict__ rec) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (b, keypoint)
    if (i >= total) return;
    const long long b = i / k;
    const int j = (int)(i - b * k);
    float* r = rec + i * 6;
    r[0] = kpts[i * 2 + 0];
    r[1] = kpts[i * 2 + 1];
    r[2] = scores[i];
    r[3] = landmark ? (float)landmark[i] : 0.f;
    const bool m = matches0 != nullptr && j < km;
    r[4] = matches0 ? (m ? (float)matches0[b * km + j] : -1.f) : 0.f;
    r[5] = m ? mscores0[b * km + j] : 0.f;
}

extern "C" int pram_pack_record_f32(const float* kpts, const float* scores, const int* landmark, const long long* matches0,
                                    const float* mscores0, int batch, int k, int km, float* rec, void* stream) {
    PRAM_REQUIRE(kpts && scores && rec, "pram_pack_record_f32: null pointer");
    PRAM_REQUIRE((matches0 == nullptr) == (mscores0 == nullptr) && km >= 0 && km <= k, "pram_pack_record_f32: matches and scores go together, km <= k");
    const long long total = (long long)batch * k;
    if (total == 0) return PRAM_OK;
    hipLaunchKernelGGL(pack_record_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kpts, scores,
                       landmark, matches0, mscores0, k, km, total, rec);
    return pram_launch_status("pram_pack_record_f32");
}

// ---------------------------------------------------------------- frame staging (localization/loc_by_rec_online.py:86-106)
// The reference prepares a query frame on the host and device in three steps: img / 255 (numpy, float64), .cuda().float(), and
// tvf.Normalize(mean, std) = sub_(mean).div_(std) per channel of the HWC -> CHW permuted tensor (channels stay in cv2's order).  All
// three are functions of one byte and the channel, so the device side is a table lookup: lut[c][v] is built ON THE HOST with the
// reference's own operations (ops.frame_lut), the kernel only moves data — uint8 [b][h][w][3] in, fp32 [b][3][h][w] out: 3 B read
// and 12 B written per pixel.  One thread = 4 consecutive pixels of a row (12 bytes in as three dwords, one float4 per channel out).
__global__ __launch_bounds__(256) void stage_frames_kernel(const uint8_t* __restrict__ in, const float* __restrict__ lut,
                                                            float* __restrict__ out, long long hw, long long quads) {
    __shared__ float sl[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) sl[i] = lut[i];
    __syncthreads();
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // (frame, quad of pixels)
    if (q >= quads) return;
    const long long per = hw / 4;
    const long long b = q / per, p0 = (q - b * per) * 4;
    const unsigned* src = reinterpret_cast<const unsigned*>(in + (b * hw + p0) * 3);
    const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
    unsigned char v[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (w0 >> (8 * i)) & 0xff; v[4 + i] = (w1 >> (8 * i)) & 0xff; v[8 + i] = (w2 >> (8 * i)) & 0xff; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 o = make_float4(sl[c * 256 + v[c]], sl[c * 256 + v[3 + c]], sl[c * 256 + v[6 + c]], sl[c * 256 + v[9 + c]]);
        *reinterpret_cast<float4*>(out + (b * 3 + c) * hw + p0) = o;
    }
}

extern "C" int pram_stage_frames_u8(const void* frames_hwc3, const float* lut, float* out_nchw, int batch, int h, int w, void* stream) {
    PRAM_REQUIRE(frames_hwc3 && lut && out_nchw, "pram_stage_frames_u8: null pointer");
    PRAM_REQUIRE(batch >= 0 && h > 0 && w > 0 && ((long long)h * w) % 4 == 0, "pram_stage_frames_u8: h * w must be a multiple of 4");
    PRAM_REQUIRE(((size_t)frames_hwc3 & 3) == 0 && ((size_t)out_nchw & 15) == 0, "pram_stage_frames_u8: frames 4-byte aligned, output 16-byte aligned");
    if (batch == 0) return PRAM_OK;
    const long long hw = (long long)h * w, quads = (long long)batch * hw / 4;
    hipLaunchKernelGGL(stage_frames_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)frames_hwc3, lut, out_nchw, hw, quads);
    return pram_launch_status("pram_stage_frames_u8");
}

/* count 32-bit words at dst <- value (hipMemsetD32Async on the stream): the "zeros / full" initialisations of the host-side
   glue without a framework kernel. */
extern "C" int pram_fill_u32(void* dst, unsigned int value, size_t count, void* stream) {
    if (count == 0) return PRAM_OK;
    PRAM_REQUIRE(dst, "pram_fill_u32: null pointer");
    if (hipMemsetD32Async((hipDeviceptr_t)dst, (int)value, count, (hipStream_t)stream) != hipSuccess) return pram_launch_status("pram_fill_u32");
    return PRAM_OK;
}
