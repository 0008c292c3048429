// AdaGML token pruning and result scatter (nets/adagml.py:354-372,382-396,516-531).
// The bookkeeping (scan) runs one workgroup per token set; the row copies run on the whole chip.
#include "common.h"
#include <math.h>

namespace {

constexpr int PT = 1024;
constexpr int T_MAX = 8192;

__device__ __forceinline__ int blk_scan(int v, int* sbuf, int* total) {
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
        for (int i = 0; i < PT / 64; ++i) { const int t = sbuf[i]; sbuf[i] = run; run += t; }
        sbuf[16] = run;
    }
    __syncthreads();
    const int res = sbuf[wave] + inc - v;
    *total = sbuf[16];
    __syncthreads();
    return res;
}

struct PruneArgs {
    const float* logit; float thr; int n_min;
    const int* lens_in; const float* x_in; const float* cos_in; const float* sin_in; const int* ind_in;
    float* x_out; float* cos_out; float* sin_out; int* ind_out; int* lens_out; int* n_below; float* conf_out;
    int* row_map;
    int t_max, ldx;
};

// Pass 1, one workgroup per token set (latency-bound bookkeeping): confidences, the two counts, and the source row of every
// surviving destination row: row_map[s][d] = t for d < lens_out[s].
__global__ __launch_bounds__(PT) void prune_scan_kernel(PruneArgs p) {
    __shared__ int sbuf[17];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int len = p.lens_in ? p.lens_in[s] : p.t_max;
    const bool do_prune = len >= p.n_min;
    const float* lg = p.logit + (size_t)s * p.t_max;
    int* map = p.row_map + (size_t)s * p.t_max;
    int kept = 0, below = 0;
    for (int base = 0; base < len; base += PT) {
        const int t = base + tid;
        bool keep = false;
        int bl = 0;
        if (t < len) {
            const float conf = 1.0f / (1.0f + expf(-lg[t]));    // torch.sigmoid
            if (p.conf_out) p.conf_out[(size_t)s * p.t_max + t] = conf;
            keep = do_prune ? (conf > p.thr) : true;
            bl = conf < p.thr;
        }
        int tot, totb;
        const int pos = blk_scan(keep ? 1 : 0, sbuf, &tot);
        blk_scan(bl, sbuf, &totb);
        if (keep) map[kept + pos] = t;
        kept += tot;
        below += totb;
    }
    if (tid == 0) {
        p.lens_out[s] = kept;
        p.n_below[s] = below;
    }
}

// Pass 2, the whole chip: one wave per destination row (x row = ldx floats, cos / sin 32 floats each, the original token id).
constexpr int GR = 16;      // destination rows per workgroup (4 waves x 4 rows)
__global__ __launch_bounds__(256) void prune_gather_kernel(PruneArgs p) {
    const int s = blockIdx.y;
    const int kept = p.lens_out[s];
    const int d0 = blockIdx.x * GR;
    if (d0 >= kept) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t sx = (size_t)s * p.t_max;
#pragma unroll
    for (int i = 0; i < GR / 4; ++i) {
        const int d = d0 + wave * (GR / 4) + i;
        if (d >= kept) break;
        const int t = p.row_map[sx + d];
        const float* xi = p.x_in + (sx + t) * p.ldx;
        float* xo = p.x_out + (sx + d) * p.ldx;
        for (int c = lane * 4; c < p.ldx; c += 256) *reinterpret_cast<float4*>(xo + c) = *reinterpret_cast<const float4*>(xi + c);
        if (lane < 8) *reinterpret_cast<float4*>(p.cos_out + (sx + d) * 32 + lane * 4) =
                          *reinterpret_cast<const float4*>(p.cos_in + (sx + t) * 32 + lane * 4);
        else if (lane < 16) *reinterpret_cast<float4*>(p.sin_out + (sx + d) * 32 + (lane - 8) * 4) =
                                *reinterpret_cast<const float4*>(p.sin_in + (sx + t) * 32 + (lane - 8) * 4);
        if (lane == 16) p.ind_out[sx + d] = p.ind_in[sx + t];
    }
}

// matches0_full[ind0[i]] = ind1[matches0[i]] (valid only); scores_full[ind0[i]] = ms0[i]
__global__ void scatter_kernel(const long long* __restrict__ m0, const float* __restrict__ ms0, const int* __restrict__ ind0,
                               const int* __restrict__ ind1, const int* __restrict__ lens0, int t_max, int m_full,
                               long long* __restrict__ out_m, float* __restrict__ out_s) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens0 ? lens0[b] : t_max;
    if (i >= len) return;
    const int row = ind0[(size_t)b * t_max + i];
    const long long j = m0[(size_t)b * t_max + i];
    out_s[(size_t)b * m_full + row] = ms0[(size_t)b * t_max + i];
    if (j >= 0) out_m[(size_t)b * m_full + row] = ind1[(size_t)b * t_max + j];
}

}  // namespace

extern "C" int pram_adagml_prune_f32(const float* conf_logit, float thr, int n_min_tokens, const int* lens_in,
                                     const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                                     float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                                     int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream) {
    PRAM_REQUIRE(conf_logit && x_in && cos_in && sin_in && ind_in && x_out && cos_out && sin_out && ind_out && lens_out && n_below && row_map,
                 "pram_adagml_prune_f32: null pointer");
    PRAM_REQUIRE(t_max <= T_MAX && ldx % 4 == 0, "pram_adagml_prune_f32: t_max=%d exceeds %d or ldx not a multiple of 4", t_max, T_MAX);
    PRAM_REQUIRE(x_in != x_out, "pram_adagml_prune_f32: in-place compaction is not supported (ping-pong the buffers)");
    if (sets == 0 || t_max == 0) return PRAM_OK;
    PruneArgs p{conf_logit, thr, n_min_tokens, lens_in, x_in, cos_in, sin_in, ind_in, x_out, cos_out, sin_out, ind_out,
                lens_out, n_below, conf_out, row_map, t_max, ldx};
    hipLaunchKernelGGL(prune_scan_kernel, dim3(sets), dim3(PT), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(prune_gather_kernel, dim3(cdiv(t_max, GR), sets), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_adagml_prune_f32");
}

extern "C" int pram_adagml_scatter_f32(const long long* matches0, const float* mscores0, const int* ind0, const int* ind1,
                                       const int* lens0, int batch, int t_max, int m_full, long long* out_matches,
                                       float* out_scores, void* stream) {
    PRAM_REQUIRE(matches0 && mscores0 && ind0 && ind1 && out_matches && out_scores, "pram_adagml_scatter_f32: null pointer");
    if (batch == 0 || t_max == 0) return PRAM_OK;
    hipLaunchKernelGGL(scatter_kernel, dim3(cdiv(t_max, 256), batch), dim3(256), 0, (hipStream_t)stream, matches0, mscores0,
                       ind0, ind1, lens0, t_max, m_full, out_matches, out_scores);
    return pram_launch_status("pram_adagml_scatter_f32");
}
