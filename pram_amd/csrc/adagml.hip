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
    int ld_logit;      // stride of the logits in floats (the pooling head's last Linear is padded to 4 outputs: column 0 of [rows][4])
};

// Pass 1, one workgroup per token set (latency-bound bookkeeping): confidences, the two counts, and the source row of every
// surviving destination row: row_map[s][d] = t for d < lens_out[s].
__global__ __launch_bounds__(PT) void prune_scan_kernel(PruneArgs p) {
    __shared__ int sbuf[17];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int len = p.lens_in ? p.lens_in[s] : p.t_max;
    const bool do_prune = len >= p.n_min;
    const float* lg = p.logit + (size_t)s * p.t_max * p.ld_logit;
    int* map = p.row_map + (size_t)s * p.t_max;
    int kept = 0, below = 0;
    for (int base = 0; base < len; base += PT) {
        const int t = base + tid;
        bool keep = false;
        int bl = 0;
        if (t < len) {
            const float conf = 1.0f / (1.0f + expf(-lg[(size_t)t * p.ld_logit]));    // torch.sigmoid
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

// Per-layer bookkeeping of the batched AdaGML loop (nets/adagml.py:352-380 per pair, here for B pairs at once and without a host
// read): the pruned token counts are committed for the pairs that are still active, check_if_stop (adagml.py:522-531) is
// evaluated in the same fp32 arithmetic, and the pairs that stop at this layer commit their token counts / survivor ids /
// layer index.  One workgroup per token set s (pair s % B); every workgroup derives its pair's state from the INPUT buffers
// and writes the OUTPUT buffers (ping-pong: no workgroup reads what another writes).
struct StateArgs {
    const int* active_in; int* active_out;
    const int* lens_in; int* lens_out;                  // [2B]
    const int* lens_new; const int* n_below;            // [2B] from the prune kernel, or nullptr on the first layer
    const float* num_points;                            // [B] m + n of the original sets
    int* tiny; int* stop_layer;                         // [B]
    int* lens_final; int* lens_stop; int* lens_eff;     // [2B]
    const int* ind; int* ind_final;                     // [2B][T]
    int pairs, t_max, layer, last;
};

__global__ __launch_bounds__(256) void layer_state_kernel(StateArgs p) {
    const int s = blockIdx.x, B = p.pairs, b = s % B;
    const bool act = p.active_in[b] != 0;
    int l0 = p.lens_in[b], l1 = p.lens_in[B + b];
    bool stop = false, tiny = false;
    if (p.lens_new) {
        if (act) { l0 = p.lens_new[b]; l1 = p.lens_new[B + b]; }
        tiny = act && (l0 <= 5 || l1 <= 5);
        const float below = (float)(p.n_below[b] + p.n_below[B + b]);
        stop = act && ((1.0f - below / p.num_points[b]) > 0.95f);      // check_if_stop, fp32 like the reference's tensor arithmetic
    }
    if (p.last) stop = act;                                            // loop exhausted: the last layer's descriptors (adagml.py:374)
    const bool commit = p.lens_new != nullptr || p.last;               // layer 0 never stops a pair (adagml.py:363: ni >= 1)
    if (!commit) stop = false;
    const int mine = s < B ? l0 : l1;
    if (threadIdx.x == 0) {
        p.lens_out[s] = mine;
        p.lens_stop[s] = stop ? mine : 0;
        if (stop) p.lens_final[s] = mine;
        p.lens_eff[s] = (act && !stop) ? mine : 0;
        if (s < B) {
            p.active_out[b] = (act && !stop) ? 1 : 0;
            if (tiny) p.tiny[b] = 1;
            if (stop) p.stop_layer[b] = p.layer;
        }
    }
    if (stop)
        for (int t = threadIdx.x; t < p.t_max; t += blockDim.x) p.ind_final[(size_t)s * p.t_max + t] = p.ind[(size_t)s * p.t_max + t];
}

// score4[s][t] = (col_self[s][t], col_cross[s][t], 0, 0): the pooling head's two scores per token (adagml.py:132), padded to the
// 16-byte rows its first Linear (K 2 -> 4) reads
__global__ __launch_bounds__(256) void scores4_kernel(const float* __restrict__ col_self, const float* __restrict__ col_cross,
                                                      float4* __restrict__ out, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = make_float4(col_self[i], col_cross[i], 0.f, 0.f);
}

}  // namespace

extern "C" int pram_adagml_layer_state(const int* active_in, int* active_out, const int* lens_in, int* lens_out, const int* lens_new,
                                       const int* n_below, const float* num_points, int* tiny, int* stop_layer, int* lens_final,
                                       int* lens_stop, int* lens_eff, const int* ind, int* ind_final, int pairs, int t_max, int layer,
                                       int last, void* stream) {
    PRAM_REQUIRE(active_in && active_out && lens_in && lens_out && num_points && tiny && stop_layer && lens_final && lens_stop && lens_eff &&
                 ind && ind_final, "pram_adagml_layer_state: null pointer");
    PRAM_REQUIRE((lens_new == nullptr) == (n_below == nullptr), "pram_adagml_layer_state: lens_new and n_below go together");
    PRAM_REQUIRE(active_in != active_out && lens_in != lens_out, "pram_adagml_layer_state: state buffers must ping-pong");
    if (pairs == 0) return PRAM_OK;
    StateArgs p{active_in, active_out, lens_in, lens_out, lens_new, n_below, num_points, tiny, stop_layer, lens_final, lens_stop, lens_eff,
                ind, ind_final, pairs, t_max, layer, last};
    hipLaunchKernelGGL(layer_state_kernel, dim3(2 * pairs), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_adagml_layer_state");
}

extern "C" int pram_adagml_scores4_f32(const float* col_self, const float* col_cross, float* score4, long long tokens, void* stream) {
    PRAM_REQUIRE(col_self && col_cross && score4, "pram_adagml_scores4_f32: null pointer");
    if (tokens == 0) return PRAM_OK;
    hipLaunchKernelGGL(scores4_kernel, dim3((unsigned)((tokens + 255) / 256)), dim3(256), 0, (hipStream_t)stream, col_self, col_cross,
                       reinterpret_cast<float4*>(score4), tokens);
    return pram_launch_status("pram_adagml_scores4_f32");
}

static int adagml_prune_impl(const float* conf_logit, int ld_logit, float thr, int n_min_tokens, const int* lens_in,
                             const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                             float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                             int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream);

/* pram_adagml_prune_f32 with the logits at stride ld_logit (floats): column 0 of the padded [rows][4] output of the pooling
   head's last Linear, read in place. */
extern "C" int pram_adagml_prune_ld_f32(const float* conf_logit, int ld_logit, float thr, int n_min_tokens, const int* lens_in,
                                        const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                                        float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                                        int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream) {
    PRAM_REQUIRE(ld_logit >= 1, "pram_adagml_prune_ld_f32: bad logit stride");
    return adagml_prune_impl(conf_logit, ld_logit, thr, n_min_tokens, lens_in, x_in, cos_in, sin_in, ind_in, x_out, cos_out, sin_out, ind_out,
                             lens_out, n_below, conf_out, row_map, sets, t_max, ldx, stream);
}

extern "C" int pram_adagml_prune_f32(const float* conf_logit, float thr, int n_min_tokens, const int* lens_in,
                                     const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                                     float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                                     int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream) {
    return adagml_prune_impl(conf_logit, 1, thr, n_min_tokens, lens_in, x_in, cos_in, sin_in, ind_in, x_out, cos_out, sin_out, ind_out,
                             lens_out, n_below, conf_out, row_map, sets, t_max, ldx, stream);
}

static int adagml_prune_impl(const float* conf_logit, int ld_logit, float thr, int n_min_tokens, const int* lens_in,
                             const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                             float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                             int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream) {
    PRAM_REQUIRE(conf_logit && x_in && cos_in && sin_in && ind_in && x_out && cos_out && sin_out && ind_out && lens_out && n_below && row_map,
                 "pram_adagml_prune_f32: null pointer");
    PRAM_REQUIRE(t_max <= T_MAX && ldx % 4 == 0, "pram_adagml_prune_f32: t_max=%d exceeds %d or ldx not a multiple of 4", t_max, T_MAX);
    PRAM_REQUIRE(x_in != x_out, "pram_adagml_prune_f32: in-place compaction is not supported (ping-pong the buffers)");
    if (sets == 0 || t_max == 0) return PRAM_OK;
    PruneArgs p{conf_logit, thr, n_min_tokens, lens_in, x_in, cos_in, sin_in, ind_in, x_out, cos_out, sin_out, ind_out,
                lens_out, n_below, conf_out, row_map, t_max, ldx, ld_logit};
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
