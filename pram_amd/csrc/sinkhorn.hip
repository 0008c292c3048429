// Plain-domain Sinkhorn on the dustbin-augmented score matrix + mutual-NN match extraction.
// Reference: sink_algorithm / sinkhorn nets/gml.py:27-46, dual_softmax :20-24,
// compute_matches :304-319 (identical copies in nets/gm.py and nets/adagml.py).
//
// HBM/L2-bound.  The reference streams the (m+1)x(n+1) matrix ~44 times (softmax, 2 x 20
// masked reductions with temporaries, final scaling).  Here each Sinkhorn iteration reads P
// ONCE: a wave owns a row, keeps it in registers, forms u_i = r_i / (P_i·v + eps) with a wave
// reduction and immediately accumulates P_ij * u_i into per-lane column accumulators; the
// per-wave column partials are summed by a small second kernel that also forms v.  Row blocks
// are contiguous per workgroup, so with the XCD round-robin each XCD keeps re-reading the same
// 1/8 of P from its own L2 across iterations (16.8 MB at 2049^2 -> 2.1 MB per 4-MiB L2).
// compute_matches is fused into the final pass (row max/argmax by wave reduction, column
// max/argmax by the same partial scheme); argmax ties resolve to the lowest index like
// torch.max(dim).
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr float SINK_EPS = 1e-8f;  // nets/gml.py:17
constexpr int NBLK = 32;           // row blocks per batch element -> 128 waves, 128 column partials

struct SinkWs {
    float* P;       // [batch][m_max+1][ldw]
    float* u;       // [batch][m_max+1]
    float* v;       // [batch][ldw]
    float* part;    // [batch][NBLK*4][ldw]   column partial sums / partial max values
    int* parti;     // [batch][NBLK*4][ldw]   column partial argmax
    float* rowval;  // [batch][m_max]
    int* rowidx;    // [batch][m_max]
    float* colval;  // [batch][n_max]
    int* colidx;    // [batch][n_max]
    float* rlse;    // [batch][m_max+1]  (dual-softmax only)
    float* clse;    // [batch][ldw]
    int ldw;
};

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static SinkWs carve(void* ws, int batch, int m_max, int n_max, size_t* total) {
    SinkWs w;
    w.ldw = ((n_max + 1 + 3) / 4) * 4;
    size_t off = 0;
    char* base = (char*)ws;
    auto take = [&](size_t bytes) { char* p = base + off; off += align256(bytes); return p; };
    w.P = (float*)take((size_t)batch * (m_max + 1) * w.ldw * 4);
    w.u = (float*)take((size_t)batch * (m_max + 1) * 4);
    w.v = (float*)take((size_t)batch * w.ldw * 4);
    w.part = (float*)take((size_t)batch * NBLK * 4 * w.ldw * 4);
    w.parti = (int*)take((size_t)batch * NBLK * 4 * w.ldw * 4);
    w.rowval = (float*)take((size_t)batch * m_max * 4);
    w.rowidx = (int*)take((size_t)batch * m_max * 4);
    w.colval = (float*)take((size_t)batch * n_max * 4);
    w.colidx = (int*)take((size_t)batch * n_max * 4);
    w.rlse = (float*)take((size_t)batch * (m_max + 1) * 4);
    w.clse = (float*)take((size_t)batch * w.ldw * 4);
    if (total) *total = off;
    return w;
}

// ---- row softmax of the augmented matrix (one wave per row), u, v <- 1 ------------------------
// mode 0: write softmax probabilities (Sinkhorn);  mode 1: write raw augmented scores and the row
// log-sum-exp (dual softmax).
template <int NV>
__global__ __launch_bounds__(256) void sink_init_kernel(const float* __restrict__ dist, int ldd, long long sdist,
                                                        const int* __restrict__ m_lens, const int* __restrict__ n_lens,
                                                        const float* __restrict__ bin, SinkWs w, int m_max, int n_max,
                                                        int mode) {
    // The row (at most NV * 256 columns: lane l owns columns l, l + 64, ...) is read ONCE and kept in registers; the first version
    // walked it three times (max, sum of exponentials, write) with two exponentials per element: 218 -> ~120 us at 16 pairs of
    // 2049 x 2049.  Every lane adds its exponentials in the same order as before, so the row sums keep their bits.
    constexpr int EPL = NV * 4;
    const int b = blockIdx.y;
    const int m = m_lens ? m_lens[b] : m_max, n = n_lens ? n_lens[b] : n_max;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i == 0)
        for (int j = lane; j < w.ldw; j += 64) w.v[(size_t)b * w.ldw + j] = 1.0f;
    // u <- 1 as well: with zero iterations the final pass returns the plain row softmax (u = v = 1), like the reference
    if (i <= m_max && lane == 0) w.u[(size_t)b * (m_max + 1) + i] = 1.0f;
    if (i > m) return;
    const float bs = bin[0];
    const float* src = dist + b * sdist + (size_t)min(i, m_max - 1) * ldd;      // the bin row (i == m) reads nothing it uses
    float* dst = w.P + ((size_t)b * (m_max + 1) + i) * w.ldw;
    const bool binrow = (i == m);
    float x[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) x[t] = src[min(lane + 64 * t, max(n - 1, 0))];      // unpredicated; masked below
    float mx = bs;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int j = lane + 64 * t;
        if (binrow || j >= n) x[t] = bs;          // the dust-bin column j == n (and the padding, never used)
        if (!binrow && j < n) mx = fmaxf(mx, x[t]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    if (mode == 0) {
        float e[EPL];
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            e[t] = expf(x[t] - mx);
            if (lane + 64 * t <= n) sum += e[t];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int j = lane + 64 * t;
            if (j < w.ldw) dst[j] = (j <= n) ? e[t] / sum : 0.f;
        }
    } else {
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            if (lane + 64 * t <= n) sum += expf(x[t] - mx);
        sum = wave_sum(sum);
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int j = lane + 64 * t;
            if (j < w.ldw) dst[j] = (j <= n) ? x[t] : -INFINITY;
        }
        if (lane == 0) w.rlse[(size_t)b * (m_max + 1) + i] = mx + logf(sum);
    }
}

// ---- one Sinkhorn iteration: u = r / (P v + eps), column partials of P^T u -----------------------
// PF = rows a wave requests ahead.  A row is one dependent chain (load -> dot -> wave reduction -> u_i -> column accumulation) and a
// one-pair launch is 32 workgroups with 16 rows per wave, so what matters there is that the row ahead is REALLY in flight.  It
// was not in the first version (28 us per iteration at 2049 x 2049 = 0.6 TB/s; now 14 us): (a) loads under a per-lane predicate
// sit in exec branches and the compiler waits with vmcnt(0) at the joins; (b) a store inside the loop (u_i) beside outstanding
// loads forces vmcnt(0) — loads and stores return out of order with respect to each other on gfx9; (c) a loop the compiler takes
// for divergent carries the row slots through copies, and a copy of a register a load is still writing is a wait; (d) v was
// copied to LDS by a scalar loop: nine dependent round trips.  More rows ahead (PF = 2, 4), one wave per workgroup on 128 CUs, and two
// rows reduced side by side (their six-step wave reductions and divisions interleaved) all measured the same or worse: 128 waves
// pull 17 MB at ~10 GB/s each whatever they keep in flight, and the 128 row groups are fixed by the bits (the order in which
// rows are added into a column partial).  The rows of a wave are reduced in the same order whatever PF is.
template <int NV, int PF>
__global__ __launch_bounds__(256) void sink_iter_kernel(const int* __restrict__ m_lens, const int* __restrict__ n_lens,
                                                        SinkWs w, int m_max, int n_max) {
    extern __shared__ __attribute__((aligned(16))) float sv[];
    const int b = blockIdx.y;
    const int m = m_lens ? m_lens[b] : m_max;
    constexpr int NTHR = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the row loop is wave-uniform
    const int blk = blockIdx.x;
    const int ldw = w.ldw;
    {   // v -> LDS, zero-padded to NV * 256 (no column guard below); every load is out before the first is stored — written as a
        // scalar loop this prologue was nine dependent round trips, half of the kernel at one pair
        constexpr int NQ = (NV * 64 + NTHR - 1) / NTHR;
        float4 q4[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) q4[t] = *reinterpret_cast<const float4*>(w.v + (size_t)b * ldw + min((tid + t * NTHR) * 4, ldw - 4));
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const int j = (tid + t * NTHR) * 4;
            const float mk = j < ldw ? 1.f : 0.f;
            if (j < NV * 256) *reinterpret_cast<float4*>(sv + j) = make_float4(q4[t].x * mk, q4[t].y * mk, q4[t].z * mk, q4[t].w * mk);
        }
    }
    __syncthreads();
    const int rows = m + 1;
    const int rpb = (rows + NBLK - 1) / NBLK;
    const int rbeg = blk * rpb;
    const int rend = min(rows, rbeg + rpb);
    float4 cacc[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) cacc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Pb = w.P + (size_t)b * (m_max + 1) * ldw;
    float4 nx[PF][NV];
    float ukeep = 0.f;
    // every lane loads from a clamped address and the value is masked afterwards: a load under a per-lane predicate sits in its own
    // exec branch and the compiler then waits with vmcnt(0) at the join — for the rows requested ahead too, which is how the first
    // version of this loop paid a full memory round trip per row with the "prefetch" in place
    auto rload = [&](int i, float4 (&dst)[NV]) {
        const float* rp = Pb + (size_t)max(min(i, rend - 1), 0) * ldw;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int c = (t * 64 + lane) * 4;
            dst[t] = *reinterpret_cast<const float4*>(rp + min(c, ldw - 4));
        }
    };
    // ... masked where the row is used, by a multiplication: a select would let the compiler put the load back under the predicate
    auto rmask = [&](int i, const float4 (&src)[NV], float4 (&dst)[NV]) {
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const float mk = ((t * 64 + lane) * 4 < ldw && i < rend) ? 1.f : 0.f;
            dst[t] = make_float4(src[t].x * mk, src[t].y * mk, src[t].z * mk, src[t].w * mk);
        }
    };
    // the slots are requested in slot order here too (pinned): the wait at the loop's top is the stricter of what this block and
    // the loop's end leave outstanding, and interleaved requests would make it "everything"
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        rload(rbeg + wave + 4 * q, nx[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int i0 = rbeg + wave; i0 < rend; i0 += 4 * PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            // no early exit: rows past the end run masked (all zeros: they add nothing, and their u is not stored) — a branch
            // here makes the row slots loop-carried through copies, and copies of registers a load is still writing are waits
            const int i = i0 + 4 * q;
            float4 pr[NV];
            float dot = 0.f;
            rmask(i, nx[q], pr);
            __builtin_amdgcn_sched_barrier(0);      // the refill is issued after the slot was read: same registers, no copies (and no wait for them) at the loop's end
            rload(i + 4 * PF, nx[q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const float4 vv = *reinterpret_cast<const float4*>(sv + (t * 64 + lane) * 4);
                dot += (pr[t].x * vv.x + pr[t].y * vv.y) + (pr[t].z * vv.z + pr[t].w * vv.w);
            }
            dot = wave_sum(dot);
            const float ri = (i == m) ? (float)(m + 1) : 1.0f;
            const float ui = ri / (dot + SINK_EPS);
            // u_i is kept by lane (row number within the wave) and stored after the loop: with a store outstanding beside the
            // loads the compiler has to wait with vmcnt(0) (loads and stores return out of order with respect to each other on
            // gfx9), which would put a full round trip back on every row
            if (lane == ((i - rbeg) >> 2)) ukeep = ui;      // (rows past the end land in lanes whose store is masked)
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                cacc[t].x += pr[t].x * ui;
                cacc[t].y += pr[t].y * ui;
                cacc[t].z += pr[t].z * ui;
                cacc[t].w += pr[t].w * ui;
            }
        }
    }
    {
        const int i = rbeg + wave + 4 * lane;      // at most ceil(4352 / 32 / 4) = 34 rows per wave
        if (i < rend) w.u[(size_t)b * (m_max + 1) + i] = ukeep;
    }
    float* part = w.part + ((size_t)b * NBLK * 4 + blk * 4 + wave) * ldw;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = (t * 64 + lane) * 4;
        if (c < ldw) *reinterpret_cast<float4*>(part + c) = cacc[t];
    }
}

__global__ __launch_bounds__(256) void sink_colreduce_kernel(const int* __restrict__ n_lens, SinkWs w, int n_max) {
    const int b = blockIdx.y;
    const int n = n_lens ? n_lens[b] : n_max;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.ldw) return;
    const float* part = w.part + (size_t)b * NBLK * 4 * w.ldw + j;
    // all partials requested before the first is added (the sum itself stays in partial order: it fixes the bits)
    float pv[NBLK * 4];
#pragma unroll
    for (int k = 0; k < NBLK * 4; ++k) pv[k] = part[(size_t)k * w.ldw];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NBLK * 4; ++k) s += pv[k];
    const float cj = (j == n) ? (float)(n + 1) : 1.0f;
    w.v[(size_t)b * w.ldw + j] = (j <= n) ? cj / (s + SINK_EPS) : 0.f;
}

// ---- final pass: p = P*u*v (or the dual-softmax score), row max/argmax, column partial max ----------
// Built like sink_iter_kernel (see the rules there): the row ahead is requested through unpredicated loads from clamped
// addresses, the wave index is scalar, and nothing is stored inside the row loop — a row's (max, arg-max) is kept by the lane
// with the row's number and stored after the loop.  WANT_P (the caller asked for the full assignment matrix) stores p inside
// the loop and pays a memory round trip per row for it, as every row did before: 135 -> 70 us at 16 pairs of 2049 x 2049.
template <int NV, int MODE, bool WANT_P>
__global__ __launch_bounds__(256) void sink_final_kernel(const int* __restrict__ m_lens, const int* __restrict__ n_lens,
                                                         SinkWs w, int m_max, int n_max, float* __restrict__ p_out,
                                                         int ldp) {
    extern __shared__ __attribute__((aligned(16))) float sv[];
    const int b = blockIdx.y;
    const int m = m_lens ? m_lens[b] : m_max, n = n_lens ? n_lens[b] : n_max;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ldw = w.ldw;
    const float* colvec = (MODE == 0) ? w.v : w.clse;
    {   // column vector -> LDS, padded to NV * 256 (the padding is never compared: j < n below); all loads before the first store
        constexpr int NQ = (NV * 64 + 255) / 256;
        float4 q4[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) q4[t] = *reinterpret_cast<const float4*>(colvec + (size_t)b * ldw + min((tid + t * 256) * 4, ldw - 4));
#pragma unroll
        for (int t = 0; t < NQ; ++t)
            if ((tid + t * 256) * 4 < NV * 256) *reinterpret_cast<float4*>(sv + (tid + t * 256) * 4) = q4[t];
    }
    __syncthreads();
    const int rows = m + 1;
    const int rpb = (rows + NBLK - 1) / NBLK;
    const int rbeg = blockIdx.x * rpb;
    const int rend = min(rows, rbeg + rpb);
    float cmax[NV * 4];
    int cidx[NV * 4];
#pragma unroll
    for (int t = 0; t < NV * 4; ++t) { cmax[t] = -INFINITY; cidx[t] = 0x7fffffff; }
    const float* Pb = w.P + (size_t)b * (m_max + 1) * ldw;
    const float* rowvec = (MODE == 0) ? w.u : w.rlse;
    float4 nx[NV];
    float nu;
    auto rload = [&](int i) {
        const int ic = max(min(i, rend - 1), 0);
        nu = rowvec[(size_t)b * (m_max + 1) + ic];
        const float* rp = Pb + (size_t)ic * ldw;
#pragma unroll
        for (int t = 0; t < NV; ++t) nx[t] = *reinterpret_cast<const float4*>(rp + min((t * 64 + lane) * 4, ldw - 4));
    };
    rload(rbeg + wave);
    __builtin_amdgcn_sched_barrier(0);
    float rkeep = -INFINITY;
    int ikeep = 0x7fffffff;
    for (int i = rbeg + wave; i < rend; i += 4) {
        // the row's values first (they consume the prefetched registers), then the request for the next row, then the comparisons
        float pv[NV][4];
        const float ui = nu;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const float4 pr = nx[t];
            const float4 vv = *reinterpret_cast<const float4*>(sv + (t * 64 + lane) * 4);
            if (MODE == 0) {
                pv[t][0] = (pr.x * ui) * vv.x; pv[t][1] = (pr.y * ui) * vv.y;
                pv[t][2] = (pr.z * ui) * vv.z; pv[t][3] = (pr.w * ui) * vv.w;
            } else {  // exp(log_softmax_row + log_softmax_col)
                pv[t][0] = expf((pr.x - ui) + (pr.x - vv.x)); pv[t][1] = expf((pr.y - ui) + (pr.y - vv.y));
                pv[t][2] = expf((pr.z - ui) + (pr.z - vv.z)); pv[t][3] = expf((pr.w - ui) + (pr.w - vv.w));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        rload(i + 4);
        __builtin_amdgcn_sched_barrier(0);
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int c = (t * 64 + lane) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = c + k;
                if constexpr (WANT_P) {
                    if (j <= n) p_out[((size_t)b * (m_max + 1) + i) * ldp + j] = pv[t][k];
                }
                if (i < m && j < n) {
                    if (pv[t][k] > best) { best = pv[t][k]; bidx = j; }
                    if (pv[t][k] > cmax[t * 4 + k]) { cmax[t * 4 + k] = pv[t][k]; cidx[t * 4 + k] = i; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (lane == ((i - rbeg) >> 2)) { rkeep = best; ikeep = bidx; }
    }
    {
        const int i = rbeg + wave + 4 * lane;      // at most ceil(4352 / 32 / 4) = 34 rows per wave
        if (i < rend && i < m) {
            w.rowval[(size_t)b * m_max + i] = rkeep;
            w.rowidx[(size_t)b * m_max + i] = ikeep;
        }
    }
    float* part = w.part + ((size_t)b * NBLK * 4 + blockIdx.x * 4 + wave) * ldw;
    int* parti = w.parti + ((size_t)b * NBLK * 4 + blockIdx.x * 4 + wave) * ldw;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = (t * 64 + lane) * 4;
        if (c < ldw) {
            *reinterpret_cast<float4*>(part + c) = make_float4(cmax[t * 4], cmax[t * 4 + 1], cmax[t * 4 + 2], cmax[t * 4 + 3]);
            *reinterpret_cast<int4*>(parti + c) = make_int4(cidx[t * 4], cidx[t * 4 + 1], cidx[t * 4 + 2], cidx[t * 4 + 3]);
        }
    }
}

// column max / argmax over the NBLK * 4 partials of sink_final_kernel.  (value descending, row index ascending) is a total order,
// so the partials may be combined in any grouping: 32 columns per workgroup, eight threads per column with 16 partials each
// (all 16 loads in flight), then a reduction through LDS — the one-thread-per-column loop was a chain of 128 dependent round
// trips, 43 us whatever the batch.
__global__ __launch_bounds__(256) void sink_colmax_kernel(const int* __restrict__ n_lens, SinkWs w, int n_max) {
    __shared__ float sval[8][32];
    __shared__ int sidx[8][32];
    const int b = blockIdx.y;
    const int n = n_lens ? n_lens[b] : n_max;
    const int jj = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jj;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if (j < n) {
        const float* part = w.part + ((size_t)b * NBLK * 4 + g * (NBLK / 2)) * w.ldw + j;
        const int* parti = w.parti + ((size_t)b * NBLK * 4 + g * (NBLK / 2)) * w.ldw + j;
        float v[NBLK / 2];
        int i[NBLK / 2];
#pragma unroll
        for (int k = 0; k < NBLK / 2; ++k) { v[k] = part[(size_t)k * w.ldw]; i[k] = parti[(size_t)k * w.ldw]; }
#pragma unroll
        for (int k = 0; k < NBLK / 2; ++k)
            if (v[k] > best || (v[k] == best && i[k] < bidx)) { best = v[k]; bidx = i[k]; }
    }
    sval[g][jj] = best;
    sidx[g][jj] = bidx;
    __syncthreads();
    if (g == 0 && j < n) {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const float v = sval[q][jj];
            const int i = sidx[q][jj];
            if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
        }
        w.colval[(size_t)b * n_max + j] = best;
        w.colidx[(size_t)b * n_max + j] = bidx;
    }
}

// ---- mutual check + threshold (compute_matches) ------------------------------------------------------
__global__ __launch_bounds__(256) void sink_mutual_kernel(const int* __restrict__ m_lens, const int* __restrict__ n_lens,
                                                          SinkWs w, int m_max, int n_max, float thr,
                                                          long long* __restrict__ matches0, long long* __restrict__ matches1,
                                                          float* __restrict__ ms0, float* __restrict__ ms1) {
    const int b = blockIdx.y;
    const int m = m_lens ? m_lens[b] : m_max, n = n_lens ? n_lens[b] : n_max;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float* rowval = w.rowval + (size_t)b * m_max;
    const int* rowidx = w.rowidx + (size_t)b * m_max;
    const int* colidx = w.colidx + (size_t)b * n_max;
    if (t < m_max) {
        long long mi = -1;
        float sc = 0.f;
        // A row / column whose scores are all NaN (a non-finite operand upstream) never wins a comparison and keeps the "no index"
        // sentinel: it is reported unmatched with score 0 (torch.max would name the first NaN; every `> p` test on it fails there
        // too, so the MATCHES agree) — and is never used as an address.
        if (t < m && n > 0 && (unsigned)rowidx[t] < (unsigned)n) {
            const int j = rowidx[t];
            const bool mutual = colidx[j] == t;
            sc = mutual ? rowval[t] : 0.f;
            if (mutual && sc > thr) mi = j;
        }
        if (matches0) matches0[(size_t)b * m_max + t] = mi;
        if (ms0) ms0[(size_t)b * m_max + t] = sc;
    }
    if (t < n_max) {
        long long mj = -1;
        float sc = 0.f;
        if (t < n && m > 0 && (unsigned)colidx[t] < (unsigned)m && (unsigned)rowidx[colidx[t]] < (unsigned)n) {
            const int i = colidx[t];
            const bool mutual1 = rowidx[i] == t;
            const bool mutual0 = colidx[rowidx[i]] == i;
            const float s0 = mutual0 ? rowval[i] : 0.f;
            sc = mutual1 ? s0 : 0.f;
            if (mutual1 && mutual0 && s0 > thr) mj = i;
        }
        if (matches1) matches1[(size_t)b * n_max + t] = mj;
        if (ms1) ms1[(size_t)b * n_max + t] = sc;
    }
}

// ---- dual softmax: column log-sum-exp of the augmented matrix (partials per wave, then combine) -----
template <int NV>
__global__ __launch_bounds__(256) void dual_colpart_kernel(const int* __restrict__ m_lens, SinkWs w, int m_max) {
    const int b = blockIdx.y;
    const int m = m_lens ? m_lens[b] : m_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ldw = w.ldw;
    const int rows = m + 1;
    const int rpb = (rows + NBLK - 1) / NBLK;
    const int rbeg = blockIdx.x * rpb;
    const int rend = min(rows, rbeg + rpb);
    float cm[NV * 4], cs[NV * 4];
#pragma unroll
    for (int t = 0; t < NV * 4; ++t) { cm[t] = -INFINITY; cs[t] = 0.f; }
    const float* Pb = w.P + (size_t)b * (m_max + 1) * ldw;
    for (int i = rbeg + wave; i < rend; i += 4) {
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int c = (t * 64 + lane) * 4;
            if (c < ldw) {
                const float4 pr = *reinterpret_cast<const float4*>(Pb + (size_t)i * ldw + c);
                const float x[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float nm = fmaxf(cm[t * 4 + k], x[k]);
                    if (nm > -INFINITY) {
                        cs[t * 4 + k] = cs[t * 4 + k] * expf(cm[t * 4 + k] - nm) + expf(x[k] - nm);
                        cm[t * 4 + k] = nm;
                    }
                }
            }
        }
    }
    float* part = w.part + ((size_t)b * NBLK * 4 + blockIdx.x * 4 + wave) * ldw;
    float* parts = reinterpret_cast<float*>(w.parti) + ((size_t)b * NBLK * 4 + blockIdx.x * 4 + wave) * ldw;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = (t * 64 + lane) * 4;
        if (c < ldw) {
            *reinterpret_cast<float4*>(part + c) = make_float4(cm[t * 4], cm[t * 4 + 1], cm[t * 4 + 2], cm[t * 4 + 3]);
            *reinterpret_cast<float4*>(parts + c) = make_float4(cs[t * 4], cs[t * 4 + 1], cs[t * 4 + 2], cs[t * 4 + 3]);
        }
    }
}

__global__ __launch_bounds__(256) void dual_colreduce_kernel(SinkWs w) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.ldw) return;
    const float* part = w.part + (size_t)b * NBLK * 4 * w.ldw + j;
    const float* parts = reinterpret_cast<const float*>(w.parti) + (size_t)b * NBLK * 4 * w.ldw + j;
    float mx = -INFINITY;
    for (int k = 0; k < NBLK * 4; ++k) mx = fmaxf(mx, part[(size_t)k * w.ldw]);
    float s = 0.f;
    if (mx > -INFINITY)
        for (int k = 0; k < NBLK * 4; ++k) {
            const float pm = part[(size_t)k * w.ldw];
            if (pm > -INFINITY) s += parts[(size_t)k * w.ldw] * expf(pm - mx);
        }
    w.clse[(size_t)b * w.ldw + j] = (mx > -INFINITY) ? mx + logf(s) : INFINITY;
}

template <int NV>
int run_sinkhorn(const float* dist, int ldd, const int* m_lens, const int* n_lens, const float* bin, int iters,
                 float thr, float* p_out, int ldp, long long* matches0, long long* matches1, float* ms0, float* ms1,
                 int batch, int m_max, int n_max, SinkWs w, hipStream_t st, bool dual) {
    const size_t shm_it = (size_t)NV * 256 * 4;      // the row kernels keep the column vector in LDS, padded to NV * 256
    dim3 grow(cdiv(m_max + 1, 4), batch), giter(NBLK, batch), gcol(cdiv(w.ldw, 256), batch);
    hipLaunchKernelGGL(sink_init_kernel<NV>, grow, dim3(256), 0, st, dist, ldd, (long long)m_max * ldd, m_lens, n_lens, bin,
                       w, m_max, n_max, dual ? 1 : 0);
    if (!dual) {
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL((sink_iter_kernel<NV, 1>), giter, dim3(256), shm_it, st, m_lens, n_lens, w, m_max, n_max);
            hipLaunchKernelGGL(sink_colreduce_kernel, gcol, dim3(256), 0, st, n_lens, w, n_max);
        }
        if (p_out) hipLaunchKernelGGL((sink_final_kernel<NV, 0, true>), giter, dim3(256), shm_it, st, m_lens, n_lens, w, m_max, n_max, p_out, ldp);
        else hipLaunchKernelGGL((sink_final_kernel<NV, 0, false>), giter, dim3(256), shm_it, st, m_lens, n_lens, w, m_max, n_max, p_out, ldp);
    } else {
        hipLaunchKernelGGL(dual_colpart_kernel<NV>, giter, dim3(256), 0, st, m_lens, w, m_max);
        hipLaunchKernelGGL(dual_colreduce_kernel, gcol, dim3(256), 0, st, w);
        if (p_out) hipLaunchKernelGGL((sink_final_kernel<NV, 1, true>), giter, dim3(256), shm_it, st, m_lens, n_lens, w, m_max, n_max, p_out, ldp);
        else hipLaunchKernelGGL((sink_final_kernel<NV, 1, false>), giter, dim3(256), shm_it, st, m_lens, n_lens, w, m_max, n_max, p_out, ldp);
    }
    hipLaunchKernelGGL(sink_colmax_kernel, dim3(cdiv(n_max, 32), batch), dim3(256), 0, st, n_lens, w, n_max);
    hipLaunchKernelGGL(sink_mutual_kernel, dim3(cdiv(m_max > n_max ? m_max : n_max, 256), batch), dim3(256), 0, st, m_lens, n_lens, w,
                       m_max, n_max, thr, matches0, matches1, ms0, ms1);
    return pram_launch_status(dual ? "pram_dual_softmax_match_f32" : "pram_sinkhorn_match_f32");
}

int dispatch(const float* dist, int ldd, const int* m_lens, const int* n_lens, const float* bin, int iters, float thr,
             float* p_out, int ldp, long long* matches0, long long* matches1, float* ms0, float* ms1, int batch,
             int m_max, int n_max, void* workspace, void* stream, bool dual) {
    PRAM_REQUIRE(dist && bin && workspace, "sinkhorn: null pointer");
    PRAM_REQUIRE(batch >= 0 && m_max > 0 && n_max > 0 && iters >= 0, "sinkhorn: bad sizes");
    PRAM_REQUIRE(n_max + 1 <= 17 * 256, "sinkhorn: n_max=%d exceeds 4351 columns", n_max);
    PRAM_REQUIRE(!p_out || ldp >= n_max + 1, "sinkhorn: ldp too small");
    if (batch == 0) return PRAM_OK;
    SinkWs w = carve(workspace, batch, m_max, n_max, nullptr);
    hipStream_t st = (hipStream_t)stream;
#define RUN(NV) return run_sinkhorn<NV>(dist, ldd, m_lens, n_lens, bin, iters, thr, p_out, ldp, matches0, matches1, ms0, ms1, batch, m_max, n_max, w, st, dual)
    if (w.ldw <= 2 * 256) RUN(2);
    if (w.ldw <= 5 * 256) RUN(5);
    if (w.ldw <= 9 * 256) RUN(9);
    RUN(17);
#undef RUN
}

}  // namespace

extern "C" size_t pram_sinkhorn_workspace_bytes(int batch, int m_max, int n_max) {
    size_t total = 0;
    carve(nullptr, batch, m_max, n_max, &total);
    return total;
}

extern "C" int pram_sinkhorn_match_f32(const float* dist, int ldd, const int* m_lens, const int* n_lens,
                                       const float* bin_score, int iters, float match_threshold, float* p_out, int ldp,
                                       long long* matches0, long long* matches1, float* mscores0, float* mscores1,
                                       int batch, int m_max, int n_max, void* workspace, void* stream) {
    return dispatch(dist, ldd, m_lens, n_lens, bin_score, iters, match_threshold, p_out, ldp, matches0, matches1,
                    mscores0, mscores1, batch, m_max, n_max, workspace, stream, false);
}

extern "C" int pram_dual_softmax_match_f32(const float* dist, int ldd, const int* m_lens, const int* n_lens,
                                           const float* bin_score, float match_threshold, float* p_out, int ldp,
                                           long long* matches0, long long* matches1, float* mscores0, float* mscores1,
                                           int batch, int m_max, int n_max, void* workspace, void* stream) {
    return dispatch(dist, ldd, m_lens, n_lens, bin_score, 0, match_threshold, p_out, ldp, matches0, matches1, mscores0,
                    mscores1, batch, m_max, n_max, workspace, stream, true);
}
