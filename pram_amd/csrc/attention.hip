// Flash-style multi-head attention in exact fp32 on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32).  Replaces the materialised einsum -> softmax -> einsum of
// nets/segnetvit.py:73-76 (self), nets/gml.py:175-179 (cross: both directions in one launch, kv_shift) and the column means of
// nets/adagml.py:148,229.
//
// Work decomposition: one workgroup = 4 waves = 128 query rows of one (batch, head); each wave
// owns 32 query rows and walks the keys in tiles of 64 staged through LDS (K XOR-swizzled for
// conflict-free ds_read_b128, V with its two 32-column halves swapped on rows with key bit 2 set, so the
// two half-waves of a ds_read_b32 — rows 4 apart — use different banks), double buffered, one
// barrier per tile.
//
// Register-level layout (the point of the design): everything is computed TRANSPOSED so that
// the query row is the lane index in every accumulator —
//   Sᵀ[key][q] = mfma(A = K-fragment, B = Q-fragment): lane (r = l&31, h = l>>5) holds, for
//       query r, the 16 keys (e&3) + 8(e>>2) + 4h of the 32-key sub-tile;
//   Oᵀ[d][q]  = mfma(A = Vᵀ-fragment, B = Pᵀ-fragment): the B operand of step e is exactly the
//       register e of exp(Sᵀ) (lane r supplies P[q = r][key(e, h)]), and the accumulator holds
//       O[q = r][d(e, h)].
// So the online-softmax state (row max m, row sum l, rescale factor) is lane-local, P never
// moves between lanes or through LDS, and the only cross-lane op per tile is one
// __shfl_xor(…, 32) for the row max.  At the f32 MFMA rate (64 cycles per instruction) the
// kernel is matrix-pipe bound: 128 MFMA = 8192 cycles per 64-key tile per wave against ~64
// v_exp + ~200 VALU and 80 LDS reads.
//
// Key chunks.  The keys of a sequence are processed in chunks of CHUNK_TILES tiles (512 keys): every chunk runs the
// online softmax from a fresh state, is normalised, and is folded into the running result with fold_chunk() — a
// fixed left fold in chunk order.  One workgroup normally walks all chunks of its 128 query rows ("fused").  When a
// launch has too few (batch, head, q-tile) units to fill 2 x 256 workgroup slots — one or two query frames — the
// chunks become a grid dimension instead ("split"): each workgroup writes its normalised chunk result to a
// workspace and combine_kernel applies the SAME fold in the SAME order, so the output does not depend on which
// mode ran, bit for bit.  That is what lets a padded batch element still equal its B = 1 run exactly.
#include "common.h"
#include <math.h>

namespace {

constexpr int D = 64;     // head dim (fixed: hidden 256 / 4 heads)
constexpr int QW = 32;    // query rows per wave
constexpr int NW = 4;     // waves per workgroup
constexpr int BQ = QW * NW;
constexpr int BKV = 64;   // keys per LDS tile
constexpr int CHUNK_TILES = 8;                  // tiles per key chunk
constexpr int CHUNK = CHUNK_TILES * BKV;        // 512 keys
constexpr int SPLIT_BELOW = 2 * 256;            // fused launches with fewer workgroups than this are split
constexpr float LOG2E = 1.4426950408889634f;

// Fold one normalised chunk result (oc, lc = log2-sum-exp of its scores) into the running (ot, lt).  Used by the
// attention kernel (fused mode) and by combine_kernel (split mode): identical operations in identical order.
// Starting from lt = -inf, ot = 0 the first fold returns (oc, lc) exactly.
__device__ __forceinline__ void fold_weights(float lt, float lc, float* at, float* ac, float* lnew) {
    const float mx = fmaxf(lt, lc);
    const float wt = exp2f(lt - mx), wc = exp2f(lc - mx);
    const float den = wt + wc;
    *at = wt / den;
    *ac = wc / den;
    *lnew = mx + log2f(den);
}
__device__ __forceinline__ float fold_value(float ot, float oc, float at, float ac) { return ot * at + oc * ac; }

struct AttnArgs {
    const float* q; const float* k; const float* v;
    float* out; float* lse2;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, ldv, ldo;
    int batch, heads, m_max, n_max;
    float scale2;
    int q_tiles;
    int kv_shift;   // keys / values of batch element b come from element (b + kv_shift) % batch (cross attention: both directions in one launch)
    int nsplit;     // 1: fused (a workgroup folds all key chunks); > 1: blockIdx.y = key chunk, results go to part_o / part_l
    float* part_o;  // [nsplit][batch * m_max][heads * D]   normalised chunk outputs
    float* part_l;  // [nsplit][batch][heads][m_max]        their log2-sum-exp
};

struct Smem {
    float k[2][BKV * D];
    float v[2][BKV * D];
};  // 64 KiB -> two workgroups per CU

__device__ __forceinline__ int key_of(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Softmax exponentials: arguments are <= 0 and results in [0, 1], so the bare v_exp_f32 is enough.  exp2f() wraps
// it in a denormal-range rescue (compare, two selects, add, ldexp: 6 instructions instead of 1) that only matters
// for probabilities below 2^-126; that wrapper was half of the softmax's VALU work (attention 121 -> 128 TFLOP/s).
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__global__ __launch_bounds__(256, 2) void attention_kernel(AttnArgs p) {
    __shared__ Smem s;
    const int nblk = p.batch * p.heads * p.q_tiles;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int qt = id % p.q_tiles;
    const int bh = id / p.q_tiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qt * BQ >= qlen) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * BQ + wave * QW;          // first query row of this wave (within batch b)
    const bool wave_active = q0 < qlen;
    const int qrow = q0 + r;
    const bool q_ok = qrow < qlen;
    if (klen <= 0) {
        // empty key set (a pruned-away / empty opposite set in a ragged batch): the context of the valid query rows is
        // defined as 0 (and lse as 0) instead of being left as uninitialised memory for the MLP tail to read
        if (q_ok && (p.nsplit <= 1 || blockIdx.y == 0)) {
            float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<float4*>(op + c * 8 + h * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = 0.f;
        }
        return;
    }

    const float* qp = p.q + ((size_t)b * p.m_max + qrow) * p.ldq + head * D;
    const float* kp = p.k + (size_t)kb * p.n_max * p.ldk + head * D;
    const float* vp = p.v + (size_t)kb * p.n_max * p.ldv + head * D;

    // Q fragments: qf[c] = Q[qrow][8c + 4h .. +3]
    float4 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        qf[c] = q_ok ? *reinterpret_cast<const float4*>(qp + c * 8 + h * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int lrow = tid >> 4, lslot = tid & 15;   // staging: row lrow + 16p, 16-B slot lslot
    float4 kr[4], vr[4];
    // staging loads are branch-free (clamped, always-legal row; rows >= klen are zeroed when they go to LDS), so
    // they can be scheduled in between the first MFMAs of the tile instead of in a burst in front of them
    auto gload = [&](int kt) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int key = min(kt * BKV + lrow + 16 * pp, klen - 1);
            kr[pp] = *reinterpret_cast<const float4*>(kp + (size_t)key * p.ldk + lslot * 4);
            vr[pp] = *reinterpret_cast<const float4*>(vp + (size_t)key * p.ldv + lslot * 4);
        }
    };
    auto lstore = [&](int buf, int kt) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int row = lrow + 16 * pp;
            if (kt * BKV + row >= klen) { kr[pp] = make_float4(0.f, 0.f, 0.f, 0.f); vr[pp] = kr[pp]; }
            *reinterpret_cast<float4*>(&s.k[buf][row * D + ((lslot ^ (row & 15)) << 2)]) = kr[pp];
            *reinterpret_cast<float4*>(&s.v[buf][row * D + ((lslot ^ (((row >> 2) & 1) << 3)) << 2)]) = vr[pp];
        }
    };

    const int nkt_all = (klen + BKV - 1) / BKV;
    const int split = p.nsplit > 1 ? (int)blockIdx.y : 0;
    const int kt0 = split * CHUNK_TILES;                                   // first tile of this workgroup
    const int nkt = p.nsplit > 1 ? min(nkt_all, kt0 + CHUNK_TILES) : nkt_all;   // one past its last tile
    if (kt0 >= nkt) return;                                                // this chunk lies beyond klen
    gload(kt0);
    lstore(kt0 & 1, kt0);
    __syncthreads();

    float m_run = -1.0e30f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
    float o_tot[2][16], l_tot2 = -INFINITY;      // running fold over finished chunks
#pragma unroll
    for (int e = 0; e < 16; ++e) { o_tot[0][e] = 0.f; o_tot[1][e] = 0.f; }

    for (int kt = kt0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        const int lt = more ? kt + 1 : kt;     // the last tile re-issues its own (legal) rows: no branch in the MFMA region
        if (!wave_active) gload(lt);

        if (wave_active) {
            // ---- Sᵀ = K · Qᵀ for the two 32-key sub-tiles
            f32x16 st[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[0][e] = 0.f; st[1][e] = 0.f; }
            const float* sk = s.k[cur];
            // K fragments are software-prefetched one chunk ahead: a wave issues the ds_read_b128 pair of
            // chunk c+1 before the 8 MFMAs of chunk c, so the LDS latency hides under 512 matrix cycles.
            float4 kA0, kA1, kB0, kB1;
            auto kload = [&](int c, float4& k0, float4& k1) {
                const int slot = (2 * c + h) ^ (r & 15);   // (key & 15) == (r & 15) for key = 32t + r
                k0 = *reinterpret_cast<const float4*>(sk + r * D + (slot << 2));
                k1 = *reinterpret_cast<const float4*>(sk + (32 + r) * D + (slot << 2));
            };
            auto kmma = [&](int c, const float4& k0, const float4& k1) {
                const float a0[4] = {k0.x, k0.y, k0.z, k0.w};
                const float a1[4] = {k1.x, k1.y, k1.z, k1.w};
                const float bq[4] = {qf[c].x, qf[c].y, qf[c].z, qf[c].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bq[j], st[0], 0, 0, 0);
                    st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bq[j], st[1], 0, 0, 0);
                }
            };
            // sched_barrier(0) pins "prefetch, then MFMAs": hipcc otherwise sinks each ds_read next to its use
            __builtin_amdgcn_s_setprio(3);
            kload(0, kA0, kA1);
            kload(1, kB0, kB1);
            __builtin_amdgcn_sched_barrier(0);
            // next tile's K / V rows: 8 global loads + their address arithmetic, two per pair of MFMAs of the first chunk
            gload(lt);
            kmma(0, kA0, kA1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            kload(2, kA0, kA1);
            __builtin_amdgcn_sched_barrier(0);
            kmma(1, kB0, kB1);
#pragma unroll
            for (int c = 2; c < 8; c += 2) {
                kload(c + 1, kB0, kB1);
                __builtin_amdgcn_sched_barrier(0);
                kmma(c, kA0, kA1);
                if (c + 2 < 8) kload(c + 2, kA0, kA1);
                __builtin_amdgcn_sched_barrier(0);
                kmma(c + 1, kB0, kB1);
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- mask keys beyond klen (last tile of the sequence only)
            if (kt + 1 == nkt_all && (klen & (BKV - 1))) {
                const int kbase = kt * BKV;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + t * 32 + key_of(e, h) >= klen) st[t][e] = -INFINITY;
            }
            // ---- online softmax, lane-local per query row
            float tmax = st[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, st[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[1][e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * p.scale2);
            const float alpha = fast_exp2(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    // the ROUNDED product, like the row maximum above (m_new >= tmax * scale2, rounded): the argument is <= 0 whatever
                    // the scores' magnitude.  (An fma subtracts m_new from the exact product; at |score| ~ 1e9 — a hot residual stream on
                    // the range guard's last resort — the product's rounding error alone is up to +128 and exp2 of it is inf.)
                    const float pv = fast_exp2(st[t][e] * p.scale2 - m_new);
                    st[t][e] = pv;
                    psum += pv;
                }
            l_run = fmaf(l_run, alpha, psum);
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
            // ---- Oᵀ += Vᵀ · Pᵀ   (V fragments prefetched 4 keys = 8 MFMAs ahead)
            const float* sv = s.v[cur];
            float vA[4][2], vB[4][2];
            auto vload = [&](int t, int e0, float (&vv)[4][2]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // bit 2 of the key is h: the two half-waves read rows 4 apart; the 32-column XOR puts them on
                    // different halves of the 64 banks (a row is exactly 64 banks wide)
                    const int key = t * 32 + key_of(e0 + i, h);
                    vv[i][0] = sv[key * D + (r ^ (h << 5))];
                    vv[i][1] = sv[key * D + ((32 + r) ^ (h << 5))];
                }
            };
            auto vmma = [&](int t, int e0, const float (&vv)[4][2]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[i][0], st[t][e0 + i], oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[i][1], st[t][e0 + i], oacc[1], 0, 0, 0);
                }
            };
            __builtin_amdgcn_s_setprio(3);
            vload(0, 0, vA);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                vload(t, 4, vB);
                __builtin_amdgcn_sched_barrier(0);
                vmma(t, 0, vA);
                vload(t, 8, vA);
                __builtin_amdgcn_sched_barrier(0);
                vmma(t, 4, vB);
                vload(t, 12, vB);
                __builtin_amdgcn_sched_barrier(0);
                vmma(t, 8, vA);
                if (t == 0) vload(1, 0, vA);
                __builtin_amdgcn_sched_barrier(0);
                vmma(t, 12, vB);
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- end of a key chunk: normalise it, fold it into the running result, start afresh
            if (((kt + 1) % CHUNK_TILES) == 0 || !more) {
                const float l_c = l_run + __shfl_xor(l_run, 32, 64);
                const float inv = 1.0f / l_c;
                const float lse_c = m_run + log2f(l_c);
                if (p.nsplit > 1) {          // split mode: the chunk result itself is the output of this workgroup
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o_tot[0][e] = oacc[0][e] * inv; o_tot[1][e] = oacc[1][e] * inv; }
                    l_tot2 = lse_c;
                } else {
                    float at, ac, lnew;
                    fold_weights(l_tot2, lse_c, &at, &ac, &lnew);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        o_tot[0][e] = fold_value(o_tot[0][e], oacc[0][e] * inv, at, ac);
                        o_tot[1][e] = fold_value(o_tot[1][e], oacc[1][e] * inv, at, ac);
                    }
                    l_tot2 = lnew;
                }
                m_run = -1.0e30f;
                l_run = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
            }
        }
        if (more) lstore(cur ^ 1, kt + 1);
        __syncthreads();
    }

    if (!wave_active || !q_ok) return;
    const size_t row = (size_t)b * p.m_max + qrow;
    float* op = p.nsplit > 1 ? p.part_o + ((size_t)split * p.batch * p.m_max + row) * (p.heads * D) + head * D
                             : p.out + row * p.ldo + head * D;
#pragma unroll
    for (int dn = 0; dn < 2; ++dn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 o = make_float4(o_tot[dn][4 * g + 0], o_tot[dn][4 * g + 1], o_tot[dn][4 * g + 2], o_tot[dn][4 * g + 3]);
            *reinterpret_cast<float4*>(op + dn * 32 + 8 * g + 4 * h) = o;
        }
    if (h == 0) {
        const size_t li = ((size_t)b * p.heads + head) * p.m_max + qrow;
        if (p.nsplit > 1) p.part_l[(size_t)split * p.batch * p.heads * p.m_max + li] = l_tot2;
        else if (p.lse2) p.lse2[li] = l_tot2;
    }
}

// Split mode, second step: one wave per (query row, head), one lane per output dim; the left fold over the key
// chunks of that row in chunk order — fold_weights / fold_value exactly as the fused kernel applies them.
__global__ __launch_bounds__(256) void combine_kernel(AttnArgs p) {
    const int lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, qrow, head)
    const long long units = (long long)p.batch * p.m_max * p.heads;
    if (unit >= units) return;
    const int head = (int)(unit % p.heads);
    const long long row = unit / p.heads;                                       // b * m_max + qrow
    const int b = (int)(row / p.m_max), qrow = (int)(row - (long long)b * p.m_max);
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qrow >= qlen || klen <= 0) return;
    const int nchunk = (klen + CHUNK - 1) / CHUNK;
    const size_t li = ((size_t)b * p.heads + head) * p.m_max + qrow;
    float ot = 0.f, lt = -INFINITY;
    for (int c = 0; c < nchunk; ++c) {
        const float oc = p.part_o[((size_t)c * p.batch * p.m_max + row) * (p.heads * D) + head * D + lane];
        const float lc = p.part_l[(size_t)c * p.batch * p.heads * p.m_max + li];
        float at, ac, lnew;
        fold_weights(lt, lc, &at, &ac, &lnew);
        ot = fold_value(ot, oc, at, ac);
        lt = lnew;
    }
    p.out[(size_t)row * p.ldo + head * D + lane] = ot;
    if (p.lse2 && lane == 0) p.lse2[li] = lt;
}

// ---------------------------------------------------------------- column means (AdaGML)
// colmean[b][j] = 1/(H * m_b) * sum_h sum_i exp2(scale2 * q_i·k_j - lse2[b,h,i]).
// One workgroup = 128 keys (32 per wave, K fragments in registers), loops over heads and over
// 64-row Q tiles staged through LDS.  Sᵀ[key][q]: the query is the lane, the key the register,
// so the per-key sums accumulate in registers and cross lanes once at the end.
struct ColArgs {
    const float* q; const float* k; const float* lse2; float* colmean;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, batch, heads, m_max, n_max;
    float scale2;
    int kv_shift;   // as in AttnArgs; the means are written to the KEY side's row of colmean
};

__global__ __launch_bounds__(256, 2) void colmean_kernel(ColArgs p) {
    __shared__ float sq[BKV * D];
    const int b = blockIdx.y;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int key0 = blockIdx.x * 128 + wave * 32;
    if (blockIdx.x * 128 >= klen) return;
    const int lrow = tid >> 4, lslot = tid & 15;
    const int nqt = (qlen + BKV - 1) / BKV;

    float cacc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cacc[e] = 0.f;

    for (int head = 0; head < p.heads; ++head) {
        const int key = key0 + r;
        const float* kp = p.k + ((size_t)kb * p.n_max + key) * p.ldk + head * D;
        float4 kf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            kf[c] = (key < klen) ? *reinterpret_cast<const float4*>(kp + c * 8 + h * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* qbase = p.q + (size_t)b * p.m_max * p.ldq + head * D;
        const float* lse = p.lse2 + ((size_t)b * p.heads + head) * p.m_max;
        for (int qt = 0; qt < nqt; ++qt) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int row = lrow + 16 * pp;
                const int qi = qt * BKV + row;
                const float4 v = (qi < qlen) ? *reinterpret_cast<const float4*>(qbase + (size_t)qi * p.ldq + lslot * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&sq[row * D + ((lslot ^ (row & 15)) << 2)]) = v;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 st;
#pragma unroll
                for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int slot = (2 * c + h) ^ (r & 15);
                    const float4 qv = *reinterpret_cast<const float4*>(sq + (t * 32 + r) * D + (slot << 2));
                    const float a[4] = {kf[c].x, kf[c].y, kf[c].z, kf[c].w};
                    const float bq[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) st = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bq[j], st, 0, 0, 0);
                }
                const int qi = qt * BKV + t * 32 + r;
                const float l2 = (qi < qlen) ? lse[qi] : INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) cacc[e] += fast_exp2(fmaf(st[e], p.scale2, -l2));
            }
            __syncthreads();
        }
    }
    const float norm = qlen > 0 ? 1.0f / ((float)p.heads * (float)qlen) : 0.f;   // no queries: the mean over an empty set is reported as 0, not NaN
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float v = cacc[e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const int key = key0 + key_of(e, h);
        if (r == 0 && key < klen) p.colmean[(size_t)kb * p.n_max + key] = v * norm;
    }
}

}  // namespace

// fused or split launch (see "Key chunks" above).  The mode never changes the result.
static size_t split_bytes(int batch, int heads, int m_max, int n_max) {
    const long units = (long)batch * heads * cdiv(m_max, BQ);
    const int nsplit = cdiv(n_max, CHUNK);
    if (units >= SPLIT_BELOW || nsplit < 2) return 0;
    return (size_t)nsplit * batch * m_max * (heads * D + heads) * sizeof(float);
}

static void launch_attention(AttnArgs& p, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const size_t need = split_bytes(p.batch, p.heads, p.m_max, p.n_max);
    const int units = p.batch * p.heads * p.q_tiles;
    if (need == 0 || workspace == nullptr || workspace_bytes < need) {
        p.nsplit = 1;
        p.part_o = p.part_l = nullptr;
        hipLaunchKernelGGL(attention_kernel, dim3(units), dim3(256), 0, st, p);
        return;
    }
    p.nsplit = cdiv(p.n_max, CHUNK);
    p.part_o = (float*)workspace;
    p.part_l = p.part_o + (size_t)p.nsplit * p.batch * p.m_max * p.heads * D;
    hipLaunchKernelGGL(attention_kernel, dim3(units, p.nsplit), dim3(256), 0, st, p);
    const long long rows = (long long)p.batch * p.m_max * p.heads;
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
}

extern "C" size_t pram_attention_workspace_bytes(int batch, int heads, int m_max, int n_max) {
    return split_bytes(batch, heads, m_max, n_max);
}

extern "C" int pram_attention_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                  float* out, int ldo, float* lse2, const int* q_lens, const int* k_lens, int batch,
                                  int heads, int m_max, int n_max, float scale, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    PRAM_REQUIRE(q && k && v && out, "pram_attention_f32: null pointer");
    PRAM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "pram_attention_f32: ld must be a multiple of 4");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0, "pram_attention_f32: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_f32: empty key set");
    AttnArgs p{q, k, v, out, lse2, q_lens, k_lens, ldq, ldk, ldv, ldo, batch, heads, m_max, n_max, scale * LOG2E,
               cdiv(m_max, BQ), 0, 1, nullptr, nullptr};
    launch_attention(p, workspace, workspace_bytes, (hipStream_t)stream);
    return pram_launch_status("pram_attention_f32");
}

extern "C" int pram_attention_cross_f32(const float* qk, int ldqk, const float* v, int ldv, float* out, int ldo, float* lse2,
                                        const int* lens, int pairs, int heads, int t_max, float scale, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    PRAM_REQUIRE(qk && v && out, "pram_attention_cross_f32: null pointer");
    PRAM_REQUIRE(ldqk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "pram_attention_cross_f32: ld must be a multiple of 4");
    PRAM_REQUIRE(pairs >= 0 && heads > 0 && t_max >= 0, "pram_attention_cross_f32: bad sizes");
    if (pairs == 0 || t_max == 0) return PRAM_OK;
    AttnArgs p{qk, qk, v, out, lse2, lens, lens, ldqk, ldqk, ldv, ldo, 2 * pairs, heads, t_max, t_max, scale * LOG2E,
               cdiv(t_max, BQ), pairs, 1, nullptr, nullptr};
    launch_attention(p, workspace, workspace_bytes, (hipStream_t)stream);
    return pram_launch_status("pram_attention_cross_f32");
}

extern "C" int pram_attention_colmean_f32(const float* q, int ldq, const float* k, int ldk, const float* lse2,
                                          float* colmean, const int* q_lens, const int* k_lens, int batch, int heads,
                                          int m_max, int n_max, float scale, void* stream) {
    PRAM_REQUIRE(q && k && lse2 && colmean, "pram_attention_colmean_f32: null pointer");
    PRAM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0, "pram_attention_colmean_f32: ld must be a multiple of 4");
    if (batch == 0 || n_max == 0 || m_max == 0) return PRAM_OK;
    ColArgs p{q, k, lse2, colmean, q_lens, k_lens, ldq, ldk, batch, heads, m_max, n_max, scale * LOG2E, 0};
    hipLaunchKernelGGL(colmean_kernel, dim3(cdiv(n_max, 128), batch), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_colmean_f32");
}

extern "C" int pram_attention_cross_colmean_f32(const float* qk, int ldqk, const float* lse2, float* colmean, const int* lens,
                                                int pairs, int heads, int t_max, float scale, void* stream) {
    PRAM_REQUIRE(qk && lse2 && colmean, "pram_attention_cross_colmean_f32: null pointer");
    PRAM_REQUIRE(ldqk % 4 == 0, "pram_attention_cross_colmean_f32: ld must be a multiple of 4");
    if (pairs == 0 || t_max == 0) return PRAM_OK;
    ColArgs p{qk, qk, lse2, colmean, lens, lens, ldqk, ldqk, 2 * pairs, heads, t_max, t_max, scale * LOG2E, pairs};
    hipLaunchKernelGGL(colmean_kernel, dim3(cdiv(t_max, 128), 2 * pairs), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_cross_colmean_f32");
}
