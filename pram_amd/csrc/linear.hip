// Token-side linear algebra: nn.Linear (+concat input, +bias, +alpha, +residual, +rotary
// epilogue), batched A·Bᵀ, LayerNorm+GELU, Fourier positional encoding.
// Reference sites: nets/segnetvit.py:87-106,157-164; nets/gml.py:118-186,220,235,278-282.
#include "gemm_core.h"
#include "gemm_core_f16.h"
#include "gemm_core_x3.h"
#include "gemm_core_x3w.h"

namespace {

struct LinArgs {
    const float* a0; int lda0; int k0;
    const float* a1; int lda1; int k1;
    const float* w;
    const float* bias;
    const float* residual; int ldr;
    float* out; int ldo;
    int m, n;
    float alpha;
    int flags;
    const float* rcos; const float* rsin; int rot_cols;
    // batching (bgemm): per-z strides in floats; 0 for plain linear
    long long sa, sw, so;
    int tiles_m, tiles_n;
    // optional fp16 copy of the output (same values, rounded to nearest even) for the fp16 attention kernel, which
    // would otherwise round the fp32 output itself while staging it; `out` may be null when only the copy is wanted
    void* out16; int ldo16;
    // split-fp16 output planes (x3 path): out16 = hi plane, out16_lo = lo plane of out * out16_scale (out16_lo == nullptr:
    // out16 is the plain fp16 copy of the C5 path)
    void* out16_lo; float out16_scale;
    // ragged token matrices: rows are sequences of t_pad rows, sequence s has lens[s] valid ones; output tiles without a valid
    // row are skipped (their outputs are left untouched).  lens == nullptr: every row counts.
    const int* lens; int t_pad;
    // value columns of a q | k | v projection written TRANSPOSED (the V^T planes of attention_x3.hip, key-permuted) instead of
    // row-major: columns >= vt_col0 are vt_heads heads of 64 value dims, rows are sequences of vt_t (% 64 == 0) tokens
    void* vt_hi; void* vt_lo; int vt_col0, vt_heads, vt_t, vt_tv;
    // range guard of the split-fp16 path (common.h): the device status word of pram_set_status_word, or nullptr
    unsigned int* status;
    // LayerNorm statistics across a GEMM pair (the MLP tail: Linear -> LayerNorm -> GELU -> Linear, nets/segnetvit.py:87-95).
    //   row_ssq (first GEMM, out): [n / 64][m] partial sums of squares of the output rows, one partial per 64-column block; the
    //     first GEMM's weights are CENTRED over its outputs on the host, so its output IS h - mean(h) and the variance is mean(out^2).
    //   ln_ssq / ln_parts / ln_gamma / ln_beta / ln_eps (second GEMM, in): its A operand is GELU(a * rstd * gamma + beta), applied
    //     while the operand is staged; rstd = 1 / sqrt(sum_p ln_ssq[p][row] / K + eps).
    float* row_ssq;
    int a0_f16, a1_f16;      // fp16 MFMA path: the A segment is fp16 in HBM (a0 / a1 point at halves, lda in halves) instead of fp32
    const float* ln_ssq; int ln_parts; const float* ln_gamma; const float* ln_beta; float ln_eps;
    float act_scale = gemmx3::ACT_SCALE;      // split-fp16 path: scale of the A operand's planes (pram_act_scale() at launch)
};

// GELU(t) = t Phi(t) with the Gaussian tail written as s(|t|) = 0.5 erfc(|t| / sqrt 2) = 2^-q(|t|), q a degree-7 polynomial
// fitted on [0, 6] (least squares weighted by |t| s, coefficients rounded to fp32; profiles/tools/gelu_fit.py): Phi = 1 - s for
// t >= 0, s below.  |t ds| <= 7.5e-8 over the whole line — the class of erff's own rounding times t — for seven multiply-adds,
// one v_exp and four other instructions, where erff costs ~31: the GELU of the hidden layer is applied inside a GEMM's staging
// path, where every VALU instruction competes with the split arithmetic.  Beyond |t| = 6 the tail is held at s(6) = 1e-9.
// tests/test_gpu_guard_chunks_mlp.py pins it against an fp64 GELU on a dense grid.
__device__ __forceinline__ float gelu_erf(float t) {
    const float a = fminf(fabsf(t), 6.0f);
    float q = fmaf(-3.327738795633195e-06f, a, 1.992317265830934e-06f);
    q = fmaf(q, a, 0.0006240683724172413f);
    q = fmaf(q, a, -0.0077792988158762455f);
    q = fmaf(q, a, 0.053078699856996536f);
    q = fmaf(q, a, 0.4589627683162689f);
    q = fmaf(q, a, 1.1511503458023071f);
    q = fmaf(q, a, 0.9999977350234985f);
    const float sgm = __builtin_amdgcn_exp2f(-q);
    return t * (t >= 0.f ? 1.0f - sgm : sgm);
}

// The A-operand transform of the second GEMM of an MLP tail (gemm_core_x3.h::mainloop, AXf): v = GELU(v * rstd[row] * gamma + beta)
// on four consecutive k of row slot p.  gamma | beta live in LDS (gb: K floats each); rstd per row slot is computed once per thread.
template <int PA, int BKC = 32>
struct LnGeluXf {
    const float* gb; int K; int kq;      // kq = this thread's float4 column inside a BKC-deep chunk
    float rstd[PA];
    __device__ __forceinline__ void operator()(float4& v, int p, int kt) const {
#if defined(PRAM_LNA_ABLATE) && (PRAM_LNA_ABLATE == 1 || PRAM_LNA_ABLATE == 4)      // profiling: no transform at all (prologue kept)
        (void)p; (void)kt;
#elif defined(PRAM_LNA_ABLATE) && PRAM_LNA_ABLATE == 2     // profiling: LayerNorm's affine only
        const int k = kt * BKC + kq * 4;
        const float4 g = *reinterpret_cast<const float4*>(gb + k);
        const float4 b = *reinterpret_cast<const float4*>(gb + K + k);
        const float rs = rstd[p];
        v.x = v.x * rs * g.x + b.x; v.y = v.y * rs * g.y + b.y; v.z = v.z * rs * g.z + b.z; v.w = v.w * rs * g.w + b.w;
#elif defined(PRAM_LNA_ABLATE) && PRAM_LNA_ABLATE == 3     // profiling: GELU only (no gamma | beta reads)
        const float rs = rstd[p];
        (void)kt;
        v.x = gelu_erf(v.x * rs); v.y = gelu_erf(v.y * rs); v.z = gelu_erf(v.z * rs); v.w = gelu_erf(v.w * rs);
#else
        const int k = kt * BKC + kq * 4;
        const float4 g = *reinterpret_cast<const float4*>(gb + k);
        const float4 b = *reinterpret_cast<const float4*>(gb + K + k);
        const float rs = rstd[p];
        v.x = gelu_erf(v.x * rs * g.x + b.x);
        v.y = gelu_erf(v.y * rs * g.y + b.y);
        v.z = gelu_erf(v.z * rs * g.z + b.z);
        v.w = gelu_erf(v.w * rs * g.w + b.w);
#endif
    }
};

// rstd of row `row` from the first GEMM's partial sums of squares (fixed order: deterministic)
// (the partials are all requested before the first is added — up to 16 at a time, hidden <= 1024 — one memory round trip instead of
//  one per partial in the prologue of every workgroup; the sum keeps its ascending order: same bits)
__device__ __forceinline__ float ln_rstd(const LinArgs& p, int row, int K) {
    constexpr int QMAX = 16;
    float s = 0.f;
    for (int q0 = 0; q0 < p.ln_parts; q0 += QMAX) {
        float v[QMAX];
#pragma unroll
        for (int i = 0; i < QMAX; ++i) v[i] = p.ln_ssq[(size_t)min(q0 + i, p.ln_parts - 1) * p.m + row];
#pragma unroll
        for (int i = 0; i < QMAX; ++i)
            if (q0 + i < p.ln_parts) s += v[i];
    }
    return 1.0f / sqrtf(s / (float)K + p.ln_eps);
}

__device__ __forceinline__ bool tile_has_rows(const int* __restrict__ lens, int t_pad, int row0, int bm, int m) {
    if (!lens) return true;
    const int rend = min(row0 + bm, m);
    for (int s = row0 / t_pad; s * t_pad < rend; ++s)
        if (max(row0, s * t_pad) - s * t_pad < lens[s]) return true;
    return false;
}

// a row of a (possibly ragged) token matrix that carries data: rows beyond their sequence's length hold whatever the producer
// left there (ragged producers never write them) and are staged as zeros — deterministic, and invisible to the range guard
__device__ __forceinline__ bool row_valid(const int* __restrict__ lens, int t_pad, int row, int m) {
    if (row >= m) return false;
    if (!lens) return true;
    const int sq = row / t_pad;
    return row - sq * t_pad < lens[sq];
}

// Shared epilogue (fp32 and fp16 main loops produce the same accumulator layout): bias, alpha, rotary, residual.
template <int MI, int WN>
__device__ __forceinline__ void linear_epilogue(const LinArgs& p, f32x16 (&acc)[MI][2], float* out, int row0, int col0,
                                                int BM, int BN, _Float16* stage = nullptr) {
    using gemm::acc_row;
    const int tid = threadIdx.x;
    const int mlast = p.m - 1, nlast = p.n - 1;
    // all loads first (clamped addresses), then arithmetic, then predicated stores
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: tile bases live in scalar registers
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int cbase = col0 + wn * 64;
    const bool rot = (p.flags & PRAM_LIN_ROTARY) && cbase < p.rot_cols;
    const int c0 = cbase + r, c1 = cbase + 32 + r;
    const bool c0ok = c0 < p.n, c1ok = c1 < p.n;
    const int c0c = min(c0, nlast), c1c = min(c1, nlast);
    const float b0 = p.bias ? p.bias[c0c] : 0.f, b1 = p.bias ? p.bias[c1c] : 0.f;
    // ragged mode: only the valid rows of a tile are stored (`out` may be a persistent buffer whose other rows belong to
    // sequences that are not part of this call — AdaGML commits the matching descriptors of the pairs stopping at a layer)
    const bool ragged = p.lens != nullptr;
    const bool full = (row0 + BM <= p.m) && (col0 + BN <= p.n) && !ragged;
    // Full tiles (every row and column inside the matrix: all of them at the bench's shapes) address their loads and stores as a
    // wave-uniform tile base (scalar registers) plus a 32-bit byte offset per lane — row e of a 32-row block is (e & 3) + 8 (e >> 2)
    // leading dimensions further, a scalar product — instead of a 64-bit multiply-add per element and row (48 quarter-rate
    // instructions per 32 stores before; the epilogue is vector-issue-bound, profiles/r05_gemm_epilogue.txt).
    const unsigned unit_alpha = p.alpha == 1.0f;
    float one = 1.0f;       // a 1.0 the optimiser cannot see through: fma(s, 1, -hi) stays an fma and selects v_fma_mixlo_f16
    asm volatile("" : "+s"(one));
    float emax = 0.f;       // range guard: largest |value * out16_scale| this lane turned into split planes (valid rows only)
    // valid rows of a ragged tile: a tile inside one sequence (the rule: t_pad is a multiple of the tile height) has them up to a
    // row limit; a tile that straddles sequences asks row_valid per row
    int rlimit = p.m;
    bool straddle = false;
    if (ragged) {
        const int sq = row0 / p.t_pad;
        if ((min(row0 + BM, p.m) - 1) / p.t_pad == sq) rlimit = min(p.m, sq * p.t_pad + p.lens[sq]);
        else straddle = true;
    }
    auto rvalid = [&](int row) -> bool { return straddle ? row_valid(p.lens, p.t_pad, row, p.m) : row < rlimit; };
#pragma clang loop unroll(full)      // acc[mi] must stay in registers: a rolled loop would index it dynamically (scratch)
    for (int mi = 0; mi < MI; ++mi) {
        const int rbase = row0 + wm * 32 * MI;
        float rc_[16], rs_[16], q0[16], q1[16];
        if (rot) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t rr = (size_t)min(rbase + acc_row(mi, e, h), mlast) * 32 + r;
                rc_[e] = p.rcos[rr];
                rs_[e] = p.rsin[rr];
            }
        }
        if (p.residual && full) {
            const char* tile = reinterpret_cast<const char*>(p.residual + (size_t)(rbase + 32 * mi) * p.ldr + cbase);
            const unsigned lane_off = ((unsigned)(4 * h) * (unsigned)p.ldr + r) * 4u, ld4 = (unsigned)p.ldr * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned o = lane_off + (unsigned)((e & 3) + 8 * (e >> 2)) * ld4;
                q0[e] = *reinterpret_cast<const float*>(tile + o);
                q1[e] = *reinterpret_cast<const float*>(tile + o + 128);
            }
        } else if (p.residual) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t rr = (size_t)min(rbase + acc_row(mi, e, h), mlast) * p.ldr;
                q0[e] = p.residual[rr + c0c];
                q1[e] = p.residual[rr + c1c];
            }
        }
        float a0_[16], a1_[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0_[e] = acc[mi][0][e] + b0; a1_[e] = acc[mi][1][e] + b1; }
        if (!unit_alpha) {      // (x * 1.0f is x: skipping the multiplication changes no bit)
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0_[e] *= p.alpha; a1_[e] *= p.alpha; }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v0 = a0_[e];
            float v1 = a1_[e];
            if (rot) {
                const float e0 = v0 * rc_[e] - v1 * rs_[e];   // even dim: t0*cos + (-t1)*sin
                const float o0 = v1 * rc_[e] + v0 * rs_[e];   // odd dim : t1*cos + t0*sin
                v0 = e0;
                v1 = o0;
            }
            if (p.residual) { v0 += q0[e]; v1 += q1[e]; }
            q0[e] = v0;
            q1[e] = v1;
        }
        if (p.row_ssq) {
            // Sum of squares of this wave's 64 columns of every row (lanes 0..31 / 32..63 hold the same rows' other columns),
            // one partial per 64-column block: every tile configuration has 64-column wave tiles, so a block's partial — and
            // the consumer's ascending sum over the blocks — is the same bits whichever tile ran (batch == B = 1).
            // 16 rows (registers) x 32 lanes (columns): a halving butterfly — at every step a lane hands the half of its rows it is
            // not responsible for to its partner and adds what it receives — needs 8 + 4 + 2 + 1 + 1 = 16 cross-lane moves where a
            // plain all-reduce of every register needs 80.
            float v8[8], v4[4], v2[2], v1;
            {
                float sq[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) sq[e] = (c0ok ? q0[e] * q0[e] : 0.f) + (c1ok ? q1[e] * q1[e] : 0.f);
                const bool up = (lane & 16) != 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float keep = up ? sq[i + 8] : sq[i], send = up ? sq[i] : sq[i + 8];
                    v8[i] = keep + __shfl_xor(send, 16, 64);
                }
            }
            {
                const bool up = (lane & 8) != 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float keep = up ? v8[i + 4] : v8[i], send = up ? v8[i] : v8[i + 4];
                    v4[i] = keep + __shfl_xor(send, 8, 64);
                }
            }
            {
                const bool up = (lane & 4) != 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float keep = up ? v4[i + 2] : v4[i], send = up ? v4[i] : v4[i + 2];
                    v2[i] = keep + __shfl_xor(send, 4, 64);
                }
            }
            {
                const bool up = (lane & 2) != 0;
                const float keep = up ? v2[1] : v2[0], send = up ? v2[0] : v2[1];
                v1 = keep + __shfl_xor(send, 2, 64);
            }
            v1 += __shfl_xor(v1, 1, 64);
            // lane bits 4..1 = register bits 3..0 of the row this lane ended up with
            const int e_own = (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1);
            const int row = rbase + acc_row(mi, e_own, h);
            if ((lane & 1) == 0 && row < p.m && cbase < p.n) p.row_ssq[(size_t)(cbase >> 6) * p.m + row] = v1;
        }
        if (p.vt_hi && cbase >= p.vt_col0) {
            // Value head of the projection: straight into the V^T planes.  The accumulator layout IS the key permutation of
            // attention_x3.hip (pos_of_key): registers 8g .. 8g+7 of a lane are eight consecutive positions of one head dim, so a
            // lane writes 16 bytes per plane, head dim and register half.  Tokens beyond their sequence's length become zeros
            // (a masked key must meet a finite value).
            typedef _Float16 half8v __attribute__((ext_vector_type(8)));
            const int r0 = rbase + mi * 32;
            if (r0 < p.m) {
                const int sq = r0 / p.vt_t, t0 = r0 - sq * p.vt_t;
                const int len = p.lens ? p.lens[sq] : p.vt_t;
                const int head = (cbase - p.vt_col0) >> 6;
                const size_t drow = ((size_t)sq * p.vt_heads + head) * 64;
                const int pos0 = (t0 & ~63) + ((t0 >> 5) & 1) * 32 + h * 8;
                _Float16* vh = reinterpret_cast<_Float16*>(p.vt_hi);
                _Float16* vl = reinterpret_cast<_Float16*>(p.vt_lo);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    half8v h0, l0, h1, l1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = 8 * g + i;
                        const bool ok = t0 + (i & 3) + 8 * (2 * g + (i >> 2)) + 4 * h < len;
                        const float s0 = ok ? q0[e] * p.out16_scale : 0.f, s1 = ok ? q1[e] * p.out16_scale : 0.f;
                        emax = fmaxf(fmaxf(emax, fabsf(s0)), fabsf(s1));
                        // lo = fp16(s - hi) as one v_fma_mix (the difference is exact: same bits as convert, subtract, convert)
                        const _Float16 hh0 = (_Float16)s0, hh1 = (_Float16)s1;
                        h0[i] = hh0;
                        l0[i] = (_Float16)__builtin_fmaf(s0, one, -(float)hh0);
                        h1[i] = hh1;
                        l1[i] = (_Float16)__builtin_fmaf(s1, one, -(float)hh1);
                    }
                    const size_t d0 = (drow + r) * p.vt_tv + pos0 + 16 * g, d1 = (drow + 32 + r) * p.vt_tv + pos0 + 16 * g;
                    *reinterpret_cast<half8v*>(vh + d0) = h0;
                    *reinterpret_cast<half8v*>(vh + d1) = h1;
                    if (vl) {      // the single-product path carries one plane
                        *reinterpret_cast<half8v*>(vl + d0) = l0;
                        *reinterpret_cast<half8v*>(vl + d1) = l1;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (p.out16 && p.out16_lo && stage) {
            // split planes through LDS: a lane owns single columns of 16 rows, so direct stores are 2-byte scatters (128 bytes per
            // instruction: 256 instructions per thread of a 256 x 256 tile); each wave transposes its 32 x 64 block of both planes
            // in its own 9 KB of the (now idle) staging memory and writes whole rows, 16 bytes per lane
            typedef _Float16 half8v __attribute__((ext_vector_type(8)));
            constexpr int LD = 72;                                  // halves per staged row: 144 B, the two lane halves land 16 banks apart
            _Float16* sh = stage + wave * (2 * 32 * LD);
            _Float16* sl = sh + 32 * LD;
            const bool block_full = full || (!straddle && rbase + mi * 32 + 32 <= rlimit);      // wave-uniform: all 32 rows count
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = acc_row(0, e, h);
                const float s0 = q0[e] * p.out16_scale, s1 = q1[e] * p.out16_scale;
                if (block_full || rvalid(rbase + mi * 32 + rr)) emax = fmaxf(fmaxf(emax, fabsf(s0)), fabsf(s1));
                const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
                sh[rr * LD + r] = h0;
                sh[rr * LD + 32 + r] = h1;
                sl[rr * LD + r] = (_Float16)__builtin_fmaf(s0, one, -(float)h0);
                sl[rr * LD + 32 + r] = (_Float16)__builtin_fmaf(s1, one, -(float)h1);
            }
            _Float16* oh = reinterpret_cast<_Float16*>(p.out16);
            _Float16* ol = reinterpret_cast<_Float16*>(p.out16_lo);
            const int seg = lane & 7;
            const bool cok = cbase + seg * 8 < p.n;                 // n % 8 == 0 on this path
            if (full) {
                char* th = reinterpret_cast<char*>(oh + (size_t)(rbase + mi * 32) * p.ldo16 + cbase);
                char* tl = reinterpret_cast<char*>(ol + (size_t)(rbase + mi * 32) * p.ldo16 + cbase);
                const unsigned lane_off = ((unsigned)(lane >> 3) * (unsigned)p.ldo16 + seg * 8) * 2u, ld16 = (unsigned)p.ldo16 * 16u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = (lane >> 3) + 8 * j;
                    const half8v vh = *reinterpret_cast<const half8v*>(sh + rr * LD + seg * 8);
                    const half8v vl = *reinterpret_cast<const half8v*>(sl + rr * LD + seg * 8);
                    *reinterpret_cast<half8v*>(th + lane_off + j * ld16) = vh;
                    *reinterpret_cast<half8v*>(tl + lane_off + j * ld16) = vl;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = (lane >> 3) + 8 * j;
                    const int row = rbase + mi * 32 + rr;
                    const half8v vh = *reinterpret_cast<const half8v*>(sh + rr * LD + seg * 8);
                    const half8v vl = *reinterpret_cast<const half8v*>(sl + rr * LD + seg * 8);
                    const bool ok = cok && rvalid(row);      // as the fp32 output below
                    if (ok) {
                        *reinterpret_cast<half8v*>(oh + (size_t)row * p.ldo16 + cbase + seg * 8) = vh;
                        *reinterpret_cast<half8v*>(ol + (size_t)row * p.ldo16 + cbase + seg * 8) = vl;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // one 32-row block at a time: hoisting the next block's work up here spills
        } else if (p.out16 && p.out16_lo) {      // split planes: hi = fp16(v s), lo = fp16(v s - hi)
            _Float16* oh = reinterpret_cast<_Float16*>(p.out16);
            _Float16* ol = reinterpret_cast<_Float16*>(p.out16_lo);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + acc_row(mi, e, h);
                if (rvalid(row)) {
                    const float s0 = q0[e] * p.out16_scale, s1 = q1[e] * p.out16_scale;
                    emax = fmaxf(fmaxf(emax, fabsf(s0)), fabsf(s1));
                    const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
                    if (c0ok) { oh[(size_t)row * p.ldo16 + c0] = h0; ol[(size_t)row * p.ldo16 + c0] = (_Float16)__builtin_fmaf(s0, one, -(float)h0); }
                    if (c1ok) { oh[(size_t)row * p.ldo16 + c1] = h1; ol[(size_t)row * p.ldo16 + c1] = (_Float16)__builtin_fmaf(s1, one, -(float)h1); }
                }
            }
        } else if (p.out16 && stage) {
            // plain fp16 copy through LDS, like the split planes above: a lane owns single columns of 16 rows, so direct stores are
            // 2-byte scatters; each wave transposes its 32 x 64 block in 4.5 KB of the (idle) staging memory and writes 16 bytes per lane
            typedef _Float16 half8v __attribute__((ext_vector_type(8)));
            constexpr int LD = 72;
            _Float16* sh = stage + wave * (32 * LD);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = acc_row(0, e, h);
                sh[rr * LD + r] = (_Float16)q0[e];
                sh[rr * LD + 32 + r] = (_Float16)q1[e];
            }
            _Float16* o16 = reinterpret_cast<_Float16*>(p.out16);
            const int seg = lane & 7;
            const bool cok = cbase + seg * 8 < p.n;                 // n % 8 == 0 on this path
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = (lane >> 3) + 8 * j;
                const int row = rbase + mi * 32 + rr;
                const half8v v = *reinterpret_cast<const half8v*>(sh + rr * LD + seg * 8);
                if (cok && row < p.m) *reinterpret_cast<half8v*>(o16 + (size_t)row * p.ldo16 + cbase + seg * 8) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (p.out16) {
            _Float16* o16 = reinterpret_cast<_Float16*>(p.out16);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + acc_row(mi, e, h);
                if (row < p.m) {
                    if (c0ok) o16[(size_t)row * p.ldo16 + c0] = (_Float16)q0[e];
                    if (c1ok) o16[(size_t)row * p.ldo16 + c1] = (_Float16)q1[e];
                }
            }
        }
        if (out == nullptr) continue;
        if (full) {   // block-uniform fast path: no per-element predicates
            char* tile = reinterpret_cast<char*>(out + (size_t)(rbase + 32 * mi) * p.ldo + cbase);
            const unsigned lane_off = ((unsigned)(4 * h) * (unsigned)p.ldo + r) * 4u, ld4 = (unsigned)p.ldo * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned o = lane_off + (unsigned)((e & 3) + 8 * (e >> 2)) * ld4;
                *reinterpret_cast<float*>(tile + o) = q0[e];
                *reinterpret_cast<float*>(tile + o + 128) = q1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + acc_row(mi, e, h);
                if (rvalid(row)) {
                    if (c0ok) out[(size_t)row * p.ldo + c0] = q0[e];
                    if (c1ok) out[(size_t)row * p.ldo + c1] = q1[e];
                }
            }
        }
    }
    if (p.out16_lo) x3_range_flag(p.status, emax);
}

template <int MI, int WN, int BKT>
__global__ __launch_bounds__(gemm::NT, (gemm::Cfg<MI, WN, BKT>::WAVES)) void linear_kernel(LinArgs p) {
    using namespace gemm;
    using C = Cfg<MI, WN, BKT>;
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, RPP = C::RPP;
    __shared__ Smem<C> smem;
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int z = blockIdx.y;
    const float* a0 = p.a0 + z * p.sa;
    const float* w = p.w + z * p.sw;
    float* out = p.out + z * p.so;
    const int K = p.k0 + p.k1;
    const int tid = threadIdx.x;
    const int srow = C::stage_row(tid), skq = C::stage_kq(tid);
    const int row0 = tm * BM, col0 = tn * BN;
    if (!tile_has_rows(p.lens, p.t_pad, row0, BM, p.m)) return;

    // loaders: clamped (always legal) addresses + select, no branches around the loads; row bases hoisted
    const int mlast = p.m - 1, nlast = p.n - 1, klast = K - 4;
    const float* arow0[C::PA];
    const float* arow1[C::PA];
    const float* brow[C::PB];
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int rc = min(row0 + srow + RPP * pp, mlast);
        arow0[pp] = a0 + (size_t)rc * p.lda0;
        arow1[pp] = p.a1 ? p.a1 + (size_t)rc * p.lda1 - p.k0 : arow0[pp];
    }
#pragma unroll
    for (int pp = 0; pp < C::PB; ++pp) brow[pp] = w + (size_t)min(col0 + srow + RPP * pp, nlast) * K;
    auto adv = [](int) {};
    auto la = [&](int pp, int kt) -> float4 {
        const int kc = min(kt * BK + skq * 4, klast);
        const bool second = (kt * BK >= p.k0) && p.k1 > 0;            // wave-uniform: k0 % BK == 0
        return *reinterpret_cast<const float4*>((second ? arow1[pp] : arow0[pp]) + kc);
    };
    auto oka = [&](int pp, int kt) -> bool { return (row0 + srow + RPP * pp) < p.m && (kt * BK + skq * 4) < K; };
    auto lb = [&](int pp, int kt) -> float4 { return *reinterpret_cast<const float4*>(brow[pp] + min(kt * BK + skq * 4, klast)); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + srow + RPP * pp) < p.n && (kt * BK + skq * 4) < K; };

    f32x16 acc[MI][2];
    mainloop<C, MI>(smem, adv, la, oka, lb, okb, (K + BK - 1) / BK, acc);

    linear_epilogue<MI, WN>(p, acc, out, row0, col0, BM, BN);
}

// fp16-operand variant (BASELINE C5 "fp16 MFMA path"): w16 is the weight matrix pre-converted to fp16.
// LNA: the A operand (fp16 hidden layer of an MLP tail, centred) is normalised and GELU-ed while it is staged (LnGeluXf)
template <int MI, int WN, bool LNA = false>
__global__ __launch_bounds__(gemm16::NT, 2) void linear_f16_kernel(LinArgs p, const _Float16* __restrict__ w16) {
    using namespace gemm16;
    using C = Cfg<MI, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    __shared__ Smem<MI, WN> smem;
    __shared__ __attribute__((aligned(16))) float lngb[LNA ? 2 * 1024 : 4];
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int K = p.k0 + p.k1;
    const int tid = threadIdx.x;
    const int arow = tid >> 4, akq = tid & 15, brow = tid >> 3, bsl = tid & 7;
    const int row0 = tm * BM, col0 = tn * BN;
    // ragged token matrices (pram_linear_f16_ragged_f32): tiles without a valid row are skipped, the epilogue stores valid rows only.
    // Rows beyond a sequence's length are multiplied like the others (whatever they hold stays in their own output rows, which
    // are never stored).  Not for the V^T-writing projection: its planes need their zeros beyond every sequence's length.
    if (!p.vt_hi && !tile_has_rows(p.lens, p.t_pad, row0, BM, p.m)) return;
    const int mlast = p.m - 1, nlast = p.n - 1;
    auto la = [&](int pp, int kt) -> float4 {
        const int rc = min(row0 + arow + 16 * pp, mlast);
        const int kc = min(kt * BK + akq * 4, K - 4);
        const bool second = (kt * BK >= p.k0) && p.k1 > 0;            // wave-uniform: k0 % 64 == 0
        if (WN == 2 && (second ? p.a1_f16 : p.a0_f16)) {               // fp16 segment (wave-uniform): four halves, widened losslessly.  Wide outputs only:
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));     // in the 256-row tile (WN = 1) the second path costs 43 spilled registers
            const _Float16* src = second ? (reinterpret_cast<const _Float16*>(p.a1) + (size_t)rc * p.lda1 + (kc - p.k0))
                                         : (reinterpret_cast<const _Float16*>(p.a0) + (size_t)rc * p.lda0 + kc);
            const half4v hv = *reinterpret_cast<const half4v*>(src);
            return make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
        }
        const float* src = second ? (p.a1 + (size_t)rc * p.lda1 + (kc - p.k0)) : (p.a0 + (size_t)rc * p.lda0 + kc);
        return *reinterpret_cast<const float4*>(src);
    };
    auto oka = [&](int pp, int kt) -> bool { return (row0 + arow + 16 * pp) < p.m && (kt * BK + akq * 4) < K; };
    auto lb = [&](int pp, int kt) -> uint4 {
        const int cc = min(col0 + brow + 32 * pp, nlast);
        const int kc = min(kt * BK + bsl * 8, K - 8);
        return *reinterpret_cast<const uint4*>(w16 + (size_t)cc * K + kc);
    };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + brow + 32 * pp) < p.n && (kt * BK + bsl * 8) < K; };
    auto adv = [](int) {};
    f32x16 acc[MI][2];
    if constexpr (LNA) {
        for (int i = tid; i < K; i += NT) { lngb[i] = p.ln_gamma[i]; lngb[K + i] = p.ln_beta[i]; }
        LnGeluXf<C::PA, BK> xf;
        xf.gb = lngb; xf.K = K; xf.kq = akq;
#pragma unroll
        for (int pp = 0; pp < C::PA; ++pp) xf.rstd[pp] = (row0 + arow + 16 * pp) < p.m ? ln_rstd(p, row0 + arow + 16 * pp, K) : 0.f;
        __syncthreads();
        mainloop<MI, WN>(smem, adv, la, oka, lb, okb, (K + BK - 1) / BK, acc, xf);
    } else {
        mainloop<MI, WN>(smem, adv, la, oka, lb, okb, (K + BK - 1) / BK, acc);
    }
    // fp16 outputs (q / k, the hidden layer, ...) leave as whole 16-byte row segments through the now idle staging memory
    _Float16* stage = (p.out16 && !p.out16_lo && p.n % 8 == 0 && p.ldo16 % 8 == 0) ? reinterpret_cast<_Float16*>(&smem) : nullptr;
    linear_epilogue<MI, WN>(p, acc, p.out, row0, col0, BM, BN, stage);
}

// split-fp16 variant (gemm_core_x3.h): wh / wl = the weight matrix * w_scale split into two fp16 planes [n][K] on the
// host; activations are split while they are staged.  inv = 1 / (ACT_SCALE * w_scale) undoes both scales (exact).
// LNA: the A operand is the hidden layer of an MLP tail before its LayerNorm + GELU, applied while it is staged (LnGeluXf)
template <int MI, int WN, bool LNA = false>
__global__ __launch_bounds__(gemmx3::NT, 2) void linear_x3_kernel(LinArgs p, const _Float16* __restrict__ wh,
                                                                   const _Float16* __restrict__ wl, float inv) {
    using namespace gemmx3;
    using C = Cfg<MI, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    __shared__ Smem<MI, WN> smem;
    __shared__ __attribute__((aligned(16))) float lngb[LNA ? 2 * 1024 : 4];
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int K = p.k0 + p.k1;
    const int tid = threadIdx.x;
    const int arow = tid >> 3, akq = tid & 7, brow = tid >> 2, bsl = tid & 3;
    const int row0 = tm * BM, col0 = tn * BN;
    if (!tile_has_rows(p.lens, p.t_pad, row0, BM, p.m)) return;
    const int mlast = p.m - 1, nlast = p.n - 1;
    const float* arow0[C::PA];
    const float* arow1[C::PA];
    unsigned rowok = 0u;
#pragma unroll
    for (int pp = 0; pp < C::PA; ++pp) {
        const int rc = min(row0 + arow + 32 * pp, mlast);
        arow0[pp] = p.a0 + (size_t)rc * p.lda0;
        arow1[pp] = p.a1 ? p.a1 + (size_t)rc * p.lda1 - p.k0 : arow0[pp];
        rowok |= (row_valid(p.lens, p.t_pad, row0 + arow + 32 * pp, p.m) ? 1u : 0u) << pp;
    }
    size_t boff[C::PB];
#pragma unroll
    for (int pp = 0; pp < C::PB; ++pp) boff[pp] = (size_t)min(col0 + brow + 64 * pp, nlast) * K;
    auto la = [&](int pp, int kt) -> float4 {
        const int kc = min(kt * BK + akq * 4, K - 4);
        const bool second = (kt * BK >= p.k0) && p.k1 > 0;            // wave-uniform: k0 % 32 == 0
        return *reinterpret_cast<const float4*>((second ? arow1[pp] : arow0[pp]) + kc);
    };
    auto oka = [&](int pp, int kt) -> bool { return ((rowok >> pp) & 1u) && (kt * BK + akq * 4) < K; };
    auto lb = [&](int pp, int kt, int plane) -> uint4 {
        const int kc = min(kt * BK + bsl * 8, K - 8);
        return *reinterpret_cast<const uint4*>((plane ? wl : wh) + boff[pp] + kc);
    };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + brow + 64 * pp) < p.n && (kt * BK + bsl * 8) < K; };
    auto adv = [](int) {};
    f32x16 acc[MI][2];
    float amax = 0.f;
    if constexpr (LNA) {
        // prologue loads all in flight together (gamma | beta, the rows' partial sums of squares): unpredicated, clamped addresses
        {
            float gv[4], bv[4];      // K <= 1024, NT = 256
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int i = min(tid + NT * j, K - 1); gv[j] = p.ln_gamma[i]; bv[j] = p.ln_beta[i]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (tid + NT * j < K) { lngb[tid + NT * j] = gv[j]; lngb[K + tid + NT * j] = bv[j]; }
        }
        LnGeluXf<C::PA> xf;
        xf.gb = lngb; xf.K = K; xf.kq = akq;
#pragma unroll
        for (int pp = 0; pp < C::PA; ++pp) {
            const float rs = ln_rstd(p, min(row0 + arow + 32 * pp, mlast), K);
            xf.rstd[pp] = ((rowok >> pp) & 1u) ? rs : 0.f;
        }
        __syncthreads();
        mainloop<MI, WN>(smem, adv, la, oka, lb, okb, (K + BK - 1) / BK, p.act_scale, acc, amax, xf);
    } else {
        mainloop<MI, WN>(smem, adv, la, oka, lb, okb, (K + BK - 1) / BK, p.act_scale, acc, amax);
    }
    x3_range_flag(p.status, amax);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= inv;
    linear_epilogue<MI, WN>(p, acc, p.out, row0, col0, BM, BN);
}

// Both operands as split planes (gemm_core_x3.h::mainloop_planes): A = [a0 | a1] with planes (a0h, a0l) [m][lda0] and, for the
// concatenated input, (a1h, a1l) [m][lda1] — halves, written by the producing kernels' epilogues.
struct PlaneArgs {
    const _Float16* a0h; const _Float16* a0l; int lda0;
    const _Float16* a1h; const _Float16* a1l; int lda1;
};

template <int MI, int WN>
__global__ __launch_bounds__(gemmx3::NT, 2) void linear_x3p_kernel(LinArgs p, PlaneArgs a, const _Float16* __restrict__ wh,
                                                                    const _Float16* __restrict__ wl, float inv) {
    using namespace gemmx3;
    using C = Cfg<MI, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    __shared__ Smem<MI, WN> smem;
    if (const int z = blockIdx.y) {      // batched (pram_bgemm_nt_x3p_f32): per-z strides, in halves for the planes, floats for out
        a.a0h += z * p.sa; a.a0l += z * p.sa; wh += z * p.sw; wl += z * p.sw; p.out += z * p.so;
    }
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int K = p.k0 + p.k1;
    const int tid = threadIdx.x;
    const int srow = tid >> 2, ssl = tid & 3;
    const int row0 = tm * BM, col0 = tn * BN;
    const int mlast = p.m - 1, nlast = p.n - 1;
    size_t aoff0[BM / 64], aoff1[BM / 64], boff[BN / 64];
#pragma unroll
    for (int pp = 0; pp < BM / 64; ++pp) {
        const size_t rc = (size_t)min(row0 + srow + 64 * pp, mlast);
        aoff0[pp] = rc * a.lda0 + ssl * 8;
        aoff1[pp] = rc * a.lda1 + ssl * 8;
    }
#pragma unroll
    for (int pp = 0; pp < BN / 64; ++pp) boff[pp] = (size_t)min(col0 + srow + 64 * pp, nlast) * K + ssl * 8;
    auto la = [&](int pp, int kt, int plane) -> uint4 {
        const int k = kt * BK;
        if (k >= p.k0 && p.k1 > 0)                                       // wave-uniform: k0 % 32 == 0
            return *reinterpret_cast<const uint4*>((plane ? a.a1l : a.a1h) + aoff1[pp] + (k - p.k0));
        return *reinterpret_cast<const uint4*>((plane ? a.a0l : a.a0h) + aoff0[pp] + k);
    };
    auto oka = [&](int pp, int kt) -> bool { return (row0 + srow + 64 * pp) < p.m; };
    auto lb = [&](int pp, int kt, int plane) -> uint4 { return *reinterpret_cast<const uint4*>((plane ? wl : wh) + boff[pp] + kt * BK); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + srow + 64 * pp) < p.n; };
    f32x16 acc[MI][2];
    mainloop_planes<MI, WN>(smem, la, oka, lb, okb, K / BK, acc);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= inv;
    linear_epilogue<MI, WN>(p, acc, p.out, row0, col0, BM, BN);
}

// Wide-tile split-fp16 GEMM (gemm_core_x3w.h): 256 x 256 or 128 x 256 outputs per 512-thread workgroup, A as fp32 (split while
// staged) or as pre-split planes.  Same arithmetic and accumulation order as linear_x3_kernel: bit-identical results.
template <int MI, int WM, int WN, bool APLANES, int ABL = 0, int DMA = 1, bool LNA = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void linear_x3w_kernel(LinArgs p, PlaneArgs a, const _Float16* __restrict__ wh,
                                                                      const _Float16* __restrict__ wl, float inv) {
    using namespace gemmx3w;
    using C = Cfg<MI, WM, WN>;
    constexpr int BM = C::BM, BN = C::BN;
    static_assert(!(LNA && APLANES), "the LayerNorm + GELU transform applies to an fp32 A operand");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<MI, WM, WN>& smem = *reinterpret_cast<Smem<MI, WM, WN>*>(smem_raw);
    float* lngb = reinterpret_cast<float*>(smem_raw + sizeof(Smem<MI, WM, WN>));      // LNA: gamma | beta, 2 K floats behind the stages
    // the plane pointers as four separate scalars: selected as `plane ? lo : hi` out of the (by-value) argument struct, the pair
    // was read through a dynamically indexed private copy of the struct (56 bytes of scratch, round 2's ISA metadata)
    const _Float16* a0h = a.a0h;
    const _Float16* a0l = a.a0l;
    const _Float16* a1h = a.a1h;
    const _Float16* a1l = a.a1l;
    const int alda0 = a.lda0, alda1 = a.lda1;
    if constexpr (APLANES) {
        if (const int z = blockIdx.y) {  // batched (pram_bgemm_nt_x3p_f32): per-z strides, in halves for the planes, floats for out
            a0h += z * p.sa; a0l += z * p.sa; wh += z * p.sw; wl += z * p.sw; p.out += z * p.so;
        }
    }
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int tn = id % p.tiles_n, tm = id / p.tiles_n;
    const int K = p.k0 + p.k1;
    const int tid = threadIdx.x;
    const int arow = tid >> 3, akq = tid & 7, qrow = tid >> 2, qsl = tid & 3;
    const int row0 = tm * BM, col0 = tn * BN;
    if (!tile_has_rows(p.lens, p.t_pad, row0, BM, p.m)) return;
    const int mlast = p.m - 1, nlast = p.n - 1;
    size_t boff[C::QB];
#pragma unroll
    for (int pp = 0; pp < C::QB; ++pp) boff[pp] = (size_t)min(col0 + qrow + C::RQ * pp, nlast) * K + qsl * 8;
    auto lb = [&](int pp, int kt, int plane) -> uint4 { return *reinterpret_cast<const uint4*>((plane ? wl : wh) + boff[pp] + kt * BK); };
    auto okb = [&](int pp, int kt) -> bool { return (col0 + qrow + C::RQ * pp) < p.n; };
    auto adv = [](int) {};
    auto bptr = [&](int row, int plane, int kt) -> const _Float16* {
        return (plane ? wl : wh) + (size_t)min(col0 + row, nlast) * K + kt * BK;
    };
    f32x16 acc[MI][2];
    if constexpr (APLANES) {
        size_t aoff0[C::QA], aoff1[C::QA];
#pragma unroll
        for (int pp = 0; pp < C::QA; ++pp) {
            const size_t rc = (size_t)min(row0 + qrow + C::RQ * pp, mlast);
            aoff0[pp] = rc * alda0 + qsl * 8;
            aoff1[pp] = rc * alda1 + qsl * 8;
        }
        auto la = [&](int pp, int kt, int plane) -> uint4 {
            const int k = kt * BK;
            if (k >= p.k0 && p.k1 > 0)                                       // wave-uniform: k0 % 32 == 0
                return *reinterpret_cast<const uint4*>((plane ? a1l : a1h) + aoff1[pp] + (k - p.k0));
            return *reinterpret_cast<const uint4*>((plane ? a0l : a0h) + aoff0[pp] + k);
        };
        auto oka = [&](int pp, int kt) -> bool { return (row0 + qrow + C::RQ * pp) < p.m; };
        auto aptr = [&](int row, int plane, int kt) -> const _Float16* {
            const size_t rc = (size_t)min(row0 + row, mlast);
            const int k = kt * BK;
            if (k >= p.k0 && p.k1 > 0) return (plane ? a1l : a1h) + rc * alda1 + (k - p.k0);
            return (plane ? a0l : a0h) + rc * alda0 + k;
        };
        float amax = 0.f;      // planes in: nothing is split here
        mainloop<MI, WM, WN, true, ABL, (DMA ? 2 : 0)>(smem, adv, la, oka, lb, okb, aptr, bptr, K / BK, p.act_scale, acc, amax);
    } else {
        const float* arow0[C::PA];
        const float* arow1[C::PA];
        unsigned rowok = 0u;
#pragma unroll
        for (int pp = 0; pp < C::PA; ++pp) {
            const int rc = min(row0 + arow + C::RA * pp, mlast);
            arow0[pp] = p.a0 + (size_t)rc * p.lda0;
            arow1[pp] = p.a1 ? p.a1 + (size_t)rc * p.lda1 - p.k0 : arow0[pp];
            rowok |= (row_valid(p.lens, p.t_pad, row0 + arow + C::RA * pp, p.m) ? 1u : 0u) << pp;
        }
        auto la = [&](int pp, int kt) -> float4 {
            const int kc = kt * BK + akq * 4;
            const bool second = (kt * BK >= p.k0) && p.k1 > 0;            // wave-uniform: k0 % 32 == 0
            return *reinterpret_cast<const float4*>((second ? arow1[pp] : arow0[pp]) + kc);
        };
        auto oka = [&](int pp, int kt) -> bool { return ((rowok >> pp) & 1u) != 0u; };
        auto aptr = [](int, int, int) -> const _Float16* { return nullptr; };
        float amax = 0.f;
        if constexpr (LNA) {
            // prologue loads all in flight together, as in linear_x3_kernel (K <= 1024, 512 threads: two rounds of gamma | beta)
#if defined(PRAM_LNA_ABLATE) && PRAM_LNA_ABLATE == 4      // profiling: no prologue, no transform
            LnGeluXf<C::PA> xf0;
            xf0.gb = lngb; xf0.K = K; xf0.kq = akq;
            for (int pp = 0; pp < C::PA; ++pp) xf0.rstd[pp] = 1.0f;
            mainloop<MI, WM, WN, false, ABL, (DMA ? 1 : 0)>(smem, adv, la, oka, lb, okb, aptr, bptr, K / BK, p.act_scale, acc, amax, xf0);
#else
            {
                float gv[2], bv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { const int i = min(tid + C::NT * j, K - 1); gv[j] = p.ln_gamma[i]; bv[j] = p.ln_beta[i]; }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (tid + C::NT * j < K) { lngb[tid + C::NT * j] = gv[j]; lngb[K + tid + C::NT * j] = bv[j]; }
            }
            LnGeluXf<C::PA> xf;
            xf.gb = lngb; xf.K = K; xf.kq = akq;
#pragma unroll
            for (int pp = 0; pp < C::PA; ++pp) {
                const float rs = ln_rstd(p, min(row0 + arow + C::RA * pp, mlast), K);
                xf.rstd[pp] = ((rowok >> pp) & 1u) ? rs : 0.f;
            }
            __syncthreads();
            mainloop<MI, WM, WN, false, ABL, (DMA ? 1 : 0)>(smem, adv, la, oka, lb, okb, aptr, bptr, K / BK, p.act_scale, acc, amax, xf);
#endif
        } else {
            mainloop<MI, WM, WN, false, ABL, (DMA ? 1 : 0)>(smem, adv, la, oka, lb, okb, aptr, bptr, ABL == 128 ? 1 : (ABL == 256 && (blockIdx.x & 1) && blockIdx.x < 256) ? K / BK / 2 : K / BK, p.act_scale, acc, amax);
        }
        x3_range_flag(p.status, amax);
    }
    if constexpr (ABL == 64) {      // profiling: no epilogue (the accumulators kept alive by a store that never happens)
        float sacc = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += acc[mi][ni][e];
        if (sacc == 1.2345e-30f) p.out[tid] = sacc;
        return;
    }
    unsigned long long te0 = 0ull;
    if constexpr ((ABL & 4) != 0) te0 = __builtin_readcyclecounter();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= inv;
    // the staging memory is idle now (the main loop ends on a barrier): the plane epilogue transposes through it
    _Float16* stage = (p.out16_lo && p.n % 8 == 0 && p.ldo16 % 8 == 0) ? reinterpret_cast<_Float16*>(smem_raw) : nullptr;
    linear_epilogue<MI, WN>(p, acc, p.out, ABL == 512 ? 0 : row0, col0, BM, BN, stage);      // (512, profiling: every row tile stores to rows 0..255: the stores stay in the L2)
    if constexpr ((ABL & 4) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) atomicAdd(&gemmx3w::prof[6], (unsigned long long)__builtin_readcyclecounter() - te0);
    }
}

// ---------------------------------------------------------------- LayerNorm + GELU
// One wave per row; the row (<= 1024 floats) lives in registers, mean then centred variance
// (two-pass, like torch's RowwiseMoments result to fp32 rounding), exact erf GELU.
template <int NV>  // float4 per lane
__global__ __launch_bounds__(256) void ln_gelu_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                      int ldy, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int rows, int cols, float eps,
                                                      const int* __restrict__ lens, int t_pad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (lens && (row % t_pad) >= lens[row / t_pad]) return;      // a row beyond its sequence's length: never read afterwards
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (c < cols) ? *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c) : make_float4(0, 0, 0, 0);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            float t[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                          (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = 0.5f * t[j] * (1.0f + erff(t[j] * 0.70710678118654752440f));
            *reinterpret_cast<float4*>(y + (size_t)row * ldy + c) = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

// ---------------------------------------------------------------- Fourier positional encoding
__global__ void fourier_kernel(const float* __restrict__ kpts, const float* __restrict__ wr, float cx, float cy,
                               float scale, float* __restrict__ co, float* __restrict__ si, int rows) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = idx >> 5, f = idx & 31;
    if (row >= rows) return;
    const float x = (kpts[row * 2 + 0] - cx) / scale;
    const float y = (kpts[row * 2 + 1] - cy) / scale;
    const float pr = x * wr[f * 2 + 0] + y * wr[f * 2 + 1];
    co[idx] = cosf(pr);
    si[idx] = sinf(pr);
}

template <int MI, int WN, int BKT>
void launch_linear_t(LinArgs& p, int batch, hipStream_t st) {
    using C = gemm::Cfg<MI, WN, BKT>;
    p.tiles_m = cdiv(p.m, C::BM);
    p.tiles_n = cdiv(p.n, C::BN);
    hipLaunchKernelGGL((linear_kernel<MI, WN, BKT>), dim3(p.tiles_m * p.tiles_n, batch), dim3(gemm::NT), 0, st, p);
}

// Full grids (>= 512 tiles) take the 16-deep chunk (three workgroups per CU); small ones are latency-bound per
// workgroup and keep the 32-deep chunk (half the barriers).  The concatenated input needs k0 % BK == 0.
void launch_linear(LinArgs& p, int batch, hipStream_t st) {
    int mi, wn;
    gemm::choose_tile(p.m * batch, p.n, &mi, &wn);
    if (wn == 2) { if (mi == 2) launch_linear_t<2, 2, 16>(p, batch, st); else launch_linear_t<1, 2, 32>(p, batch, st); }
    else         { if (mi == 2) launch_linear_t<2, 1, 16>(p, batch, st); else launch_linear_t<1, 1, 32>(p, batch, st); }
}

template <int MI, int WN>
void launch_linear_f16_t(LinArgs& p, const _Float16* w16, hipStream_t st) {
    using C = gemm16::Cfg<MI, WN>;
    p.tiles_m = cdiv(p.m, C::BM);
    p.tiles_n = cdiv(p.n, C::BN);
    if constexpr (WN == 2) {
        if (p.ln_ssq) {
            hipLaunchKernelGGL((linear_f16_kernel<MI, WN, true>), dim3(p.tiles_m * p.tiles_n, 1), dim3(gemm16::NT), 0, st, p, w16);
            return;
        }
    }
    hipLaunchKernelGGL((linear_f16_kernel<MI, WN>), dim3(p.tiles_m * p.tiles_n, 1), dim3(gemm16::NT), 0, st, p, w16);
}

template <int MI, int WN>
void launch_linear_x3_t(LinArgs& p, const _Float16* wh, const _Float16* wl, float inv, hipStream_t st) {
    using C = gemmx3::Cfg<MI, WN>;
    p.tiles_m = cdiv(p.m, C::BM);
    p.tiles_n = cdiv(p.n, C::BN);
    if constexpr (WN == 2) {      // the LayerNorm + GELU operand transform exists for outputs wider than 64 columns (pram_linear_x3_lngelu_f32 checks)
        if (p.ln_ssq) {
            hipLaunchKernelGGL((linear_x3_kernel<MI, WN, true>), dim3(p.tiles_m * p.tiles_n, 1), dim3(gemmx3::NT), 0, st, p, wh, wl, inv);
            return;
        }
    }
    hipLaunchKernelGGL((linear_x3_kernel<MI, WN>), dim3(p.tiles_m * p.tiles_n, 1), dim3(gemmx3::NT), 0, st, p, wh, wl, inv);
}

template <int MI, int WM, int WN, bool APLANES>
void launch_linear_x3w_t(LinArgs& p, PlaneArgs& a, const _Float16* wh, const _Float16* wl, float inv, hipStream_t st, int batch = 1) {
    using C = gemmx3w::Cfg<MI, WM, WN>;
    p.tiles_m = cdiv(p.m, C::BM);
    p.tiles_n = cdiv(p.n, C::BN);
    const size_t shm = sizeof(gemmx3w::Smem<MI, WM, WN>);
    static bool attr_set = false;      // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)linear_x3w_kernel<MI, WM, WN, APLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    if constexpr (!APLANES) {
        if (p.ln_ssq) {      // A = GELU(LayerNorm(hidden)) applied while staged: gamma | beta ride behind the stages
            const size_t shm_ln = shm + 2 * (size_t)(p.k0 + p.k1) * sizeof(float);
            static bool ln_attr_set = false;
            if (!ln_attr_set) {
                (void)hipFuncSetAttribute((const void*)linear_x3w_kernel<MI, WM, WN, false, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(shm + 2 * 1024 * sizeof(float)));
                ln_attr_set = true;
            }
            hipLaunchKernelGGL((linear_x3w_kernel<MI, WM, WN, false, 0, 1, true>), dim3(p.tiles_m * p.tiles_n, batch), dim3(C::NT), shm_ln, st, p, a, wh, wl, inv);
            return;
        }
    }
#ifdef PRAM_PROFILING      // ablations (garbage results, PRAM_OK): profiling builds only (build_variants.py TAG:linear.hip:-DPRAM_PROFILING)
    static const char* abl = getenv("PRAM_GEMM_ABLATE");
    const int ab = abl ? atoi(abl) : 0;
    if (ab == 0) { hipLaunchKernelGGL((linear_x3w_kernel<MI, WM, WN, APLANES>), dim3(p.tiles_m * p.tiles_n, batch), dim3(C::NT), shm, st, p, a, wh, wl, inv); return; }
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(C::NT), shm, st, p, a, wh, wl, inv);
    };
    if (ab == 8) go(linear_x3w_kernel<MI, WM, WN, APLANES, 0, 0>);       // register staging for both operands (no LDS-DMA)
    else if (ab == 1) go(linear_x3w_kernel<MI, WM, WN, APLANES, 1>);      // no staging after the first chunk
    else if (ab == 2) go(linear_x3w_kernel<MI, WM, WN, APLANES, 2>);      // fragments read once per chunk
    else if (ab == 4) go(linear_x3w_kernel<MI, WM, WN, APLANES, 4>);      // phase clocks -> pram_debug_gemm_phases
    else if (ab == 16) go(linear_x3w_kernel<MI, WM, WN, APLANES, 16>);    // no A staging
    else if (ab == 32) go(linear_x3w_kernel<MI, WM, WN, APLANES, 32>);    // no B DMA
    else if (ab == 64) go(linear_x3w_kernel<MI, WM, WN, APLANES, 64>);    // the shipped loop, no epilogue
    else if (ab == 128) go(linear_x3w_kernel<MI, WM, WN, APLANES, 128>);  // ONE chunk of the shipped loop + the epilogue (fp32 A only)
    else if (ab == 512) go(linear_x3w_kernel<MI, WM, WN, APLANES, 512>);  // every row tile stores to the first one's rows: the epilogue without its HBM writes
    else if (ab == 256) go(linear_x3w_kernel<MI, WM, WN, APLANES, 256>);  // odd workgroups of the first round walk half their chunks: do de-synchronised CUs overlap store bursts with main loops?
    else go(linear_x3w_kernel<MI, WM, WN, APLANES, 3>);
#else
    hipLaunchKernelGGL((linear_x3w_kernel<MI, WM, WN, APLANES>), dim3(p.tiles_m * p.tiles_n, batch), dim3(C::NT), shm, st, p, a, wh, wl, inv);
#endif
}

// wide tiles (gemm_core_x3w.h) for outputs at least 256 columns wide: 256 x 256 when that still gives every CU a workgroup,
// 128 x 256 otherwise.  PRAM_X3_TILE=narrow|w256|w128 overrides (profiling).
template <bool APLANES>
bool launch_linear_x3_wide(LinArgs& p, PlaneArgs& a, const _Float16* wh, const _Float16* wl, float inv, hipStream_t st, int batch = 1) {
    static const char* force = prof_env("PRAM_X3_TILE");
    if (force && force[0] == 'n') return false;
    if (p.n < 256 || (p.k0 + p.k1) % 32 != 0) return false;
    const long big = (long)cdiv(p.m, 256) * cdiv(p.n, 256) * batch, small = (long)cdiv(p.m, 128) * cdiv(p.n, 256) * batch;
    if (!force && small < 192) return false;      // too few wide tiles for 256 CUs: narrow tiles fill the chip better
    // One workgroup per CU: a launch takes ceil(tiles / 256) rounds.  A 128-row tile costs ~0.55 of a 256-row one (measured), so
    // the 256-row tiles lose when their last round is mostly empty — 32 768 x 768 (SegNetViT's q | k | v projection) is 384 tiles =
    // 2 rounds against 768 half tiles = 3 x 0.55 rounds.
    const bool use256 = force ? (force[1] == '2') : (big >= 224 && 100 * cdiv((int)big, 256) <= 55 * cdiv((int)small, 256));
#ifdef PRAM_PROFILING
    if constexpr (!APLANES) {
        if (force && force[0] == 'f') { launch_linear_x3w_t<2, 1, 8, false>(p, a, wh, wl, inv, st, batch); return true; }      // full-row probe: 64 x 512 tiles
    }
#endif
    if (use256) launch_linear_x3w_t<4, 2, 4, APLANES>(p, a, wh, wl, inv, st, batch);
    else launch_linear_x3w_t<2, 2, 4, APLANES>(p, a, wh, wl, inv, st, batch);
    return true;
}

template <int MI, int WN>
void launch_linear_x3p_t(LinArgs& p, PlaneArgs& a, const _Float16* wh, const _Float16* wl, float inv, hipStream_t st, int batch = 1) {
    using C = gemmx3::Cfg<MI, WN>;
    p.tiles_m = cdiv(p.m, C::BM);
    p.tiles_n = cdiv(p.n, C::BN);
    hipLaunchKernelGGL((linear_x3p_kernel<MI, WN>), dim3(p.tiles_m * p.tiles_n, batch), dim3(gemmx3::NT), 0, st, p, a, wh, wl, inv);
}

}  // namespace

/* pram_linear_x3_f32 with the activations already split ("planes": value * 16 = hi + lo, two fp16 matrices [m][lda] written
   by pram_linear_x3_f32 / pram_attention_x3_f32 / pram_layernorm_gelu_x3): the main loop stages both operands with plain
   16-byte copies.  K = k0 + k1 and k0 must be multiples of 32; lda multiples of 8. */
extern "C" int pram_linear_x3p_f32(const void* a0_hi, const void* a0_lo, int lda0, int k0, const void* a1_hi, const void* a1_lo,
                                   int lda1, int k1, const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                                   const float* residual, int ldr, float* out, int ldo, void* out_hi, void* out_lo, int ldo16,
                                   int m, int n, float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                                   void* stream) {
    PRAM_REQUIRE(a0_hi && a0_lo && w_hi && w_lo && (out || (out_hi && out_lo)), "pram_linear_x3p_f32: null pointer");
    PRAM_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), "pram_linear_x3p_f32: the split output needs both planes");
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0 && w_scale > 0.f, "pram_linear_x3p_f32: bad sizes");
    PRAM_REQUIRE(k0 % 32 == 0 && k1 % 32 == 0 && lda0 % 8 == 0, "pram_linear_x3p_f32: k0, k1 must be multiples of 32, lda of 8");
    PRAM_REQUIRE(k1 == 0 || (a1_hi && a1_lo && lda1 % 8 == 0), "pram_linear_x3p_f32: second segment needs both planes");
    if (flags & PRAM_LIN_ROTARY)
        PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "pram_linear_x3p_f32: rotary needs cos/sin and rot_cols %% 64 == 0");
    if (m == 0) return PRAM_OK;
    LinArgs p{nullptr, lda0, k0, nullptr, lda1, k1, nullptr, bias, residual, ldr, out, ldo, m, n, alpha, flags,
              rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0, out_hi, ldo16, out_lo, pram_act_scale()};
    PlaneArgs a{(const _Float16*)a0_hi, (const _Float16*)a0_lo, lda0, (const _Float16*)a1_hi, (const _Float16*)a1_lo, lda1};
    p.status = pram_status_ptr();
    p.act_scale = pram_act_scale();
    int mi, wn;
    gemm::choose_tile(m, n, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* wh = (const _Float16*)w_hi;
    const _Float16* wl = (const _Float16*)w_lo;
    const float inv = 1.0f / (pram_act_scale() * w_scale);
    if (launch_linear_x3_wide<true>(p, a, wh, wl, inv, st)) return pram_launch_status("pram_linear_x3p_f32");
    if (wn == 2) { if (mi == 2) launch_linear_x3p_t<2, 2>(p, a, wh, wl, inv, st); else launch_linear_x3p_t<1, 2>(p, a, wh, wl, inv, st); }
    else         { if (mi == 2) launch_linear_x3p_t<2, 1>(p, a, wh, wl, inv, st); else launch_linear_x3p_t<1, 1>(p, a, wh, wl, inv, st); }
    return pram_launch_status("pram_linear_x3p_f32");
}

/* Profiling aid: the per-phase shader-clock totals the wide split-fp16 GEMM accumulates when PRAM_GEMM_ABLATE=4 (see
   gemm_core_x3w.h); out72 = host array of 72 counters, reset != 0 clears them afterwards. */
extern "C" int pram_debug_gemm_phases(unsigned long long* out72, int reset) {
    PRAM_REQUIRE(out72, "pram_debug_gemm_phases: null pointer");
    if (hipMemcpyFromSymbol(out72, HIP_SYMBOL(gemmx3w::prof), 72 * sizeof(unsigned long long)) != hipSuccess) return pram_launch_status("pram_debug_gemm_phases");
    if (reset) {
        unsigned long long z[72] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gemmx3w::prof), z, sizeof(z)) != hipSuccess) return pram_launch_status("pram_debug_gemm_phases");
    }
    return PRAM_OK;
}

struct VtOut { void* hi; void* lo; int col0, heads, t_seq; };
// LayerNorm coupling of an MLP tail's two GEMMs (LinArgs::row_ssq / ln_*): ssq_out for the first, the rest for the second
struct LnIo { float* ssq_out; const float* ssq_in; int parts; const float* gamma; const float* beta; float eps; };

// partials per row the first GEMM of an MLP tail writes: one per 64-column block of its output (pram_linear_x3_ssq_parts)
static int x3_ssq_parts(int n) { return cdiv(n, 64); }

static int linear_x3_impl(const LnIo* ln, const VtOut* vt, const int* lens, int t_pad, const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi,
                                  const void* w_lo, float w_scale, const float* bias, const float* residual, int ldr,
                                  float* out, int ldo, void* out_hi, void* out_lo, int ldo16, int m, int n, float alpha,
                                  int flags, const float* rot_cos, const float* rot_sin, int rot_cols, void* stream) {
    PRAM_REQUIRE(a0 && w_hi && w_lo && (out || (out_hi && out_lo)), "pram_linear_x3_f32: null pointer");
    PRAM_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), "pram_linear_x3_f32: the split output needs both planes");
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0 && w_scale > 0.f, "pram_linear_x3_f32: bad sizes");
    PRAM_REQUIRE((k0 + k1) % 8 == 0 && lda0 % 4 == 0, "pram_linear_x3_f32: K must be a multiple of 8, lda of 4");
    PRAM_REQUIRE(k1 == 0 || (a1 && k0 % gemmx3::BK == 0 && lda1 % 4 == 0), "pram_linear_x3_f32: concat needs k0 %% 32 == 0");
    if (flags & PRAM_LIN_ROTARY)
        PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "pram_linear_x3_f32: rotary needs cos/sin and rot_cols %% 64 == 0");
    if (m == 0) return PRAM_OK;
    LinArgs p{a0, lda0, k0, a1, lda1, k1, nullptr, bias, residual, ldr, out, ldo, m, n, alpha, flags,
              rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0, out_hi, ldo16, out_lo, pram_act_scale(), lens, t_pad};
    PRAM_REQUIRE(!lens || t_pad > 0, "pram_linear_x3_f32: lens needs t_pad > 0");
    p.status = pram_status_ptr();
    p.act_scale = pram_act_scale();
    if (ln) {
        PRAM_REQUIRE(!(ln->ssq_out && (out_hi || vt)), "pram_linear_x3_ssq_f32: row sums go with an fp32 output");
        p.row_ssq = ln->ssq_out;
        if (ln->ssq_in) {
            PRAM_REQUIRE(ln->gamma && ln->beta && ln->parts > 0 && k1 == 0 && k0 % 32 == 0 && k0 <= 1024,
                         "pram_linear_x3_lngelu_f32: needs gamma / beta / parts, one input segment, K %% 32 == 0, K <= 1024");
            p.ln_ssq = ln->ssq_in; p.ln_parts = ln->parts; p.ln_gamma = ln->gamma; p.ln_beta = ln->beta; p.ln_eps = ln->eps;
        }
    }
    if (vt) {
        PRAM_REQUIRE(vt->hi && vt->lo && out_hi && out_lo, "pram_linear_x3_qkv_f32: null pointer");
        PRAM_REQUIRE(vt->t_seq > 0 && vt->t_seq % 64 == 0 && m % vt->t_seq == 0 && (!lens || t_pad == vt->t_seq),
                     "pram_linear_x3_qkv_f32: sequences must be a multiple of 64 tokens long (and t_pad == t_seq)");
        PRAM_REQUIRE(vt->heads > 0 && vt->col0 % 64 == 0 && n == vt->col0 + vt->heads * 64 && ldo16 >= vt->col0,
                     "pram_linear_x3_qkv_f32: the value heads must be the last heads * 64 columns");
        p.vt_hi = vt->hi; p.vt_lo = vt->lo; p.vt_col0 = vt->col0; p.vt_heads = vt->heads; p.vt_t = vt->t_seq; p.vt_tv = vt->t_seq;
    }
    int mi, wn;
    gemm::choose_tile(m, n, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* wh = (const _Float16*)w_hi;
    const _Float16* wl = (const _Float16*)w_lo;
    const float inv = 1.0f / (pram_act_scale() * w_scale);
    {
        PlaneArgs none{nullptr, nullptr, 0, nullptr, nullptr, 0};
        if (launch_linear_x3_wide<false>(p, none, wh, wl, inv, st)) return pram_launch_status("pram_linear_x3_f32");
    }
    if (wn == 2) { if (mi == 2) launch_linear_x3_t<2, 2>(p, wh, wl, inv, st); else launch_linear_x3_t<1, 2>(p, wh, wl, inv, st); }
    else         { if (mi == 2) launch_linear_x3_t<2, 1>(p, wh, wl, inv, st); else launch_linear_x3_t<1, 1>(p, wh, wl, inv, st); }
    return pram_launch_status("pram_linear_x3_f32");
}

extern "C" int pram_linear_x3_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi,
                                  const void* w_lo, float w_scale, const float* bias, const float* residual, int ldr,
                                  float* out, int ldo, void* out_hi, void* out_lo, int ldo16, int m, int n, float alpha,
                                  int flags, const float* rot_cos, const float* rot_sin, int rot_cols, void* stream) {
    return linear_x3_impl(nullptr, nullptr, nullptr, 0, a0, lda0, k0, a1, lda1, k1, w_hi, w_lo, w_scale, bias, residual, ldr, out, ldo, out_hi, out_lo, ldo16,
                          m, n, alpha, flags, rot_cos, rot_sin, rot_cols, stream);
}

/* pram_linear_x3_f32 on a ragged token matrix: rows are sequences of t_pad rows, sequence s has lens[s] (device int32) valid
   ones; output tiles that contain no valid row are skipped and left untouched (AdaGML: pruned / stopped pairs cost nothing). */
extern "C" int pram_linear_x3_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi,
                                         const void* w_lo, float w_scale, const float* bias, const float* residual, int ldr,
                                         float* out, int ldo, void* out_hi, void* out_lo, int ldo16, int m, int n, float alpha,
                                         int flags, const float* rot_cos, const float* rot_sin, int rot_cols, const int* lens,
                                         int t_pad, void* stream) {
    return linear_x3_impl(nullptr, nullptr, lens, t_pad, a0, lda0, k0, a1, lda1, k1, w_hi, w_lo, w_scale, bias, residual, ldr, out, ldo, out_hi, out_lo, ldo16,
                          m, n, alpha, flags, rot_cos, rot_sin, rot_cols, stream);
}

/* MLP tail, first GEMM (nets/segnetvit.py:87-95 mlp.0; nets/gml.py:118-126; the seg head segnetvit.py:157-164): pram_linear_x3_ragged_f32
   that also writes, per output row, the sum of squares of its outputs — one partial per 64-column block, row_ssq [parts][m] with
   parts = pram_linear_x3_ssq_parts(m, n, k0 + k1) = ceil(n / 64): summed in ascending order by the consumer, the same bits for
   every tile configuration.  With the weights CENTRED over the outputs on the host (w - mean_j w[j], b - mean b)
   the output is h - mean(h) and sum(row_ssq) / n is the LayerNorm's variance: the second GEMM normalises while it stages. */
extern "C" int pram_linear_x3_ssq_parts(int m, int n, int k) { (void)m; (void)k; return x3_ssq_parts(n); }

extern "C" int pram_linear_x3_ssq_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi, const void* w_lo,
                                      float w_scale, const float* bias, float* out, int ldo, float* row_ssq, int m, int n,
                                      const int* lens, int t_pad, void* stream) {
    PRAM_REQUIRE(row_ssq && out, "pram_linear_x3_ssq_f32: null pointer");
    LnIo ln{row_ssq, nullptr, 0, nullptr, nullptr, 0.f};
    return linear_x3_impl(&ln, nullptr, lens, t_pad, a0, lda0, k0, a1, lda1, k1, w_hi, w_lo, w_scale, bias, nullptr, 0, out, ldo, nullptr, nullptr, 0,
                          m, n, 1.0f, 0, nullptr, nullptr, 0, stream);
}

/* MLP tail, second GEMM: out = GELU(LayerNorm(hidden)) w^T + bias + residual with the LayerNorm + GELU (nn.LayerNorm + nn.GELU of
   the reference's nn.Sequential) applied to the A operand while it is staged: hidden [m][k] = the CENTRED output of
   pram_linear_x3_ssq_f32, ln_ssq [parts][m] its row sums, gamma / beta [k], rstd = 1 / sqrt(sum_p ln_ssq[p][row] / k + eps).
   GELU through a degree-7 fit of the Gaussian tail (gelu_erf: 7.5e-8).  k % 32 == 0, k <= 1024. */
extern "C" int pram_linear_x3_lngelu_f32(const float* hidden, int ldh, int k, const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                                         const float* residual, int ldr, float* out, int ldo, int m, int n, const float* ln_ssq, int parts,
                                         const float* gamma, const float* beta, float eps, const int* lens, int t_pad, void* stream) {
    PRAM_REQUIRE(ln_ssq && out, "pram_linear_x3_lngelu_f32: null pointer");
    PRAM_REQUIRE(n > 64, "pram_linear_x3_lngelu_f32: n = %d must exceed 64 (narrower outputs: pram_layernorm_gelu_f32 + pram_linear_x3_f32)", n);
    LnIo ln{nullptr, ln_ssq, parts, gamma, beta, eps};
    return linear_x3_impl(&ln, nullptr, lens, t_pad, hidden, ldh, k, nullptr, 0, 0, w_hi, w_lo, w_scale, bias, residual, ldr, out, ldo, nullptr, nullptr, 0,
                          m, n, 1.0f, 0, nullptr, nullptr, 0, stream);
}

/* The q | k | v projection of an attention block in one call (nets/segnetvit.py:87-95, nets/gml.py:151-159): columns
   [0, vt_col0) leave as row-major split planes [m][ldo16] (the q / k operands of pram_attention_x3_f32), the last heads * 64
   columns — the values — leave as the transposed, key-permuted planes [m / t_seq][heads][64][t_seq] that
   pram_attention_x3_vt would build from them, zeros for tokens >= lens[s].  t_seq % 64 == 0.  lens may be NULL. */
extern "C" int pram_linear_x3_qkv_f32(const float* a0, int lda0, int k0, const void* w_hi, const void* w_lo, float w_scale,
                                      const float* bias, void* out_hi, void* out_lo, int ldo16, void* vt_hi, void* vt_lo, int vt_col0,
                                      int heads, int t_seq, int m, int n, int flags, const float* rot_cos, const float* rot_sin,
                                      int rot_cols, const int* lens, void* stream) {
    VtOut vt{vt_hi, vt_lo, vt_col0, heads, t_seq};
    return linear_x3_impl(nullptr, &vt, lens, lens ? t_seq : 0, a0, lda0, k0, nullptr, 0, 0, w_hi, w_lo, w_scale, bias, nullptr, 0, nullptr, 0, out_hi,
                          out_lo, ldo16, m, n, 1.0f, flags, rot_cos, rot_sin, rot_cols, stream);
}

static int linear_f16_f32_impl(const char* who, const int* lens, int t_pad, const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                               const void* w16, const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                               float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols, void* stream) {
    PRAM_REQUIRE(a0 && w16 && out, "%s: null pointer", who);
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0, "%s: bad sizes", who);
    PRAM_REQUIRE((k0 + k1) % 8 == 0 && lda0 % 4 == 0, "%s: K must be a multiple of 8, lda of 4", who);
    PRAM_REQUIRE(k1 == 0 || (a1 && k0 % gemm16::BK == 0 && lda1 % 4 == 0), "%s: concat needs k0 %% 64 == 0", who);
    if (flags & PRAM_LIN_ROTARY)
        PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "%s: rotary needs cos/sin and rot_cols %% 64 == 0", who);
    if (m == 0) return PRAM_OK;
    LinArgs p{a0, lda0, k0, a1, lda1, k1, nullptr, bias, residual, ldr, out, ldo, m, n, alpha, flags,
              rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0};
    p.lens = lens;
    p.t_pad = t_pad;
    int mi, wn;
    gemm::choose_tile(m, n, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* w = (const _Float16*)w16;
    if (wn == 2) { if (mi == 2) launch_linear_f16_t<2, 2>(p, w, st); else launch_linear_f16_t<1, 2>(p, w, st); }
    else launch_linear_f16_t<1, 1>(p, w, st);      // 128-row tiles for narrow outputs: the 256-row instantiation spills (43 registers) on this path
    return pram_launch_status(who);
}

extern "C" int pram_linear_f16_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w16,
                                   const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                                   float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                                   void* stream) {
    return linear_f16_f32_impl("pram_linear_f16_f32", nullptr, 0, a0, lda0, k0, a1, lda1, k1, w16, bias, residual, ldr, out, ldo, m, n, alpha, flags,
                               rot_cos, rot_sin, rot_cols, stream);
}

/* pram_linear_f16_f32 on a ragged token matrix (as pram_linear_x3_ragged_f32 / pram_linear_ragged_f32): rows are sequences of
   t_pad rows, sequence s has lens[s] (device int32) valid ones; output tiles without a valid row are skipped and only valid rows
   are stored — `out` may be a persistent buffer whose other rows belong to someone else (AdaGML's matching descriptors). */
extern "C" int pram_linear_f16_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w16,
                                          const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                                          float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                                          const int* lens, int t_pad, void* stream) {
    PRAM_REQUIRE(!lens || t_pad > 0, "pram_linear_f16_ragged_f32: lens needs t_pad > 0");
    return linear_f16_f32_impl("pram_linear_f16_ragged_f32", lens, t_pad, a0, lda0, k0, a1, lda1, k1, w16, bias, residual, ldr, out, ldo, m, n, alpha,
                               flags, rot_cos, rot_sin, rot_cols, stream);
}

extern "C" int pram_linear_f16_h16(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w16,
                                   const float* bias, const float* residual, int ldr, float* out, int ldo, void* out16,
                                   int ldo16, int m, int n, float alpha, int flags, const float* rot_cos,
                                   const float* rot_sin, int rot_cols, void* stream) {
    PRAM_REQUIRE(a0 && w16 && out16, "pram_linear_f16_h16: null pointer");
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0, "pram_linear_f16_h16: bad sizes");
    PRAM_REQUIRE((k0 + k1) % 8 == 0 && lda0 % 4 == 0, "pram_linear_f16_h16: K must be a multiple of 8, lda of 4");
    PRAM_REQUIRE(k1 == 0 || (a1 && k0 % gemm16::BK == 0 && lda1 % 4 == 0), "pram_linear_f16_h16: concat needs k0 %% 64 == 0");
    if (flags & PRAM_LIN_ROTARY)
        PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "pram_linear_f16_h16: rotary needs cos/sin and rot_cols %% 64 == 0");
    if (m == 0) return PRAM_OK;
    LinArgs p{a0, lda0, k0, a1, lda1, k1, nullptr, bias, residual, ldr, out, ldo, m, n, alpha, flags,
              rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0, out16, ldo16};
    int mi, wn;
    gemm::choose_tile(m, n, &mi, &wn);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* w = (const _Float16*)w16;
    if (wn == 2) { if (mi == 2) launch_linear_f16_t<2, 2>(p, w, st); else launch_linear_f16_t<1, 2>(p, w, st); }
    else launch_linear_f16_t<1, 1>(p, w, st);      // 128-row tiles for narrow outputs: the 256-row instantiation spills (43 registers) on this path
    return pram_launch_status("pram_linear_f16_h16");
}

static int linear_f32_impl(const int* lens, int t_pad, const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                               const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                               float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                               void* stream) {
    PRAM_REQUIRE(a0 && w && out, "pram_linear_f32: null pointer");
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0, "pram_linear_f32: bad sizes m=%d n=%d k0=%d k1=%d", m, n, k0, k1);
    PRAM_REQUIRE((k0 + k1) % 4 == 0 && lda0 % 4 == 0, "pram_linear_f32: K and lda must be multiples of 4");
    PRAM_REQUIRE(k1 == 0 || (a1 && k0 % 32 == 0 && lda1 % 4 == 0), "pram_linear_f32: concat needs k0 %% 32 == 0");
    if (flags & PRAM_LIN_ROTARY)
        PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "pram_linear_f32: rotary needs cos/sin and rot_cols %% 64 == 0");
    if (m == 0) return PRAM_OK;
    LinArgs p{a0, lda0, k0, a1, lda1, k1, w, bias, residual, ldr, out, ldo, m, n, alpha, flags,
              rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0};
    PRAM_REQUIRE(!lens || t_pad > 0, "pram_linear_f32: lens needs t_pad > 0");
    p.lens = lens;
    p.t_pad = t_pad;
    launch_linear(p, 1, (hipStream_t)stream);
    return pram_launch_status("pram_linear_f32");
}

extern "C" int pram_linear_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                               const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                               float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                               void* stream) {
    return linear_f32_impl(nullptr, 0, a0, lda0, k0, a1, lda1, k1, w, bias, residual, ldr, out, ldo, m, n, alpha, flags, rot_cos, rot_sin,
                           rot_cols, stream);
}

extern "C" int pram_linear_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                                      const float* bias, const float* residual, int ldr, float* out, int ldo, int m, int n,
                                      float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                                      const int* lens, int t_pad, void* stream) {
    return linear_f32_impl(lens, t_pad, a0, lda0, k0, a1, lda1, k1, w, bias, residual, ldr, out, ldo, m, n, alpha, flags, rot_cos, rot_sin,
                           rot_cols, stream);
}

extern "C" int pram_bgemm_nt_f32(const float* a, int lda, long long stride_a, const float* b, int ldb,
                                 long long stride_b, float* c, int ldc, long long stride_c, int batch, int m_max,
                                 int n_max, int k, float alpha, void* stream) {
    PRAM_REQUIRE(a && b && c, "pram_bgemm_nt_f32: null pointer");
    PRAM_REQUIRE(k % 4 == 0 && lda % 4 == 0 && ldb == k, "pram_bgemm_nt_f32: need k %% 4 == 0, lda %% 4 == 0, ldb == k");
    if (batch == 0 || m_max == 0 || n_max == 0) return PRAM_OK;
    LinArgs p{a, lda, k, nullptr, 0, 0, b, nullptr, nullptr, 0, c, ldc, m_max, n_max, alpha, 0,
              nullptr, nullptr, 0, stride_a, stride_b, stride_c, 0, 0};
    launch_linear(p, batch, (hipStream_t)stream);
    return pram_launch_status("pram_bgemm_nt_f32");
}

/* pram_bgemm_nt_f32 on the split-fp16 path: c[z] = alpha * a[z] b[z]^T with both operands given as split planes (value * 16 =
   hi + lo, as the projection epilogues write them): a [m][lda], b [n][ldb == k] per batch element, strides in halves (a, b)
   and floats (c).  k % 32 == 0.  The matcher's score matrix mdesc0 . mdesc1^T (nets/gml.py:253). */
extern "C" int pram_bgemm_nt_x3p_f32(const void* a_hi, const void* a_lo, int lda, long long stride_a, const void* b_hi, const void* b_lo,
                                     int ldb, long long stride_b, float* c, int ldc, long long stride_c, int batch, int m_max, int n_max,
                                     int k, float alpha, void* stream) {
    PRAM_REQUIRE(a_hi && a_lo && b_hi && b_lo && c, "pram_bgemm_nt_x3p_f32: null pointer");
    PRAM_REQUIRE(k > 0 && k % 32 == 0 && lda % 8 == 0 && ldb == k && stride_a % 8 == 0 && stride_b % 8 == 0,
                 "pram_bgemm_nt_x3p_f32: need k %% 32 == 0, lda %% 8 == 0, ldb == k, plane strides %% 8 == 0");
    if (batch == 0 || m_max == 0 || n_max == 0) return PRAM_OK;
    LinArgs p{nullptr, lda, k, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, c, ldc, m_max, n_max, alpha, 0,
              nullptr, nullptr, 0, stride_a, stride_b, stride_c, 0, 0, nullptr, 0, nullptr, pram_act_scale()};
    PlaneArgs a{(const _Float16*)a_hi, (const _Float16*)a_lo, lda, nullptr, nullptr, 0};
    const _Float16* wh = (const _Float16*)b_hi;
    const _Float16* wl = (const _Float16*)b_lo;
    const float inv = 1.0f / (pram_act_scale() * pram_act_scale());
    hipStream_t st = (hipStream_t)stream;
    if (launch_linear_x3_wide<true>(p, a, wh, wl, inv, st, batch)) return pram_launch_status("pram_bgemm_nt_x3p_f32");
    int mi, wn;
    gemm::choose_tile(m_max, n_max, &mi, &wn);
    if (wn == 2) { if (mi == 2) launch_linear_x3p_t<2, 2>(p, a, wh, wl, inv, st, batch); else launch_linear_x3p_t<1, 2>(p, a, wh, wl, inv, st, batch); }
    else         { if (mi == 2) launch_linear_x3p_t<2, 1>(p, a, wh, wl, inv, st, batch); else launch_linear_x3p_t<1, 1>(p, a, wh, wl, inv, st, batch); }
    return pram_launch_status("pram_bgemm_nt_x3p_f32");
}

static int layernorm_gelu_impl(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, int rows, int cols,
                               float eps, const int* lens, int t_pad, void* stream, const char* who) {
    PRAM_REQUIRE(x && y && gamma && beta, "%s: null pointer", who);
    PRAM_REQUIRE(cols > 0 && cols <= 1024 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
                 "%s: cols=%d must be <= 1024 and a multiple of 4", who, cols);
    PRAM_REQUIRE(!lens || t_pad > 0, "%s: lens needs t_pad > 0", who);
    if (rows == 0) return PRAM_OK;
    dim3 grid(cdiv(rows, 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (cols <= 256) hipLaunchKernelGGL(ln_gelu_kernel<1>, grid, blk, 0, st, x, ldx, y, ldy, gamma, beta, rows, cols, eps, lens, t_pad);
    else if (cols <= 512) hipLaunchKernelGGL(ln_gelu_kernel<2>, grid, blk, 0, st, x, ldx, y, ldy, gamma, beta, rows, cols, eps, lens, t_pad);
    else hipLaunchKernelGGL(ln_gelu_kernel<4>, grid, blk, 0, st, x, ldx, y, ldy, gamma, beta, rows, cols, eps, lens, t_pad);
    return pram_launch_status(who);
}

extern "C" int pram_layernorm_gelu_f32(const float* x, int ldx, float* y, int ldy, const float* gamma,
                                       const float* beta, int rows, int cols, float eps, void* stream) {
    return layernorm_gelu_impl(x, ldx, y, ldy, gamma, beta, rows, cols, eps, nullptr, 0, stream, "pram_layernorm_gelu_f32");
}

extern "C" int pram_layernorm_gelu_ragged_f32(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                              int rows, int cols, float eps, const int* lens, int t_pad, void* stream) {
    return layernorm_gelu_impl(x, ldx, y, ldy, gamma, beta, rows, cols, eps, lens, t_pad, stream, "pram_layernorm_gelu_ragged_f32");
}

extern "C" int pram_fourier_encoding_f32(const float* kpts, const float* wr, float cx, float cy, float scale,
                                         float* cos_out, float* sin_out, int rows, void* stream) {
    PRAM_REQUIRE(kpts && wr && cos_out && sin_out, "pram_fourier_encoding_f32: null pointer");
    if (rows == 0) return PRAM_OK;
    hipLaunchKernelGGL(fourier_kernel, dim3(cdiv(rows * 32, 256)), dim3(256), 0, (hipStream_t)stream, kpts, wr, cx, cy,
                       scale, cos_out, sin_out, rows);
    return pram_launch_status("pram_fourier_encoding_f32");
}

/* ---- fp16 MFMA path (BASELINE C5) with fp16 intermediates in HBM: the values a consuming GEMM would round to fp16 while staging
   them are rounded where they are produced instead — q / k / v, the attention context and the MLP's hidden layer travel as 2 bytes
   per element, the residual stream stays fp32.  Same results as the fp32-intermediate form except for the hidden layer, which is
   rounded before its LayerNorm instead of after its GELU (this path's own tolerance). */
static int linear_f16_impl(const char* who, const float* a0, int lda0, int k0, int a0_f16, const float* a1, int lda1, int k1, int a1_f16,
                           const void* w16, const float* bias, const float* residual, int ldr, float* out, int ldo, void* out16, int ldo16,
                           const LnIo* ln, const VtOut* vt, const int* lens, int m, int n, int flags, const float* rot_cos,
                           const float* rot_sin, int rot_cols, void* stream) {
    PRAM_REQUIRE(a0 && w16 && (out || out16), "%s: null pointer", who);
    PRAM_REQUIRE(m >= 0 && n > 0 && k0 > 0 && k1 >= 0, "%s: bad sizes", who);
    PRAM_REQUIRE((k0 + k1) % 8 == 0 && lda0 % 4 == 0 && (k1 == 0 || (a1 && k0 % gemm16::BK == 0 && lda1 % 4 == 0)), "%s: K %% 8, lda %% 4, concat needs k0 %% 64 == 0", who);
    if (flags & PRAM_LIN_ROTARY) PRAM_REQUIRE(rot_cos && rot_sin && rot_cols % 64 == 0, "%s: rotary needs cos/sin and rot_cols %% 64 == 0", who);
    if (m == 0) return PRAM_OK;
    LinArgs p{a0, lda0, k0, a1, lda1, k1, nullptr, bias, residual, ldr, out, ldo, m, n, 1.0f, flags, rot_cos, rot_sin, rot_cols, 0, 0, 0, 0, 0, out16, ldo16};
    p.a0_f16 = a0_f16;
    p.a1_f16 = a1_f16;
    if (ln) {
        p.row_ssq = ln->ssq_out;
        if (ln->ssq_in) {
            PRAM_REQUIRE(ln->gamma && ln->beta && ln->parts > 0 && k1 == 0 && k0 % 64 == 0 && k0 <= 1024 && n > 64, "%s: needs gamma / beta / parts, one input segment, K %% 64 == 0, K <= 1024, n > 64", who);
            p.ln_ssq = ln->ssq_in; p.ln_parts = ln->parts; p.ln_gamma = ln->gamma; p.ln_beta = ln->beta; p.ln_eps = ln->eps;
        }
    }
    if (vt) {
        PRAM_REQUIRE(vt->hi && out16 && vt->t_seq > 0 && vt->t_seq % 64 == 0 && m % vt->t_seq == 0 && vt->heads > 0 && vt->col0 % 64 == 0 &&
                     n == vt->col0 + vt->heads * 64 && ldo16 >= vt->col0, "%s: the value heads must be the last heads * 64 columns, sequences a multiple of 64 tokens", who);
        p.vt_hi = vt->hi; p.vt_lo = nullptr; p.vt_col0 = vt->col0; p.vt_heads = vt->heads; p.vt_t = vt->t_seq; p.vt_tv = vt->t_seq;
        p.out16_scale = 1.0f;
        p.lens = lens;      // only the V^T epilogue looks at it (zeros beyond a sequence's length); the fp16 GEMM computes every row
        p.t_pad = vt->t_seq;
    }
    int mi, wn;
    gemm::choose_tile(m, n, &mi, &wn);
    PRAM_REQUIRE(wn == 2 || !(a0_f16 || a1_f16), "%s: fp16 input segments need an output wider than 64 columns (n = %d)", who, n);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* w = (const _Float16*)w16;
    if (wn == 2) { if (mi == 2) launch_linear_f16_t<2, 2>(p, w, st); else launch_linear_f16_t<1, 2>(p, w, st); }
    else launch_linear_f16_t<1, 1>(p, w, st);      // 128-row tiles for narrow outputs: the 256-row instantiation spills (43 registers) on this path
    return pram_launch_status(who);
}

/* q | k | v projection of an attention block on the fp16 path: columns [0, vt_col0) as fp16 row-major [m][ldo16], the value heads as
   the transposed key-permuted fp16 values [m / t_seq][heads][64][t_seq] pram_attention_h16t_* reads (zeros beyond lens[s]). */
extern "C" int pram_linear_f16_qkv_h16(const float* a0, int lda0, int k0, const void* w16, const float* bias, void* out16, int ldo16, void* vt16,
                                       int vt_col0, int heads, int t_seq, int m, int n, int flags, const float* rot_cos, const float* rot_sin,
                                       int rot_cols, const int* lens, void* stream) {
    VtOut vt{vt16, nullptr, vt_col0, heads, t_seq};
    return linear_f16_impl("pram_linear_f16_qkv_h16", a0, lda0, k0, 0, nullptr, 0, 0, 0, w16, bias, nullptr, 0, nullptr, 0, out16, ldo16, nullptr, &vt, lens,
                           m, n, flags, rot_cos, rot_sin, rot_cols, stream);
}

/* MLP tail on the fp16 path, first GEMM: [a0 (fp32) | a1 (fp16, lda1 in halves)] w^T + bias with host-centred weights -> the hidden
   layer as fp16 [m][ldo16] and its rows' sums of squares (taken from the fp32 values), parts = ceil(n / 64). */
extern "C" int pram_linear_f16_ssq_h16(const float* a0, int lda0, int k0, const void* a1_16, int lda1, int k1, const void* w16, const float* bias,
                                       void* out16, int ldo16, float* row_ssq, int m, int n, void* stream) {
    PRAM_REQUIRE(row_ssq && out16, "pram_linear_f16_ssq_h16: null pointer");
    LnIo ln{row_ssq, nullptr, 0, nullptr, nullptr, 0.f};
    return linear_f16_impl("pram_linear_f16_ssq_h16", a0, lda0, k0, 0, reinterpret_cast<const float*>(a1_16), lda1, k1, 1, w16, bias, nullptr, 0, nullptr, 0,
                           out16, ldo16, &ln, nullptr, nullptr, m, n, 0, nullptr, nullptr, 0, stream);
}

/* MLP tail on the fp16 path, second GEMM: out (fp32) = GELU(LayerNorm(hidden16)) w^T + bias + residual, LayerNorm + GELU applied
   while the fp16 hidden layer is staged (as pram_linear_x3_lngelu_f32). */
extern "C" int pram_linear_f16_lngelu_f32(const void* hidden16, int ldh, int k, const void* w16, const float* bias, const float* residual, int ldr,
                                          float* out, int ldo, int m, int n, const float* ln_ssq, int parts, const float* gamma,
                                          const float* beta, float eps, void* stream) {
    PRAM_REQUIRE(ln_ssq && out, "pram_linear_f16_lngelu_f32: null pointer");
    LnIo ln{nullptr, ln_ssq, parts, gamma, beta, eps};
    return linear_f16_impl("pram_linear_f16_lngelu_f32", reinterpret_cast<const float*>(hidden16), ldh, k, 1, nullptr, 0, 0, 0, w16, bias, residual, ldr, out, ldo,
                           nullptr, 0, &ln, nullptr, nullptr, m, n, 0, nullptr, nullptr, 0, stream);
}
