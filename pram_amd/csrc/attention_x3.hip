// Split-fp16 ("x3") flash attention on v_mfma_f32_32x32x16_f16: fp32-class results at the fp16 matrix rate.
// Replaces, like attention.hip, the materialised einsum -> softmax -> einsum of nets/segnetvit.py:73-76 (self) and
// nets/gml.py:175-179 (cross: both directions in one launch, kv_shift).
//
// Q, K and V arrive as the split planes the projection GEMM writes (pram_linear_x3_f32, out_hi / out_lo):
//     x * 16 = hi + lo,  hi = fp16(16 x),  lo = fp16(16 x - hi)
// and every product is three MFMAs accumulated in fp32 (gemm_core_x3.h has the error analysis):
//     S^T = K_hi Q_hi + K_hi Q_lo + K_lo Q_hi            (= 256 K Q^T; the 1/256 rides in the softmax scale)
//     O^T = V^T_lo P_hi + V^T_hi P_hi + V^T_hi P_lo       (P = 2^7 exp2(s - m) = P_hi + P_lo, = 2^11 V^T P)
// The probabilities are scaled by 2^7 (an offset in the exponent argument: free) and the running maximum m is LAZY: it follows the
// row maximum only when it is left behind by more than 2^8 (LAZY_T), so P <= 2^15 fits fp16, the two parts keep an absolute
// 2^-25 (fp16 subnormals are kept by the MFMA), and the rescale of the output accumulators — exactly 1 on almost every tile — is
// skipped by a wave-uniform branch.  PSPLIT = false (pram_attention_x3_set_p_split(0), from 1024 keys on) drops the P_lo
// product: P is then ONE fp16 whose rounding (2^-12 relative per probability) does not average out of flat attention — the
// output is a small difference of large terms — and shows as ~1e-3 on SegNetViT's logits against 4e-5; the row sum then
// accumulates the same rounded values, so scale and rounding bias cancel in the normalisation.
//
// Register design as attention.hip / attention_f16.hip: everything transposed so that the query row is the lane
// index in every accumulator (online-softmax state lane-local, P never leaves registers); K tile [key][d] and V tile
// TRANSPOSED and key-permuted [d][pos(key)] in LDS so that both MFMA operands are single ds_read_b128s.
// Per 64-key tile per wave: 48 MFMA x 32 cycles = 1536 matrix cycles (40 / 1280 without the P_lo product; the f32-MFMA kernel: 8192).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int D = 64, QW = 32, NW = 4, BQ = QW * NW, BKV = 64;
constexpr float LOG2E = 1.4426950408889634f;
// (the q / k / v planes carry value * pram_act_scale(), 16 by default: ArgsX::in_scale / scale2 get it at launch)
constexpr float LAZY_T = 8.0f;          // lazy running maximum: it follows the row maximum only when left behind by more than 2^8
#ifndef AX_XPRIO
#define AX_XPRIO 1
#endif
// tile-loop form per launch shape (see the kernel): 1 = vector / matrix phases, 0 = interleaved.  Defaults as measured
// (profiles/r05_attn_phases_ab.txt); the macros exist for that A/B.
#ifndef AX_STYLE4
#define AX_STYLE4 0      // 128-row workgroups, one chunk (grids that do not fill the chip)
#endif
#ifndef AX_STYLE8
#define AX_STYLE8 1      // 256-row workgroups
#endif
#ifndef AX_STYLEC
#define AX_STYLEC 1      // key chunks (fused and split): 1 = by the grid (phases from two workgroups per CU on), 2 = always phases, 0 = never
#endif
constexpr int SPREAD_MAXG = 7;          // interleaved form: leading MFMA groups of a tile's score phase that carry the row maximum
#ifndef AX_ABL
#define AX_ABL 0      // profiling only (results are garbage): 1 no soft-max, 2 no MFMA, 3 no fragment reads, 4 soft-max of the first tile only,
                      // 5 half of the K fragment reads (odd k-steps reuse the even ones' registers), 6 half of the K and V fragment reads
#endif

constexpr float P_EXP_SHIFT = 7.0f;     // probabilities carried as 2^7 p, at most 2^15 with the lazy maximum LAZY_T behind

struct ArgsX {
    const _Float16* qh; const _Float16* ql; const _Float16* kh; const _Float16* kl;
    const _Float16* vh; const _Float16* vl;   // V^T planes [batch][heads][64][tv] with the key permutation of pos_of_key (vt_kernel)
    float* out; float* lse2;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, tv, ldo;           // ldq / ldk in halves; tv = padded key count of a V^T row (multiple of 64)
    int batch, heads, m_max, n_max;
    float scale2;                    // scale * log2(e) / IN_SCALE^2
    int q_tiles;
    int kv_shift;                    // as in attention.hip
    float in_scale;                  // scale of the planes in the single-product mode (1: plain fp16 q / k / v)
    int nsplit;                      // 1: a workgroup walks all key chunks; > 1: blockIdx.y = group of group_tiles / 8 key chunks
    int group_tiles;                 // split mode: 64-key tiles per workgroup (a multiple of chunk_tiles)
    int chunk_tiles;                 // 64-key tiles per key chunk (g_chunk_tiles: even, the same for every launch of the process)
    float* part_o;                   // [nsplit][batch * m_max][heads * D]   normalised chunk outputs
    float* part_l;                   // [nsplit][batch][heads][m_max]        their log2-sum-exp
    _Float16* out16; int ldo16;      // single-product mode: the context as fp16 [batch * m_max][ldo16] instead of `out` (the next GEMM's operand)
};

// Key chunks (as attention.hip).  From 1024 keys on, the keys of a sequence are processed in chunks of chunk_tiles tiles: every
// chunk runs the online soft-max from a fresh state, is normalised, and all chunks are folded by fold_weights / fold_value — a
// fixed left fold in chunk order.  One workgroup normally walks all chunks of its 128 query rows ("fused").  A launch with too
// few (batch, head, q-tile) units to fill the chip — one or two query frames, the reference's online loop
// (localization/loc_by_rec_online.py:109-133) — makes groups of chunks a grid dimension instead ("split"): each workgroup parks
// its normalised chunk results in a caller-owned workspace and combine_x3_kernel applies the SAME fold in the SAME order, so the
// output does not depend on which mode ran, bit for bit: a padded batch element still equals its B = 1 run exactly.
// Chunk size.  A fused walk folds a running total at every chunk end; the total does not fit in the 256 registers of a wave
// beside the pipeline's two score tiles, so the chunked kernel spills: with two-part probabilities it is 5 % slower than the
// unchunked one before it folds anything, +17 % with 512-key chunks at 2048 keys, +6 % with two 2048-key chunks at 4096 keys
// (profiles/r03_x3_attention_chunks.txt).  The default is therefore 4096 keys — every shipped configuration (2048 and 4096
// keypoints) runs ONE chunk, the unchunked kernel, and pays nothing — and pram_attention_x3_set_chunk_keys lowers it for
// deployments that want the split mode for one-frame launches (bench.py --latency: 512; it moves the chunk boundaries of EVERY
// launch of the process, so results change in their last bits consistently, never between batch sizes).
constexpr int DEFAULT_CHUNK_TILES = 64;         // 4096 keys (g_chunk_tiles; pram_attention_x3_set_chunk_keys)
constexpr int SPLIT_TARGET = 256;               // split launches aim at this many workgroups: one per CU (g_split_target)

__device__ __forceinline__ void fold_weights(float lt, float lc, float* at, float* ac, float* lnew) {
    const float mx = fmaxf(lt, lc);
    const float wt = exp2f(lt - mx), wc = exp2f(lc - mx);
    const float den = wt + wc;
    *at = wt / den;
    *ac = wc / den;
    *lnew = mx + log2f(den);
}
__device__ __forceinline__ float fold_value(float ot, float oc, float at, float ac) { return ot * at + oc * ac; }

struct alignas(16) Smem {
    _Float16 kh[2][BKV * D];    // [key][d], slot-swizzled
    _Float16 kl[2][BKV * D];
    _Float16 vth[2][D * BKV];   // [d][pos(key)], slot-swizzled
    _Float16 vtl[2][D * BKV];
};  // 64 KiB -> two workgroups per CU

struct alignas(16) SmemHi {     // single-product mode: hi planes only
    _Float16 kh[2][BKV * D];
    _Float16 vth[2][D * BKV];
};
template <bool HI> struct SmemSel { typedef Smem type; };
template <> struct SmemSel<true> { typedef SmemHi type; };

__device__ __forceinline__ int key_of(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }
// position of key (0..63) inside a V^T row: inverse of key = 32t + (i&3) + 16u + 8(i>>2) + 4h
__device__ __forceinline__ int pos_of_key(int key) {
    const int t = key >> 5, w = key & 31;
    const int u = w >> 4, h = (w >> 2) & 1, i = (w & 3) | (((w >> 3) & 1) << 2);
    return t * 32 + u * 16 + h * 8 + i;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The kernel.  Per 64-key tile a wave issues 48 MFMAs (1 608 clocks of its SIMD's matrix pipe) and ~190 vector instructions
// (~1 000 clocks of vector issue: a VOP2 costs a wave 4 clocks, a VOP3 5, v_exp_f32 and v_fma_mix 8; beside one MFMA = 33.5 clocks a
// wave hides ~23 of them — profiles/r05_mfma_valu_overlap.txt).  Both pipes are nearly full, so the tile loop is built around
// WHERE the vector work sits (round 5; profiles/r05_x3_attention_valu_diet.txt has every step's A/B):
//   * software pipeline across tiles: the scores of tile j + 1 are multiplied while the soft-max of tile j runs (a second
//     32-register score tile; K staged one tile ahead of V: K_{j+2} and V_{j+1} land while K_{j+1} and V_j are read);
//   * the soft-max is split over BOTH matrix phases of a tile (softmax_a beside S_{j+1} = K_{j+1} Q^T, the fp16 parts and the row
//     sums inside pv_b beside O += V_j P_j), one MFMA : 3-4 vector instructions by sched_group_barrier — bunched behind the score
//     MFMAs alone it made that phase issue-bound (42 clocks per MFMA) and left the P V MFMAs bare: the younger wave of every SIMD
//     needed 3 760 clocks for the phase while the older one waited 1 900 at the barrier (profiles/r05_attn_phases.txt);
//   * no packed fp32 (v_pk_*: +18 clocks beside an MFMA, measured) — the file is built with -fno-slp-vectorize; the lo parts by
//     v_fma_mixlo / mixhi_f16 (one instruction per element instead of convert, subtract, convert); the half-waves trade row maxima
//     by v_permlane32_swap instead of ds_bpermute; the lazy maximum above.
// HI: single-product mode (BASELINE C5's fp16 path): only the hi planes exist (q / k / v rounded to fp16, scale p.in_scale = 1),
// one MFMA per product; half the LDS, so the co-residency is bounded by registers only.
// MODE 0: one online soft-max over all keys (one key chunk: every launch at the default chunk size); 1: fused, key chunks folded
// into a running total in registers / scratch; 2: split — blockIdx.y = group of key chunks, the chunk results go to the workspace,
// combine_x3_kernel folds — see "Key chunks" above.
// NWV: waves per workgroup.  Four (128 query rows, two workgroups per CU) everywhere but on full grids of MODE 0, where eight
// (256 rows, one workgroup per CU: the same 64 KB of LDS, the same registers per wave) stage every K / V tile once per 256 query
// rows instead of once per 128 — the staging is what the tile loop pays for beside its MFMAs (profiles/r02_x3_attention_ablation.txt).
#ifdef PRAM_PROFILING
// per-workgroup timeline of the last launch (profiles/tools/x3_attn_timeline.py): [blockIdx.x][0] start, [1] end (100 MHz wall
// clock), [2] XCC id << 16 | HW_ID bits, [3] shader clocks spent
static __device__ unsigned long long attn_prof[4096][4];
static __device__ unsigned long long attn_phase[8][4];      // per wave of the workgroup: shader clocks per tile phase, summed over workgroups
#endif

constexpr bool style_phases(int mode, int nwv) { return mode != 0 ? AX_STYLEC != 0 : nwv == NW ? AX_STYLE4 != 0 : AX_STYLE8 != 0; }

template <bool PSPLIT, bool HI = false, int MODE = 0, int NWV = NW, bool PHASES = style_phases(MODE, NWV)>
// (MODE 1, the fused chunk walk of launches without a workspace, keeps a second set of output accumulators: one workgroup per CU —
//  512 registers per lane — where two would spill 53-92 of them; the other 128-row forms run two workgroups per CU)
__global__ __launch_bounds__(NWV * 64, (NWV == NW && MODE != 1) ? 2 : 1) void attention_x3_pipe_kernel(ArgsX p) {
#ifdef PRAM_PROFILING
    const unsigned long long prof_t0 = wall_clock64(), prof_c0 = __builtin_readcyclecounter();
#endif
    constexpr int BQV = QW * NWV;            // query rows of the workgroup
    constexpr int SROWS = NWV * 8;           // K rows / V^T rows one staging pass of the workgroup covers (8 threads per row)
    constexpr int PPN = BKV / SROWS;         // staging passes per tile
    static_assert(!(HI && PSPLIT), "the single-product mode carries one plane of everything");
    __shared__ typename SmemSel<HI>::type s;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    float one = 1.0f;      // a 1.0 the optimiser cannot see through (quarter(): keeps fma(p, 1, -hi) an fma)
    asm volatile("" : "+s"(one));
    const int nblk = p.batch * p.heads * p.q_tiles;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int qt = id % p.q_tiles;
    const int bh = id / p.q_tiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qt * BQV >= qlen) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * BQV + wave * QW;
    const bool wave_active = q0 < qlen;
    const int qrow = q0 + r;
    const bool q_ok = qrow < qlen;
    if (klen <= 0) {   // empty key set: context defined as 0 (see attention.hip); the split mode's first chunk reports it
        if (MODE == 2 && blockIdx.y != 0) return;
        if (q_ok && HI && p.out16) {
            _Float16* o16 = p.out16 + ((size_t)b * p.m_max + qrow) * p.ldo16 + head * D;
#pragma unroll
            for (int c = 0; c < 32; ++c) o16[c * 2 + h] = (_Float16)0.f;
        } else if (q_ok) {
            float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<float4*>(op + c * 8 + h * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = 0.f;
        }
        return;
    }

    const size_t qoff = ((size_t)b * p.m_max + min(qrow, p.m_max - 1)) * p.ldq + head * D;
    const size_t koff = (size_t)kb * p.n_max * p.ldk + head * D;
    const size_t voff = ((size_t)kb * p.heads + head) * D * p.tv;

    half8 qh[4], ql[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qh[c] = *reinterpret_cast<const half8*>(p.qh + qoff + c * 16 + h * 8);
        if constexpr (!HI) ql[c] = *reinterpret_cast<const half8*>(p.ql + qoff + c * 16 + h * 8);
        if (!q_ok)
#pragma unroll
            for (int i = 0; i < 8; ++i) { qh[c][i] = (_Float16)0.f; if constexpr (!HI) ql[c][i] = (_Float16)0.f; }
    }

    const int lrow = tid >> 3, lseg = tid & 7;
    half8 krh[PPN], krl[PPN], vrh[PPN], vrl[PPN];
    // Staging addresses as a uniform base (scalar registers) plus a 32-bit byte offset per thread: one or two vector instructions per
    // tile instead of the 64-bit multiply-adds of a per-thread pointer (they sat in the vector phase, which bounds the tile loop).
    // min(row, klen - 1) * ldk == min(row * ldk, (klen - 1) * ldk): the clamp of the last tile costs one v_min.
    const char* const kh_b = reinterpret_cast<const char*>(p.kh + koff);
    const char* const kl_b = reinterpret_cast<const char*>(HI ? p.kh : p.kl + koff);
    const char* const vh_b = reinterpret_cast<const char*>(p.vh + voff);
    const char* const vl_b = reinterpret_cast<const char*>(HI ? p.vh : p.vl + voff);
    unsigned kofs[PPN], vofs[PPN];
#pragma unroll
    for (int pp = 0; pp < PPN; ++pp) {
        kofs[pp] = ((unsigned)(lrow + SROWS * pp) * (unsigned)p.ldk + lseg * 8) * 2u;
        vofs[pp] = ((unsigned)(lrow + SROWS * pp) * (unsigned)p.tv + lseg * 8) * 2u;
    }
    const unsigned klast = ((unsigned)(klen - 1) * (unsigned)p.ldk + lseg * 8) * 2u;
    const unsigned ktile_bytes = (unsigned)BKV * (unsigned)p.ldk * 2u;
    auto gload_k = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < PPN; ++pp) {
            const unsigned o = min(kofs[pp] + (unsigned)kt * ktile_bytes, klast);
            krh[pp] = *reinterpret_cast<const half8*>(kh_b + o);
            if constexpr (!HI) krl[pp] = *reinterpret_cast<const half8*>(kl_b + o);
        }
    };
    auto gload_v = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < PPN; ++pp) {
            const unsigned o = vofs[pp] + (unsigned)kt * (BKV * 2u);
            vrh[pp] = *reinterpret_cast<const half8*>(vh_b + o);
            if constexpr (!HI) vrl[pp] = *reinterpret_cast<const half8*>(vl_b + o);
        }
    };
    auto lstore_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < PPN; ++pp) {
            const int row = lrow + SROWS * pp;
            const int off = row * D + ((lseg ^ ((row >> 1) & 7)) << 3);
            *reinterpret_cast<half8*>(&s.kh[buf][off]) = krh[pp];
            if constexpr (!HI) *reinterpret_cast<half8*>(&s.kl[buf][off]) = krl[pp];
        }
    };
    auto lstore_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < PPN; ++pp) {
            const int row = lrow + SROWS * pp;
            const int off = row * D + ((lseg ^ ((row >> 1) & 7)) << 3);
            *reinterpret_cast<half8*>(&s.vth[buf][off]) = vrh[pp];
            if constexpr (!HI) *reinterpret_cast<half8*>(&s.vtl[buf][off]) = vrl[pp];
        }
    };

    struct KFrag { half8 h0, h1, l0, l1; };
    auto kload = [&](int buf, int c, KFrag& f) __attribute__((always_inline)) {
#if AX_ABL == 5 || AX_ABL == 6      // odd k-steps reuse the fragments of the even ones: half of the K fragment reads, real operands
        if (c & 1) return;
#endif
#if AX_ABL == 3
        asm volatile("" : "=v"(f.h0), "=v"(f.h1), "=v"(f.l0), "=v"(f.l1));
        return;
#endif
        const int slot = ((2 * c + h) ^ ((r >> 1) & 7)) << 3;
        f.h0 = *reinterpret_cast<const half8*>(&s.kh[buf][r * D + slot]);
        f.h1 = *reinterpret_cast<const half8*>(&s.kh[buf][(32 + r) * D + slot]);
        if constexpr (!HI) {
            f.l0 = *reinterpret_cast<const half8*>(&s.kl[buf][r * D + slot]);
            f.l1 = *reinterpret_cast<const half8*>(&s.kl[buf][(32 + r) * D + slot]);
        }
    };
    auto kmma = [&](f32x16 (&st)[2], int c, const KFrag& f) __attribute__((always_inline)) {
#if AX_ABL == 2
        asm volatile("" :: "v"(f.h0), "v"(f.h1), "v"(f.l0), "v"(f.l1));
        return;
#endif
        if constexpr (!HI) {
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l0, qh[c], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l1, qh[c], st[1], 0, 0, 0);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, ql[c], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, ql[c], st[1], 0, 0, 0);
        }
        st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, qh[c], st[0], 0, 0, 0);
        st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, qh[c], st[1], 0, 0, 0);
    };
    struct VFrag { half8 h0, h1, l0, l1; };
    auto vload = [&](int buf, int t, int u, VFrag& f) __attribute__((always_inline)) {
#if AX_ABL == 6      // half of the V fragment reads (and ABL 5's half of the K reads): what a 64-row wave would save
        if (u & 1) return;
#endif
#if AX_ABL == 3
        asm volatile("" : "=v"(f.h0), "=v"(f.h1), "=v"(f.l0), "=v"(f.l1));
        return;
#endif
        const int slot = ((t * 4 + u * 2 + h) ^ ((r >> 1) & 7)) << 3;
        f.h0 = *reinterpret_cast<const half8*>(&s.vth[buf][r * BKV + slot]);
        f.h1 = *reinterpret_cast<const half8*>(&s.vth[buf][(32 + r) * BKV + slot]);
        if constexpr (!HI) {
            f.l0 = *reinterpret_cast<const half8*>(&s.vtl[buf][r * BKV + slot]);
            f.l1 = *reinterpret_cast<const half8*>(&s.vtl[buf][(32 + r) * BKV + slot]);
        }
    };

    const int nkt_all = (klen + BKV - 1) / BKV;
    const int CT = p.chunk_tiles;
    const int t0 = MODE == 2 ? (int)blockIdx.y * p.group_tiles : 0;               // first tile of this workgroup (a multiple of 8)
    const int nkt = MODE == 2 ? min(nkt_all, t0 + p.group_tiles) : nkt_all;       // one past its last tile
    if (t0 >= nkt) return;                                                         // this chunk lies beyond klen
    float m_run = -1.0e30f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
    // Chunk bookkeeping.  MODE 1 (fused): a running total (o_tot, l_tot2) is folded at every chunk end by the left fold
    // combine_x3_kernel applies; it does not fit beside the two score tiles of the software pipeline (256 registers at two waves
    // per SIMD), the compiler spills and reloads it around the chunk end: +6 % for two 2048-key chunks, +17 % at 512.  MODE 2 (split):
    // every chunk of the workgroup's key group is PARKED in the workspace, [chunk][row][head * 64 + d] / [chunk][batch][head][row].
    // MODE 0: one chunk, nothing to fold.
    // (A fused variant that parked its chunks like MODE 2 and folded them after the last tile — no spill — was dropped: compiled
    // with two-part probabilities it returned, on ~1 wave in 2500, one output register with the earlier chunks' share missing in
    // lanes 48..63; inputs in memory were right, draining every counter before the fold did not help, the cause was not found.
    // tests/test_gpu_guard_chunks_mlp.py::test_x3_attention_many_workgroups_every_mode repeats the launch that showed it.)
    float l_tot2 = -INFINITY;
    float o_tot[2][MODE == 1 ? 16 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { o_tot[0][e] = 0.f; o_tot[1][e] = 0.f; }
    }
    // The workspace addresses are recomputed where they are used, behind an opaque zero: hoisted out of the tile loop they would be
    // four more registers live across it — spilled, and reloaded with the round trip exposed at every chunk end.
    auto part_ptr = [&](int c) __attribute__((always_inline)) -> float* {
        int z = 0;
        asm volatile("" : "+v"(z));
        const size_t row = (size_t)b * p.m_max + min(qt * BQV + wave * QW + (int)(threadIdx.x & 31) + z, p.m_max - 1);
        return p.part_o + ((size_t)c * p.batch * p.m_max + row) * (p.heads * D) + head * D;
    };
    auto part_lse = [&](int c) __attribute__((always_inline)) -> float* {
        int z = 0;
        asm volatile("" : "+v"(z));
        const size_t li = ((size_t)b * p.heads + head) * p.m_max + min(qt * BQV + wave * QW + (int)(threadIdx.x & 31) + z, p.m_max - 1);
        return p.part_l + (size_t)c * p.batch * p.heads * p.m_max + li;
    };
    // end of a key chunk inside the walk (MODE 1 / 2): normalise it, fold it (1) or park it (2), start afresh
    auto chunk_end = [&](int c) __attribute__((always_inline)) {
        const float l_c = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = (1.0f / p.in_scale) / l_c;      // undoes the 2^P_EXP_SHIFT of P (carried by l_c) and the scale of V
        const float lse_c = m_run + (log2f(l_c) - P_EXP_SHIFT);
        if constexpr (MODE == 2) {
            if (q_ok) {
                float* op = part_ptr(c);
#pragma unroll
                for (int dn = 0; dn < 2; ++dn)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(op + dn * 32 + 8 * g + 4 * h) =
                            make_float4(oacc[dn][4 * g + 0] * inv, oacc[dn][4 * g + 1] * inv, oacc[dn][4 * g + 2] * inv, oacc[dn][4 * g + 3] * inv);
                if (h == 0) *part_lse(c) = lse_c;
            }
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
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
        m_run = -1.0e30f;
        l_run = 0.f;
    };
    // after the last tile: the last chunk normalised in place (oacc), l_tot2 = its log2-sum-exp; MODE 1 then folds it into the total
    auto finish = [&]() __attribute__((always_inline)) {
        const float l_c = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = (1.0f / p.in_scale) / l_c;
        const float lse_c = m_run + (log2f(l_c) - P_EXP_SHIFT);
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[0][e] *= inv; oacc[1][e] *= inv; }
        if constexpr (MODE == 1) {
            float at, ac, lnew;
            fold_weights(l_tot2, lse_c, &at, &ac, &lnew);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                oacc[0][e] = fold_value(o_tot[0][e], oacc[0][e], at, ac);
                oacc[1][e] = fold_value(o_tot[1][e], oacc[1][e], at, ac);
            }
            l_tot2 = lnew;
        } else {
            l_tot2 = lse_c;
        }
    };
    // O *= alpha, between the two matrix phases of a tile; skipped when no lane's running maximum moved (alpha is exactly 1 on
    // almost every tile: the maximum is lazy, see softmax_a)
    auto rescale = [&](float alpha) __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) == 0) return;
#pragma unroll
        for (int dn = 0; dn < 2; ++dn)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dn][e] *= alpha;
    };
    // The vector ALU is the second bound of this kernel (profiles/r05_mfma_valu_overlap.txt: beside one v_mfma_f32_32x32x16_f16 =
    // 33.5 clocks a wave hides ~23 clocks of vector issue — a VOP2 costs 4, a VOP3 5, v_exp_f32 and v_fma_mix 8 — and the two waves
    // of a SIMD share the issue port).  One tile's soft-max is ~1 000 such clocks; bunched behind the 24 score MFMAs (42 per MFMA)
    // it made that phase issue-bound and left the 24 P V MFMAs bare (profiles/r05_attn_phases.txt: the younger wave of every SIMD
    // needed 3 760 clocks for the phase, the older one waited 1 900 at the barrier).  Here it is split where the data allows:
    //   part A (beside S_{j+1} = K_{j+1} Q^T): row maximum, p = 2^(s c + shift) left in place of the scores       ~22 clocks per MFMA
    //   part B (beside O += V_j P_j): per 16-key quarter, right in front of the MFMAs that take them: fp16 hi parts
    //          (v_cvt_pk), lo parts (v_fma_mix), row sums                                                          ~19 clocks per MFMA
    auto softmax_a = [&](f32x16 (&st)[2]) __attribute__((always_inline)) -> float {
        float tmax = st[0][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, st[0][e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[1][e]);
        {
            float a = tmax, b = tmax;
            asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            tmax = fmaxf(a, b);
        }
        const float t2 = tmax * p.scale2;
        const float m_new = (t2 > m_run + LAZY_T) ? t2 : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float shift = P_EXP_SHIFT - m_new;
        const float sc = p.scale2;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[t][e] = __builtin_amdgcn_exp2f(fmaf(st[t][e], sc, shift));      // argument <= P_EXP_SHIFT + LAZY_T
        m_run = m_new;
        return alpha;
    };
    // the fp16 hi parts of one 16-key quarter (four v_cvt_pk_f16_f32; the single-product mode sums the rounded values here)
    // (kept as four separate packed pairs: taking the fp16 halves back out of an int4 / half8 vector for v_fma_mix made this
    // compiler read every pair's hi part from element 0 — profiles/r05_x3_attention_valu_diet.txt)
    struct Hi4 { half2_t v[4]; };
    auto hi_parts = [&](int t, int u, f32x16 (&st)[2], float& psum) __attribute__((always_inline)) -> Hi4 {
        Hi4 hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi.v[j] = __builtin_convertvector((f2){st[t][8 * u + 2 * j], st[t][8 * u + 2 * j + 1]}, half2_t);      // v_cvt_pk_f16_f32
            if constexpr (!PSPLIT) psum = __builtin_amdgcn_fdot2(hi.v[j], (half2_t){(_Float16)1.f, (_Float16)1.f}, psum, false);
        }
        return hi;
    };
    auto pack4 = [&](const Hi4& x) __attribute__((always_inline)) -> half8 {
        return (half8){x.v[0][0], x.v[0][1], x.v[1][0], x.v[1][1], x.v[2][0], x.v[2][1], x.v[3][0], x.v[3][1]};
    };
#ifdef PRAM_PROFILING
    // shader clocks per phase of a tile, summed over the tile loop.  Phases form: [0] V phases, [1] the barrier behind them, [2] X phases,
    // [3] the barrier behind them.  Interleaved form: [0] scores of tile j+1 beside the soft-max of tile j, [1] P V, [2] the staged tiles
    // written to LDS (waits for the loads), [3] barrier
    unsigned long long prof_ts[5] = {0, 0, 0, 0, 0}, prof_ph[4] = {0, 0, 0, 0};
#define PROF_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); prof_ts[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROF_STAMP(i) do { } while (0)
#endif
    // ---- The tile loop in PHASES (round 5, second step; the form of every launch that puts two waves on a SIMD).  A SIMD serves
    // its two waves with STRICT priority for the older one on each pipe, but an MFMA-only wave and a vector-only wave overlap almost
    // perfectly whichever is older (profiles/r05_mfma_valu_two_waves.txt: 33.0 clocks per MFMA beside 8.2 per vector instruction,
    // against 32.0 / 6.9 alone; two MFMA streams: the younger gets NOTHING until the older is through).  A tile has 48 MFMAs and
    // ~190 vector instructions per wave — four per MFMA, exactly what that overlap carries.  So a tile is cut into
    //   V_j : soft-max of S_j -> P_j (hi, lo parts, row sums; the rare rescale of O), the staged K / V tiles written to LDS, the
    //         next ones requested                                                                            (no MFMA)
    //   X_j : O += V_j P_j and S_{j+1} = K_{j+1} Q^T                                  (48 MFMAs = 1 608 clocks, no vector work)
    // with a barrier behind each, and the younger half of an eight-wave workgroup (waves 4-7, the SIMD partners of waves 0-3) runs
    // the SAME sequence one phase later: while one wave of a SIMD multiplies, its partner does its vector work.
    //   phase        ... 2j            2j+1          2j+2 ...
    //   waves 0-3        V_j           X_j           V_{j+1}
    //   waves 4-7        X_{j-1}       V_j           X_j
    // The wave in X raises its priority (s_setprio 1): left to age order the older wave won BOTH its phases and waited ~1 300 clocks
    // per tile at the barriers for its partner (no gain over the interleaved form); with it the two take equal time (-4 %).
    // Measured (profiles/r05_attn_phases_ab.txt, 32 x 2048 keys): interleaved 391 us, phases 391, + priority 376, + 32-bit staging
    // offsets 364; X alone (soft-max of the first tile only) 266 us, V alone (no MFMA) 108 us: about 40 % of the vector work is
    // hidden, and what is left scales with the vector ISSUE CLOCKS of a tile whichever wave carries them — moving a quarter or half
    // of the conversions into the head of X, a second K fragment slot, priority for the younger half only: no gain, not kept.
    // LDS: K_t is read in phases 2t-1 (older half) and 2t (younger), V_t in 2t+1 and 2t+2; every thread stages its 16-byte share of
    // each plane; the older half writes K_{j+1} / V_j in its V_j (phase 2j), the younger half K_{j+2} / V_{j+1} in its V_j (phase
    // 2j+1): every tile is complete a barrier before its first reader, and its stage (two per operand, as before) was last read two
    // phases before its first writer.  Four-wave workgroups (two per CU: the key-chunk kernels) run the older half's sequence; their
    // SIMD partners belong to another workgroup and drift into the complementary phase on their own (-6 .. -10 % against the
    // interleaved form at 8192 / 16384 keys).  A grid that leaves ONE wave per SIMD has no partner to overlap with and keeps the
    // interleaved form below (phases there: +17 % at one query of 2048 keys).
    // Same MFMA order per accumulator in both forms: bit-identical results.
    if constexpr (PHASES) {
        constexpr bool TWO_HALVES = NWV == 2 * NW;
        const bool upper = TWO_HALVES && wave >= NW;
        Hi4 phi[2][2], plo[2][2];
        // V_j on score tile st
        auto stage = [&](int j, auto dly_t) __attribute__((always_inline)) {      // the staging part of V_j: the data were requested two phases ago
            constexpr int dly = decltype(dly_t)::value;
            const int tv = j + dly, tk = j + 1 + dly;
            if (tv < nkt) lstore_v(tv & 1);
            if (tk < nkt) lstore_k(tk & 1);
            if (tv + 1 < nkt) gload_v(tv + 1);
            if (tk + 1 < nkt) gload_k(tk + 1);
        };
        // P of one 16-key quarter as fp16 parts (phi, plo) and its share of the row sums
        float ps4c[4], psumc, alphac;
        auto convert = [&](int t, int u, f32x16 (&st)[2]) __attribute__((always_inline)) {
            phi[t][u] = hi_parts(t, u, st, psumc);
            if constexpr (PSPLIT) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float p0 = st[t][8 * u + 2 * jj], p1 = st[t][8 * u + 2 * jj + 1];
                    // lo = fp16(p - hi) as ONE instruction per element: fma(p, 1, -hi) with the 1 hidden from the optimiser
                    // selects v_fma_mixlo / mixhi_f16 (hi read as fp16 from its half of the packed register); a plain
                    // p - (float)hi would be convert, subtract, convert.  Same value bit for bit (the difference is exact).
                    const half2_t hk = phi[t][u].v[jj];
                    plo[t][u].v[jj] = (half2_t){(_Float16)__builtin_fmaf(p0, one, -(float)hk[0]), (_Float16)__builtin_fmaf(p1, one, -(float)hk[1])};
                    ps4c[jj] = (t == 0 && u == 0) ? p0 + p1 : ps4c[jj] + (p0 + p1);
                }
            }
        };
        auto sums_done = [&]() __attribute__((always_inline)) {
            if constexpr (PSPLIT) psumc = (ps4c[0] + ps4c[1]) + (ps4c[2] + ps4c[3]);
            l_run = fmaf(l_run, alphac, psumc);
        };
        auto phase_v = [&](int j, f32x16 (&st)[2], auto last_t, auto dly_t) __attribute__((always_inline)) {
            stage(j, dly_t);
            if (decltype(last_t)::value && (klen & (BKV - 1)) && nkt == nkt_all) {      // last tile of the sequence: keys beyond klen
                const int kbase = j * BKV;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + t * 32 + key_of(e, h) >= klen) st[t][e] = -INFINITY;
            }
#if AX_ABL == 1
            alphac = 1.f; psumc = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) asm volatile("" : "=v"(phi[q >> 1][q & 1].v[jj]), "=v"(plo[q >> 1][q & 1].v[jj]));
            return;
#endif
#if AX_ABL == 4      // the vector work of the first tile only: real P in every product, no vector phase afterwards
            if (j != t0) return;
#endif
            alphac = softmax_a(st);
            rescale(alphac);
            psumc = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) convert(q >> 1, q & 1, st);
            sums_done();
            // every lo part is computed HERE: left alone the compiler sinks half of the v_fma_mix into the head of the matrix phase (-0.5 .. -1 %)
            if constexpr (PSPLIT) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        half2_t& lo_ref = plo[q >> 1][q & 1].v[jj];
                        asm volatile("" : "+v"(lo_ref));
                    }
            }
        };
        // one 16-key quarter of P V: the four products that take P_hi, then (PSPLIT) the two that take P_lo
        auto vq = [&](int t, int u, const VFrag& f) __attribute__((always_inline)) {
#if AX_ABL == 2
            asm volatile("" :: "v"(f.h0), "v"(f.h1), "v"(f.l0), "v"(f.l1));
            return;
#endif
            if constexpr (!HI) {
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l0, pack4(phi[t][u]), oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l1, pack4(phi[t][u]), oacc[1], 0, 0, 0);
            }
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, pack4(phi[t][u]), oacc[0], 0, 0, 0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, pack4(phi[t][u]), oacc[1], 0, 0, 0);
            if constexpr (PSPLIT) {
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, pack4(plo[t][u]), oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, pack4(plo[t][u]), oacc[1], 0, 0, 0);
            }
        };
        // X_j: O += V_j P_j (P_j from V_j) and, when a tile follows, S_{j+1} into sn; fragment reads one step ahead of their MFMAs
        auto phase_x = [&](int j, f32x16 (&sn)[2], auto do_s_t) __attribute__((always_inline)) {
            constexpr bool do_s = decltype(do_s_t)::value;
            const int vbuf = j & 1, kbuf = (j + 1) & 1;
            // two fragment slots in all (32 registers): each is refilled right behind the six MFMAs that consumed it, one group of six
            // (~200 clocks) ahead of the group that needs it
            VFrag va;
            KFrag fa;
            // the wave in its matrix phase goes first: its partner's vector work fills the gaps
            if (AX_XPRIO) __builtin_amdgcn_s_setprio(1);
            vload(vbuf, 0, 0, va);
            if (do_s) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { sn[0][e] = 0.f; sn[1][e] = 0.f; }
                kload(kbuf, 0, fa);
                __builtin_amdgcn_sched_barrier(0);
                vq(0, 0, va);
                vload(vbuf, 0, 1, va);
                __builtin_amdgcn_sched_barrier(0);
                kmma(sn, 0, fa);
                kload(kbuf, 1, fa);
                __builtin_amdgcn_sched_barrier(0);
                vq(0, 1, va);
                vload(vbuf, 1, 0, va);
                __builtin_amdgcn_sched_barrier(0);
                kmma(sn, 1, fa);
                kload(kbuf, 2, fa);
                __builtin_amdgcn_sched_barrier(0);
                vq(1, 0, va);
                vload(vbuf, 1, 1, va);
                __builtin_amdgcn_sched_barrier(0);
                kmma(sn, 2, fa);
                kload(kbuf, 3, fa);
                __builtin_amdgcn_sched_barrier(0);
                vq(1, 1, va);
                kmma(sn, 3, fa);
            } else {
                __builtin_amdgcn_sched_barrier(0);
                vq(0, 0, va);
                vload(vbuf, 0, 1, va);
                __builtin_amdgcn_sched_barrier(0);
                vq(0, 1, va);
                vload(vbuf, 1, 0, va);
                __builtin_amdgcn_sched_barrier(0);
                vq(1, 0, va);
                vload(vbuf, 1, 1, va);
                __builtin_amdgcn_sched_barrier(0);
                vq(1, 1, va);
            }
            if (AX_XPRIO) __builtin_amdgcn_s_setprio(0);
        };
        // one tile: V_j, barrier, X_j, barrier (the younger half's very last X needs none: nobody waits for it)
        auto tile = [&](int j, f32x16 (&sc)[2], f32x16 (&sn)[2], auto do_s_t, auto dly_t) __attribute__((always_inline)) {
            PROF_STAMP(0);
            phase_v(j, sc, std::integral_constant<bool, !decltype(do_s_t)::value>{}, dly_t);
            PROF_STAMP(1);
            __syncthreads();
            PROF_STAMP(2);
            phase_x(j, sn, do_s_t);
            PROF_STAMP(3);
            if (!(decltype(dly_t)::value == 1 && !decltype(do_s_t)::value)) __syncthreads();
            PROF_STAMP(4);
#ifdef PRAM_PROFILING
            prof_ph[0] += prof_ts[1] - prof_ts[0]; prof_ph[1] += prof_ts[2] - prof_ts[1];
            prof_ph[2] += prof_ts[3] - prof_ts[2]; prof_ph[3] += prof_ts[4] - prof_ts[3];
#endif
        };

        // ---- prologue (tile numbers are absolute: t0 is even, the stage of a tile is its parity): K_{t0} staged by everybody; the
        // younger half then spends its idle first phase putting its share of K_{t0+1} / V_{t0} into LDS and requesting K_{t0+2} / V_{t0+1};
        // the older half requests K_{t0+1} / V_{t0} (written in its V_{t0}) and multiplies S_{t0}
        gload_k(t0);
        lstore_k(0);
        __syncthreads();
        f32x16 sa[2], sb[2];
        if (upper) {
            if (t0 + 1 < nkt) gload_k(t0 + 1);
            gload_v(t0);
            if (t0 + 1 < nkt) lstore_k(1);
            lstore_v(0);
            if (t0 + 2 < nkt) gload_k(t0 + 2);
            if (t0 + 1 < nkt) gload_v(t0 + 1);
            __syncthreads();
        } else {
            if (t0 + 1 < nkt) gload_k(t0 + 1);
            gload_v(t0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) { sa[0][e] = 0.f; sa[1][e] = 0.f; }
        if (wave_active) {      // "X_{t0 - 1}": only the scores of the first tile
            KFrag fa, fb;
            kload(0, 0, fa);
            kload(0, 1, fb);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 0, fa);
            kload(0, 2, fa);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 1, fb);
            kload(0, 3, fb);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 2, fa);
            kmma(sa, 3, fb);
        }
        __syncthreads();

        // the two halves of an eight-wave workgroup run the same sequence with a different staging offset: instantiated per half, so that
        // every LDS stage index is the tile's parity plus a constant
        auto run = [&](auto dly_t) __attribute__((always_inline)) -> bool {
            if (!wave_active) {
                // a wave whose 32 query rows lie beyond the sequence: it still stages its share of every tile and meets every barrier
                for (int j = t0; j < nkt; ++j) {
                    stage(j, dly_t);
                    __syncthreads();
                    if (!(decltype(dly_t)::value == 1 && j == nkt - 1)) __syncthreads();
                }
                return false;
            }
            int j = t0;
            const std::true_type more_t{};
            const std::false_type last_t{};
            if constexpr (MODE != 0) {
                // Whole key chunks that are followed by at least one more tile: an even number of tiles (the score-tile ping-pong comes back
                // to sa), then the chunk's end in straight-line code BETWEEN the tile loops — written as a branch inside the tile loop it
                // made every accumulator a loop-carried phi of two definitions and cost ~100 register copies per tile.
                while (nkt - j > CT) {
        #pragma unroll 1
                    for (int i = 0; i < CT; i += 2) {
                        tile(j + i, sa, sb, more_t, dly_t);
                        tile(j + i + 1, sb, sa, more_t, dly_t);
                    }
                    j += CT;
                    chunk_end(j / CT - 1);
                }
            }
            for (; j + 2 < nkt; j += 2) {
                tile(j, sa, sb, more_t, dly_t);
                tile(j + 1, sb, sa, more_t, dly_t);
            }
            if (j + 2 == nkt) {
                tile(j, sa, sb, more_t, dly_t);
                tile(j + 1, sb, sa, last_t, dly_t);
            } else {
                tile(j, sa, sb, last_t, dly_t);
            }
            return true;
        };
        const bool live = upper ? run(std::integral_constant<int, 1>{}) : run(std::integral_constant<int, 0>{});
        if (!live) return;
    } else {
        // ---- The INTERLEAVED tile loop (round 5, first step): every wave runs S_{j+1} beside part A of the soft-max of tile j and
        // O += V_j P_j beside part B, one barrier per tile.  It hides the vector work in the wave's OWN MFMA shadow, so it does not
        // need a partner wave in the complementary phase: the form for grids that leave one wave per SIMD (a query or two).
        // part B + P V of one tile, written in the order it is meant to issue (the sched_group_barriers behind the call pin it): per
        // 16-key quarter the four products that take P_hi, each followed by one element pair's lo parts (v_fma_mixlo / mixhi) and its
        // sum; then the two products that take P_lo, followed by the row-sum updates and the NEXT quarter's hi parts.
        // va / vb: the V^T fragments of the first two quarters, already requested.
        auto pv_b = [&](int vbuf, f32x16 (&st)[2], float alpha, VFrag& va, VFrag& vb) {
            float ps4[4] = {0.f, 0.f, 0.f, 0.f}, psum = 0.f;
            Hi4 hi = hi_parts(0, 0, st, psum);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int t = qd >> 1, u = qd & 1;
                VFrag& f = u ? vb : va;
                Hi4 lo;
                float pair[4];
                auto lo_pair = [&](int j) __attribute__((always_inline)) {
                    if constexpr (PSPLIT) {
                        const float p0 = st[t][8 * u + 2 * j], p1 = st[t][8 * u + 2 * j + 1];
                        // lo = fp16(p - hi) as ONE instruction per element: fma(p, 1, -hi) with the 1 hidden from the optimiser selects
                        // v_fma_mixlo / mixhi_f16 (hi read as fp16 from its half of the packed register); a plain p - (float)hi would be
                        // convert, subtract, convert.  Same value bit for bit (the difference is exact in fp32).
                        const half2_t hk = hi.v[j];
                        lo.v[j] = (half2_t){(_Float16)__builtin_fmaf(p0, one, -(float)hk[0]), (_Float16)__builtin_fmaf(p1, one, -(float)hk[1])};
                        pair[j] = p0 + p1;
                    }
                };
                if constexpr (!HI) {
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l0, pack4(hi), oacc[0], 0, 0, 0);
                    lo_pair(0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l1, pack4(hi), oacc[1], 0, 0, 0);
                    lo_pair(1);
                }
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, pack4(hi), oacc[0], 0, 0, 0);
                lo_pair(2);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, pack4(hi), oacc[1], 0, 0, 0);
                lo_pair(3);
                Hi4 nhi = hi;
                if (qd < 3) nhi = hi_parts((qd + 1) >> 1, (qd + 1) & 1, st, psum);
                if constexpr (PSPLIT) {
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h0, pack4(lo), oacc[0], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ps4[j] += pair[j];
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h1, pack4(lo), oacc[1], 0, 0, 0);
                }
                hi = nhi;
                if (qd == 0) vload(vbuf, 1, 0, va);       // this quarter is through with va / vb: the fragments of quarter qd + 2
                if (qd == 1) vload(vbuf, 1, 1, vb);
            }
            if constexpr (PSPLIT) psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
            l_run = fmaf(l_run, alpha, psum);
        };

        // ---- prologue: K_0 | V_0, K_1 staged; S_0 multiplied on its own  (tile numbers relative to t0, which is even: the
        // stage parity of a tile is that of its absolute number)
        gload_k(t0);
        lstore_k(0);
        gload_v(t0);
        if (t0 + 1 < nkt) gload_k(t0 + 1);
        lstore_v(0);
        if (t0 + 1 < nkt) lstore_k(1);
        __syncthreads();
        f32x16 sa[2], sb[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { sa[0][e] = 0.f; sa[1][e] = 0.f; }
        if (wave_active) {
            KFrag fa, fb;
            kload(0, 0, fa);
            kload(0, 1, fb);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 0, fa);
            kload(0, 2, fa);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 1, fb);
            kload(0, 3, fb);
            __builtin_amdgcn_sched_barrier(0);
            kmma(sa, 2, fa);
            kmma(sa, 3, fb);
        }
        // tile 0's K stage is rewritten (K_2) at the END of the first loop pass, before that pass's barrier: every wave must be done
        // reading K_0 first
        if (t0 + 2 < nkt) __syncthreads();

        // tile j (not the last): S_{j+1} = K_{j+1} Q^T interleaved with softmax(S_j), then O += V_j P_j; K_{j+2} and V_{j+1} staged
        auto mid = [&](int j, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
            const int kbuf = (j + 1) & 1, vbuf = j & 1;
            const bool more_k = j + 2 < nkt;
            PROF_STAMP(0);
            gload_v(j + 1);
            if (more_k) gload_k(j + 2);
            if (wave_active) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { sn[0][e] = 0.f; sn[1][e] = 0.f; }
                KFrag fa, fb;
                VFrag va, vb;
                kload(kbuf, 0, fa);
                kload(kbuf, 1, fb);
                __builtin_amdgcn_sched_barrier(0);
                kmma(sn, 0, fa);
                kload(kbuf, 2, fa);
                kmma(sn, 1, fb);
                kload(kbuf, 3, fb);
                kmma(sn, 2, fa);
                kmma(sn, 3, fb);
                vload(vbuf, 0, 0, va);           // the first two quarters' V^T fragments travel under the end of phase A
                vload(vbuf, 0, 1, vb);
                const float alpha = softmax_a(sc);
                // (the probabilities are only used behind the rescale branch: without a use in THIS block the compiler sinks the 32
                // v_fma / v_exp pairs below it, out of the MFMAs' shadow)
                asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
                // phase A: one score MFMA, then vector work of part A in its shadow: the row maximum first (everything else waits for
                // it), then v_fma + v_exp pairs; the fragment reads go out with the first groups (K of the later k-steps, then V^T of
                // the first quarters).  0x008 MFMA, 0x100 LDS read, 0x002 vector ALU, 0x400 transcendental.
#pragma unroll
                for (int g = 0; g < (HI ? 8 : 24); ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (g < (HI ? 8 : 16)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (g < (HI ? 2 : SPREAD_MAXG)) __builtin_amdgcn_sched_group_barrier(0x002, HI ? 16 : 4, 0);
                    else {
                        __builtin_amdgcn_sched_group_barrier(0x002, HI ? 6 : 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, HI ? 6 : 2, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                PROF_STAMP(1);
                rescale(alpha);
                __builtin_amdgcn_sched_barrier(0);
                pv_b(vbuf, sc, alpha, va, vb);
                // phase B as pv_b writes it: 4 v_cvt_pk in front, then per quarter 4 x { MFMA, v_fma_mixlo, v_fma_mixhi, v_add },
                // { MFMA, the next quarter's 4 v_cvt_pk }, { MFMA, 4 v_add }; the two reloads of the fragment registers behind them
                __builtin_amdgcn_sched_group_barrier(0x002, HI ? 8 : 4, 0);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    if constexpr (HI) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (qd < 3) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (qd < 3) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, PSPLIT ? 3 : 1, 0);
                        }
                        if constexpr (PSPLIT) {
                            if (qd < 3) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    }
                    if (qd < 2) __builtin_amdgcn_sched_group_barrier(0x100, HI ? 2 : 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                PROF_STAMP(2);
            }
            lstore_v((j + 1) & 1);
            if (more_k) lstore_k(j & 1);
            PROF_STAMP(3);
            __syncthreads();
            PROF_STAMP(4);
#ifdef PRAM_PROFILING
            prof_ph[0] += prof_ts[1] - prof_ts[0]; prof_ph[1] += prof_ts[2] - prof_ts[1];
            prof_ph[2] += prof_ts[3] - prof_ts[2]; prof_ph[3] += prof_ts[4] - prof_ts[3];
#endif
        };
        int j = t0;
        if constexpr (MODE != 0) {
            // Whole key chunks that are followed by at least one more tile: eight tiles (the score-tile ping-pong comes back to
            // sa), then the chunk's end in straight-line code BETWEEN the tile loops — written as a branch inside the tile loop it
            // made every accumulator a loop-carried phi of two definitions and cost ~100 register copies per tile (+13 % kernel
            // time).  Its parking stores go out after the tile's staged loads were waited for; the next load wait is a tile away.
            while (nkt - j > CT) {
#pragma unroll 1
                for (int i = 0; i < CT; i += 2) {
                    mid(j + i, sa, sb);
                    mid(j + i + 1, sb, sa);
                }
                j += CT;
                if (wave_active) chunk_end(j / CT - 1);
            }
        }
        for (; j + 2 < nkt; j += 2) {
            mid(j, sa, sb);
            mid(j + 1, sb, sa);
        }
        bool in_a = true;
        if (j + 1 < nkt) { mid(j, sa, sb); in_a = false; ++j; }
        if (!wave_active) return;
        // last tile: mask the keys beyond klen, soft-max, P V
        auto last = [&](f32x16 (&sc)[2]) {
            if ((klen & (BKV - 1)) && nkt == nkt_all) {
                const int kbase = (nkt - 1) * BKV;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + t * 32 + key_of(e, h) >= klen) sc[t][e] = -INFINITY;
            }
            VFrag va, vb;
            vload((nkt - 1) & 1, 0, 0, va);
            vload((nkt - 1) & 1, 0, 1, vb);
            const float alpha = softmax_a(sc);
            rescale(alpha);
            __builtin_amdgcn_sched_barrier(0);
            pv_b((nkt - 1) & 1, sc, alpha, va, vb);
        };
        if (in_a) last(sa); else last(sb);
    }
    finish();
#ifdef PRAM_PROFILING
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        attn_prof[blockIdx.x][0] = prof_t0;
        attn_prof[blockIdx.x][1] = wall_clock64();
        attn_prof[blockIdx.x][2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
        attn_prof[blockIdx.x][3] = __builtin_readcyclecounter() - prof_c0;
    }
    if (lane == 0 && blockIdx.y == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&attn_phase[wave & 7][i], prof_ph[i]);
#endif

    if (q_ok) {
        const size_t row = (size_t)b * p.m_max + qrow;
        const int clast = (nkt - 1) / CT;          // split mode: the workgroup's last chunk is parked like the others
        if (HI && MODE == 0 && p.out16) {          // fp16 context (what the consuming GEMM would round it to while staging)
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            _Float16* o16 = p.out16 + row * p.ldo16 + head * D;
#pragma unroll
            for (int dn = 0; dn < 2; ++dn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4v hv;
                    hv[0] = (_Float16)oacc[dn][4 * g + 0]; hv[1] = (_Float16)oacc[dn][4 * g + 1];
                    hv[2] = (_Float16)oacc[dn][4 * g + 2]; hv[3] = (_Float16)oacc[dn][4 * g + 3];
                    *reinterpret_cast<half4v*>(o16 + dn * 32 + 8 * g + 4 * h) = hv;
                }
        } else {
            float* op = MODE == 2 ? p.part_o + ((size_t)clast * p.batch * p.m_max + row) * (p.heads * D) + head * D
                                  : p.out + row * p.ldo + head * D;
#pragma unroll
            for (int dn = 0; dn < 2; ++dn)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(op + dn * 32 + 8 * g + 4 * h) =
                        make_float4(oacc[dn][4 * g + 0], oacc[dn][4 * g + 1], oacc[dn][4 * g + 2], oacc[dn][4 * g + 3]);
        }
        if (h == 0) {
            const size_t li = ((size_t)b * p.heads + head) * p.m_max + qrow;
            if (MODE == 2) p.part_l[(size_t)((nkt - 1) / CT) * p.batch * p.heads * p.m_max + li] = l_tot2;
            else if (p.lse2) p.lse2[li] = l_tot2;
        }
    }
}

// Split mode, second step: the left fold over the key chunks of a (query row, head) in chunk order — fold_weights / fold_value
// exactly as the fused kernel (MODE 1) applies them.  One lane per four output dims (16 lanes per (row, head)); the chunks' partial
// outputs and log-sums are all requested BEFORE the first fold (up to CMAX chunks at a time: independent loads, one round trip) —
// the first version asked for chunk c + 1 only after folding chunk c, one L2 round trip per chunk on the one-query path's critical
// path (9 us per launch for a 4-chunk fold, profiles/r05_latency_step_anatomy.md).
__global__ __launch_bounds__(256) void combine_x3_kernel(ArgsX p) {
    constexpr int CMAX = 8;
    const int sub = threadIdx.x & 15;                                           // dims 4 sub .. 4 sub + 3
    const long long unit = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);     // (b, qrow, head)
    const long long units = (long long)p.batch * p.m_max * p.heads;
    if (unit >= units) return;
    const int head = (int)(unit % p.heads);
    const long long row = unit / p.heads;                                       // b * m_max + qrow
    const int b = (int)(row / p.m_max), qrow = (int)(row - (long long)b * p.m_max);
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qrow >= qlen || klen <= 0) return;            // empty key set: the attention kernel wrote the zeros itself
    const int nchunk = (klen + p.chunk_tiles * BKV - 1) / (p.chunk_tiles * BKV);
    const size_t li = ((size_t)b * p.heads + head) * p.m_max + qrow;
    const size_t ostride = (size_t)p.batch * p.m_max * (p.heads * D), lstride = (size_t)p.batch * p.heads * p.m_max;
    const float* po = p.part_o + (size_t)row * (p.heads * D) + head * D + 4 * sub;
    const float* pl = p.part_l + li;
    float4 ot = make_float4(0.f, 0.f, 0.f, 0.f);
    float lt = -INFINITY;
    for (int c0 = 0; c0 < nchunk; c0 += CMAX) {
        float4 oc[CMAX];
        float lc[CMAX];
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            const int c = min(c0 + i, nchunk - 1);      // (clamped: a legal address, the value is not used)
            oc[i] = *reinterpret_cast<const float4*>(po + (size_t)c * ostride);
            lc[i] = pl[(size_t)c * lstride];
        }
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            if (c0 + i < nchunk) {
                float at, ac, lnew;
                fold_weights(lt, lc[i], &at, &ac, &lnew);
                ot.x = fold_value(ot.x, oc[i].x, at, ac);
                ot.y = fold_value(ot.y, oc[i].y, at, ac);
                ot.z = fold_value(ot.z, oc[i].z, at, ac);
                ot.w = fold_value(ot.w, oc[i].w, at, ac);
                lt = lnew;
            }
        }
    }
    *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldo + head * D + 4 * sub) = ot;
    if (p.lse2 && sub == 0) p.lse2[li] = lt;
}

// V^T planes for attention_x3_kernel: [seq][head][64 dims][tv positions] fp16, position = 64 * (t / 64) + pos_of_key(t % 64),
// zeros for t >= lens[seq] (a masked key then meets a finite value).  One workgroup = 64 tokens of one (sequence, head):
// 16-byte reads along d, a 64 x 64 transpose through LDS, 16-byte writes along the positions.
struct VtArgs {
    const _Float16* vh; const _Float16* vl; int ldv;      // row-major value planes [seq * t_max][ldv], head h at columns 64 h ..
    _Float16* oh; _Float16* ol;
    const int* lens;
    int t_max, tv, heads;
};

__global__ __launch_bounds__(256) void vt_kernel(VtArgs p) {
    __shared__ _Float16 sh[2][64][72];      // [plane][d][pos], rows padded to 144 B: the transposing 2-byte writes spread over the banks
    const int blk = blockIdx.x, head = blockIdx.y, seq = blockIdx.z;
    const int len = p.lens ? min(p.lens[seq], p.t_max) : p.t_max;
    const int tid = threadIdx.x;
    const int key = tid >> 2, seg = tid & 3;                 // token blk*64 + key, dims 16 seg .. 16 seg + 15
    const int t = blk * 64 + key;
    const int pos = pos_of_key(key);
    half8 a0, a1, b0, b1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)0.f; a1[i] = a0[i]; b0[i] = a0[i]; b1[i] = a0[i]; }
    const bool two = p.vl != nullptr;                        // single-product mode: one plane
    if (t < len) {
        const size_t src = ((size_t)seq * p.t_max + t) * p.ldv + head * D + seg * 16;
        a0 = *reinterpret_cast<const half8*>(p.vh + src);
        a1 = *reinterpret_cast<const half8*>(p.vh + src + 8);
        if (two) {
            b0 = *reinterpret_cast<const half8*>(p.vl + src);
            b1 = *reinterpret_cast<const half8*>(p.vl + src + 8);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sh[0][seg * 16 + i][pos] = a0[i];
        sh[0][seg * 16 + 8 + i][pos] = a1[i];
        sh[1][seg * 16 + i][pos] = b0[i];
        sh[1][seg * 16 + 8 + i][pos] = b1[i];
    }
    __syncthreads();
    const int d = tid >> 2, part = tid & 3;                  // dim d, positions 16 part .. 16 part + 15
    const size_t dst = (((size_t)seq * p.heads + head) * D + d) * p.tv + blk * 64 + part * 16;
    *reinterpret_cast<half8*>(p.oh + dst) = *reinterpret_cast<const half8*>(&sh[0][d][part * 16]);
    *reinterpret_cast<half8*>(p.oh + dst + 8) = *reinterpret_cast<const half8*>(&sh[0][d][part * 16 + 8]);
    if (two) {
        *reinterpret_cast<half8*>(p.ol + dst) = *reinterpret_cast<const half8*>(&sh[1][d][part * 16]);
        *reinterpret_cast<half8*>(p.ol + dst + 8) = *reinterpret_cast<const half8*>(&sh[1][d][part * 16 + 8]);
    }
}

// ---------------------------------------------------------------- column means of the attention matrix (AdaGML)
// colmean[kb][j] = 1 / (H m_b) * sum_h sum_i exp2(scale2 * q_i . k_j - lse2[b, h, i])   (nets/adagml.py:92-104: the mean over heads
// and query rows of the soft-max matrix, one number per key token — what the pooling MLP turns into a keep / prune logit).
// Second pass over Q K^T with the row normalisers of the first (lse2 of attention_x3_kernel); the operands are the same split
// planes, three MFMAs per product.  One workgroup = 128 keys (32 per wave, their K fragments live in registers as the MFMA's
// row operand); the queries stream through LDS in 64-row tiles as the column operand, so a lane owns one query per 32-column
// block — its lse2 is one scalar load — and the per-key sums accumulate in registers across the whole loop; lanes are
// combined once at the end.  No atomics: the result does not depend on the launch geometry.
struct ColArgsX {
    const _Float16* qh; const _Float16* ql; const _Float16* kh; const _Float16* kl;
    const float* lse2; float* colmean;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, batch, heads, m_max, n_max;
    float scale2;
    int kv_shift;
};

__global__ __launch_bounds__(256, 2) void colmean_x3_kernel(ColArgsX p) {
    __shared__ __attribute__((aligned(16))) _Float16 sqh[2][BKV * D];
    __shared__ __attribute__((aligned(16))) _Float16 sql[2][BKV * D];
    const int b = blockIdx.y;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if ((int)blockIdx.x * 128 >= klen) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int key0 = blockIdx.x * 128 + wave * 32;
    const int lrow = tid >> 3, lseg = tid & 7;
    const int nqt = (qlen + BKV - 1) / BKV;

    float cacc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cacc[e] = 0.f;

    half8 grh[2], grl[2];
    // Q rows beyond qlen are duplicates of the last valid row: their (finite) scores meet lse2 = +inf below
    auto gload = [&](int head, int qt) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const size_t qi = (size_t)min(qt * BKV + lrow + 32 * pp, qlen - 1);
            const size_t off = ((size_t)b * p.m_max + qi) * p.ldq + head * D + lseg * 8;
            grh[pp] = *reinterpret_cast<const half8*>(p.qh + off);
            grl[pp] = *reinterpret_cast<const half8*>(p.ql + off);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int row = lrow + 32 * pp;
            const int off = row * D + ((lseg ^ ((row >> 1) & 7)) << 3);
            *reinterpret_cast<half8*>(&sqh[buf][off]) = grh[pp];
            *reinterpret_cast<half8*>(&sql[buf][off]) = grl[pp];
        }
    };
    const int total = p.heads * nqt;      // (head, q tile) steps, double buffered across head boundaries
    if (total > 0) {
        gload(0, 0);
        lstore(0);
    }
    __syncthreads();
    half8 kh[4], kl[4];
    int head = 0, qt = 0;
    for (int it = 0; it < total; ++it) {
        const int cur = it & 1;
        if (qt == 0) {
            const size_t koff = ((size_t)kb * p.n_max + min(key0 + r, klen - 1)) * p.ldk + head * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                kh[c] = *reinterpret_cast<const half8*>(p.kh + koff + c * 16 + h * 8);
                kl[c] = *reinterpret_cast<const half8*>(p.kl + koff + c * 16 + h * 8);
            }
        }
        int nhead = head, nqt_ = qt + 1;
        if (nqt_ == nqt) { nqt_ = 0; ++nhead; }
        const bool more = it + 1 < total;
        if (more) gload(nhead, nqt_);
        const float* lse = p.lse2 + ((size_t)b * p.heads + head) * p.m_max;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qi = qt * BKV + t * 32 + r;
            const float l2 = (qi < qlen) ? lse[qi] : INFINITY;
            f32x16 st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = ((2 * c + h) ^ ((r >> 1) & 7)) << 3;
                const half8 qh = *reinterpret_cast<const half8*>(&sqh[cur][(t * 32 + r) * D + slot]);
                const half8 ql = *reinterpret_cast<const half8*>(&sql[cur][(t * 32 + r) * D + slot]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[c], qh, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], ql, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], qh, st, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) cacc[e] += __builtin_amdgcn_exp2f(fmaf(st[e], p.scale2, -l2));
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        head = nhead;
        qt = nqt_;
    }
    const float norm = qlen > 0 ? 1.0f / ((float)p.heads * (float)qlen) : 0.f;      // no queries: reported as 0, not NaN
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

/* Column means of the soft-max matrix of pram_attention_x3_f32 (AdaGML's token scores): colmean[kb][j] = mean over heads and
   over the q_lens[b] query rows of softmax_row(scale q k^T)[i][j], kb = (b + kv_shift) % batch, keys j >= k_lens[kb] untouched.
   q / k: the split planes the attention call used; lse2: its row log-sum-exps (log2 domain), [batch][heads][m_max]. */
extern "C" int pram_attention_x3_colmean_f32(const void* q_hi, const void* q_lo, int ldq, const void* k_hi, const void* k_lo, int ldk,
                                             const float* lse2, float* colmean, const int* q_lens, const int* k_lens, int batch,
                                             int heads, int m_max, int n_max, float scale, int kv_shift, void* stream) {
    PRAM_REQUIRE(q_hi && q_lo && k_hi && k_lo && lse2 && colmean, "pram_attention_x3_colmean_f32: null pointer");
    PRAM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0, "pram_attention_x3_colmean_f32: ld of the fp16 planes must be a multiple of 8");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0 && kv_shift >= 0, "pram_attention_x3_colmean_f32: bad sizes");
    if (batch == 0 || n_max == 0) return PRAM_OK;
    ColArgsX p{(const _Float16*)q_hi, (const _Float16*)q_lo, (const _Float16*)k_hi, (const _Float16*)k_lo, lse2, colmean, q_lens, k_lens,
               ldq, ldk, batch, heads, m_max, n_max, scale * LOG2E / (pram_act_scale() * pram_act_scale()), kv_shift};
    hipLaunchKernelGGL(colmean_x3_kernel, dim3(cdiv(n_max, 128), batch), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_x3_colmean_f32");
}

/* Value planes [seqs * t_max][ldv] (head h at columns 64 h .., as pram_linear_x3_f32 writes them) -> transposed, key-permuted
   planes [seqs][heads][64][tv], tv = t_max rounded up to a multiple of 64; tokens t >= lens[seq] (NULL = t_max) become zeros. */
extern "C" int pram_attention_x3_vt(const void* v_hi, const void* v_lo, int ldv, void* vt_hi, void* vt_lo, const int* lens,
                                    int seqs, int heads, int t_max, void* stream) {
    PRAM_REQUIRE(v_hi && vt_hi && ((v_lo == nullptr) == (vt_lo == nullptr)), "pram_attention_x3_vt: null pointer (the lo planes go together)");
    PRAM_REQUIRE(ldv % 8 == 0 && heads > 0 && seqs >= 0 && t_max >= 0, "pram_attention_x3_vt: bad sizes");
    if (seqs == 0 || t_max == 0) return PRAM_OK;
    const int tv = cdiv(t_max, 64) * 64;
    VtArgs p{(const _Float16*)v_hi, (const _Float16*)v_lo, ldv, (_Float16*)vt_hi, (_Float16*)vt_lo, lens, t_max, tv, heads};
    hipLaunchKernelGGL(vt_kernel, dim3(tv / 64, heads, seqs), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_x3_vt");
}

// fused or split launch (see "Key chunks" above).  The mode never changes the result.  Both park chunk results in the workspace
// ([chunks][rows][heads * 64] + [chunks][batch][heads][rows] floats) when a sequence has more than one chunk.
static int g_chunk_tiles = 0;
static int chunk_tiles() {
    if (g_chunk_tiles == 0) {
        const char* e = getenv("PRAM_ATTN_CHUNK_KEYS");
        int t = e ? atoi(e) / BKV : DEFAULT_CHUNK_TILES;
        g_chunk_tiles = (t >= 2 && t % 2 == 0) ? t : DEFAULT_CHUNK_TILES;
    }
    return g_chunk_tiles;
}
static size_t x3_ws_bytes(int batch, int heads, int m_max, int n_max) {
    const int nchunks = cdiv(n_max, chunk_tiles() * BKV);
    if (n_max < 1024 || nchunks < 2) return 0;
    return (size_t)nchunks * batch * m_max * (heads * D + heads) * sizeof(float);
}

/* Keys per chunk of pram_attention_x3_f32 (a multiple of 128; default 4096, or PRAM_ATTN_CHUNK_KEYS): process-wide, to be set
   before the first launch — it fixes where every launch folds its partial soft-maxes.  0 keeps the current value; returns it. */
extern "C" int pram_attention_x3_set_chunk_keys(int keys) {
    if (keys >= 128 && keys % 128 == 0) g_chunk_tiles = keys / BKV;
    return chunk_tiles() * BKV;
}

static int g_split_target = SPLIT_TARGET;
// workgroup groups a launch is split into along the keys (1 = fused): as many as bring the grid to ~one workgroup per CU, each a
// whole number of 512-key chunks
static int x3_split_groups(int batch, int heads, int m_max, int n_max) {
    if (n_max < 1024) return 1;
    const long units = (long)batch * heads * cdiv(m_max, BQ);
    const int nchunks = cdiv(n_max, chunk_tiles() * BKV);
    const long g = units > 0 ? g_split_target / units : 1;
    return (int)(g < 2 ? 1 : (g > nchunks ? nchunks : g));
}

/* Tuning / test knob: under-filled launches of pram_attention_x3_f32 are split along the keys into as many workgroup groups as bring
   the grid to `workgroups` (default 256 = one per CU; 0 = never split; negative restores the default).  The mode never changes a
   result bit; returns the previous value. */
extern "C" int pram_attention_x3_set_split_target(int workgroups) {
    const int old = g_split_target;
    g_split_target = workgroups < 0 ? SPLIT_TARGET : workgroups;
    return old;
}

extern "C" size_t pram_attention_x3_workspace_bytes(int batch, int heads, int m_max, int n_max) {
    return x3_ws_bytes(batch, heads, m_max, n_max);
}

/* > 1 if pram_attention_x3_f32 with a workspace would run this launch in the split mode (that many key groups), else 1. */
extern "C" int pram_attention_x3_is_split(int batch, int heads, int m_max, int n_max) { return x3_split_groups(batch, heads, m_max, n_max); }

/* MFMA instructions (v_mfma_f32_32x32x16_f16) the kernel pram_attention_x3_f32 launches for n_max keys issues per 64-key tile and
   32-query wave; a single-product fp16 attention needs 16 (8 for K Q^T, 8 for P V).  40 = three products for the scores, two
   for P V (probabilities as one fp16); 48 = three and three (below 1024 keys).  bench.py prices its roofline with this. */
static int g_p_split = -1;
static bool p_split_always() {
    if (g_p_split < 0) {
        const char* e = getenv("PRAM_ATTN_P");      // "split" / "fp16": see pram_attention_x3_set_p_split
        g_p_split = (e && e[0] == 'f') ? 0 : 1;
    }
    return g_p_split != 0;
}

/* How the probabilities enter P V from 1024 keys on (below, always split): 1 = as two fp16 parts like every other operand (three
   MFMAs per product, 48 per tile: the default), 0 = as ONE fp16 (two MFMAs, 40 per tile: ~15 % less attention time, but the
   2^-12 rounding of every probability shows: SegNetViT logits 7e-4..1.3e-3 from the fp32 oracle instead of 4e-5 (not inside the 1e-3 parity bar everywhere: an opt-in, not parity-gated) on the synthetic token
   sets of tests/, 0.15 % of the landmark arg-maxes flipped).  Process-wide; negative = query.  Returns the value in force. */
extern "C" int pram_attention_x3_set_p_split(int split) {
    if (split >= 0) g_p_split = split ? 1 : 0;
    return p_split_always() ? 1 : 0;
}

#ifdef PRAM_PROFILING
/* profiling builds only: the per-workgroup timeline of the last pipelined launch (4096 x 4 uint64, see attn_prof) */
extern "C" int pram_debug_attention_timeline(unsigned long long* out16384) {
    if (hipMemcpyFromSymbol(out16384, HIP_SYMBOL(attn_prof), sizeof(unsigned long long) * 4096 * 4) != hipSuccess) return pram_launch_status("pram_debug_attention_timeline");
    return PRAM_OK;
}
/* ... and the per-wave phase clocks (8 x 4 uint64, see attn_phase), summed since the last reset */
extern "C" int pram_debug_attention_phases(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(attn_phase), sizeof(unsigned long long) * 32) != hipSuccess) return pram_launch_status("pram_debug_attention_phases");
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(attn_phase), z, sizeof(z)) != hipSuccess) return pram_launch_status("pram_debug_attention_phases");
    }
    return PRAM_OK;
}
#endif

extern "C" int pram_attention_x3_mfma_per_tile(int n_max) { return (n_max < 1024 || p_split_always()) ? 48 : 40; }

/* Split-fp16 flash attention.  q / k: row-major planes written by pram_linear_x3_f32 (value * 16 = hi + lo; ld* in halves,
   16-byte aligned rows and head offsets); vt: the V^T planes of pram_attention_x3_vt for the KEY side ([batch][heads][64][tv],
   tv = n_max rounded up to 64).  kv_shift > 0: keys / values of sequence s come from sequence (s + kv_shift) % batch (cross
   attention, both directions in one launch).  workspace (may be NULL): pram_attention_x3_workspace_bytes(...) bytes for the
   split mode of under-filled launches; without it the launch is fused — same bits either way. */
extern "C" int pram_attention_x3_f32(const void* q_hi, const void* q_lo, int ldq, const void* k_hi, const void* k_lo, int ldk,
                                     const void* vt_hi, const void* vt_lo, float* out, int ldo, float* lse2,
                                     const int* q_lens, const int* k_lens, int batch, int heads, int m_max, int n_max,
                                     float scale, int kv_shift, void* workspace, size_t workspace_bytes, void* stream) {
    PRAM_REQUIRE(q_hi && q_lo && k_hi && k_lo && vt_hi && vt_lo && out, "pram_attention_x3_f32: null pointer");
    PRAM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "pram_attention_x3_f32: ld of the fp16 planes must be a multiple of 8");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0 && kv_shift >= 0, "pram_attention_x3_f32: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_x3_f32: empty key set");
    // K / V tiles are addressed by 32-bit byte offsets from a per-sequence base
    PRAM_REQUIRE((long long)n_max * ldk * 2 < (1ll << 32) && 64ll * (cdiv(n_max, 64) * 64) * 2 < (1ll << 32), "pram_attention_x3_f32: a sequence's K rows / V^T planes must span < 4 GiB");
    ArgsX p{(const _Float16*)q_hi, (const _Float16*)q_lo, (const _Float16*)k_hi, (const _Float16*)k_lo, (const _Float16*)vt_hi,
            (const _Float16*)vt_lo, out, lse2, q_lens, k_lens, ldq, ldk, cdiv(n_max, 64) * 64, ldo, batch, heads, m_max, n_max,
            scale * LOG2E / (pram_act_scale() * pram_act_scale()), cdiv(m_max, BQ), kv_shift, pram_act_scale(), 1, 0, chunk_tiles(), nullptr, nullptr};
    const dim3 grid(batch * heads * p.q_tiles), blk(256);
    hipStream_t st = (hipStream_t)stream;
    {
        if (n_max < 1024) {
            // short key sets (always two-part probabilities): 256-row workgroups in the phases form when they fill the chip — many small
            // pairs in one grouped call (the matcher's real call pattern) — the 128-row interleaved kernel otherwise
            static const char* wv4 = prof_env("PRAM_ATTN_WAVES");
            const long u256 = (long)batch * heads * cdiv(m_max, 2 * BQ);
            if (u256 >= 256 && !(wv4 && wv4[0] == '4')) {
                p.q_tiles = cdiv(m_max, 2 * BQ);
                hipLaunchKernelGGL((attention_x3_pipe_kernel<true, false, 0, 2 * NW>), dim3(batch * heads * p.q_tiles), dim3(2 * NW * 64), 0, st, p);
            } else {
                hipLaunchKernelGGL((attention_x3_pipe_kernel<true, false, 0>), grid, blk, 0, st, p);
            }
            return pram_launch_status("pram_attention_x3_f32");
        }
        const bool psplit = p_split_always();      // probabilities as two fp16 parts (three MFMAs per P V product) also from 1024 keys on
        // the key-chunk kernels (128-row workgroups) run the phases form when the grid puts two workgroups on a CU — two waves on
        // a SIMD — and a workgroup walks at least 2048 keys; the interleaved form otherwise (one frame split four ways: one wave
        // per SIMD, nothing to overlap with; 512-key groups: the longer prologue of the phases form is not won back)
#define PRAM_LAUNCH_PIPE_(MODE_, GRID_, PH_)                                                                                  \
    do {                                                                                                                      \
        if (psplit) hipLaunchKernelGGL((attention_x3_pipe_kernel<true, false, MODE_, NW, PH_>), GRID_, blk, 0, st, p);        \
        else hipLaunchKernelGGL((attention_x3_pipe_kernel<false, false, MODE_, NW, PH_>), GRID_, blk, 0, st, p);              \
    } while (0)
#define PRAM_LAUNCH_PIPE(MODE_, GRID_, TILES_)                                                                                \
    do {                                                                                                                      \
        const dim3 g_ = GRID_;                                                                                                \
        const bool ph_ = MODE_ != 0 ? (AX_STYLEC == 1 ? (long)g_.x * g_.y >= 2 * 256 && (TILES_) >= 32 : AX_STYLEC != 0)      \
                                    : AX_STYLE4 != 0;                                                                         \
        if (ph_) PRAM_LAUNCH_PIPE_(MODE_, g_, true);                                                                          \
        else PRAM_LAUNCH_PIPE_(MODE_, g_, false);                                                                             \
    } while (0)
        const size_t need = x3_ws_bytes(batch, heads, m_max, n_max);
        static const char* force = prof_env("PRAM_ATTN_MODE");      // profiling only: 0 = the unchunked kernel (different last bits), 1 = fused, 2 = split
        const int fm = force ? atoi(force) : -1;
        const int nchunks = cdiv(n_max, chunk_tiles() * BKV);
        if (fm == 0 || (nchunks < 2 && fm != 1)) {      // one chunk per sequence: the walk is the unchunked kernel's (a fold from (0, -inf) is exact)
            // a grid that fills the chip with 256-row workgroups runs eight waves per workgroup (NWV): every K / V tile is staged once
            // per 256 query rows.  PRAM_ATTN_WAVES=4 keeps the 128-row workgroups (profiling); the choice never changes a bit.
            static const char* wv = prof_env("PRAM_ATTN_WAVES");
            const long units256 = (long)batch * heads * cdiv(m_max, 2 * BQ);
            if (units256 >= 256 && !(wv && wv[0] == '4')) {
                p.q_tiles = cdiv(m_max, 2 * BQ);
                const dim3 grid8(batch * heads * p.q_tiles), blk8(2 * NW * 64);
                if (psplit) hipLaunchKernelGGL((attention_x3_pipe_kernel<true, false, 0, 2 * NW>), grid8, blk8, 0, st, p);
                else hipLaunchKernelGGL((attention_x3_pipe_kernel<false, false, 0, 2 * NW>), grid8, blk8, 0, st, p);
                return pram_launch_status("pram_attention_x3_f32");
            }
            PRAM_LAUNCH_PIPE(0, grid, 0);
            return pram_launch_status("pram_attention_x3_f32");
        }
        int groups = (workspace == nullptr || workspace_bytes < need || need == 0) ? 1 : x3_split_groups(batch, heads, m_max, n_max);
        if (fm == 2 && groups < 2 && workspace != nullptr && need != 0 && workspace_bytes >= need) groups = nchunks;
        if (groups < 2 || fm == 1 || fm == 3) {          // fused: the workgroup folds its chunks in registers (no workspace needed)
            PRAM_LAUNCH_PIPE(1, grid, cdiv(n_max, BKV));
            return pram_launch_status("pram_attention_x3_f32");
        }
        p.part_o = (float*)workspace;
        p.part_l = p.part_o + (size_t)nchunks * batch * m_max * heads * D;
        p.group_tiles = cdiv(nchunks, groups) * p.chunk_tiles;
        p.nsplit = cdiv(nchunks * p.chunk_tiles, p.group_tiles);
        PRAM_LAUNCH_PIPE(2, dim3(grid.x, p.nsplit), p.group_tiles);
#undef PRAM_LAUNCH_PIPE
#undef PRAM_LAUNCH_PIPE_
        const long long rows = (long long)batch * m_max * heads;
        hipLaunchKernelGGL(combine_x3_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, p);
        return pram_launch_status("pram_attention_x3_f32");
    }
    return pram_launch_status("pram_attention_x3_f32");
}

// single-product launches: eight waves per workgroup on grids that fill the chip with 256-row workgroups (as pram_attention_x3_f32)
static void launch_h16t(ArgsX& p, hipStream_t st) {
    static const char* wv = prof_env("PRAM_ATTN_WAVES");
    const long units256 = (long)p.batch * p.heads * cdiv(p.m_max, 2 * BQ);
    if (units256 >= 256 && !(wv && wv[0] == '4')) {
        p.q_tiles = cdiv(p.m_max, 2 * BQ);
        hipLaunchKernelGGL((attention_x3_pipe_kernel<false, true, 0, 2 * NW>), dim3(p.batch * p.heads * p.q_tiles), dim3(2 * NW * 64), 0, st, p);
        return;
    }
    hipLaunchKernelGGL((attention_x3_pipe_kernel<false, true, 0>), dim3(p.batch * p.heads * p.q_tiles), dim3(256), 0, st, p);
}

/* Single-product ("fp16 MFMA path", BASELINE C5) flash attention on the pipelined kernel: q / k = fp16 row-major [rows][ld]
   (what pram_linear_f16_h16 writes), vt = the transposed, key-permuted fp16 values of pram_attention_x3_vt called with NULL lo
   planes.  One fp16 MFMA per product, fp32 accumulation and soft-max; the probabilities are rounded to fp16 (scaled by 2^P_EXP_SHIFT = 2^7 under the lazy maximum)
   like the inputs.  Output fp32.  Same arguments otherwise as pram_attention_x3_f32. */
extern "C" int pram_attention_h16t_f32(const void* q16, int ldq, const void* k16, int ldk, const void* vt16, float* out, int ldo,
                                       float* lse2, const int* q_lens, const int* k_lens, int batch, int heads, int m_max,
                                       int n_max, float scale, int kv_shift, void* stream) {
    PRAM_REQUIRE(q16 && k16 && vt16 && out, "pram_attention_h16t_f32: null pointer");
    PRAM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "pram_attention_h16t_f32: ld of the fp16 operands must be a multiple of 8");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0 && kv_shift >= 0, "pram_attention_h16t_f32: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_h16t_f32: empty key set");
    // K / V tiles are addressed by 32-bit byte offsets from a per-sequence base
    PRAM_REQUIRE((long long)n_max * ldk * 2 < (1ll << 32) && 64ll * (cdiv(n_max, 64) * 64) * 2 < (1ll << 32), "pram_attention_h16t_f32: a sequence's K rows / V^T planes must span < 4 GiB");
    ArgsX p{(const _Float16*)q16, nullptr, (const _Float16*)k16, nullptr, (const _Float16*)vt16, nullptr, out, lse2, q_lens, k_lens,
            ldq, ldk, cdiv(n_max, 64) * 64, ldo, batch, heads, m_max, n_max, scale * LOG2E, cdiv(m_max, BQ), kv_shift, 1.0f, 1, 0, 0, nullptr, nullptr};
    launch_h16t(p, (hipStream_t)stream);
    return pram_launch_status("pram_attention_h16t_f32");
}

/* pram_attention_h16t_f32 with the context written as fp16 [batch * m_max][ldo16] (ldo16 % 4 == 0): the operand format of the
   fp16 path's next GEMM (pram_linear_f16_ssq_h16) — the same values that GEMM would round the fp32 context to while staging it. */
extern "C" int pram_attention_h16t_h16(const void* q16, int ldq, const void* k16, int ldk, const void* vt16, void* out16, int ldo16,
                                       float* lse2, const int* q_lens, const int* k_lens, int batch, int heads, int m_max,
                                       int n_max, float scale, int kv_shift, void* stream) {
    PRAM_REQUIRE(q16 && k16 && vt16 && out16, "pram_attention_h16t_h16: null pointer");
    PRAM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo16 % 4 == 0, "pram_attention_h16t_h16: ld of the fp16 operands must be a multiple of 8 (output: 4)");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0 && kv_shift >= 0, "pram_attention_h16t_h16: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_h16t_h16: empty key set");
    // K / V tiles are addressed by 32-bit byte offsets from a per-sequence base
    PRAM_REQUIRE((long long)n_max * ldk * 2 < (1ll << 32) && 64ll * (cdiv(n_max, 64) * 64) * 2 < (1ll << 32), "pram_attention_h16t_h16: a sequence's K rows / V^T planes must span < 4 GiB");
    ArgsX p{(const _Float16*)q16, nullptr, (const _Float16*)k16, nullptr, (const _Float16*)vt16, nullptr, nullptr, lse2, q_lens, k_lens,
            ldq, ldk, cdiv(n_max, 64) * 64, 0, batch, heads, m_max, n_max, scale * LOG2E, cdiv(m_max, BQ), kv_shift, 1.0f, 1, 0, 0, nullptr, nullptr,
            (_Float16*)out16, ldo16};
    launch_h16t(p, (hipStream_t)stream);
    return pram_launch_status("pram_attention_h16t_h16");
}
