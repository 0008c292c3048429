// fp16-input / fp32-accumulate flash attention on v_mfma_f32_32x32x16_f16 — the "fp16 MFMA path" of
// BASELINE config C5 (Aachen, 4096 keypoints).  NOT used by the fp32 parity configs: inputs are rounded
// to fp16 (11-bit mantissa), so results carry ~1e-3 relative error and have their own documented tolerance.
//
// Same interface and the same transposed register design as attention.hip (query row = lane in every
// accumulator, P never leaves registers): Q/K/V/O stay fp32 in HBM; Q is rounded once per workgroup, K and V
// while they are staged into LDS.  One MFMA contracts 16 values of k: lane (r = l & 31, h = l >> 5) supplies
// 8 consecutive fp16 per operand.  Which k a (half-wave, slot) pair stands for is free as long as A and B
// agree, which is what makes both products operand-ready:
//   S^T[key][q]:  A = K[key = r][d = 16c + 8h + i],  B = Q[q = r][same d]            (4 MFMA per 32 keys)
//   O^T[d][q]  :  B = the 8 accumulator registers e = 8u .. 8u+7 of exp(S^T), i.e. keys
//                 (i & 3) + 16u + 8(i >> 2) + 4h, rounded to fp16;  A = V^T[d = r][those keys], read as one
//                 ds_read_b128 because V is staged TRANSPOSED and key-permuted: LDS V^T[d][pos] with
//                 pos(key) = 32t + 16u + 8h + i                                      (4 MFMA per 32 keys)
// Per 64-key tile per wave: 16 MFMA x 32 cycles = 512 matrix cycles against ~600 VALU cycles of softmax +
// conversions, so this kernel is VALU/MFMA balanced, not matrix-bound like the fp32 one; LDS is 16 KiB per
// buffer pair (4+ workgroups per CU) so co-resident waves overlap the two pipes.
// LDS rows are 128 B (64 fp16); the 16-B slot is XOR-swizzled with (row >> 1) & 7 so every 16-lane group of
// ds_read_b128 covers 16 distinct (bank-row half, slot) pairs.
#include "common.h"
#include <math.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int D = 64, QW = 32, NW = 4, BQ = QW * NW, BKV = 64;
constexpr float LOG2E = 1.4426950408889634f;

struct Args {
    const float* q; const float* k; const float* v;
    float* out; float* lse2;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, ldv, ldo;
    int batch, heads, m_max, n_max;
    float scale2;
    int q_tiles;
    int kv_shift;   // as in attention.hip
};

struct alignas(16) Smem {
    _Float16 k[2][BKV * D];    // [key][d], slot-swizzled
    _Float16 vt[2][D * BKV];   // [d][pos(key)], slot-swizzled
};  // 32 KiB

__device__ __forceinline__ int key_of(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }
// position of key (0..63) inside a V^T row: inverse of key = 32t + (i&3) + 16u + 8(i>>2) + 4h
__device__ __forceinline__ int pos_of_key(int key) {
    const int t = key >> 5, w = key & 31;
    const int u = w >> 4, h = (w >> 2) & 1, i = (w & 3) | (((w >> 3) & 1) << 2);
    return t * 32 + u * 16 + h * 8 + i;
}

__global__ __launch_bounds__(256, 3) void attention_f16_kernel(Args p) {
    __shared__ Smem s;
    const int nblk = p.batch * p.heads * p.q_tiles;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int qt = id % p.q_tiles;
    const int bh = id / p.q_tiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qt * BQ >= qlen) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * BQ + wave * QW;
    const bool wave_active = q0 < qlen;
    const int qrow = q0 + r;
    const bool q_ok = qrow < qlen;
    if (klen <= 0) {   // empty key set: context defined as 0 (see attention.hip)
        if (q_ok) {
            float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<float4*>(op + c * 8 + h * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = 0.f;
        }
        return;
    }

    const float* qp = p.q + ((size_t)b * p.m_max + min(qrow, p.m_max - 1)) * p.ldq + head * D;
    const float* kp = p.k + (size_t)kb * p.n_max * p.ldk + head * D;
    const float* vp = p.v + (size_t)kb * p.n_max * p.ldv + head * D;

    // Q fragments: qf[c][i] = fp16(Q[qrow][16c + 8h + i])
    half8 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 a = *reinterpret_cast<const float4*>(qp + c * 16 + h * 8);
        float4 bq = *reinterpret_cast<const float4*>(qp + c * 16 + h * 8 + 4);
        if (!q_ok) { a = make_float4(0.f, 0.f, 0.f, 0.f); bq = a; }
        qf[c][0] = (_Float16)a.x; qf[c][1] = (_Float16)a.y; qf[c][2] = (_Float16)a.z; qf[c][3] = (_Float16)a.w;
        qf[c][4] = (_Float16)bq.x; qf[c][5] = (_Float16)bq.y; qf[c][6] = (_Float16)bq.z; qf[c][7] = (_Float16)bq.w;
    }

    const int lrow = tid >> 4, lq = tid & 15;      // staging: key lrow + 16p, floats 4*lq .. 4*lq+3
    float4 kr[4], vr[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int key = kt * BKV + lrow + 16 * pp;
            const int kc = min(key, klen - 1);
            kr[pp] = *reinterpret_cast<const float4*>(kp + (size_t)kc * p.ldk + lq * 4);
            vr[pp] = *reinterpret_cast<const float4*>(vp + (size_t)kc * p.ldv + lq * 4);
        }
    };
    auto lstore = [&](int buf, int kt) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int row = lrow + 16 * pp;
            const bool ok = kt * BKV + row < klen;
            float4 kv = kr[pp], vv = vr[pp];
            if (!ok) { kv = make_float4(0.f, 0.f, 0.f, 0.f); vv = kv; }
            // K: 4 fp16 = 8 bytes at [row][d = 4 lq ..]: slot = lq >> 1, half-slot = lq & 1
            half4 k4;
            k4[0] = (_Float16)kv.x; k4[1] = (_Float16)kv.y; k4[2] = (_Float16)kv.z; k4[3] = (_Float16)kv.w;
            const int kslot = (lq >> 1) ^ ((row >> 1) & 7);
            *reinterpret_cast<half4*>(&s.k[buf][row * D + kslot * 8 + (lq & 1) * 4]) = k4;
            // V^T: element (d = 4 lq + j, pos(row)) -> row d of vt, slot = pos >> 3 (swizzled by d), lane = pos & 7
            const int pos = pos_of_key(row);
            const float vj[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = lq * 4 + j;
                const int vslot = (pos >> 3) ^ ((d >> 1) & 7);
                s.vt[buf][d * BKV + vslot * 8 + (pos & 7)] = (_Float16)vj[j];
            }
        }
    };

    const int nkt = (klen + BKV - 1) / BKV;
    gload(0);
    lstore(0, 0);
    __syncthreads();

    float m_run = -1.0e30f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) gload(kt + 1);

        if (wave_active) {
            f32x16 st[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[0][e] = 0.f; st[1][e] = 0.f; }
            const _Float16* sk = s.k[cur];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = (2 * c + h) ^ ((r >> 1) & 7);   // ((32 + r) >> 1) & 7 == (r >> 1) & 7
                const half8 k0 = *reinterpret_cast<const half8*>(sk + r * D + slot * 8);
                const half8 k1 = *reinterpret_cast<const half8*>(sk + (32 + r) * D + slot * 8);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[c], st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[c], st[1], 0, 0, 0);
            }
            if (!more && (klen & (BKV - 1))) {
                const int kbase = kt * BKV;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + t * 32 + key_of(e, h) >= klen) st[t][e] = -INFINITY;
            }
            float tmax = st[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, st[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[1][e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * p.scale2);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            half8 pf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][e], p.scale2, -m_new));   // v_exp_f32: argument <= 0
                    psum += pv;
                    pf[t][e >> 3][e & 7] = (_Float16)pv;
                }
            l_run = fmaf(l_run, alpha, psum);
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
            const _Float16* sv = s.vt[cur];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int slot = t * 4 + u * 2 + h;
                    const half8 v0 = *reinterpret_cast<const half8*>(sv + r * BKV + ((slot ^ ((r >> 1) & 7)) << 3));
                    const half8 v1 = *reinterpret_cast<const half8*>(sv + (32 + r) * BKV + ((slot ^ ((r >> 1) & 7)) << 3));
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[t][u], oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[t][u], oacc[1], 0, 0, 0);
                }
        }
        if (more) lstore(cur ^ 1, kt + 1);
        __syncthreads();
    }

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
        for (int dn = 0; dn < 2; ++dn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = make_float4(oacc[dn][4 * g + 0] * inv, oacc[dn][4 * g + 1] * inv,
                                             oacc[dn][4 * g + 2] * inv, oacc[dn][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + dn * 32 + 8 * g + 4 * h) = o;
            }
        if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = m_run + log2f(l_tot);
    }
}


// ---------------------------------------------------------------- fp16 inputs in HBM
// Same arithmetic as attention_f16_kernel — it rounds Q / K / V to fp16 while staging them; here the producing GEMM
// has already written exactly those fp16 values (pram_linear_f16_h16), so the outputs are bit-identical — but half
// the bytes cross L2 -> LDS, the staging needs no conversion and half the registers, and that buys the thing this
// kernel is actually short of: with 512 matrix cycles per tile, one K/V tile in flight leaves the loop waiting on a
// global-load round trip per tile.  Tile t+2 is requested while tile t is multiplied (two register sets, tile c in
// set c & 1; the LDS stage stays double buffered).
struct ArgsH {
    const _Float16* q; const _Float16* k; const _Float16* v;
    float* out; float* lse2;
    const int* q_lens; const int* k_lens;
    int ldq, ldk, ldv, ldo;          // ldq / ldk / ldv in halves
    int batch, heads, m_max, n_max;
    float scale2;
    int q_tiles;
    int kv_shift;
};

__global__ __launch_bounds__(256, 3) void attention_h16_kernel(ArgsH p) {
    __shared__ Smem s;
    const int nblk = p.batch * p.heads * p.q_tiles;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int qt = id % p.q_tiles;
    const int bh = id / p.q_tiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int qlen = p.q_lens ? p.q_lens[b] : p.m_max;
    const int kb = p.kv_shift ? (b + p.kv_shift) % p.batch : b;
    const int klen = p.k_lens ? p.k_lens[kb] : p.n_max;
    if (qt * BQ >= qlen) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * BQ + wave * QW;
    const bool wave_active = q0 < qlen;
    const int qrow = q0 + r;
    const bool q_ok = qrow < qlen;
    if (klen <= 0) {   // empty key set: context defined as 0 (see attention.hip)
        if (q_ok) {
            float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<float4*>(op + c * 8 + h * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = 0.f;
        }
        return;
    }

    const _Float16* qp = p.q + ((size_t)b * p.m_max + min(qrow, p.m_max - 1)) * p.ldq + head * D;
    const _Float16* kp = p.k + (size_t)kb * p.n_max * p.ldk + head * D;
    const _Float16* vp = p.v + (size_t)kb * p.n_max * p.ldv + head * D;

    half8 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = *reinterpret_cast<const half8*>(qp + c * 16 + h * 8);
        if (!q_ok)
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[c][i] = (_Float16)0.f;
    }

    const int lrow = tid >> 3, lseg = tid & 7;      // staging: key lrow + 32p, halves 8*lseg .. 8*lseg+7
    auto gload = [&](int kt, half8 (&kr)[2], half8 (&vr)[2]) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int kc = min(kt * BKV + lrow + 32 * pp, klen - 1);
            kr[pp] = *reinterpret_cast<const half8*>(kp + (size_t)kc * p.ldk + lseg * 8);
            vr[pp] = *reinterpret_cast<const half8*>(vp + (size_t)kc * p.ldv + lseg * 8);
        }
    };
    auto lstore = [&](int buf, int kt, const half8 (&kr)[2], const half8 (&vr)[2]) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int row = lrow + 32 * pp;
            const bool ok = kt * BKV + row < klen;
            half8 kv = kr[pp], vv = vr[pp];
            if (!ok)
#pragma unroll
                for (int i = 0; i < 8; ++i) { kv[i] = (_Float16)0.f; vv[i] = (_Float16)0.f; }
            *reinterpret_cast<half8*>(&s.k[buf][row * D + ((lseg ^ ((row >> 1) & 7)) << 3)]) = kv;
            const int pos = pos_of_key(row);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = lseg * 8 + j;
                const int vslot = (pos >> 3) ^ ((d >> 1) & 7);
                s.vt[buf][d * BKV + vslot * 8 + (pos & 7)] = vv[j];
            }
        }
    };

    const int nkt = (klen + BKV - 1) / BKV;
    half8 kA[2], vA[2], kB[2], vB[2];
    gload(0, kA, vA);
    if (nkt > 1) gload(1, kB, vB);
    lstore(0, 0, kA, vA);
    __syncthreads();

    float m_run = -1.0e30f, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }

    auto tile = [&](int kt, half8 (&kn)[2], half8 (&vn)[2], const half8 (&kc)[2], const half8 (&vc)[2]) {
        const int cur = kt & 1;
        if (kt + 2 < nkt) gload(kt + 2, kn, vn);       // into the set tile kt has left
        if (wave_active) {
            f32x16 st[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[0][e] = 0.f; st[1][e] = 0.f; }
            const _Float16* sk = s.k[cur];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = (2 * c + h) ^ ((r >> 1) & 7);
                const half8 k0 = *reinterpret_cast<const half8*>(sk + r * D + slot * 8);
                const half8 k1 = *reinterpret_cast<const half8*>(sk + (32 + r) * D + slot * 8);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[c], st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[c], st[1], 0, 0, 0);
            }
            if (kt + 1 == nkt && (klen & (BKV - 1))) {
                const int kbase = kt * BKV;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + t * 32 + key_of(e, h) >= klen) st[t][e] = -INFINITY;
            }
            float tmax = st[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, st[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[1][e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * p.scale2);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            half8 pf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][e], p.scale2, -m_new));
                    psum += pv;
                    pf[t][e >> 3][e & 7] = (_Float16)pv;
                }
            l_run = fmaf(l_run, alpha, psum);
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
            const _Float16* sv = s.vt[cur];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int slot = t * 4 + u * 2 + h;
                    const half8 v0 = *reinterpret_cast<const half8*>(sv + r * BKV + ((slot ^ ((r >> 1) & 7)) << 3));
                    const half8 v1 = *reinterpret_cast<const half8*>(sv + (32 + r) * BKV + ((slot ^ ((r >> 1) & 7)) << 3));
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[t][u], oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[t][u], oacc[1], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) lstore(cur ^ 1, kt + 1, kc, vc);   // tile kt+1, requested one iteration ago
        __syncthreads();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, kA, vA, kB, vB);
        if (kt + 1 < nkt) tile(kt + 1, kB, vB, kA, vA);
    }

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        float* op = p.out + ((size_t)b * p.m_max + qrow) * p.ldo + head * D;
#pragma unroll
        for (int dn = 0; dn < 2; ++dn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = make_float4(oacc[dn][4 * g + 0] * inv, oacc[dn][4 * g + 1] * inv,
                                             oacc[dn][4 * g + 2] * inv, oacc[dn][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + dn * 32 + 8 * g + 4 * h) = o;
            }
        if (p.lse2 && h == 0) p.lse2[((size_t)b * p.heads + head) * p.m_max + qrow] = m_run + log2f(l_tot);
    }
}

}  // namespace

extern "C" int pram_attention_f16_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
                                      int ldo, float* lse2, const int* q_lens, const int* k_lens, int batch, int heads,
                                      int m_max, int n_max, float scale, void* stream) {
    PRAM_REQUIRE(q && k && v && out, "pram_attention_f16_f32: null pointer");
    PRAM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "pram_attention_f16_f32: ld must be a multiple of 4");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0, "pram_attention_f16_f32: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_f16_f32: empty key set");
    Args p{q, k, v, out, lse2, q_lens, k_lens, ldq, ldk, ldv, ldo, batch, heads, m_max, n_max, scale * LOG2E, cdiv(m_max, BQ), 0};
    hipLaunchKernelGGL(attention_f16_kernel, dim3(batch * heads * p.q_tiles), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_f16_f32");
}

extern "C" int pram_attention_cross_f16_f32(const float* qk, int ldqk, const float* v, int ldv, float* out, int ldo, float* lse2,
                                            const int* lens, int pairs, int heads, int t_max, float scale, void* stream) {
    PRAM_REQUIRE(qk && v && out, "pram_attention_cross_f16_f32: null pointer");
    PRAM_REQUIRE(ldqk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "pram_attention_cross_f16_f32: ld must be a multiple of 4");
    PRAM_REQUIRE(pairs >= 0 && heads > 0 && t_max >= 0, "pram_attention_cross_f16_f32: bad sizes");
    if (pairs == 0 || t_max == 0) return PRAM_OK;
    Args p{qk, qk, v, out, lse2, lens, lens, ldqk, ldqk, ldv, ldo, 2 * pairs, heads, t_max, t_max, scale * LOG2E, cdiv(t_max, BQ), pairs};
    hipLaunchKernelGGL(attention_f16_kernel, dim3(2 * pairs * heads * p.q_tiles), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_cross_f16_f32");
}


/* fp16 q / k / v in HBM (written by pram_linear_f16_h16): ld* in halves, 16-byte aligned rows and head offsets. */
extern "C" int pram_attention_h16_f32(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv, float* out,
                                      int ldo, float* lse2, const int* q_lens, const int* k_lens, int batch, int heads,
                                      int m_max, int n_max, float scale, int kv_shift, void* stream) {
    PRAM_REQUIRE(q16 && k16 && v16 && out, "pram_attention_h16_f32: null pointer");
    PRAM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "pram_attention_h16_f32: ld of the fp16 operands must be a multiple of 8");
    PRAM_REQUIRE(batch >= 0 && heads > 0 && m_max >= 0 && n_max >= 0 && kv_shift >= 0, "pram_attention_h16_f32: bad sizes");
    if (batch == 0 || m_max == 0) return PRAM_OK;
    PRAM_REQUIRE(n_max > 0, "pram_attention_h16_f32: empty key set");
    ArgsH p{(const _Float16*)q16, (const _Float16*)k16, (const _Float16*)v16, out, lse2, q_lens, k_lens, ldq, ldk, ldv, ldo,
            batch, heads, m_max, n_max, scale * LOG2E, cdiv(m_max, BQ), kv_shift};
    hipLaunchKernelGGL(attention_h16_kernel, dim3(batch * heads * p.q_tiles), dim3(256), 0, (hipStream_t)stream, p);
    return pram_launch_status("pram_attention_h16_f32");
}
