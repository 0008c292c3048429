// Exact-fp32 MFMA GEMM main loop for gfx950, shared by the token linear layers and the
// implicit-GEMM convolutions.
//
//   C[128 x 128] per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave, 2 x 2 tiles of
//   v_mfma_f32_32x32x2_f32), K consumed in chunks of 32 through a double-buffered LDS stage.
//
// Both operands are "K-contiguous rows" (activations [row][k], weights [col][k]), so A and B
// fragments use the same LDS image and the same read: lane l (r = l & 31, h = l >> 5) reads one
// b128 = 4 consecutive k at k-offset 8*kk + 4*h of row r; MFMA step j then contracts
// k = 8*kk + j (lower half-wave) and k = 8*kk + 4 + j (upper half-wave).  The operand maps
// (A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]) only require that A and B agree on which k a
// half-wave supplies, so one ds_read_b128 feeds four MFMAs.
//
// LDS rows are padded to 36 floats (144 B): the 16-B slot of row r is 9r + const (mod 16), a
// bijection over any 16 rows distinct mod 16, so every 16-lane group of ds_read_b128 is
// conflict-free, and the staging ds_write_b128 (8 lanes = one contiguous 128-B row) is too.
//
// f32 MFMA runs at the f32 vector rate (64 cycles per 32x32x2 per SIMD): the loop is
// matrix-pipe bound by a wide margin (64 MFMA = 4096 cycles per chunk per wave against
// 16 ds_read_b128 and 8 global float4 loads), so plain register staging with one barrier per
// chunk is enough; no LDS-DMA or counted-vmcnt pipeline is needed at this rate.
#pragma once
#include "common.h"

namespace gemm {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 36, NT = 256;

struct alignas(16) Smem {
    float a[2][BM * LDT];
    float b[2][BN * LDT];
};  // 73,728 B -> two workgroups per CU

// Branch-free guarded float4 load: the address is always legal (clamped by the caller), the value is
// zeroed when the element is out of range.  Branching around each load makes hipcc serialise them.
__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
    float4 v = *reinterpret_cast<const float4*>(p);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// ALoad: float4 operator()(int row_in_tile_slot p (0..3), int kt) -> A[row = tid/8 + 32p][kt*32 + (tid%8)*4 ..+3]
// BLoad: same for the weight rows.
template <class ALoad, class BLoad>
__device__ __forceinline__ void mainloop(Smem& s, ALoad& la, BLoad& lb, int nk, f32x16 (&acc)[2][2]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, skq = tid & 7;

#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    float4 ra[4], rb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ra[p] = la(p, 0);
        rb[p] = lb(p, 0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<float4*>(&s.a[0][(srow + 32 * p) * LDT + skq * 4]) = ra[p];
        *reinterpret_cast<float4*>(&s.b[0][(srow + 32 * p) * LDT + skq * 4]) = rb[p];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ra[p] = la(p, kt + 1);
                rb[p] = lb(p, kt + 1);
            }
        }
        const float* sa = &s.a[cur][(wm * 64 + r) * LDT + h * 4];
        const float* sb = &s.b[cur][(wn * 64 + r) * LDT + h * 4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 af[2], bf[2];
            af[0] = *reinterpret_cast<const float4*>(sa + kk * 8);
            af[1] = *reinterpret_cast<const float4*>(sa + 32 * LDT + kk * 8);
            bf[0] = *reinterpret_cast<const float4*>(sb + kk * 8);
            bf[1] = *reinterpret_cast<const float4*>(sb + 32 * LDT + kk * 8);
            const float a0[4] = {af[0].x, af[0].y, af[0].z, af[0].w};
            const float a1[4] = {af[1].x, af[1].y, af[1].z, af[1].w};
            const float b0[4] = {bf[0].x, bf[0].y, bf[0].z, bf[0].w};
            const float b1[4] = {bf[1].x, bf[1].y, bf[1].z, bf[1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<float4*>(&s.a[cur ^ 1][(srow + 32 * p) * LDT + skq * 4]) = ra[p];
                *reinterpret_cast<float4*>(&s.b[cur ^ 1][(srow + 32 * p) * LDT + skq * 4]) = rb[p];
            }
        }
        __syncthreads();
    }
}

// Accumulator element e of tile (mi, ni) of this lane lives at
//   row = 64*wm + 32*mi + (e & 3) + 8*(e >> 2) + 4*h ,  col = 64*wn + 32*ni + (lane & 31)
__device__ __forceinline__ int acc_row(int mi, int e, int h) { return 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h; }

}  // namespace gemm
