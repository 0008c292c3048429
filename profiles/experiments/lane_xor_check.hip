// Device check of lane_xor<K> (below): every lane must receive lane (l ^ K)'s value, in 256-thread blocks (four waves, so that
// threadIdx.x & 16 / & 32 are tested beyond wave 0).   hipcc --offload-arch=gfx950 -O3 -w -o profiles/bin/lane_xor_check profiles/experiments/lane_xor_check.hip
#include <hip/hip_runtime.h>
// lane (l ^ K)'s value without the LDS crossbar (round 6's experiment: profiles/r06_ssq_dpp_ab.txt — correct, and no faster where it was tried)
template <int K>
__device__ __forceinline__ float lane_xor(float v) {
    const int iv = __builtin_bit_cast(int, v);
    if constexpr (K == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    else if constexpr (K == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    else if constexpr (K == 4) {
        int r = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xF, 0x5, false);      // row_shl:4 into banks 0, 2
        r = __builtin_amdgcn_update_dpp(r, iv, 0x114, 0xF, 0xA, false);           // row_shr:4 into banks 1, 3
        return __builtin_bit_cast(float, r);
    } else if constexpr (K == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, false));   // row_ror:8
    else if constexpr (K == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)iv, (unsigned)iv, false, false);
        return __builtin_bit_cast(float, (int)((threadIdx.x & 16) ? r[0] : r[1]));
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)iv, (unsigned)iv, false, false);
        return __builtin_bit_cast(float, (int)((threadIdx.x & 32) ? r[0] : r[1]));
    }
}
#include <cstdio>
__global__ void chk(int* bad) {
    const float v = (float)(threadIdx.x & 63) + 100.f * (float)(threadIdx.x >> 6);
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto want = [&](int k) { return (float)(l ^ k) + 100.f * (float)w; };
    if (lane_xor<1>(v) != want(1)) atomicOr(bad, 1);
    if (lane_xor<2>(v) != want(2)) atomicOr(bad, 2);
    if (lane_xor<4>(v) != want(4)) atomicOr(bad, 4);
    if (lane_xor<8>(v) != want(8)) atomicOr(bad, 8);
    if (lane_xor<16>(v) != want(16)) atomicOr(bad, 16);
    if (lane_xor<32>(v) != want(32)) atomicOr(bad, 32);
}
int main() {
    int* d; int h = -1;
    hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(chk, dim3(4), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("lane_xor check: bad mask = %d (0 = every K maps lane l to l ^ K)\n", h);
    return h != 0;
}
