// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate with random (non-zero) operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a0 = in[t], a1 = in[t + 1], b0 = in[t + 2], b1 = in[t + 3];
    f32x16 c0, c1, c2, c3;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[t] = s;
}
int main() {
    const int blocks_per_cu[2] = {1, 2};
    float *in, *out;
    const int n = 256 * 2048 + 8;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    for (int v = 0; v < 2; ++v) {
        const int blocks = 256 * blocks_per_cu[v], iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 4096.0;
        printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s\n", blocks_per_cu[v], ms, flops / ms / 1e9);
    }
    return 0;
}
