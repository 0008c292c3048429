#include <hip/hip_runtime.h>
#include <cstdio>
#include <math.h>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, float mean, float rstd) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    float v[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) v[i] = in[(t + i * 977) & 0xffff];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 128; ++i) {
        const float y = (v[i] - mean) * rstd * 1.01f + 0.02f;
        s += 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
    out[t] = s;
}
int main() {
    float *in, *out; hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 6.f - 3.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, 0.1f, 1.3f);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, 0.1f, 1.3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LN-normalise + erf GELU of 128 values per lane, one 512-thread workgroup per CU: %.1f us per launch\n", ms / 20 * 1e3);
    return 0;
}
