// micro-benchmark: sustained v_mfma_i32_32x32x32_i8 rate with random vs zero operands (the int8 matrix pipe under the power cap),
// next to mfma_f16_chain_microbench.hip.  Register-only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const int* in, int* out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    i32x4 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) { a[j][i] = in[(t * 4 + i + 64 * j) & 0xffff]; b[j][i] = in[(t * 4 + i + 17 + 32 * j) & 0xffff]; }
    i32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) c[n][e] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(n + rep) & 3], b[(n + 2 * rep) & 3], c[n], 0, 0, 0);
    }
    int s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) s += c[n][e];
    out[t] = s;
    if (t == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }
}
template <int NACC>
void run(const int* in, int* out, int bpc, const char* tag) {
    const int blocks = 256 * bpc, iters = 4000;
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, 500, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 32;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%s chains/wave %d, waves/SIMD %d: %.3f ms, %.1f TOP/s (s_memtime: %.1f ticks per MFMA of one wave, %.0f MHz tick rate)\n", tag, NACC, bpc, ms,
           ops / ms / 1e9, (double)hc[0] / (iters * 8.0), (double)hc[0] / (double)hc[1] * 100.0);
}
int main() {
    int *in, *out;
    hipMalloc(&in, 65536 * 4 + 64); hipMalloc(&out, 256 * 2048 * 4);
    int* h = (int*)malloc(65536 * 4);
    for (int z = 0; z < 2; ++z) {
        for (int i = 0; i < 65536; ++i) h[i] = z ? 0 : (int)((unsigned)rand() * 2654435761u);
        hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
        const char* tag = z ? "zero  " : "random";
        for (int bpc = 1; bpc <= 2; ++bpc) { run<4>(in, out, bpc, tag); run<8>(in, out, bpc, tag); }
    }
    return 0;
}
