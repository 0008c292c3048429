// micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate as a function of (a) the number of independent accumulator chains per
// wave, (b) waves per SIMD, (c) random vs zero operands (DVFS), (d) whether the three split-product terms share an accumulator.
// Register-only: no LDS, no global traffic inside the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const _Float16* in, float* out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    half8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { a[j][i] = in[(t * 8 + i + 64 * j) & 0xffff]; b[j][i] = in[(t * 8 + i + 17 + 32 * j) & 0xffff]; }
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) c[n][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(n + rep) & 3], b[(n + 2 * rep) & 3], c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) s += c[n][e];
    out[t] = s;
    if (t == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }      // s_memtime ticks, 100 MHz ticks
}
template <int NACC>
void run(const _Float16* in, float* out, int bpc, const char* tag) {
    const int blocks = 256 * bpc, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, 500, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * 32768.0;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    // workgroup 0's own view: s_memtime ticks per MFMA it issued, and the tick rate against the 100 MHz s_memrealtime
    printf("%s chains/wave %d, waves/SIMD %d: %.3f ms, %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz; s_memtime: %.1f ticks per MFMA of one wave, %.0f MHz tick rate)\n",
           tag, NACC, bpc, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)bpc * iters * 8), (double)hc[0] / (iters * 8.0), (double)hc[0] / (double)hc[1] * 100.0);
}
int main() {
    _Float16 *in; float *out;
    hipMalloc(&in, 65536 * 2 + 64); hipMalloc(&out, 256 * 2048 * 4);
    _Float16* h = (_Float16*)malloc(65536 * 2);
    // z = 0 random, 1 zeros, 2 / 3 / 4: random with the low 3 / 6 / 9 mantissa bits cleared (does operand entropy set the power?)
    for (int z = 0; z < 5; ++z) {
        for (int i = 0; i < 65536; ++i) {
            _Float16 v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
            if (z == 1) v = (_Float16)0.f;
            if (z >= 2) { unsigned short b; __builtin_memcpy(&b, &v, 2); b &= (unsigned short)(0xffff << (3 * (z - 1))); __builtin_memcpy(&v, &b, 2); }
            h[i] = v;
        }
        hipMemcpy(in, h, 65536 * 2, hipMemcpyHostToDevice);
        const char* tags[5] = {"random", "zero  ", "rnd-3b", "rnd-6b", "rnd-9b"};
        const char* tag = tags[z];
        for (int bpc = 1; bpc <= 2; ++bpc) {
            if (z >= 2 && bpc == 1) continue;
            if (z < 2) { run<1>(in, out, bpc, tag); run<2>(in, out, bpc, tag); }
            run<4>(in, out, bpc, tag); run<8>(in, out, bpc, tag);
        }
    }
    return 0;
}
