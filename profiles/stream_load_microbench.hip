// micro-benchmark: HBM read bandwidth of a kernel shaped like the wide GEMM's activation stream — ONE 512-thread workgroup per CU
// (two waves per SIMD, what 232 VGPRs allow), each thread keeping K independent 16-byte loads in flight, rows of 128 bytes
// (32 fp32) at a stride of `ld` floats like the A operand's 32-column chunks.  No arithmetic.  Answers: how many bytes must a CU
// keep in flight to stream at the rate a copy kernel reaches with full occupancy?
#include <hip/hip_runtime.h>
#include <cstdio>
// shared != 0: every workgroup walks the SAME rows (the weight operand: L2 hits after the first touch) `shared` times over
template <int K>
__global__ __launch_bounds__(512) void k(const float* __restrict__ a, float* out, int rows_per_wg, int ld, int chunks, int shared = 0) {
    const int tid = threadIdx.x, row = tid >> 3, q = tid & 7;
    const float* base = a + (shared ? 0 : (size_t)blockIdx.x * rows_per_wg * ld);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rep = 0; rep < (shared ? shared : 1); ++rep)
    // walk the tile's rows in 64-row passes, `chunks` 32-column chunks per row block, K loads in flight per thread
    for (int r0 = 0; r0 < rows_per_wg; r0 += 64 * K) {
        for (int c = 0; c < chunks; ++c) {
            float4 v[K];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = *reinterpret_cast<const float4*>(base + (size_t)(r0 + row + 64 * i) * ld + c * 32 + q * 4);
#pragma unroll
            for (int i = 0; i < K; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    if (acc.x == 12345.678f) out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
}
template <int K>
void run(const float* a, float* out, int ld) {
    const int wgs = 256, rows_per_wg = 64 * 16 * 4, chunks = ld / 32;          // 4096 rows per workgroup
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<K>, dim3(wgs), dim3(512), 0, 0, a, out, rows_per_wg, ld, chunks);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<K>, dim3(wgs), dim3(512), 0, 0, a, out, rows_per_wg, ld, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 10.0 * wgs * rows_per_wg * (double)ld * 4;
    printf("ld %4d floats, %2d x 16-byte loads in flight per thread (%3d KB per CU): %6.2f TB/s\n", ld, K, K * 8, bytes / ms / 1e9);
}
template <int K>
void run_shared(const float* a, float* out) {
    // 1024 rows x 256 floats = 1 MB (a weight matrix), read 64 times by each of the 256 workgroups
    const int wgs = 256, rows = 1024, ld = 256, reps = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<K>, dim3(wgs), dim3(512), 0, 0, a, out, rows, ld, ld / 32, reps);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<K>, dim3(wgs), dim3(512), 0, 0, a, out, rows, ld, ld / 32, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 10.0 * wgs * reps * rows * (double)ld * 4;
    printf("L2-resident 1 MB read by every workgroup, %2d loads in flight per thread: %6.2f TB/s chip-wide = %5.1f bytes per clock and CU at 2.1 GHz\n",
           K, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}
int main() {
    const size_t n = (size_t)256 * 4096 * 512;        // 2 GB of fp32 at ld = 512
    float *a, *out; hipMalloc(&a, n * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(a, 0, n * 4);
    for (int ld : {256, 512}) { run<1>(a, out, ld); run<2>(a, out, ld); run<4>(a, out, ld); run<8>(a, out, ld); run<16>(a, out, ld); }
    run_shared<1>(a, out); run_shared<2>(a, out); run_shared<4>(a, out); run_shared<8>(a, out);
    return 0;
}
