// What bounds a workgroup's output burst?  One 512-thread workgroup per CU (128 KB of LDS claimed so that two never share one) writes
// 256 KB tiles the way the GEMM epilogue does, G workgroups at a time (G = 32 .. 256 active CUs), in four store forms:
//   0  dword  : a wave instruction writes 2 rows x 128 B (column-per-lane accumulator layout, the shipped epilogue)
//   1  dwordx4: a wave instruction writes 4 rows x 256 B (16 lanes per row: the "rows through LDS" form, without the LDS pass)
//   2  dwordx4: a wave instruction writes 32 rows x 32 B (row-per-lane, operand-swapped MFMA layout)
//   3  dwordx4: a wave instruction writes 1 KB contiguous (the ceiling of the store path)
// Per form and G: microseconds per 256 KB tile and bytes per clock per CU (clock from the wall_clock64 / s_memtime ratio is not
// needed: the shader clock is read with s_memtime around the burst of workgroup 0's wave 0).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/bin/store_path profiles/store_path_microbench.hip && profiles/bin/store_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int TILE_ROWS = 256, TILE_COLS = 256, REPS = 32;      // fp32 tile, REPS tiles per workgroup (distinct addresses)

template <int FORM>
__global__ __launch_bounds__(512) void store_kernel(float* out, int ld, unsigned long long* clk) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;      // 2 x 4 waves, 128 x 64 outputs each
    if (tid == 0) smem[0] = 1;
    const float v = (float)tid;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REPS; ++rep) {
        float* tile = out + ((size_t)(blockIdx.x * REPS + rep) * TILE_ROWS + wm * 128) * ld + wn * 64;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float* blk = tile + (size_t)mi * 32 * ld;
            if constexpr (FORM == 0) {
                const int r = lane & 31, h = lane >> 5;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    blk[(size_t)row * ld + r] = v;
                    blk[(size_t)row * ld + 32 + r] = v;
                }
            } else if constexpr (FORM == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<float4*>(blk + (size_t)((lane >> 4) + 4 * j) * ld + (lane & 15) * 4) = make_float4(v, v, v, v);
            } else if constexpr (FORM == 2) {
                const int r = lane & 31, h = lane >> 5;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(blk + (size_t)r * ld + ni * 32 + 8 * g + 4 * h) = make_float4(v, v, v, v);
            } else {
                float* lin = out + ((size_t)(blockIdx.x * REPS + rep) * TILE_ROWS * TILE_COLS) + (wave * 4 + mi) * 2048;
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(lin + j * 256 + lane * 4) = make_float4(v, v, v, v);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int FORM>
void run(float* buf, unsigned long long* clk, int G, int ld) {
    hipFuncSetAttribute((const void*)store_kernel<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<FORM>, dim3(G), dim3(512), 128 * 1024, 0, buf, ld, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<FORM>, dim3(G), dim3(512), 128 * 1024, 0, buf, ld, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), clk, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cs = 0;
    for (auto c : h) cs += (double)c;
    cs /= G;
    const double bytes = (double)REPS * TILE_ROWS * TILE_COLS * 4;
    printf("form %d  G %3d : %7.2f us per 256 KB tile (wall / launch / REPS), %6.2f B/clk/CU (shader clocks of the burst), chip %.2f TB/s\n", FORM, G,
           ms / 10 * 1e3 / REPS, bytes / cs, bytes * G / (ms / 10 * 1e-3) / 1e12);
}

int main() {
    const int ld = 768;      // the step's widest fp32 output
    float* buf;
    unsigned long long* clk;
    const size_t n = (size_t)256 * REPS * TILE_ROWS * 768;
    hipMalloc(&buf, n * sizeof(float));
    hipMalloc(&clk, 256 * sizeof(unsigned long long));
    for (int G : {32, 64, 128, 256}) {
        run<0>(buf, clk, G, ld);
        run<1>(buf, clk, G, ld);
        run<2>(buf, clk, G, ld);
        run<3>(buf, clk, G, ld);
    }
    return 0;
}
