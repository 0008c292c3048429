// micro-benchmark: do the two waves of a SIMD overlap when one issues only MFMAs and the other only vector instructions?
// One 512-thread workgroup per CU (waves w and w + 4 share SIMD w; 100 KB of LDS keep a second workgroup off the CU).  Roles by wave
// half: M = a stream of v_mfma_f32_32x32x16_f16 (two alternating accumulators), V = a stream of vector instructions (the soft-max mix:
// 1 v_exp_f32 : 1 v_fma_f32 : 1 v_fma_mixlo : 2 v_add per element), I = idle (exits at once).  Reported: shader clocks per MFMA /
// per vector instruction of wave 0 and wave 4, for (lower half, upper half) in MI, IM, VI, IV, MV, VM, MM, VV.
//   hipcc --offload-arch=gfx950 -O3 profiles/mfma_valu_two_waves_microbench.hip -o two && ./two
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void kern(const float* in, float* out, int iters, int role_lo, int role_hi, unsigned long long* clk) {
    __shared__ float pad[25600];
    const int t = blockIdx.x * 512 + threadIdx.x, wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;      // 0 idle, 1 MFMA, 2 VALU
    if (threadIdx.x == 0) pad[0] = in[0];
    __syncthreads();
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(t + i) & 1023]; b[i] = (_Float16)in[(t + 7 * i) & 1023]; }
    f32x16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    float r[10];
    for (int i = 0; i < 10; ++i) r[i] = in[(t + i) & 1023];
    const float y = in[(t + 3) & 1023], z = in[(t + 5) & 1023];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g)      // 5 instructions: the per-element mix of the attention soft-max
                asm volatile("v_fma_f32 %0, %5, %6, %0\n\tv_exp_f32 %1, %5\n\tv_fma_mixlo_f16 %2, %5, 1.0, -%6 op_sel_hi:[0,0,1]\n\tv_add_f32 %3, %5, %6\n\tv_add_f32 %4, %6, %5"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "v"(y), "v"(z));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = pad[0];
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    for (int i = 0; i < 10; ++i) s += r[i];
    out[t] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
}

int main() {
    float *d_in, *d_out; unsigned long long* d_clk;
    (void)hipMalloc(&d_in, 4096); (void)hipMalloc(&d_out, 256 * 512 * 4); (void)hipMalloc(&d_clk, 64);
    (void)hipMemset(d_in, 0, 4096);
    const char* names[3] = {"idle", "MFMA", "VALU"};
    const int iters = 2000;
    const int combos[8][2] = {{1, 0}, {0, 1}, {2, 0}, {0, 2}, {1, 2}, {2, 1}, {1, 1}, {2, 2}};
    for (auto& c : combos) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d_in, d_out, 200, c[0], c[1], d_clk);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d_in, d_out, iters, c[0], c[1], d_clk);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, d_clk, 64, hipMemcpyDeviceToHost);
        auto per = [&](int w, int role) { return role == 1 ? (double)h[w] / (iters * 8.0) : role == 2 ? (double)h[w] / (iters * 40.0) : 0.0; };
        printf("waves 0-3 %s, waves 4-7 %s:  wave 0 %6.2f clocks per %s   wave 4 %6.2f clocks per %s\n", names[c[0]], names[c[1]],
               per(0, c[0]), c[0] == 1 ? "MFMA" : "vector instruction", per(4, c[1]), c[1] == 1 ? "MFMA" : "vector instruction");
    }
    return 0;
}
