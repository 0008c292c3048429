// Which compute units does a hipExtStreamCreateWithCUMask stream run on?  For a few masks, 2048 one-wave workgroups record
// their XCC id (HW_REG_XCC_ID) and HW_ID (CU / SH / SE ids); printed: workgroups per XCC and distinct (xcc, se, sh, cu) places seen.
//   hipcc --offload-arch=gfx950 -O3 -w -o profiles/bin/cu_mask_probe profiles/cu_mask_probe.hip && profiles/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // burn a little time so that the grid spreads over every CU the stream may use
    float x = (float)threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
    if (x == 1.2345f) out[0] = 0;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 2 * 2048 * sizeof(uint32_t));
    struct M { const char* name; uint32_t w[8]; };
    M masks[] = {
        {"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"low128", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
        {"high128", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
        {"even", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"odd", {0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau}},
        {"first32", {~0u, 0, 0, 0, 0, 0, 0, 0}},
        {"bits0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"every8th", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"words0,2,4,6", {~0u, 0, ~0u, 0, ~0u, 0, ~0u, 0}},
    };
    for (auto& m : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
        if (e != hipSuccess) { printf("%s: create failed %d\n", m.name, (int)e); continue; }
        hipMemsetAsync(d, 0xff, 2 * 2048 * sizeof(uint32_t), st);
        hipLaunchKernelGGL(where, dim3(2048), dim3(64), 0, st, d);
        hipStreamSynchronize(st);
        std::vector<uint32_t> h(2 * 2048);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int per[16] = {0};
        std::set<uint64_t> places;
        for (int b = 0; b < 2048; ++b) {
            const uint32_t x = h[2 * b], hw = h[2 * b + 1];
            per[x & 15]++;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            places.insert(((uint64_t)x << 32) | (se << 8) | (sh << 4) | cu);
        }
        printf("%-13s: per XCC", m.name);
        for (int i = 0; i < 8; ++i) printf(" %4d", per[i]);
        printf("  | distinct CUs seen %zu\n", places.size());
        hipStreamDestroy(st);
    }
    return 0;
}
