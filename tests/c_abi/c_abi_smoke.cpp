// Stand-alone consumer of the C ABI (include/pram_hip.h): no Python, no torch — plain HIP runtime calls, raw device
// pointers, the library's entry points, and a CPU check of what comes back.  Built and run by
// tests/test_gpu_c_abi.py:   hipcc c_abi_smoke.cpp -I include -L pram_amd/csrc -lpram_hip -o c_abi_smoke
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "pram_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define PRAM_OK_(x) do { int rc_ = (x); if (rc_ != PRAM_OK) { printf("pram error %d: %s\n", rc_, pram_last_error()); return 3; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main() {
    printf("libpram_hip version %d\n", pram_hip_version());
    unsigned seed = 12345u;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // ---- pram_linear_f32: out = x @ w^T + b   (m = 300 rows, k = 64, n = 96: ragged tiles on purpose)
    const int m = 300, k = 64, n = 96;
    std::vector<float> x(m * k), w(n * k), b(n), out(m * n);
    for (auto& v : x) v = frand(seed);
    for (auto& v : w) v = frand(seed);
    for (auto& v : b) v = frand(seed);
    float *dx, *dw, *db, *dout;
    HIP_OK(hipMalloc(&dx, x.size() * 4)); HIP_OK(hipMalloc(&dw, w.size() * 4)); HIP_OK(hipMalloc(&db, b.size() * 4)); HIP_OK(hipMalloc(&dout, out.size() * 4));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    PRAM_OK_(pram_linear_f32(dx, k, k, nullptr, 0, 0, dw, db, nullptr, 0, dout, n, m, n, 1.0f, 0, nullptr, nullptr, 0, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = b[j];
            for (int t = 0; t < k; ++t) acc += (double)x[i * k + t] * w[j * k + t];
            worst = fmax(worst, fabs(acc - out[i * n + j]));
        }
    printf("pram_linear_f32     max |err| vs fp64 = %.3e\n", worst);
    if (!(worst < 1e-5)) return 4;

    // ---- pram_linear_x3_f32 (the default precision): the caller splits the weight matrix into two fp16 planes on the host
    {
        float wmax = 0.f;
        for (float e : w) wmax = fmaxf(wmax, fabsf(e));
        const float w_scale = exp2f(floorf(log2f(16384.0f / wmax)));          // max|w| * w_scale in [2^13, 2^14)
        std::vector<_Float16> whi(w.size()), wlo(w.size());
        for (size_t i = 0; i < w.size(); ++i) {
            const float sv = w[i] * w_scale;
            whi[i] = (_Float16)sv;
            wlo[i] = (_Float16)(sv - (float)whi[i]);
        }
        void *dwh, *dwl;
        HIP_OK(hipMalloc(&dwh, whi.size() * 2)); HIP_OK(hipMalloc(&dwl, wlo.size() * 2));
        HIP_OK(hipMemcpy(dwh, whi.data(), whi.size() * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dwl, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMemset(dout, 0, out.size() * 4));
        PRAM_OK_(pram_linear_x3_f32(dx, k, k, nullptr, 0, 0, dwh, dwl, w_scale, db, nullptr, 0, dout, n, nullptr, nullptr, 0, m, n,
                                    1.0f, 0, nullptr, nullptr, 0, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double worst3 = 0.0;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = b[j];
                for (int t = 0; t < k; ++t) acc += (double)x[i * k + t] * w[j * k + t];
                worst3 = fmax(worst3, fabs(acc - out[i * n + j]));
            }
        printf("pram_linear_x3_f32  max |err| vs fp64 = %.3e\n", worst3);
        if (!(worst3 < 1e-5)) return 9;
    }

    // ---- pram_attention_f32: 2 sequences x 4 heads x 64, ragged lengths, with and without the split workspace
    const int B = 2, H = 4, T = 200, C = H * 64;
    const int lens_h[2] = {200, 77};
    std::vector<float> q(B * T * C), kk(B * T * C), v(B * T * C), o(B * T * C), o2(B * T * C);
    for (auto& e : q) e = frand(seed);
    for (auto& e : kk) e = frand(seed);
    for (auto& e : v) e = frand(seed);
    float *dq, *dk, *dv, *dob; int* dl;
    HIP_OK(hipMalloc(&dq, q.size() * 4)); HIP_OK(hipMalloc(&dk, q.size() * 4)); HIP_OK(hipMalloc(&dv, q.size() * 4)); HIP_OK(hipMalloc(&dob, q.size() * 4));
    HIP_OK(hipMalloc(&dl, sizeof(lens_h)));
    HIP_OK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dk, kk.data(), q.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dv, v.data(), q.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dl, lens_h, sizeof(lens_h), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dob, 0, q.size() * 4));
    PRAM_OK_(pram_attention_f32(dq, C, dk, C, dv, C, dob, C, nullptr, dl, dl, B, H, T, T, 0.125f, nullptr, 0, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(o.data(), dob, q.size() * 4, hipMemcpyDeviceToHost));
    worst = 0.0;
    std::vector<double> p(T);
    for (int bb = 0; bb < B; ++bb)
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < lens_h[bb]; ++i) {
                double mx = -1e300, sum = 0.0;
                for (int j = 0; j < lens_h[bb]; ++j) {
                    double s = 0.0;
                    for (int d = 0; d < 64; ++d) s += (double)q[(bb * T + i) * C + h * 64 + d] * kk[(bb * T + j) * C + h * 64 + d];
                    p[j] = s * 0.125;
                    mx = fmax(mx, p[j]);
                }
                for (int j = 0; j < lens_h[bb]; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
                for (int d = 0; d < 64; ++d) {
                    double acc = 0.0;
                    for (int j = 0; j < lens_h[bb]; ++j) acc += p[j] * v[(bb * T + j) * C + h * 64 + d];
                    worst = fmax(worst, fabs(acc / sum - o[(bb * T + i) * C + h * 64 + d]));
                }
            }
    printf("pram_attention_f32  max |err| vs fp64 = %.3e\n", worst);
    if (!(worst < 2e-5)) return 5;

    // a caller-owned workspace switches small launches to the split mode; the bits must not change
    const int T2 = 1200;
    std::vector<float> big(T2 * C);
    for (auto& e : big) e = frand(seed);
    float *dq2, *do2a, *do2b; void* ws;
    HIP_OK(hipMalloc(&dq2, big.size() * 4)); HIP_OK(hipMalloc(&do2a, big.size() * 4)); HIP_OK(hipMalloc(&do2b, big.size() * 4));
    HIP_OK(hipMemcpy(dq2, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    const size_t nb = pram_attention_workspace_bytes(1, H, T2, T2);
    if (nb == 0) { printf("expected a split workspace size for a one-sequence launch\n"); return 6; }
    HIP_OK(hipMalloc(&ws, nb));
    PRAM_OK_(pram_attention_f32(dq2, C, dq2, C, dq2, C, do2a, C, nullptr, nullptr, nullptr, 1, H, T2, T2, 0.125f, nullptr, 0, st));
    PRAM_OK_(pram_attention_f32(dq2, C, dq2, C, dq2, C, do2b, C, nullptr, nullptr, nullptr, 1, H, T2, T2, 0.125f, ws, nb, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> a(big.size()), bsplit(big.size());
    HIP_OK(hipMemcpy(a.data(), do2a, big.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bsplit.data(), do2b, big.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i] != bsplit[i]) { printf("fused and split attention differ at %zu: %.9g vs %.9g\n", i, a[i], bsplit[i]); return 7; }
    printf("pram_attention_f32  fused == split, bit for bit (%zu bytes of workspace)\n", nb);

    // ---- error path: bad arguments come back as a code + message, nothing is launched
    const int rc = pram_linear_f32(dx, k, 63, nullptr, 0, 0, dw, db, nullptr, 0, dout, n, m, n, 1.0f, 0, nullptr, nullptr, 0, st);
    if (rc == PRAM_OK) { printf("expected an argument error for K = 63\n"); return 8; }
    printf("argument check: rc = %d, \"%s\"\n", rc, pram_last_error());
    printf("c_abi_smoke ok\n");
    return 0;
}
