// Error channel and version of libpram_hip.so.
#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void pram_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int pram_hip_version(void) { return 110; }

// ---- status word (include/pram_hip.h, "range guard"): one caller-owned device word per device, ORed into by the kernels
static unsigned int* g_status[64] = {nullptr};

unsigned int* pram_status_ptr(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_status[dev];
}

// compute units of the current device (persistent kernels size their grids with it); 256 when the query fails
int pram_cu_count(void) {
    static int g_cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (g_cus[dev] == 0) {
        int n = 0;
        g_cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return g_cus[dev];
}

// ---- scale of the split-fp16 activation planes (include/pram_hip.h, "activation scale"): per host thread
static thread_local float g_act_scale = 16.0f;

float pram_act_scale(void) { return g_act_scale; }

extern "C" float pram_x3_set_act_scale(float scale) {
    const float prev = g_act_scale;
    if (scale > 0.f) {
        int e = 0;
        const float m = frexpf(scale, &e);      // a power of two in [2^-12, 2^4]: scaling stays exact, the planes stay inside fp16
        if (m == 0.5f && e - 1 >= -12 && e - 1 <= 4) g_act_scale = scale;
        else return -1.0f;
    }
    return prev;
}

extern "C" int pram_set_status_word(unsigned int* device_word) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        pram_set_error("pram_set_status_word: no current device");
        return PRAM_E_ARG;
    }
    g_status[dev] = device_word;
    return PRAM_OK;
}

extern "C" int pram_read_status_word(unsigned int* host_out, int reset, void* stream) {
    PRAM_REQUIRE(host_out, "pram_read_status_word: null pointer");
    unsigned int* w = pram_status_ptr();
    *host_out = 0u;
    if (!w) return PRAM_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(host_out, w, sizeof(unsigned int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return pram_launch_status("pram_read_status_word");
    if (reset && *host_out) {
        if (hipMemsetAsync(w, 0, sizeof(unsigned int), st) != hipSuccess) return pram_launch_status("pram_read_status_word");
    }
    return PRAM_OK;
}
extern "C" const char* pram_last_error(void) { return g_err; }
