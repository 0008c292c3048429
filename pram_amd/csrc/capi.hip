// Error channel and version of libpram_hip.so.
#include <stdarg.h>
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
