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

extern "C" int pram_hip_version(void) { return 100; }
extern "C" const char* pram_last_error(void) { return g_err; }
