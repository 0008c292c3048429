// Shared helpers for the gfx950 kernels of libpram_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pram_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void pram_set_error(const char* fmt, ...);

#define PRAM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            pram_set_error(__VA_ARGS__);   \
            return PRAM_E_ARG;             \
        }                                  \
    } while (0)

static inline int pram_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pram_set_error("%s: %s", what, hipGetErrorString(e));
        return PRAM_E_LAUNCH;
    }
    return PRAM_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Behaviour-changing environment switches (tile / kernel-form overrides) exist in profiling builds only (-DPRAM_PROFILING, e.g.
// profiles/tools/build_variants.py TAG:file.hip:-DPRAM_PROFILING): the production library reads none of them.
#include <stdlib.h>
static inline const char* prof_env(const char* name) {
#ifdef PRAM_PROFILING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// The caller-owned status word of the current device (pram_set_status_word), or nullptr.
unsigned int* pram_status_ptr(void);
// compute units of the current device (grid size of the persistent kernels)
int pram_cu_count(void);
// Scale of the split-fp16 ACTIVATION planes in force for the calling host thread (pram_x3_set_act_scale; default 16 =
// gemmx3::ACT_SCALE): every kernel that writes or reads activation planes gets it by value at launch, so producers and consumers
// launched under one setting agree, and a captured graph keeps the value it was captured with.
float pram_act_scale(void);

// Range guard of the split-fp16 path: a kernel that turns fp32 values into fp16(value * scale) parts tracks the largest
// |value * scale| it met; 65520 and above round to +-inf in fp16 (a finite result would be garbage, usually NaN), which is
// reported in the status word instead of silently.  NaN inputs never compare >= and are not flagged: they propagate as NaN,
// like the reference's fp32 arithmetic.
__device__ __forceinline__ void x3_range_flag(unsigned int* status, float amax_scaled) {
    // one atomic per wave at most: the flag is rare, the test is not
    if (status != nullptr && amax_scaled >= 65520.0f) atomicOr(status, PRAM_STATUS_X3_RANGE);
}


// Bijective XCD-aware remap of a 1-D block id: hardware places block b on XCD b % 8; after the
// remap consecutive logical ids share an XCD (and therefore its L2).  Placement is used for
// speed only, never for correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
