// Shared device/host helpers for the gfx950 SGCLIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sgclip.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

namespace sgclip {

// thread-local error text behind sgclip_last_error()
void set_error(const char *fmt, ...);
int launch_check(const char *what);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for `fn` on the CURRENT device, once per (kernel, device): the
// attribute belongs to the device's own function object, so a second GPU in the process needs its own call.
// false + error text when HIP refuses.
bool lds_attr(const void *fn, int bytes);
// compute units of the current device (cached per device)
int device_cus();

// Opt-in per-kernel timing (sgclip_timing_*): when enabled, a KernelTimer in a launch
// function brackets the launch with two HIP events on the launch stream.
enum { TK_GEMM0 = 0, TK_CROP = 8, TK_ATTN = 9, TK_LN = 10, TK_SIM = 11, TK_COUNT = 12 };
struct KernelTimer {
    KernelTimer(int kind, double work, hipStream_t stream);
    ~KernelTimer();
    int slot;
    hipStream_t stream;
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

}  // namespace sgclip
