// Shared helpers for the gfx950 kernels of libsdt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>

#include "../../include/sdt_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SDT_NXCD 8

void sdt_set_error(const char* fmt, ...);

#define SDT_CHECK_ARG(cond, msg)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            sdt_set_error("%s: %s", __func__, msg);      \
            return SDT_ERR_ARG;                          \
        }                                                \
    } while (0)

#define SDT_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            sdt_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return SDT_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bijective XCD-aware remap of a 1-D block id: blocks that the dispatcher places on one XCD
// (id % 8) get a contiguous chunk of the tile space, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk / SDT_NXCD, r = nblk % SDT_NXCD;
    int xcd = bid % SDT_NXCD, pos = bid / SDT_NXCD;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float act_fwd(float u, float slope) { return u > 0.f ? u : u * slope; }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.f ? 1.f : slope; }
