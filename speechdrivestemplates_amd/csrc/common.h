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

// ---- exact three-way bf16 split of an fp32 value (presplit.hip): x = x1 + x2 + x3, each piece 8 significand bits, by truncation
// (x1 = top 8 bits of x; r = x - x1 is exact and has <= 16 significant bits; x2 = top 8 bits of r; r - x2 has <= 8 bits left)
typedef unsigned sdt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sdt_trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned sdt_pack_hi16(float lo, float hi) {  // {bf16 trunc(lo), bf16 trunc(hi)} in one dword
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
    const float x1 = sdt_trunc_bf16(x), r = x - x1;
    const float x2 = sdt_trunc_bf16(r), r2 = r - x2;
    a = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(x1) >> 16));
    b = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(x2) >> 16));
    c = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r2) >> 16));
}
// "Planes" layout (presplit.hip): the three bf16 pieces of a channels-last tensor (rows, C), C % 32 == 0, are interleaved per row
// and per 32-channel chunk: element (row, c, piece p) lives at  row*3C + (c/32)*96 + p*32 + (c%32).  A K step of the conv kernel
// (one 32-channel chunk of a row) then reads 192 CONTIGUOUS bytes -- with one plane-major array per piece it read three 64-byte
// half-lines from three distant addresses, wasting half of every 128-byte L2 line it touched.
__device__ __forceinline__ size_t planes_index(size_t row, int c, int C) { return row * (size_t)(3 * C) + (size_t)((c >> 5) * 96 + (c & 31)); }
// four consecutive channels c..c+3 (c % 4 == 0) of one row -> the three pieces (8-byte store each); dst = planes + planes_index(row, c, C)
__device__ __forceinline__ void store_planes4(const f32x4 v, __bf16* __restrict__ dst) {
    f32x4 r = v;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        sdt_u32x2 w;
        w[0] = sdt_pack_hi16(r[0], r[1]);
        w[1] = sdt_pack_hi16(r[2], r[3]);
        *(sdt_u32x2*)(dst + p * 32) = w;
        if (p < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] - sdt_trunc_bf16(r[e]);
        }
    }
}
