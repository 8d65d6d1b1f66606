// What does the fp32 -> 3 x bf16 split cost on one SIMD of gfx950, alone and under MFMAs?  (round 6)
// One workgroup of 512 threads per CU (two waves per SIMD), as the split-fp32 conv kernels run.  Per iteration and wave: 24 v_mfma_f32_32x32x16_bf16
// (two independent accumulators, as the 64 x 32 wave tile) and / or NSPLIT = 4 splits of a f32x4 (one K step of the 128 x 128 kernel: RA + RB = 4 rows
// per loader thread) with the results XOR-folded into registers (no LDS, no memory: VALU only).
//   KIND 0: round-to-nearest split (v_cvt_pk_bf16_f32 + shift / and + v_pk_add_f32), the kernels' x3_stage
//   KIND 1: truncation split (v_and + v_pk_add_f32 + v_perm_b32) -- also an exact three-term decomposition
//   KIND 2..6: 72 independent instructions of ONE type (cvt_pk, and, perm, pk_add, lshl) -- the issue rate of each
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_valu_probe tools/debug/split_valu_probe.hip && /tmp/split_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_rn(const f32x4 v, unsigned (&o)[6]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const f32x2 r = x - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
        const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
        const f32x2 t = r - f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
        const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
        o[p] = h, o[2 + p] = m, o[4 + p] = l;
    }
}
__device__ __forceinline__ void split_tr(const f32x4 v, unsigned (&o)[6]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const unsigned x0 = __float_as_uint(v[2 * p]), x1 = __float_as_uint(v[2 * p + 1]);
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        const f32x2 r = x - f32x2{__uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 & 0xffff0000u)};
        const unsigned r0 = __float_as_uint(r[0]), r1 = __float_as_uint(r[1]);
        const f32x2 t = r - f32x2{__uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 & 0xffff0000u)};
        o[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        o[2 + p] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
        o[4 + p] = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
    }
}

template <int MODE, int KIND>  // MODE 1 MFMA only, 2 VALU only, 3 both (sched_group_barrier: 3 VALU behind every MFMA)
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc, int rep) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
    f32x4 v[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) v[i][e] = 1.0f + threadIdx.x * 1.37e-3f + i * 0.77f + e * 0.31f;
    unsigned fold[6] = {0, 0, 0, 0, 0, 0};
    unsigned w[24];
    for (int i = 0; i < 24; ++i) w[i] = threadIdx.x * 2654435761u + i;
    f32x2 pk[8];
    for (int i = 0; i < 8; ++i) pk[i] = f32x2{1.0f + i + threadIdx.x * 1e-3f, 0.5f + i};
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int r = 0; r < rep; ++r) {
        if (MODE & 2) {
            if (KIND <= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned o[6];
                    if (KIND == 0) split_rn(v[i], o);
                    else split_tr(v[i], o);
                    // keep the inputs changing (one VALU per row: a new "loaded" value) and the outputs alive
                    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]));
                    asm volatile("" : "+v"(v[i]));
                    fold[i & 3] = o[5];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 72; ++k) {
                    unsigned& x = w[k % 24];
                    if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x));
                    if (KIND == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
                    if (KIND == 4) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "s"(0x07060302u));
                    if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
                }
                if (KIND == 5) {
#pragma unroll
                    for (int k = 0; k < 72; ++k) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(pk[k % 8]));
                }
            }
        }
        if (MODE & 1) {
#pragma unroll
            for (int m = 0; m < 24; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 1], 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 4; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    for (int i = 0; i < 6; ++i) s += (float)fold[i];
    for (int i = 0; i < 24; ++i) s += (float)w[i];
    for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int KIND>
static void run(const char* what, float* out, long long* cyc, int rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, rep);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, rep);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s launch %7.1f us = %7.1f ns per iteration\n", what, ms * 1e3, ms * 1e6 / rep);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    const int rep = 4000;
    printf("# per iteration and SIMD: 2 waves x (24 MFMA, 4 f32x4 splits = 72 VALU)   [one K step of the 128 x 128 split-fp32 kernel]\n");
    run<1, 0>("MFMA only (48 per SIMD)", out, cyc, rep);
    run<2, 0>("split RN only (v_cvt_pk_bf16_f32 form, 144 VALU per SIMD)", out, cyc, rep);
    run<2, 1>("split TRUNC only (v_and / v_perm form, 144 VALU per SIMD)", out, cyc, rep);
    run<3, 0>("MFMA + split RN interleaved", out, cyc, rep);
    run<3, 1>("MFMA + split TRUNC interleaved", out, cyc, rep);
    run<2, 2>("144 x v_cvt_pk_bf16_f32", out, cyc, rep);
    run<2, 3>("144 x v_and_b32", out, cyc, rep);
    run<2, 4>("144 x v_perm_b32", out, cyc, rep);
    run<2, 5>("144 x v_pk_add_f32", out, cyc, rep);
    run<2, 6>("144 x v_lshlrev_b32", out, cyc, rep);
    run<3, 2>("MFMA + 144 x v_cvt_pk_bf16_f32", out, cyc, rep);
    run<3, 3>("MFMA + 144 x v_and_b32", out, cyc, rep);
    run<3, 4>("MFMA + 144 x v_perm_b32", out, cyc, rep);
    run<3, 5>("MFMA + 144 x v_pk_add_f32", out, cyc, rep);
    return 0;
}
