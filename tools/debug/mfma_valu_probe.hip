// Does VALU work overlap with MFMA work on one SIMD of gfx950?  One workgroup of 512 threads per CU (two waves per SIMD), each wave runs REP
// iterations of { NM x v_mfma_f32_32x32x16_bf16 (independent accumulators), NV x VALU (independent v_fma_f32 / v_cvt_pk_bf16_f32 chains) } with the
// two kinds interleaved one MFMA : NV / NM VALU.  Times the whole LAUNCH (hipEvents: every wave has finished), per iteration, for MFMA only, VALU
// only, both in every wave, and both with the waves SPECIALISED (waves 0-3, one per SIMD, run all the MFMAs; waves 4-7 all the VALU work).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_probe tools/debug/mfma_valu_probe.hip && /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int MODE_, int KIND>  // MODE 1 MFMA only, 2 VALU only, 3 both, 4 both with specialised waves;  KIND 0: v_fma_f32, 1: the split's mix
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc, int rep) {
    const int role = MODE_ == 4 ? ((threadIdx.x >> 6) < 4 ? 1 : 2) : MODE_;  // wave-uniform
    if (MODE_ == 4) rep *= 2;  // a specialised wave does its kind of work for two
    const int MODE = role;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE & 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (KIND == 0) {
                        v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
                    } else {
                        bf16x2 pk;
                        pk[0] = (__bf16)v[k], pk[1] = (__bf16)v[k + 1];
                        const unsigned h = __builtin_bit_cast(unsigned, pk);
                        v[k] = v[k] - __uint_as_float(h << 16) + 1.0f;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cyc[0] = c1 - c0;
        cyc[1] = t1 - t0;
    }
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
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s launch %7.1f us = %6.1f ns per iteration (16 MFMA + 96 VALU per SIMD and iteration)   wave 0 alone: %6.1f ns\n", what, ms * 1e3,
           ms * 1e6 / rep, h[1] * 10.0 / rep);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    const int rep = 4000;
    run<1, 0>("MFMA only", out, cyc, rep);
    run<2, 0>("VALU only (v_fma_f32)", out, cyc, rep);
    run<3, 0>("MFMA + v_fma_f32 interleaved in every wave", out, cyc, rep);
    run<4, 0>("MFMA waves + v_fma_f32 waves (specialised)", out, cyc, rep);
    run<2, 1>("VALU only (cvt_pk_bf16 + shift + sub + add)", out, cyc, rep);
    run<3, 1>("MFMA + the split's VALU mix in every wave", out, cyc, rep);
    run<4, 1>("MFMA waves + split-mix waves (specialised)", out, cyc, rep);
    return 0;
}
