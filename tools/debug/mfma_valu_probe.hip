// Does VALU work overlap with MFMA work on one SIMD of gfx950?  One workgroup of 512 threads per CU (two waves per SIMD), each wave runs REP
// iterations of { NM x v_mfma_f32_32x32x16_bf16 (independent accumulators), NV x VALU (independent v_fma_f32 / v_cvt_pk_bf16_f32 chains) } with the
// two kinds interleaved one MFMA : NV / NM VALU.  Prints cycles per iteration for MFMA only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_probe tools/debug/mfma_valu_probe.hip && /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int MODE, int KIND>  // MODE 1 MFMA only, 2 VALU only, 3 both;  KIND 0: v_fma_f32, 1: v_cvt_pk_bf16_f32 + shifts + subs (the split's mix)
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc, int rep) {
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
    hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, rep);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, rep);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s %8.1f shader cycles per iteration (8 MFMA = 256 pipe cycles per wave, 2 waves per SIMD; 48 VALU per wave)   wall %.2f us total\n", what,
           (double)h[0] / rep, h[1] * 0.01);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    const int rep = 2000;
    run<1, 0>("MFMA only", out, cyc, rep);
    run<2, 0>("VALU only (v_fma_f32)", out, cyc, rep);
    run<3, 0>("MFMA + v_fma_f32 interleaved", out, cyc, rep);
    run<2, 1>("VALU only (cvt_pk_bf16 + shift + sub + add)", out, cyc, rep);
    run<3, 1>("MFMA + the split's VALU mix interleaved", out, cyc, rep);
    return 0;
}
