// Probe of ds_read_b64_tr_b16 on gfx950 (round 4, design input for the bf16 weight-gradient kernel):
//   1. semantics: which (lane, element) of the 16-lane group's 16 x 8-byte reads lands in which (lane, element) of the result;
//   2. a 32x32x16 bf16 MFMA whose A and B operands are transposed-read from m-major LDS tiles, against a host reference;
//   3. cycles per instruction of the fragment read pattern for three LDS layouts (plain 256-byte pitch, 64-byte XOR swizzle, 320-byte pitch).
// Build: hipcc --offload-arch=gfx950 -O3 tools/debug/tr16_probe.hip -o tools/bin/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr16(const short* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p); }

__global__ void sem_kernel(short* out) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = tr16(lds + threadIdx.x * 4);  // lane l addresses elements 4l .. 4l+3
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

// layout: 0 plain pitch (cols*2 bytes), 1 XOR of the 64-byte segment index with (row & 3), 2 pitch + 64 bytes
template <int LAYOUT>
__device__ __forceinline__ int toff(int row, int col, int cols) {  // element offset of (row, col) in an m-major [rows][cols] tile
    if (LAYOUT == 0) return row * cols + col;
    if (LAYOUT == 1) return row * cols + ((((col >> 5) ^ (row & 3)) << 5) | (col & 31));
    return row * (cols + 32) + col;
}

// D[n][c] = sum_m A[m][n] * B[m][c], m = 0..15, n, c = 0..31: both operands m-major in LDS, fragments by transposed reads
template <int LAYOUT>
__global__ void mfma_kernel(const short* A, const short* B, float* D) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    constexpr int COLS = 128;
    short* sA = lds;
    short* sB = lds + 64 * (COLS + 32);
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 32; i += 64) {
        const int m = i / 32, n = i % 32;
        sA[toff<LAYOUT>(m, n, COLS)] = A[i];
        sB[toff<LAYOUT>(m, n, COLS)] = B[i];
    }
    __syncthreads();
    const int g = lane >> 4, p = lane & 15;
    const int col = (g & 1) * 16 + 4 * (p & 3), row = 8 * (g >> 1) + (p >> 2);
    s16x4 a0 = tr16(sA + toff<LAYOUT>(row, col, COLS)), a1 = tr16(sA + toff<LAYOUT>(row + 4, col, COLS));
    s16x4 b0 = tr16(sB + toff<LAYOUT>(row, col, COLS)), b1 = tr16(sB + toff<LAYOUT>(row + 4, col, COLS));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    s16x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), c = lane & 31;
        D[n * 32 + c] = acc[r];
    }
}

// timing: one workgroup of 4 waves, each wave reads the fragments of a 64 x 64 wave tile (2 + 2 sub-tiles) for 4 k16 blocks of a
// [64][128] m-major tile pair, ITER times
template <int LAYOUT>
__global__ void time_kernel(long long* cycles, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    constexpr int COLS = 128;
    short* sA = lds;
    short* sB = lds + 64 * (COLS + 32);
    for (int i = threadIdx.x; i < 2 * 64 * (COLS + 32); i += 256) lds[i] = (short)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, p = lane & 15;
    const int col = (g & 1) * 16 + 4 * (p & 3), row = 8 * (g >> 1) + (p >> 2);
    int s = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s16x4 a = tr16(sA + toff<LAYOUT>(16 * j + 4 * h + row, wm * 64 + t * 32 + col, COLS));
                    s16x4 b = tr16(sB + toff<LAYOUT>(16 * j + 4 * h + row, wn * 64 + t * 32 + col, COLS));
                    s += a[0] + a[3] + b[1] + b[2];
                }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    sink[threadIdx.x] = (float)s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int LAYOUT>
static void run_layout(const short* dA, const short* dB, float* dD, const std::vector<float>& ref, long long* dcyc, float* dsink) {
    const int lds_bytes = 2 * 64 * (128 + 32) * 2;
    hipLaunchKernelGGL(mfma_kernel<LAYOUT>, dim3(1), dim3(64), lds_bytes, 0, dA, dB, dD);
    std::vector<float> D(32 * 32);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < 32 * 32; ++i) worst = std::max(worst, (double)std::fabs(D[i] - ref[i]));
    const int iters = 2000;
    hipLaunchKernelGGL(time_kernel<LAYOUT>, dim3(1), dim3(256), lds_bytes, 0, dcyc, dsink, iters);
    hipLaunchKernelGGL(time_kernel<LAYOUT>, dim3(1), dim3(256), lds_bytes, 0, dcyc, dsink, iters);
    long long cyc;
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    // 32 tr reads per wave and iteration, 4 waves on one CU
    printf("layout %d: mfma max|err| %.3g   %.2f cycles per tr16 wave-instruction (4 waves, 32 reads per iteration)\n", LAYOUT, worst,
           (double)cyc / iters / 32.0);
}

int main() {
    short* dout;
    CK(hipMalloc(&dout, 256 * 2));
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 8192, 0, dout);
    std::vector<short> out(256);
    CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("semantics (lane l addresses elements 4l..4l+3; value = element index): out[lane][j]\n");
    for (int l = 0; l < 64; ++l) {
        printf("  lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" %4d (lane %2d e%d)", out[l * 4 + j], out[l * 4 + j] / 4, out[l * 4 + j] % 4);
        printf("\n");
    }
    std::vector<unsigned short> A(16 * 32), B(16 * 32);
    std::vector<float> ref(32 * 32, 0.f);
    srand(1);
    for (auto& v : A) v = f2bf((rand() % 2001 - 1000) / 500.f);
    for (auto& v : B) v = f2bf((rand() % 2001 - 1000) / 500.f);
    for (int n = 0; n < 32; ++n)
        for (int c = 0; c < 32; ++c) {
            double s = 0;
            for (int m = 0; m < 16; ++m) s += (double)bf2f(A[m * 32 + n]) * bf2f(B[m * 32 + c]);
            ref[n * 32 + c] = (float)s;
        }
    short *dA, *dB;
    float *dD, *dsink;
    long long* dcyc;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096)); CK(hipMalloc(&dsink, 1024)); CK(hipMalloc(&dcyc, 8));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    run_layout<0>(dA, dB, dD, ref, dcyc, dsink);
    run_layout<1>(dA, dB, dD, ref, dcyc, dsink);
    run_layout<2>(dA, dB, dD, ref, dcyc, dsink);
    return 0;
}
