// Micro-benchmark: what the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) of one MI355X sustains under the instruction mixes of
// conv_taps_kernel's K loop, WITHOUT the per-workgroup prologue / epilogue and tile quantisation of a real launch.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
//
// Persistent grids of 256 x R workgroups (R workgroups per CU, 4 waves each = R waves per SIMD), every wave runs ITERS iterations
// of one "K step" = 16 MFMAs (1024 pipe cycles):
//   mode 0  4 independent accumulators, nothing else                              (the guide's 155 TFLOP/s measurement)
//   mode 1  ONE accumulator (16 dependent MFMAs per step), nothing else           (conv_taps' accumulator structure)
//   mode 2  mode 1 + two s_barrier per step
//   mode 3  mode 2 + 8 ds_read_b128 per step (the A / B fragments of a 64x64x32 tile)
//   mode 4  mode 3 + 4 ds_write_b128 per step (register-staged tile -> LDS)
//   mode 5  mode 4 + 4 16-byte global loads per thread and step from a 256 MB buffer (= the whole K-loop skeleton)
// Reported: TFLOP/s (hipEvent time), and the shader clock = s_memtime cycles / s_memrealtime (100 MHz) seen by one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDP 36

template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ clk,
                                                 int iters, size_t src_floats) {
    __shared__ __attribute__((aligned(16))) float sA[64 * LDP];
    __shared__ __attribute__((aligned(16))) float sB[64 * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 64 * LDP; i += 256) {
        sA[i] = src[(i * 7 + blockIdx.x) % 4096];
        sB[i] = src[(i * 13 + blockIdx.x) % 4096];
    }
    __syncthreads();
    const float* pa = sA + (wm * 32 + (lane & 31)) * LDP + (lane >> 5) * 4;
    const float* pb = sB + (wn * 32 + (lane & 31)) * LDP + (lane >> 5) * 4;
    const int kv = tid & 7, r0 = tid >> 3;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    f32x4 a[4], b[4], ra[2], rb[2];
    for (int j = 0; j < 4; ++j) {
        a[j] = *(const f32x4*)(pa + j * 8);
        b[j] = *(const f32x4*)(pb + j * 8);
    }
    for (int i = 0; i < 2; ++i) ra[i] = a[i], rb[i] = b[i];
    // every workgroup streams its own 16 KB-per-step window through the buffer (wraps around)
    size_t goff = ((size_t)blockIdx.x * 7919u * 4096u + (size_t)tid * 4) % (src_floats - 16384);
    unsigned long long t0 = 0, c0 = 0;
    if (tid == 0) {
        c0 = __builtin_readcyclecounter();
        t0 = wall_clock64();
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *(f32x4*)&sA[(r0 + 32 * i) * LDP + kv * 4] = ra[i];
                *(f32x4*)&sB[(r0 + 32 * i) * LDP + kv * 4] = rb[i];
            }
        }
        if constexpr (MODE >= 2) __syncthreads();
        if constexpr (MODE >= 5) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = *(const f32x4*)(src + goff + i * 1024);
                rb[i] = *(const f32x4*)(src + goff + 2048 + i * 1024);
            }
            goff += 4096;
            if (goff >= src_floats - 16384) goff -= (src_floats - 16384);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE >= 3) {
                a[j] = *(const f32x4*)(pa + j * 8);
                b[j] = *(const f32x4*)(pb + j * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (MODE == 0)
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], b[j][e], acc[e], 0, 0, 0);
                else
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], b[j][e], acc[0], 0, 0, 0);
            }
        }
        if constexpr (MODE >= 2) __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = t1 - t0;
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if constexpr (MODE >= 4) s += ra[0][0] + rb[1][3];
    if (s == 123.456f) sink[tid] = s;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <int MODE>
static void run(const float* src, float* sink, unsigned long long* clk, int R, int iters, size_t src_floats, const char* what) {
    const int grid = 256 * R;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters / 4, src_floats);  // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(256), 0, 0, src, sink, clk, iters, src_floats);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> h(2 * grid);
    CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, ticks = 0;
    for (int i = 0; i < grid; ++i) cyc += (double)h[2 * i], ticks += (double)h[2 * i + 1];
    const double ghz = cyc / (ticks * 10.0);  // 100 MHz realtime counter: 10 ns per tick
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    const double tf = flop / (best * 1e-3) / 1e12;
    printf("mode %d  R=%d waves/SIMD  %-58s %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)  clock %.2f GHz  pipe-busy-at-that-clock %.3f\n", MODE, R,
           what, best, tf, tf / 157.3, ghz, tf / (157.3 * ghz / 2.4));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t src_floats = (size_t)64 << 20;  // 256 MB
    float *src, *sink;
    unsigned long long* clk;
    CK(hipMalloc(&src, src_floats * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * 256 * 8));
    std::vector<float> h(src_floats);
    unsigned s = 12345u;
    for (size_t i = 0; i < src_floats; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    }
    CK(hipMemcpy(src, h.data(), src_floats * 4, hipMemcpyHostToDevice));
    const int Rs[] = {1, 2, 4, 7};
    for (int R : Rs) {
        const int it = iters * 4 / (R < 4 ? R : 4) / (R == 7 ? 2 : 1);
        run<0>(src, sink, clk, R, it, src_floats, "4 accumulators, MFMA only");
        run<1>(src, sink, clk, R, it, src_floats, "1 accumulator, MFMA only");
        run<2>(src, sink, clk, R, it, src_floats, "+ 2 barriers / step");
        run<3>(src, sink, clk, R, it, src_floats, "+ 8 ds_read_b128 / step");
        run<4>(src, sink, clk, R, it, src_floats, "+ 4 ds_write_b128 / step");
        run<5>(src, sink, clk, R, it, src_floats, "+ 4 global 16-B loads / thread / step (K-loop skeleton)");
    }
    return 0;
}
