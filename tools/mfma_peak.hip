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
//   mode 5  mode 4 + 4 16-byte global loads per thread and step (= the whole K-loop skeleton); the loads stream through a window of
//           `win` floats per workgroup-group: 5a 1 MB shared by all workgroups (L2 hits), 5b 64 MB (Infinity-Cache resident),
//           5c 256 MB (HBM: 16 KB per workgroup and step with no reuse = 7 TB/s at full MFMA rate -- NOT what the conv does)
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
    size_t goff = ((size_t)blockIdx.x * 7919u * 4096u + (size_t)tid * 4) % (src_floats - 16384);  // src_floats = the window
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


// ---- the skeleton of a software-pipelined big-tile K loop: BM x BN x 32 tiles, 4 waves in a 2x2 grid (wave tile BM/2 x BN/2 =
// TM x TN accumulators of 32x32), TWO LDS buffers and ONE barrier per step: tile s+1 goes registers -> LDS[next] and the global
// loads of tile s+2 are issued while tile s is multiplied out of LDS[cur].  Global loads as in mode 5 (window = src_floats).
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_loop(const float* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ clk,
                                                 int iters, size_t src_floats) {
    constexpr int TM = BM / 64, TN = BN / 64, RA = BM / 32, RB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                  // [2][BM * LDP]
    float* sB = smem + 2 * BM * LDP;   // [2][BN * LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * BM * LDP; i += 256) sA[i] = src[(i * 7 + blockIdx.x) % 4096];
    for (int i = tid; i < 2 * BN * LDP; i += 256) sB[i] = src[(i * 13 + blockIdx.x) % 4096];
    __syncthreads();
    const int kv = tid & 7, r0 = tid >> 3;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[RA], rb[RB];
    for (int i = 0; i < RA; ++i) ra[i] = *(const f32x4*)&sA[(r0 + 32 * i) * LDP + kv * 4];
    for (int i = 0; i < RB; ++i) rb[i] = *(const f32x4*)&sB[(r0 + 32 * i) * LDP + kv * 4];
    const size_t span = src_floats - 65536;
    size_t goff = ((size_t)blockIdx.x * 7919u * 4096u + (size_t)tid * 4) % span;
    unsigned long long t0 = 0, c0 = 0;
    if (tid == 0) {
        c0 = __builtin_readcyclecounter();
        t0 = wall_clock64();
    }
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        float* wA = sA + (cur ^ 1) * BM * LDP;
        float* wB = sB + (cur ^ 1) * BN * LDP;
#pragma unroll
        for (int i = 0; i < RA; ++i) *(f32x4*)&wA[(r0 + 32 * i) * LDP + kv * 4] = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(f32x4*)&wB[(r0 + 32 * i) * LDP + kv * 4] = rb[i];
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = *(const f32x4*)(src + goff + i * 1024);
#pragma unroll
        for (int i = 0; i < RB; ++i) rb[i] = *(const f32x4*)(src + goff + (RA + i) * 1024);
        goff += (RA + RB) * 1024;
        if (goff >= span) goff -= span;
        __builtin_amdgcn_sched_barrier(0);  // keep the loads of tile s+2 up here: hipcc otherwise sinks them to the end of the step
        const float* pa = sA + cur * BM * LDP + (wm * (BM / 2) + (lane & 31)) * LDP + (lane >> 5) * 4;
        const float* pb = sB + cur * BN * LDP + (wn * (BN / 2) + (lane & 31)) * LDP + (lane >> 5) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = *(const f32x4*)(pa + tm * 32 * LDP + j * 8);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = *(const f32x4*)(pb + tn * 32 * LDP + j * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = t1 - t0;
    }
    float s = ra[0][0] + rb[0][1];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[tid] = s;
}


// ---- the same big tile with the feeding instructions INTERLEAVED between the MFMAs in program order (an in-order wave overlaps its
// own MFMAs only with what follows them in the stream): per K step of 4 k-groups (16*TM*TN/4 MFMAs each)
//   group 0: MFMAs + the ds_writes of tile s+1 (-> LDS[next]) + the fragment reads of group 1
//   group 1: MFMAs + the global loads of tile s+2                + the fragment reads of group 2
//   group 2: MFMAs                                               + the fragment reads of group 3
//   group 3: a few MFMAs, lgkmcnt(0) + s_barrier, the rest of the MFMAs + the fragment reads of group 0 of tile s+1 (LDS[next])
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_loop_p(const float* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ clk,
                                                   int iters, size_t src_floats) {
    constexpr int TM = BM / 64, TN = BN / 64, RA = BM / 32, RB = BN / 32, NM = TM * TN * 4;  // NM MFMAs per k-group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + 2 * BM * LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * BM * LDP; i += 256) sA[i] = src[(i * 7 + blockIdx.x) % 4096];
    for (int i = tid; i < 2 * BN * LDP; i += 256) sB[i] = src[(i * 13 + blockIdx.x) % 4096];
    __syncthreads();
    const int kv = tid & 7, r0 = tid >> 3;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[RA], rb[RB];
    for (int i = 0; i < RA; ++i) ra[i] = *(const f32x4*)&sA[(r0 + 32 * i) * LDP + kv * 4];
    for (int i = 0; i < RB; ++i) rb[i] = *(const f32x4*)&sB[(r0 + 32 * i) * LDP + kv * 4];
    const size_t span = src_floats - 65536;
    size_t goff = ((size_t)blockIdx.x * 7919u * 4096u + (size_t)tid * 4) % span;
    const int fa = (wm * (BM / 2) + (lane & 31)) * LDP + (lane >> 5) * 4, fb = (wn * (BN / 2) + (lane & 31)) * LDP + (lane >> 5) * 4;
    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a0[tm] = *(const f32x4*)(sA + fa + tm * 32 * LDP);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b0[tn] = *(const f32x4*)(sB + fb + tn * 32 * LDP);
    unsigned long long t0 = 0, c0 = 0;
    if (tid == 0) {
        c0 = __builtin_readcyclecounter();
        t0 = wall_clock64();
    }
#define MFMA_GROUP(A, B)                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)              \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                        \
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tm][e], B[tn][e], acc[tm][tn], 0, 0, 0)
#define READ_FRAGS(A, B, PA, PB, J)                                                                              \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = *(const f32x4*)((PA) + tm * 32 * LDP + (J) * 8);   \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) B[tn] = *(const f32x4*)((PB) + tn * 32 * LDP + (J) * 8)
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        float* wA = sA + (cur ^ 1) * BM * LDP;
        float* wB = sB + (cur ^ 1) * BN * LDP;
        const float* pa = sA + cur * BM * LDP + fa;
        const float* pb = sB + cur * BN * LDP + fb;
        // ---- group 0
#pragma unroll
        for (int i = 0; i < RA; ++i) *(f32x4*)&wA[(r0 + 32 * i) * LDP + kv * 4] = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(f32x4*)&wB[(r0 + 32 * i) * LDP + kv * 4] = rb[i];
        READ_FRAGS(a1, b1, pa, pb, 1);
        MFMA_GROUP(a0, b0);
        // ---- group 1
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = *(const f32x4*)(src + goff + i * 1024);
#pragma unroll
        for (int i = 0; i < RB; ++i) rb[i] = *(const f32x4*)(src + goff + (RA + i) * 1024);
        goff += (RA + RB) * 1024;
        if (goff >= span) goff -= span;
        READ_FRAGS(a0, b0, pa, pb, 2);
        MFMA_GROUP(a1, b1);
        // ---- group 2
        READ_FRAGS(a1, b1, pa, pb, 3);
        MFMA_GROUP(a0, b0);
#pragma unroll
        for (int q = 0; q < RA + RB; ++q) { SGB(0x8, 1); SGB(0x200, 1); }
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) { SGB(0x8, 1); SGB(0x100, 1); }
        SGB(0x8, NM - (RA + RB) - (TM + TN));
#pragma unroll
        for (int q = 0; q < RA + RB; ++q) { SGB(0x8, 1); SGB(0x20, 1); }
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) { SGB(0x8, 1); SGB(0x100, 1); }
        SGB(0x8, NM - (RA + RB) - (TM + TN));
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) { SGB(0x8, 1); SGB(0x100, 1); }
        SGB(0x8, NM - (TM + TN));
        // ---- group 3: everybody is done reading LDS[cur] and writing LDS[next] once the fragment reads above have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        READ_FRAGS(a0, b0, wA + fa, wB + fb, 0);
        MFMA_GROUP(a1, b1);
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) { SGB(0x8, 1); SGB(0x100, 1); }
        SGB(0x8, NM - (TM + TN));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = t1 - t0;
    }
    float s = ra[0][0] + rb[0][1] + a0[0][0] + b0[0][0];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
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

typedef void (*big_kernel_t)(const float*, float*, unsigned long long*, int, size_t);
template <int BM, int BN, bool PIPE>
static void run_big(const float* src, float* sink, unsigned long long* clk, int R, int iters, size_t src_floats, const char* what) {
    const int grid = 256 * R;
    const size_t lds = (size_t)2 * (BM + BN) * LDP * 4;
    big_kernel_t kern = PIPE ? (big_kernel_t)gemm_loop_p<BM, BN> : (big_kernel_t)gemm_loop<BM, BN>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, sink, clk, iters / 4, src_floats);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, sink, clk, iters, src_floats);
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
    const double ghz = cyc / (ticks * 10.0);
    const double flop = (double)grid * iters * 2.0 * BM * BN * 32;
    const double tf = flop / (best * 1e-3) / 1e12;
    printf("tile %3dx%3d %s R=%d WG/CU  %-56s %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)  clock %.2f GHz  pipe-busy-at-that-clock %.3f\n", BM, BN, PIPE ? "interleaved" : "plain      ", R,
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
        run<5>(src, sink, clk, R, it, (size_t)1 << 18, "+ 4 global 16-B loads / thread / step, 1 MB window (L2)");
        run<5>(src, sink, clk, R, it, (size_t)16 << 20, "+ 4 global 16-B loads / thread / step, 64 MB window (MALL)");
        run<5>(src, sink, clk, R, it, src_floats, "+ 4 global 16-B loads / thread / step, 256 MB window (HBM)");
    }
    const int RB[] = {1, 2};
    for (int R : RB) {
        const int it = iters / R;
        run_big<128, 128, false>(src, sink, clk, R, it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<128, 128, true>(src, sink, clk, R, it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<128, 128, true>(src, sink, clk, R, it, (size_t)16 << 20, "2 LDS buffers, 1 barrier; 64 MB window (MALL)");
        run_big<128, 64, false>(src, sink, clk, R, 2 * it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<128, 64, true>(src, sink, clk, R, 2 * it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<128, 64, true>(src, sink, clk, R, 2 * it, (size_t)16 << 20, "2 LDS buffers, 1 barrier; 64 MB window (MALL)");
        run_big<256, 64, true>(src, sink, clk, R, it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<64, 64, true>(src, sink, clk, R, 4 * it, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
    }
    for (int R : {4, 6}) {
        run_big<64, 64, true>(src, sink, clk, R, 4 * iters / R, (size_t)1 << 18, "2 LDS buffers, 1 barrier; 1 MB window (L2)");
        run_big<64, 64, true>(src, sink, clk, R, 4 * iters / R, (size_t)16 << 20, "2 LDS buffers, 1 barrier; 64 MB window (MALL)");
    }
    return 0;
}
