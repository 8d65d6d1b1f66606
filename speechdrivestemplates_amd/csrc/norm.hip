// Normalisation + activation kernels (HBM-bound; 16-B coalesced channel vectors, wave-shuffle and
// LDS reductions, fp64 accumulation of the per-(group,channel) statistics).
//
//  colnorm_* : statistics per (g, c) over the R rows of a (G, R, C) channels-last view
//              G=B,R=HW  -> nn.InstanceNorm2d  (building_blocks.py:26)
//              G=1,R=B*L -> nn.BatchNorm1d/2d in training mode (building_blocks.py:24,39)
//  rownorm_* : statistics per row over C -> the reference's InstanceNorm1d on the permuted tensor
//              (building_blocks.py:50-51), i.e. a per-(b,t) LayerNorm without affine
//  both followed by LeakyReLU(0.2) / ReLU (building_blocks.py:46).
#include <stdlib.h>

#include "common.h"

#define MAXC 1024
#define UNR 4  // rows (16-byte vectors) in flight per thread in the streaming passes

// Four consecutive channels of a channels-last tensor as fp32, whatever it is stored as: fp32 (16-byte access) or bf16 (8-byte access; the
// bf16-storage path keeps the Conv2d chain's activations in HBM as bf16, statistics and arithmetic stay fp32)
typedef unsigned nm_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 nm_bf16x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* __restrict__ p) {
    if constexpr (sizeof(T) == 4) {
        return *(const f32x4*)p;
    } else {
        const nm_u32x2 w = *(const nm_u32x2*)p;
        return (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
    }
}
template <typename T>
__device__ __forceinline__ void st4(T* __restrict__ p, const f32x4 v) {
    if constexpr (sizeof(T) == 4) {
        *(f32x4*)p = v;
    } else {
        nm_bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];  // round to nearest even
        *(nm_bf16x4*)p = h;
    }
}

// ---------------------------------------------------------------------------------------------
template <bool BWD, typename TA = float, typename TB = float>
__global__ __launch_bounds__(256) void colstats_kernel(const TA* __restrict__ a,   // fwd: y      bwd: dz
                                                       const TB* __restrict__ y,   // bwd only
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float slope, double* __restrict__ sums, int64_t R, int C,
                                                       int rows_per_block) {
    __shared__ double sS[MAXC], sQ[MAXC];
    const int tid = threadIdx.x, g = blockIdx.y;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int cv = tid % tpr, rr = tid / tpr;
    for (int c = tid; c < C; c += 256) sS[c] = 0.0, sQ[c] = 0.0;
    __syncthreads();
    if (rr < rpp) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r1 = min(R, r0 + rows_per_block);
        const size_t base = (size_t)g * R * C + 4 * cv;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu, ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BWD) {
            mu = *(const f32x4*)(mean + (size_t)g * C + 4 * cv);
            rs = *(const f32x4*)(rstd + (size_t)g * C + 4 * cv);
            if (gamma) ga = *(const f32x4*)(gamma + 4 * cv);
            if (beta) be = *(const f32x4*)(beta + 4 * cv);
        }
        // UNR rows in flight per thread (16-byte loads issued back to back before any of them is used): with one load per
        // loop trip a workgroup keeps only 4 KB in flight and the pass runs at ~2.5 TB/s whatever the grid
        for (int64_t r = r0 + rr; r < r1; r += (int64_t)UNR * rpp) {
            f32x4 v[UNR], yv[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int64_t rj = r + (int64_t)j * rpp;
                const bool ok = rj < r1;
                v[j] = ok ? ld4(a + base + (size_t)rj * C) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (BWD) yv[j] = ok ? ld4(y + base + (size_t)rj * C) : mu;  // masked rows: dz == 0
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if constexpr (!BWD) {
                    s += v[j];
                    q += v[j] * v[j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yh = (yv[j][e] - mu[e]) * rs[e];
                        const float gg = v[j][e] * act_grad(yh * ga[e] + be[e], slope);
                        s[e] += gg;
                        q[e] += gg * yh;
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&sS[4 * cv + e], (double)s[e]);
            atomicAdd(&sQ[4 * cv + e], (double)q[e]);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        atomicAdd(&sums[((size_t)g * C + c) * 2], sS[c]);
        atomicAdd(&sums[((size_t)g * C + c) * 2 + 1], sQ[c]);
    }
}

template <typename TY = float, typename TZ = float>
__global__ __launch_bounds__(256) void colnorm_apply_fwd_kernel(const TY* __restrict__ y, TZ* __restrict__ z,
                                                                const double* __restrict__ sums, float* __restrict__ mean,
                                                                float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                                int64_t R, int C, int rows_per_block, float eps,
                                                                float momentum, float slope) {
    const int tid = threadIdx.x, g = blockIdx.y;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int cv = tid % tpr, rr = tid / tpr;
    if (rr >= rpp) return;
    f32x4 mu, rs, ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    double var[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double s = sums[((size_t)g * C + 4 * cv + e) * 2], q = sums[((size_t)g * C + 4 * cv + e) * 2 + 1];
        const double m = s / (double)R;
        double v = q / (double)R - m * m;
        v = v > 0.0 ? v : 0.0;
        var[e] = v;
        mu[e] = (float)m;
        rs[e] = (float)(1.0 / sqrt(v + (double)eps));
    }
    if (gamma) ga = *(const f32x4*)(gamma + 4 * cv);
    if (beta) be = *(const f32x4*)(beta + 4 * cv);
    if (blockIdx.x == 0 && rr == 0) {
        *(f32x4*)(mean + (size_t)g * C + 4 * cv) = mu;
        *(f32x4*)(rstd + (size_t)g * C + 4 * cv) = rs;
        if (rmean != nullptr && g == 0) {  // nn.BatchNorm training-mode running statistics (unbiased variance)
            // G > 1 with running statistics: G calls of the module on G equal slices of the batch, batched into one launch (the two no-grad
            // pose-encoder passes of a train step, voice2pose.py:160-176): the running statistics take the G updates in call order
            const double unb = R > 1 ? (double)R / (double)(R - 1) : 1.0;
            const int G = (int)gridDim.y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * cv + e;
                float rm = (1.f - momentum) * rmean[c] + momentum * mu[e];
                float rv = (1.f - momentum) * rvar[c] + momentum * (float)(var[e] * unb);
                for (int gg = 1; gg < G; ++gg) {
                    const double s = sums[((size_t)gg * C + c) * 2], q = sums[((size_t)gg * C + c) * 2 + 1];
                    const double m = s / (double)R;
                    double v = q / (double)R - m * m;
                    v = v > 0.0 ? v : 0.0;
                    rm = (1.f - momentum) * rm + momentum * (float)m;
                    rv = (1.f - momentum) * rv + momentum * (float)(v * unb);
                }
                rmean[c] = rm;
                rvar[c] = rv;
            }
            if (nbt != nullptr && cv == 0) nbt[0] += G;
        }
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    const size_t base = (size_t)g * R * C + 4 * cv;
    for (int64_t r = r0 + rr; r < r1; r += (int64_t)UNR * rpp) {
        f32x4 v[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t rj = r + (int64_t)j * rpp;
            if (rj < r1) v[j] = ld4(y + base + (size_t)rj * C);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t rj = r + (int64_t)j * rpp;
            if (rj < r1) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_fwd(((v[j][e] - mu[e]) * rs[e]) * ga[e] + be[e], slope);
                st4(z + base + (size_t)rj * C, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void colnorm_eval_kernel(const float* __restrict__ y, float* __restrict__ z,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                           int64_t rows, int C, float eps, float slope) {
    const int64_t nvec = rows * (C >> 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % (C >> 2));
        const f32x4 v = *(const f32x4*)(y + 4 * i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * cv + e;
            const float rs = 1.f / sqrtf(rvar[c] + eps);
            const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
            o[e] = act_fwd(((v[e] - rmean[c]) * rs) * ga + be, slope);
        }
        *(f32x4*)(z + 4 * i) = o;
    }
}

template <typename TZ = float, typename TY = float, typename TD = float>
__global__ __launch_bounds__(256) void colnorm_apply_bwd_kernel(const TZ* __restrict__ dz, const TY* __restrict__ y,
                                                                TD* __restrict__ dy, const double* __restrict__ sums,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int64_t R, int C, int rows_per_block, float slope) {
    const int tid = threadIdx.x, g = blockIdx.y;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int cv = tid % tpr, rr = tid / tpr;
    if (rr >= rpp) return;
    f32x4 mu = *(const f32x4*)(mean + (size_t)g * C + 4 * cv);
    f32x4 rs = *(const f32x4*)(rstd + (size_t)g * C + 4 * cv);
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f}, mg, mgy;
    if (gamma) ga = *(const f32x4*)(gamma + 4 * cv);
    if (beta) be = *(const f32x4*)(beta + 4 * cv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double sg = sums[((size_t)g * C + 4 * cv + e) * 2], sgy = sums[((size_t)g * C + 4 * cv + e) * 2 + 1];
        mg[e] = (float)(sg / (double)R);
        mgy[e] = (float)(sgy / (double)R);
        if (blockIdx.x == 0 && rr == 0) {
            if (dgamma) atomicAdd(&dgamma[4 * cv + e], (float)sgy);
            if (dbeta) atomicAdd(&dbeta[4 * cv + e], (float)sg);
        }
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    const size_t base = (size_t)g * R * C + 4 * cv;
    for (int64_t r = r0 + rr; r < r1; r += (int64_t)UNR * rpp) {
        f32x4 gz[UNR], yv[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t rj = r + (int64_t)j * rpp;
            if (rj < r1) {
                gz[j] = ld4(dz + base + (size_t)rj * C);
                yv[j] = ld4(y + base + (size_t)rj * C);
            }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t rj = r + (int64_t)j * rpp;
            if (rj < r1) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yh = (yv[j][e] - mu[e]) * rs[e];
                    const float gg = gz[j][e] * act_grad(yh * ga[e] + be[e], slope);
                    o[e] = ga[e] * rs[e] * (gg - mg[e] - yh * mgy[e]);
                }
                st4(dy + base + (size_t)rj * C, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One wave per row; NV float4 per lane (C <= 256*NV).
// SLABS (forward only): ``a`` holds nslab split-K partial outputs of the producing conv, slab_stride floats apart; they are
// summed here in slab order (exactly what splitk_reduce_kernel would have done) and the sum is also stored to ``ysum``
// (the raw conv output backward needs), so conv -> reduce -> norm is two launches instead of three.
template <int NV, bool BWD, bool SLABS = false>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ a,  // fwd: y   bwd: dz
                                                      const float* __restrict__ y, float* __restrict__ out,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                      int C, float eps, float slope, int nslab = 1, size_t slab_stride = 0,
                                                      float* __restrict__ ysum = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * C;
    f32x4 v[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        ok[i] = c < C;
        v[i] = ok[i] ? *(const f32x4*)(a + base + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SLABS) {
            if (ok[i]) {
                for (int z = 1; z < nslab; ++z) v[i] += *(const f32x4*)(a + (size_t)z * slab_stride + base + c);
                *(f32x4*)(ysum + base + c) = v[i];
            }
        }
    }
    const float invC = 1.f / (float)C;
    if constexpr (!BWD) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mu;
                    q += d * d;
                }
            }
        const float var = wave_sum(q) * invC;
        const float rs = 1.f / sqrtf(var + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_fwd((v[i][e] - mu) * rs, slope);
                *(f32x4*)(out + base + 4 * (lane + 64 * i)) = o;
            }
    } else {
        const float mu = mean[row], rs = rstd[row];
        f32x4 yh[NV];
        float sg = 0.f, sgy = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 yv = ok[i] ? *(const f32x4*)(y + base + 4 * (lane + 64 * i)) : (f32x4){mu, mu, mu, mu};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yh[i][e] = (yv[e] - mu) * rs;
                v[i][e] = v[i][e] * act_grad(yh[i][e], slope);  // dz==0 on masked lanes
                sg += v[i][e];
                sgy += v[i][e] * yh[i][e];
            }
        }
        sg = wave_sum(sg) * invC;
        sgy = wave_sum(sgy) * invC;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (v[i][e] - sg - yh[i][e] * sgy);
                *(f32x4*)(out + base + 4 * (lane + 64 * i)) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
static int colnorm_rows_per_block(int C, int G, int64_t R, bool stats) {
    // Streaming passes: ~2048 workgroups over all groups (8 per CU = every wave slot) with UNR 16-byte loads in flight per
    // thread, i.e. >= 32 KB in flight per CU -- what it takes to cover HBM latency at several TB/s.  A workgroup takes at least
    // one full unrolled trip (UNR * rpp rows) and at most 64 rows per thread (fp32 partial sums stay accurate).
    const int rpp = 256 / (C >> 2);
    // statistics of ONE group (BatchNorm): every workgroup funnels 2*C fp64 atomics into the same addresses -> fewer, longer ones
    const int64_t target = (stats && G == 1) ? 512 : 2048;
    const int64_t want = cdiv64(R, std::max<int64_t>(1, target / G));
    const int64_t unit = (int64_t)UNR * rpp;
    const int64_t rpb = cdiv64(want, unit) * unit;
    return (int)std::min<int64_t>(std::max<int64_t>(rpb, unit), (int64_t)rpp * 64);
}
static int check_colnorm(int G, int64_t R, int C) {
    SDT_CHECK_ARG(G > 0 && R > 0, "non-positive dims");
    SDT_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= MAXC, "C must be a multiple of 4 in [4,1024]");
    return SDT_OK;
}

static bool dtype_ok(int d) { return d == SDT_F32 || d == SDT_BF16; }

// y (y_dtype) -> z (z_dtype); everything else as sdt_colnorm_fwd_f32
extern "C" int sdt_colnorm_fwd_t(const void* y, int y_dtype, void* z, int z_dtype, double* sums, float* mean, float* rstd,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, int G, int64_t R, int C, float eps, float momentum,
                                 float slope, int stats_ready, void* stream) {
    int rc = check_colnorm(G, R, C);
    if (rc) return rc;
    SDT_CHECK_ARG(y && z && sums && mean && rstd, "null pointer");
    SDT_CHECK_ARG(dtype_ok(y_dtype) && dtype_ok(z_dtype), "unknown element type");
    SDT_CHECK_ARG(y_dtype == SDT_BF16 || z_dtype == SDT_F32, "fp32 y with a bf16 z is not built");
    hipStream_t s = (hipStream_t)stream;
    if (!stats_ready) {  // otherwise the producing conv's epilogue already accumulated them (sdt_conv_taps_stats_f32)
        const int rps = colnorm_rows_per_block(C, G, R, true);
        const dim3 sgrid((unsigned)cdiv64(R, rps), G);
        if (y_dtype == SDT_F32)
            hipLaunchKernelGGL((colstats_kernel<false, float, float>), sgrid, dim3(256), 0, s, (const float*)y, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, slope, sums, R, C, rps);
        else
            hipLaunchKernelGGL((colstats_kernel<false, __bf16, float>), sgrid, dim3(256), 0, s, (const __bf16*)y, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, slope, sums, R, C, rps);
    }
    const int rpb = colnorm_rows_per_block(C, G, R, false);
    dim3 grid((unsigned)cdiv64(R, rpb), G);
#define FWD_GO(TY_, TZ_)                                                                                                              \
    hipLaunchKernelGGL((colnorm_apply_fwd_kernel<TY_, TZ_>), grid, dim3(256), 0, s, (const TY_*)y, (TZ_*)z, sums, mean, rstd, gamma, beta, \
                       running_mean, running_var, num_batches_tracked, R, C, rpb, eps, momentum, slope)
    if (y_dtype == SDT_F32) FWD_GO(float, float);
    else if (z_dtype == SDT_BF16) FWD_GO(__bf16, __bf16);
    else FWD_GO(__bf16, float);
#undef FWD_GO
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
extern "C" int sdt_colnorm_fwd_f32(const float* y, float* z, double* sums, float* mean, float* rstd,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, int G, int64_t R, int C, float eps, float momentum,
                                   float slope, int stats_ready, void* stream) {
    return sdt_colnorm_fwd_t(y, SDT_F32, z, SDT_F32, sums, mean, rstd, gamma, beta, running_mean, running_var, num_batches_tracked, G, R, C, eps,
                             momentum, slope, stats_ready, stream);
}

extern "C" int sdt_colnorm_eval_f32(const float* y, float* z, const float* gamma, const float* beta,
                                    const float* running_mean, const float* running_var, int64_t rows, int C, float eps,
                                    float slope, void* stream) {
    int rc = check_colnorm(1, rows, C);
    if (rc) return rc;
    SDT_CHECK_ARG(y && z && running_mean && running_var, "null pointer");
    const int64_t nvec = rows * (C >> 2);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(nvec, 256), 2048);
    hipLaunchKernelGGL(colnorm_eval_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, z, gamma, beta, running_mean,
                       running_var, rows, C, eps, slope);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

// dz (dz_dtype), y (y_dtype) -> dy (dy_dtype); everything else as sdt_colnorm_bwd_f32.  Built: all fp32; y and dy bf16 with dz bf16 or fp32.
extern "C" int sdt_colnorm_bwd_t(const void* dz, int dz_dtype, const void* y, int y_dtype, void* dy, int dy_dtype, double* sums, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                 int G, int64_t R, int C, float slope, int stats_ready, void* stream) {
    int rc = check_colnorm(G, R, C);
    if (rc) return rc;
    SDT_CHECK_ARG(dz && y && dy && sums && mean && rstd, "null pointer");
    SDT_CHECK_ARG(!(dgamma || dbeta) || G == 1, "affine gradients need G == 1");
    SDT_CHECK_ARG(dtype_ok(dz_dtype) && dtype_ok(y_dtype) && dtype_ok(dy_dtype), "unknown element type");
    const bool all32 = dz_dtype == SDT_F32 && y_dtype == SDT_F32 && dy_dtype == SDT_F32;
    SDT_CHECK_ARG(all32 || (y_dtype == SDT_BF16 && dy_dtype == SDT_BF16), "element type combination not built");
    hipStream_t s = (hipStream_t)stream;
    if (!stats_ready) {
        const int rps = colnorm_rows_per_block(C, G, R, true);
        const dim3 sgrid((unsigned)cdiv64(R, rps), G);
#define ST_GO(TA_, TB_) \
    hipLaunchKernelGGL((colstats_kernel<true, TA_, TB_>), sgrid, dim3(256), 0, s, (const TA_*)dz, (const TB_*)y, mean, rstd, gamma, beta, slope, sums, R, C, rps)
        if (all32) ST_GO(float, float);
        else if (dz_dtype == SDT_F32) ST_GO(float, __bf16);
        else ST_GO(__bf16, __bf16);
#undef ST_GO
    }
    const int rpb = colnorm_rows_per_block(C, G, R, false);
    dim3 grid((unsigned)cdiv64(R, rpb), G);
#define BWD_GO(TZ_, TY_, TD_)                                                                                                        \
    hipLaunchKernelGGL((colnorm_apply_bwd_kernel<TZ_, TY_, TD_>), grid, dim3(256), 0, s, (const TZ_*)dz, (const TY_*)y, (TD_*)dy, sums, mean, rstd, \
                       gamma, beta, dgamma, dbeta, R, C, rpb, slope)
    if (all32) BWD_GO(float, float, float);
    else if (dz_dtype == SDT_F32) BWD_GO(float, __bf16, __bf16);
    else BWD_GO(__bf16, __bf16, __bf16);
#undef BWD_GO
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
extern "C" int sdt_colnorm_bwd_f32(const float* dz, const float* y, float* dy, double* sums, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                   int G, int64_t R, int C, float slope, int stats_ready, void* stream) {
    return sdt_colnorm_bwd_t(dz, SDT_F32, y, SDT_F32, dy, SDT_F32, sums, mean, rstd, gamma, beta, dgamma, dbeta, G, R, C, slope, stats_ready, stream);
}

template <bool BWD>
static int launch_rownorm(const float* a, const float* y, float* out, float* mean, float* rstd, int64_t rows, int C,
                          float eps, float slope, hipStream_t s) {
    const unsigned grid = (unsigned)cdiv64(rows, 4);
    const int nv = cdiv(C, 256);
    switch (nv) {
        case 1: hipLaunchKernelGGL((rownorm_kernel<1, BWD>), dim3(grid), dim3(256), 0, s, a, y, out, mean, rstd, rows, C, eps, slope); break;
        case 2: hipLaunchKernelGGL((rownorm_kernel<2, BWD>), dim3(grid), dim3(256), 0, s, a, y, out, mean, rstd, rows, C, eps, slope); break;
        case 3:
        case 4: hipLaunchKernelGGL((rownorm_kernel<4, BWD>), dim3(grid), dim3(256), 0, s, a, y, out, mean, rstd, rows, C, eps, slope); break;
        default: return SDT_ERR_UNSUPPORTED;
    }
    return SDT_OK;
}

extern "C" int sdt_rownorm_fwd_f32(const float* y, float* z, float* mean, float* rstd, int64_t rows, int C, float eps,
                                   float slope, void* stream) {
    SDT_CHECK_ARG(y && z && mean && rstd && rows > 0, "bad argument");
    SDT_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= MAXC, "C must be a multiple of 4 in [4,1024]");
    int rc = launch_rownorm<false>(y, nullptr, z, mean, rstd, rows, C, eps, slope, (hipStream_t)stream);
    if (rc) return rc;
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_rownorm_slabs_fwd_f32(const float* partial, int nslab, float* y, float* z, float* mean, float* rstd,
                                         int64_t rows, int C, float eps, float slope, void* stream) {
    SDT_CHECK_ARG(partial && y && z && mean && rstd && rows > 0 && nslab >= 1 && nslab <= 64, "bad argument");
    SDT_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= MAXC, "C must be a multiple of 4 in [4,1024]");
    const unsigned grid = (unsigned)cdiv64(rows, 4);
    const size_t stride = (size_t)rows * C;
    hipStream_t s = (hipStream_t)stream;
    switch (cdiv(C, 256)) {
        case 1: hipLaunchKernelGGL((rownorm_kernel<1, false, true>), dim3(grid), dim3(256), 0, s, partial, (const float*)nullptr, z, mean, rstd, rows, C, eps, slope, nslab, stride, y); break;
        case 2: hipLaunchKernelGGL((rownorm_kernel<2, false, true>), dim3(grid), dim3(256), 0, s, partial, (const float*)nullptr, z, mean, rstd, rows, C, eps, slope, nslab, stride, y); break;
        case 3:
        case 4: hipLaunchKernelGGL((rownorm_kernel<4, false, true>), dim3(grid), dim3(256), 0, s, partial, (const float*)nullptr, z, mean, rstd, rows, C, eps, slope, nslab, stride, y); break;
        default: sdt_set_error("%s: unsupported channel count", __func__); return SDT_ERR_UNSUPPORTED;
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_rownorm_bwd_f32(const float* dz, const float* y, const float* mean, const float* rstd, float* dy,
                                   int64_t rows, int C, float slope, void* stream) {
    SDT_CHECK_ARG(dz && y && mean && rstd && dy && rows > 0, "bad argument");
    SDT_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= MAXC, "C must be a multiple of 4 in [4,1024]");
    int rc = launch_rownorm<true>(dz, y, dy, (float*)mean, (float*)rstd, rows, C, 0.f, slope, (hipStream_t)stream);
    if (rc) return rc;
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
