// First audio-encoder block (Conv2d 1->64 k3 s1 p1 + InstanceNorm2d/BatchNorm2d + LeakyReLU, generator.py:16,
// building_blocks.py:15-26,46) fused for the single-input-channel case.  Its output is the largest tensor of the
// network (B x 80 x 427 x 64 = 280 MB at B=32) and every pass over it is HBM time, while the convolution itself is 9
// MACs per element.  Two facts make a one-pass forward possible:
//   * y[c] = sum_t w[c][t] * x_t  (x_t = the 9 zero-padded neighbours of the mel pixel), hence
//       sum_pos y[c]   = sum_t w[c][t] S_t                    S_t    = sum_pos x_t
//       sum_pos y[c]^2 = sum_{t,u} w[c][t] w[c][u] R_{t,u}    R_{t,u} = sum_pos x_t x_u
//     i.e. the per-(clip, channel) statistics of ALL 64 channels follow from 9 + 45 moments of the 4.4 MB mel image
//     (accumulated in fp64); the normalised, activated output is then written exactly once, and the raw conv output is
//     never stored;
//   * in backward the normalised pre-activation is recomputed from the mel image (36 FMAs per 4 channels) instead of
//     being re-read: statistics of the incoming gradient and the weight gradient each read dz once.
// Forward: l0_moments -> l0_finalize -> l0_fwd (1 write of 280 MB).  Backward: l0_bwd_stats (1 read) -> l0_dw (1 read).
#include "common.h"

#define L0_C 64
#define L0_T 9
#define L0_NMOM 54  // 9 first moments + 45 upper-triangular second moments

__device__ __forceinline__ void l0_gather(const float* __restrict__ mel, int H, int W, int y, int x, float (&nb)[L0_T]) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            nb[(dy + 1) * 3 + (dx + 1)] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? mel[(size_t)yy * W + xx] : 0.f;
        }
}

// Sliding 3x3 window along one image row: each step shifts the window one pixel to the right and loads the 3 new
// right-column values (zero outside the image) -- 3 loads per pixel instead of 9 and no integer division.
struct L0Window {
    float nb[L0_T];
    const float* r0;
    const float* r1;
    const float* r2;  // rows y-1, y, y+1 (nullptr when outside the image)
    int W;
    __device__ __forceinline__ float at(const float* r, int x) const { return (r != nullptr && (unsigned)x < (unsigned)W) ? r[x] : 0.f; }
    __device__ __forceinline__ void init(const float* img, int H, int W_, int y, int x) {
        W = W_;
        r0 = (y - 1 >= 0) ? img + (size_t)(y - 1) * W : nullptr;
        r1 = img + (size_t)y * W;
        r2 = (y + 1 < H) ? img + (size_t)(y + 1) * W : nullptr;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            nb[d] = at(r0, x - 1 + d);
            nb[3 + d] = at(r1, x - 1 + d);
            nb[6 + d] = at(r2, x - 1 + d);
        }
    }
    __device__ __forceinline__ void advance(int xnew) {  // window now centred on xnew (= previous centre + 1)
        nb[0] = nb[1]; nb[1] = nb[2]; nb[2] = at(r0, xnew + 1);
        nb[3] = nb[4]; nb[4] = nb[5]; nb[5] = at(r1, xnew + 1);
        nb[6] = nb[7]; nb[7] = nb[8]; nb[8] = at(r2, xnew + 1);
    }
};

// grid (chunks, B); mom[b][54] doubles (zeroed by the caller)
__global__ __launch_bounds__(256) void l0_moments_kernel(const float* __restrict__ mel, double* __restrict__ mom, int H, int W,
                                                         int pix_per_block) {
    __shared__ double sM[L0_NMOM];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < L0_NMOM) sM[tid] = 0.0;
    __syncthreads();
    const float* img = mel + (size_t)b * H * W;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(H * W, p0 + pix_per_block);
    float acc[L0_NMOM];
#pragma unroll
    for (int i = 0; i < L0_NMOM; ++i) acc[i] = 0.f;
    for (int p = p0 + tid; p < p1; p += 256) {
        float nb[L0_T];
        l0_gather(img, H, W, p / W, p % W, nb);
        int k = L0_T;
#pragma unroll
        for (int t = 0; t < L0_T; ++t) {
            acc[t] += nb[t];
#pragma unroll
            for (int u = t; u < L0_T; ++u) acc[k++] += nb[t] * nb[u];
        }
    }
#pragma unroll
    for (int i = 0; i < L0_NMOM; ++i) {
        const double s = wave_sum_d((double)acc[i]);
        if ((tid & 63) == 0) atomicAdd(&sM[i], s);
    }
    __syncthreads();
    if (tid < L0_NMOM) atomicAdd(&mom[(size_t)b * L0_NMOM + tid], sM[tid]);
}

// one thread per (group, channel): mean / rstd from the moments (groups == B: InstanceNorm; groups == 1: BatchNorm)
__global__ void l0_finalize_kernel(const double* __restrict__ mom, const float* __restrict__ w, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                   int64_t* __restrict__ nbt, int B, int groups, double n_per_group, float eps, float momentum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * L0_C) return;
    const int g = i / L0_C, c = i % L0_C;
    double M[L0_NMOM];
    for (int k = 0; k < L0_NMOM; ++k) {
        double s = 0.0;
        if (groups == 1)
            for (int b = 0; b < B; ++b) s += mom[(size_t)b * L0_NMOM + k];
        else
            s = mom[(size_t)g * L0_NMOM + k];
        M[k] = s;
    }
    double wv[L0_T];
    for (int t = 0; t < L0_T; ++t) wv[t] = (double)w[c * L0_T + t];
    double s = 0.0, q = 0.0;
    int k = L0_T;
    for (int t = 0; t < L0_T; ++t) {
        s += wv[t] * M[t];
        for (int u = t; u < L0_T; ++u) q += (u == t ? 1.0 : 2.0) * wv[t] * wv[u] * M[k++];
    }
    const double m = s / n_per_group;
    double var = q / n_per_group - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean != nullptr && groups == 1) {
        const double unb = n_per_group > 1.0 ? n_per_group / (n_per_group - 1.0) : 1.0;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unb);
        if (nbt != nullptr && c == 0) nbt[0] += 1;
    }
}

// thread = (pixel, channel quad): 16 threads per pixel, 16 pixels per 256-thread pass
struct L0Thread {
    float w[4][L0_T];
    f32x4 mu, rs, ga, be;
};
__device__ __forceinline__ void l0_setup(L0Thread& t, const float* __restrict__ w, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int g, int cq) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < L0_T; ++k) t.w[e][k] = w[(4 * cq + e) * L0_T + k];
    t.mu = *(const f32x4*)(mean + (size_t)g * L0_C + 4 * cq);
    t.rs = *(const f32x4*)(rstd + (size_t)g * L0_C + 4 * cq);
    t.ga = gamma ? *(const f32x4*)(gamma + 4 * cq) : (f32x4){1.f, 1.f, 1.f, 1.f};
    t.be = beta ? *(const f32x4*)(beta + 4 * cq) : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 l0_yhat(const L0Thread& t, const float (&nb)[L0_T]) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < L0_T; ++k) s = fmaf(nb[k], t.w[e][k], s);
        y[e] = (s - t.mu[e]) * t.rs[e];
    }
    return y;
}

// grid (H, B): one image row per workgroup; thread = (segment of the row, channel quad): 16 segments x 16 quads
__global__ __launch_bounds__(256) void l0_fwd_kernel(const float* __restrict__ mel, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ z, int H, int W, int groups, float slope) {
    const int b = blockIdx.y, y = blockIdx.x, tid = threadIdx.x, cq = tid & 15, seg = tid >> 4;
    L0Thread t;
    l0_setup(t, w, mean, rstd, gamma, beta, groups == 1 ? 0 : b, cq);
    const int len = (W + 15) / 16, x0 = seg * len, x1 = min(W, x0 + len);
    if (x0 >= x1) return;
    float* out = z + ((size_t)(b * H + y) * W) * L0_C + 4 * cq;
    L0Window win;
    win.init(mel + (size_t)b * H * W, H, W, y, x0);
    for (int x = x0; x < x1; ++x) {
        if (x > x0) win.advance(x);
        const f32x4 yh = l0_yhat(t, win.nb);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_fwd(yh[e] * t.ga[e] + t.be[e], slope);
        *(f32x4*)(out + (size_t)x * L0_C) = o;
    }
}

// sums[g][c][2] doubles (zeroed by the caller): sum g, sum g*yhat with g = dz * act'(gamma*yhat+beta)
__global__ __launch_bounds__(256) void l0_bwd_stats_kernel(const float* __restrict__ dz, const float* __restrict__ mel,
                                                           const float* __restrict__ w, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, double* __restrict__ sums, int H,
                                                           int W, int groups, float slope, int rows_per_block) {
    __shared__ double sS[L0_C], sQ[L0_C];
    const int b = blockIdx.y, tid = threadIdx.x, cq = tid & 15, seg = tid >> 4;
    const int y_beg = blockIdx.x * rows_per_block, y_end = min(H, y_beg + rows_per_block);
    const int g = groups == 1 ? 0 : b;
    if (tid < L0_C) sS[tid] = 0.0, sQ[tid] = 0.0;
    __syncthreads();
    L0Thread t;
    l0_setup(t, w, mean, rstd, gamma, beta, g, cq);
    const int len = (W + 15) / 16, x0 = seg * len, x1 = min(W, x0 + len);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int y = y_beg; y < y_end && x0 < x1; ++y) {
        const float* gin = dz + ((size_t)(b * H + y) * W) * L0_C + 4 * cq;
        L0Window win;
        win.init(mel + (size_t)b * H * W, H, W, y, x0);
        for (int xb = x0; xb < x1; xb += 4) {  // 4 gradient vectors in flight per thread (HBM latency hiding)
            f32x4 gzv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gzv[j] = (xb + j < x1) ? *(const f32x4*)(gin + (size_t)(xb + j) * L0_C) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xb + j;
                if (x < x1) {
                    if (x > x0) win.advance(x);
                    const f32x4 yh = l0_yhat(t, win.nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gg = gzv[j][e] * act_grad(yh[e] * t.ga[e] + t.be[e], slope);
                        s[e] += gg;
                        q[e] += gg * yh[e];
                    }
                }
            }
        }
    }
    // the 4 segment-lanes of a wave that share a channel quad first (bits 4,5 of the lane id), then LDS / global fp64
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double sd = (double)s[e], qd = (double)q[e];
        sd += __shfl_xor(sd, 16, 64);
        sd += __shfl_xor(sd, 32, 64);
        qd += __shfl_xor(qd, 16, 64);
        qd += __shfl_xor(qd, 32, 64);
        if ((tid & 48) == 0) {
            atomicAdd(&sS[4 * cq + e], sd);
            atomicAdd(&sQ[4 * cq + e], qd);
        }
    }
    __syncthreads();
    if (tid < L0_C) {
        atomicAdd(&sums[((size_t)g * L0_C + tid) * 2], sS[tid]);
        atomicAdd(&sums[((size_t)g * L0_C + tid) * 2 + 1], sQ[tid]);
    }
}

// dW[c][t] += sum_pos dy[c] * x_t  with dy = gamma*rstd*(g - mean_g - yhat*mean_gy); dgamma/dbeta accumulated (BN)
__global__ __launch_bounds__(256) void l0_dw_kernel(const float* __restrict__ dz, const float* __restrict__ mel,
                                                    const float* __restrict__ w, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const double* __restrict__ sums,
                                                    float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    int H, int W, int groups, double n_per_group, float slope,
                                                    int rows_per_block) {
    __shared__ float sD[L0_C * L0_T];
    const int b = blockIdx.y, tid = threadIdx.x, cq = tid & 15, pl = tid >> 4;
    const int y_beg = blockIdx.x * rows_per_block, y_end = min(H, y_beg + rows_per_block);
    const int g = groups == 1 ? 0 : b;
    for (int i = tid; i < L0_C * L0_T; i += 256) sD[i] = 0.f;
    __syncthreads();
    L0Thread t;
    l0_setup(t, w, mean, rstd, gamma, beta, g, cq);
    f32x4 mg, mgy;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double sg = sums[((size_t)g * L0_C + 4 * cq + e) * 2], sgy = sums[((size_t)g * L0_C + 4 * cq + e) * 2 + 1];
        mg[e] = (float)(sg / n_per_group);
        mgy[e] = (float)(sgy / n_per_group);
        if (blockIdx.x == 0 && blockIdx.y == 0 && pl == 0) {
            if (dgamma) atomicAdd(&dgamma[4 * cq + e], (float)sgy);
            if (dbeta) atomicAdd(&dbeta[4 * cq + e], (float)sg);
        }
    }
    const int len = (W + 15) / 16, x0 = pl * len, x1 = min(W, x0 + len);
    float acc[4][L0_T];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < L0_T; ++k) acc[e][k] = 0.f;
    for (int y = y_beg; y < y_end && x0 < x1; ++y) {
        const float* gin = dz + ((size_t)(b * H + y) * W) * L0_C + 4 * cq;
        L0Window win;
        win.init(mel + (size_t)b * H * W, H, W, y, x0);
        for (int xb = x0; xb < x1; xb += 4) {
            f32x4 gzv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gzv[j] = (xb + j < x1) ? *(const f32x4*)(gin + (size_t)(xb + j) * L0_C) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xb + j;
                if (x < x1) {
                    if (x > x0) win.advance(x);
                    const f32x4 yh = l0_yhat(t, win.nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gg = gzv[j][e] * act_grad(yh[e] * t.ga[e] + t.be[e], slope);
                        const float dy = t.ga[e] * t.rs[e] * (gg - mg[e] - yh[e] * mgy[e]);
#pragma unroll
                        for (int k = 0; k < L0_T; ++k) acc[e][k] = fmaf(dy, win.nb[k], acc[e][k]);
                    }
                }
            }
        }
    }
    // reduce over the 16 pixel lanes that share a channel quad: lanes tid = pl*16 + cq -> xor over bits 4,5 inside a
    // wave, then LDS atomics across the 4 waves
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < L0_T; ++k) {
            float v = acc[e][k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if ((tid & 48) == 0) atomicAdd(&sD[(4 * cq + e) * L0_T + k], v);
        }
    __syncthreads();
    for (int i = tid; i < L0_C * L0_T; i += 256) atomicAdd(&dW[i], sD[i]);
}

// ---------------------------------------------------------------------------------------------
static int l0_ppb(int HW) { return std::max(1024, std::min(4096, cdiv(HW, 8) / 256 * 256)); }  // moments kernel only

extern "C" int sdt_l0_block_fwd_f32(const float* mel, const float* w, float* z, double* mom, float* mean, float* rstd,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, int B, int H, int W, int groups, float eps, float momentum,
                                    float slope, void* stream) {
    SDT_CHECK_ARG(mel && w && z && mom && mean && rstd, "null pointer");
    SDT_CHECK_ARG(B > 0 && H > 0 && W > 0 && (groups == B || groups == 1), "bad dims (groups must be B or 1)");
    SDT_CHECK_ARG((int64_t)B * H * W * L0_C * 4 < (1ll << 40), "tensor too large");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W, ppb = l0_ppb(HW);
    dim3 grid(cdiv(HW, ppb), B);
    hipLaunchKernelGGL(l0_moments_kernel, grid, dim3(256), 0, s, mel, mom, H, W, ppb);
    const double n = groups == 1 ? (double)B * HW : (double)HW;
    hipLaunchKernelGGL(l0_finalize_kernel, dim3(cdiv(groups * L0_C, 64)), dim3(64), 0, s, mom, w, mean, rstd, running_mean,
                       running_var, num_batches_tracked, B, groups, n, eps, momentum);
    hipLaunchKernelGGL(l0_fwd_kernel, dim3(H, B), dim3(256), 0, s, mel, w, mean, rstd, gamma, beta, z, H, W, groups, slope);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_l0_block_bwd_f32(const float* dz, const float* mel, const float* w, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, double* sums, float* dw, float* dgamma,
                                    float* dbeta, int B, int H, int W, int groups, float slope, void* stream) {
    SDT_CHECK_ARG(dz && mel && w && mean && rstd && sums && dw, "null pointer");
    SDT_CHECK_ARG(B > 0 && H > 0 && W > 0 && (groups == B || groups == 1), "bad dims (groups must be B or 1)");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    // several image rows per workgroup: ~2 workgroups per CU keeps HBM busy while bounding the number of workgroups
    // that push their partial sums through the same global atomics
    const int rpb = std::max(1, (H * B) / 1280);
    dim3 grid(cdiv(H, rpb), B);
    hipLaunchKernelGGL(l0_bwd_stats_kernel, grid, dim3(256), 0, s, dz, mel, w, mean, rstd, gamma, beta, sums, H, W, groups, slope, rpb);
    const double n = groups == 1 ? (double)B * HW : (double)HW;
    hipLaunchKernelGGL(l0_dw_kernel, grid, dim3(256), 0, s, dz, mel, w, mean, rstd, gamma, beta, sums, dw, dgamma, dbeta, H, W,
                       groups, n, slope, rpb);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
