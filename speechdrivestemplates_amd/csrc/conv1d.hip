// The Conv1d stage of the sdt generator (U-Net + decoder, generator.py:70-85,98-116) as ONE launch per layer and direction.
//
// The stage is latency-bound: M = B*T <= 2048 output rows cannot fill 256 CUs with 64x64 tiles, so round 1 split the K loop over
// workgroups (slabs + a reduce pass) and ran conv -> reduce+norm (-> upsample-add) as 2-3 dependent launches per layer, ~11 us
// each including the launch boundary: 452 us forward + 641 us backward exposed on the step's critical path for 0.24 GFLOP per clip.
// Here a layer is one launch:
//   * 32 x 64 output tiles, full K per workgroup -> 256 workgroups at T = 64 without any split; inside a workgroup 8 waves =
//     2 (rows) x 2 (cols) x 2 (halves of each 64-wide K step) on v_mfma_f32_16x16x4_f32 (exact fp32), K halves combined through LDS;
//   * the per-(b,t) normalisation over channels (InstanceNorm1d on the permuted tensor, building_blocks.py:50-51) is never a pass
//     of its own: the producing launch stores the RAW conv output y plus, per row, partial (sum, sum of squares) of its 64
//     columns; the CONSUMING launch derives mean / rstd of the rows it needs in its prologue and normalises + LeakyReLUs while it
//     stages the A tile ("normalise on load").  The linear x2 upsample + skip add in front of the decoder convs (generator.py:79-83)
//     happens in the same loader;
//   * backward: the input gradient dz of a layer is the same GEMM with the mirrored weights; its A operand
//     gy = rstd * (g - mean(g) - yhat * mean(g * yhat)),  g = dz * act'(yhat),  is formed on load from (dz, y) and per-row
//     (mean, rstd, sum g, sum g*yhat); the epilogue adds the skip-path gradient where there is one and emits the partial
//     (sum g, sum g*yhat) of the layer below, so the chain needs no normalisation-backward pass either.
// Weight gradients stay on the side stream with the generic kernels (conv.hip), fed by materialised z / gy tensors that are also
// produced off the critical path (c1d_rownorm_partials_kernel, rownorm_kernel<BWD>).
#include <stdlib.h>

#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

#define C1_BM 32
#define C1_BN 64
#define C1_BK 64
#define C1_PITCH 68      // floats per LDS row of a [row][64] tile (16-byte aligned rows, 4-float skew against bank conflicts)
#define C1_NPART_MAX 16  // partial statistics per row: Cout / 64 <= 16
#define C1_MAXROWS 80    // input rows a tile can touch: 32 * stride + taps - 1 <= 67; 2 source resolutions in the upsample mode

enum { C1_IN_PLAIN = 0, C1_IN_NORM = 1, C1_IN_UPADD = 2, C1_IN_NORMBWD = 3 };

// per-row (mean, rstd) from the NP partial (sum, sumsq) pairs of a raw conv output row
__device__ __forceinline__ void c1_row_stats(const float* __restrict__ ps, int np, int C, float eps, float& mean, float& rstd) {
    float s = 0.f, q = 0.f;
    for (int i = 0; i < np; ++i) {
        s += ps[2 * i];
        q += ps[2 * i + 1];
    }
    mean = s / (float)C;
    float var = q / (float)C - mean * mean;
    var = var > 0.f ? var : 0.f;
    rstd = 1.f / sqrtf(var + eps);
}

// Y[m][n] = (bias[n]) + sum_{t, c} A(m, t, c) * W[n][t][c]          m = (b, to), A = the (transformed) input row at time to*s + t - p
//   in_mode C1_IN_PLAIN  : A = X[b, ti, c]
//           C1_IN_NORM   : A = act((X - mean) * rstd), row statistics from xstats (partials of the layer that produced X raw)
//           C1_IN_UPADD  : A = lerp2(act(norm(X2)))[ti] + act(norm(X))[ti]   (X2 at half resolution T2 with x2stats; X = skip at Ti)
//           C1_IN_NORMBWD: transposed-conv form for the input gradient: the "input" is the gradient dz (X) of a layer's OUTPUT
//                          (B, Ti = To_layer, Cin = Cout_layer), A = gy on load from (X = dz, X2 = y of that layer, xstats = fwd
//                          partials, x2stats = backward partials (sum g, sum g*yhat)); W = mirrored weights (Cout_gemm = Cin_layer);
//                          output time index to -> contributing input time (to + p - t) / s when divisible
// epilogue: bias (nullable); add (nullable, same shape as Y) is added to the result; ystats (nullable) receives per row and
// 64-column tile the partial (sum, sumsq) of the stored values (forward) -- or, when bw_y != nullptr, the partial
// (sum g, sum g * yhat) of the layer whose raw output is bw_y (shape of Y) with forward partials bw_stats.
struct c1d_args {
    const float* X;
    const float* X2;
    const float* xstats;
    const float* x2stats;
    const float* W;
    const float* bias;
    const float* add;
    float* Y;
    float* ystats;
    const float* bw_y;
    const float* bw_stats;
    int B, Ti, T2, Cin, To, Cout, taps, stride, pad;
    int in_mode, np_in, np_in2, np_bw;
    float eps, slope;
};

__global__ __launch_bounds__(512) void c1d_kernel(const c1d_args a) {
    __shared__ __attribute__((aligned(16))) float sA[C1_BM * C1_PITCH];
    __shared__ __attribute__((aligned(16))) float sB[C1_BN * C1_PITCH];
    __shared__ float sStat[2][C1_MAXROWS][4];  // per source row: mean, rstd, (s1 / C, s2 / C in the backward mode)
    __shared__ float sPart[2][C1_BM][2];       // per wn: row partials of the epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int M = a.B * a.To;
    const int nnb = (a.Cout + C1_BN - 1) / C1_BN;
    const int m0 = (blockIdx.x / nnb) * C1_BM, n0 = (blockIdx.x % nnb) * C1_BN;
    const int Ktot = a.taps * a.Cin;
    const int nit = (Ktot + C1_BK - 1) / C1_BK;
    const bool bwd = a.in_mode == C1_IN_NORMBWD;

    // ---- prologue: statistics of the source rows this tile touches.  Rows of the primary source: a contiguous window per batch
    // item; the tile's 32 output rows may straddle two batch items, so the table is indexed by (global row - first global row).
    const int b_first = m0 / a.To, to_first = m0 % a.To;
    int row_lo, row_lo2 = 0;
    if (!bwd) {
        row_lo = b_first * a.Ti + max(0, to_first * a.stride - a.pad);
    } else {  // contributing layer-output rows for input rows [to_first, ...): (to + p - t) / s
        row_lo = b_first * a.Ti + max(0, (to_first + a.pad - (a.taps - 1)) / a.stride);
    }
    if (a.in_mode == C1_IN_UPADD) row_lo2 = b_first * a.T2 + max(0, (max(0, to_first * a.stride - a.pad)) / 2 - 1);
    if (a.in_mode != C1_IN_PLAIN) {
        const int total = a.B * a.Ti;
        for (int j = tid; j < C1_MAXROWS; j += 512) {
            const int r = row_lo + j;
            if (r < total) {
                float mu, rs;
                c1_row_stats(a.xstats + (size_t)r * a.np_in * 2, a.np_in, a.Cin, a.eps, mu, rs);
                sStat[0][j][0] = mu;
                sStat[0][j][1] = rs;
                if (bwd) {
                    float s1 = 0.f, s2 = 0.f;
                    for (int i = 0; i < a.np_in2; ++i) {
                        s1 += a.x2stats[((size_t)r * a.np_in2 + i) * 2];
                        s2 += a.x2stats[((size_t)r * a.np_in2 + i) * 2 + 1];
                    }
                    sStat[0][j][2] = s1 / (float)a.Cin;
                    sStat[0][j][3] = s2 / (float)a.Cin;
                }
            }
        }
        if (a.in_mode == C1_IN_UPADD) {
            const int total2 = a.B * a.T2;
            for (int j = tid; j < C1_MAXROWS; j += 512) {
                const int r = row_lo2 + j;
                if (r < total2) {
                    float mu, rs;
                    c1_row_stats(a.x2stats + (size_t)r * a.np_in2 * 2, a.np_in2, a.Cin, a.eps, mu, rs);
                    sStat[1][j][0] = mu;
                    sStat[1][j][1] = rs;
                }
            }
        }
    }

    // ---- loader mapping: A: 32 rows x 16 float4 = 512 threads; B: 64 rows x 16 float4 = 2 per thread
    const int lr = tid >> 4, lq = tid & 15;
    const int am = m0 + lr;
    const bool am_ok = am < M;
    const int ab = am_ok ? am / a.To : 0, ato = am_ok ? am % a.To : 0;
    const int bn0 = n0 + lr, bn1 = n0 + lr + 32;
    __syncthreads();

    auto act = [&](float u) { return u > 0.f ? u : u * a.slope; };
    // one float4 of the A operand: flattened k -> (tap t, channel c..c+3) of output row (ab, ato)
    auto load_a = [&](int k) -> f32x4v {
        f32x4v v = {0.f, 0.f, 0.f, 0.f};
        if (!am_ok || k >= Ktot) return v;
        const int t = k / a.Cin, c = k - t * a.Cin;
        if (!bwd) {
            const int ti = ato * a.stride + t - a.pad;
            if ((unsigned)ti >= (unsigned)a.Ti) return v;
            const int r = ab * a.Ti + ti;
            v = *(const f32x4v*)(a.X + (size_t)r * a.Cin + c);
            if (a.in_mode == C1_IN_PLAIN) return v;
            const float mu = sStat[0][r - row_lo][0], rs = sStat[0][r - row_lo][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act((v[e] - mu) * rs);
            if (a.in_mode == C1_IN_UPADD) {
                // F.interpolate(prev, Ti, mode='linear', align_corners=False) at output index ti (generator.py:79-83)
                const float src = ((float)ti + 0.5f) * ((float)a.T2 / (float)a.Ti) - 0.5f;
                const float sc = src < 0.f ? 0.f : src;
                int i0 = (int)sc;
                i0 = i0 < a.T2 - 1 ? i0 : a.T2 - 1;
                const int i1 = i0 + 1 < a.T2 ? i0 + 1 : a.T2 - 1;
                const float w1 = sc - (float)i0, w0 = 1.f - w1;
                const int r0 = ab * a.T2 + i0, r1 = ab * a.T2 + i1;
                const f32x4v p0 = *(const f32x4v*)(a.X2 + (size_t)r0 * a.Cin + c);
                const f32x4v p1 = *(const f32x4v*)(a.X2 + (size_t)r1 * a.Cin + c);
                const float m0_ = sStat[1][r0 - row_lo2][0], s0_ = sStat[1][r0 - row_lo2][1];
                const float m1_ = sStat[1][r1 - row_lo2][0], s1_ = sStat[1][r1 - row_lo2][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += w0 * act((p0[e] - m0_) * s0_) + w1 * act((p1[e] - m1_) * s1_);
            }
            return v;
        }
        // backward: the layer-output row that tap t maps output row ato to
        const int num = ato + a.pad - t;
        if (num < 0 || num % a.stride != 0) return v;
        const int ti = num / a.stride;
        if (ti >= a.Ti) return v;
        const int r = ab * a.Ti + ti;
        const f32x4v dz = *(const f32x4v*)(a.X + (size_t)r * a.Cin + c);
        const f32x4v yv = *(const f32x4v*)(a.X2 + (size_t)r * a.Cin + c);
        const float* st = sStat[0][r - row_lo];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float yh = (yv[e] - st[0]) * st[1];
            const float gg = dz[e] * (yh > 0.f ? 1.f : a.slope);
            v[e] = st[1] * (gg - st[2] - yh * st[3]);
        }
        return v;
    };
    auto load_b = [&](int n, int k) -> f32x4v {
        if (n >= a.Cout || k >= Ktot) return (f32x4v){0.f, 0.f, 0.f, 0.f};
        return *(const f32x4v*)(a.W + (size_t)n * Ktot + k);  // W is (Cout, taps, Cin): K-contiguous
    };

    // ---- K loop: one 64-wide step per iteration, wave half wk multiplies k in [32 wk, 32 wk + 32)
    f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4v ra = load_a(4 * lq), rb0 = load_b(bn0, 4 * lq), rb1 = load_b(bn1, 4 * lq);
    // fragment addresses: lane = (row l & 15, k-slot h = l >> 4); sub-step (j, e) consumes k = 16 j + 4 h + e for A and B alike
    const float* pa = sA + (wm * 16 + (lane & 15)) * C1_PITCH + wk * 32 + (lane >> 4) * 4;
    const float* pb = sB + (wn * 32 + (lane & 15)) * C1_PITCH + wk * 32 + (lane >> 4) * 4;
    for (int it = 0; it < nit; ++it) {
        *(f32x4v*)&sA[lr * C1_PITCH + 4 * lq] = ra;
        *(f32x4v*)&sB[lr * C1_PITCH + 4 * lq] = rb0;
        *(f32x4v*)&sB[(lr + 32) * C1_PITCH + 4 * lq] = rb1;
        __syncthreads();
        if (it + 1 < nit) {
            const int k = (it + 1) * C1_BK + 4 * lq;
            ra = load_a(k);
            rb0 = load_b(bn0, k);
            rb1 = load_b(bn1, k);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4v av = *(const f32x4v*)(pa + 16 * j);
            const f32x4v b0 = *(const f32x4v*)(pb + 16 * j);
            const f32x4v b1 = *(const f32x4v*)(pb + 16 * C1_PITCH + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b1[e], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- combine the two K halves through LDS (reuse sB: 4 (wm, wn) pairs x 2 tiles x 256 floats = 8 KB)
    float* red = sB;
    if (wk == 1) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) *(f32x4v*)&red[(((wm * 2 + wn) * 2 + tl) * 64 + lane) * 4] = acc[tl];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) acc[tl] += *(const f32x4v*)&red[(((wm * 2 + wn) * 2 + tl) * 64 + lane) * 4];

        // ---- epilogue (waves wk == 0).  C/D layout of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
        float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int n = n0 + wn * 32 + tl * 16 + (lane & 15);
            const bool nok = n < a.Cout;
            const float bv = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 16 + (lane >> 4) * 4 + r;
                const int m = m0 + row;
                if (m < M && nok) {
                    float v = acc[tl][r] + bv;
                    const size_t o = (size_t)m * a.Cout + n;
                    if (a.add != nullptr) v += a.add[o];
                    a.Y[o] = v;
                    if (a.ystats != nullptr) {
                        if (a.bw_y == nullptr) {
                            ps[r] += v;
                            pq[r] = fmaf(v, v, pq[r]);
                        } else {  // partial (sum g, sum g*yhat) of the layer below: v is the total gradient w.r.t. its activated output
                            float mu, rs;
                            c1_row_stats(a.bw_stats + (size_t)m * a.np_bw * 2, a.np_bw, a.Cout, a.eps, mu, rs);
                            const float yh = (a.bw_y[o] - mu) * rs;
                            const float gg = v * (yh > 0.f ? 1.f : a.slope);
                            ps[r] += gg;
                            pq[r] = fmaf(gg, yh, pq[r]);
                        }
                    }
                }
            }
        }
        if (a.ystats != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    ps[r] += __shfl_xor(ps[r], o, 64);
                    pq[r] += __shfl_xor(pq[r], o, 64);
                }
                if ((lane & 15) == 0) {
                    const int row = wm * 16 + (lane >> 4) * 4 + r;
                    sPart[wn][row][0] = ps[r];
                    sPart[wn][row][1] = pq[r];
                }
            }
        }
    }
    __syncthreads();  // every wave of the workgroup reaches this barrier (uniform control flow above)
    if (a.ystats != nullptr && tid < C1_BM) {
        const int m = m0 + tid;
        if (m < M) {
            float* d = a.ystats + ((size_t)m * nnb + (n0 / C1_BN)) * 2;
            d[0] = sPart[0][tid][0] + sPart[1][tid][0];
            d[1] = sPart[0][tid][1] + sPart[1][tid][1];
        }
    }
}

// z = act((y - mean) * rstd) with the row statistics taken from the partials; also stores mean / rstd (for the generic
// normalisation-backward / weight-gradient kernels that run on the side stream).  One wave per row, C <= 1024.
__global__ __launch_bounds__(256) void c1d_rownorm_partials_kernel(const float* __restrict__ y, const float* __restrict__ stats, int np,
                                                                   float* __restrict__ z, float* __restrict__ mean,
                                                                   float* __restrict__ rstd, int64_t rows, int C, float eps, float slope) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mu, rs;
    c1_row_stats(stats + (size_t)row * np * 2, np, C, eps, mu, rs);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
    for (int c = 4 * lane; c < C; c += 256) {
        const f32x4v v = *(const f32x4v*)(y + (size_t)row * C + c);
        f32x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_fwd((v[e] - mu) * rs, slope);
        *(f32x4v*)(z + (size_t)row * C + c) = o;
    }
}

// Adjoint of the linear x2 upsample (upsample_add backward, generator.py:79-83) fused with the statistics of the normalisation
// backward below it: dprev[b, i] = sum_j w(j -> i) g[b, j] and, with y / fwd partials of the layer that produced prev,
// bstats[row] = {sum gg, sum gg * yhat, 0...}.  One wave per (b, i) row of the low-resolution tensor.
__global__ __launch_bounds__(256) void c1d_upsample_bwd_stats_kernel(const float* __restrict__ g, float* __restrict__ dprev,
                                                                     const float* __restrict__ y, const float* __restrict__ stats, int np,
                                                                     float* __restrict__ bstats, int np_out, int B, int Ti, int To, int C,
                                                                     float eps, float slope) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Ti) return;
    const int b = row / Ti, i = row % Ti;
    float mu, rs;
    c1_row_stats(stats + (size_t)row * np * 2, np, C, eps, mu, rs);
    // output rows j whose interpolation touches input row i: j in [2i - 2, 2i + 2] for a x2 upsample; evaluate the forward map
    const float scale = (float)Ti / (float)To;
    const int jlo = max(0, (int)((float)(i - 1) / scale) - 2), jhi = min(To - 1, (int)((float)(i + 1) / scale) + 2);
    float s1 = 0.f, s2 = 0.f;
    for (int c = 4 * lane; c < C; c += 256) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = jlo; j <= jhi; ++j) {
            const float src = ((float)j + 0.5f) * scale - 0.5f;
            const float sc = src < 0.f ? 0.f : src;
            int i0 = (int)sc;
            i0 = i0 < Ti - 1 ? i0 : Ti - 1;
            const int i1 = i0 + 1 < Ti ? i0 + 1 : Ti - 1;
            const float w1 = sc - (float)i0, w0 = 1.f - w1;
            float w = 0.f;
            if (i0 == i) w += w0;
            if (i1 == i) w += w1;
            if (w != 0.f) {
                const f32x4v gv = *(const f32x4v*)(g + ((size_t)b * To + j) * C + c);
                acc += gv * w;
            }
        }
        *(f32x4v*)(dprev + (size_t)row * C + c) = acc;
        const f32x4v yv = *(const f32x4v*)(y + (size_t)row * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float yh = (yv[e] - mu) * rs;
            const float gg = acc[e] * act_grad(yh, slope);
            s1 += gg;
            s2 = fmaf(gg, yh, s2);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane < 2 * np_out) bstats[(size_t)row * np_out * 2 + lane] = lane == 0 ? s1 : (lane == 1 ? s2 : 0.f);
}

// ---------------------------------------------------------------------------------------------
extern "C" int sdt_c1d_layer_f32(const sdt_c1d* p, void* stream) {
    SDT_CHECK_ARG(p != nullptr, "null descriptor");
    SDT_CHECK_ARG(p->X && p->W && p->Y, "null pointer");
    SDT_CHECK_ARG(p->B > 0 && p->Ti > 0 && p->To > 0 && p->Cin > 0 && p->Cout > 0 && p->taps >= 1 && p->taps <= 8 && p->stride >= 1 && p->stride <= 2,
                  "bad dims");
    SDT_CHECK_ARG(p->Cin % 4 == 0, "Cin must be a multiple of 4");
    SDT_CHECK_ARG(p->in_mode >= C1_IN_PLAIN && p->in_mode <= C1_IN_NORMBWD, "unknown input mode");
    SDT_CHECK_ARG(p->in_mode == C1_IN_PLAIN || (p->xstats && p->np_in >= 1 && p->np_in <= C1_NPART_MAX), "input statistics missing");
    SDT_CHECK_ARG((p->in_mode != C1_IN_UPADD && p->in_mode != C1_IN_NORMBWD) || (p->X2 && p->x2stats && p->np_in2 >= 1 && p->np_in2 <= C1_NPART_MAX),
                  "second source missing");
    SDT_CHECK_ARG(p->in_mode != C1_IN_UPADD || (p->T2 > 0 && p->stride == 1), "upsample-add input needs T2 and stride 1");
    SDT_CHECK_ARG(p->ystats == nullptr || (p->Cout % C1_BN == 0 && p->Cout / C1_BN <= C1_NPART_MAX), "statistics epilogue needs Cout % 64 == 0");
    SDT_CHECK_ARG(p->bw_y == nullptr || (p->ystats && p->bw_stats && p->np_bw >= 1 && p->np_bw <= C1_NPART_MAX), "backward-statistics epilogue arguments");
    SDT_CHECK_ARG(C1_BM * p->stride + p->taps + 2 <= C1_MAXROWS, "tile touches too many input rows");
    SDT_CHECK_ARG((int64_t)p->B * p->To * p->Cout < (1ll << 31) && (int64_t)p->B * p->Ti * p->Cin < (1ll << 31), "tensor too large");
    c1d_args a;
    a.X = p->X; a.X2 = p->X2; a.xstats = p->xstats; a.x2stats = p->x2stats; a.W = p->W; a.bias = p->bias; a.add = p->add; a.Y = p->Y;
    a.ystats = p->ystats; a.bw_y = p->bw_y; a.bw_stats = p->bw_stats;
    a.B = p->B; a.Ti = p->Ti; a.T2 = p->T2; a.Cin = p->Cin; a.To = p->To; a.Cout = p->Cout; a.taps = p->taps; a.stride = p->stride; a.pad = p->pad;
    a.in_mode = p->in_mode; a.np_in = p->np_in; a.np_in2 = p->np_in2; a.np_bw = p->np_bw; a.eps = p->eps; a.slope = p->slope;
    const int M = p->B * p->To;
    const unsigned grid = (unsigned)(cdiv(M, C1_BM) * cdiv(p->Cout, C1_BN));
    hipLaunchKernelGGL(c1d_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_c1d_rownorm_partials_f32(const float* y, const float* stats, int np, float* z, float* mean, float* rstd, int64_t rows,
                                            int C, float eps, float slope, void* stream) {
    SDT_CHECK_ARG(y && stats && z && mean && rstd && rows > 0 && np >= 1 && np <= C1_NPART_MAX, "bad argument");
    SDT_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= 1024, "C must be a multiple of 4 in [4,1024]");
    hipLaunchKernelGGL(c1d_rownorm_partials_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, stats, np, z, mean,
                       rstd, rows, C, eps, slope);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_c1d_upsample_bwd_stats_f32(const float* g, float* dprev, const float* y, const float* stats, int np, float* bstats,
                                              int np_out, int B, int Ti, int To, int C, float eps, float slope, void* stream) {
    SDT_CHECK_ARG(g && dprev && y && stats && bstats && B > 0 && Ti > 0 && To > 0, "bad argument");
    SDT_CHECK_ARG(np >= 1 && np <= C1_NPART_MAX && np_out >= 1 && np_out <= C1_NPART_MAX && C % 4 == 0 && C <= 1024, "bad argument");
    hipLaunchKernelGGL(c1d_upsample_bwd_stats_kernel, dim3((unsigned)cdiv(B * Ti, 4)), dim3(256), 0, (hipStream_t)stream, g, dprev, y, stats,
                       np, bstats, np_out, B, Ti, To, C, eps, slope);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
