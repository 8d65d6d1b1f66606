// fp32-equivalent convolution products on the bf16 MFMA from PRE-SPLIT operands ("bf16x6" arithmetic without the in-kernel split).
//
// An fp32 value x is split EXACTLY into three bf16 pieces by truncation, x = x1 + x2 + x3 (8 + 8 + 8 significand bits; see
// split3 in common.h), and a product a*b is evaluated as the six bf16 x bf16 MFMA products
//      a1b1 | a1b2 + a2b1 + a2b2 + a1b3 + a3b1            (dropped: a2b3 + a3b2 + a3b3 < 2^-23 |ab|)
// accumulated in fp32 (large term and corrections in separate accumulators, added once at the end) -- the arithmetic of
// conv_taps_bf_kernel<6> (conv.hip), which passes the same float64-calibrated parity bar as the exact-fp32 MFMA path
// (tests/test_fullsize_gpu.py).  Round 1 split both operand tiles inside the conv kernel, 5.5 VALU operations per element per
// K step, and was VALU-bound (+11 % only).  Here the split is done ONCE by the kernel that produces a tensor:
//   activations   colnorm_apply_fwd / l0_fwd write z as three bf16 planes next to the fp32 tensor        (norm.hip, l0.hip)
//   gradients     colnorm_apply_bwd writes dy as three bf16 planes                                         (norm.hip)
//   weights       weight_planes_batched_kernel: W and its (Cin,taps,Cout) mirror, once per optimiser step  (here)
// and conv_taps_pre_kernel below is a pure bf16-MFMA implicit GEMM: 16-byte plane loads -> swizzled LDS -> ds_read_b128
// fragments -> 6 MFMAs per (32x32x16) block.  A "planes" tensor is [3][numel] bf16, plane-major.
//
// Same tap-table contract, tap culling, XCD remap, multi-class launches and statistics epilogues as conv_taps_kernel.
#include <stdlib.h>

#include "common.h"

#define BKP 32  // K chunk (channels) per step
#define SDT_OOB 0x80000000u
#define SDT_MAX_CLASSES 4

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct geom_pack_p {
    sdt_conv_geom g[SDT_MAX_CLASSES];
};
struct norm_bwd_args_p {
    const float* y;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    double* sums;
    float slope;
    int groups;
};

// element offset of the 16-byte chunk c (0..3) of row r in a [rows][32] bf16 tile: chunk position XOR-swizzled by (r >> 2) & 3
// (conflict-free for the 16-byte stores -- 8 lanes cover two rows = one 128-byte bank window -- and for the ds_read_b128
// fragment reads, see conv.hip bf_tile_off)
__device__ __forceinline__ int ptile_off(int row, int c) { return row * BKP + ((c ^ ((row >> 2) & 3)) << 3); }

// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void conv_taps_pre_kernel(const __bf16* __restrict__ Xp, const size_t xplane,
                                                            const __bf16* __restrict__ Wp, const size_t wplane,
                                                            float* __restrict__ Y, const geom_pack_p gp,
                                                            double* __restrict__ stats, const int rows_per_group,
                                                            const norm_bwd_args_p nb) {
    constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 accumulator tiles per wave (2x2 waves)
    constexpr int RA = BM / 64, RB = BN / 64;   // 16-row groups per wave: wave w stages rows [16 (w + 4 i), +16) of each operand tile
    constexpr int PLANEA = BM * BKP, PLANEB = BN * BKP;
    // Staging is asynchronous global -> LDS (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass (the LDS write
    // port was the bound of the register-staged form: 24 KB of ds_write_b128 per K step at ~80 B/clk), two LDS buffers and ONE
    // barrier per K step.  An LDS-DMA writes wave-uniform base + lane * 16 B, so a wave instruction fills 16 rows x 64 B of one
    // plane and the bank swizzle moves to the SOURCE side: lane (row, position q) fetches chunk q ^ ((row >> 2) & 3) of its row.
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3 * PLANEA];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3 * PLANEB];
    __shared__ int sOut[BM];
    __shared__ int sTap[3 * SDT_MAX_TAPS];
    __shared__ int sLive[SDT_MAX_TAPS + 1];

    const sdt_conv_geom& gt = gp.g[blockIdx.y];
    struct {
        int B, Hi, Wi, Cin, Ho, Wo, Hy, Wy, Cout, sy, sx, osy, osx, ooy, oox, ntaps, Tw;
    } g;
#define SDT_SGPR(f) g.f = __builtin_amdgcn_readfirstlane(gt.f)
    SDT_SGPR(B); SDT_SGPR(Hi); SDT_SGPR(Wi); SDT_SGPR(Cin); SDT_SGPR(Ho); SDT_SGPR(Wo); SDT_SGPR(Hy); SDT_SGPR(Wy); SDT_SGPR(Cout);
    SDT_SGPR(sy); SDT_SGPR(sx); SDT_SGPR(osy); SDT_SGPR(osx); SDT_SGPR(ooy); SDT_SGPR(oox); SDT_SGPR(ntaps); SDT_SGPR(Tw);
#undef SDT_SGPR

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = g.B * g.Ho * g.Wo;
    const int nmb = (M + BM - 1) / BM;
    const int nnb = (g.Cout + BN - 1) / BN;
    if ((int)blockIdx.x >= nmb * nnb) return;
    const int lin = xcd_remap(blockIdx.x, nmb * nnb);
    const int m0 = (lin / nnb) * BM;
    const int n0 = (lin % nnb) * BN;

    if (tid < g.ntaps) {
        sTap[tid] = gt.dy[tid];
        sTap[SDT_MAX_TAPS + tid] = gt.dx[tid];
        sTap[2 * SDT_MAX_TAPS + tid] = gt.wt[tid];
    }
    if (tid <= SDT_MAX_TAPS) sLive[tid] = 0;
    if (tid < BM) {
        int m = m0 + tid, off = -1;
        if (m < M) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            off = ((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout;
        }
        sOut[tid] = off;
    }
    // loader mapping: lane = (row within the 16-row group, 16-byte position); wave w owns the row groups w + 4 i
    const int kc4 = lane & 3, r0 = wave * 16 + (lane >> 2);
    int rbH[RA], riy[RA], rix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 64 * i;
        if (m < M) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            rbH[i] = b * g.Hi;
            riy[i] = oy * g.sy;
            rix[i] = ox * g.sx;
        } else {
            rbH[i] = 0;
            riy[i] = -(1 << 20);
            rix[i] = 0;
        }
    }
    __syncthreads();
    if (kc4 == 0) {  // tap culling (conv_taps_kernel): drop taps whose inputs are out of range for every row of this tile
        for (int t = 0; t < g.ntaps; ++t) {
            const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t];
            bool any = false;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                any |= (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
            if (any) sLive[t] = 1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int t = 0; t < g.ntaps; ++t)
            if (sLive[t]) sLive[n++] = t;
        sLive[SDT_MAX_TAPS] = n;
    }
    __syncthreads();
    const int ntl = sLive[SDT_MAX_TAPS];
    const int nkc = g.Cin / BKP;
    const int nsteps = ntl * nkc;

    // one buffer resource per operand covering all three planes; byte offsets; masked rows use SDT_OOB (hardware returns zeros)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)Xp, 0, (int)(3u * (unsigned)xplane * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)(3u * (unsigned)wplane * 2u), 0x00020000);
    const int xps = (int)((unsigned)xplane * 2u), wps = (int)((unsigned)wplane * 2u);  // plane strides in bytes (uniform)
    unsigned aoff[RA], boff[RB], abase[RA], bbase[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) abase[i] = (unsigned)(((rbH[i] + riy[i]) * g.Wi + rix[i]) * g.Cin) * 2u;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 64 * i;
        bbase[i] = n < g.Cout ? (unsigned)(n * g.Tw * g.Cin) * 2u : SDT_OOB;
    }
    int cur_tl = -1, nxt_tl = 0, nxt_kc = 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned swz = (unsigned)((kc4 ^ ((r0 >> 2) & 3)) * 16);  // byte offset of the chunk this lane fetches (rows r0 + 64 i share it)
    auto issue = [&](int buf) {
        const int tl = nxt_tl, kc = nxt_kc;
        if (++nxt_kc == nkc) nxt_kc = 0, ++nxt_tl;
        const int cs = kc * BKP * 2;
        if (tl != cur_tl) {
            cur_tl = tl;
            const int t = sLive[tl];
            const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
            const unsigned ashift = (unsigned)((dy * g.Wi + dx) * g.Cin) * 2u + swz;
            const unsigned bshift = (unsigned)(wt * g.Cin) * 2u + swz;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                aoff[i] = ok ? abase[i] + ashift : SDT_OOB;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) boff[i] = bbase[i] == SDT_OOB ? SDT_OOB : bbase[i] + bshift;
        }
        // each instruction: 1 KiB = 16 rows x 64 B of one plane, lane * 16 B apart, at rows 16 (wave + 4 i) of the tile
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(&sA[buf][p * PLANEA + (wave * 16 + 64 * i) * BKP]), 16, (int)aoff[i],
                                                         cs + p * xps, 0, 0);
#pragma unroll
            for (int i = 0; i < RB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(&sB[buf][p * PLANEB + (wave * 16 + 64 * i) * BKP]), 16, (int)boff[i],
                                                         cs + p * wps, 0, 0);
        }
    };

    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f, accl[i][j][r] = 0.f;

    // fragment reads: MFMA k-slot e of lane half h <-> k = 16j + 8h + e, i.e. the 16-byte chunk 2j + h of row (.. + lane & 31)
    const int fsw = (lane >> 2) & 3, fh = lane >> 5;
    const int foff[2] = {((fh ^ fsw) << 3), (((2 + fh) ^ fsw) << 3)};
    const int rowA = (wm * (BM / 2) + (lane & 31)) * BKP, rowB = (wn * (BN / 2) + (lane & 31)) * BKP;

    if (nsteps > 0) issue(0);
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA pieces of `step` have landed
        __syncthreads();                      // ... everyone's have, and everyone is done reading the other buffer
        if (step + 1 < nsteps) issue(buf ^ 1);
        const __bf16* pa = &sA[buf][rowA];
        const __bf16* pb = &sB[buf][rowB];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm][p] = *(const bf16x8*)(pa + p * PLANEA + tm * 32 * BKP + foff[j]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn][p] = *(const bf16x8*)(pb + p * PLANEB + tn * 32 * BKP + foff[j]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], acc[tm][tn], 0, 0, 0);
                    accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], accl[tm][tn], 0, 0, 0);
                    accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], accl[tm][tn], 0, 0, 0);
                    accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], accl[tm][tn], 0, 0, 0);
                    accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], accl[tm][tn], 0, 0, 0);
                    accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], accl[tm][tn], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accl[i][j][r];

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)   (as conv_taps_kernel, incl. its statistics modes)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * (BN / 2) + tn * 32 + (lane & 31);
            const bool nok = n < g.Cout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int off = sOut[row];
                if (off >= 0 && nok) Y[(size_t)off + n] = acc[tm][tn][r];
            }
            if constexpr (EPI == 1) {  // forward statistics of the normalisation that follows
                const int g0 = m0 / rows_per_group;
                const int mb = (g0 + 1) * rows_per_group;
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (sOut[row] >= 0 && nok) {
                        const float v = acc[tm][tn][r];
                        if (m0 + row < mb) {
                            s0 += v;
                            q0 = fmaf(v, v, q0);
                        } else {
                            s1 += v;
                            q1 = fmaf(v, v, q1);
                        }
                    }
                }
                s0 += __shfl_xor(s0, 32, 64);
                q0 += __shfl_xor(q0, 32, 64);
                s1 += __shfl_xor(s1, 32, 64);
                q1 += __shfl_xor(q1, 32, 64);
                if (lane < 32 && nok) {
                    double* d = stats + ((size_t)g0 * g.Cout + n) * 2;
                    atomicAdd(d, (double)s0);
                    atomicAdd(d + 1, (double)q0);
                    if (mb < m0 + BM && mb < M) {
                        atomicAdd(d + 2 * (size_t)g.Cout, (double)s1);
                        atomicAdd(d + 2 * (size_t)g.Cout + 1, (double)q1);
                    }
                }
            }
            if constexpr (EPI == 2) {  // statistics of the normalisation backward that consumes this gradient
                const int rpg = nb.groups == 1 ? M : g.Ho * g.Wo;
                const int g0 = m0 / rpg;
                const int mb = (g0 + 1) * rpg;
                const bool two = mb < m0 + BM && mb < M;
                float mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f, ga = 1.f, be = 0.f;
                if (nok) {
                    mu0 = nb.mean[(size_t)g0 * g.Cout + n];
                    rs0 = nb.rstd[(size_t)g0 * g.Cout + n];
                    if (two) {
                        mu1 = nb.mean[(size_t)(g0 + 1) * g.Cout + n];
                        rs1 = nb.rstd[(size_t)(g0 + 1) * g.Cout + n];
                    }
                    if (nb.gamma != nullptr) ga = nb.gamma[n];
                    if (nb.beta != nullptr) be = nb.beta[n];
                }
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int off = sOut[row];
                    if (off >= 0 && nok) {
                        const float yv = nb.y[(size_t)off + n];
                        const bool second = m0 + row >= mb;
                        const float yh = (yv - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
                        const float gg = acc[tm][tn][r] * act_grad(yh * ga + be, nb.slope);
                        if (!second) {
                            s0 += gg;
                            q0 = fmaf(gg, yh, q0);
                        } else {
                            s1 += gg;
                            q1 = fmaf(gg, yh, q1);
                        }
                    }
                }
                s0 += __shfl_xor(s0, 32, 64);
                q0 += __shfl_xor(q0, 32, 64);
                s1 += __shfl_xor(s1, 32, 64);
                q1 += __shfl_xor(q1, 32, 64);
                if (lane < 32 && nok) {
                    double* d = nb.sums + ((size_t)g0 * g.Cout + n) * 2;
                    atomicAdd(d, (double)s0);
                    atomicAdd(d + 1, (double)q0);
                    if (two) {
                        atomicAdd(d + 2 * (size_t)g.Cout, (double)s1);
                        atomicAdd(d + 2 * (size_t)g.Cout + 1, (double)q1);
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// x (n floats) -> three bf16 planes (planes + p * n): standalone split, for tensors whose producer is not one of ours.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, size_t n) {
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(x + 4 * i);
        store_planes4(v, planes + 4 * i, n);
    }
}

// W (cout,taps,cin) -> planes of W and planes of its (cin,taps,cout) mirror, for many layers in one launch (same tiling as
// weight_transpose_batched_kernel; run right after the optimiser step, together with the fp32 mirrors).
__global__ __launch_bounds__(256) void weight_planes_batched_kernel(const sdt_wp_desc* __restrict__ table, int n_layers) {
    __shared__ float tile[32][33];
    const int bid = blockIdx.x;
    int l = 0;
    while (l + 1 < n_layers && table[l + 1].tile_begin <= bid) ++l;
    const sdt_wp_desc d = table[l];
    const int nci = (d.cin + 31) >> 5, nco = (d.cout + 31) >> 5;
    int rem = bid - d.tile_begin;
    const int ci0 = (rem % nci) * 32;
    rem /= nci;
    const int co0 = (rem % nco) * 32, t = rem / nco;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t plane = (size_t)d.cout * d.taps * d.cin;
    __bf16* wp = (__bf16*)d.wp;
    __bf16* wtp = (__bf16*)d.wtp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty + 8 * i, ci = ci0 + tx;
        const bool ok = co < d.cout && ci < d.cin;
        const float v = ok ? d.w[((size_t)co * d.taps + t) * d.cin + ci] : 0.f;
        tile[ty + 8 * i][tx] = v;
        if (ok) {
            __bf16 p0, p1, p2;
            split3(v, p0, p1, p2);
            const size_t o = ((size_t)co * d.taps + t) * d.cin + ci;
            wp[o] = p0;
            wp[plane + o] = p1;
            wp[2 * plane + o] = p2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty + 8 * i, co = co0 + tx;
        if (ci < d.cin && co < d.cout) {
            __bf16 p0, p1, p2;
            split3(tile[tx][ty + 8 * i], p0, p1, p2);
            const size_t o = ((size_t)ci * d.taps + t) * d.cout + co;
            wtp[o] = p0;
            wtp[plane + o] = p1;
            wtp[2 * plane + o] = p2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int check_geom_p(const sdt_conv_geom* g) {
    SDT_CHECK_ARG(g != nullptr, "null geometry");
    SDT_CHECK_ARG(g->B > 0 && g->Hi > 0 && g->Wi > 0 && g->Cin > 0 && g->Ho > 0 && g->Wo > 0 && g->Cout > 0, "non-positive dims");
    SDT_CHECK_ARG(g->ntaps > 0 && g->ntaps <= SDT_MAX_TAPS && g->Tw >= 1, "bad tap count");
    SDT_CHECK_ARG(g->Cin % BKP == 0, "pre-split kernels need Cin % 32 == 0");
    SDT_CHECK_ARG((int64_t)g->B * g->Hy * g->Wy * g->Cout < (1ll << 31), "output tensor too large for 32-bit offsets");
    SDT_CHECK_ARG((g->Ho - 1) * g->osy + g->ooy < g->Hy && (g->Wo - 1) * g->osx + g->oox < g->Wy, "output grid exceeds Y");
    for (int t = 0; t < g->ntaps; ++t) SDT_CHECK_ARG(g->wt[t] >= 0 && g->wt[t] < g->Tw, "weight tap out of range");
    const int64_t lim = (1ll << 31) - 65536;  // one buffer resource spans the three planes of an operand
    SDT_CHECK_ARG((int64_t)g->B * g->Hi * g->Wi * g->Cin * 6 < lim, "input planes exceed 2 GiB (split the batch)");
    SDT_CHECK_ARG((int64_t)g->Cout * g->Tw * g->Cin * 6 < lim, "weight planes exceed 2 GiB");
    return SDT_OK;
}

static int g_pre_tile = 0;  // 0 = automatic; 64064 / 128064 / 128128 force a tile (developer switch through sdt_set_pre_tile)
extern "C" int sdt_set_pre_tile(int tile) {
    SDT_CHECK_ARG(tile == 0 || tile == 64064 || tile == 128064 || tile == 128128, "unknown tile");
    g_pre_tile = tile;
    return SDT_OK;
}

template <int BM, int BN>
static void launch_pre(const __bf16* xp, size_t xplane, const __bf16* wp, size_t wplane, float* y, const sdt_conv_geom* const* gs,
                       int ncls, double* stats, int rpg, const norm_bwd_args_p& nb, hipStream_t s) {
    geom_pack_p gp;
    int tiles = 0;
    for (int c = 0; c < SDT_MAX_CLASSES; ++c) gp.g[c] = *gs[c < ncls ? c : 0];
    for (int c = 0; c < ncls; ++c) tiles = std::max(tiles, cdiv(gs[c]->B * gs[c]->Ho * gs[c]->Wo, BM) * cdiv(gs[c]->Cout, BN));
    dim3 grid(tiles, ncls, 1);
    if (stats != nullptr)
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, 1>), grid, dim3(256), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
    else if (nb.sums != nullptr)
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, 2>), grid, dim3(256), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
    else
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, 0>), grid, dim3(256), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
}

// Forward conv / input gradient from pre-split operands (see the header of this file and include/sdt_hip.h).
extern "C" int sdt_conv_taps_pre_f32(const void* x_planes, int64_t x_plane_elems, const void* w_planes, int64_t w_plane_elems,
                                     float* y, const sdt_conv_geom* geoms, int ncls, double* stats, int rows_per_group,
                                     const sdt_norm_bwd* nbw, void* stream) {
    SDT_CHECK_ARG(geoms && ncls >= 1 && ncls <= SDT_MAX_CLASSES, "1..4 geometries per launch");
    const sdt_conv_geom* gs[SDT_MAX_CLASSES];
    for (int c = 0; c < ncls; ++c) {
        gs[c] = geoms + c;
        int rc = check_geom_p(gs[c]);
        if (rc) return rc;
        SDT_CHECK_ARG(gs[c]->B == gs[0]->B && gs[c]->Hi == gs[0]->Hi && gs[c]->Wi == gs[0]->Wi && gs[c]->Cin == gs[0]->Cin &&
                          gs[c]->Hy == gs[0]->Hy && gs[c]->Wy == gs[0]->Wy && gs[c]->Cout == gs[0]->Cout && gs[c]->Tw == gs[0]->Tw,
                      "the classes of one launch must share the X, W and Y tensors");
    }
    SDT_CHECK_ARG(x_planes && w_planes && y, "null pointer");
    SDT_CHECK_ARG((((uintptr_t)x_planes | (uintptr_t)w_planes) % 16) == 0 && x_plane_elems % 8 == 0 && w_plane_elems % 8 == 0,
                  "planes must be 16-byte aligned");
    SDT_CHECK_ARG(x_plane_elems == (int64_t)gs[0]->B * gs[0]->Hi * gs[0]->Wi * gs[0]->Cin, "x plane size does not match the geometry");
    SDT_CHECK_ARG(w_plane_elems == (int64_t)gs[0]->Cout * gs[0]->Tw * gs[0]->Cin, "w plane size does not match the geometry");
    SDT_CHECK_ARG(!(stats && nbw), "one statistics epilogue per launch");
    if (stats) {
        const int64_t M = (int64_t)gs[0]->B * gs[0]->Ho * gs[0]->Wo;
        SDT_CHECK_ARG(ncls == 1 && rows_per_group >= 128 && M % rows_per_group == 0 && gs[0]->osy == 1 && gs[0]->osx == 1 &&
                          gs[0]->ooy == 0 && gs[0]->oox == 0 && gs[0]->Hy == gs[0]->Ho && gs[0]->Wy == gs[0]->Wo,
                      "geometry not supported by the fused-statistics epilogue");
    }
    norm_bwd_args_p nb = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    if (nbw != nullptr) {
        SDT_CHECK_ARG(nbw->y && nbw->mean && nbw->rstd && nbw->sums, "null pointer in sdt_norm_bwd");
        SDT_CHECK_ARG(nbw->groups == 1 || nbw->groups == gs[0]->B, "groups must be 1 (BatchNorm) or B (InstanceNorm)");
        for (int c = 0; c < ncls; ++c)
            SDT_CHECK_ARG(nbw->groups == 1 ? (int64_t)gs[c]->B * gs[c]->Ho * gs[c]->Wo >= 128 : gs[c]->Ho * gs[c]->Wo >= 128,
                          "a group must span at least one 128-row tile");
        nb = {nbw->y, nbw->mean, nbw->rstd, nbw->gamma, nbw->beta, nbw->sums, nbw->slope, nbw->groups};
    }
    hipStream_t s = (hipStream_t)stream;
    const __bf16* xp = (const __bf16*)x_planes;
    const __bf16* wp = (const __bf16*)w_planes;
    int tile = g_pre_tile;
    if (tile == 0) {  // enough workgroups for >= 2 waves of the chip at 128x128, else smaller tiles
        int64_t m = 0;
        for (int c = 0; c < ncls; ++c) m += (int64_t)gs[c]->B * gs[c]->Ho * gs[c]->Wo;
        const int64_t t128 = cdiv64(m, 128) * cdiv(gs[0]->Cout, 128);
        tile = (gs[0]->Cout % 128 == 0 && t128 >= 1024) ? 128128 : ((cdiv64(m, 128) * cdiv(gs[0]->Cout, 64) >= 1024) ? 128064 : 64064);
    }
    switch (tile) {
        case 128128: launch_pre<128, 128>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        case 128064: launch_pre<128, 64>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        default: launch_pre<64, 64>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_split_planes_f32(const float* x, void* planes, int64_t n, void* stream) {
    SDT_CHECK_ARG(x && planes && n > 0 && n % 8 == 0, "bad argument (n must be a multiple of 8)");
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n / 4, 256), 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)planes, (size_t)n);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_weight_planes_batched(const sdt_wp_desc* table, int n_layers, int total_tiles, void* stream) {
    SDT_CHECK_ARG(table && n_layers > 0 && total_tiles > 0, "bad argument");
    hipLaunchKernelGGL(weight_planes_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n_layers);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
