// Implicit-GEMM convolution kernels for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   conv_taps_kernel : Y[m][n] = sum_k im2col(X)[m][k] * W[n][k]      (forward conv and input-gradient)
//   conv_dw_kernel   : dW[n][k] += sum_m dY[m][n] * im2col(X)[m][k]   (weight gradient, split over m)
//
// Data layout: channels-last activations, (Cout, taps, Cin) weights -> both GEMM operands of the
// forward/dX product are K-contiguous, so global loads are 16-B vectors along K and LDS tiles are
// [row][K] with a 36-float pitch (conflict-free ds_read_b128).  A workgroup is 4 waves
// in a 2x2 grid; each wave owns a (BM/2)x(BN/2) sub-tile as TMxTN 32x32 MFMA accumulators.
//
// Replaces the ATen/cuDNN convolution calls made by building_blocks.py:15-22,31-38 (ConvNormRelu),
// generator.py:103 and discriminator.py:16 (k1 / k3 head convs) in forward and backward.
#include <stdlib.h>

#include "common.h"

#define BK 32
#define LDP 36  // LDS pitch of a [row][BK] tile, in floats
// Byte offset used for masked buffer loads: every tensor handed to the vector loaders is < 2^31 - 64 KiB bytes
// (checked on the host), so SDT_OOB + (in-row offset) is always past num_records and the load returns zeros.
#define SDT_OOB 0x80000000u

// Up to 4 geometries per launch: the output parity classes of a strided layer's input gradient run as ONE grid
// (blockIdx.y = class) instead of 4 (2 in the 1-D stage) short launches -- one prologue / tail / launch boundary instead of four.
#define SDT_MAX_CLASSES 4
struct geom_pack {
    sdt_conv_geom g[SDT_MAX_CLASSES];
    // > 0: rotate the m-tile index by mrot * (block of 32 / n-tiles m-tiles) inside each such block (launch_taps: layers whose tap
    // culling makes the work of a tile depend strongly on its image row, i.e. the p = 0 (6,3) layer's input gradient)
    int mrot;
};
// Epilogue of an input-gradient launch that also accumulates the statistics of the normalisation BACKWARD that consumes it
// (what colstats_kernel<true> would compute in a separate pass over dz and y):
//   sums[(grp*C + n)*2 + {0,1}] += sum gg, sum gg*yhat,   gg = dz * act'(gamma*yhat + beta),  yhat = (y - mean)*rstd
// y = raw forward output of the layer below (same shape as the dz tensor this launch writes), grp = batch item (groups == B,
// InstanceNorm) or 0 (groups == 1, BatchNorm).  sums == nullptr: off.
struct norm_bwd_args {
    const float* y;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    double* sums;
    float slope;
    int groups;
};

#ifdef SDT_TUNING
// PRIO 30 (tools/debug/taps_timeline.py): lane 0 of every workgroup stamps the 100 MHz real-time counter at five points of its
// life and its hardware id into sdt_dbg_tl[8 * linear workgroup id ...] -- where a launch's time goes (prologue / K loop / epilogue,
// workgroups per CU, generations).
__device__ unsigned long long* sdt_dbg_tl = nullptr;
extern "C" int sdt_debug_set_timeline(void* p) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(sdt_dbg_tl), &p, sizeof(p));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define SDT_TL(slot)                                                                                                   \
    do {                                                                                                               \
        if constexpr (PRIO >= 30 && PRIO < 40) {                                                                       \
            if (threadIdx.x == 0 && sdt_dbg_tl != nullptr)                                                             \
                sdt_dbg_tl[8 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + (slot)] = wall_clock64(); \
        }                                                                                                              \
    } while (0)
#else
#define SDT_TL(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// stats != nullptr: the epilogue also accumulates sum(y) and sum(y^2) per (group, output channel) into ``stats`` (fp64
// atomics; the buffer must be zero on entry), group = output row index / rows_per_group -- the statistics pass of the
// InstanceNorm2d / BatchNorm that follows (sdt_colnorm_fwd_f32 with stats_ready = 1) then never reads y.
// rows_per_group >= BM, so a tile touches at most two groups.  (A run-time switch, not a template argument: one kernel
// symbol for every forward / input-gradient launch; the branch is uniform and outside the K loop.)
// EPI selects the epilogue at compile time (each statistics epilogue costs registers: a run-time switch took the plain kernel
// from 7 to 5 waves per SIMD): 0 = store only, 1 = + forward statistics (stats), 2 = + normalisation-backward statistics (nb).
// The 64x64 instantiations are held to 7 waves per SIMD (72 registers): what hides a workgroup's prologue and epilogue is its six neighbours.
template <int BM, int BN, bool VEC4, int PRIO = 0, int EPI = 0>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64) ? 7 : 1) void conv_taps_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                        const geom_pack gp, const int splitk,
                                                        float* __restrict__ partial, const size_t ysize,
                                                        double* __restrict__ stats, const int rows_per_group,
                                                        const norm_bwd_args nb) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;
    // the class this workgroup belongs to (uniform).  Its scalar fields are pinned into SGPRs: read through the dynamically
    // indexed kernel-argument pack they otherwise end up in VGPRs (+8-10 VGPRs = one wave per SIMD less)
    const sdt_conv_geom& gt = gp.g[blockIdx.y];  // tap tables
    struct {
        int B, Hi, Wi, Cin, Ho, Wo, Hy, Wy, Cout, sy, sx, osy, osx, ooy, oox, ntaps, Tw;
    } g;
#define SDT_SGPR(f) g.f = __builtin_amdgcn_readfirstlane(gt.f)
    SDT_SGPR(B); SDT_SGPR(Hi); SDT_SGPR(Wi); SDT_SGPR(Cin); SDT_SGPR(Ho); SDT_SGPR(Wo); SDT_SGPR(Hy); SDT_SGPR(Wy); SDT_SGPR(Cout);
    SDT_SGPR(sy); SDT_SGPR(sx); SDT_SGPR(osy); SDT_SGPR(osx); SDT_SGPR(ooy); SDT_SGPR(oox); SDT_SGPR(ntaps); SDT_SGPR(Tw);
#undef SDT_SGPR
    constexpr bool DBUF = PRIO == 20;  // experiment: two LDS buffers, ONE barrier per K step (tile k+1 is written while tile k computes)
    __shared__ __attribute__((aligned(16))) float sA[(DBUF ? 2 : 1) * BM * LDP];
    __shared__ __attribute__((aligned(16))) float sB[(DBUF ? 2 : 1) * BN * LDP];
    __shared__ int sOut[BM];
    __shared__ int sTap[3 * SDT_MAX_TAPS];
    __shared__ int sLive[SDT_MAX_TAPS + 1];  // flags, then the ordered list of the taps that reach at least one in-range input for this tile; [MAX] = count
    __shared__ int sFlag[SDT_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = g.B * g.Ho * g.Wo;
    const int nmb = (M + BM - 1) / BM;
    if constexpr (PRIO == 2) {  // static per-workgroup priority: de-synchronises the co-resident workgroups' phases
        switch ((blockIdx.x >> 3) & 3) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(3); break;
            default: break;
        }
    }
    // 1-D grid over (m-tile, n-tile) pairs.  Each XCD (private 4 MiB L2) gets a contiguous range of pairs with the
    // n-tile index fastest: the workgroups that share an A tile (same m-tile, different n-tiles) and the m-tiles that
    // share input rows across vertical taps run on the same XCD close in time, so A is fetched into that L2 once
    // (measured before this ordering: L2-miss traffic 9-16x the input bytes on the Cout=256 layers).
    const int nnb = (g.Cout + BN - 1) / BN;
    if ((int)blockIdx.x >= nmb * nnb) return;  // the grid is sized for the largest class of the launch
    // experiments 31 / 32: waves outside the K loop (prologue, epilogue) issue at raised priority -- the youngest wave of a SIMD
    // otherwise loses every VALU / LDS / memory issue slot to the older waves' MFMAs (timeline: prologue 6 us alone, 30 us loaded)
    if constexpr (PRIO == 31 || PRIO == 32) __builtin_amdgcn_s_setprio(3);
    SDT_TL(0);
#ifdef SDT_TUNING
    if constexpr (PRIO >= 30 && PRIO < 40) {
        if (threadIdx.x == 0 && sdt_dbg_tl != nullptr)
            sdt_dbg_tl[8 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 5] =
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif
    const int lin = xcd_remap(blockIdx.x, nmb * nnb);
    int mt = lin / nnb;
    if constexpr (VEC4) {
        // Load balance under tap culling.  A CU's co-resident workgroups are 32 positions apart in `lin`, i.e. 32 / nnb m-tiles; when
        // that stride is close to the number of tiles per image (8 vs 8.28 on the (6,3) p=0 layer's input gradient) every tile of a
        // CU sits at the same image row: one CU gets 4 tiles with 15-18 live taps, another 4 tiles with 3-6, and the launch waits for
        // the first (293 us at 66 TFLOP/s).  Rotating the tile index inside consecutive blocks of 32 / nnb tiles by mrot * block
        // spreads a CU's tiles evenly over the image rows.
        const int mrot = __builtin_amdgcn_readfirstlane(gp.mrot);
        if (mrot > 0) {
            const int S = 32 / nnb, blk = mt / S;
            if ((blk + 1) * S <= nmb) mt = blk * S + (mt - blk * S + mrot * blk) % S;
        }
    }
    const int m0 = mt * BM;
    const int n0 = (lin % nnb) * BN;

    if (tid < g.ntaps) {
        sTap[tid] = gt.dy[tid];
        sTap[SDT_MAX_TAPS + tid] = gt.dx[tid];
        sTap[2 * SDT_MAX_TAPS + tid] = gt.wt[tid];
    }
    if (tid < SDT_MAX_TAPS) sFlag[tid] = 0;
    if (tid < BM) {
        int m = m0 + tid, off = -1;
        if (m < M) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            off = ((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout;
        }
        sOut[tid] = off;
    }
    // loader mapping: 8 threads x 16 B cover one 32-float K row; 32 rows per pass
    const int kv = tid & 7, r0 = tid >> 3;
    int rbH[RA], riy[RA], rix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m < M) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            rbH[i] = b * g.Hi;
            riy[i] = oy * g.sy;
            rix[i] = ox * g.sx;
        } else {
            rbH[i] = 0;
            riy[i] = -(1 << 20);  // always out of range
            rix[i] = 0;
        }
    }
    __syncthreads();
    int ntl = g.ntaps;
    // tap order (uniform per launch; PRIO 12 / 14 force table / row-residue order for A/B runs), see the note below
    const bool rotate = PRIO == 14 || (PRIO != 12 && (unsigned)(g.Cout * g.Tw) * (unsigned)g.Cin <= (3u << 17));  // weights <= 1.5 MiB
    if (VEC4 && (g.Hi == 1 || (g.ntaps <= 4 && PRIO != 16)) && PRIO != 11) {
        // 1-D stage: a tap can only be dead for a whole tile when T is tiny, and these launches are latency-bound --
        // skip the culling passes (two barriers and two serial loops of the prologue).  Likewise the parity classes of a
        // strided layer's input gradient (2x2 taps, K loop of only 8-32 steps: the prologue is a visible share of a
        // workgroup's life, culling could only trim border tiles, and their re-reads already hit the L2)
        if (tid < g.ntaps) sLive[tid] = tid;
        if (tid == 0) sLive[SDT_MAX_TAPS] = g.ntaps;
        __syncthreads();
    } else if constexpr (VEC4) {
        // tap culling: a tap whose input coordinates are out of range for EVERY row of this tile contributes only
        // zeros -- skip its K steps (halves the work of the p=0 (6,3) layer's input gradient; trims padded borders)
        if (kv == 0) {
            for (int t = 0; t < g.ntaps; ++t) {
                const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t];
                bool any = false;
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    any |= (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                if (any) sFlag[t] = 1;  // benign race: every writer stores 1
            }
        }
        __syncthreads();
        // Order of the live taps, chosen per launch so that re-reads hit the XCD's 4 MiB L2 (fabric-side traffic measured
        // with tools/fetch_calibration.py --encoder; the kernel is MFMA-bound either way, its speed does not change):
        //  * small weight tensor, big activations (L1..L4): by the residue of the input row a tap reads,
        //    (oy0*sy + dy) mod (number of tap rows), oy0 = output row of the tile's first pixel -- NOT by dy.  An input row
        //    is needed by every output row within the kernel's vertical reach; their tiles are co-resident on one XCD, and
        //    in dy-major order they would touch that row a third (k3) or a quarter (k4) of a workgroup's lifetime apart, by
        //    which time the L2 has streamed several times its size and the row is fetched again (1.9x / 2.6x the input
        //    bytes on the 3x3 / 4x4-stride-2 layers; 1.05x / 1.2x with this order: all co-resident tiles read input rows of
        //    one residue class in the same phase of their K loops);
        //  * weights that alone fill an L2 (L5..L7, 2.4-4.7 MB): table order, so that all tiles read the SAME weight tap
        //    at the same time (rotating them re-streams the weights: 2.7x more traffic on L7).  A channel-chunk-major
        //    K loop would fix these layers' re-reads too (measured: L5 forward 368 -> 145 MB, L7 input gradient 230 -> 85 MB)
        //    but rebuilds the tap offsets every K step and costs 7-17 % of the launch time -- rejected.
        if (tid < 64) {
            int dymin = sTap[0], dymax = sTap[0];
            for (int u = 1; u < g.ntaps; ++u) {
                dymin = min(dymin, sTap[u]);
                dymax = max(dymax, sTap[u]);
            }
            const int nd = dymax - dymin + 1;
            const int b0 = rotate ? (((m0 / g.Wo) % g.Ho) * g.sy) % nd : 0;  // uniform
            const bool mine = tid < g.ntaps && sFlag[tid] != 0;
            int kt = 0;
            if (mine && rotate) {
                kt = b0 + sTap[tid] - dymin;
                kt -= kt >= nd ? nd : 0;
            }
            int rank = 0, total = 0;
            for (int u = 0; u < g.ntaps; ++u) {
                if (sFlag[u] == 0) continue;
                int ku = 0;
                if (rotate) {
                    ku = b0 + sTap[u] - dymin;
                    ku -= ku >= nd ? nd : 0;
                }
                rank += (ku < kt || (ku == kt && u < tid)) ? 1 : 0;
                ++total;
            }
            if (mine) sLive[rank] = tid;
            if (tid == 0) sLive[SDT_MAX_TAPS] = total;
        }
        __syncthreads();
        ntl = sLive[SDT_MAX_TAPS];
    }

    const int nkc = (g.Cin + BK - 1) / BK;
    const int Ktot = g.ntaps * g.Cin;
    const int nsteps_all = VEC4 ? ntl * nkc : (Ktot + BK - 1) / BK;
    // split-K: slice z of the K loop (deterministic: partial tiles go to their own slab, summed by splitk_reduce)
    const int step0 = (int)(((long)blockIdx.z * nsteps_all) / splitk);
    const int nsteps = (int)(((long)(blockIdx.z + 1) * nsteps_all) / splitk);

    // VEC4 loader: raw buffer loads with 32-bit byte offsets; masked rows / taps use an out-of-range offset and the
    // hardware returns zeros (no branches, no 64-bit address arithmetic in the K loop).  Row offsets are rebuilt
    // only when the tap changes (every Cin/32 steps).
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((unsigned)g.Cout * g.Tw * g.Cin * 4u), 0x00020000);
    unsigned aoff[RA], boff[RB], abase[RA], bbase[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) abase[i] = (unsigned)(((rbH[i] + riy[i]) * g.Wi + rix[i]) * g.Cin) * 4u;  // tap (0,0); masked rows never use it
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 32 * i;
        bbase[i] = n < g.Cout ? (unsigned)(n * g.Tw * g.Cin) * 4u : SDT_OOB;
    }
    int cur_tl = -1;
    f32x4 ra[RA], rb[RB];
    // (tap, channel-chunk) of the next step to load: steps are requested consecutively, so the pair is advanced
    // incrementally instead of dividing step by nkc in every K step
    int nxt_tl = step0 / nkc, nxt_kc = step0 - nxt_tl * nkc;
    auto load = [&](int step) {
        if constexpr (VEC4) {
            int tl, kc;
            if constexpr (PRIO == 11) {  // A/B: the division form
                tl = step / nkc;
                kc = step - tl * nkc;
            } else {
                tl = nxt_tl;
                kc = nxt_kc;
                if (++nxt_kc == nkc) nxt_kc = 0, ++nxt_tl;
            }
            const unsigned cb = (unsigned)(kc * BK + kv * 4) * 4u;
            const int cs = kc * BK * 4;  // uniform part of cb: goes into the load's scalar offset operand
            if (tl != cur_tl) {
                cur_tl = tl;
                const int t = sLive[tl];
                const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
                if constexpr (PRIO == 11) {  // A/B: offsets rebuilt from scratch
#pragma unroll
                    for (int i = 0; i < RA; ++i) {
                        const int iy = riy[i] + dy, ix = rix[i] + dx;
                        const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                        aoff[i] = ok ? (unsigned)(((rbH[i] + iy) * g.Wi + ix) * g.Cin) * 4u : SDT_OOB;
                    }
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        const int n = n0 + r0 + 32 * i;
                        boff[i] = n < g.Cout ? (unsigned)((n * g.Tw + wt) * g.Cin) * 4u : SDT_OOB;
                    }
                } else {  // row bases are fixed for the tile: a tap only adds a uniform shift (and decides the mask)
                    const unsigned ashift = (unsigned)((dy * g.Wi + dx) * g.Cin) * 4u + (unsigned)kv * 16u;
                    const unsigned bshift = (unsigned)(wt * g.Cin) * 4u + (unsigned)kv * 16u;
#pragma unroll
                    for (int i = 0; i < RA; ++i) {
                        const bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                        aoff[i] = ok ? abase[i] + ashift : SDT_OOB;
                    }
#pragma unroll
                    for (int i = 0; i < RB; ++i) boff[i] = bbase[i] == SDT_OOB ? SDT_OOB : bbase[i] + bshift;
                }
            }
            if constexpr (PRIO == 11) {
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(aoff[i] + cb), 0, 0));
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(boff[i] + cb), 0, 0));
            } else {  // per-lane part of the column offset lives in aoff/boff, the K-chunk part in the scalar operand
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)aoff[i], cs, 0));
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)boff[i], cs, 0));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = step * BK + kv * 4 + e;
                const bool kok = kk < Ktot;
                const int t = kok ? kk / g.Cin : 0;
                const int c = kk - t * g.Cin;
                const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int iy = riy[i] + dy, ix = rix[i] + dx;
                    const bool ok = kok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                    ra[i][e] = ok ? X[((size_t)(rbH[i] + iy) * g.Wi + ix) * g.Cin + c] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int n = n0 + r0 + 32 * i;
                    rb[i][e] = (kok && n < g.Cout) ? W[((size_t)n * g.Tw + wt) * g.Cin + c] : 0.f;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr bool TWO_ACC = (PRIO == 5 || PRIO == 6);  // experiment: even/odd k sub-steps into separate accumulators
    f32x16 acc2[TWO_ACC ? TM : 1][TWO_ACC ? TN : 1];
    if constexpr (TWO_ACC) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }

    const float* pa = sA + (wm * (BM / 2) + (lane & 31)) * LDP + (lane >> 5) * 4;
    const float* pb = sB + (wn * (BN / 2) + (lane & 31)) * LDP + (lane >> 5) * 4;

    if constexpr (DBUF) {
        if (step0 < nsteps) {
            load(step0);
#pragma unroll
            for (int i = 0; i < RA; ++i) *(f32x4*)&sA[(r0 + 32 * i) * LDP + kv * 4] = ra[i];
#pragma unroll
            for (int i = 0; i < RB; ++i) *(f32x4*)&sB[(r0 + 32 * i) * LDP + kv * 4] = rb[i];
            if (step0 + 1 < nsteps) load(step0 + 1);
        }
        __syncthreads();
        for (int step = step0; step < nsteps; ++step) {
            const int cur = (step - step0) & 1;
            if (step + 1 < nsteps) {  // regs hold tile step+1: into the other buffer (nobody reads it: its last readers passed the barrier below)
#pragma unroll
                for (int i = 0; i < RA; ++i) *(f32x4*)&sA[(cur ^ 1) * BM * LDP + (r0 + 32 * i) * LDP + kv * 4] = ra[i];
#pragma unroll
                for (int i = 0; i < RB; ++i) *(f32x4*)&sB[(cur ^ 1) * BN * LDP + (r0 + 32 * i) * LDP + kv * 4] = rb[i];
                if (step + 2 < nsteps) load(step + 2);
            }
            const float* qa = pa + cur * BM * LDP;
            const float* qb = pb + cur * BN * LDP;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = *(const f32x4*)(qa + tm * 32 * LDP + j * 8);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = *(const f32x4*)(qb + tn * 32 * LDP + j * 8);
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
    } else {
    SDT_TL(1);
    if (step0 < nsteps) load(step0);
    for (int step = step0; step < nsteps; ++step) {
        if constexpr (PRIO == 32 || PRIO == 33) __builtin_amdgcn_s_setprio(2);  // feeding part of a K step above the MFMA bursts
        if (!(PRIO == 7 || PRIO == 8 || PRIO == 9) || step == step0) {
#pragma unroll
            for (int i = 0; i < RA; ++i) *(f32x4*)&sA[(r0 + 32 * i) * LDP + kv * 4] = ra[i];
#pragma unroll
            for (int i = 0; i < RB; ++i) *(f32x4*)&sB[(r0 + 32 * i) * LDP + kv * 4] = rb[i];
        }
        if (PRIO != 4 && (PRIO != 9 || step == step0)) __syncthreads();
        if (step == step0) SDT_TL(2);  // the first tile is in LDS
        if constexpr (PRIO != 3 && PRIO != 6 && PRIO != 7 && PRIO != 8 && PRIO != 9) {
            if (step + 1 < nsteps) load(step + 1);
        }
        // K order inside the 32-wide tile is permuted identically for A and B: MFMA k-slot h=lane>>5 of
        // sub-step (j,e) consumes k = 8j + 4h + e, so each lane feeds 4 MFMAs from one ds_read_b128.
        if constexpr (PRIO == 31 || PRIO == 32 || PRIO == 33) __builtin_amdgcn_s_setprio(0);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = ((PRIO == 8 || PRIO == 9) && step != step0) ? ra[tm] : *(const f32x4*)(pa + tm * 32 * LDP + j * 8);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = ((PRIO == 8 || PRIO == 9) && step != step0) ? rb[tn] : *(const f32x4*)(pb + tn * 32 * LDP + j * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        if (TWO_ACC && (e & 1))
                            acc2[TWO_ACC ? tm : 0][TWO_ACC ? tn : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc2[TWO_ACC ? tm : 0][TWO_ACC ? tn : 0], 0, 0, 0);
                        else
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
                    }
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (PRIO != 4 && (PRIO != 9 || step == step0)) __syncthreads();
    }
    }

    if constexpr (TWO_ACC) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
    }
    if constexpr (PRIO == 31 || PRIO == 32) __builtin_amdgcn_s_setprio(3);
    SDT_TL(3);  // K loop done
    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (splitk > 1) {
        Y = partial + (size_t)blockIdx.z * ysize;
        bias = nullptr;
    }
    // EPI 2 reads the forward output y at every position of the tile: all of those loads are issued first, branch-free (positions outside
    // the tensor read element 0 and are masked below), so the tile pays ONE memory latency instead of one per group of four rows behind
    // the stores -- as in convsk.hip, where the same change was worth 19 % on the input-gradient launches
    float yv[EPI == 2 ? TM * TN * 16 : 1];
    if constexpr (EPI == 2) {
        // (buffer loads: one offset register per load instead of a 64-bit address -- the kernel has to stay at 72 registers, 7 workgroups per CU)
        const __amdgpu_buffer_rsrc_t rsNY = __builtin_amdgcn_make_buffer_rsrc((void*)nb.y, 0, 0x7ffffffc, 0x00020000);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = n0 + wn * (BN / 2) + tn * 32 + (lane & 31);
                const bool nok = n < g.Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = sOut[wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                    yv[(tm * TN + tn) * 16 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsNY, (off >= 0 && nok) ? (off + n) * 4 : (int)0x80000000, 0, 0));
                }
            }
    }
    // stores: branch-free buffer stores when the tensor's byte offsets fit 31 bits (rows / columns outside the tensor get an out-of-range
    // offset and the hardware drops the store); the guarded 64-bit form otherwise
    const bool ysmall = (size_t)g.B * g.Hy * g.Wy * g.Cout * 4 < (size_t)0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t rsYs = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, 0x7ffffffc, 0x00020000);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * (BN / 2) + tn * 32 + (lane & 31);
            const bool nok = n < g.Cout;
            const float bv = (bias != nullptr && nok) ? bias[n] : 0.f;
            if (ysmall) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = sOut[wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tm][tn][r] + bv), rsYs, (off >= 0 && nok) ? (off + n) * 4 : (int)0x80000000, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int off = sOut[row];
                    if (off >= 0 && nok) Y[(size_t)off + n] = acc[tm][tn][r] + bv;
                }
            }
            if constexpr (EPI == 1) {
                const int g0 = m0 / rows_per_group;
                const int mb = (g0 + 1) * rows_per_group;  // first row of the next group
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (sOut[row] >= 0 && nok) {
                        const float v = acc[tm][tn][r] + bv;
                        if (m0 + row < mb) {
                            s0 += v;
                            q0 = fmaf(v, v, q0);
                        } else {
                            s1 += v;
                            q1 = fmaf(v, v, q1);
                        }
                    }
                }
                s0 += __shfl_xor(s0, 32, 64);  // the two lane halves hold different rows of the same column
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
                const int rpg = nb.groups == 1 ? M : g.Ho * g.Wo;  // >= BM (host-checked): a tile touches at most two groups
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
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int off = sOut[row];
                    if (off >= 0 && nok) {
                        const bool second = m0 + row >= mb;
                        const float yh = (yv[(tm * TN + tn) * 16 + r] - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
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
    SDT_TL(4);  // every store / atomic of the epilogue issued
}


// ---------------------------------------------------------------------------------------------
// The same contract for the launches of the 1-D stage (U-Net / decoder / pose encoder / discriminator: Hi = 1, M = B*T <= 2048 rows,
// K split over blockIdx.z so that a workgroup has at most NS K steps).  These launches are latency-bound: the per-workgroup timeline of
// conv_taps_kernel on them (tools/debug/taps_timeline.py --only c1d_k3_T64) reads 2.3 us of dispatch, 2.5 us prologue (tap tables through
// LDS, two barriers), 0.9 us until the first tile is in LDS, 7.8 us for 6 K steps whose MFMA floor at two workgroups per CU is 5.1 us (each
// step waits for the loads issued one step earlier), 2.2 us epilogue.  Here a workgroup issues the loads of ALL its K steps up front
// (NS * 4 16-byte loads per thread: the whole latency is paid once), derives its row offsets with one division per row (1-D rows are
// contiguous), and stores through branch-free buffer stores.  The products are accumulated in the order of conv_taps_kernel (same LDS
// layout, same k permutation, same split points): results are BIT-identical to that kernel's.
template <int NS>
__global__ __launch_bounds__(256, 2) void conv1d_small_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ Y, const geom_pack gp,
                                                               const int splitk, float* __restrict__ partial, const size_t ysize) {
    constexpr int BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDP];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDP];
    __shared__ __attribute__((aligned(16))) int sOut[BM];
    const sdt_conv_geom& gt = gp.g[blockIdx.y];
    const int B = __builtin_amdgcn_readfirstlane(gt.B), Wi = __builtin_amdgcn_readfirstlane(gt.Wi), Cin = __builtin_amdgcn_readfirstlane(gt.Cin);
    const int Wo = __builtin_amdgcn_readfirstlane(gt.Wo), Wy = __builtin_amdgcn_readfirstlane(gt.Wy), Cout = __builtin_amdgcn_readfirstlane(gt.Cout);
    const int sx = __builtin_amdgcn_readfirstlane(gt.sx), osx = __builtin_amdgcn_readfirstlane(gt.osx), oox = __builtin_amdgcn_readfirstlane(gt.oox);
    const int ntaps = __builtin_amdgcn_readfirstlane(gt.ntaps), Tw = __builtin_amdgcn_readfirstlane(gt.Tw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = B * Wo;
    const int nmb = (M + BM - 1) / BM, nnb = Cout / BN;
    if ((int)blockIdx.x >= nmb * nnb) return;  // the grid is sized for the largest class of the launch
    const int lin = xcd_remap(blockIdx.x, nmb * nnb);
    const int m0 = (lin / nnb) * BM, n0 = (lin % nnb) * BN;
    const int nkc = Cin / BK;
    const int nsteps_all = ntaps * nkc;
    const int step0 = (int)(((long)blockIdx.z * nsteps_all) / splitk);
    const int nsteps = (int)(((long)(blockIdx.z + 1) * nsteps_all) / splitk) - step0;  // <= NS (host-checked)

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((unsigned)B * Wi * Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((unsigned)Cout * Tw * Cin * 4u), 0x00020000);
    const int kv = tid & 7, r0 = tid >> 3;
    // this thread's two A rows: element index of input position ix0 = ox * sx (tap dx = 0) and ix0 itself; rows past M never load
    int abase[2], aix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int b = m / Wo, ox = m - b * Wo;
        aix[i] = m < M ? ox * sx : -(1 << 24);
        abase[i] = (b * Wi + ox * sx) * Cin;
    }
    const int bbase[2] = {(n0 + r0) * Tw * Cin, (n0 + r0 + 32) * Tw * Cin};
    f32x4 ra[NS][2], rb[NS][2];
    {
        int t = step0 / nkc, kc = step0 - t * nkc;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // branch-free masks (a uniform branch around a load makes hipcc drain vmcnt(0) at the merge): bit 31 of an offset = "no load"
            const unsigned offm = (unsigned)(nsteps - 1 - s) & SDT_OOB;  // s >= nsteps
            const int tt = min(t, ntaps - 1);
            const int dx = gt.dx[tt], wt = gt.wt[tt];  // uniform: scalar loads from the kernel arguments
            const int cs = (kc * BK + kv * 4) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ix = aix[i] + dx;
                const unsigned rowm = ((unsigned)ix | (unsigned)(Wi - 1 - ix)) & SDT_OOB;  // ix < 0 or ix >= Wi (rows past M: ix << 0)
                const unsigned ao = ((unsigned)((abase[i] + dx * Cin) * 4 + cs) & 0x7fffffffu) | rowm | offm;
                ra[s][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)ao, 0, 0));
                const unsigned bo = (unsigned)((bbase[i] + wt * Cin) * 4 + cs) | offm;
                rb[s][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)bo, 0, 0));
            }
            if (++kc == nkc) kc = 0, ++t;
        }
    }
    if (tid < BM) {  // byte offset of the tile's output rows (SDT_OOB: past M)
        const int m = m0 + tid;
        const int b = m / Wo, ox = m - b * Wo;
        sOut[tid] = m < M ? (int)((unsigned)((b * Wy + ox * osx + oox) * Cout) * 4u) : (int)SDT_OOB;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wofs = r0 * LDP + kv * 4;
    const int fa = (wm * 32 + (lane & 31)) * LDP + (lane >> 5) * 4, fb = (wn * 32 + (lane & 31)) * LDP + (lane >> 5) * 4;
    auto stage = [&](int s, int buf) {
        *(f32x4*)&sA[buf][wofs] = ra[s][0];
        *(f32x4*)&sA[buf][wofs + 32 * LDP] = ra[s][1];
        *(f32x4*)&sB[buf][wofs] = rb[s][0];
        *(f32x4*)&sB[buf][wofs + 32 * LDP] = rb[s][1];
    };
    stage(0, 0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s < nsteps) {  // uniform
            if (s + 1 < NS && s + 1 < nsteps) stage(s + 1, (s + 1) & 1);  // the other buffer: its last readers passed the barrier below
            const float* qa = &sA[s & 1][fa];
            const float* qb = &sB[s & 1][fb];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = *(const f32x4*)(qa + j * 8), b = *(const f32x4*)(qb + j * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); rows past M carry the out-of-range offset
    float* out = splitk > 1 ? partial + (size_t)blockIdx.z * ysize : Y;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((unsigned)ysize * 4u), 0x00020000);
    const int n = n0 + wn * 32 + (lane & 31);
    const float bv = (bias != nullptr && splitk == 1) ? bias[n] : 0.f;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int4 o4 = *(const int4*)&sOut[wm * 32 + 8 * qq + 4 * (lane >> 5)];
        const unsigned nb4 = (unsigned)n * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[4 * qq + 0] + bv), rsY, (int)((unsigned)o4.x + nb4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[4 * qq + 1] + bv), rsY, (int)((unsigned)o4.y + nb4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[4 * qq + 2] + bv), rsY, (int)((unsigned)o4.z + nb4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[4 * qq + 3] + bv), rsY, (int)((unsigned)o4.w + nb4), 0, 0);
    }
}

#ifdef SDT_TUNING  // rejected experiment, kept only in the tuning build (tools/conv_bench.py)
// ---------------------------------------------------------------------------------------------
// conv_taps with asynchronous global->LDS staging (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass, two
// LDS buffers and ONE barrier per K step.  An LDS-DMA writes wave-uniform base + lane*16 B, so a tile is stored as
// unpadded 128-byte rows and the bank-conflict avoidance moves to the SOURCE side: lane (row, position c') fetches the
// 16-byte chunk c' ^ ((row >> 1) & 7) of its row, and fragment reads address chunk q at position q ^ ((row >> 1) & 7)
// (16 consecutive rows then hit 16 distinct 16-byte slots of the 256-byte bank window).  64x64 tile, Cin % 32 == 0.
__global__ __launch_bounds__(256) void conv_taps_dma_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ Y,
                                                            const sdt_conv_geom g, const int splitk,
                                                            float* __restrict__ partial, const size_t ysize) {
    constexpr int BM = 64, BN = 64, RA = 2, RB = 2;
    __shared__ __attribute__((aligned(16))) float sA[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * BK];
    __shared__ int sOut[BM];
    __shared__ int sTap[3 * SDT_MAX_TAPS];
    __shared__ int sLive[SDT_MAX_TAPS + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = g.B * g.Ho * g.Wo;
    const int nmb = (M + BM - 1) / BM;
    const int nnb = (g.Cout + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, nmb * nnb);
    const int m0 = (lin / nnb) * BM;
    const int n0 = (lin % nnb) * BN;

    if (tid < g.ntaps) {
        sTap[tid] = g.dy[tid];
        sTap[SDT_MAX_TAPS + tid] = g.dx[tid];
        sTap[2 * SDT_MAX_TAPS + tid] = g.wt[tid];
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
    const int kv = tid & 7, r0 = tid >> 3;
    int rbH[RA], riy[RA], rix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 32 * i;
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
    if (kv == 0) {
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
    const int nkc = g.Cin / BK;
    const int nsteps_all = ntl * nkc;
    const int step0 = (int)(((long)blockIdx.z * nsteps_all) / splitk);
    const int nsteps = (int)(((long)(blockIdx.z + 1) * nsteps_all) / splitk);

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((unsigned)g.Cout * g.Tw * g.Cin * 4u), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned swz = (unsigned)((kv ^ ((r0 >> 1) & 7)) * 16);  // byte offset of the chunk this lane fetches (r0 + 32 i keeps it)
    unsigned aoff[RA], boff[RB];
    int cur_tl = -1;
    auto issue = [&](int step, int buf) {
        const int tl = step / nkc;
        const unsigned cb = (unsigned)((step - tl * nkc) * BK) * 4u + swz;
        if (tl != cur_tl) {
            cur_tl = tl;
            const int t = sLive[tl];
            const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int iy = riy[i] + dy, ix = rix[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                aoff[i] = ok ? (unsigned)(((rbH[i] + iy) * g.Wi + ix) * g.Cin) * 4u : SDT_OOB;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int n = n0 + r0 + 32 * i;
                boff[i] = n < g.Cout ? (unsigned)((n * g.Tw + wt) * g.Cin) * 4u : SDT_OOB;
            }
        }
        // wave w fills rows [8w + 32i, 8w + 32i + 8) of each tile: 1 KiB, lane*16 B apart
#pragma unroll
        for (int i = 0; i < RA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(&sA[buf][(wave * 8 + 32 * i) * BK]), 16, (int)(aoff[i] + cb), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(&sB[buf][(wave * 8 + 32 * i) * BK]), 16, (int)(boff[i] + cb), 0, 0, 0);
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // fragment chunk q = 2j + h of row (lane & 31) sits at position q ^ ((row >> 1) & 7)
    const int fsw = (lane >> 1) & 7, fh = lane >> 5;
    int foff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) foff[j] = ((2 * j + fh) ^ fsw) * 4;
    const int rowA = (wm * 32 + (lane & 31)) * BK, rowB = (wn * 32 + (lane & 31)) * BK;

    if (step0 < nsteps) issue(step0, 0);
    for (int step = step0; step < nsteps; ++step) {
        const int buf = (step - step0) & 1;
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA for `step` has landed
        __syncthreads();                      // ... everyone's has, and everyone is done reading the other buffer
        if (step + 1 < nsteps) issue(step + 1, buf ^ 1);
        const float* pa = &sA[buf][rowA];
        const float* pb = &sB[buf][rowB];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = *(const f32x4*)(pa + foff[j]);
            const f32x4 b = *(const f32x4*)(pb + foff[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
    }

    if (splitk > 1) {
        Y = partial + (size_t)blockIdx.z * ysize;
        bias = nullptr;
    }
    const int n = n0 + wn * 32 + (lane & 31);
    const bool nok = n < g.Cout;
    const float bv = (bias != nullptr && nok) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int off = sOut[row];
        if (off >= 0 && nok) Y[(size_t)off + n] = acc[r] + bv;
    }
}

#endif  // SDT_TUNING

// ---------------------------------------------------------------------------------------------
// conv_taps on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate), fp32 in HBM, fp32 accumulate.
//   NS = 1: operands rounded to bf16 (RNE)                      -> "bf16" math (BASELINE config 4)
//   NS = 3: x = x1 + x2 (+ dropped x3), terms a1b1 + a1b2 + a2b1 -> ~16 significant bits ("bf16x3")
//   NS = 6: x = x1 + x2 + x3 EXACTLY (three 8-bit pieces of the 24-bit significand, split by truncation), terms
//           a1b1 | a1b2 a2b1 a2b2 a1b3 a3b1 (dropped: < 2^-23 |ab|) -> fp32-equivalent products ("bf16x6")
// The split happens when the register-staged global tile is written to LDS; LDS holds one [row][32] bf16 image per
// piece (pitch 40 elements = 80 B: conflict-free ds_read_b128).  The large term and the correction terms go to
// separate accumulators and are added once at the end.  Same tap table, loaders, culling and epilogue as
// conv_taps_kernel; 64x64 tile, Cin % 32 == 0 only.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define LDPB 32  // a [row][32] bf16 LDS tile has 64-byte rows, no padding: the four 16-byte chunks of row r are stored at
                 // chunk position c ^ ((r >> 2) & 3), which makes both the 8-byte stores (two rows fill one 128-byte bank
                 // window) and the ds_read_b128 fragment reads (16 rows -> 16 distinct slots of the 256-byte window)
                 // conflict-free (the padded 80-byte pitch measured SQ_LDS_BANK_CONFLICT ~ 90 % of the LDS cycles)
__device__ __forceinline__ int bf_tile_off(int row, int k8) {  // element offset of the 8-byte unit k8 (0..7) of a row
    return row * LDPB + (((k8 >> 1) ^ ((row >> 2) & 3)) << 3) + ((k8 & 1) << 2);
}

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {  // {bf16 trunc(lo), bf16 trunc(hi)} in one dword
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

template <int NS>
__device__ __forceinline__ void split_store(const f32x4 v, __bf16* __restrict__ dst, const int plane_stride) {
    constexpr int NP = NS == 1 ? 1 : (NS == 3 ? 2 : 3);
    if constexpr (NS == 1) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];  // round to nearest even
        *(bf16x4*)dst = h;
    } else {
        f32x4 r = v;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            u32x2 w;
            w[0] = pack_hi16(r[0], r[1]);
            w[1] = pack_hi16(r[2], r[3]);
            *(u32x2*)(dst + p * plane_stride) = w;
            if (p + 1 < NP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = r[e] - trunc_bf16(r[e]);  // exact
            }
        }
    }
}

template <int NS, int BM = 64, int BN = 64>
__global__ __launch_bounds__(256) void conv_taps_bf_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y,
                                                           const sdt_conv_geom g, const int splitk,
                                                           float* __restrict__ partial, const size_t ysize) {
    constexpr int RA = BM / 32, RB = BN / 32, TM = BM / 64, TN = BN / 64;
    constexpr int NP = NS == 1 ? 1 : (NS == 3 ? 2 : 3);
    constexpr int PLANEA = BM * LDPB, PLANEB = BN * LDPB;
    __shared__ __attribute__((aligned(16))) __bf16 sA[NP * PLANEA];
    __shared__ __attribute__((aligned(16))) __bf16 sB[NP * PLANEB];
    __shared__ int sOut[BM];
    __shared__ int sTap[3 * SDT_MAX_TAPS];
    __shared__ int sLive[SDT_MAX_TAPS + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = g.B * g.Ho * g.Wo;
    const int nmb = (M + BM - 1) / BM;
    const int nnb = (g.Cout + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, nmb * nnb);
    const int m0 = (lin / nnb) * BM;
    const int n0 = (lin % nnb) * BN;

    if (tid < g.ntaps) {
        sTap[tid] = g.dy[tid];
        sTap[SDT_MAX_TAPS + tid] = g.dx[tid];
        sTap[2 * SDT_MAX_TAPS + tid] = g.wt[tid];
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
    const int kv = tid & 7, r0 = tid >> 3;
    int rbH[RA], riy[RA], rix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 32 * i;
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
    if (kv == 0) {  // tap culling, as in conv_taps_kernel
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
    const int nkc = g.Cin / BK;
    const int nsteps_all = ntl * nkc;
    const int step0 = (int)(((long)blockIdx.z * nsteps_all) / splitk);
    const int nsteps = (int)(((long)(blockIdx.z + 1) * nsteps_all) / splitk);

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((unsigned)g.Cout * g.Tw * g.Cin * 4u), 0x00020000);
    // same lean loader as conv_taps_kernel: per-tile row bases + a uniform shift per tap, (tap, chunk) advanced
    // incrementally, the K-chunk offset in the load's scalar operand
    unsigned aoff[RA], boff[RB], abase[RA], bbase[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) abase[i] = (unsigned)(((rbH[i] + riy[i]) * g.Wi + rix[i]) * g.Cin) * 4u;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 32 * i;
        bbase[i] = n < g.Cout ? (unsigned)(n * g.Tw * g.Cin) * 4u : SDT_OOB;
    }
    int cur_tl = -1;
    int nxt_tl = step0 / nkc, nxt_kc = step0 - nxt_tl * nkc;
    f32x4 ra[RA], rb[RB];
    auto load = [&](int step) {
        (void)step;  // consecutive steps: the pair is carried
        const int tl = nxt_tl, kc = nxt_kc;
        if (++nxt_kc == nkc) nxt_kc = 0, ++nxt_tl;
        const int cs = kc * BK * 4;
        if (tl != cur_tl) {
            cur_tl = tl;
            const int t = sLive[tl];
            const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
            const unsigned ashift = (unsigned)((dy * g.Wi + dx) * g.Cin) * 4u + (unsigned)kv * 16u;
            const unsigned bshift = (unsigned)(wt * g.Cin) * 4u + (unsigned)kv * 16u;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                aoff[i] = ok ? abase[i] + ashift : SDT_OOB;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) boff[i] = bbase[i] == SDT_OOB ? SDT_OOB : bbase[i] + bshift;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)aoff[i], cs, 0));
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)boff[i], cs, 0));
    };

    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f, accl[i][j][r] = 0.f;

    // fragment read offsets for the two k16 blocks of a step: logical chunk 2j + h of row (.. + lane&31)
    const int fsw = (lane >> 2) & 3, fh = lane >> 5;
    const int foff[2] = {((fh ^ fsw) << 3), (((2 + fh) ^ fsw) << 3)};
    const __bf16* pa = sA + (wm * (BM / 2) + (lane & 31)) * LDPB;
    const __bf16* pb = sB + (wn * (BN / 2) + (lane & 31)) * LDPB;
    const int wofs = bf_tile_off(r0, kv);  // (row + 32 i) keeps (row >> 2) & 3: the same swizzle for every i

    if (step0 < nsteps) load(step0);
    for (int step = step0; step < nsteps; ++step) {
#pragma unroll
        for (int i = 0; i < RA; ++i) split_store<NS>(ra[i], &sA[wofs + 32 * i * LDPB], PLANEA);
#pragma unroll
        for (int i = 0; i < RB; ++i) split_store<NS>(rb[i], &sB[wofs + 32 * i * LDPB], PLANEB);
        __syncthreads();
        if (step + 1 < nsteps) load(step + 1);
        // MFMA k-slot e of lane half h <-> k = 16j + 8h + e for A and B alike
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm][p] = *(const bf16x8*)(pa + p * PLANEA + tm * 32 * LDPB + foff[j]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn][p] = *(const bf16x8*)(pb + p * PLANEB + tn * 32 * LDPB + foff[j]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], acc[tm][tn], 0, 0, 0);
                    if constexpr (NS >= 3) {
                        accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][NP > 1 ? 1 : 0], accl[tm][tn], 0, 0, 0);
                        accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 1 ? 1 : 0], b[tn][0], accl[tm][tn], 0, 0, 0);
                    }
                    if constexpr (NS == 6) {
                        accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 1 ? 1 : 0], b[tn][NP > 1 ? 1 : 0], accl[tm][tn], 0, 0, 0);
                        accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][NP > 2 ? 2 : 0], accl[tm][tn], 0, 0, 0);
                        accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 2 ? 2 : 0], b[tn][0], accl[tm][tn], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += accl[i][j][r];
    }

    if (splitk > 1) {
        Y = partial + (size_t)blockIdx.z * ysize;
        bias = nullptr;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * (BN / 2) + tn * 32 + (lane & 31);
            const bool nok = n < g.Cout;
            const float bv = (bias != nullptr && nok) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int off = sOut[row];
                if (off >= 0 && nok) Y[(size_t)off + n] = acc[tm][tn][r] + bv;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient.  GEMM with the reduction over output positions m; operands are m-major in HBM
// (dY rows are Cout-contiguous, X rows Cin-contiguous) so LDS tiles are [k=m][row] and MFMA operands
// are read with conflict-free ds_read_b32.  grid = (m-splits, column tiles, Cout tiles); partial
// products are accumulated into dW with fp32 global atomics.
// NS > 0 (VEC4 only): the products run on the bf16 MFMA; each wave splits the fp32 fragments it reads from LDS into
// 1 / 2 / 3 bf16 pieces (see conv_taps_bf_kernel) -- 8 consecutive m per lane and k-slot, gathered with ds_read_b32.
template <int NS>
__device__ __forceinline__ void split_frag(const float (&v)[8], bf16x8 (&out)[NS == 1 ? 1 : (NS == 3 ? 2 : 3)]) {
    constexpr int NP = NS == 1 ? 1 : (NS == 3 ? 2 : 3);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (NS == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) out[0][e] = (__bf16)v[e];
    } else {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            u32x4 w;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = pack_hi16(r[2 * q], r[2 * q + 1]);
            out[p] = __builtin_bit_cast(bf16x8, w);
            if (p + 1 < NP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = r[e] - trunc_bf16(r[e]);
            }
        }
    }
}

// SLAB: the deterministic variant -- every row range writes its partial product with plain stores into its own slab
// dW + range * (Cout * Tw * Cin) and dw_slab_reduce_kernel adds the slabs to the gradient in a fixed order (no atomics, bit-identical
// from run to run).
// (the body is a device function so that the grouped launch below -- many small geometries in one grid -- shares it; ``lin``: the workgroup's
// index within its geometry's grid of ``nblk`` workgroups)
template <int BM, int BN, bool VEC4, int NS, bool SLAB>
__device__ __forceinline__ void conv_dw_body(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dW,
                                             const sdt_conv_geom& g, const int rows_per_split, const int lin) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;      // float4 loads per thread per tile
    constexpr int PA = 1024 / BM, PB = 1024 / BN;  // tile rows covered per pass
    __shared__ __attribute__((aligned(16))) float sA[BK * BM];
    __shared__ __attribute__((aligned(16))) float sB[BK * BN];
    __shared__ int sRow[2][4][BK];
    __shared__ int sTap[3 * SDT_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = g.B * g.Ho * g.Wo;
    // 1-D grid over (row range, column tile, n tile); an XCD gets a contiguous range with the row range slowest, so
    // every workgroup that reads the same dY / X rows runs on the same XCD at about the same time (L2 reuse)
    const int ncb = (g.Cin + BN - 1) / BN;
    const int ncol = VEC4 ? g.ntaps * ncb : (g.ntaps * g.Cin + BN - 1) / BN;
    const int nnt = (g.Cout + BM - 1) / BM;
    const int bx = lin / (ncol * nnt), by = (lin / nnt) % ncol, bz = lin % nnt;
    const int mbeg = bx * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    const int n0 = bz * BM;
    const int tapv = VEC4 ? by / ncb : 0;
    const int c0 = VEC4 ? (by % ncb) * BN : by * BN;  // scalar path: flattened (tap,c)
    const int Ktot = g.ntaps * g.Cin;

    if (tid < g.ntaps) {
        sTap[tid] = g.dy[tid];
        sTap[SDT_MAX_TAPS + tid] = g.dx[tid];
        sTap[2 * SDT_MAX_TAPS + tid] = g.wt[tid];
    }
    auto decode = [&](int step) {  // called by tid < BK
        int m = mbeg + step * BK + tid;
        int offY = -1, bH = 0, iy0 = -(1 << 20), ix0 = 0;
        if (m < mend) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            offY = ((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout;
            bH = b * g.Hi;
            iy0 = oy * g.sy;
            ix0 = ox * g.sx;
        }
        int (*R)[BK] = sRow[step & 1];
        if constexpr (VEC4) {  // byte offsets for the buffer loads (this block's tap is fixed); masked rows -> SDT_OOB
            const int iy = iy0 + g.dy[tapv];
            const int ix = ix0 + g.dx[tapv];
            const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            R[0][tid] = offY >= 0 ? (int)((unsigned)offY * 4u) : (int)SDT_OOB;
            R[1][tid] = ok ? (int)((unsigned)(((bH + iy) * g.Wi + ix) * g.Cin) * 4u) : (int)SDT_OOB;
        } else {
            R[0][tid] = offY;
            R[1][tid] = bH;
            R[2][tid] = iy0;
            R[3][tid] = ix0;
        }
    };

    const int nva = tid % (BM / 4), rra = tid / (BM / 4);
    const int nvb = tid % (BN / 4), rrb = tid / (BN / 4);
    const int nsteps = (mend - mbeg + BK - 1) / BK;
    if (nsteps <= 0) return;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)((unsigned)g.B * g.Hy * g.Wy * g.Cout * 4u), 0x00020000);
    // per-thread column byte offsets; columns past Cout / Cin are masked with SDT_OOB (added to any row offset -> OOB)
    const unsigned colA = (n0 + 4 * nva) < g.Cout ? (unsigned)(n0 + 4 * nva) * 4u : SDT_OOB;
    const unsigned colB = (c0 + 4 * nvb) < g.Cin ? (unsigned)(c0 + 4 * nvb) * 4u : SDT_OOB;
    f32x4 ra[RA], rb[RB];
    auto load = [&](int step) {
        int (*R)[BK] = sRow[step & 1];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int r = rra + PA * i;
            const int offY = R[0][r];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int n = n0 + 4 * nva;
            if constexpr (VEC4) {
                // offY is a byte offset or SDT_OOB; SDT_OOB + SDT_OOB would wrap to 0, hence the saturating add
                const unsigned o = __builtin_elementwise_add_sat((unsigned)offY, colA);  // saturating: OOB + OOB stays out of range
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)o, 0, 0));
                (void)n;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (offY >= 0 && n + e < g.Cout) v[e] = dY[(size_t)offY + n + e];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = rrb + PB * i;
            const int bH = R[1][r], iy0 = R[2][r], ix0 = R[3][r];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (VEC4) {
                const unsigned xo = (unsigned)bH;  // byte offset of the gathered X row for this block's tap, or SDT_OOB
                const unsigned o = __builtin_elementwise_add_sat(xo, colB);
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, 0, 0));
                (void)iy0;
                (void)ix0;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = c0 + 4 * nvb + e;
                    if (j < Ktot) {
                        const int t = j / g.Cin, c = j - t * g.Cin;
                        const int iy = iy0 + sTap[t], ix = ix0 + sTap[SDT_MAX_TAPS + t];
                        if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi)
                            v[e] = X[((size_t)(bH + iy) * g.Wi + ix) * g.Cin + c];
                    }
                }
            }
            rb[i] = v;
        }
    };

    f32x16 acc[TM][TN];
    f32x16 accl[NS >= 3 ? TM : 1][NS >= 3 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (NS >= 3) accl[i][j][r] = 0.f;
            }

    if (tid < BK) decode(0);
    __syncthreads();
    load(0);
    const float* pa = sA + (lane >> 5) * BM + wm * (BM / 2) + (lane & 31);
    const float* pb = sB + (lane >> 5) * BN + wn * (BN / 2) + (lane & 31);
    for (int step = 0; step < nsteps; ++step) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(f32x4*)&sA[(rra + PA * i) * BM + 4 * nva] = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(f32x4*)&sB[(rrb + PB * i) * BN + 4 * nvb] = rb[i];
        if (step + 1 < nsteps && tid < BK) decode(step + 1);
        __syncthreads();
        if (step + 1 < nsteps) load(step + 1);
        if constexpr (NS > 0) {
            constexpr int NP = NS == 1 ? 1 : (NS == 3 ? 2 : 3);
            const float* qa = sA + wm * (BM / 2) + (lane & 31) + 8 * (lane >> 5) * BM;
            const float* qb = sB + wn * (BN / 2) + (lane & 31) + 8 * (lane >> 5) * BN;
#pragma unroll
            for (int j = 0; j < BK / 16; ++j) {  // MFMA k-slot e of lane half h <-> tile row m = 16j + 8h + e
                bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = qa[(16 * j + e) * BM + tm * 32];
                    split_frag<NS>(v, a[tm]);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = qb[(16 * j + e) * BN + tn * 32];
                    split_frag<NS>(v, b[tn]);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], acc[tm][tn], 0, 0, 0);
                        if constexpr (NS >= 3) {
                            accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][NP > 1 ? 1 : 0], accl[tm][tn], 0, 0, 0);
                            accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 1 ? 1 : 0], b[tn][0], accl[tm][tn], 0, 0, 0);
                        }
                        if constexpr (NS == 6) {
                            accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 1 ? 1 : 0], b[tn][NP > 1 ? 1 : 0], accl[tm][tn], 0, 0, 0);
                            accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][NP > 2 ? 2 : 0], accl[tm][tn], 0, 0, 0);
                            accl[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][NP > 2 ? 2 : 0], b[tn][0], accl[tm][tn], 0, 0, 0);
                        }
                    }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = pa[2 * kk * BM + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = pb[2 * kk * BN + tn * 32];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        }
        __syncthreads();
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += accl[i][j][r];
    }

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int jl = wn * (BN / 2) + tn * 32 + (lane & 31);
            int wt, c;
            bool cok;
            if constexpr (VEC4) {
                c = c0 + jl;
                cok = c < g.Cin;
                wt = sTap[2 * SDT_MAX_TAPS + tapv];
            } else {
                const int j = c0 + jl;
                cok = j < Ktot;
                const int t = cok ? j / g.Cin : 0;
                c = j - t * g.Cin;
                wt = sTap[2 * SDT_MAX_TAPS + t];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (cok && n < g.Cout) {
                    const size_t o = ((size_t)n * g.Tw + wt) * g.Cin + c;
                    if constexpr (SLAB)
                        dW[(size_t)bx * ((size_t)g.Cout * g.Tw * g.Cin) + o] = acc[tm][tn][r];
                    else
                        atomicAdd(&dW[o], acc[tm][tn][r]);
                }
            }
        }
}

template <int BM, int BN, bool VEC4, int NS = 0, bool SLAB = false>
__global__ __launch_bounds__(256) void conv_dw_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                      float* __restrict__ dW, const sdt_conv_geom g,
                                                      const int rows_per_split) {
    conv_dw_body<BM, BN, VEC4, NS, SLAB>(X, dY, dW, g, rows_per_split, xcd_remap(blockIdx.x, (int)gridDim.x));
}

// ---------------------------------------------------------------------------------------------
// GROUPED deterministic weight gradient: the weight gradients of many small layers -- the generator's sixteen Conv1d blocks, 0.8 GFLOP each --
// in ONE grid + one ordered reduce instead of a kernel and a reduce per layer.  Launched alone, each of those layers split its 2048 rows over
// ~32 row ranges to fill the chip (1536 workgroups), i.e. wrote and re-read 32 slabs of its 786 KB gradient: 0.8 GB of slab traffic per step, which
// is what its 17 us were spent on.  In one grid the layers fill the chip together and a layer needs 2 row ranges.
#define SDT_DW_GROUP_MAX 24
struct dw_group_item {
    sdt_conv_geom g;
    int32_t rows, nsplit, block0, nblocks;
    int64_t slab_off, n, vec0;  // floats: start of this layer's slabs; elements of its gradient; first 16-byte vector of it in the reduce's index space
};
struct dw_group_args {
    const float* x[SDT_DW_GROUP_MAX];
    const float* dy[SDT_DW_GROUP_MAX];
    float* dw[SDT_DW_GROUP_MAX];
    const dw_group_item* items;
    float* slabs;
    int n;
};

__global__ __launch_bounds__(256) void conv_dw_group_kernel(const dw_group_args A) {
    int i = 0;
    while (i + 1 < A.n && (int)blockIdx.x >= A.items[i + 1].block0) ++i;
    const dw_group_item& it = A.items[i];
    conv_dw_body<64, 64, true, 0, true>(A.x[i], A.dy[i], A.slabs + it.slab_off, it.g, it.rows, xcd_remap((int)blockIdx.x - it.block0, it.nblocks));
}

__global__ __launch_bounds__(256) void dw_slab_reduce_group_kernel(const dw_group_args A, const int64_t total_vec) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * 256) {
        int i = 0;
        while (i + 1 < A.n && v >= A.items[i + 1].vec0) ++i;
        const dw_group_item& it = A.items[i];
        const int64_t e = 4 * (v - it.vec0);
        const float* sl = A.slabs + it.slab_off + e;
        f32x4 a = *(const f32x4*)sl;
        for (int sidx = 1; sidx < it.nsplit; ++sidx) a += *(const f32x4*)(sl + (size_t)sidx * it.n);
        *(f32x4*)(A.dw[i] + e) += a;
    }
}

// dw[i] += slab[0][i] + slab[1][i] + ... in that order (deterministic weight gradient)
__global__ __launch_bounds__(256) void dw_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, size_t n, int nsplit) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((n & 3) == 0) {
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            f32x4 a = *(const f32x4*)(slabs + 4 * i);
            for (int sidx = 1; sidx < nsplit; ++sidx) a += *(const f32x4*)(slabs + (size_t)sidx * n + 4 * i);
            *(f32x4*)(dw + 4 * i) += a;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float a = slabs[i];
            for (int sidx = 1; sidx < nsplit; ++sidx) a += slabs[(size_t)sidx * n + i];
            dw[i] += a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ W, float* __restrict__ Wt,
                                                               int cout, int taps, int cin) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int co = co0 + ty + 8 * i, ci = ci0 + tx;
        tile[ty + 8 * i][tx] = (co < cout && ci < cin) ? W[((size_t)co * taps + t) * cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ci = ci0 + ty + 8 * i, co = co0 + tx;
        if (ci < cin && co < cout) Wt[((size_t)ci * taps + t) * cout + co] = tile[tx][ty + 8 * i];
    }
}

// one bf16 copy (round to nearest even), or -- planes == 3 -- the exact three-way split the split-fp32 kernels make in their loaders (convbf.hip
// x3_stage: the same conversions and the same exact fp32 differences, so a launch on pre-split weights is bit-identical to one that splits itself)
__device__ __forceinline__ void wt_store16(__bf16* dst, const size_t o, const size_t nel, const int planes, const float v) {
    const __bf16 hi = (__bf16)v;
    dst[o] = hi;
    if (planes == 3) {
        const float r = v - (float)hi;
        const __bf16 mid = (__bf16)r;
        const float t = r - (float)mid;
        dst[nel + o] = mid;
        dst[2 * nel + o] = (__bf16)t;
    }
}

__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const sdt_wt_desc* __restrict__ table, int n_layers) {
    __shared__ float tile[32][33];
    const int bid = blockIdx.x;
    int l = 0;
    while (l + 1 < n_layers && table[l + 1].tile_begin <= bid) ++l;  // <= a few dozen layers
    const sdt_wt_desc d = table[l];
    const int nci = (d.cin + 31) >> 5, nco = (d.cout + 31) >> 5;
    const size_t nel = (size_t)d.cout * d.taps * d.cin;
    int rem = bid - d.tile_begin;
    const int ci0 = (rem % nci) * 32;
    rem /= nci;
    const int co0 = (rem % nco) * 32, t = rem / nco;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty + 8 * i, ci = ci0 + tx;
        const bool ok = co < d.cout && ci < d.cin;
        const size_t o = ((size_t)co * d.taps + t) * d.cin + ci;
        const float v = ok ? d.w[o] : 0.f;
        tile[ty + 8 * i][tx] = v;
        if (ok && d.w16 != nullptr) wt_store16((__bf16*)d.w16, o, nel, d.planes, v);  // bf16 copy of W itself (the bf16-storage path's forward operand) / its split
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty + 8 * i, co = co0 + tx;
        if (ci < d.cin && co < d.cout) {
            const size_t o = ((size_t)ci * d.taps + t) * d.cout + co;
            const float v = tile[tx][ty + 8 * i];
            if (d.wt != nullptr) d.wt[o] = v;
            if (d.wt16 != nullptr) wt_store16((__bf16*)d.wt16, o, nel, d.planes, v);  // and of the mirror (its input-gradient operand)
        }
    }
}

// out[c] += sum over rows of x[r][c], in a FIXED order (bias gradients; no atomics: bit-identical from run to run).  One workgroup per
// 16 columns; thread (column, q = 0..15) sums the rows r = q (mod 16) with eight independent partial sums (rows 16 i + q, i mod 8), which
// are then combined in a fixed tree -- 128 row classes in flight per column instead of one dependent chain (the head's bias gradient, 2048
// rows of 274 columns, is pure load latency: 25 us with 5 workgroups of 32 classes, 6 us like this).
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int C) {
    __shared__ float part[16][17];
    const int cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, q = threadIdx.x >> 4;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (c < C) {
        int64_t r = q;
        for (; r + 112 < rows; r += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[(r + 16 * u) * C + c];
        }
        for (int u = 0; r < rows; r += 16, ++u) acc[u] += x[r * C + c];
    }
    part[q][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0 && c < C) {
        float s[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] = part[u][cl];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) s[u] += s[u + w];
        out[c] += s[0];
    }
}

// ---------------------------------------------------------------------------------------------
static int check_geom(const sdt_conv_geom* g) {
    SDT_CHECK_ARG(g != nullptr, "null geometry");
    SDT_CHECK_ARG(g->B > 0 && g->Hi > 0 && g->Wi > 0 && g->Cin > 0 && g->Ho > 0 && g->Wo > 0 && g->Cout > 0, "non-positive dims");
    SDT_CHECK_ARG(g->ntaps > 0 && g->ntaps <= SDT_MAX_TAPS && g->Tw >= 1, "bad tap count");
    SDT_CHECK_ARG((int64_t)g->B * g->Hy * g->Wy * g->Cout < (1ll << 31), "output tensor too large for 32-bit offsets");
    SDT_CHECK_ARG((int64_t)g->B * g->Ho * g->Wo < (1ll << 31), "too many output positions");
    SDT_CHECK_ARG((g->Ho - 1) * g->osy + g->ooy < g->Hy && (g->Wo - 1) * g->osx + g->oox < g->Wy, "output grid exceeds Y");
    for (int t = 0; t < g->ntaps; ++t) SDT_CHECK_ARG(g->wt[t] >= 0 && g->wt[t] < g->Tw, "weight tap out of range");
    const int64_t lim = (1ll << 31) - 65536;  // 32-bit byte offsets + SDT_OOB masking in the buffer-load paths
    SDT_CHECK_ARG((int64_t)g->B * g->Hi * g->Wi * g->Cin * 4 < lim, "input tensor exceeds 2 GiB (split the batch)");
    SDT_CHECK_ARG((int64_t)g->B * g->Hy * g->Wy * g->Cout * 4 < lim, "output tensor exceeds 2 GiB (split the batch)");
    SDT_CHECK_ARG((int64_t)g->Cout * g->Tw * g->Cin * 4 < lim, "weight tensor exceeds 2 GiB");
    return SDT_OK;
}

static int g_conv_math = SDT_MATH_F32;
extern "C" int sdt_set_conv_math(int mode) {
    SDT_CHECK_ARG(mode == SDT_MATH_F32 || mode == SDT_MATH_BF16 || mode == SDT_MATH_BF16X3 || mode == SDT_MATH_BF16X6, "unknown math mode");
    g_conv_math = mode;
    return SDT_OK;
}
extern "C" int sdt_get_conv_math(void) { return g_conv_math; }

static const norm_bwd_args kNoNormBwd = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};

static geom_pack pack_of(const sdt_conv_geom* const* gs, int n) {
    geom_pack gp;
    for (int i = 0; i < SDT_MAX_CLASSES; ++i) gp.g[i] = *gs[i < n ? i : 0];
    gp.mrot = 0;
    return gp;
}

// conv1d_small_kernel takes a launch when every class is 1-D with full 32-channel chunks and 64-column tiles, the K slice of a workgroup
// is at most SMALL1D_NS steps, the arithmetic is exact fp32 and no statistics epilogue is asked for
#define SMALL1D_NS 8
static bool g_small1d = true;
#ifdef SDT_TUNING
extern "C" int sdt_debug_set_small1d(int on) {
    g_small1d = on != 0;
    return SDT_OK;
}
#endif
static bool small1d_ok(bool vec4, const sdt_conv_geom* const* gs, int ncls, int splitk, const norm_bwd_args& nb) {
    if (!g_small1d || !vec4 || g_conv_math != SDT_MATH_F32 || nb.sums != nullptr) return false;
    for (int c = 0; c < ncls; ++c) {
        const sdt_conv_geom& g = *gs[c];
        if (g.Hi != 1 || g.Ho != 1 || g.Hy != 1 || g.Cin % BK != 0 || g.Cout % 64 != 0) return false;
        const int nsteps_all = g.ntaps * (g.Cin / BK);
        if (cdiv(nsteps_all, splitk) > SMALL1D_NS) return false;
        for (int t = 0; t < g.ntaps; ++t)
            if (g.dy[t] != 0) return false;
    }
    return true;
}

// One launch for ``ncls`` geometries that share X, W and Y (blockIdx.y = class; the x extent covers the largest class).
template <int BM, int BN>
static void launch_taps(bool vec4, const float* x, const float* w, const float* bias, float* y,
                        const sdt_conv_geom* const* gs, int ncls, int splitk, float* partial, const norm_bwd_args& nb, hipStream_t s) {
    const sdt_conv_geom& g = *gs[0];
    const size_t ysize = (size_t)g.B * g.Hy * g.Wy * g.Cout;
    int tiles = 0;
    for (int c = 0; c < ncls; ++c) tiles = std::max(tiles, cdiv(gs[c]->B * gs[c]->Ho * gs[c]->Wo, BM) * cdiv(gs[c]->Cout, BN));
    dim3 grid(tiles, ncls, splitk);
    geom_pack gp = pack_of(gs, ncls);
    if (vec4 && ncls == 1 && splitk == 1 && g.ntaps >= 9) {  // see the kernel: tile rotation for tall valid-correlation launches
        int dymin = g.dy[0], dymax = g.dy[0];
        for (int t = 1; t < g.ntaps; ++t) dymin = std::min(dymin, g.dy[t]), dymax = std::max(dymax, g.dy[t]);
        const int nnb = cdiv(g.Cout, BN), nmb = cdiv(g.B * g.Ho * g.Wo, BM);
        if (dymax - dymin + 1 > g.Hi && 32 % nnb == 0 && 32 / nnb >= 4) {
            const int S = 32 / nnb;
            const double per_image = (double)g.Ho * g.Wo / BM, per_cu = std::max(1.0, (double)nmb * nnb / 256.0);
            gp.mrot = (int)std::max(1l, lround(per_image / per_cu)) % S;
        }
    }
    if (vec4 && g_conv_math != SDT_MATH_F32 && ncls == 1 && nb.sums == nullptr) {
        switch (g_conv_math) {
            case SDT_MATH_BF16: hipLaunchKernelGGL((conv_taps_bf_kernel<1, BM, BN>), dim3(tiles, 1, splitk), dim3(256), 0, s, x, w, bias, y, g, splitk, partial, ysize); break;
            case SDT_MATH_BF16X3: hipLaunchKernelGGL((conv_taps_bf_kernel<3, BM, BN>), dim3(tiles, 1, splitk), dim3(256), 0, s, x, w, bias, y, g, splitk, partial, ysize); break;
            default: hipLaunchKernelGGL((conv_taps_bf_kernel<6, BM, BN>), dim3(tiles, 1, splitk), dim3(256), 0, s, x, w, bias, y, g, splitk, partial, ysize); break;
        }
        return;
    }
    if (small1d_ok(vec4, gs, ncls, splitk, nb) && BM == 64 && BN == 64) {
        hipLaunchKernelGGL((conv1d_small_kernel<SMALL1D_NS>), grid, dim3(256), 0, s, x, w, bias, y, gp, splitk, partial, ysize);
        return;
    }
#define SDT_TAPS(PRIO) hipLaunchKernelGGL((conv_taps_kernel<BM, BN, true, PRIO>), grid, dim3(256), 0, s, x, w, bias, y, gp, splitk, partial, ysize, (double*)nullptr, 0, nb)
    if (nb.sums != nullptr) {  // host-checked: vector path, fp32 math, no split-K
        hipLaunchKernelGGL((conv_taps_kernel<BM, BN, true, 0, 2>), grid, dim3(256), 0, s, x, w, bias, y, gp, splitk, partial, ysize, (double*)nullptr, 0, nb);
        return;
    }
#ifdef SDT_TUNING  // ablation / A-B instantiations (some compute WRONG results by design): tuning build only
    static const int prio = getenv("SDT_CONV_PRIO") ? atoi(getenv("SDT_CONV_PRIO")) : 0;
    if (vec4 && prio == 1) SDT_TAPS(1);
    else if (vec4 && prio == 2) SDT_TAPS(2);
    else if (vec4 && prio == 3) SDT_TAPS(3);   // ablation: no global loads in the K loop (wrong results, timing only)
    else if (vec4 && prio == 4) SDT_TAPS(4);   // ablation: no barriers (wrong results, timing only)
    else if (vec4 && prio == 5) SDT_TAPS(5);   // experiment: two accumulators per 32x32 sub-tile
    else if (vec4 && prio == 6) SDT_TAPS(6);   // ablation: two accumulators, no global loads
    else if (vec4 && prio == 7) SDT_TAPS(7);   // ablation: no global loads, no LDS stores
    else if (vec4 && prio == 8) SDT_TAPS(8);   // ablation: no global loads, no LDS traffic at all
    else if (vec4 && prio == 9) SDT_TAPS(9);   // ablation: as 8, and no barriers after the first K step
    else if (vec4 && prio == 10 && BM == 64 && BN == 64 && ncls == 1)  // experiment: asynchronous global->LDS staging
        hipLaunchKernelGGL(conv_taps_dma_kernel, dim3(tiles, 1, splitk), dim3(256), 0, s, x, w, bias, y, g, splitk, partial, ysize);
    else if (vec4 && prio == 11) SDT_TAPS(11);  // A/B: tap culling also on 1-D launches (the previous behaviour)
    else if (vec4 && prio == 12) SDT_TAPS(12);  // A/B: live taps in table (dy-major) order
    else if (vec4 && prio == 14) SDT_TAPS(14);  // A/B: row-residue tap order on every launch
    else if (vec4 && prio == 16) SDT_TAPS(16);  // A/B: tap culling also on launches with <= 4 taps
    else if (vec4 && prio == 20) SDT_TAPS(20);  // experiment: double-buffered LDS, one barrier per K step
    else if (vec4 && prio == 30) SDT_TAPS(30);  // per-workgroup timeline stamps (tools/debug/taps_timeline.py)
    else if (vec4 && prio == 31) SDT_TAPS(31);  // experiment: prologue / epilogue at priority 3
    else if (vec4 && prio == 32) SDT_TAPS(32);  // experiment: 31 + the feeding part of every K step at priority 2
    else if (vec4 && prio == 33) SDT_TAPS(33);  // experiment: only the feeding part of every K step at priority 2
    else
#endif
    if (vec4)
        SDT_TAPS(0);
    else
        hipLaunchKernelGGL((conv_taps_kernel<BM, BN, false>), grid, dim3(256), 0, s, x, w, bias, y, gp, splitk, partial, ysize, (double*)nullptr, 0, nb);
#undef SDT_TAPS
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ y, size_t n, int cout, int splitk) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = partial[i];
        for (int z = 1; z < splitk; ++z) s += partial[(size_t)z * n + i];
        if (bias != nullptr) s += bias[i % cout];
        y[i] = s;
    }
}

// tile selection: returns BM*1000 + BN (+1 when the 16-B vector loader is usable)
static int taps_variant(const sdt_conv_geom* g, bool aligned) {
    const int vec4 = ((g->Cin % BK == 0) && aligned) ? 1 : 0;  // whole 32-channel K chunks: no channel mask in the loop
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
#ifdef SDT_TUNING
    static const int forced = getenv("SDT_CONV_TILE") ? atoi(getenv("SDT_CONV_TILE")) : 0;
    if (forced == 64064 || forced == 64128 || forced == 128064 || forced == 128128) return forced * 10 + vec4;
#endif
    // Measured on MI355X (profiles/r01_tile_sweep.txt): the 64x64 tile (54 VGPR + 16 AGPR -> 7 waves/SIMD) beats the
    // 128-wide tiles on every layer of the hot path (92-121 vs 55-113 TFLOP/s): the fp32 MFMA is slow enough that LDS
    // reuse is irrelevant, while occupancy hides the gather latency and the small tile quantises better over 256 CUs.
    // (also in the bf16 modes: larger tiles win some isolated layers but lose 3 % over the whole step)
    return 64064 * 10 + vec4;
}
extern "C" int sdt_conv_taps_variant(const sdt_conv_geom* g) { return g ? taps_variant(g, true) : SDT_ERR_ARG; }
// 1 when a launch of these classes with this K split runs conv1d_small_kernel (16-byte aligned operands assumed), else 0: lets the host
// label its per-launch timings with the kernel that a trace will show
extern "C" int sdt_conv1d_small_used(const sdt_conv_geom* geoms, int ncls, int splitk) {
    if (!geoms || ncls < 1 || ncls > SDT_MAX_CLASSES || splitk < 1) return SDT_ERR_ARG;
    const sdt_conv_geom* gs[SDT_MAX_CLASSES];
    for (int c = 0; c < ncls; ++c) gs[c] = geoms + c;
    const int var = taps_variant(gs[0], true);
    return (var / 10 == 64064 && small1d_ok(var % 10, gs, ncls, splitk, kNoNormBwd)) ? 1 : 0;
}

// Suggested split of the K loop for launches that cannot fill 256 CUs with output tiles (the 1-D stage:
// M = B*T <= 2048 rows): enough slices for >= 2 workgroups per CU while keeping >= 4 K-steps per slice.
extern "C" int sdt_conv_taps_splitk_hint(const sdt_conv_geom* g) {
    if (!g) return SDT_ERR_ARG;
    const int var = taps_variant(g, true);
    const int bm = var / 10 / 1000, bn = var / 10 % 1000;
    const int64_t tiles = cdiv64((int64_t)g->B * g->Ho * g->Wo, bm) * cdiv(g->Cout, bn);
    // 1-D launches keep splitting up to ~1024 workgroups: with 192-767 tiles (one to three per CU of a kernel that fits seven) every workgroup
    // walks its whole K alone -- the 256-tile first block of the paired pose-encoder pass (242 input channels, scalar loader): 78 us unsplit
    const bool one_d = g->Hi == 1 && g->Ho == 1;
    if (tiles >= (one_d ? 768 : 192)) return 1;
    const int nsteps = (var % 10) ? g->ntaps * cdiv(g->Cin, BK) : cdiv(g->ntaps * g->Cin, BK);
    int k = (int)std::min<int64_t>(std::min<int64_t>(cdiv64(tiles >= 192 ? 1024 : 512, tiles), nsteps / 4), 16);
    return std::max(k, 1);
}

static int taps_dispatch(const float* x, const float* w, const float* bias, float* y, const sdt_conv_geom* const* gs, int ncls,
                         int splitk, float* partial, const norm_bwd_args& nb, void* stream) {
    const int var = taps_variant(gs[0], (((uintptr_t)x | (uintptr_t)w) % 16) == 0);
    const bool vec4 = var % 10;
    hipStream_t s = (hipStream_t)stream;
    switch (var / 10) {
#ifdef SDT_TUNING  // the tile sweep of profiles/r01_tile_sweep.txt; production uses 64x64 everywhere
        case 128064: launch_taps<128, 64>(vec4, x, w, bias, y, gs, ncls, splitk, partial, nb, s); break;
        case 64128: launch_taps<64, 128>(vec4, x, w, bias, y, gs, ncls, splitk, partial, nb, s); break;
        case 128128: launch_taps<128, 128>(vec4, x, w, bias, y, gs, ncls, splitk, partial, nb, s); break;
#endif
        default: launch_taps<64, 64>(vec4, x, w, bias, y, gs, ncls, splitk, partial, nb, s); break;
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_conv_taps_splitk_f32(const float* x, const float* w, const float* bias, float* y,
                                        const sdt_conv_geom* g, int splitk, float* partial, void* stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    SDT_CHECK_ARG(x && w && y, "null pointer");
    SDT_CHECK_ARG(splitk >= 1 && splitk <= 64 && (splitk == 1 || partial != nullptr), "bad split-K arguments");
    return taps_dispatch(x, w, bias, y, &g, 1, splitk, partial, kNoNormBwd, stream);
}

// Input gradient of a (possibly strided) convolution in ONE launch: ``ncls`` output parity classes (sdt_conv_geom each; same
// X = dY, W = transposed weights and Y = dX tensor, disjoint output positions).  Optionally the epilogue accumulates the
// statistics of the normalisation backward that consumes dX (see sdt_norm_bwd in include/sdt_hip.h).
extern "C" int sdt_conv_taps_multi_f32(const float* x, const float* w, float* y, const sdt_conv_geom* geoms, int ncls, int splitk,
                                       float* partial, const sdt_norm_bwd* nbw, void* stream) {
    SDT_CHECK_ARG(geoms && ncls >= 1 && ncls <= SDT_MAX_CLASSES, "1..4 geometries per launch");
    const sdt_conv_geom* gs[SDT_MAX_CLASSES];
    for (int c = 0; c < ncls; ++c) {
        gs[c] = geoms + c;
        int rc = check_geom(gs[c]);
        if (rc) return rc;
        SDT_CHECK_ARG(gs[c]->B == gs[0]->B && gs[c]->Hi == gs[0]->Hi && gs[c]->Wi == gs[0]->Wi && gs[c]->Cin == gs[0]->Cin &&
                          gs[c]->Hy == gs[0]->Hy && gs[c]->Wy == gs[0]->Wy && gs[c]->Cout == gs[0]->Cout && gs[c]->Tw == gs[0]->Tw,
                      "the classes of one launch must share the X, W and Y tensors");
    }
    SDT_CHECK_ARG(x && w && y, "null pointer");
    SDT_CHECK_ARG(splitk >= 1 && splitk <= 64 && (splitk == 1 || partial != nullptr), "bad split-K arguments");
    SDT_CHECK_ARG(g_conv_math == SDT_MATH_F32 || (ncls == 1 && nbw == nullptr), "multi-class / fused-statistics launches exist for fp32 math only");
    norm_bwd_args nb = kNoNormBwd;
    if (nbw != nullptr) {
        SDT_CHECK_ARG(nbw->y && nbw->mean && nbw->rstd && nbw->sums, "null pointer in sdt_norm_bwd");
        SDT_CHECK_ARG(splitk == 1 && (gs[0]->Cin % BK == 0), "fused backward statistics need an unsplit vector-path launch");
        SDT_CHECK_ARG(nbw->groups == 1 || nbw->groups == gs[0]->B, "groups must be 1 (BatchNorm) or B (InstanceNorm)");
        for (int c = 0; c < ncls; ++c)
            SDT_CHECK_ARG(nbw->groups == 1 ? (int64_t)gs[c]->B * gs[c]->Ho * gs[c]->Wo >= 64 : gs[c]->Ho * gs[c]->Wo >= 64,
                          "a group must span at least one 64-row tile");
        // (the epilogue reads y through a buffer resource: 32-bit byte offsets)
        SDT_CHECK_ARG((int64_t)gs[0]->B * gs[0]->Hy * gs[0]->Wy * gs[0]->Cout * 4 < (1ll << 31) - 4, "output tensor too large for the fused backward statistics");
        nb = {(const float*)nbw->y, nbw->mean, nbw->rstd, nbw->gamma, nbw->beta, nbw->sums, nbw->slope, nbw->groups};
    }
    return taps_dispatch(x, w, nullptr, y, gs, ncls, splitk, partial, nb, stream);
}

// Forward conv + per-(group, channel) sum / sum-of-squares of its output in the epilogue (64x64 tile, vector path, fp32 math).
extern "C" int sdt_conv_taps_stats_supported(const sdt_conv_geom* g, int rows_per_group) {
    if (!g || check_geom(g) || rows_per_group < 64) return 0;
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
    return (g->Cin % BK == 0) && g_conv_math == SDT_MATH_F32 && M % rows_per_group == 0 &&
           sdt_conv_taps_splitk_hint(g) == 1 && g->osy == 1 && g->osx == 1 && g->ooy == 0 && g->oox == 0 && g->Hy == g->Ho &&
           g->Wy == g->Wo;
}
extern "C" int sdt_conv_taps_stats_f32(const float* x, const float* w, const float* bias, float* y, const sdt_conv_geom* g,
                                       double* stats, int rows_per_group, void* stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    SDT_CHECK_ARG(x && w && y && stats, "null pointer");
    SDT_CHECK_ARG(sdt_conv_taps_stats_supported(g, rows_per_group), "geometry not supported by the fused-statistics epilogue");
    SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w) % 16) == 0, "x and w must be 16-byte aligned");
    const int M = g->B * g->Ho * g->Wo;
    const size_t ysize = (size_t)g->B * g->Hy * g->Wy * g->Cout;
    dim3 grid(cdiv(M, 64) * cdiv(g->Cout, 64), 1, 1);
    const geom_pack gp = pack_of(&g, 1);
#ifdef SDT_TUNING
    static const int order = getenv("SDT_CONV_PRIO") ? atoi(getenv("SDT_CONV_PRIO")) : 0;  // A/B only
    if (order == 12)
        hipLaunchKernelGGL((conv_taps_kernel<64, 64, true, 12, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, gp, 1,
                           (float*)nullptr, ysize, stats, rows_per_group, kNoNormBwd);
    else if (order == 14)
        hipLaunchKernelGGL((conv_taps_kernel<64, 64, true, 14, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, gp, 1,
                           (float*)nullptr, ysize, stats, rows_per_group, kNoNormBwd);
    else
#endif
        hipLaunchKernelGGL((conv_taps_kernel<64, 64, true, 0, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, gp, 1,
                           (float*)nullptr, ysize, stats, rows_per_group, kNoNormBwd);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_conv_taps_f32(const float* x, const float* w, const float* bias, float* y,
                                 const sdt_conv_geom* g, void* stream) {
    return sdt_conv_taps_splitk_f32(x, w, bias, y, g, 1, nullptr, stream);
}

extern "C" int sdt_splitk_reduce_f32(const float* partial, const float* bias, float* y, int64_t n, int cout, int splitk,
                                     void* stream) {
    SDT_CHECK_ARG(partial && y && n > 0 && cout > 0 && splitk >= 1, "bad argument");
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 256), 2048));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partial, bias, y, (size_t)n, cout, splitk);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

template <int BM, int BN>
static void dw_split(bool vec4, const sdt_conv_geom& g, int& nsplit, int& rows, int& coltiles, int& ntiles) {
    const int M = g.B * g.Ho * g.Wo;
    coltiles = vec4 ? g.ntaps * cdiv(g.Cin, BN) : cdiv(g.ntaps * g.Cin, BN);
    ntiles = cdiv(g.Cout, BM);
    const int tiles = coltiles * ntiles;
    // All workgroups of a launch are co-resident (8 per CU fit) and the kernel is MFMA-bound, so the launch lasts as long as the
    // fullest CU: pick the row split whose workgroup count sits just BELOW a multiple of 256 (cdiv(1536, tiles) put 6.05-6.19
    // workgroups on a CU for the 3x3 layers: 13-16 % of the launch spent with 7 workgroups on a few CUs and 6 on the rest).
    const int max_split = std::max(1, M / (4 * BK));  // at least 4 K-steps per workgroup: bounds the atomic traffic
#ifdef SDT_TUNING  // A/B: the round-1 rule
    static const int old_rule = getenv("SDT_DW_SPLIT_OLD") ? atoi(getenv("SDT_DW_SPLIT_OLD")) : 0;
    if (old_rule) {
        nsplit = std::max(1, std::min(cdiv(1536, tiles), max_split));
        rows = cdiv(cdiv(M, nsplit), BK) * BK;
        nsplit = cdiv(M, rows);
        return;
    }
#endif
    int best = 1, best_rows = cdiv(M, BK) * BK;
    double best_cost = 1e30;
    for (int ns = std::max(1, 1024 / tiles); ns <= std::max(1, 2048 / tiles); ++ns) {
        const int n0 = std::min(ns, max_split);
        const int r = cdiv(cdiv(M, n0), BK) * BK;
        const int n1 = cdiv(M, r);
        const double per_cu = (double)n1 * tiles / 256.0;
        // time ~ ceil(per_cu) * (work per workgroup ~ r); a mild preference for ~6 workgroups per CU (atomic traffic vs tail)
        const double cost = std::ceil(per_cu) * (double)r * (1.0 + 0.01 * std::fabs(per_cu - 6.0));
        if (cost < best_cost) best_cost = cost, best = n1, best_rows = r;
    }
    nsplit = best;
    rows = best_rows;
}

// slabs != nullptr: deterministic (plain stores into nsplit slabs + an ordered reduce), fp32 math only
template <int BM, int BN>
static void launch_dw(bool vec4, const float* x, const float* dy, float* dw, const sdt_conv_geom& g, hipStream_t s, float* slabs = nullptr) {
    int nsplit, rows, coltiles, ntiles;
    dw_split<BM, BN>(vec4, g, nsplit, rows, coltiles, ntiles);
    dim3 grid(nsplit * coltiles * ntiles);
    if (slabs != nullptr) {
        if (vec4)
            hipLaunchKernelGGL((conv_dw_kernel<BM, BN, true, 0, true>), grid, dim3(256), 0, s, x, dy, slabs, g, rows);
        else
            hipLaunchKernelGGL((conv_dw_kernel<BM, BN, false, 0, true>), grid, dim3(256), 0, s, x, dy, slabs, g, rows);
        const size_t n = (size_t)g.Cout * g.Tw * g.Cin;
        const unsigned rgrid = (unsigned)std::max<size_t>(1, std::min<size_t>((n / 4 + 255) / 256, 2048));
        hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3(rgrid), dim3(256), 0, s, (const float*)slabs, dw, n, nsplit);
        return;
    }
    if (vec4 && g_conv_math == SDT_MATH_BF16)
        hipLaunchKernelGGL((conv_dw_kernel<BM, BN, true, 1>), grid, dim3(256), 0, s, x, dy, dw, g, rows);
    else if (vec4 && g_conv_math == SDT_MATH_BF16X3)
        hipLaunchKernelGGL((conv_dw_kernel<BM, BN, true, 3>), grid, dim3(256), 0, s, x, dy, dw, g, rows);
    else if (vec4 && g_conv_math == SDT_MATH_BF16X6)
        hipLaunchKernelGGL((conv_dw_kernel<BM, BN, true, 6>), grid, dim3(256), 0, s, x, dy, dw, g, rows);
    else if (vec4)
        hipLaunchKernelGGL((conv_dw_kernel<BM, BN, true>), grid, dim3(256), 0, s, x, dy, dw, g, rows);
    else
        hipLaunchKernelGGL((conv_dw_kernel<BM, BN, false>), grid, dim3(256), 0, s, x, dy, dw, g, rows);
}

static int dw_variant(const sdt_conv_geom* g, bool aligned) {
    const int vec4 = ((g->Cin % 4 == 0) && (g->Cout % 4 == 0) && aligned) ? 1 : 0;
#ifdef SDT_TUNING
    static const int forced = getenv("SDT_DW_TILE") ? atoi(getenv("SDT_DW_TILE")) : 0;
    if (forced == 64064 || forced == 128064 || forced == 64128 || forced == 128128) return forced * 10 + vec4;
#endif
    // 64x64 everywhere (42 VGPR + 16 AGPR -> 8 waves/SIMD): on par or better than the larger tiles on every layer and
    // a quarter of the atomic traffic per workgroup (profiles/r01_tile_sweep.txt)
    return 64064 * 10 + vec4;
}
extern "C" int sdt_conv_dw_variant(const sdt_conv_geom* g) { return g ? dw_variant(g, true) : SDT_ERR_ARG; }

extern "C" int sdt_conv_dw_f32(const float* x, const float* dy, float* dw, const sdt_conv_geom* g, void* stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    SDT_CHECK_ARG(x && dy && dw, "null pointer");
    const int var = dw_variant(g, (((uintptr_t)x | (uintptr_t)dy) % 16) == 0);
    const bool vec4 = var % 10;
    hipStream_t s = (hipStream_t)stream;
    switch (var / 10) {
#ifdef SDT_TUNING
        case 128128: launch_dw<128, 128>(vec4, x, dy, dw, *g, s); break;
        case 128064: launch_dw<128, 64>(vec4, x, dy, dw, *g, s); break;
        case 64128: launch_dw<64, 128>(vec4, x, dy, dw, *g, s); break;
#endif
        default: launch_dw<64, 64>(vec4, x, dy, dw, *g, s); break;
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

// Deterministic weight gradient (fp32 MFMA): same tiling as sdt_conv_dw_f32, but every row range stores its partial product into
// its own slab of `workspace` and the slabs are added to dw in a fixed order.  workspace >= sdt_conv_dw_workspace_bytes(g).
extern "C" int64_t sdt_conv_dw_workspace_bytes(const sdt_conv_geom* g) {
    if (check_geom(g)) return -1;
    int nsplit, rows, coltiles, ntiles;
    dw_split<64, 64>((g->Cin % 4 == 0) && (g->Cout % 4 == 0), *g, nsplit, rows, coltiles, ntiles);
    return (int64_t)nsplit * g->Cout * g->Tw * g->Cin * (int64_t)sizeof(float);
}

extern "C" int sdt_conv_dw_det_f32(const float* x, const float* dy, float* dw, const sdt_conv_geom* g, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    SDT_CHECK_ARG(x && dy && dw && workspace, "null pointer");
    SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace | (uintptr_t)dw) % 16) == 0, "operands must be 16-byte aligned");
    SDT_CHECK_ARG(workspace_bytes >= sdt_conv_dw_workspace_bytes(g), "workspace too small");
    // every element of every slab has to be written by some workgroup: each weight tap appears exactly once in the tap table
    SDT_CHECK_ARG(g->ntaps == g->Tw, "deterministic weight gradient needs a tap table that covers every weight tap once");
    for (int t = 0; t < g->ntaps; ++t)
        for (int u = 0; u < t; ++u) SDT_CHECK_ARG(g->wt[t] != g->wt[u], "deterministic weight gradient needs distinct weight taps");
    const bool vec4 = (g->Cin % 4 == 0) && (g->Cout % 4 == 0);
    launch_dw<64, 64>(vec4, x, dy, dw, *g, (hipStream_t)stream, (float*)workspace);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

// ---- grouped form: plan (host, once per set of geometries), then launches with fresh tensor pointers
static int dw_group_ok(const sdt_conv_geom* g) {
    if (check_geom(g)) return 0;
    if (g->Cin % 4 || g->Cout % 4 || g->ntaps != g->Tw) return 0;
    for (int t = 0; t < g->ntaps; ++t)
        for (int u = 0; u < t; ++u)
            if (g->wt[t] == g->wt[u]) return 0;
    return 1;
}
extern "C" int64_t sdt_conv_dw_group_plan_bytes(int n) { return n >= 1 && n <= SDT_DW_GROUP_MAX ? (int64_t)n * (int64_t)sizeof(dw_group_item) : -1; }
// plan_out: n items (host memory, sdt_conv_dw_group_plan_bytes(n)); *workspace_bytes: slabs of all layers.  Every geometry must be one the
// deterministic single-layer entry point takes (full 4-channel vectors, each weight tap once).
extern "C" int sdt_conv_dw_group_plan(const sdt_conv_geom* const* geoms, int n, void* plan_out, int64_t* workspace_bytes) {
    SDT_CHECK_ARG(geoms && plan_out && workspace_bytes && n >= 1 && n <= SDT_DW_GROUP_MAX, "1..24 geometries");
    dw_group_item* it = (dw_group_item*)plan_out;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        SDT_CHECK_ARG(geoms[i] && dw_group_ok(geoms[i]), "geometry not supported by the deterministic weight gradient");
        tiles += geoms[i]->ntaps * cdiv(geoms[i]->Cin, 64) * cdiv(geoms[i]->Cout, 64);
    }
    // ~6 workgroups per CU over the whole group (what dw_split aims at for one layer), at least 4 K steps per workgroup
    const int want = std::max(1, (1536 + tiles / 2) / tiles);
    int block = 0;
    int64_t slab = 0, vec = 0;
    for (int i = 0; i < n; ++i) {
        const sdt_conv_geom& g = *geoms[i];
        const int M = g.B * g.Ho * g.Wo;
        const int ns0 = std::max(1, std::min(want, M / (4 * BK)));
        const int rows = cdiv(cdiv(M, ns0), BK) * BK;
        const int nsplit = cdiv(M, rows);
        const int64_t nel = (int64_t)g.Cout * g.Tw * g.Cin;
        it[i].g = g;
        it[i].rows = rows, it[i].nsplit = nsplit, it[i].block0 = block;
        it[i].nblocks = nsplit * g.ntaps * cdiv(g.Cin, 64) * cdiv(g.Cout, 64);
        it[i].slab_off = slab, it[i].n = nel, it[i].vec0 = vec;
        block += it[i].nblocks;
        slab += (int64_t)nsplit * nel;
        vec += nel / 4;
    }
    *workspace_bytes = slab * (int64_t)sizeof(float);
    return SDT_OK;
}
extern "C" int sdt_conv_dw_group_f32(const void* const* x, const void* const* dy, void* const* dw, int n, const void* plan_host,
                                     const void* plan_dev, void* workspace, void* stream) {
    SDT_CHECK_ARG(x && dy && dw && plan_host && plan_dev && workspace && n >= 1 && n <= SDT_DW_GROUP_MAX, "bad argument");
    SDT_CHECK_ARG(g_conv_math == SDT_MATH_F32, "the grouped weight gradient is exact fp32 only");
    const dw_group_item* it = (const dw_group_item*)plan_host;
    dw_group_args A;
    uintptr_t bits = (uintptr_t)workspace;
    for (int i = 0; i < n; ++i) {
        SDT_CHECK_ARG(x[i] && dy[i] && dw[i], "null tensor");
        A.x[i] = (const float*)x[i], A.dy[i] = (const float*)dy[i], A.dw[i] = (float*)dw[i];
        bits |= (uintptr_t)x[i] | (uintptr_t)dy[i] | (uintptr_t)dw[i];
    }
    SDT_CHECK_ARG(bits % 16 == 0, "operands must be 16-byte aligned");
    A.items = (const dw_group_item*)plan_dev;
    A.slabs = (float*)workspace;
    A.n = n;
    const int blocks = it[n - 1].block0 + it[n - 1].nblocks;
    const int64_t total_vec = it[n - 1].vec0 + it[n - 1].n / 4;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_dw_group_kernel, dim3(blocks), dim3(256), 0, s, A);
    hipLaunchKernelGGL(dw_slab_reduce_group_kernel, dim3((unsigned)std::min<int64_t>((total_vec + 255) / 256, 4096)), dim3(256), 0, s, A, total_vec);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_weight_transpose_f32(const float* w, float* wt, int cout, int taps, int cin, void* stream) {
    SDT_CHECK_ARG(w && wt && cout > 0 && taps > 0 && cin > 0, "bad argument");
    dim3 grid(cdiv(cin, 32), cdiv(cout, 32), taps);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, cout, taps, cin);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_weight_transpose_batched_f32(const sdt_wt_desc* table, int n_layers, int total_tiles, void* stream) {
    SDT_CHECK_ARG(table && n_layers > 0 && total_tiles > 0, "bad argument");
    hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n_layers);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_col_sum_f32(const float* x, float* out, int64_t rows, int c, void* stream) {
    SDT_CHECK_ARG(x && out && rows > 0 && c > 0, "bad argument");
    hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)cdiv(c, 16)), dim3(256), 0, (hipStream_t)stream, x, out, rows, c);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
