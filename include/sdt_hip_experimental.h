/*
 * sdt_hip_experimental.h -- entry points of two MEASURED EXPERIMENTS that lose to the default path and are therefore exported by the
 * -DSDT_TUNING library (libsdt_hip_tuning.so) only, not by libsdt_hip.so:
 *   csrc/presplit.hip : the bf16x6 pre-split operand pipeline          (4060 vs 4445 clips/s for the split-in-kernel form)
 *   csrc/conv1d.hip   : the Conv1d stage as one launch per layer       (-2 % end to end)
 * Host side: speechdrivestemplates_amd/experimental/.  Same conventions as sdt_hip.h.
 */
#ifndef SDT_HIP_EXPERIMENTAL_H
#define SDT_HIP_EXPERIMENTAL_H

#include "sdt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * fp32-equivalent conv products on the bf16 MFMA from PRE-SPLIT operands (csrc/presplit.hip): an fp32 value is split exactly
 * into three bf16 pieces by truncation (x = x1 + x2 + x3) and a product is the six MFMA products a1b1, a1b2, a2b1, a2b2, a1b3,
 * a3b1 accumulated in fp32 (dropped terms < 2^-23 |ab|).  A "planes" tensor holds 3*n bf16 for the n elements of a channels-last
 * fp32 tensor (rows, C), C % 32 == 0: piece p of element (row, c) at row*3C + (c/32)*96 + p*32 + c%32 (the three pieces of a
 * 32-channel chunk are adjacent 64-byte runs, so a K step of the conv kernel reads 192 contiguous bytes per row).  Replaces the same ATen convolution calls as sdt_conv_taps_f32
 * (building_blocks.py:15-22) for 2-D layers with Cin % 32 == 0; fp32 storage and accumulation are unchanged.
 *   sdt_split_planes_f32    : x (n floats = rows x C) -> planes (standalone split; the normalisation kernels emit planes themselves)
 *   sdt_weight_planes_batched: planes of W (cout,taps,cin) AND of its (cin,taps,cout) mirror for many layers in one launch;
 *                              tile_begin / total_tiles as in sdt_weight_transpose_batched_f32
 *   sdt_conv_taps_pre_f32   : forward conv (ncls = 1; stats != NULL accumulates the forward statistics as
 *                              sdt_conv_taps_stats_f32) or input gradient (ncls parity classes; nb != NULL accumulates the
 *                              normalisation-backward statistics as sdt_conv_taps_multi_f32) from x planes and w planes.
 */
typedef struct sdt_wp_desc {
    const float* w; /* (cout, taps, cin) fp32 */
    void* wp;       /* [3](cout, taps, cin) bf16 */
    void* wtp;      /* [3](cin, taps, cout) bf16 */
    int32_t cout, taps, cin, tile_begin;
} sdt_wp_desc;
int sdt_split_planes_f32(const float* x, void* planes, int64_t n, int C, void* stream);
int sdt_weight_planes_batched(const sdt_wp_desc* table, int n_layers, int total_tiles, void* stream);
int sdt_conv_taps_pre_f32(const void* x_planes, int64_t x_plane_elems, const void* w_planes, int64_t w_plane_elems, float* y,
                          const sdt_conv_geom* geoms, int ncls, double* stats, int rows_per_group, const sdt_norm_bwd* nb,
                          void* stream);
/* developer switch: force the tile of sdt_conv_taps_pre_f32 (0 = automatic, 64064, 128064, 128128) */
int sdt_set_pre_tile(int tile);

/*
 * The Conv1d stage of the sdt generator as one launch per layer and direction (csrc/conv1d.hip): Conv1d (k3 s1 p1 | k4 s2 p1 | k1,
 * building_blocks.py:8-12,31-38) whose per-(b,t) normalisation over channels + LeakyReLU (building_blocks.py:46,50-51) and the
 * linear x2 upsample + skip add in front of the U-Net decoder convs (generator.py:79-83) are applied while the A operand is
 * staged ("normalise on load"), and whose epilogue emits the row statistics the NEXT launch needs.
 *   Y[b,to,n] = bias[n] + add[b,to,n] + sum_{t<taps, c<Cin} A(b, to*stride + t - pad, c) * W[n,t,c]       W (Cout,taps,Cin)
 * in_mode 0: A = X (B,Ti,Cin).   1: A = act(norm(X)), row statistics of X from xstats.
 *         2: A = linear_upsample(act(norm(X2 (B,T2,Cin))))[ti] + act(norm(X))[ti]  (x2stats / xstats).
 *         3: input gradient of a layer: X = dz w.r.t. the layer's ACTIVATED output (B,Ti,Cin), X2 = its raw output y, xstats =
 *            forward statistics, x2stats = backward statistics (sum g, sum g*yhat); A = the normalisation backward on load;
 *            W = the (Cin_layer,taps,Cout_layer) mirror; output row to gathers layer-output rows (to + pad - t) / stride.
 * Row statistics are PARTIALS: stats[row][np][2], one (sum, sum of squares) -- or (sum g, sum g*yhat) -- per 64-column tile of the
 * launch that produced them (np = C / 64); consumers add the np pairs.  ystats (nullable, Cout % 64 == 0): forward partials of Y,
 * or, when bw_y != NULL, backward partials of the layer below (raw output bw_y, forward partials bw_stats, shape of Y).
 */
typedef struct sdt_c1d {
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
    int32_t B, Ti, T2, Cin, To, Cout, taps, stride, pad;
    int32_t in_mode, np_in, np_in2, np_bw;
    float eps, slope;
    /* split-K inside the launch (1 = none): K slices of a tile store partial tiles into slabs [splitk][B*To][Cout]; the last slice
     * to arrive (counters [tiles], uint32, zero on entry, left zero) adds them in slice order and runs the epilogue. */
    int32_t splitk;
    float* slabs;
    uint32_t* counters;
} sdt_c1d;
int sdt_c1d_layer_f32(const sdt_c1d* p, void* stream);
/* z = act(norm(y)) from partial statistics, also mean / rstd per row (feeds the generic weight-gradient kernels). */
int sdt_c1d_rownorm_partials_f32(const float* y, const float* stats, int np, float* z, float* mean, float* rstd, int64_t rows,
                                 int C, float eps, float slope, void* stream);
/* Adjoint of F.interpolate(prev (B,Ti,C), To, 'linear') applied to g (B,To,C) -> dprev, fused with the backward statistics of
 * the layer that produced prev (raw output y, forward partials stats): bstats[row][np_out][2] = {sum g', sum g'*yhat, 0...}.
 * Ti == To degenerates to a copy + statistics. */
int sdt_c1d_upsample_bwd_stats_f32(const float* g, float* dprev, const float* y, const float* stats, int np, float* bstats,
                                   int np_out, int B, int Ti, int To, int C, float eps, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif
