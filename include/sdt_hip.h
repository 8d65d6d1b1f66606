/*
 * sdt_hip.h -- C ABI of libsdt_hip.so: the MI355X (gfx950) kernels behind the SDT voice2pose
 * training hot path.  Plain pointers + sizes + a hipStream_t (passed as void*); no torch types.
 *
 * All activation tensors are CHANNELS-LAST fp32 in HBM:
 *     2-D stage  (B, H, W, C)      1-D stage (B, T, C)  == (B, 1, T, C)
 * Conv weights are (Cout, taps, Cin) with Cin contiguous (taps = kh*kw, row-major).  The host
 * mirror keeps the reference's logical shapes (Cout,Cin,kh,kw)/(Cout,Cin,k) as strided views of
 * that storage, so reference checkpoints load unchanged.
 *
 * Every entry point: allocates nothing, launches on the caller's stream, returns 0 on success or
 * a negative sdt_status; sdt_last_error() gives the message.  Each comment names the reference
 * operator (file:line in ShenhanQian/SpeechDrivesTemplates) the entry point replaces.
 */
#ifndef SDT_HIP_H
#define SDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDT_MAX_TAPS 20

enum sdt_status { SDT_OK = 0, SDT_ERR_ARG = -1, SDT_ERR_LAUNCH = -2, SDT_ERR_UNSUPPORTED = -3 };
/* Element type of a tensor argument of the *_t / *_bf16 entry points (the bf16-storage path of BASELINE config 4: activations of the
 * Conv2d chain and the conv operands' weight copies live in HBM as bf16, statistics / accumulation / master weights / gradients fp32) */
enum sdt_dtype { SDT_F32 = 0, SDT_BF16 = 1 };

const char* sdt_last_error(void);
int sdt_abi_version(void);

/*
 * Tap-table convolution geometry (implicit GEMM):
 *   Y[b, oy*osy+ooy, ox*osx+oox, n] = bias[n] +
 *       sum_{t<ntaps} sum_{c<Cin} X[b, oy*sy+dy[t], ox*sx+dx[t], c] * W[n, wt[t], c]
 * for (b,oy,ox) in B x Ho x Wo; out-of-range X reads are 0.  X is (B,Hi,Wi,Cin), Y is
 * (B,Hy,Wy,Cout), W is (Cout,Tw,Cin).  One geometry describes a forward conv (any stride), and --
 * with transposed weights and flipped taps -- the input-gradient of a stride-1 conv or one output
 * parity class of a strided conv.
 */
typedef struct sdt_conv_geom {
    int32_t B, Hi, Wi, Cin;
    int32_t Ho, Wo;
    int32_t Hy, Wy, Cout;
    int32_t sy, sx;
    int32_t osy, osx, ooy, oox;
    int32_t ntaps, Tw;
    int32_t dy[SDT_MAX_TAPS], dx[SDT_MAX_TAPS], wt[SDT_MAX_TAPS];
} sdt_conv_geom;

/* nn.Conv2d / nn.Conv1d forward and input-gradient (building_blocks.py:15-22,31-38; ATen conv). */
int sdt_conv_taps_f32(const float* x, const float* w, const float* bias, float* y,
                      const sdt_conv_geom* g, void* stream);
/* The same forward conv with the statistics pass of the normalisation that follows fused into its epilogue:
 * stats[(g * Cout + n) * 2 + {0,1}] += sum / sum of squares of Y[., n] over the output rows m (= (b*Ho+oy)*Wo+ox) with
 * m / rows_per_group == g (InstanceNorm2d: rows_per_group = Ho*Wo; BatchNorm: B*Ho*Wo).  stats must be zero on entry.
 * sdt_conv_taps_stats_supported: 1 if the geometry qualifies (dense output, Cin % 32 == 0, no split-K, fp32 math). */
int sdt_conv_taps_stats_supported(const sdt_conv_geom* g, int rows_per_group);
int sdt_conv_taps_stats_f32(const float* x, const float* w, const float* bias, float* y, const sdt_conv_geom* g,
                            double* stats, int rows_per_group, void* stream);
/* Split-K form for launches with too few output tiles to fill the chip (the 1-D stage): slice z of `splitk`
 * writes its partial sums to partial + z*numel(Y) (no bias); after ALL launches that share the Y tensor (the parity
 * classes of an input-gradient) sdt_splitk_reduce_f32 sums the slabs in a fixed order (deterministic) and adds bias.
 * splitk == 1 is sdt_conv_taps_f32.  sdt_conv_taps_splitk_hint returns a suitable slice count for a geometry. */
int sdt_conv_taps_splitk_hint(const sdt_conv_geom* g);
int sdt_conv_taps_splitk_f32(const float* x, const float* w, const float* bias, float* y,
                             const sdt_conv_geom* g, int splitk, float* partial, void* stream);
int sdt_splitk_reduce_f32(const float* partial, const float* bias, float* y, int64_t n, int cout, int splitk, void* stream);
/* Input gradient of a strided convolution in ONE launch: `ncls` (1..4) output parity classes -- geometries that share X (= dY),
 * W (= transposed weights) and the Y (= dX) tensor and write disjoint output positions (ATen's conv backward-data,
 * building_blocks.py:15-22,31-38).  Same split-K contract as sdt_conv_taps_splitk_f32.
 * nb != NULL (fp32 math, splitk == 1, Cin % 32 == 0): the epilogue also accumulates the per-(group, channel) sums the backward
 * of the InstanceNorm2d / BatchNorm + LeakyReLU that produced this conv's input needs (building_blocks.py:24-26,46), so that
 * sdt_colnorm_bwd_f32(stats_ready = 1) skips its statistics pass over dz and y:
 *   sums[(grp*C + n)*2 + 0] += sum gg,  [..+1] += sum gg*yhat,  gg = dX * act'(gamma*yhat + beta), yhat = (y - mean)*rstd
 * y = raw output of the conv below (same shape as dX), grp = batch item (groups == B) or 0 (groups == 1); sums zero on entry. */
typedef struct sdt_norm_bwd {
    const void* y;      /* element type of the launch's x (fp32; bf16 for sdt_convsk_bf16) */
    const float* mean;  /* [groups*C] */
    const float* rstd;  /* [groups*C] */
    const float* gamma; /* [C] or NULL */
    const float* beta;  /* [C] or NULL */
    double* sums;       /* [groups*C*2] */
    float slope;
    int32_t groups;
} sdt_norm_bwd;
int sdt_conv_taps_multi_f32(const float* x, const float* w, float* y, const sdt_conv_geom* geoms, int ncls, int splitk,
                            float* partial, const sdt_norm_bwd* nb, void* stream);
/*
 * Persistent stream-K form of the same convolution (csrc/convsk.hip; round 3) for the MFMA-bound Conv2d launches of the audio
 * encoder (generator.py:15-30 through building_blocks.py:15-22): forward (ncls = 1) or input gradient (ncls parity classes) with
 * Cin % 32 == 0 and Cout % 64 == 0.  One or two resident workgroups per CU each walk a contiguous range of the launch's (tile, live K step) list, so every
 * CU gets the same number of K steps whatever the tile count; tiles that straddle two ranges are combined in a fixed order
 * (bit-identical from run to run).  128x128x32 or 128x64x32 tiles, software-pipelined, fp32 accumulation in chunks of 256 products.
 *   sdt_convsk_supported     1 if the geometry pack qualifies
 *   sdt_convsk_plan_bytes    size of the PLAN of a geometry pack: per GEMM row {X byte offset, mask of the taps outside X, Y byte offset,
 *                            statistics group} (the row ORDER is the plan's choice: image-row-major where that culls >= 5 % of the K
 *                            steps), per m-tile {live-tap mask, tap rotation}, prefix sums of live K steps, first tile of
 *                            each workgroup's range.  Built on the host once per geometry (sdt_convsk_plan_build), kept by the caller
 *                            in host AND device memory and handed to every launch
 *   rows_per_group > 0       statistics group of output row m = m / rows_per_group (forward statistics, as sdt_conv_taps_stats_f32);
 *   rows_per_group <= 0      group = batch item (bwd_groups == B) or 0 (bwd_groups == 1) (backward statistics, as sdt_norm_bwd)
 *   sdt_convsk_workspace_bytes  partial-tile slabs + flags: one buffer per stream, ZERO-FILLED ONCE by the caller, then passed to every
 *                            launch on that stream.  epoch >= 1 is the flag value of the launch; the workgroup that consumes a flag lowers it
 *                            again, so every flag is zero between launches and the same epoch may be passed every time (a launch recorded
 *                            into a hipGraph replays correctly).  The word after the flags is an error code: non-zero = the owner of a
 *                            split tile gave up waiting for a partner (sdt_convsk_set_spin_limit polls: the partner was never dispatched,
 *                            e.g. another process holds its slot); that tile is stored as NaN, never with a partial sum missing, and the
 *                            host mirror raises when it sees the word (core/pipelines/trainer.py)
 *   sdt_convsk_f32           the launch; stats / nb as in sdt_conv_taps_stats_f32 / sdt_conv_taps_multi_f32 (at most one of them);
 *                            xbytes / wbytes / ybytes: sizes of the X, W and Y tensors
 */
/*   sdt_convsk_f32_w3        the same launch for a split-fp32 plan (sdt_convsk_set_f32_split) with the weights ALREADY split: w3 = three bf16 planes
 *                            [hi | mid | lo] of the (N, Tw, Cin) weight tensor (sdt_wt_desc.planes = 3 makes them); wbytes = the fp32 tensor's size.
 *                            Bit-identical results to sdt_convsk_f32 on the fp32 weights (the same split, made once per optimiser step instead of in
 *                            every tile on every K step).  ABI 5. */
int sdt_convsk_supported(const sdt_conv_geom* geoms, int ncls);
int sdt_convsk_grid(void);
int sdt_convsk_set_wg_per_cu(int n); /* 1 or 2 persistent workgroups per CU for plans built afterwards (default 2) */
/* fp32 plans built afterwards with ONE workgroup per CU: the split-fp32 form of the 8-wave kernel (csrc/convbf.hip: each fp32 operand = three bf16
 * planes made by the loader, six bf16 MFMAs per fragment pair -- products exact to 2^-23, fp32 accumulation).  Default 0. */
int sdt_convsk_set_f32_split(int on);
/* Workgroup slots (of the GPU's 512 two-per-CU slots; multiple of 8, <= 256 = half of the GPU) that plans built afterwards leave free: a persistent
 * launch that fills the GPU cannot share it with another long-lived kernel (a collective's); data-parallel runs plan their backward launches with a
 * reserve.  A one-per-CU workgroup (the 8-wave kernels) counts as two slots.  Default 0. */
int sdt_convsk_set_reserved_slots(int n);
/* K order of the 8-wave kernels' tiles for launches made afterwards (a launch-time setting, not a plan property): 0 = tap-major (all 128-byte channel
 * chunks of a tap, then the next tap), 1 = chunk-major (all live taps of a chunk, then the next chunk: neighbouring taps re-read the cache lines of
 * the step before while the CU's vector L1 still holds them).  ABI 5. */
int sdt_convsk_set_k_order(int order);
/* Split-fp32 weight-gradient plans built afterwards (sdt_convsk_dw_plan_build_t with sdt_convsk_set_f32_split(1)): 1 = 128-wide column tiles with a
 * ragged last one when taps * Cin = 64 (mod 128); 0 (default) = tiles by divisibility, the rule of rounds 3-5.  ABI 5. */
int sdt_convsk_set_dw_wide_tiles(int on);
int sdt_convsk_f32_w3(const float* x, const void* w3, const float* bias, float* y, const void* plan_host, const void* plan_dev, void* workspace,
                      unsigned epoch, double* stats, const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes, int64_t ybytes, void* stream);
/* Polls (each ~1 us under load) of a partner's flag before the owner of a split tile declares the launch failed.  Default 1 << 22. */
int sdt_convsk_set_spin_limit(unsigned polls);
unsigned sdt_convsk_get_spin_limit(void);
int64_t sdt_convsk_plan_bytes(const sdt_conv_geom* geoms, int ncls);
int64_t sdt_convsk_workspace_bytes(void);
int sdt_convsk_plan_build(const sdt_conv_geom* geoms, int ncls, int rows_per_group, int bwd_groups, void* out, int64_t out_bytes);
/* The same for a given element type of x / w (x_dtype) and of y (y_dtype): the plan's byte offsets and its K step (128 bytes of an input
 * row: 32 fp32 or 64 bf16 channels) depend on them.  bf16 needs Cin % 64 == 0. */
int sdt_convsk_supported_t(const sdt_conv_geom* geoms, int ncls, int x_dtype);
int64_t sdt_convsk_plan_bytes_t(const sdt_conv_geom* geoms, int ncls, int x_dtype);
int sdt_convsk_plan_build_t(const sdt_conv_geom* geoms, int ncls, int rows_per_group, int bwd_groups, int x_dtype, int y_dtype, void* out,
                            int64_t out_bytes);
/* Weight gradient of a forward geometry on the same persistent machinery (dense dY, Cout % 64 == 0, Cin % 64 == 0; 128- or 64-wide tiles):
 * the reduction over the output positions is split over the workgroups, partial tiles go to slabs of `workspace`
 * (sdt_convsk_dw_workspace_bytes() bytes, contents irrelevant) and a second kernel adds them to dw (Cout, Tw, Cin) in a fixed order:
 * no atomics, bit-identical from run to run (what the reference asks of cuDNN with cudnn.deterministic = True, main.py:37-38).
 * ACCUMULATES into dw like sdt_conv_dw_f32.  Plan: sdt_convsk_dw_plan_bytes / _build, host + device copy as above. */
int sdt_convsk_dw_supported(const sdt_conv_geom* g);
int64_t sdt_convsk_dw_plan_bytes(const sdt_conv_geom* g);
int64_t sdt_convsk_dw_workspace_bytes(void);
int sdt_convsk_dw_plan_build(const sdt_conv_geom* g, void* out, int64_t out_bytes);
int sdt_convsk_dw_f32(const float* x, const float* dy, float* dw, const void* plan_host, const void* plan_dev, void* workspace,
                      int64_t xbytes, int64_t ybytes, void* stream);
int sdt_convsk_f32(const float* x, const float* w, const float* bias, float* y, const void* plan_host, const void* plan_dev,
                   void* workspace, unsigned epoch, double* stats, const sdt_norm_bwd* nb, int64_t xbytes, int64_t wbytes, int64_t ybytes,
                   void* stream);
/* bf16-storage path: x, w, y and nb->y are bf16 tensors (plan from sdt_convsk_plan_build_t(.., SDT_BF16, SDT_BF16, ..)); products on
 * v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), fp32 accumulation, bias and statistics (taken from the fp32 accumulators before the
 * output is rounded).  sdt_convsk_dw_bf16: bf16 x / dy, fp32 gradient accumulated into dw; 64 output positions per K step, operands
 * transposed on the way out of LDS by ds_read_b64_tr_b16. */
int sdt_convsk_bf16(const void* x, const void* w, const float* bias, void* y, const void* plan_host, const void* plan_dev,
                    void* workspace, unsigned epoch, double* stats, const sdt_norm_bwd* nb, int64_t xbytes, int64_t wbytes, int64_t ybytes,
                    void* stream);
int sdt_convsk_dw_supported_t(const sdt_conv_geom* g, int dtype);
int64_t sdt_convsk_dw_plan_bytes_t(const sdt_conv_geom* g, int dtype);
int sdt_convsk_dw_plan_build_t(const sdt_conv_geom* g, int dtype, void* out, int64_t out_bytes);
int sdt_convsk_dw_bf16(const void* x, const void* dy, float* dw, const void* plan_host, const void* plan_dev, void* workspace,
                       int64_t xbytes, int64_t ybytes, void* stream);
/* Weight gradient, ACCUMULATED into dw (Cout,Tw,Cin):
 *   dw[n, wt[t], c] += sum_{b,oy,ox} dY[b, oy*osy+ooy, ox*osx+oox, n] * X[b, oy*sy+dy[t], ox*sx+dx[t], c] */
int sdt_conv_dw_f32(const float* x, const float* dy, float* dw, const sdt_conv_geom* g, void* stream);
/* The same weight gradient without atomics: every row range of the split reduction stores its partial product into its own slab of
 * `workspace` (>= sdt_conv_dw_workspace_bytes(g) bytes, 16-byte aligned) and the slabs are added to dw in a fixed order, so the
 * result is bit-identical from run to run (torch.use_deterministic_algorithms territory; the reference's cuDNN weight gradient
 * is not deterministic either, core/pipelines/trainer.py never asks for it).  fp32 MFMA only. */
int64_t sdt_conv_dw_workspace_bytes(const sdt_conv_geom* g);
int sdt_conv_dw_det_f32(const float* x, const float* dy, float* dw, const sdt_conv_geom* g, void* workspace, int64_t workspace_bytes,
                        void* stream);
/* Which kernel instantiation the two entry points above pick for a geometry (16-B aligned operands assumed):
 * (BM*1000+BN)*10 + vec4, e.g. 1281281 = conv_taps_kernel<128,128,true>.  Used by bench.py to attribute timings. */
int sdt_conv_taps_variant(const sdt_conv_geom* g);
/* 1 when sdt_conv_taps_f32 / _splitk_f32 / _multi_f32 on these classes with this K split run the 1-D stage's small-K kernel
 * (conv1d_small_kernel: Hi = 1, Cin % 32 == 0, Cout % 64 == 0, <= 8 K steps per workgroup, exact fp32, no statistics epilogue) */
int sdt_conv1d_small_used(const sdt_conv_geom* geoms, int ncls, int splitk);
int sdt_conv_dw_variant(const sdt_conv_geom* g);
/* (Cout,T,Cin) -> (Cin,T,Cout): operand layout for the input-gradient GEMM. */
int sdt_weight_transpose_f32(const float* w, float* wt, int cout, int taps, int cin, void* stream);
/* Multiplication arithmetic of sdt_conv_taps(_splitk)_f32 for geometries with Cin % 32 == 0 (process-wide switch; tensors
 * stay fp32 in HBM and accumulation stays fp32 in every mode):
 *   SDT_MATH_F32    exact fp32 products on v_mfma_f32_32x32x2_f32 (default; what every parity claim refers to)
 *   SDT_MATH_BF16   operands rounded to bf16 (v_mfma_f32_32x32x16_bf16) -- BASELINE config 4's precision
 *   SDT_MATH_BF16X3 2-piece split, 3 bf16 products per fp32 product (~16 significant bits)
 *   SDT_MATH_BF16X6 exact 3-piece split of the 24-bit significand, 6 bf16 products (dropped terms < 2^-23 |ab|) */
enum { SDT_MATH_F32 = 0, SDT_MATH_BF16 = 1, SDT_MATH_BF16X3 = 3, SDT_MATH_BF16X6 = 6 };
int sdt_set_conv_math(int mode);
int sdt_get_conv_math(void);


/* Deterministic weight gradients of up to 24 SMALL layers in one grid + one ordered reduce (the generator's sixteen Conv1d blocks: launched one by
 * one each layer splits its rows ~32 ways to fill the chip and spends its time on slab traffic; together they need 2 row ranges each).
 * plan: host-built once per set of geometries (sdt_conv_dw_group_plan_bytes(n) bytes; the caller keeps a host and a device copy);
 * *workspace_bytes: slab space, no initialisation needed.  Accumulates into every dw[i] like sdt_conv_dw_det_f32 (fixed order, exact fp32).  */
int64_t sdt_conv_dw_group_plan_bytes(int n);
int sdt_conv_dw_group_plan(const sdt_conv_geom* const* geoms, int n, void* plan_out, int64_t* workspace_bytes);
int sdt_conv_dw_group_f32(const void* const* x, const void* const* dy, void* const* dw, int n, const void* plan_host, const void* plan_dev,
                          void* workspace, void* stream);
/* The same transposition for many layers in ONE launch (all mirrors of an optimiser group are refreshed right after its
 * Adam step).  table: device array of n_layers descriptors; tile_begin = running sum of
 * ceil(cin/32)*ceil(cout/32)*taps over the preceding layers, total_tiles = that sum over all layers. */
typedef struct sdt_wt_desc {
    const float* w; /* (cout, taps, cin) */
    float* wt;      /* (cin, taps, cout); nullable */
    void* w16;      /* nullable: bf16 copy of w (round to nearest even), the bf16-storage path's forward / weight operand */
    void* wt16;     /* nullable: bf16 copy of the mirror */
    int32_t cout, taps, cin, tile_begin;
    int32_t planes; /* what w16 / wt16 receive: 1 (or 0) = one bf16 copy; 3 = the exact three-way bf16 split [hi | mid | lo], three planes of
                       cout * taps * cin elements each: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) -- the pre-split B operand of
                       sdt_convsk_f32_w3 (ABI 5) */
    int32_t reserved;
} sdt_wt_desc;
int sdt_weight_transpose_batched_f32(const sdt_wt_desc* table, int n_layers, int total_tiles, void* stream);

/* out[c] += sum_rows x[row, c]  (bias gradient of the k1 head conv, generator.py:103). */
int sdt_col_sum_f32(const float* x, float* out, int64_t rows, int c, void* stream);

/*
 * Column-statistics normalisation over a (G, R, C) view: statistics per (g, c) over R rows.
 *   G=B, R=H*W  -> nn.InstanceNorm2d            (building_blocks.py:26)
 *   G=1, R=B*HW -> nn.BatchNorm{1,2}d, training (building_blocks.py:24,39)
 * followed by LeakyReLU(slope) (slope = 0 -> ReLU)  (building_blocks.py:46).
 * sums: workspace of 2*G*C doubles that MUST BE ZERO ON ENTRY (fp64 atomics accumulate into it; the call leaves it
 * dirty -- callers carve it from a region zeroed once per step instead of paying a memset per layer).
 * num_batches_tracked (nullable) is the
 * BatchNorm int64 counter, incremented on the device.  gamma/beta/running_* may be NULL (IN).
 * stats_ready != 0: sums already holds sum(y), sum(y^2) per (g, c) (sdt_conv_taps_stats_f32) -- the statistics pass is skipped.
 * *_t: the same with explicit element types (enum sdt_dtype) of the activation tensors -- the bf16-storage path: statistics, mean / rstd
 * and the arithmetic stay fp32, only what moves through HBM is bf16.  Built combinations: all fp32; forward y bf16 -> z bf16 | fp32;
 * backward y and dy bf16 with dz bf16 | fp32.
 * fwd writes z, mean[G*C], rstd[G*C]; if running_mean != NULL updates running stats with
 * momentum (unbiased variance), as nn.BatchNorm does in training mode -- with G > 1 as G consecutive calls of the module on the G
 * slices would: G updates in slice order, num_batches_tracked += G (the two no-grad pose-encoder passes of a train step in one launch).
 * eval: z = act(gamma*(y-running_mean)/sqrt(running_var+eps)+beta).
 */
int sdt_colnorm_fwd_f32(const float* y, float* z, double* sums, float* mean, float* rstd,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, int G, int64_t R, int C, float eps, float momentum,
                        float slope, int stats_ready, void* stream);
int sdt_colnorm_fwd_t(const void* y, int y_dtype, void* z, int z_dtype, double* sums, float* mean, float* rstd,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, int G, int64_t R, int C, float eps, float momentum,
                      float slope, int stats_ready, void* stream);
int sdt_colnorm_eval_f32(const float* y, float* z, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var,
                         int64_t rows, int C, float eps, float slope, void* stream);
/* bwd: dy <- d(loss)/dy given dz; dgamma/dbeta (nullable) are ACCUMULATED. dy may alias dz.
 * stats_ready != 0: sums already holds sum gg, sum gg*yhat per (g, c) (accumulated by the epilogue of the input-gradient
 * conv that produced dz, sdt_conv_taps_multi_f32 with nb) -- the statistics pass over dz and y is skipped. */
int sdt_colnorm_bwd_f32(const float* dz, const float* y, float* dy, double* sums, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, float* dgamma,
                        float* dbeta, int G, int64_t R, int C, float slope, int stats_ready, void* stream);
int sdt_colnorm_bwd_t(const void* dz, int dz_dtype, const void* y, int y_dtype, void* dy, int dy_dtype, double* sums, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, float* dgamma,
                      float* dbeta, int G, int64_t R, int C, float slope, int stats_ready, void* stream);

/*
 * First audio-encoder block fused for Cin == 1: Conv2d(1,64,k3,s1,p1,bias=False) -> InstanceNorm2d (groups = B) or
 * training-mode BatchNorm2d (groups = 1) -> LeakyReLU(slope)   (generator.py:16, building_blocks.py:15-26,46).
 * The statistics of all 64 channels are derived from 9+45 fp64 moments of the mel image, so z (B,H,W,64) is written once
 * and the raw conv output is never stored; backward recomputes the normalised pre-activation from mel.
 *   (both workspaces MUST BE ZERO ON ENTRY and are left dirty, like sdt_colnorm_*'s)
 *   fwd: mom = workspace of 54*B doubles; writes z, mean[groups*64], rstd[groups*64] (+ BN running stats / counter).
 *   bwd: mom = the moments fwd left in its workspace (read-only); sums = workspace of 11*groups*64 doubles;
 *        dw (64,9) and dgamma/dbeta (nullable) are ACCUMULATED.  One pass over dz: the weight gradient is assembled
 *        from 11 sums per (group, channel) and the mel moments.
 */
int sdt_l0_block_fwd_f32(const float* mel, const float* w, float* z, double* mom, float* mean, float* rstd,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, int B, int H, int W, int groups, float eps, float momentum,
                         float slope, void* stream);
int sdt_l0_block_bwd_f32(const float* dz, const float* mel, const float* w, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, const double* mom, double* sums, float* dw, float* dgamma,
                         float* dbeta, int B, int H, int W, int groups, float slope, void* stream);
/* The same with z written / dz read as z_dtype / dz_dtype (enum sdt_dtype): the block's output is the first bf16 tensor of the bf16-storage
 * path, its backward reads the bf16 gradient the L1 input-gradient launch wrote.  mel, weights, moments, statistics: fp32 / fp64 as above. */
int sdt_l0_block_fwd_t(const float* mel, const float* w, void* z, int z_dtype, double* mom, float* mean, float* rstd,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, int B, int H, int W, int groups, float eps, float momentum,
                       float slope, void* stream);
int sdt_l0_block_bwd_t(const void* dz, int dz_dtype, const float* mel, const float* w, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, const double* mom, double* sums, float* dw, float* dgamma,
                       float* dbeta, int B, int H, int W, int groups, float slope, void* stream);

/*
 * Row normalisation over C for each of `rows` rows + LeakyReLU: the reference's InstanceNorm1d
 * applied to the (B,T,C)-permuted tensor (building_blocks.py:50-51) == per-(b,t) LayerNorm, no affine.
 */
int sdt_rownorm_fwd_f32(const float* y, float* z, float* mean, float* rstd, int64_t rows, int C,
                        float eps, float slope, void* stream);
int sdt_rownorm_bwd_f32(const float* dz, const float* y, const float* mean, const float* rstd, float* dy,
                        int64_t rows, int C, float slope, void* stream);
/* rownorm_fwd fused with the split-K reduction of the producing conv: partial = nslab slabs of (rows, C) written by
 * sdt_conv_taps_splitk_f32; y <- their sum in slab order (what sdt_splitk_reduce_f32 computes), z/mean/rstd as above. */
int sdt_rownorm_slabs_fwd_f32(const float* partial, int nslab, float* y, float* z, float* mean, float* rstd,
                              int64_t rows, int C, float eps, float slope, void* stream);

/*
 * F.interpolate(x,(1,T),'bilinear') on the (B,H,W,C) encoder output, squeezed, with the gathered
 * clip code broadcast-concatenated along channels (generator.py:41-42,110-111; voice2pose.py:94):
 *   out (B,T,C+D);  out[b,t,C+d] = table[idx[b], d]   (D may be 0, table/idx NULL).
 */
int sdt_resize_concat_fwd_f32(const float* x, const float* table, const int64_t* idx, float* out,
                              int B, int H, int W, int C, int T, int D, void* stream);
/* dx (B,H,W,C) is fully written; dtable rows are ACCUMULATED (dense gradient of the code table). */
int sdt_resize_concat_bwd_f32(const float* dout, const int64_t* idx, float* dx, float* dtable,
                              int B, int H, int W, int C, int T, int D, void* stream);

/* F.interpolate(prev, To, 'linear') (+ skip) on (B,Ti,C)->(B,To,C) (generator.py:79-83, autoencoder.py:62-66). */
int sdt_upsample_add_fwd_f32(const float* prev, const float* skip, float* out, int B, int Ti, int To, int C, void* stream);
int sdt_upsample_add_bwd_f32(const float* dout, float* dprev, int B, int Ti, int To, int C, void* stream);

/* ---- The generator's Conv1d stage as one persistent launch per direction (csrc/chain1d.hip) ------------------------------------------
 * A CHAIN of ConvNormRelu('1d', norm='IN') blocks (building_blocks.py:31-51: Conv1d without bias -> normalisation of every (clip, frame) over
 * the channels -> LeakyReLU) with the wiring of UNet_1D + the decoder stack (generator.py:53-85,96-103): a block's input is the external
 * tensor (block 0), the activated output of an earlier block, or F.interpolate(earlier block, Ti, 'linear') + the activated output of another
 * one (generator.py:79-83).  Every block has 256 output channels; <= 64 frames per clip; Cin a multiple of 32 (<= 320, block 0 only: the
 * others read 256-channel block outputs).  Only the RAW conv outputs y travel between blocks; normalisation and activation are applied by
 * the consumer on load.  8 workgroups own a clip (see the file header); a launch holds 8 * (CUs / 64) clips (32 on MI355X) with every cluster
 * co-resident, a larger batch runs as consecutive launches.                                                                                  */
enum { SDT_CHAIN_PLAIN = 0, SDT_CHAIN_NORM = 1, SDT_CHAIN_UPADD = 2 };
typedef struct sdt_chain1d_layer {
    int32_t Ti, To, Cin, k, stride, pad;
    int32_t in_mode;        /* SDT_CHAIN_PLAIN: x0 | _NORM: act(norm(y[src_a])) | _UPADD: upsample(act(norm(y[src_a])), Ti) + act(norm(y[src_b])) */
    int32_t src_a, src_b;   /* indices of earlier blocks (-1: unused) */
    int32_t reserved;
    const float* w;         /* (256, k, Cin) weights */
    const float* wt;        /* (Cin, k, 256) mirror (backward only) */
    float* y;               /* (B, To, 256) raw conv output: written by forward, read by backward */
    float* x;               /* (B, Ti, 256) the conv's input as consumed, for the weight-gradient launch (NULL: not wanted; unused for block 0) */
    float* dy;              /* (B, To, 256) backward: gradient of y (the weight gradient's other operand) */
    float* dx;              /* (B, Ti, Cin) backward: gradient of the conv's input */
} sdt_chain1d_layer;
/* 1 when every block maps to a built K loop and the current device can hold a window of clusters, else 0. */
int sdt_chain1d_supported(const sdt_chain1d_layer* layers, int nlayers, int B);
/* zout (B, To_last, 256) = act(norm(y[nlayers-1])).  counters: >= min(B, 8 * (CUs / 64)) zero-initialised uint32 (zero again when a launch ends); err: one uint32
 * that a launch sets non-zero when a workgroup gave up waiting for its cluster (sdt_convsk_set_spin_limit) -- results are then invalid. */
int sdt_chain1d_fwd_f32(const sdt_chain1d_layer* layers, int nlayers, const float* x0, float* zout, int B, float slope, float eps,
                        int math, void* counters, void* err, void* stream);
/* math: SDT_MATH_F32 (exact fp32 products, the default everywhere) or SDT_MATH_BF16 (products of bf16-rounded operands on the bf16 MFMA,
 * fp32 tensors and accumulation: what the bf16-storage step -- BASELINE config 4 -- runs; the global sdt_set_conv_math is NOT consulted).
 * gz (B, To_last, 256): gradient of zout.  Writes dy of every block and dx of every block (block 0 only when need_dx0). */
int sdt_chain1d_bwd_f32(const sdt_chain1d_layer* layers, int nlayers, const float* gz, int B, float slope, float eps, int need_dx0,
                        int math, void* counters, void* err, void* stream);

/* nn.L1Loss(reduction='none')(pred,gt)*lambda .mean() (voice2pose.py:141-142). partial: >=256 doubles. */
int sdt_l1_loss_fwd_f32(const float* pred, const float* gt, int64_t n, float lambda, double* partial, float* loss, void* stream);
int sdt_l1_loss_bwd_f32(const float* pred, const float* gt, const float* gout, int64_t n, float lambda, float* dpred, void* stream);
/* LSGAN terms (voice2pose.py:171-189, nn.MSELoss against a constant): loss = lambda * mean((scores - target)^2);
 * dscores = gout * 2 * lambda / n * (scores - target). */
int sdt_mse_const_fwd_f32(const float* scores, int64_t n, float target, float lambda, float* loss, void* stream);
int sdt_mse_const_bwd_f32(const float* scores, const float* gout, int64_t n, float target, float lambda, float* dscores, void* stream);

/*
 * Clip-code batch KL (voice2pose.py:147-157): code = table[idx] (B,D); mu/unbiased var over the batch;
 * loss = 0.5*mean(-log v + mu^2 + v - 1)*lambda if every v != 0 else 0; valid[0] = that predicate
 * (kept on the device: no host sync).  code_out (B,D) receives the gathered rows.
 * N = rows of the table: an index outside [0, N) reads nothing (the reference raises IndexError there) -- its code row
 * becomes NaN, which poisons the losses, and its gradient row is dropped.
 */
int sdt_code_kl_fwd_f32(const float* table, const int64_t* idx, int N, int B, int D, float lambda,
                        float* code_out, float* loss, int32_t* valid, void* stream);
int sdt_code_kl_bwd_f32(const float* code, const int32_t* valid, const float* gout, const int64_t* idx,
                        int N, int B, int D, float lambda, float* dtable, void* stream);

/*
 * GestureDataset.get_final_results x2 + Voice2Pose.evaluate_step (gesture_dataset.py:193-220,
 * voice2pose.py:412-430), float64 like the reference: final_* (B,T,2,K) f64 (nullable),
 * metrics[0]=L2_dist, metrics[1]=lip_sync_error_n.  work: >= 3*B*T + 4 doubles, contents irrelevant (per-frame
 * partial sums, combined in a fixed order: no atomics, bit-identical from run to run).
 */
int sdt_final_metrics_f64(const float* pred, const float* gt, const double* mean, const double* std,
                          const double* scale, int hierarchical, int B, int T, int K,
                          double* final_pred, double* final_gt, double* work, double* metrics, void* stream);

/* torch.optim.Adam(betas, eps, weight_decay) step over a flat fp32 buffer (voice2pose.py:249-279,302-304).
 * lr_dev: device float (so that LR schedules do not invalidate a captured hipGraph);
 * state_dev: 16 device bytes {int64 step; float bc1; float bc2_sqrt}, zero-initialised by the caller;
 * the call increments `step` on the device and derives the bias corrections from it.
 * grad_scale multiplies g on load (1/world_size after a summing all-reduce: DDP's gradient averaging). */
int sdt_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1,
                      float beta2, float eps, float weight_decay, float grad_scale, void* state_dev, void* stream);

/*
 * Mel front end (torchaudio 0.7 MelSpectrogram as configured at voice2pose.py:27-30):
 * reflect-pad 256, 512-sample frames every 160, periodic-Hann(400) centred, |rFFT|^2, HTK filterbank.
 * The STFT is run as an MFMA GEMM by sdt_conv_taps_f32 over the hop matrix:
 *   sdt_stft_frames_f32 : audio (B,L) -> hops (B, nhops, 160), hops[b, j] = reflect_pad(audio)[j + 56]
 *   sdt_conv_taps_f32   : X = hops as (B,1,nhops,160), W = windowed DFT basis (514, 3, 160)
 *                         (row 2f = w*cos, 2f+1 = -w*sin of bin f; taps >= 400 samples are zero) -> spec (B,F,514)
 *   sdt_mel_fb_f32      : spec (B,F,2*nfreq) interleaved re/im -> mel (B, nmel, F) = fb^T |spec|^2 (sparse rows only)
 */
int sdt_stft_frames_f32(const float* audio, float* hops, int B, int L, int nhops, void* stream);
/* bin_lo/bin_hi (nmel ints each, device): [lo,hi) range of non-zero filterbank rows of every mel filter (triangles). */
int sdt_mel_fb_f32(const float* spec, const float* fb, const int32_t* bin_lo, const int32_t* bin_hi, float* mel, int B, int F,
                   int nfreq, int nmel, void* stream);

/* dst[idx[b], :] += src[b, :]  -- dense gradient of the clip-code table for `clips_code[clip_indices]` (voice2pose.py:94).
 * dst has N rows; rows with idx outside [0, N) are skipped. */
int sdt_rows_scatter_add_f32(const float* src, const int64_t* idx, float* dst, int N, int B, int D, void* stream);

/* y[i] = a[i+stride]-a[i] helper for the motion discriminator input (voice2pose.py:187-188): (B,T,C)->(B,T-1,C) */
int sdt_time_diff_fwd_f32(const float* x, float* y, int B, int T, int C, void* stream);
int sdt_time_diff_bwd_f32(const float* dy, float* dx, int B, int T, int C, void* stream);

/* Batch assembly from a clip store resident in HBM -- the device-side form of GestureDataset.__getitem__ + collate
 * (core/datasets/gesture_dataset.py:85-119): for every b, clip = idx[b]:
 *   raw (N, Tstore, 3, 137) OpenPose x / y / confidence  -> first T frames, 137 -> 122 -> 121 keypoints (:124-145),
 *   relative to the root joint, optionally hierarchical ("parted": head / hand offsets, :157-165), then
 *   (x - mean) / std with fp32 statistics (242,) (:167-176);  poses (B,T,2,121), score (B,T,2,121) = confidence twice.
 * Same fp32 operation order as the reference's torch code -> bit-identical results.
 *   sdt_rows_gather_f32: dst[b, :] = src[idx[b], :]   (the cropped / zero-padded audio rows, n_cols % 4 == 0 not required) */
int sdt_clip_poses_prepare_f32(const float* raw, const int64_t* idx, const float* mean, const float* std, float* poses,
                               float* score, int N, int Tstore, int B, int T, int hierarchical, void* stream);
int sdt_rows_gather_f32(const float* src, const int64_t* idx, float* dst, int N, int B, int64_t n_cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDT_HIP_H */
