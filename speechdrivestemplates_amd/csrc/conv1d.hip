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
#include "../../include/sdt_hip_experimental.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

#define C1_BM 32
#define C1_BN 64
#define C1_BK 64
#define C1_PITCH 68      // floats per LDS row of a [row][64] tile (16-byte aligned rows, 4-float skew against bank conflicts)
#define C1_NPART_MAX 8   // partial statistics per row: Cout / 64 <= 8
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
    unsigned cin_magic;  // floor(2^32 / Cin) + 1: k / Cin == umulhi(k, cin_magic) for the k the kernel forms
    int splitk;          // K slices per output tile (1 = none)
    float* slabs;        // [splitk][M][Cout] partial products (splitk > 1)
    unsigned* counters;  // [tiles] arrival counters, zero on entry, left zero (splitk > 1)
#ifdef SDT_TUNING
    int dbg_mode;             // 1: multipliers skip the MFMAs, 2: loaders skip the transform + LDS stores, 4: loaders skip the global loads
    unsigned long long* dbg;  // 64 timestamps per launch (tools/debug/c1d_timeline.py): [0,32) loader wave 0, [32,64) multiplier wave 4
#endif
};
#ifdef SDT_TUNING
#define C1_STAMP(slot)                                                                                             \
    do {                                                                                                           \
        if (a.dbg != nullptr && blockIdx.x == 0 && (tid & 255) == 0 && (slot) < 32) a.dbg[(loader ? 0 : 32) + (slot)] = wall_clock64(); \
    } while (0)
static unsigned long long* g_c1d_dbg = nullptr;
static int g_c1d_dbg_launch = 0, g_c1d_dbg_max = 0, g_c1d_dbg_mode = 0;
extern "C" void sdt_c1d_debug_mode(int m) { g_c1d_dbg_mode = m; }
#define C1_DBG_MODE(bit) (a.dbg_mode & (bit))
extern "C" void sdt_c1d_debug_buffer(void* p, int max_launches) {
    g_c1d_dbg = (unsigned long long*)p;
    g_c1d_dbg_launch = 0;
    g_c1d_dbg_max = max_launches;
}
#else
#define C1_STAMP(slot) do { } while (0)
#define C1_DBG_MODE(bit) 0
#endif

typedef float f32x2v __attribute__((ext_vector_type(2)));

// How the kernel got its shape (measured on MI355X, 12-16 K steps per launch, one 512-thread workgroup per CU):
//   v1  every thread loads, transforms and multiplies; one tile of prefetch ..................... 30 us / launch whatever the grid
//   v2  3-deep register ring of raw loads: no change -- hipcc drained vmcnt to 0 at every control-flow merge around a load
//   v3  static load counts (template on mode, out-of-range buffer loads instead of branches, loop exits instead of guards,
//       18 steps unrolled so that no back edge is taken): counted vmcnt(N) waits ................. 19 us
//       what was left: the two waves of a SIMD run in lockstep between the barriers, so a step cost (loader VALU) + (MFMA);
//       sched_group_barrier hints did not make hipcc interleave them inside a wave
//   v4  (this) wave specialisation: waves 0-3 are LOADERS (global -> registers -> transform -> LDS), waves 4-7 are MULTIPLIERS
//       (LDS -> MFMA); one of each per SIMD, so the hardware overlaps the loader's VALU with the multiplier's MFMA pipe.
// Loads: raw buffer loads whose offset is pushed out of range for masked elements (zeros come back, no memory access), issued
// C1 ring-depth steps ahead, also past the end of K (masked), so the number in flight is static and the waits are counted.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t c1_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4v c1_ld4(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}
__device__ __forceinline__ float c1_ld1(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
}
// sum of the np <= C1_NPART_MAX partial (a, b) pairs of one row, all loads in flight at once
__device__ __forceinline__ void c1_sum_partials(const __amdgpu_buffer_rsrc_t rs, unsigned row, int np, bool ok, float& s, float& q) {
    f32x2v v[C1_NPART_MAX];
#pragma unroll
    for (int i = 0; i < C1_NPART_MAX; ++i)
        v[i] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rs, (ok && i < np) ? (int)((row * np + i) * 8u) : (int)0x80000000u, 0, 0));
    s = 0.f;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < C1_NPART_MAX; ++i) {
        s += v[i][0];
        q += v[i][1];
    }
}
__device__ __forceinline__ void c1_mean_rstd(float s, float q, int C, float eps, float& mean, float& rstd) {
    mean = s / (float)C;
    float var = q / (float)C - mean * mean;
    var = var > 0.f ? var : 0.f;
    rstd = 1.f / sqrtf(var + eps);
}
// LDS-visible workgroup barrier usable inside the (wave-uniform) role branches: every wave executes the same NUMBER of them
__device__ __forceinline__ void c1_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// raw ring slot of one loader thread: 2 A elements (rows lr, lr + 16) and 4 B elements (rows lr + 16 j) of a 64-wide K step
template <int MODE>
struct c1d_raw {
    f32x4v x[2], b[4];
    int r[2];  // source row (< 0: masked)
};
template <>
struct c1d_raw<C1_IN_UPADD> {
    f32x4v x[2], b[4], p0[2], p1[2];
    int r[2];
};
template <>
struct c1d_raw<C1_IN_NORMBWD> {
    f32x4v x[2], b[4], p0[2];
    int r[2];
};

template <int MODE, bool BW>
__global__ __launch_bounds__(512) void c1d_kernel(const c1d_args a) {
    constexpr int D = MODE == C1_IN_NORMBWD ? 3 : (MODE == C1_IN_UPADD ? 4 : 5);  // ring depth in K steps (VGPR budget: 256 per wave)
    constexpr int U = D == 3 ? 6 : (D == 4 ? 4 : 10);                                                       // unrolled steps: a multiple of D and of 2
    constexpr bool bwd = MODE == C1_IN_NORMBWD;
    typedef c1d_raw<MODE> Raw;
    __shared__ __attribute__((aligned(16))) float sA[2][C1_BM * C1_PITCH];
    __shared__ __attribute__((aligned(16))) float sB[2][C1_BN * C1_PITCH];
    __shared__ __attribute__((aligned(16))) float sStat[2][C1_MAXROWS][4];  // per source row: mean, rstd, (s1 / C, s2 / C: backward mode)
    __shared__ float sBw[C1_BM][2];       // mean, rstd of the rows of bw_y this tile stores (backward-statistics epilogue)
    __shared__ float sPart[2][C1_BM][2];  // per wn: row partials of the epilogue

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < 4;
    const int M = a.B * a.To;
    const int nnb = (a.Cout + C1_BN - 1) / C1_BN;
    // split-K: consecutive workgroups are the K slices of one output tile.  Every slice stores its partial tile into its slab; the
    // LAST one to arrive (arrival counter) adds the slabs in slice order -- the result does not depend on the arrival order -- and
    // runs the epilogue.  One launch, no reduce pass, and the tiny-T layers (8-64 tiles) spread their 12-16 K steps over the chip.
    const int tile = (int)blockIdx.x / a.splitk, zslice = (int)blockIdx.x - tile * a.splitk;
    const int m0 = (tile / nnb) * C1_BM, n0 = (tile % nnb) * C1_BN;
    const int Ktot = a.taps * a.Cin;
    const int nit_all = (Ktot + C1_BK - 1) / C1_BK;
    const int step0 = (zslice * nit_all) / a.splitk, step1 = ((zslice + 1) * nit_all) / a.splitk;
    const int nit = step1 - step0;                    // K steps of this slice (>= 1: the host keeps splitk <= nit_all)
    const int kbeg = step0 * C1_BK, kend = min(Ktot, step1 * C1_BK);
    const unsigned OOB = 0x80000000u;
    const int sshift = a.stride - 1;  // stride is 1 or 2
    __shared__ int sLast;

    // source-row windows of the statistics tables.  Rows of the primary source: a contiguous window per batch item; the tile's 32
    // output rows may straddle two batch items, so the tables are indexed by (global row - first global row).
    const int b_first = m0 / a.To, to_first = m0 % a.To;
    int row_lo = 0, row_lo2 = 0;
    if constexpr (!bwd) {
        row_lo = b_first * a.Ti + max(0, to_first * a.stride - a.pad);
    } else {  // contributing layer-output rows for input rows [to_first, ...): (to + p - t) / s
        row_lo = b_first * a.Ti + max(0, (to_first + a.pad - (a.taps - 1)) >> sshift);
    }
    if constexpr (MODE == C1_IN_UPADD) row_lo2 = b_first * a.T2 + max(0, (max(0, to_first * a.stride - a.pad)) / 2 - 1);

    if (loader) {
        // =========================================================== LOADER WAVES ===========================================================
        const int lr = tid >> 4, lq = tid & 15;  // lr in [0, 16)
        bool am_ok[2];
        int ab[2], ato[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int am = m0 + lr + 16 * u;
            am_ok[u] = am < M;
            ab[u] = am_ok[u] ? am / a.To : 0;
            ato[u] = am_ok[u] ? am % a.To : 0;
        }
        const __amdgpu_buffer_rsrc_t rsX = c1_rsrc(a.X, (unsigned)a.B * a.Ti * a.Cin * 4u);
        const __amdgpu_buffer_rsrc_t rsX2 =
            c1_rsrc(MODE == C1_IN_UPADD || bwd ? a.X2 : a.X, (unsigned)a.B * (MODE == C1_IN_UPADD ? a.T2 : a.Ti) * a.Cin * 4u);
        const __amdgpu_buffer_rsrc_t rsW = c1_rsrc(a.W, (unsigned)a.Cout * Ktot * 4u);
        unsigned wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = n0 + lr + 16 * j < a.Cout ? (unsigned)(n0 + lr + 16 * j) * Ktot * 4u : OOB;
        const float up_scale = (float)a.T2 / (float)a.Ti;

        // F.interpolate(prev, Ti, mode='linear', align_corners=False) at output index ti (generator.py:79-83)
        auto lerp_src = [&](int ti, int& i0, int& i1, float& w1) {
            const float src = ((float)ti + 0.5f) * up_scale - 0.5f;
            const float sc = src < 0.f ? 0.f : src;
            i0 = (int)sc;
            i0 = i0 < a.T2 - 1 ? i0 : a.T2 - 1;
            i1 = i0 + 1 < a.T2 ? i0 + 1 : a.T2 - 1;
            w1 = sc - (float)i0;
        };
        // branch-free on purpose (bitwise & on the predicates, selects on the offsets)
        auto issue = [&](int k, Raw& q) {
            const bool kok = k < kend;
#pragma unroll
            for (int j = 0; j < 4; ++j) q.b[j] = c1_ld4(rsW, (kok & (wb[j] != OOB)) ? wb[j] + (unsigned)k * 4u : OOB);
            const int t = (int)__umulhi((unsigned)k, a.cin_magic), c = k - t * a.Cin;  // k / Cin, k % Cin (exactness checked on the host)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int ti;
                bool ok = am_ok[u] & kok;
                if constexpr (!bwd) {
                    ti = ato[u] * a.stride + t - a.pad;
                    ok = ok & ((unsigned)ti < (unsigned)a.Ti);
                } else {  // the layer-output row that tap t maps output row ato to
                    const int num = ato[u] + a.pad - t;
                    ti = num >> sshift;
                    ok = ok & (num >= 0) & ((num & sshift) == 0) & (ti < a.Ti);
                }
                const int row = ab[u] * a.Ti + ti;
                q.r[u] = ok ? row : -1;
                const unsigned xo = ok ? (unsigned)(row * a.Cin + c) * 4u : OOB;
                q.x[u] = c1_ld4(rsX, xo);
                if constexpr (bwd) q.p0[u] = c1_ld4(rsX2, xo);
                if constexpr (MODE == C1_IN_UPADD) {
                    int i0, i1;
                    float w1;
                    lerp_src(ok ? ti : 0, i0, i1, w1);
                    q.p0[u] = c1_ld4(rsX2, ok ? (unsigned)((ab[u] * a.T2 + i0) * a.Cin + c) * 4u : OOB);
                    q.p1[u] = c1_ld4(rsX2, ok ? (unsigned)((ab[u] * a.T2 + i1) * a.Cin + c) * 4u : OOB);
                }
            }
        };

        // ---- everything the loaders read from global memory starts here, in one burst
        C1_STAMP(0);
        Raw ring[D];
#pragma unroll
        for (int s0 = 0; s0 < D; ++s0) issue(kbeg + s0 * C1_BK + 4 * lq, ring[s0]);
        C1_STAMP(1);

        // statistics tables: threads 0..79 the primary source, threads 128..207 the half-resolution source
        if constexpr (MODE != C1_IN_PLAIN) {
            if (tid < C1_MAXROWS) {
                const int r = row_lo + tid;
                const bool ok = r < a.B * a.Ti;
                const __amdgpu_buffer_rsrc_t rs = c1_rsrc(a.xstats, (unsigned)a.B * a.Ti * a.np_in * 8u);
                float s, q, mu, rsd;
                c1_sum_partials(rs, (unsigned)r, a.np_in, ok, s, q);
                c1_mean_rstd(s, q, a.Cin, a.eps, mu, rsd);
                sStat[0][tid][0] = mu;
                sStat[0][tid][1] = rsd;
                if constexpr (bwd) {
                    const __amdgpu_buffer_rsrc_t rs2 = c1_rsrc(a.x2stats, (unsigned)a.B * a.Ti * a.np_in2 * 8u);
                    c1_sum_partials(rs2, (unsigned)r, a.np_in2, ok, s, q);
                    sStat[0][tid][2] = s / (float)a.Cin;
                    sStat[0][tid][3] = q / (float)a.Cin;
                }
            }
        }
        if constexpr (MODE == C1_IN_UPADD) {
            if (tid >= 128 && tid < 128 + C1_MAXROWS) {
                const int j = tid - 128, r = row_lo2 + j;
                const __amdgpu_buffer_rsrc_t rs = c1_rsrc(a.x2stats, (unsigned)a.B * a.T2 * a.np_in2 * 8u);
                float s, q, mu, rsd;
                c1_sum_partials(rs, (unsigned)r, a.np_in2, r < a.B * a.T2, s, q);
                c1_mean_rstd(s, q, a.Cin, a.eps, mu, rsd);
                sStat[1][j][0] = mu;
                sStat[1][j][1] = rsd;
            }
        }
        C1_STAMP(2);
        c1_barrier();  // (1) tables complete
        C1_STAMP(3);

        auto act = [&](float u) { return u > 0.f ? u : u * a.slope; };
        auto finish = [&](const Raw& q, int u) -> f32x4v {  // the A element after the input transform
            f32x4v v = q.x[u];  // zeros when masked
            if constexpr (MODE == C1_IN_PLAIN) {
                return v;
            } else {
                const bool ok = q.r[u] >= 0;
                const f32x4v st = *(const f32x4v*)sStat[0][ok ? q.r[u] - row_lo : 0];
                if constexpr (bwd) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yh = (q.p0[u][e] - st[0]) * st[1];
                        const float gg = q.x[u][e] * (yh > 0.f ? 1.f : a.slope);
                        v[e] = ok ? st[1] * (gg - st[2] - yh * st[3]) : 0.f;
                    }
                    return v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act((v[e] - st[0]) * st[1]);
                    if constexpr (MODE == C1_IN_UPADD) {
                        int i0, i1;
                        float w1;
                        lerp_src(ok ? q.r[u] - ab[u] * a.Ti : 0, i0, i1, w1);  // recomputed: cheaper than 3 more ring registers
                        const int j0 = ok ? ab[u] * a.T2 + i0 - row_lo2 : 0, j1 = ok ? ab[u] * a.T2 + i1 - row_lo2 : 0;
                        const f32x2v t0 = *(const f32x2v*)sStat[1][j0], t1 = *(const f32x2v*)sStat[1][j1];
                        const float w0 = 1.f - w1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += w0 * act((q.p0[u][e] - t0[0]) * t0[1]) + w1 * act((q.p1[u][e] - t1[0]) * t1[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
                    return v;
                }
            }
        };
        auto to_lds = [&](const Raw& q, int buf) {
#pragma unroll
            for (int u = 0; u < 2; ++u) *(f32x4v*)&sA[buf][(lr + 16 * u) * C1_PITCH + 4 * lq] = finish(q, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4v*)&sB[buf][(lr + 16 * j) * C1_PITCH + 4 * lq] = q.b[j];
        };

        to_lds(ring[0], 0);
        C1_STAMP(4);
        c1_barrier();  // (2) tile 0 in LDS
        C1_STAMP(5);
        // U steps unrolled back to back with exits (a guard around the body or a back edge would make hipcc drain vmcnt to 0 at the
        // merge).  The layers of the sdt generator have 4..16 K steps, so the back edge is never taken there.
        for (int it0 = 0;; it0 += U) {
#pragma unroll
            for (int s0 = 0; s0 < U; ++s0) {
                const int it = it0 + s0;
                if (!C1_DBG_MODE(2)) to_lds(ring[(s0 + 1) % D], (s0 & 1) ^ 1);        // tile it+1 (zeros past the end of K): issued D-1 steps ago
                if (!C1_DBG_MODE(4)) issue(kbeg + (it + D) * C1_BK + 4 * lq, ring[s0 % D]);  // slot of tile `it`, moved to LDS one step ago
                C1_STAMP(6 + it);
                c1_barrier();                                    // (3 + it)
                if (it + 1 >= nit) goto loaders_done;
            }
        }
    loaders_done:
        C1_STAMP(30);
        if (a.splitk > 1) {
            c1_barrier();  // (F1) partial tile stored
            c1_barrier();  // (F2) sLast published
            if (!sLast) return;
        }
        c1_barrier();  // (E) the multipliers' epilogue partials
        C1_STAMP(31);
    } else {
        // ========================================================= MULTIPLIER WAVES =========================================================
        const int mw = wave - 4, wm = mw >> 1, wn = mw & 1;  // 16 rows x 32 columns per wave, full K
        C1_STAMP(0);
        // epilogue operands of this lane's 8 outputs (C/D layout of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg)
        const __amdgpu_buffer_rsrc_t rsAdd = c1_rsrc(a.add != nullptr ? a.add : a.Y, a.add != nullptr ? (unsigned)M * a.Cout * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rsBwy = c1_rsrc(BW ? a.bw_y : a.Y, BW ? (unsigned)M * a.Cout * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rsBias = c1_rsrc(a.bias != nullptr ? a.bias : a.Y, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
        float eadd[2][4], ebwy[2][4], ebias[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int n = n0 + wn * 32 + tl * 16 + (lane & 15);
            ebias[tl] = c1_ld1(rsBias, n < a.Cout ? (unsigned)n * 4u : OOB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
                const unsigned o = ((m < M) & (n < a.Cout)) ? (unsigned)(m * a.Cout + n) * 4u : OOB;
                eadd[tl][r] = c1_ld1(rsAdd, o);
                if constexpr (BW) ebwy[tl][r] = c1_ld1(rsBwy, o);
            }
        }
        if constexpr (BW) {
            if (tid >= 256 && tid < 256 + C1_BM) {
                const int j = tid - 256, m = m0 + j;
                const __amdgpu_buffer_rsrc_t rs = c1_rsrc(a.bw_stats, (unsigned)M * a.np_bw * 8u);
                float s, q, mu, rsd;
                c1_sum_partials(rs, (unsigned)m, a.np_bw, m < M, s, q);
                c1_mean_rstd(s, q, a.Cout, a.eps, mu, rsd);
                sBw[j][0] = mu;
                sBw[j][1] = rsd;
            }
        }
        C1_STAMP(2);
        c1_barrier();  // (1)
        C1_STAMP(3);
        c1_barrier();  // (2)
        C1_STAMP(5);

        f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // fragment addresses: lane = (row l & 15, k-slot h = l >> 4); sub-step (j, e) consumes k = 16 j + 4 h + e for A and B alike
        const int fa = (wm * 16 + (lane & 15)) * C1_PITCH + (lane >> 4) * 4;
        const int fb = (wn * 32 + (lane & 15)) * C1_PITCH + (lane >> 4) * 4;
        for (int it = 0;; ++it) {
            const float* pa = sA[it & 1] + fa;
            const float* pb = sB[it & 1] + fb;
            if (!C1_DBG_MODE(1))
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4v av = *(const f32x4v*)(pa + 16 * j);
                const f32x4v b0 = *(const f32x4v*)(pb + 16 * j);
                const f32x4v b1 = *(const f32x4v*)(pb + 16 * C1_PITCH + 16 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b0[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b1[e], acc[1], 0, 0, 0);
                }
            }
            C1_STAMP(6 + it);
            c1_barrier();  // (3 + it)
            if (it + 1 >= nit) break;
        }

        if (a.splitk > 1) {  // ---- split-K fix-up
            const size_t slab = (size_t)M * a.Cout;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int n = n0 + wn * 32 + tl * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
                    if (m < M && n < a.Cout) a.slabs[(size_t)zslice * slab + (size_t)m * a.Cout + n] = acc[tl][r];
                }
            }
            __threadfence();  // the partial tile is visible device-wide before the arrival is counted
            c1_barrier();     // (F1)
            if (tid == 256) {
                const unsigned old = atomicAdd(&a.counters[tile], 1u);
                sLast = old == (unsigned)a.splitk - 1u;
                if (sLast) a.counters[tile] = 0u;  // every slice has arrived: leave the counter ready for the next launch
            }
            c1_barrier();  // (F2)
            if (!sLast) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int n = n0 + wn * 32 + tl * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
                    float v = 0.f;
                    if (m < M && n < a.Cout) {
                        const float* ps_ = a.slabs + (size_t)m * a.Cout + n;
                        for (int z = 0; z < a.splitk; ++z) v += __builtin_nontemporal_load(ps_ + (size_t)z * slab);
                    }
                    acc[tl][r] = v;
                }
            }
        }

        // ---- epilogue
        float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int n = n0 + wn * 32 + tl * 16 + (lane & 15);
            const bool nok = n < a.Cout;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 16 + (lane >> 4) * 4 + r;
                const int m = m0 + row;
                if (m < M && nok) {
                    const float v = acc[tl][r] + ebias[tl] + eadd[tl][r];
                    a.Y[(size_t)m * a.Cout + n] = v;
                    if constexpr (!BW) {
                        ps[r] += v;
                        pq[r] = fmaf(v, v, pq[r]);
                    } else {  // partial (sum g, sum g*yhat) of the layer below: v is the total gradient w.r.t. its activated output
                        const float yh = (ebwy[tl][r] - sBw[row][0]) * sBw[row][1];
                        const float gg = v * (yh > 0.f ? 1.f : a.slope);
                        ps[r] += gg;
                        pq[r] = fmaf(gg, yh, pq[r]);
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
        C1_STAMP(30);
        c1_barrier();  // (E)
        C1_STAMP(31);
    }
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
    const int nit_all = cdiv(p->taps * p->Cin, C1_BK);
    SDT_CHECK_ARG(p->splitk >= 1 && p->splitk <= 16 && p->splitk <= nit_all, "splitk must be in [1, min(16, K steps)]");
    SDT_CHECK_ARG(p->splitk == 1 || (p->slabs != nullptr && p->counters != nullptr), "split-K needs the slab workspace and the counters");
    a.splitk = p->splitk; a.slabs = p->slabs; a.counters = p->counters;
    a.cin_magic = (unsigned)((1ull << 32) / (unsigned)p->Cin + 1ull);
    SDT_CHECK_ARG((uint64_t)(p->taps * p->Cin + 64 * C1_BK) * (uint64_t)p->Cin < (1ull << 31), "K extent too large for the reciprocal division");
#ifdef SDT_TUNING
    a.dbg_mode = g_c1d_dbg_mode;
    a.dbg = (g_c1d_dbg != nullptr && g_c1d_dbg_launch < g_c1d_dbg_max) ? g_c1d_dbg + 64 * (g_c1d_dbg_launch++) : nullptr;
#endif
    a.in_mode = p->in_mode; a.np_in = p->np_in; a.np_in2 = p->np_in2; a.np_bw = p->np_bw; a.eps = p->eps; a.slope = p->slope;
    const int M = p->B * p->To;
    const unsigned grid = (unsigned)(cdiv(M, C1_BM) * cdiv(p->Cout, C1_BN) * p->splitk);
    const bool bw = p->bw_y != nullptr;
#define C1_LAUNCH(MODE)                                                                                      \
    if (bw) hipLaunchKernelGGL((c1d_kernel<MODE, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);  \
    else hipLaunchKernelGGL((c1d_kernel<MODE, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a)
    switch (p->in_mode) {
        case C1_IN_PLAIN: C1_LAUNCH(C1_IN_PLAIN); break;
        case C1_IN_NORM: C1_LAUNCH(C1_IN_NORM); break;
        case C1_IN_UPADD: C1_LAUNCH(C1_IN_UPADD); break;
        default: C1_LAUNCH(C1_IN_NORMBWD); break;
    }
#undef C1_LAUNCH
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
