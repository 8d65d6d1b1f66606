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
// fragments -> 6 MFMAs per (32x32x16) block.  A "planes" tensor holds 3 * numel bf16 in the interleaved layout of common.h
// (planes_index): per row and 32-channel chunk, the three pieces are adjacent 64-byte runs.
//
// Same tap-table contract, tap culling, XCD remap, multi-class launches and statistics epilogues as conv_taps_kernel.
#include <stdlib.h>

#include "common.h"
#include "../../include/sdt_hip_experimental.h"

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
// Workgroup = WM x WN waves, each owning a (BM/WM) x (BN/WN) sub-tile as 32x32 accumulators.
template <int BM, int BN, int WM, int WN, int EPI, int NBUF>
__global__ __launch_bounds__(64 * WM * WN) void conv_taps_pre_kernel(const __bf16* __restrict__ Xp, const size_t xplane,
                                                            const __bf16* __restrict__ Wp, const size_t wplane,
                                                            float* __restrict__ Y, const geom_pack_p gp,
                                                            double* __restrict__ stats, const int rows_per_group,
                                                            const norm_bwd_args_p nb) {
    constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;            // 32x32 accumulator tiles per wave
    constexpr int RA = BM / 16 / NW, RB = BN / 16 / NW;    // 16-row groups per wave: wave w stages rows [16 (w + NW i), +16) of each tile
    static_assert(TM >= 1 && TN >= 1 && RA >= 1 && RB >= 1 && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile / wave grid mismatch");
    constexpr int PLANEA = BM * BKP, PLANEB = BN * BKP;
    // Staging is asynchronous global -> LDS (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  An LDS-DMA
    // writes wave-uniform base + lane * 16 B, so a wave instruction fills 16 rows x 64 B of one piece and the bank swizzle moves
    // to the SOURCE side: lane (row, position q) fetches chunk q ^ ((row >> 2) & 3) of its row.  The operand tiles live in a ring
    // of NBUF LDS buffers: the DMA of K step s + NBUF - 1 is issued while step s computes, a wave waits only for its OLDEST
    // outstanding step (counted s_waitcnt vmcnt) and there is ONE raw s_barrier per step -- __syncthreads() would drain every
    // DMA in flight (hipcc emits vmcnt(0) in front of it), and so would a second __shared__ object next to the ring (hipcc then
    // waits vmcnt(0) before every ds_read): everything sits in ONE LDS array (cdna_hip_programming.md section 5).
    constexpr int STAGE = 3 * (PLANEA + PLANEB);                       // bf16 elements per ring slot: A pieces, then B pieces
    constexpr int MISC_INTS = BM + 3 * SDT_MAX_TAPS + SDT_MAX_TAPS + 1;
    __shared__ __attribute__((aligned(16))) __bf16 smem[NBUF * STAGE + 2 * MISC_INTS + 8];
    int* const sOut = (int*)(smem + NBUF * STAGE);
    int* const sTap = sOut + BM;
    int* const sLive = sTap + 3 * SDT_MAX_TAPS;

    const sdt_conv_geom& gt = gp.g[blockIdx.y];
    struct {
        int B, Hi, Wi, Cin, Ho, Wo, Hy, Wy, Cout, sy, sx, osy, osx, ooy, oox, ntaps, Tw;
    } g;
#define SDT_SGPR(f) g.f = __builtin_amdgcn_readfirstlane(gt.f)
    SDT_SGPR(B); SDT_SGPR(Hi); SDT_SGPR(Wi); SDT_SGPR(Cin); SDT_SGPR(Ho); SDT_SGPR(Wo); SDT_SGPR(Hy); SDT_SGPR(Wy); SDT_SGPR(Cout);
    SDT_SGPR(sy); SDT_SGPR(sx); SDT_SGPR(osy); SDT_SGPR(osx); SDT_SGPR(ooy); SDT_SGPR(oox); SDT_SGPR(ntaps); SDT_SGPR(Tw);
#undef SDT_SGPR

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
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
    for (int rr = tid; rr < BM; rr += 64 * NW) {
        int m = m0 + rr, off = -1;
        if (m < M) {
            int ox = m % g.Wo, t = m / g.Wo;
            int oy = t % g.Ho, b = t / g.Ho;
            off = ((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout;
        }
        sOut[rr] = off;
    }
    // loader mapping: lane = (row within the 16-row group, 16-byte position); wave w owns the row groups w + 4 i
    const int kc4 = lane & 3, r0 = wave * 16 + (lane >> 2);
    int rbH[RA], riy[RA], rix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 16 * NW * i;
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

    // planes layout (common.h planes_index): a row is 3*Cin bf16 = 6*Cin bytes, a 32-channel chunk 192 bytes = three 64-byte pieces;
    // byte offsets; masked rows use SDT_OOB (hardware returns zeros)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)Xp, 0, (int)(3u * (unsigned)xplane * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)(3u * (unsigned)wplane * 2u), 0x00020000);
    unsigned aoff[RA], boff[RB], abase[RA], bbase[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) abase[i] = (unsigned)(((rbH[i] + riy[i]) * g.Wi + rix[i]) * g.Cin) * 6u;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 16 * NW * i;
        bbase[i] = n < g.Cout ? (unsigned)(n * g.Tw * g.Cin) * 6u : SDT_OOB;
    }
    int cur_tl = -1, nxt_tl = 0, nxt_kc = 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned swz = (unsigned)((kc4 ^ ((r0 >> 2) & 3)) * 16);  // byte offset of the chunk this lane fetches (rows r0 + 64 i share it)
    // A step's DMA is issued in two parts: issue_prep picks the (tap, chunk) and rebuilds the row offsets when the tap changes;
    // issue_piece(i) launches ONE 1-KiB piece.  The K loop spreads the pieces of step s + NBUF - 1 between the MFMA groups of
    // step s: an LDS-DMA instruction sits in the issue stage for 60-190 cycles (the CU's 64 B/clk vector-memory path), and issued
    // back to back at the top of a step the 6-12 pieces of a wave stalled it for longer than its MFMAs run.
    __bf16* isa = smem;
    __bf16* isb = smem;
    int ics = 0;
    auto issue_prep = [&](int buf) {
        isa = smem + buf * STAGE;
        isb = isa + 3 * PLANEA;
        const int tl = nxt_tl, kc = nxt_kc;
        if (++nxt_kc == nkc) nxt_kc = 0, ++nxt_tl;
        ics = kc * 192;  // byte offset of the 32-channel chunk inside a row
        if (tl != cur_tl) {
            cur_tl = tl;
            const int t = sLive[tl];
            const int dy = sTap[t], dx = sTap[SDT_MAX_TAPS + t], wt = sTap[2 * SDT_MAX_TAPS + t];
            const unsigned ashift = (unsigned)((dy * g.Wi + dx) * g.Cin) * 6u + swz;
            const unsigned bshift = (unsigned)(wt * g.Cin) * 6u + swz;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                aoff[i] = ok ? abase[i] + ashift : SDT_OOB;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) boff[i] = bbase[i] == SDT_OOB ? SDT_OOB : bbase[i] + bshift;
        }
    };
    // piece index = plane * (RA + RB) + (row group: A groups first).  Each instruction: 1 KiB = 16 rows x 64 B of one plane,
    // lane * 16 B apart, at rows 16 (wave + NW i) of the tile
    auto issue_piece = [&](int idx) {
        const int p = idx / (RA + RB), r = idx - p * (RA + RB);
        if (r < RA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(isa + p * PLANEA + (wave * 16 + 16 * NW * r) * BKP), 16, (int)aoff[r],
                                                     ics + p * 64, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(isb + p * PLANEB + (wave * 16 + 16 * NW * (r - RA)) * BKP), 16,
                                                     (int)boff[r - RA], ics + p * 64, 0, 0);
    };
    auto issue = [&](int buf) {
        issue_prep(buf);
#pragma unroll
        for (int i = 0; i < 3 * (RA + RB); ++i) issue_piece(i);
    };

    // Accumulators.  The bf16 MFMA issues every 32 cycles but a DEPENDENT one (same accumulator) waits 64: five correction products
    // chained into one accumulator ran the whole K loop at half rate.  The six products of a block are therefore issued
    // product-major over the wave's tiles (consecutive MFMAs hit different accumulators), and a wave that owns a single tile keeps
    // THREE accumulators (large term | a0b1, a1b1, a2b0 | a1b0, a0b2) so that no accumulator is reused within two issue slots.
    constexpr int NA = (TM * TN == 1) ? 3 : 2;
    f32x16 acc[NA][TM][TN];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    // fragment reads: MFMA k-slot e of lane half h <-> k = 16j + 8h + e, i.e. the 16-byte chunk 2j + h of row (.. + lane & 31)
    const int fsw = (lane >> 2) & 3, fh = lane >> 5;
    const int foff[2] = {((fh ^ fsw) << 3), (((2 + fh) ^ fsw) << 3)};
    const int rowA = (wm * WTM + (lane & 31)) * BKP, rowB = (wn * WTN + (lane & 31)) * BKP;

    // vmcnt(N) immediates: N = DMA pieces of the steps that may stay in flight; gfx9 encoding: vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14
    constexpr int PIECES = 3 * (RA + RB);
#define SDT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))
    static_assert(PIECES * (NBUF - 2) < 64, "vmcnt counter range");
#pragma unroll
    for (int s0 = 0; s0 < NBUF - 1; ++s0)
        if (s0 < nsteps) issue(s0);
    int buf = 0, nbuf = NBUF - 1;  // slot of the current step / slot the next issue goes to
    for (int step = 0; step < nsteps; ++step) {
        // my pieces of `step` have landed once at most the (NBUF - 2) younger steps issued so far are still outstanding
        const int younger = min(NBUF - 2, nsteps - 1 - step);
        if (NBUF == 2 || younger == 0) SDT_VMCNT(0);
        else if (younger == 1) SDT_VMCNT(PIECES);
        else SDT_VMCNT(2 * PIECES);
        __builtin_amdgcn_s_barrier();  // everyone's pieces of `step` have landed, and everyone is done reading the slot reused below
        const bool more = step + NBUF - 1 < nsteps;
        if (more) issue_prep(nbuf);
        const __bf16* pa = smem + buf * STAGE + rowA;
        const __bf16* pb = smem + buf * STAGE + 3 * PLANEA + rowB;
        nbuf = buf;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
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
            // (A piece, B piece, accumulator) of the six products, product-major over the tiles
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0}, PC[6] = {0, 1, NA - 1, 1, NA - 1, 1};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[PC[q]][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][PA[q]], b[tn][PB[q]], acc[PC[q]][tm][tn], 0, 0, 0);
                // the DMA pieces of step + NBUF - 1 that belong behind MFMA group 6 j + q (12 groups per step)
                if (more) {
#pragma unroll
                    for (int i = 0; i < PIECES; ++i)
                        if ((i * 12) / PIECES == 6 * j + q) issue_piece(i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // corrections first (small terms), then the large term
                float c = acc[1][i][j][r];
                if constexpr (NA == 3) c += acc[2][i][j][r];
                acc[0][i][j][r] += c;
            }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)   (as conv_taps_kernel, incl. its statistics modes)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * WTN + tn * 32 + (lane & 31);
            const bool nok = n < g.Cout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int off = sOut[row];
                if (off >= 0 && nok) Y[(size_t)off + n] = acc[0][tm][tn][r];
            }
            if constexpr (EPI == 1) {  // forward statistics of the normalisation that follows
                const int g0 = m0 / rows_per_group;
                const int mb = (g0 + 1) * rows_per_group;
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WTM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (sOut[row] >= 0 && nok) {
                        const float v = acc[0][tm][tn][r];
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
                    const int row = wm * WTM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int off = sOut[row];
                    if (off >= 0 && nok) {
                        const float yv = nb.y[(size_t)off + n];
                        const bool second = m0 + row >= mb;
                        const float yh = (yv - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
                        const float gg = acc[0][tm][tn][r] * act_grad(yh * ga + be, nb.slope);
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
// x (rows, C) fp32 -> planes (layout: common.h planes_index): standalone split, for tensors whose producer is not one of ours.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, size_t n, int C) {
    const size_t nv = n >> 2;
    const int cq = C >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(x + 4 * i);
        store_planes4(v, planes + planes_index(i / cq, 4 * (int)(i % cq), C));
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
            const size_t o = planes_index((size_t)co * d.taps + t, ci, d.cin);  // rows = (cout, tap), channels = cin
            wp[o] = p0;
            wp[o + 32] = p1;
            wp[o + 64] = p2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty + 8 * i, co = co0 + tx;
        if (ci < d.cin && co < d.cout) {
            __bf16 p0, p1, p2;
            split3(tile[tx][ty + 8 * i], p0, p1, p2);
            const size_t o = planes_index((size_t)ci * d.taps + t, co, d.cout);  // mirror: rows = (cin, tap), channels = cout
            wtp[o] = p0;
            wtp[o + 32] = p1;
            wtp[o + 64] = p2;
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
    SDT_CHECK_ARG(tile == 0 || tile == 64064 || tile == 128064 || tile == 128128 || tile == 1281288 || tile == 1282568,
                  "unknown tile");
    g_pre_tile = tile;
    return SDT_OK;
}

template <int BM, int BN, int WM = 2, int WN = 2, int NBUF = 3>
static void launch_pre(const __bf16* xp, size_t xplane, const __bf16* wp, size_t wplane, float* y, const sdt_conv_geom* const* gs,
                       int ncls, double* stats, int rpg, const norm_bwd_args_p& nb, hipStream_t s) {
    geom_pack_p gp;
    int tiles = 0;
    for (int c = 0; c < SDT_MAX_CLASSES; ++c) gp.g[c] = *gs[c < ncls ? c : 0];
    for (int c = 0; c < ncls; ++c) tiles = std::max(tiles, cdiv(gs[c]->B * gs[c]->Ho * gs[c]->Wo, BM) * cdiv(gs[c]->Cout, BN));
    dim3 grid(tiles, ncls, 1);
    if (stats != nullptr)
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, WM, WN, 1, NBUF>), grid, dim3(64 * WM * WN), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
    else if (nb.sums != nullptr)
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, WM, WN, 2, NBUF>), grid, dim3(64 * WM * WN), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
    else
        hipLaunchKernelGGL((conv_taps_pre_kernel<BM, BN, WM, WN, 0, NBUF>), grid, dim3(64 * WM * WN), 0, s, xp, xplane, wp, wplane, y, gp, stats, rpg, nb);
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
    if (tile == 0) tile = 64064;  // measured (tools/pre_bench.py, B=32): 64x64 >= 128x64 > 128x128 on every audio-encoder layer
    switch (tile) {
        case 128128: launch_pre<128, 128>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        case 1281288: launch_pre<128, 128, 2, 4>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        case 1282568: launch_pre<128, 256, 2, 4, 2>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        case 128064: launch_pre<128, 64>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
        default: launch_pre<64, 64>(xp, (size_t)x_plane_elems, wp, (size_t)w_plane_elems, y, gs, ncls, stats, rows_per_group, nb, s); break;
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_split_planes_f32(const float* x, void* planes, int64_t n, int C, void* stream) {
    SDT_CHECK_ARG(x && planes && n > 0 && C > 0 && C % 32 == 0 && n % C == 0, "bad argument (C must be a multiple of 32 dividing n)");
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n / 4, 256), 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)planes, (size_t)n, C);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_weight_planes_batched(const sdt_wp_desc* table, int n_layers, int total_tiles, void* stream) {
    SDT_CHECK_ARG(table && n_layers > 0 && total_tiles > 0, "bad argument");
    hipLaunchKernelGGL(weight_planes_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n_layers);
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
