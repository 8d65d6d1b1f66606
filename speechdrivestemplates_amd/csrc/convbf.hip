// The bf16-SHAPED persistent stream-K convolution (round 5): Conv2d forward / input gradient of the audio encoder on bf16 tensors
// (building_blocks.py:15-22; generator.py:15-30; BASELINE config 4), v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Why a third conv kernel.  Round 4 ran the bf16-storage path on the fp32 kernel's shape (convsk_kernel<__bf16,128,128>: 4 waves, 64x64 per
// wave, two workgroups per CU): 0.17 of the dense bf16 matrix peak, bound by neither HBM nor the matrix pipe.  PMC of that kernel
// (profiles/r05_pmc_bf16.txt): MFMA pipe 19 % busy; a wave issues 30 % of its cycles, waits (s_waitcnt / barrier) 35 %, stalls at issue 35 %;
// the L1 forwards every request to the L2 (511 MB per launch for 100 MB of tensors: the im2col expansion) and is stalled on pending returns a
// third of the time; waves exist for 70 % of the launch (tile switches and stream-K hand-offs at different times per workgroup).  The bf16
// MFMA is 16x faster than the fp32 one, the feeding paths are not: at 2.5 PFLOP/s a 128x128x64 step needs 62 B/clk per CU from the L1 (its
// limit is ~64) and 129 % of the LDS (stores 79 B/clk, fragment reads 256 B/clk) -- the fp32 tile shape cannot be fed.
//
// Shape.  ONE workgroup of 8 waves per CU owns a 256 x BN tile (BN = 256 / 128 / 64 = the layer's Cout, so the A rows -- the expensive,
// gathered operand -- are loaded once per K step for ALL output channels):
//     BN = 256: waves 2 (M) x 4 (N), 128 x 64 per wave (8 accumulators; 6 fragment reads per 8 MFMAs)     31 B/clk of L1, 76 % of the LDS at the peak
//     BN = 128: waves 4 x 2,  64 x 64 per wave                                                            47 B/clk
//     BN =  64: waves 4 x 2,  64 x 32 per wave
// Two waves per SIMD of the SAME workgroup: while one waits for its fragments the other one's MFMAs run.  One barrier per K step of 64
// channels = per 2048 (BN 256) matrix-pipe cycles of a SIMD instead of per 512.  The plan (row tables in bytes, live-tap masks, stream-K
// ranges: sdt_convsk_plan_build_t with one workgroup per CU), the branch-free loader, the slab / flag hand-off of split tiles and the three
// epilogues are the round-3/4 kernel's (convsk.hip), re-shaped for 512 threads.
//
// Tile switch.  With one workgroup per CU nothing else covers a tile's set-up (dependent table reads), pipeline fill (one exposed HBM latency)
// and epilogue, so they are software-pipelined: the NEXT tile's row table is requested while the current tile's K loop runs, its first K step's
// operands are requested BEFORE the current tile's end phase (the staging registers are dead by then) and land under the stores / statistics.
#include <type_traits>

#include "convsk.h"

#define BF_NT 512
// cache policy of the weight (B operand) requests: 0 plain; 2 = nt -- served by the L2 without allocating in the CU's vector L1 (a tile's weight
// lines are read once per K step and never again by this CU before 32 KB of other lines have passed), which leaves the L1 to the im2col rows
#ifndef BF_B_AUX
#define BF_B_AUX 0
#endif
// BF_ABL (tools/debug/r05_bf2_ablation.sh; ablation builds compute WRONG results by design): 1 no global loads in the K loop, 2 no LDS stores,
// 4 no MFMAs, 8 no barrier in the K loop, 16 no fragment reads, 32 no epilogue stores, 64 requests never waited for, 128 / 256 A operand on every
// other step / never (split form)
#ifndef BF_ABL
#define BF_ABL 0
#endif
#if BF_ABL & 4
#define BF_MFMA(C, A, B) asm volatile("" : "+v"(C) : "v"(A), "v"(B))
#else
#define BF_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, A), __builtin_bit_cast(sk_bf16x8, B), C, 0, 0, 0)
#endif

#ifdef SDT_TUNING
// tools/debug/sk_timeline.py --kernel bf2: lane 0 of every workgroup stamps the 100 MHz counter at five points of each of its first 16 segments
__device__ unsigned long long* bf2_dbg_tl = nullptr;
extern "C" int sdt_debug_set_timeline_bf2(void* p) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(bf2_dbg_tl), &p, sizeof(p));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define BF_TL(slot, val)                                                                                                   \
    do {                                                                                                                   \
        if (threadIdx.x == 0 && bf2_dbg_tl != nullptr && seg < 16) bf2_dbg_tl[((size_t)r * 16 + seg) * 8 + (slot)] = (val); \
    } while (0)
// fault injection (tests/test_ops_gpu.py::test_streamk_lost_partner_is_loud): range `bf2_dbg_mute_range` computes its partial tile but never raises its flag
__device__ int bf2_dbg_mute_range = -1;
int convbf2_debug_mute_range(int r) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(bf2_dbg_mute_range), &r, sizeof(r));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define BF_MUTED(r) ((r) == bf2_dbg_mute_range)
#else
#define BF_TL(slot, val) do { } while (0)
#define BF_MUTED(r) false
#endif

// Statistics of ONE 32 x 32 accumulator block (rows rb0 .. rb0 + 31 of the tile, this lane's column n): sum u and sum u * v over the block's rows with
//   EPI 1: u = v = conv output;   EPI 2: v = yhat of the block below, u = dX * act'(yhat)   -- (s0, q0) for the rows of group `gfirst`, (s1, q1) for
// the rows of a second group `glast` (groups ascend with the row; rows past the end of the tensor carry group -1 and come last).
template <int EPI>
__device__ __forceinline__ void bf2_block_stats(const f32x16& a, const float* yv, const int* sOut, const int* sGrp, const int rb0, const int lane, const float bv,
                                                const int gfirst, const int glast, const float mu0, const float rs0, const float mu1, const float rs1,
                                                const float ga, const float be, const float slope, float& s0, float& q0, float& s1, float& q1) {
    if (gfirst == glast && gfirst >= 0) {  // wave-uniform fast path: one group, every row exists -- no per-element selects (the statistics were VALU-bound)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if constexpr (EPI == 1) {
                const float u = a[q] + bv;
                s0 += u;
                q0 = fmaf(u, u, q0);
            } else {
                const float v = (yv[q] - mu0) * rs0;
                const float u = a[q] * act_grad(v * ga + be, slope);
                s0 += u;
                q0 = fmaf(u, v, q0);
            }
        }
        return;
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int4 o4 = *(const int4*)&sOut[rb0 + 8 * qq + 4 * (lane >> 5)];
        const int4 g4 = *(const int4*)&sGrp[rb0 + 8 * qq + 4 * (lane >> 5)];
        const int offs[4] = {o4.x, o4.y, o4.z, o4.w}, grps[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = offs[e] >= 0, second = grps[e] != gfirst;
            float u, v;
            if constexpr (EPI == 1) {
                u = valid ? a[4 * qq + e] + bv : 0.f;
                v = u;
            } else {
                v = (yv[4 * qq + e] - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
                u = valid ? a[4 * qq + e] * act_grad(v * ga + be, slope) : 0.f;
            }
            if (!second) {
                s0 += u;
                q0 = fmaf(u, v, q0);
            } else {
                s1 += u;
                q1 = fmaf(u, v, q1);
            }
        }
    }
}

// Epilogue of accumulator rows [TMB, TME) x all TN column blocks of one wave: branch-free bf16 stores (pairs of columns as one dword, exchanged
// between neighbouring lanes by DPP) and, EPI 1 / 2, the per-(group, channel) statistics from the fp32 accumulators.  sOut / sGrp: LDS, byte
// offset of each tile row in Y (SK_OOB: none) and its statistics group.  row0: first tile row of the wave, ncol0: first output channel of the wave.
template <typename OT, int TM, int TN, int EPI, int TMB, int TME>
__device__ __forceinline__ void bf2_epilogue(f32x16 (&acc)[TM][TN], const int* sOut, const int* sGrp, const int row0, const int ncol0, const int lane,
                                             const float* __restrict__ bias, const __amdgpu_buffer_rsrc_t rsY, const int Cout,
                                             double* __restrict__ stats, const sk_norm_bwd& nb, const unsigned ybytes, float* red) {
    // red != nullptr: every row of the TILE belongs to one group -- the wave's column sums go to its LDS slice red[(tn * 32 + column) * 2 + {0, 1}]
    // (first row-block pair: store, later ones: add) and the caller adds the waves' slices: one partial per column and TILE instead of one
    // fp64 atomic per column and 64 rows (the atomics, not their arithmetic, were 2.4 us of a tile's end phase)
    float yv[(TME - TMB) * TN][16];
    if constexpr (EPI == 2) {
        // the forward output y of the block below at every position of these rows: all loads before anything else (one exposed latency)
        const __amdgpu_buffer_rsrc_t rsNY = __builtin_amdgcn_make_buffer_rsrc((void*)nb.y, 0, (int)ybytes, 0x00020000);
#pragma unroll
        for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const unsigned nb2 = (unsigned)(ncol0 + tn * 32 + (lane & 31)) * (unsigned)sizeof(OT);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int4 o4 = *(const int4*)&sOut[row0 + tm * 32 + 8 * qq + 4 * (lane >> 5)];
                    const int offs[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (sizeof(OT) == 2)
                            yv[(tm - TMB) * TN + tn][4 * qq + e] = sk_bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(rsNY, (int)((unsigned)offs[e] + nb2), 0, 0));
                        else
                            yv[(tm - TMB) * TN + tn][4 * qq + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsNY, (int)((unsigned)offs[e] + nb2), 0, 0));
                    }
                }
            }
    }
    const bool odd = lane & 1;
#pragma unroll
    for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = ncol0 + tn * 32 + (lane & 31);
            const float bv = bias != nullptr ? bias[n] : 0.f;
            const unsigned nb2 = (unsigned)(n & ~1) * 2u;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int4 o4 = *(const int4*)&sOut[row0 + tm * 32 + 8 * qq + 4 * (lane >> 5)];
                if constexpr (sizeof(OT) == 4) {  // fp32 output: a lane stores its own column, 32 lanes = 128 contiguous bytes of a row
                    const int offs[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#if BF_ABL & 32
                        asm volatile("" ::"v"(acc[tm][tn][4 * qq + e]), "v"(offs[e]));
#else
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tm][tn][4 * qq + e] + bv), rsY, (int)((unsigned)offs[e] + (unsigned)n * 4u), 0, 0);
#endif
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float v0 = acc[tm][tn][4 * qq + 2 * h] + bv, v1 = acc[tm][tn][4 * qq + 2 * h + 1] + bv;
                        const float give = odd ? v0 : v1;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
                        sk_bf16x2 pk;
                        pk[0] = (__bf16)(odd ? got : v0);
                        pk[1] = (__bf16)(odd ? v1 : got);
                        const int ro = odd ? (h ? o4.w : o4.y) : (h ? o4.z : o4.x);
#if BF_ABL & 32  // ablation (wrong results): the epilogue's stores are not issued (the packing stays)
                        asm volatile("" ::"v"(pk), "v"(ro));
#else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rsY, (int)((unsigned)ro + nb2), 0, 0);
#endif
                    }
                }
            }
        }
    if constexpr (EPI == 1 || EPI == 2) {
        double* acc_out = EPI == 1 ? stats : nb.sums;
        constexpr int STEP = ((TME - TMB) % 2 == 0) ? 2 : 1;  // row blocks are taken in PAIRS: one partial sum of <= 64 rows per atomic (the fp32 kernels' bound)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = ncol0 + tn * 32 + (lane & 31);
            const float bv = bias != nullptr ? bias[n] : 0.f;
            float ga = 1.f, be = 0.f;
            if constexpr (EPI == 2) {
                if (nb.gamma != nullptr) ga = nb.gamma[n];
                if (nb.beta != nullptr) be = nb.beta[n];
            }
#pragma unroll
            for (int tm = TMB; tm < TME; tm += STEP) {
                int gf[STEP], gl[STEP];
                float s0[STEP], q0[STEP], s1[STEP], q1[STEP];
#pragma unroll
                for (int k = 0; k < STEP; ++k) {
                    const int rb0 = row0 + (tm + k) * 32;
                    gf[k] = sGrp[rb0], gl[k] = sGrp[rb0 + 31];
                    const bool two = gl[k] != gf[k] && gl[k] >= 0;
                    float mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f;
                    if constexpr (EPI == 2) {
                        if (gf[k] >= 0) {
                            mu0 = nb.mean[(size_t)gf[k] * Cout + n];
                            rs0 = nb.rstd[(size_t)gf[k] * Cout + n];
                        }
                        if (two) {
                            mu1 = nb.mean[(size_t)gl[k] * Cout + n];
                            rs1 = nb.rstd[(size_t)gl[k] * Cout + n];
                        }
                    }
                    s0[k] = q0[k] = s1[k] = q1[k] = 0.f;
                    bf2_block_stats<EPI>(acc[tm + k][tn], yv[(tm + k - TMB) * TN + tn], sOut, sGrp, rb0, lane, bv, gf[k], gl[k], mu0, rs0, mu1, rs1, ga, be, nb.slope,
                                         s0[k], q0[k], s1[k], q1[k]);
                    s0[k] += __shfl_xor(s0[k], 32, 64);  // the two lane halves hold different rows of the same column
                    q0[k] += __shfl_xor(q0[k], 32, 64);
                }
                // two blocks of one group: ONE partial sum (half the atomics); otherwise block by block
                bool merged = false;
                if constexpr (STEP == 2) merged = gf[0] == gl[0] && gf[1] == gl[1] && gf[0] == gf[1] && gf[0] >= 0;  // wave-uniform
                if (merged && red != nullptr) {
                    if (lane < 32) {
                        float* d = red + (tn * 32 + lane) * 2;
                        if (tm == 0) {
                            d[0] = s0[0] + s0[1];
                            d[1] = q0[0] + q0[1];
                        } else {
                            d[0] += s0[0] + s0[1];
                            d[1] += q0[0] + q0[1];
                        }
                    }
                } else if (merged) {
                    if (lane < 32) {
                        double* d = acc_out + ((size_t)gf[0] * Cout + n) * 2;
                        atomicAdd(d, (double)(s0[0] + s0[1]));
                        atomicAdd(d + 1, (double)(q0[0] + q0[1]));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < STEP; ++k) {
                        const bool two = gl[k] != gf[k] && gl[k] >= 0;
                        if (two) {
                            s1[k] += __shfl_xor(s1[k], 32, 64);
                            q1[k] += __shfl_xor(q1[k], 32, 64);
                        }
                        if (lane < 32 && gf[k] >= 0) {
                            double* d = acc_out + ((size_t)gf[k] * Cout + n) * 2;
                            atomicAdd(d, (double)s0[k]);
                            atomicAdd(d + 1, (double)q0[k]);
                            if (two) {
                                double* d1 = acc_out + ((size_t)gl[k] * Cout + n) * 2;
                                atomicAdd(d1, (double)s1[k]);
                                atomicAdd(d1 + 1, (double)q1[k]);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- The SPLIT-fp32 form (X3; fp32 tensors in HBM, fp32-grade results at the bf16 matrix rate).
// A fp32 number is the exact sum of three bf16 numbers: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (8 + 8 + 8 significand bits;
// both differences are exact in fp32).  A product a * b is then nine bf16 x bf16 products, each EXACT in the fp32 accumulator; the kernel adds six
// of them -- lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- and drops mid*lo, lo*mid, lo*lo, at most 2^-24 + 2^-24 + 2^-32 of |a b|: less than the
// rounding of ONE fp32 addition (2^-24 of the running sum), of which a dot product of K terms makes K.  Six v_mfma_f32_32x32x16_bf16 do the
// work of eight v_mfma_f32_32x32x2_f32 in 6 / 64 of the matrix-pipe time... (16x the rate, 6x the instructions: 2.7x faster at the pipe's peak).
// The split is made ONCE per loaded element, by the loader thread between its global load and its LDS store (11 VALU instructions per pair of
// elements), not per MFMA: LDS holds the three planes, a K step is 32 channels = 128 bytes of a fp32 row (the fp32 plan's step) = two k-groups
// of 6 NM MFMAs.  Bytes per MFMA cycle are a third of the bf16 kernel's, which is what that kernel was short of (profiles/r05_bf2_experiments.txt).
// Inf operands turn into NaN (inf - inf in the split), as they would after one more layer in fp32.
#define X3_ROW 16  // floats per LDS row of one plane: 32 bf16 channels
#ifndef X3_NSET
#define X3_NSET 2  // staging register sets of the split form
#endif
__device__ __forceinline__ void x3_stage(float* dst, const int plane_stride, const f32x4 v) {
    unsigned h[2], m[2], l[2];
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // (vector conversions: ONE v_cvt_pk_bf16_f32 per pair and plane; element-wise casts cost hipcc two)
        const f32x2_ x = {v[2 * p], v[2 * p + 1]};
        h[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, sk_bf16x2));
        const f32x2_ r = x - f32x2_{__uint_as_float(h[p] << 16), __uint_as_float(h[p] & 0xffff0000u)};
        m[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sk_bf16x2));
        const f32x2_ t = r - f32x2_{__uint_as_float(m[p] << 16), __uint_as_float(m[p] & 0xffff0000u)};
        l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, sk_bf16x2));
    }
    *(uint2*)dst = uint2{h[0], h[1]};
    *(uint2*)(dst + plane_stride) = uint2{m[0], m[1]};
    *(uint2*)(dst + 2 * plane_stride) = uint2{l[0], l[1]};
}

// sched_group_barrier templates of an X3 K step (literal counts, hence the recursion).  Region 1, per MFMA q of NQ: one fragment read while any are
// left, the VALU share of the splits (NL x 24 instructions over the first NQ - 2 MFMAs), one LDS store every other MFMA, one request every NQ / NL.
// NSP: operand rows a thread SPLITS per step (24 VALU + 3 ds_write_b64 each), NCP: rows it only COPIES (pre-split weight planes: one ds_write_b128 each)
template <int Q, int NQ, int NR, int NSP, int NCP>
__device__ __forceinline__ void x3_pattern() {
    if constexpr (Q < NQ) {
        constexpr int NL = NSP + NCP;
        SK_SGB(0x8, 1);
        if constexpr (Q < NR) SK_SGB(0x100, 1);
        constexpr int NV = NSP * 24, per = (NV + NQ - 3) / (NQ - 2);
        if constexpr (Q * per < NV) SK_SGB(0x2, per);
        constexpr int NW = NSP * 3 + NCP, w0 = Q * NW / NQ, w1 = (Q + 1) * NW / NQ;
        if constexpr (Q >= 2 && w1 > w0) SK_SGB(0x200, w1 - w0);
        constexpr int l0 = Q * NL / NQ, l1 = (Q + 1) * NL / NQ;
        if constexpr (Q >= 3 && l1 > l0) SK_SGB(0x20, l1 - l0);
        x3_pattern<Q + 1, NQ, NR, NSP, NCP>();
    }
}
template <int Q, int NQ, int NR>
__device__ __forceinline__ void x3_pattern2() {
    if constexpr (Q < NQ) {
        SK_SGB(0x8, 1);
        constexpr int r0 = Q * NR / NQ, r1 = (Q + 1) * NR / NQ;
        if constexpr (r1 > r0) SK_SGB(0x100, r1 - r0);
        x3_pattern2<Q + 1, NQ, NR>();
    }
}

// EPI: 0 = store (+ bias), 1 = + forward statistics (stats: fp64 atomics, zero on entry), 2 = + normalisation-backward statistics
// ET = __bf16: bf16 tensors, one MFMA per fragment pair.   ET = float: the SPLIT-fp32 form (X3) -- see the note above x3_split.
// W3 (split form only): W points to the weights ALREADY split -- three bf16 planes [hi | mid | lo], each (N, Tw * Cin), made once per optimiser step
// (sdt_split3_batched_f32) instead of by every tile of every launch on every K step: the B operand's share of the split (half of a 128 x 128
// tile's 96 VALU instructions per loader thread and step) leaves the loop; a B row of a step is 3 x 64 bytes copied global -> LDS in 16-byte chunks.
template <typename ET, int BM, int BN, int WGM, int WGN, int EPI, bool W3 = false>
__global__ __launch_bounds__(BF_NT, 2) void convbf2_kernel(const ET* __restrict__ X, const ET* __restrict__ W, const float* __restrict__ bias,
                                                          ET* __restrict__ Y, const sk_args P, double* __restrict__ stats, const sk_norm_bwd nb) {
    constexpr bool X3 = sizeof(ET) == 4;
    static_assert(X3 || !W3, "pre-split weights belong to the split-fp32 form");
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32, RA = BM / 64, RB = BN / 64, NM = TM * TN, NF = TM + TN;
    constexpr int RBL = W3 ? (BN * 12 + BF_NT - 1) / BF_NT : RB;  // B requests per loader thread and step (W3: BN rows x 3 planes x 4 chunks of 16 bytes)
    // LDS floats per buffer of A / B.  bf16: [row][128 B + 16 B pad].  X3: [plane hi / mid / lo][row][64 B], 16-byte chunks XOR-swizzled by (row >> 2) & 3
    constexpr int A_BUF = X3 ? 3 * BM * X3_ROW : BM * SK_LDP, B_BUF = X3 ? 3 * BN * X3_ROW : BN * SK_LDP;
    // Staging register sets = how many K steps a request has to land before its data is stored to LDS: two when the accumulators leave room.
    // (Three / four sets were measured, profiles/r05_bf2_experiments.txt: no gain -- the waves wait for their operands at ANY prefetch depth, the
    // delivery RATE of ~24 B/clk per CU is what a step waits for, not the latency of one request.)
    constexpr int NSET = BN == 256 ? 1 : (X3 ? X3_NSET : 2);
    constexpr int UNR = (NSET % 2 == 0) ? (NSET < 2 ? 2 : NSET) : 2 * NSET;  // period of (LDS buffer parity, staging set)
    static_assert(WGM * WGN == 8 && TM >= 1 && TN >= 1 && RB >= 1, "wave grid");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                          // [2][A_BUF]
    float* sB = smem + 2 * A_BUF;              // [2][B_BUF]
    int* sOutB = (int*)(sB + 2 * B_BUF);        // [2][BM] byte offset of the tile's output rows in Y (SK_OOB, negative as int: none)
    int* sGrpB = sOutB + 2 * BM;                // [2][BM] statistics group of each row (EPI 1 / 2)
    int2* sRowB = (int2*)(sGrpB + 2 * BM);      // [2][BM] {X byte offset of the row's (0,0) tap, mask of the taps outside X}: the loader's row table
    int* sFlagOkp = (int*)(sRowB + 2 * BM);     // [4]
#define sFlagOk (sFlagOkp[0])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int kv = tid & 7, r0 = tid >> 3;  // loader: 16-byte chunk kv of rows r0 + 64 i
    const int G = P.G;
    const int bid = blockIdx.x;
    const int r = (bid & 7) * (G >> 3) + (bid >> 3);  // G % 8 == 0: blocks of one XCD own consecutive ranges
    const int s_begin = (int)((long)r * P.S / G), s_end = (int)((long)(r + 1) * P.S / G);
    if (s_end <= s_begin) return;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)P.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, W3 ? (int)(P.wbytes / 2 * 3) : (int)P.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)P.ybytes, 0x00020000);

    const int row0 = wm * (TM * 32), col0 = wn * (TN * 32);
    // fragment of k-group J: bf16 -- 16 bytes at J * 32 + (lane >> 5) * 16 of the lane's row;  X3 -- chunk (2 J + (lane >> 5)) ^ swizzle of the lane's row
    const int fsw = (lane >> 2) & 3;  // (row0, col0 and the 32-row blocks are multiples of 32: the swizzle of a lane's row is the lane's)
    const int fa = X3 ? (row0 + (lane & 31)) * X3_ROW : (row0 + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int fb = X3 ? (col0 + (lane & 31)) * X3_ROW : (col0 + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int fk0 = (((lane >> 5)) ^ fsw) * 4, fk1 = ((2 + (lane >> 5)) ^ fsw) * 4;  // X3: float offset of the lane's chunk of k-group 0 / 1
    // loader thread (r0, kv): 16 bytes = 8 bf16 channels (bf16) / 4 fp32 channels that become 8 bytes per plane (X3)
    const int wofs = X3 ? r0 * X3_ROW + ((((kv >> 1) ^ ((r0 >> 2) & 3)) * 4) + (kv & 1) * 2) : r0 * SK_LDP + kv * 4;
    // W3: request j of this thread is 16-byte chunk c = (tid + 512 j) mod (12 BN) of the step's B tile: plane c / (4 BN), row (c mod 4 BN) / 4, chunk c & 3
    // (the wrapped-around chunks of the last request duplicate another thread's load and store: same address, same data)
    // BN = 128: 4 BN = 512 = the workgroup -- request j IS plane j of row tid >> 2: one LDS offset / one global base + a constant per plane
    constexpr bool W3_PLANE_PER_REQ = W3 && BN * 4 == BF_NT;
    auto b3_plane = [&](const int j) { return W3_PLANE_PER_REQ ? j : ((tid + BF_NT * j) % (BN * 12)) / (BN * 4); };
    auto b3_row = [&](const int j) { return W3_PLANE_PER_REQ ? (tid >> 2) : (((tid + BF_NT * j) % (BN * 12)) % (BN * 4)) >> 2; };
    int wofs3[(W3 && !W3_PLANE_PER_REQ) ? RBL : 1];
    if constexpr (W3) {
#pragma unroll
        for (int j = 0; j < (W3_PLANE_PER_REQ ? 1 : RBL); ++j)
            wofs3[j] = b3_plane(j) * (BN * X3_ROW) + b3_row(j) * X3_ROW + (((tid & 3) ^ ((b3_row(j) >> 2) & 3)) * 4);
    }

    // ---- the walk over the range's tiles: cursor (tile, class c, m-tile mt, n-tile nt)
    struct Cursor {
        int tile, c, mt, nt;
    };
    Cursor cur;
    cur.tile = __builtin_amdgcn_readfirstlane(P.range_tile[r]);
    cur.c = 0;
    while (cur.c + 1 < P.ncls && cur.tile >= P.cls[cur.c + 1].tile_begin) ++cur.c;
    cur.mt = (cur.tile - P.cls[cur.c].tile_begin) / P.nnb;
    cur.nt = (cur.tile - P.cls[cur.c].tile_begin) - cur.mt * P.nnb;
    if (P.ntmajor) {
        cur.nt = cur.tile / P.cls[0].nmb;
        cur.mt = cur.tile - cur.nt * P.cls[0].nmb;
    }
    auto advance = [&](Cursor k) -> Cursor {
        ++k.tile;
        if (P.ntmajor) {
            if (++k.mt == P.cls[0].nmb) {
                k.mt = 0;
                ++k.nt;
            }
        } else if (++k.nt == P.nnb) {
            k.nt = 0;
            if (++k.mt == P.cls[k.c].nmb) {
                k.mt = 0;
                ++k.c;
            }
        }
        return k;
    };
    int pos = s_begin;

    // ---- loader state of the tile whose operands are being fetched
    int nkc = 1, ntaps = 1, rot = 0, kc = 0, left = 0, v_ash = 0, v_bsh = 0, cls_loaded = -1, cls_cout = 0, cls_wrow = 0;
    unsigned rmask = 0u, rmask_full = 0u;
    // K order of a tile's live steps (P.korder, sdt_convsk_set_k_order).  0: tap-major -- all channel chunks of a tap, then the next tap (rounds 3-5).
    // 1: CHUNK-major -- all live taps of one 128-byte channel chunk, then the next chunk.  The im2col rows of tap (dy, dx + 1) are the rows of
    // (dy, dx) shifted by one pixel: 127 of a tile's 128 cache lines of a step were requested by the step BEFORE it, one step = 32 KB = the vector
    // L1's size ago, instead of nkc steps (64 - 256 KB) ago -- they hit (or merge with the miss in flight) instead of going to the L2 again.
    const bool chunk_major = P.korder != 0;
    unsigned abase[RA], inval[RA], bbase[W3_PLANE_PER_REQ ? 1 : RBL];
    unsigned b3_plane_bytes = 0u;  // W3_PLANE_PER_REQ: distance of the weight planes (scalar)
    f32x4 ra[NSET][RA], rb[NSET][RBL];
    // per-tile facts the end phase needs (set by `setup`, which runs BEFORE the previous tile's end phase when tiles are pipelined)
    struct TileFacts {
        int Cout, n0, a, b, tbeg, tend;
    };
    // one K step's requests: load_prep() (wave-uniform scalars of the step at (tap, kc)), load_op(set, i) (operand row i: A rows first, then B rows),
    // load_advance().  The K loop places the ops one by one between its MFMAs; `load` is all of them in a row (pipeline fill).
    int ld_ash = 0, ld_cs = 0, ld_sh = 0;
#if BF_ABL & 64
    f32x4 abl_sink[RA + RBL];
#endif
    unsigned ld_bsh = 0u;
    auto load_prep = [&]() {
        // past the end of the segment everything is masked (loads return zeros)
        const bool on = left > 0 && rmask != 0u;
        int t = (rmask != 0u ? __builtin_ctz(rmask) : 0) + rot;
        t = t >= ntaps ? t - ntaps : t;
        t = on ? t : 0;
        ld_ash = __builtin_amdgcn_readlane(v_ash, t);
        ld_bsh = (unsigned)__builtin_amdgcn_readlane(v_bsh, t) + (on ? 0u : SK_OOB);  // bbase + bshift < 2^31: the top bit pushes it out of range
        ld_cs = kc * 128;  // a K step is 128 bytes of a row
        if constexpr (W3) {    // ... and 64 bytes of a row of each weight plane; byte shifts of the bf16 planes are half the fp32 tensor's
            ld_bsh = ((unsigned)__builtin_amdgcn_readlane(v_bsh, t) >> 1) + (on ? 0u : SK_OOB);
        }
        // the row's invalid-tap bit moves to bit 31 of the offset: out of range, the load returns zeros.  Loader off: shift 0 brings the always-set bit 31 there
        ld_sh = on ? 31 - t : 0;
    };
    auto load_op = [&](auto SI, const int i) {
        constexpr int S_ = decltype(SI)::value;
        if (i < RA) {
            const unsigned o = ((abase[i] + (unsigned)ld_ash) & 0x7fffffffu) | ((inval[i] << ld_sh) & 0x80000000u);
#if BF_ABL & 1
            ra[S_][i][0] = __uint_as_float(o + (unsigned)ld_cs);
#elif BF_ABL & 64  // the request is issued but nobody waits for it (the LDS stores take stale registers): the cost of ISSUING requests vs WAITING for them
            abl_sink[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, ld_cs, 0));
#else
            ra[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, ld_cs, 0));
#endif
        } else if (i < RA + RBL) {
            // W3: 64 bytes of a plane's row per step; plane-per-request: the plane's distance rides in the scalar offset
            const int bcs = W3 ? (ld_cs >> 1) + (W3_PLANE_PER_REQ ? (i - RA) * (int)b3_plane_bytes : 0) : ld_cs;
            const unsigned bb = bbase[W3_PLANE_PER_REQ ? 0 : i - RA];
#if BF_ABL & 1
            rb[S_][i - RA][0] = __uint_as_float(bb + ld_bsh + (unsigned)bcs);
#elif BF_ABL & 64
            abl_sink[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(bb + ld_bsh), bcs, 0));
#else
            rb[S_][i - RA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(bb + ld_bsh), bcs, BF_B_AUX));
#endif
        }
    };
    auto load_advance = [&]() {
        --left;
        if (chunk_major) {  // (wave-uniform: scalar branch)
            const unsigned rest = rmask & (rmask - 1u);
            kc = rest == 0u ? kc + 1 : kc;            // past the last chunk only when `left` has run out: those loads are masked
            rmask = rest == 0u ? rmask_full : rest;
        } else {
            const bool wrap = kc + 1 == nkc;
            kc = wrap ? 0 : kc + 1;
            rmask = wrap ? (rmask & (rmask - 1u)) : rmask;
        }
    };
    auto load = [&](auto SI) {
        load_prep();
#pragma unroll
        for (int i = 0; i < RA + RBL; ++i) load_op(SI, i);
        load_advance();
    };
    auto stage_op = [&](auto SI, const int buf, const int i) {  // operand row i of the staged step: registers -> LDS[buf]
        constexpr int S_ = decltype(SI)::value;
#if BF_ABL & 2
        if (i < RA) asm volatile("" ::"v"(ra[S_][i]));
        else if (i < RA + RBL) asm volatile("" ::"v"(rb[S_][i - RA]));
#else
        if constexpr (X3) {
            if (i < RA) x3_stage(sA + buf * A_BUF + wofs + 64 * i * X3_ROW, BM * X3_ROW, ra[S_][i]);
            else if constexpr (W3) {  // already planes: a copy
                if (i < RA + RBL) *(f32x4*)(sB + buf * B_BUF + (W3_PLANE_PER_REQ ? wofs3[0] + (i - RA) * (BN * X3_ROW) : wofs3[i - RA])) = rb[S_][i - RA];
            } else if (i < RA + RB) x3_stage(sB + buf * B_BUF + wofs + 64 * (i - RA) * X3_ROW, BN * X3_ROW, rb[S_][i - RA]);
        } else {
            if (i < RA) *(f32x4*)&sA[buf * A_BUF + wofs + 64 * i * SK_LDP] = ra[S_][i];
            else if (i < RA + RB) *(f32x4*)&sB[buf * B_BUF + wofs + 64 * (i - RA) * SK_LDP] = rb[S_][i - RA];
        }
#endif
    };
    auto stage = [&](auto SI, const int buf) {
        constexpr int S_ = decltype(SI)::value;
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < RA + RBL; ++i) stage_op(SI, buf, i);
            return;
        }
        float* wA = sA + buf * A_BUF + wofs;
        float* wB = sB + buf * B_BUF + wofs;
#if BF_ABL & 2
#pragma unroll
        for (int i = 0; i < RA; ++i) asm volatile("" ::"v"(ra[S_][i]));
#pragma unroll
        for (int i = 0; i < RB; ++i) asm volatile("" ::"v"(rb[S_][i]));
        (void)wA;
        (void)wB;
#else
#pragma unroll
        for (int i = 0; i < RA; ++i) *(f32x4*)&wA[64 * i * SK_LDP] = ra[S_][i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(f32x4*)&wB[64 * i * SK_LDP] = rb[S_][i];
#endif
    };
    // LDS hand-over between waves WITHOUT draining the global loads in flight (__syncthreads() waits for vmcnt(0) as well)
#define BF_LDS_BARRIER()                      \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    // Tables of a tile, in two halves so that their latency hides under the PREVIOUS tile's K loop:
    //   request(k): the row table entry of row `tid` (one 16-byte load: X offset, tap mask, Y offset, group) and the tile's scalars (live-step
    //               prefix sums, live-tap mask) -- issued when the previous tile's K loop starts, 4 VGPRs + 4 SGPRs in flight;
    //   commit(k):  after that K loop -- the rows' entries go to LDS (sRow / sOut / sGrp[sbuf]), one barrier, then every loader thread picks up
    //               the rows it fetches and the class's tap tables: the loader state is the new tile's, nothing waited for.
    struct TileTables {
        int4 row;
        int tbeg, tend, ti_mask, ti_rot;
    };
    auto request = [&](const Cursor& k) -> TileTables {
        TileTables t;
        const sk_class& cl = P.cls[k.c];
        t.row = tid < BM ? P.rowinfo[cl.row_begin + k.mt * BM + tid] : int4{0, 0, 0, 0};
        t.tbeg = P.tilecum[k.tile];
        t.tend = P.tilecum[k.tile + 1];
        const int2 ti = P.tileinfo[cl.mt_begin + k.mt];
        t.ti_mask = ti.x, t.ti_rot = ti.y;
        return t;
    };
    auto commit = [&](const Cursor& k, const TileTables& t, const int sbuf) -> TileFacts {
        const sk_class& cl = P.cls[k.c];
        TileFacts f;
        if (tid < BM) {
            sRowB[sbuf * BM + tid] = int2{t.row.x, t.row.y};
            sOutB[sbuf * BM + tid] = t.row.z;
            sGrpB[sbuf * BM + tid] = t.row.w;
        }
        if (k.c != cls_loaded) {  // the class's tables (kernel arguments: scalar loads, two vector loads) only when the class changes -- never, for most launches
            cls_loaded = k.c;
            nkc = __builtin_amdgcn_readfirstlane(cl.nkc);
            ntaps = __builtin_amdgcn_readfirstlane(cl.ntaps);
            cls_cout = __builtin_amdgcn_readfirstlane(cl.Cout);
            cls_wrow = __builtin_amdgcn_readfirstlane(cl.Tw * cl.Cin);
            v_ash = lane < SDT_MAX_TAPS ? cl.ashift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
            v_bsh = lane < SDT_MAX_TAPS ? cl.bshift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
        }
        f.Cout = cls_cout;
        f.tbeg = __builtin_amdgcn_readfirstlane(t.tbeg);
        f.tend = __builtin_amdgcn_readfirstlane(t.tend);
        f.a = pos - f.tbeg;
        f.b = min(s_end, f.tend) - f.tbeg;
        f.n0 = k.nt * BN;
        if constexpr (W3_PLANE_PER_REQ) {
            b3_plane_bytes = P.wbytes >> 1;  // plane (N, Tw * Cin) bf16; the planes are wbytes / 2 apart
            bbase[0] = (unsigned)((f.n0 + b3_row(0)) * cls_wrow) * 2u + (unsigned)(tid & 3) * 16u;
        } else if constexpr (W3) {
#pragma unroll
            for (int j = 0; j < RBL; ++j)
                bbase[j] = (unsigned)b3_plane(j) * (P.wbytes >> 1) + (unsigned)((f.n0 + b3_row(j)) * cls_wrow) * 2u + (unsigned)(tid & 3) * 16u;
        } else {
#pragma unroll
            for (int i = 0; i < RB; ++i) bbase[i] = (unsigned)((f.n0 + r0 + 64 * i) * cls_wrow) * (unsigned)sizeof(ET) + (unsigned)kv * 16u;  // W is (N, Tw, Cin)
        }
        rmask = rmask_full = (unsigned)__builtin_amdgcn_readfirstlane(t.ti_mask);
        rot = __builtin_amdgcn_readfirstlane(t.ti_rot);
        if (chunk_major) {  // step a of the tile = (chunk a / live taps, live tap a % live taps)
            const int nlive = __builtin_popcount(rmask_full);
            kc = nlive > 0 ? f.a / nlive : 0;
            for (int skip = f.a - kc * nlive; skip > 0; --skip) rmask &= rmask - 1;
        } else {
            for (int skip = f.a / nkc; skip > 0; --skip) rmask &= rmask - 1;
            kc = f.a - (f.a / nkc) * nkc;
        }
        left = f.b - f.a;
        BF_LDS_BARRIER();  // the rows' entries are in LDS
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int2 ri = sRowB[sbuf * BM + r0 + 64 * i];
            abase[i] = (unsigned)ri.x + (unsigned)kv * 16u;
            inval[i] = (unsigned)ri.y;
        }
        return f;
    };
    typedef std::integral_constant<int, 0> I0;

    f32x16 acc[TM][TN];
    f32x4 a0[X3 ? 1 : TM], b0[X3 ? 1 : TN], a1[X3 ? 1 : TM], b1[X3 ? 1 : TN];
    f32x4 xa[X3 ? 2 : 1][X3 ? 3 : 1][X3 ? TM : 1], xb[X3 ? 2 : 1][X3 ? 3 : 1][X3 ? TN : 1];  // X3: [k-group parity][plane hi / mid / lo][32-row block]
#if BF_ABL & 16
#define BF_READ(A, B, PA, PB, J)                                                                        \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(A[tm]) : "v"(PA));           \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(B[tn]) : "v"(PB))
#else
#define BF_READ(A, B, PA, PB, J)                                                                                      \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = *(const f32x4*)((PA) + tm * 32 * SK_LDP + (J) * 8);     \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) B[tn] = *(const f32x4*)((PB) + tn * 32 * SK_LDP + (J) * 8)
#endif
#define BF_MFMAS(A, B)                                                                                                \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) BF_MFMA(acc[tm][tn], A[tm], B[tn])

    // One K step on LDS buffer `cur`; the fragments of k-group 0 are in a0 / b0 on entry (and on exit, for the next step).
    // The instruction order is FIXED here, slot by slot (a sched_barrier after every slot): one MFMA, then its share of the next k-group's fragment
    // reads, of the LDS stores of step s + 1 and of the requests that re-fill the registers just stored.  Left to itself hipcc chained the four
    // k-groups of one accumulator back to back (dependent MFMAs) with an `s_waitcnt lgkmcnt(0)` in front of each: nothing overlapped, and the
    // ablation was additive -- without MFMAs the loop got shorter by exactly the MFMAs' execution time (profiles/r05_bf2_ablation_v0.txt).
    constexpr int NL = RA + RBL;
    constexpr int RPM = (NF + NM - 1) / NM;  // fragment reads per MFMA slot
    // The L1 accepts 64 B per clock: the 48 (BN 128) / 64 / 40 KB that a workgroup requests per K step keep it busy for 768 / 1024 / 640 cycles, and
    // a wave whose request is not accepted yet issues nothing else (in order) -- with all requests in the first half of the step the eight waves
    // queued up behind each other and the step cost MFMA time PLUS request time (2105 cycles for 1024 of MFMA; without the requests 700 fewer).
    // So the requests are spread over ALL slots of the step (the last ones behind the barrier: a request needs no barrier) and the LDS stores over
    // the slots in front of the barrier; store i always precedes request i, which re-fills the registers store i has read.
    // (Measured and rejected, profiles/r05_bf2_experiments.txt: a "ping-pong" step in which the waves w and w + 4 run the MFMA half and the store /
    // request half of a step in opposite order -- 2094 -> 3247 cycles per step: clustered requests wait for each other.)
    auto stage_slot = [](const int i) constexpr { return i * 3 * NM / NL; };
    auto load_slot = [](const int i) constexpr { return i * 4 * NM / NL; };
    auto rd = [&](auto& A, auto& B, const float* pa, const float* pb, const int J, const int i) {
#if BF_ABL & 16
        if (i < TM) asm volatile("" : "+v"(A[i]) : "v"(pa));
        else if (i < NF) asm volatile("" : "+v"(B[i - TM]) : "v"(pb));
#else
        if (i < TM) A[i] = *(const f32x4*)(pa + i * 32 * SK_LDP + J * 8);
        else if (i < NF) B[i - TM] = *(const f32x4*)(pb + (i - TM) * 32 * SK_LDP + J * 8);
#endif
    };
#ifdef SDT_TUNING
    long long bar_wait = 0;  // clock64 ticks this wave spent in the K steps' barriers since the last stamp (slot 7 of the timeline: summed over the 8 waves)
#endif
    // ---- X3: fragment i of a k-group's 3 NF fragments, in the order the MFMAs want them: A lo, B hi, A hi, B lo, A mid, B mid
    auto x3_rd = [&](const int set, const float* pa, const float* pb, const int fk, const int i) {
        if constexpr (X3) {
            // segments of TM / TN fragments: (A, lo) (B, hi) (A, hi) (B, lo) (A, mid) (B, mid); i is a constant after unrolling
            constexpr int PL[6] = {2, 0, 0, 2, 1, 1};
            const int pair = i / NF, in_pair = i - pair * NF;
            if (pair >= 3) return;
            const int seg = 2 * pair + (in_pair >= TM ? 1 : 0), k = in_pair >= TM ? in_pair - TM : in_pair;
            const int pl = PL[seg];
#if BF_ABL & 16
            if (!(seg & 1)) asm volatile("" : "+v"(xa[set][pl][k]) : "v"(pa));
            else asm volatile("" : "+v"(xb[set][pl][k]) : "v"(pb));
#else
            if (!(seg & 1)) xa[set][pl][k] = *(const f32x4*)(pa + pl * BM * X3_ROW + k * 32 * X3_ROW + fk);
            else xb[set][pl][k] = *(const f32x4*)(pb + pl * BN * X3_ROW + k * 32 * X3_ROW + fk);
#endif
        }
    };
    // One X3 K step (32 channels = two k-groups of 6 NM MFMAs) on LDS buffer `cur`; the fragments of k-group 0 are in set 0 on entry and on exit.
    // Two scheduling regions around the step's one barrier (after MFMA 9 NM of 12 NM):
    //   region 1: the other k-group's 3 NF fragment reads, the split + LDS stores of step s + 1 (NL x (22 VALU + 3 ds_write_b64)), the requests of
    //             step s + 2 -- spread over the 9 NM MFMAs by sched_group_barrier: the split's VALU chains must run UNDER the MFMAs (a whole
    //             operand's 25 instructions between two MFMAs left the matrix pipe idle for 70 of every 100 cycles: 40-50 % of the peak in the loop)
    //   region 2: the NEXT step's k-group-0 fragments (LDS[next] is complete behind the barrier) under the last 3 NM MFMAs.
    auto step3 = [&](auto CUR, auto SETN) __attribute__((always_inline)) {
        if constexpr (X3) {
            constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // small products first
            constexpr int NS = 12 * NM, NR = 3 * NF, BAR = 9 * NM;
            const float* pa = sA + cur * A_BUF + fa;
            const float* pb = sB + cur * B_BUF + fb;
            const float* pan = sA + nx * A_BUF + fa;
            const float* pbn = sB + nx * B_BUF + fb;
            load_prep();
            __builtin_amdgcn_sched_barrier(0);
            // ---- region 1
#pragma unroll
            for (int i = 0; i < NR; ++i) x3_rd(1, pa, pb, fk1, i);
            // BF_ABL 128 / 256 (wrong results): the A operand is requested, split and stored on every other step only / never -- the ceiling of a
            // loader that fetches an im2col row once per (dy, channel chunk) instead of once per tap (sliding window, kw = 2 / kw -> infinity)
            constexpr bool skipA = !W3 && (((BF_ABL & 128) && cur == 1) || (BF_ABL & 256));
#pragma unroll
            for (int j = skipA ? RA : 0; j < NL; ++j) {
                stage_op(SETN, nx, j);
                load_op(SETN, j);
            }
#pragma unroll
            for (int q = 0; q < BAR; ++q) {
                const int J = q / (6 * NM), pr = (q % (6 * NM)) / NM, t = q % NM;
                BF_MFMA(acc[t / TN][t % TN], xa[J][PA[pr]][t / TN], xb[J][PB[pr]][t % TN]);
            }
            x3_pattern<0, BAR, NR, W3 ? RA : (skipA ? RB : NL), W3 ? RBL : 0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of LDS[cur] and my stores to LDS[next] are done
#ifdef SDT_TUNING
            const long long tb0 = bf2_dbg_tl != nullptr ? clock64() : 0;  // (timeline runs only: how long does a wave wait at the step's barrier?)
#endif
#if !(BF_ABL & 8)
            __builtin_amdgcn_s_barrier();
#endif
#ifdef SDT_TUNING
            if (bf2_dbg_tl != nullptr) bar_wait += clock64() - tb0;
#endif
            __builtin_amdgcn_sched_barrier(0);
            // ---- region 2
#pragma unroll
            for (int i = 0; i < NR; ++i) x3_rd(0, pan, pbn, fk0, i);
#pragma unroll
            for (int q = BAR; q < NS; ++q) {
                const int J = q / (6 * NM), pr = (q % (6 * NM)) / NM, t = q % NM;
                BF_MFMA(acc[t / TN][t % TN], xa[J][PA[pr]][t / TN], xb[J][PB[pr]][t % TN]);
            }
            x3_pattern2<0, NS - BAR, NR>();
            __builtin_amdgcn_sched_barrier(0);
            load_advance();
        }
    };

    auto step = [&](auto CUR, auto SETI) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
        typedef decltype(SETI) SETN;
        if constexpr (X3) {
            step3(CUR, SETI);
            return;
        }
        const float* pa = sA + cur * A_BUF + fa;
        const float* pb = sB + cur * B_BUF + fb;
        const float* pan = sA + nx * A_BUF + fa;
        const float* pbn = sB + nx * B_BUF + fb;
        load_prep();
        __builtin_amdgcn_sched_barrier(0);
        auto ops = [&](const int q) {  // the stores / requests that belong to slot q of the step
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                if (stage_slot(j) == q) stage_op(SETN{}, nx, j);
                if (load_slot(j) == q) load_op(SETN{}, j);
            }
        };
        // k-group 0 (a0 / b0) + fragments of k-group 1 -> a1 / b1
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            BF_MFMA(acc[i / TN][i % TN], a0[i / TN], b0[i % TN]);
#pragma unroll
            for (int j = 0; j < RPM; ++j) rd(a1, b1, pa, pb, 1, i * RPM + j);
            ops(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-group 1 (a1 / b1) + fragments of k-group 2 -> a0 / b0
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            BF_MFMA(acc[i / TN][i % TN], a1[i / TN], b1[i % TN]);
#pragma unroll
            for (int j = 0; j < RPM; ++j) rd(a0, b0, pa, pb, 2, i * RPM + j);
            ops(NM + i);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-group 2 (a0 / b0) + fragments of k-group 3 -> a1 / b1
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            BF_MFMA(acc[i / TN][i % TN], a0[i / TN], b0[i % TN]);
#pragma unroll
            for (int j = 0; j < RPM; ++j) rd(a1, b1, pa, pb, 3, i * RPM + j);
            ops(2 * NM + i);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of LDS[cur] and my writes of LDS[next] are done
#if !(BF_ABL & 8)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // k-group 3 (a1 / b1) + fragments of the NEXT step's k-group 0 -> a0 / b0 + the last requests
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            BF_MFMA(acc[i / TN][i % TN], a1[i / TN], b1[i % TN]);
#pragma unroll
            for (int j = 0; j < RPM; ++j) rd(a0, b0, pan, pbn, 0, i * RPM + j);
            ops(3 * NM + i);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_advance();
    };

    // `n` (<= UNR) consecutive steps starting at period position U (compile-time recursion: buffer parity and staging set are template constants)
    auto steps_unrolled = [&](const int n) __attribute__((always_inline)) {
        auto rec = [&](auto&& rec_, auto U_) __attribute__((always_inline)) -> void {
            constexpr int u = decltype(U_)::value;
            if constexpr (u < UNR) {
                if (u < n) {
                    step(std::integral_constant<int, (u & 1)>{}, std::integral_constant<int, ((u + 1) % NSET)>{});
                    rec_(rec_, std::integral_constant<int, u + 1>{});
                }
            }
        };
        rec(rec, std::integral_constant<int, 0>{});
    };

    f32x16 tot[X3 ? TM : 1][X3 ? TN : 1];  // X3: the sum of the finished accumulation chunks
    auto x3_flush = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    tot[X3 ? i : 0][X3 ? j : 0][q] += acc[i][j][q];
                    acc[i][j][q] = 0.f;
                }
    };
    auto x3_flush_final = [&]() {  // acc = the tile's (partial) sum, where the end phase expects it
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] += tot[X3 ? i : 0][X3 ? j : 0][q];
    };

    // running column statistics of this workgroup's range (threads tid < BN: column n0 + tid of group run_g)
    int run_g = -1, run_n0 = 0, run_cout = 0;
    float run_s = 0.f, run_q = 0.f;
    auto flush_run = [&]() {
        if constexpr (EPI != 0) {
            if (run_g >= 0) {
                double* d = (EPI == 1 ? stats : nb.sums) + ((size_t)run_g * run_cout + run_n0 + tid) * 2;
                atomicAdd(d, (double)run_s);
                atomicAdd(d + 1, (double)run_q);
            }
        }
    };

    // ------------------------------------------------------------------ first tile: tables, loader state, the first request
    int sbuf = 0;
    TileFacts F = commit(cur, request(cur), 0);
    load(I0{});
    for (int seg = 0;; ++seg) {
        BF_TL(0, wall_clock64());
        // ---- the NEXT tile's tables are requested now: they arrive while this tile's K loop runs
        const bool more = F.tbeg + F.b < s_end;  // this range goes on behind this tile (the tile ends inside the range)
        const Cursor nxt = more ? advance(cur) : cur;
        TileTables NT;
        if (more) NT = request(nxt);
        // ---- pipeline fill: step 0 -> LDS[0]; steps 1 (and 2) -> registers
        BF_LDS_BARRIER();  // the previous tile's LDS tiles are no longer read by anybody
        stage(I0{}, 0);
        // steps 1 .. NSET - 1 -> sets 1 .. NSET - 1, step NSET -> set 0 (just stored)
        if constexpr (NSET >= 2) load(std::integral_constant<int, 1>{});
        if constexpr (NSET >= 3) load(std::integral_constant<int, 2>{});
        if constexpr (NSET >= 4) load(std::integral_constant<int, 3>{});
        load(I0{});
        BF_LDS_BARRIER();  // LDS[0] is written (the requests just issued stay in flight)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[i][j][q] = 0.f;
                    if constexpr (X3) tot[i][j][q] = 0.f;
                }
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < 3 * NF; ++i) x3_rd(0, sA + fa, sB + fb, fk0, i);
        } else {
            BF_READ(a0, b0, sA + fa, sB + fb, 0);
        }
        const int nsteps = F.b - F.a;
        BF_TL(1, wall_clock64());
        BF_TL(5, (unsigned long long)nsteps);
        int s = 0;
        // the K loop, unrolled over the period of (LDS buffer, staging set): step u of a period reads LDS[u & 1] and stores / re-requests set (u + 1) % NSET
        if constexpr (X3) {
            // fp32-grade sums: the accumulators are added to `tot` and cleared every SK_CHUNK steps (256 channels x taps), as the fp32 kernel does --
            // a running fp32 sum over a whole 4096-term K loop carries 1.5x the rounding error of the blocked one (tools/debug/x3_check.py)
            constexpr int CH = SK_CHUNK % UNR == 0 ? SK_CHUNK : UNR;  // chunks are whole periods (three staging sets: 6 steps)
            for (; s + CH <= nsteps; s += CH) {
#pragma unroll 1
                for (int u = 0; u < CH; u += UNR) steps_unrolled(UNR);
                x3_flush();
            }
        }
        for (; s + UNR <= nsteps; s += UNR) {
            steps_unrolled(UNR);
        }
        steps_unrolled(nsteps - s);
        if constexpr (X3) x3_flush_final();
        BF_TL(2, wall_clock64());
#ifdef SDT_TUNING
        if (lane == 0 && bf2_dbg_tl != nullptr && seg < 16) atomicAdd(&bf2_dbg_tl[((size_t)r * 16 + seg) * 8 + 7], (unsigned long long)bar_wait);
        bar_wait = 0;
#endif
#if BF_ABL & 64
#pragma unroll
        for (int i = 0; i < RA + RBL; ++i) asm volatile("" ::"v"(abl_sink[i]));
#endif

        // ---- the tile's facts for the end phase; then the NEXT tile's loader state and first request, before the end phase
        const TileFacts E = F;
        const int ebuf = sbuf;
        const bool owner = E.a == 0, whole = owner && E.b == E.tend - E.tbeg;
        if (more) {
            pos = E.tbeg + E.b;
            cur = nxt;
            sbuf ^= 1;
            F = commit(cur, NT, sbuf);  // (writes sRow / sOut / sGrp[sbuf]: the other buffer than the one the end phase below reads)
            load(I0{});                 // first K step of the next tile: lands under the end phase
        }
        BF_TL(3, wall_clock64());
        BF_TL(6, whole ? 0ull : (owner ? 1ull : 2ull));
        const int* sOut = sOutB + ebuf * BM;
        const int* sGrp = sGrpB + ebuf * BM;
        if (!owner) {
            // publish the partial tile: register order, 16 B per lane (write-through), then drain + flag
            float* slab = P.slabs + (size_t)r * (BM * BN);
            const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        const int off = ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 16;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsS, off, 0, 16);  // sc1
                    }
            // (the next tile's first loads are in flight too: vmcnt(0) waits for them as well -- once per split tile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !BF_MUTED(r)) __hip_atomic_store((gu32*)(P.flags + r), P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (!whole) {
                const int r_last = (int)((((long)E.tend) * G - 1) / P.S);
                bool lost = false;
                for (int cr = r + 1; cr <= r_last; ++cr) {
                    if (wave == 0) {
                        unsigned spins = 0;
                        bool ok = true;
                        while (__hip_atomic_load((gu32*)(P.flags + cr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.epoch) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > P.spin_limit) {
                                ok = false;
                                break;
                            }
                        }
                        if (lane == 0) {
                            sFlagOk = ok ? 1 : 0;
                            if (!ok) __hip_atomic_store((gu32*)P.err, 1u + (unsigned)cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else __hip_atomic_store((gu32*)(P.flags + cr), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    if (sFlagOk) {
                        const float* slab = P.slabs + (size_t)cr * (BM * BN);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    // L1-bypassing loads (sc1): the slab was published write-through by another CU; this CU may hold stale
                                    // lines of an earlier launch's slab at the same address
                                    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, BM * BN * 4, 0x00020000);
                                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                                  rsS, ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 16, 0, 16));
#pragma unroll
                                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
                                }
                    } else {
                        lost = true;
                    }
                    __syncthreads();
                }
                if (lost) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 16; ++q) acc[i][j][q] = __builtin_nanf("");
                }
            }
            // statistics of a tile whose 256 rows are ONE group (nearly all tiles): per-wave column sums -> LDS -> one partial per column, kept in
            // registers across the tiles of this workgroup's range while group and column block stay the same (flushed with one fp64 atomic pair
            // per column when they change and at the end of the range)
            const int tgrp = sGrp[0];
            const bool tile_uniform = EPI != 0 && TM % 2 == 0 && tgrp >= 0 && tgrp == sGrp[BM - 1];
            float* red = tile_uniform ? sA + A_BUF + (wm * BN + col0) * 2 : nullptr;  // (LDS buffer 1 of A is idle until the next tile's first step)
            if constexpr (EPI == 2 && TM > 2) {  // the reads of y for half of the rows at a time: 64 instead of 128 registers
                bf2_epilogue<ET, TM, TN, EPI, 0, TM / 2>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes, red);
                bf2_epilogue<ET, TM, TN, EPI, TM / 2, TM>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes, red);
            } else {
                bf2_epilogue<ET, TM, TN, EPI, 0, TM>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes, red);
            }
            if constexpr (EPI != 0) {
                if (tile_uniform) {
                    BF_LDS_BARRIER();
                    if (tid < BN) {
                        const float* rr = sA + A_BUF + tid * 2;
                        float ts = 0.f, tq = 0.f;
#pragma unroll
                        for (int w = 0; w < WGM; ++w) {  // fixed order
                            ts += rr[w * BN * 2];
                            tq += rr[w * BN * 2 + 1];
                        }
                        if (run_g == tgrp && run_n0 == E.n0) {
                            run_s += ts;
                            run_q += tq;
                        } else {
                            flush_run();
                            run_g = tgrp, run_n0 = E.n0, run_cout = E.Cout, run_s = ts, run_q = tq;
                        }
                    }
                }
            }
        }
        BF_TL(4, wall_clock64());
        if (!more) break;
    }
    if (tid < BN) flush_run();
#undef BF_LDS_BARRIER
#undef BF_READ
#undef BF_MFMAS
#undef sFlagOk
}

template <typename ET, int BM, int BN, int WGM, int WGN, int EPI, bool W3 = false>
static void bf2_launch_one(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, hipStream_t s) {
    // A / B tiles (double-buffered; X3: three planes of 64 B per row), sOut + sGrp + sRow (double-buffered), flag
    const size_t tiles = sizeof(ET) == 4 ? (size_t)(2 * 3 * (BM + BN) * X3_ROW) * 4 : (size_t)(2 * (BM + BN) * SK_LDP) * 4;
    const size_t lds = tiles + (size_t)8 * BM * 4 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)convbf2_kernel<ET, BM, BN, WGM, WGN, EPI, W3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((convbf2_kernel<ET, BM, BN, WGM, WGN, EPI, W3>), dim3(A.G), dim3(BF_NT), lds, s, (const ET*)x, (const ET*)w, bias, (ET*)y, A, stats, nb);
}

#define BF2_GO(ET_, BM_, BN_, WGM_, WGN_)                                                              \
    do {                                                                                                \
        if (epi == 0) bf2_launch_one<ET_, BM_, BN_, WGM_, WGN_, 0>(x, w, bias, y, A, stats, nb, s);     \
        else if (epi == 1) bf2_launch_one<ET_, BM_, BN_, WGM_, WGN_, 1>(x, w, bias, y, A, stats, nb, s); \
        else bf2_launch_one<ET_, BM_, BN_, WGM_, WGN_, 2>(x, w, bias, y, A, stats, nb, s);              \
    } while (0)

int convbf2_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, int bm, int bn, int epi,
                   hipStream_t s) {
    if (bm == 256 && bn == 256) BF2_GO(__bf16, 256, 256, 2, 4);
    else if (bm == 256 && bn == 128) BF2_GO(__bf16, 256, 128, 4, 2);
    else if (bm == 256 && bn == 64) BF2_GO(__bf16, 256, 64, 4, 2);
    else if (bm == 128 && bn == 128) BF2_GO(__bf16, 128, 128, 2, 4);  // few-row layers (L5 - L7 at 32 clips): 64 x 32 per wave, about one tile per CU
    else return SDT_ERR_ARG;
    return SDT_OK;
}

// fp32 tensors, split-fp32 products (plans built with sdt_convsk_set_f32_split(1)): 128 x 128 tiles, 256 x 64 for the 64-channel outputs -- 64 x 32
// per wave: two accumulator sets (running chunk + chunk sums) and two fragment sets of three planes fit 256 registers; 64 x 64 per wave does not
#define BF2_GO_W3(BM_, BN_, WGM_, WGN_)                                                                     \
    do {                                                                                                     \
        if (epi == 0) bf2_launch_one<float, BM_, BN_, WGM_, WGN_, 0, true>(x, w, bias, y, A, stats, nb, s);      \
        else if (epi == 1) bf2_launch_one<float, BM_, BN_, WGM_, WGN_, 1, true>(x, w, bias, y, A, stats, nb, s); \
        else bf2_launch_one<float, BM_, BN_, WGM_, WGN_, 2, true>(x, w, bias, y, A, stats, nb, s);               \
    } while (0)
int convx3_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, int bm, int bn, int epi,
                  int w3, hipStream_t s) {
    if (w3) {
        if (bm == 256 && bn == 64) BF2_GO_W3(256, 64, 4, 2);
        else if (bm == 128 && bn == 128) BF2_GO_W3(128, 128, 2, 4);
        else return SDT_ERR_ARG;
        return SDT_OK;
    }
    if (bm == 256 && bn == 64) BF2_GO(float, 256, 64, 4, 2);
    else if (bm == 128 && bn == 128) BF2_GO(float, 128, 128, 2, 4);
    else return SDT_ERR_ARG;
    return SDT_OK;
}
#undef BF2_GO
