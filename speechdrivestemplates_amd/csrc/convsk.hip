// Persistent stream-K implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) -- the Conv2d forward and
// input-gradient launches of the audio encoder (building_blocks.py:15-22; generator.py:15-30), round 3.
//
// Why a second conv kernel.  profiles/r03_mfma_pipe_microbench.txt / r03_taps_timeline_64x64.txt: the 64x64-tile kernel of conv.hip keeps
// 7 workgroups per CU resident, but under load a workgroup spends 30 us of its 120 us life in a prologue that takes 6 us alone (and 9 us in
// the epilogue): only ~3.5 of the 7 waves of a SIMD are in their K loop, the matrix pipe is 73-80 % busy, and layers with 1060 tiles put
// 5 workgroups on 36 CUs and 4 on the rest (17 % tail).  The same K loop WITHOUT prologue / epilogue / quantisation sustains 0.95 of the
// peak -- and one wave per SIMD on a 128x128 tile with the feeding instructions interleaved between its own MFMAs sustains 0.91.
//
// Design.
//   * grid = one workgroup (4 waves, 2x2) per CU slot, G = 256 * R workgroups, R = 2 (R = 1 instantiated for A/B).  The launch's work is the list of (tile, live K step)
//     pairs in tile order; workgroup r owns the contiguous range [r*S/G, (r+1)*S/G) of that list (stream-K): every CU gets the same
//     number of K steps whatever the tile count, tap culling included (the host counts LIVE steps per tile).
//   * a tile that straddles a range boundary is finished by the workgroup that owns its first step: the others store their partial
//     accumulators write-through into a slab, drain, and raise a flag; the owner -- which reaches that tile at the END of its range, when
//     the others (who meet it at the START of theirs) are long done -- polls the flags, adds the slabs in range order and runs the
//     epilogue.  Fixed split points, fixed order: bit-identical from run to run.  (MI355X_MICROARCH.md, Guideline 16 recipe R1.)
//   * tile 128x128x32 (wave tile 64x64 = 2x2 accumulators) or 128x64x32 (N = 64 layers; wave tile 64x32); two LDS buffers, ONE barrier
//     per K step; tile s+1 goes registers -> LDS[next] and the global loads of tile s+2 are issued BETWEEN the MFMAs of tile s
//     (sched_group_barrier), so the in-order wave overlaps its own feeding with its own MFMAs.
//   * no index arithmetic in the kernel: the host builds a PLAN per geometry pack (sdt_convsk_plan_*): per GEMM row {byte offset of the
//     row's (0,0) tap in X, mask of the taps that fall outside X for this row, byte offset of the output row in Y}, per m-tile {live-tap mask, first tap of the
//     residue-rotated order}, prefix sums of live steps per tile, first tile of every range.
//   * fp32 accumulation in CHUNKS of 8 K steps (256 products): a chunk starts from C = 0 and is added to the running total when it
//     ends.  A K-long single accumulator made the HIP forward 2-3.8x further from float64 than the blocked sums of the CPU reference
//     (profiles/r03_stage_errors_before.txt: the error grew with sqrt(K)); chunking brings it level.
#include <type_traits>

#include <vector>

#include "convsk.h"

#if defined(SK_BF_ABL) && (SK_BF_ABL & 4)  // ablation (wrong results): no MFMAs in the bf16 K loop (the fragments stay live)
#define SK_BF_MFMA(C, A, B) asm volatile("" : "+v"(C) : "v"(A), "v"(B))
#else
#define SK_BF_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, A), __builtin_bit_cast(sk_bf16x8, B), C, 0, 0, 0)
#endif

#ifdef SDT_TUNING
// tools/debug/sk_timeline.py: lane 0 of every workgroup stamps the 100 MHz real-time counter at five points of each of its first 16
// tile segments: sk_dbg_tl[(range * 16 + segment) * 8 + slot], slot 5 = K steps of the segment, 6 = kind (0 whole, 1 owner, 2 publish)
__device__ unsigned long long* sk_dbg_tl = nullptr;
extern "C" int sdt_debug_set_timeline_sk(void* p) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(sk_dbg_tl), &p, sizeof(p));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define SK_TL(slot, val)                                                                              \
    do {                                                                                              \
        if (threadIdx.x == 0 && sk_dbg_tl != nullptr && seg < 16) sk_dbg_tl[((size_t)r * 16 + seg) * 8 + (slot)] = (val); \
    } while (0)
// fault injection (tests/test_ops_gpu.py::test_streamk_lost_partner_is_loud): the workgroup of this range computes its partial tile but never
// raises its flag -- what a partner that was never dispatched looks like to the owner of the tile
__device__ int sk_dbg_mute_range = -1;
extern "C" int sdt_debug_convsk_mute_range(int r) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(sk_dbg_mute_range), &r, sizeof(r));
    if (e == hipSuccess && convbf2_debug_mute_range(r) != SDT_OK) return SDT_ERR_LAUNCH;  // the 8-wave kernels (convbf.hip) as well
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define SK_MUTED(r) ((r) == sk_dbg_mute_range)
#else
#define SK_MUTED(r) false
#define SK_TL(slot, val) do { } while (0)
#endif

// Epilogue of a finished output tile held in TM x TN accumulators per wave (2x2 wave grid): branch-free buffer stores (+ bias) and,
// for EPI 1 / 2, the per-(group, channel) statistics.  sOut / sGrp: LDS, [BM] byte offset of each tile row in Y (SK_OOB: none) and its
// statistics group.  C/D layout of a 32x32 accumulator: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5).
template <typename TX, typename TY, int BM, int BN, int EPI, int TMB = 0, int TME = BM / 64>
__device__ __forceinline__ void sk_epilogue(f32x16 (&tot)[BM / 64][BN / 64], const int* sOut, const int* sGrp, const int n0, const int wm,
                                            const int wn, const int lane, const float* __restrict__ bias, const __amdgpu_buffer_rsrc_t rsY,
                                            const int Cout, double* __restrict__ stats, const sk_norm_bwd& nb, const unsigned ybytes) {
    constexpr int TN = BN / 64;  // accumulator rows TMB .. TME-1 of the wave (the callers interleave other work between halves)
    // EPI 2 reads the forward output y at every position of the tile (64 dwords per lane, HBM misses): all of them are issued before
    // anything else of the epilogue -- one exposed memory latency per tile instead of one per 32x32 block (the K loop's fragment and
    // staging registers are dead here, so the 64 values have room)
    float yv[(TME - TMB) * TN * 16];
    __amdgpu_buffer_rsrc_t rsNY = rsY;
    if constexpr (EPI == 2) {
        rsNY = __builtin_amdgcn_make_buffer_rsrc((void*)nb.y, 0, (int)ybytes, 0x00020000);
#pragma unroll
        for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                // (sOut holds byte offsets of the output rows in Y; y has Y's shape, and -- host-checked -- its element size: TX == TY with EPI 2)
                const unsigned nb4 = (unsigned)(n0 + wn * (BN / 2) + tn * 32 + (lane & 31)) * (unsigned)sizeof(TX);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int4 o4 = *(const int4*)&sOut[wm * (BM / 2) + tm * 32 + 8 * qq + 4 * (lane >> 5)];
                    const int offs[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#ifdef SK_NO_YLOAD  // ablation (wrong results): cost of the epilogue's reads of the forward output
                        yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e] = (float)offs[e];
#else
                        if constexpr (sk_is_bf16<TX>::value)
                            yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e] =
                                sk_bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(rsNY, (int)((unsigned)offs[e] + nb4), 0, 0));
                        else
                            yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e] =
                                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsNY, (int)((unsigned)offs[e] + nb4), 0, 0));
#endif
                }
            }
    }
#pragma unroll
    for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * (BN / 2) + tn * 32 + (lane & 31);
            const float bv = bias != nullptr ? bias[n] : 0.f;
            // branch-free stores: the four rows (q & 3) of a register quad are consecutive tile rows -> one 16-byte LDS read
            // gives their byte offsets; rows past the end of the tensor carry SK_OOB and the hardware drops the store
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int4 o4 = *(const int4*)&sOut[wm * (BM / 2) + tm * 32 + 8 * qq + 4 * (lane >> 5)];
                if constexpr (sk_is_bf16<TY>::value) {
                    // bf16 output: a lane holds ONE column of four consecutive rows; lanes l and l ^ 1 hold neighbouring columns.  Each pair of
                    // lanes swaps one value per row pair (DPP quad_perm [1,0,3,2]) so that the even lane stores the column pair of the first
                    // row and the odd lane that of the second as ONE dword: half the stores, each 4 bytes instead of 2
                    const bool odd = lane & 1;
                    const unsigned nb2 = (unsigned)(n & ~1) * 2u;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float v0 = tot[tm][tn][4 * qq + 2 * h] + bv, v1 = tot[tm][tn][4 * qq + 2 * h + 1] + bv;
                        const float give = odd ? v0 : v1;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
                        sk_bf16x2 pk;
                        pk[0] = (__bf16)(odd ? got : v0);
                        pk[1] = (__bf16)(odd ? v1 : got);
                        const int ro = odd ? (h ? o4.w : o4.y) : (h ? o4.z : o4.x);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rsY, (int)((unsigned)ro + nb2), 0, 0);
                    }
                } else {
                    const unsigned nb4 = (unsigned)n * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[tm][tn][4 * qq + 0] + bv), rsY, (int)((unsigned)o4.x + nb4), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[tm][tn][4 * qq + 1] + bv), rsY, (int)((unsigned)o4.y + nb4), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[tm][tn][4 * qq + 2] + bv), rsY, (int)((unsigned)o4.z + nb4), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[tm][tn][4 * qq + 3] + bv), rsY, (int)((unsigned)o4.w + nb4), 0, 0);
                }
            }
            if constexpr (EPI == 1 || EPI == 2) {
                // statistics over the rows of this 32-row block: its rows belong to at most two groups (groups ascend with the
                // row; host-checked: a group spans >= 32 rows).  EPI 1: sum / sum of squares of the conv output (forward
                // normalisation statistics); EPI 2: sum gg / sum gg*yhat of the normalisation backward that consumes dX
                const int rb0 = wm * (BM / 2) + tm * 32;
                const int gfirst = sGrp[rb0], glast = sGrp[rb0 + 31];
                const bool two = glast != gfirst && glast >= 0;
                float mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f, ga = 1.f, be = 0.f;
                if constexpr (EPI == 2) {
                    if (gfirst >= 0) {
                        mu0 = nb.mean[(size_t)gfirst * Cout + n];
                        rs0 = nb.rstd[(size_t)gfirst * Cout + n];
                    }
                    if (two) {
                        mu1 = nb.mean[(size_t)glast * Cout + n];
                        rs1 = nb.rstd[(size_t)glast * Cout + n];
                    }
                    if (nb.gamma != nullptr) ga = nb.gamma[n];
                    if (nb.beta != nullptr) be = nb.beta[n];
                }
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
                // Fast path (wave-uniform): the 32 rows of the block are one group and all exist (groups ascend with the row, rows past the end of
                // the tensor carry group -1 and come last) -- nearly every block.  Same sums in the same order as the general loop below, without
                // its per-element selects: the statistics were VALU-bound (3.3 of a tile's 4.8 us end phase, profiles/r05_bf2_timeline_v0.txt).
                if (gfirst == glast && gfirst >= 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if constexpr (EPI == 1) {
                            const float u = tot[tm][tn][q] + bv;
                            s0 += u;
                            q0 = fmaf(u, u, q0);
                        } else {
                            const float v = (yv[((tm - TMB) * TN + tn) * 16 + q] - mu0) * rs0;
                            const float u = tot[tm][tn][q] * act_grad(v * ga + be, nb.slope);
                            s0 += u;
                            q0 = fmaf(u, v, q0);
                        }
                    }
                } else
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int4 o4 = *(const int4*)&sOut[rb0 + 8 * qq + 4 * (lane >> 5)];
                    const int4 g4 = *(const int4*)&sGrp[rb0 + 8 * qq + 4 * (lane >> 5)];
                    const int offs[4] = {o4.x, o4.y, o4.z, o4.w}, grps[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool valid = offs[e] >= 0, second = grps[e] != gfirst;
                        float u, v;  // accumulate (u, u*v') pairs: EPI 1 (v, v*v), EPI 2 (gg, gg*yhat)
                        if constexpr (EPI == 1) {
                            u = valid ? tot[tm][tn][4 * qq + e] + bv : 0.f;
                            v = u;
                        } else {
                            const float yvv = yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e];
                            v = (yvv - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
                            u = valid ? tot[tm][tn][4 * qq + e] * act_grad(v * ga + be, nb.slope) : 0.f;
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
                s0 += __shfl_xor(s0, 32, 64);  // the two lane halves hold different rows of the same column
                q0 += __shfl_xor(q0, 32, 64);
                s1 += __shfl_xor(s1, 32, 64);
                q1 += __shfl_xor(q1, 32, 64);
                double* acc_out = EPI == 1 ? stats : nb.sums;
#ifdef SK_NO_ATOM  // ablation (wrong results): cost of the statistics atomics
                if (lane < 32 && gfirst >= 0 && s0 == 123.456f) {
#else
                if (lane < 32 && gfirst >= 0) {
#endif
                    double* d = acc_out + ((size_t)gfirst * Cout + n) * 2;
                    atomicAdd(d, (double)s0);
                    atomicAdd(d + 1, (double)q0);
                    if (two) {
                        double* d1 = acc_out + ((size_t)glast * Cout + n) * 2;
                        atomicAdd(d1, (double)s1);
                        atomicAdd(d1 + 1, (double)q1);
                    }
                }
            }
        }
}

// EPI: 0 = store (+ bias), 1 = + forward statistics (stats: fp64 atomics, zero on entry), 2 = + normalisation-backward statistics
template <typename TX, typename TY, int BM, int BN, int EPI, int WPC>
__global__ __launch_bounds__(256, WPC) void convsk_kernel(const TX* __restrict__ X, const TX* __restrict__ W, const float* __restrict__ bias,
                                                     TY* __restrict__ Y, const sk_args P, double* __restrict__ stats,
                                                     const int rows_per_group, const sk_norm_bwd nb) {
    constexpr bool BF = sk_is_bf16<TX>::value;
    constexpr int TM = BM / 64, TN = BN / 64, RA = BM / 32, RB = BN / 32, NM = TM * TN * (BF ? 1 : 4);  // MFMAs per k-group of a step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                          // [2][BM * LDP]
    float* sB = smem + 2 * BM * SK_LDP;        // [2][BN * LDP]
    int* sOut = (int*)(sB + 2 * BN * SK_LDP);  // [BM] byte offset of the tile's output rows in Y (SK_OOB, negative as int: none)
    int* sGrp = sOut + BM;                     // [BM] statistics group of each row (EPI 1 / 2)
    int* sFlagOkp = sGrp + BM;                 // [4] (no static __shared__: it would shift the 16-byte alignment of the dynamic region)
#define sFlagOk (sFlagOkp[0])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kv = tid & 7, r0 = tid >> 3;
    const int G = P.G;
    // workgroup -> range: blocks that land on one XCD (id % 8) own consecutive ranges, i.e. consecutive tiles (shared L2)
    const int bid = blockIdx.x;
    const int r = (bid & 7) * (G >> 3) + (bid >> 3);  // G % 8 == 0
    const int s_begin = (int)((long)r * P.S / G), s_end = (int)((long)(r + 1) * P.S / G);
    if (s_end <= s_begin) return;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)P.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)P.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)P.ybytes, 0x00020000);

    // fragment read offsets (floats) inside an LDS tile
    const int fa = (wm * (BM / 2) + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int fb = (wn * (BN / 2) + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int wofs = r0 * SK_LDP + kv * 4;  // loader thread's position in an LDS tile (+ 32 * i rows)

    // ---- the walk over the range's tiles: (tile, class c, m-tile mt, n-tile nt), `pos` = next step of the launch's list to do
    int tile = __builtin_amdgcn_readfirstlane(P.range_tile[r]);
    int pos = s_begin;
    int c = 0;
    while (c + 1 < P.ncls && tile >= P.cls[c + 1].tile_begin) ++c;
    int mt = (tile - P.cls[c].tile_begin) / P.nnb;
    int nt = (tile - P.cls[c].tile_begin) - mt * P.nnb;
    if (P.ntmajor) {
        nt = tile / P.cls[0].nmb;
        mt = tile - nt * P.cls[0].nmb;
    }

    // ------------------------------------------------------------------ per tile: set-up, pipeline fill, K loop, end phase.
    // (Issuing the next tile's set-up reads and first K steps under the stores of the finished tile was built and measured: it helps
    // the 128x64 instantiations by 2-4 % but pushes the 128x128 ones past 256 registers -- two workgroups per CU -- and the spills of
    // the end phase cost 10-15 %; with two workgroups per CU the partner covers most of a tile switch anyway.)
    for (int seg = 0;; ++seg) {
        // ------------------------------------------------------------------ tile set-up (all state of a tile is local to this
        // iteration: hoisting it out of the loop made the register allocator spill inside the chunk loop, -25 %)
        SK_TL(0, wall_clock64());
        const sk_class& cl = P.cls[c];
        const int nkc = __builtin_amdgcn_readfirstlane(cl.nkc), Cout = __builtin_amdgcn_readfirstlane(cl.Cout);
        const int ntaps = __builtin_amdgcn_readfirstlane(cl.ntaps);
        const int tbeg = __builtin_amdgcn_readfirstlane(P.tilecum[tile]), tend = __builtin_amdgcn_readfirstlane(P.tilecum[tile + 1]);
        const int a = pos - tbeg, b = min(s_end, tend) - tbeg;  // this workgroup does live steps [a, b) of the tile
        int2 ti = P.tileinfo[cl.mt_begin + mt];
        const int m0 = cl.row_begin + mt * BM, n0 = nt * BN;
        // tap tables of the class: lane t holds tap t
        const int v_ash = lane < SDT_MAX_TAPS ? cl.ashift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
        const int v_bsh = lane < SDT_MAX_TAPS ? cl.bshift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
        unsigned abase[RA], bbase[RB], inval[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int2 ri = ((const int2*)P.rowinfo)[2 * (m0 + r0 + 32 * i)];  // {X byte offset, invalid-tap mask}
            abase[i] = (unsigned)ri.x + (unsigned)kv * 16u;
            inval[i] = (unsigned)ri.y;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) bbase[i] = (unsigned)((n0 + r0 + 32 * i) * cl.Tw * cl.Cin) * (unsigned)sizeof(TX) + (unsigned)kv * 16u;  // W is (N, Tw, Cin)
        if (tid < BM) {
            const int2 ro = ((const int2*)P.rowinfo)[2 * (m0 + tid) + 1];  // {Y byte offset, statistics group}
            sOut[tid] = ro.x;
            sGrp[tid] = ro.y;
        }
        // live taps in residue-rotated order: bit i of the (host-rotated) mask is tap (rot + i) mod ntaps; a K step is (lowest set
        // bit, kc).  Everything in the loader is wave-uniform and branch-free (scalar selects): the steady loop stays ONE basic block.
        unsigned rmask = (unsigned)__builtin_amdgcn_readfirstlane(ti.x);
        const int rot = __builtin_amdgcn_readfirstlane(ti.y);
        for (int skip = a / nkc; skip > 0; --skip) rmask &= rmask - 1;  // start inside the tile
        int kc = a - (a / nkc) * nkc;
        int left = b - a;  // steps the loader still has to fetch
        f32x4 ra[RA], rb[RB];
        auto load = [&]() {
            // offsets of the step at (tap, kc); past the end of the segment everything is masked (loads return zeros)
            const bool on = left > 0 && rmask != 0u;
            int t = (rmask != 0u ? __builtin_ctz(rmask) : 0) + rot;
            t = t >= ntaps ? t - ntaps : t;
            t = on ? t : 0;
            const int ash = __builtin_amdgcn_readlane(v_ash, t), bsh = __builtin_amdgcn_readlane(v_bsh, t);
            const int cs = kc * (SK_BK * 4);
            // the row's invalid-tap bit moves to bit 31 of the offset: out of range, the load returns zeros.  Loader off: shift 0
            // brings the always-set bit 31 there
            const int sh = on ? 31 - t : 0;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const unsigned o = ((abase[i] + (unsigned)ash) & 0x7fffffffu) | ((inval[i] << sh) & 0x80000000u);
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, cs, 0));
            }
            const unsigned bsh_eff = (unsigned)bsh + (on ? 0u : SK_OOB);  // bbase + bshift < 2^31: adding the top bit pushes it out of range
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(bbase[i] + bsh_eff), cs, 0));
            // advance (scalar selects)
            --left;
            const bool wrap = kc + 1 == nkc;
            kc = wrap ? 0 : kc + 1;
            rmask = wrap ? (rmask & (rmask - 1u)) : rmask;
        };
        auto stage = [&](int buf) {
            float* wA = sA + buf * BM * SK_LDP + wofs;
            float* wB = sB + buf * BN * SK_LDP + wofs;
#pragma unroll
            for (int i = 0; i < RA; ++i) *(f32x4*)&wA[32 * i * SK_LDP] = ra[i];
#pragma unroll
            for (int i = 0; i < RB; ++i) *(f32x4*)&wB[32 * i * SK_LDP] = rb[i];
        };

        f32x16 acc[1][TM][TN], tot[TM][TN];
        f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
#define SK_READ(A, B, PA, PB, J)                                                                                        \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = *(const f32x4*)((PA) + tm * 32 * SK_LDP + (J) * 8);       \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) B[tn] = *(const f32x4*)((PB) + tn * 32 * SK_LDP + (J) * 8)
#define SK_MFMA(SET, A, B)                                                                                              \
    if constexpr (BF) {                                                                                                 \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)             \
            SK_BF_MFMA(acc[SET][tm][tn], A[tm], B[tn]);                                                                 \
    } else {                                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                 \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                           \
                acc[SET][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tm][e], B[tn][e], acc[SET][tm][tn], 0, 0, 0); \
    }

        // one K step (ONE instance in the kernel: variants of this body for the first step of an accumulation chunk doubled the
        // loop nest and made the register allocator spill around the chunk loop of the 128x128 / two-workgroups-per-CU kernels)
        auto step = [&](const int cur) {
            constexpr int SET = 0;
            const float* pa = sA + cur * BM * SK_LDP + fa;
            const float* pb = sB + cur * BN * SK_LDP + fb;
            // k-group 0: its MFMAs + fragments of k-group 1 + registers (step+1) -> LDS[next] + the global loads of step+2 (re-filling
            // the registers just staged: issued as early as possible, a full step before they are needed)
            SK_READ(a1, b1, pa, pb, 1);
            stage(cur ^ 1);
            load();
            SK_MFMA(SET, a0, b0);
            // k-group 1 + fragments of k-group 2
            SK_READ(a0, b0, pa, pb, 2);
            SK_MFMA(SET, a1, b1);
            // k-group 2 + fragments of k-group 3
            SK_READ(a1, b1, pa, pb, 3);
            SK_MFMA(SET, a0, b0);
            // the interleave: one feeding instruction after each MFMA, in this order
            if constexpr (BF) {
                sk_bf_interleave<0, 0, NM, TM + TN, RA + RB>();
            } else {
                constexpr int NF = TM + TN, NL = RA + RB;
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
#pragma unroll
                for (int q = 0; q < NL; ++q) { SK_SGB(0x8, 1); SK_SGB(0x200, 1); }
#pragma unroll
                for (int q = 0; q < NL; ++q) { SK_SGB(0x8, 1); SK_SGB(0x20, 1); }
                constexpr int u1 = NF + 2 * NL;
                if constexpr (u1 < NM) SK_SGB(0x8, NM - u1);
                constexpr int v1 = u1 < NM ? NM : u1;
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
                constexpr int u2 = v1 + NF;
                if constexpr (u2 < 2 * NM) SK_SGB(0x8, 2 * NM - u2);
                constexpr int v2 = u2 < 2 * NM ? 2 * NM : u2;
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
                constexpr int u3 = v2 + NF;
                if constexpr (u3 < 3 * NM) SK_SGB(0x8, 3 * NM - u3);
            }
            // k-group 3: everybody has read LDS[cur] and written LDS[next] once the fragment reads above have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            SK_READ(a0, b0, sA + (cur ^ 1) * BM * SK_LDP + fa, sB + (cur ^ 1) * BN * SK_LDP + fb, 0);
            SK_MFMA(SET, a1, b1);
#pragma unroll
            for (int q = 0; q < TM + TN; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
            if constexpr (NM > TM + TN) SK_SGB(0x8, NM - (TM + TN));
            __builtin_amdgcn_sched_barrier(0);
        };
        // the finished chunk goes into the running total and the accumulators restart from zero
        auto flush = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        tot[i][j][q] += acc[0][i][j][q];
                        acc[0][i][j][q] = 0.f;
                    }
        };
        SK_TL(1, wall_clock64());
        const int nsteps = b - a;
        if constexpr (BF) {
            // ---- bf16: a K step is 16 MFMAs of 32 cycles per wave, about the L2 latency under load.  TWO register sets: the loads of step s + 3 are
            // issued at the top of step s into the set that was just staged, i.e. two steps before they are written to LDS (with one set
            // the wave sat in vmcnt(0) at the top of every step: 2500 cycles per step against 512 of MFMA, tools/debug/sk_timeline.py
            // --dtype bf16).  The loop is unrolled by two so that register set and LDS buffer are compile-time.  One accumulator over
            // the whole K loop: the products of bf16 operands are exact in fp32 and the operands carry 2^-9 already.
            f32x4 ra2[2][RA], rb2[2][RB];
            auto load2 = [&](auto SI) {
                constexpr int S_ = decltype(SI)::value;
                const bool on = left > 0 && rmask != 0u;
                int t = (rmask != 0u ? __builtin_ctz(rmask) : 0) + rot;
                t = t >= ntaps ? t - ntaps : t;
                t = on ? t : 0;
                const int ash = __builtin_amdgcn_readlane(v_ash, t), bsh = __builtin_amdgcn_readlane(v_bsh, t);
                const int cs = kc * (SK_BK * 4);
                const int sh = on ? 31 - t : 0;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const unsigned o = ((abase[i] + (unsigned)ash) & 0x7fffffffu) | ((inval[i] << sh) & 0x80000000u);
#if defined(SK_BF_ABL) && (SK_BF_ABL & 1)  // ablation (wrong results): no global loads in the K loop
                    ra2[S_][i][0] = __uint_as_float(o + (unsigned)cs);
#else
                    ra2[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, cs, 0));
#endif
                }
                const unsigned bsh_eff = (unsigned)bsh + (on ? 0u : SK_OOB);
#pragma unroll
                for (int i = 0; i < RB; ++i)
#if defined(SK_BF_ABL) && (SK_BF_ABL & 1)
                    rb2[S_][i][0] = __uint_as_float(bbase[i] + bsh_eff + (unsigned)cs);
#else
                    rb2[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(bbase[i] + bsh_eff), cs, 0));
#endif
                --left;
                const bool wrap = kc + 1 == nkc;
                kc = wrap ? 0 : kc + 1;
                rmask = wrap ? (rmask & (rmask - 1u)) : rmask;
            };
            auto stage2 = [&](auto SI, const int buf) {
                constexpr int S_ = decltype(SI)::value;
                float* wA = sA + buf * BM * SK_LDP + wofs;
                float* wB = sB + buf * BN * SK_LDP + wofs;
#if defined(SK_BF_ABL) && (SK_BF_ABL & 2)  // ablation (wrong results): no LDS stores in the K loop (the registers stay live through an asm use)
#pragma unroll
                for (int i = 0; i < RA; ++i) asm volatile("" ::"v"(ra2[S_][i]));
#pragma unroll
                for (int i = 0; i < RB; ++i) asm volatile("" ::"v"(rb2[S_][i]));
                (void)wA;
                (void)wB;
#else
#pragma unroll
                for (int i = 0; i < RA; ++i) *(f32x4*)&wA[32 * i * SK_LDP] = ra2[S_][i];
#pragma unroll
                for (int i = 0; i < RB; ++i) *(f32x4*)&wB[32 * i * SK_LDP] = rb2[S_][i];
#endif
            };
            auto step2 = [&](auto CUR) {
                constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
                constexpr int SET = 0;
                const float* pa = sA + cur * BM * SK_LDP + fa;
                const float* pb = sB + cur * BN * SK_LDP + fb;
                SK_READ(a1, b1, pa, pb, 1);
                stage2(std::integral_constant<int, nx>{}, nx);  // step s + 1: registers -> LDS[next]
                load2(std::integral_constant<int, nx>{});       // step s + 3 -> the registers just staged
                SK_MFMA(SET, a0, b0);
                SK_READ(a0, b0, pa, pb, 2);
                SK_MFMA(SET, a1, b1);
                SK_READ(a1, b1, pa, pb, 3);
                SK_MFMA(SET, a0, b0);
                sk_bf_interleave<0, 0, NM, TM + TN, RA + RB>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                SK_READ(a0, b0, sA + nx * BM * SK_LDP + fa, sB + nx * BN * SK_LDP + fb, 0);
                SK_MFMA(SET, a1, b1);
#pragma unroll
                for (int q = 0; q < (NM < TM + TN ? NM : TM + TN); ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
                if constexpr (NM > TM + TN) SK_SGB(0x8, NM - (TM + TN));
                __builtin_amdgcn_sched_barrier(0);
            };
            load2(std::integral_constant<int, 0>{});
            stage2(std::integral_constant<int, 0>{}, 0);
            load2(std::integral_constant<int, 1>{});
            load2(std::integral_constant<int, 0>{});
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[0][i][j][q] = 0.f;
            SK_READ(a0, b0, sA + fa, sB + fb, 0);
            SK_TL(2, wall_clock64());
            SK_TL(5, (unsigned long long)nsteps);
            int s = 0;
            for (; s + 1 < nsteps; s += 2) {
                step2(std::integral_constant<int, 0>{});
                step2(std::integral_constant<int, 1>{});
            }
            if (s < nsteps) step2(std::integral_constant<int, 0>{});
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) tot[i][j] = acc[0][i][j];
        } else {
        load();
        stage(0);
        load();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[0][i][j][q] = 0.f, tot[i][j][q] = 0.f;
        SK_READ(a0, b0, sA + fa, sB + fb, 0);
        // K loop.  Chunks are aligned to the tile's own step index (a + s): the summation structure of a segment does not depend on
        // where the workgroup's range starts
        SK_TL(2, wall_clock64());
        SK_TL(5, (unsigned long long)nsteps);
        for (int s = 0; s < nsteps; ++s) {
            if (s > 0 && ((a + s) % SK_CHUNK) == 0) flush();  // uniform
            step(s & 1);
        }
        flush();
        }

        // ------------------------------------------------------------------ end of the segment
        const bool owner = a == 0, whole = owner && b == tend - tbeg;
        SK_TL(3, wall_clock64());
        SK_TL(6, whole ? 0ull : (owner ? 1ull : 2ull));
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
                        f32x4 v = {tot[i][j][4 * q], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]};
                        const int off = ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 16;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsS, off, 0, 16);  // sc1
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !SK_MUTED(r)) __hip_atomic_store((gu32*)(P.flags + r), P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (!whole) {
                // the ranges r+1 .. r_last hold the rest of this tile: r(x) = ((x + 1) * G - 1) / S for the tile's last step x
                const int r_last = (int)((((long)tend) * G - 1) / P.S);
                bool lost = false;  // a partner never arrived: the tile is POISONED (NaN), never stored with a partial sum missing
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
                            // the flag's only consumer lowers it again: every flag is zero between launches, whatever `epoch` the next one
                            // carries -- launches recorded into a hipGraph replay with the epoch they were captured with
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
                                    const f32x4 v = *(const f32x4*)(slab + ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 4);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) tot[i][j][4 * q + e] += v[e];
                                }
                    } else {
                        lost = true;
                    }
                    __syncthreads();
                }
                if (lost) {  // the error word is set; NaN outputs (and NaN statistics) make the failure visible downstream as well
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 16; ++q) tot[i][j][q] = __builtin_nanf("");
                }
            }
            sk_epilogue<TX, TY, BM, BN, EPI>(tot, sOut, sGrp, n0, wm, wn, lane, bias, rsY, Cout, stats, nb, P.ybytes);
        }
        SK_TL(4, wall_clock64());
        // ------------------------------------------------------------------ next tile of the range
        pos = tbeg + b;
        if (pos >= s_end) break;
        ++tile;
        if (P.ntmajor) {
            if (++mt == P.cls[0].nmb) {
                mt = 0;
                ++nt;
            }
        } else if (++nt == P.nnb) {
            nt = 0;
            if (++mt == P.cls[c].nmb) {
                mt = 0;
                ++c;
            }
        }
        __syncthreads();  // the epilogue's reads of sOut / sGrp are done before the next tile's set-up overwrites them
    }
#undef SK_READ
#undef SK_MFMA
}


// ---------------------------------------------------------------------------------------------
// Weight gradient on the same machinery:  dW[n, (t, c)] = sum_m dY[m, n] * X[row(m) + tap t, c]  -- a GEMM whose reduction runs over
// the output positions m (K = M / 32 steps per 128x128 tile of dW, thousands of steps), cut into G / T chunks per tile: one (tile,
// chunk) unit per persistent workgroup.  Every workgroup stores its partial tile into a slab (plain stores: the consumer is the NEXT
// kernel) and dw_sk_reduce_kernel adds the slabs of a tile in chunk order to dW: no atomics, bit-identical from run to run -- the deterministic
// weight gradient is the default, not an option.  Both operands are m-major in HBM (a 16-byte load is 4 consecutive n resp. c of one
// m), so the LDS tiles are [k = m][128] and an MFMA operand is one conflict-free ds_read_b32 (row k + lane / 32, column lane % 32).
// The per-row tables of the FORWARD geometry's plan give, per m, the dY row offset, the X row offset and the mask of taps that fall
// outside X; they are read two K steps ahead of the data they address.
template <int BM, int BN, int WPC>
__global__ __launch_bounds__(256, WPC) void convsk_dw_kernel(const float* __restrict__ X, const float* __restrict__ dY, const sk_args P,
                                                             const int K, const int ncol, const int nchunk, float* __restrict__ slabs) {
    // BM x BN tile of dW (n x (t, c)), each 128 or 64 wide.  Per operand of width Wd: Wd / 4 16-byte chunks per k row, KPT = Wd / 32
    // consecutive k rows per thread (4 or 2)
    constexpr int TM = BM / 64 > 0 ? BM / 64 : 1, TN = BN / 64 > 0 ? BN / 64 : 1, NM = TM * TN * 4;
    constexpr int KA = BM / 32, KB = BN / 32;  // k rows (= 16-byte loads) per thread and K step, dY / X
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS tiles [k / 4][row][k % 4]: a thread loads KPT CONSECUTIVE m (= k) of its 4-column chunk, transposes the block in registers
    // (register renaming + moves) and stores, per column, its k values as one 16- or 8-byte vector -- so an MFMA operand fragment is one
    // ds_read_b128 of 4 k values of a row, exactly as in the forward kernel (a [k][row] tile costs one ds_read_b32 per MFMA operand:
    // 8x the LDS instructions, measured 5-9 % slower than the atomics kernel it was meant to replace)
    float* sA = smem;                // [2][8][BM][4]   dY tile
    float* sB = smem + 2 * 32 * BM;  // [2][8][BN][4]   gathered X tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cqa = tid % (BM / 4), ksa = tid / (BM / 4);  // loader: chunk cq of the k rows KPT * ks .. + KPT - 1
    const int cqb = tid % (BN / 4), ksb = tid / (BN / 4);
    // Work unit = (tile, chunk of the K loop): workgroup u does chunk u / T of tile u % T, so the workgroups that run side by side
    // (consecutive u on one XCD) walk the SAME rows m at the same time, each for its own tile: dY and X rows come out of the L2 for
    // all but one of them.  (A stream-K split with the tile as the outer index made every workgroup stream private rows: 1.5 GB of
    // L2 misses per launch on the 3x3 layers.)
    const int G = P.G, T = P.T;
    const int bid = blockIdx.x;
    const int u = (bid & 7) * (G >> 3) + (bid >> 3);
    if (u >= T * nchunk) return;
    const int chunk = u / T, tile = u - chunk * T;
    const sk_class& cl = P.cls[0];
    const int Cin = __builtin_amdgcn_readfirstlane(cl.Cin);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)P.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)P.ybytes, 0x00020000);
    // fragment read offsets (floats): lane half h reads the k group 2 j + h of its row
    // LDS bank swizzle: row r of a k group sits in the 16-byte slot r ^ ((r >> 3) & 7).  The loader's lanes store rows 4 apart (64 B: only two
    // of the eight 16-byte bank groups of a 128-byte LDS cycle, a 4x slower ds_write_b128: SQ_LDS_BANK_CONFLICT was non-zero for these kernels only);
    // XOR-ing the row's bits 3..5 into its low three bits spreads 8 consecutive loader lanes over all eight groups and keeps 8 consecutive
    // fragment rows (bits 3..5 equal) a permutation of them
    int fa[TM], fb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / 2) + tm * 32 + (lane & 31);
        fa[tm] = ((lane >> 5) * BM + (r ^ ((r >> 3) & 7))) * 4;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int r = wn * (BN / 2) + tn * 32 + (lane & 31);
        fb[tn] = ((lane >> 5) * BN + (r ^ ((r >> 3) & 7))) * 4;
    }
    {
        const int a = (int)((long)chunk * K / nchunk), b = (int)((long)(chunk + 1) * K / nchunk);
        const int nt = tile / ncol, ct = tile - nt * ncol;
        const int n0 = nt * BM, j0 = ct * BN;
        // this thread's B column: 4 consecutive channels of ONE tap
        const int j = j0 + cqb * 4;
        const int t = j / Cin, c = j - t * Cin;
        const unsigned acol = (unsigned)(n0 + cqa * 4) * 4u;
        const unsigned bcol = (unsigned)cl.ashift[t] + (unsigned)c * 4u;
        const int sh = 31 - t;  // the row's invalid-tap bit t -> bit 31 of the offset
        int ya[KA];             // dY row offsets of this thread's k rows
        int2 xb[KB];            // {X row offset, invalid-tap mask} of this thread's k rows
        f32x4 ra[KA], rb[KB];
        int mrowa = (a * 32) + KA * ksa, mrowb = (a * 32) + KB * ksb;  // first rows of the next table reads
        auto load_rows = [&]() {
#pragma unroll
            for (int i = 0; i < KA; ++i) ya[i] = ((const int*)P.rowinfo)[4 * (mrowa + i) + 2];
#pragma unroll
            for (int i = 0; i < KB; ++i) xb[i] = ((const int2*)P.rowinfo)[2 * (mrowb + i)];
            mrowa += 32;
            mrowb += 32;
        };
        int left = b - a;
        auto load = [&]() {  // data of the step whose table rows are in ya / xb; past the end of the segment: masked
            const unsigned off_mask = left > 0 ? 0u : SK_OOB;
#pragma unroll
            for (int i = 0; i < KA; ++i) {
                const unsigned oa = ((unsigned)ya[i] + acol) | off_mask;  // rows past M carry SK_OOB already
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)oa, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const unsigned ob = (((unsigned)xb[i].x + bcol) & 0x7fffffffu) | (((unsigned)xb[i].y << sh) & 0x80000000u) | off_mask;
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)ob, 0, 0));
            }
            --left;
        };
        auto stage = [&](int buf) {  // register transpose: column e of the block = the k values of row (chunk * 4 + e)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rwa = (cqa * 4 + e) ^ ((cqa >> 1) & 7), rwb = (cqb * 4 + e) ^ ((cqb >> 1) & 7);  // row ^ ((row >> 3) & 7)
                if constexpr (KA == 4) {
                    const f32x4 va = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]};
                    *(f32x4*)&sA[buf * 32 * BM + (ksa * BM + rwa) * 4] = va;
                } else {
                    const f32x2 va = {ra[0][e], ra[1][e]};
                    *(f32x2*)&sA[buf * 32 * BM + ((ksa >> 1) * BM + rwa) * 4 + (ksa & 1) * 2] = va;
                }
                if constexpr (KB == 4) {
                    const f32x4 vb = {rb[0][e], rb[1][e], rb[2][e], rb[3][e]};
                    *(f32x4*)&sB[buf * 32 * BN + (ksb * BN + rwb) * 4] = vb;
                } else {
                    const f32x2 vb = {rb[0][e], rb[1][e]};
                    *(f32x2*)&sB[buf * 32 * BN + ((ksb >> 1) * BN + rwb) * 4 + (ksb & 1) * 2] = vb;
                }
            }
        };
        f32x16 acc[1][TM][TN], tot[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[0][i][jj][q] = 0.f, tot[i][jj][q] = 0.f;
        // fill: tables of step a -> data of step a -> LDS[0]; tables / data of step a+1 -> registers; tables of step a+2
        load_rows();
        load();
        load_rows();
        stage(0);
        load();
        load_rows();
        __syncthreads();
        f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
#define DW_READ(A, B, PA, PB, J)                                                                                            \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = *(const f32x4*)((PA) + fa[tm] + (J) * 2 * BM * 4);              \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) B[tn] = *(const f32x4*)((PB) + fb[tn] + (J) * 2 * BN * 4)
#define DW_MFMA(A, B)                                                                                                       \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                         \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                                   \
            acc[0][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tm][e], B[tn][e], acc[0][tm][tn], 0, 0, 0)
        DW_READ(a0, b0, sA, sB, 0);
        auto step = [&](const int cur) {
            const float* pa = sA + cur * 32 * BM;
            const float* pb = sB + cur * 32 * BN;
            // as the forward kernel's step: k-group 0 + fragments of k-group 1 + registers (step+1) -> LDS[next] + the global loads of
            // step+2 + the table rows of step+3
            DW_READ(a1, b1, pa, pb, 1);
            stage(cur ^ 1);
            load();
            load_rows();
            DW_MFMA(a0, b0);
            DW_READ(a0, b0, pa, pb, 2);
            DW_MFMA(a1, b1);
            DW_READ(a1, b1, pa, pb, 3);
            DW_MFMA(a0, b0);
            {
                constexpr int NF = TM + TN, NW = 8, NL = 2 * (KA + KB);  // fragment reads / LDS stores / global loads (data + tables)
                constexpr int NMF = NM > NF ? NM : NF;                    // MFMAs of a k-group available to pair with
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
#pragma unroll
                for (int q = 0; q < NW; ++q) { SK_SGB(0x8, 1); SK_SGB(0x200, 1); }
#pragma unroll
                for (int q = 0; q < NL; ++q) { SK_SGB(0x8, 1); SK_SGB(0x20, 1); }
                constexpr int u1 = NF + NW + NL;
                constexpr int v1 = u1 < NMF ? NMF : u1;
                if constexpr (u1 < NM) SK_SGB(0x8, NM - u1);
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
                constexpr int u2 = v1 + NF;
                constexpr int v2 = u2 < 2 * NM ? 2 * NM : u2;
                if constexpr (u2 < 2 * NM) SK_SGB(0x8, 2 * NM - u2);
#pragma unroll
                for (int q = 0; q < NF; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
                constexpr int u3 = v2 + NF;
                if constexpr (u3 < 3 * NM) SK_SGB(0x8, 3 * NM - u3);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of LDS[cur] and my writes of LDS[next] are done
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            DW_READ(a0, b0, sA + (cur ^ 1) * 32 * BM, sB + (cur ^ 1) * 32 * BN, 0);
            DW_MFMA(a1, b1);
#pragma unroll
            for (int q = 0; q < TM + TN; ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
            if constexpr (NM > TM + TN) SK_SGB(0x8, NM - (TM + TN));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto flush = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        tot[i][jj][q] += acc[0][i][jj][q];
                        acc[0][i][jj][q] = 0.f;
                    }
        };
        const int nsteps = b - a;
        for (int s = 0; s < nsteps; ++s) {
            if (s > 0 && ((a + s) % SK_CHUNK) == 0) flush();  // chunks aligned to the tile's own step index
            step(s & 1);
        }
        flush();
#undef DW_READ
#undef DW_MFMA
        // partial tile -> slab of this unit (natural [n][j] layout: the reduce kernel reads it coalesced)
        float* slab = slabs + (size_t)u * (BM * BN);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int nl = wm * (BM / 2) + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    const int jl = wn * (BN / 2) + tn * 32 + (lane & 31);
                    slab[nl * BN + jl] = tot[tm][tn][q];
                }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient (bf16-storage path): the same (tile, K-chunk) decomposition, slabs and ordered reduce as convsk_dw_kernel, on
// v_mfma_f32_32x32x16_bf16.  A K step is 64 output positions m.  Both operands are m-major in HBM and the bf16 MFMA wants 8 CONSECUTIVE k
// (= m) per lane, i.e. a transposed operand: the tiles are stored m-major in LDS exactly as they are loaded ([m][BM] / [m][BN] bf16, one
// 16-byte load = 8 columns of one m) and the fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS read: the 16 lanes of a
// group address a 4 (m) x 16 (columns) block -- lane p the 4 columns 4 (p & 3) .. of row p >> 2 -- and lane i receives column i of the four
// rows (tools/debug/tr16_probe.hip: semantics, the MFMA identity and the bank behaviour measured on the GPU).  The 64-byte segments of a row
// are XOR-swizzled with the row so that the four rows of a group fall into the four quarters of the 256-byte bank window.
template <int W>
__device__ __forceinline__ int bfdw_off(const int row, const int col) {  // element offset of (row, col) in a swizzled [64][W] bf16 tile
    constexpr int S = W / 32;                                            // 64-byte segments per row
    const int f = S >= 4 ? (row & 3) : ((row >> 1) & 1);
    return row * W + ((((col >> 5) ^ f) << 5) | (col & 31));
}
typedef short sk_s16x4 __attribute__((ext_vector_type(4)));
typedef short sk_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) sk_s16x4 sk_lds_s16x4;

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void convbf_dw_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ dY, const sk_args P, const int K,
                                                           const int ncol, const int nchunk, float* __restrict__ slabs) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int CA = BM / 8, CB = BN / 8;          // 16-byte chunks per tile row
    constexpr int RPA = 256 / CA, RPB = 256 / CB;    // tile rows covered by one pass of the 256 threads
    constexpr int LA = 64 / RPA, LB = 64 / RPB;      // loads per thread and K step
    extern __shared__ __attribute__((aligned(16))) short smem_h[];
    short* sA = smem_h;                 // [2][64][BM]
    short* sB = smem_h + 2 * 64 * BM;   // [2][64][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cqa = tid % CA, rra = tid / CA, cqb = tid % CB, rrb = tid / CB;
    const int G = P.G, T = P.T;
    const int bid = blockIdx.x;
    const int u = (bid & 7) * (G >> 3) + (bid >> 3);
    if (u >= T * nchunk) return;
    const int chunk = u / T, tile = u - chunk * T;
    const sk_class& cl = P.cls[0];
    const int Cin = __builtin_amdgcn_readfirstlane(cl.Cin);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)P.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)P.ybytes, 0x00020000);
    const int a = (int)((long)chunk * K / nchunk), b = (int)((long)(chunk + 1) * K / nchunk);
    const int nt = tile / ncol, ct = tile - nt * ncol;
    const int n0 = nt * BM, j0 = ct * BN;
    // this thread's B columns: 8 consecutive channels of ONE tap (Cin % 64 == 0)
    const int j = j0 + cqb * 8;
    const int t = j / Cin, c = j - t * Cin;
    const unsigned acol = (unsigned)(n0 + cqa * 8) * 2u;
    const unsigned bcol = (unsigned)cl.ashift[t] + (unsigned)c * 2u;
    const int sh = 31 - t;  // the row's invalid-tap bit t -> bit 31 of the offset
    int ya[LA];
    int2 xb[LB];
    f32x4 ra[2][LA], rb[2][LB];  // two register sets: a load is issued two steps before its data is written to LDS (see convsk_kernel's bf16 loop)
    int mrow = a * 64;  // first row of the next table read
    auto load_rows = [&]() {
#pragma unroll
        for (int i = 0; i < LA; ++i) ya[i] = ((const int*)P.rowinfo)[4 * (mrow + rra + RPA * i) + 2];
#pragma unroll
        for (int i = 0; i < LB; ++i) xb[i] = ((const int2*)P.rowinfo)[2 * (mrow + rrb + RPB * i)];
        mrow += 64;
    };
    int left = b - a;
    auto load = [&](auto SI) {
        constexpr int S_ = decltype(SI)::value;
        const unsigned off_mask = left > 0 ? 0u : SK_OOB;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const unsigned oa = ((unsigned)ya[i] + acol) | off_mask;  // rows past M carry SK_OOB already
            ra[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)oa, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const unsigned ob = (((unsigned)xb[i].x + bcol) & 0x7fffffffu) | (((unsigned)xb[i].y << sh) & 0x80000000u) | off_mask;
            rb[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)ob, 0, 0));
        }
        --left;
    };
    auto stage = [&](auto SI, const int buf) {
        constexpr int S_ = decltype(SI)::value;
#pragma unroll
        for (int i = 0; i < LA; ++i) *(f32x4*)&sA[buf * 64 * BM + bfdw_off<BM>(rra + RPA * i, cqa * 8)] = ra[S_][i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *(f32x4*)&sB[buf * 64 * BN + bfdw_off<BN>(rrb + RPB * i, cqb * 8)] = rb[S_][i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][jj][q] = 0.f;
    // fragment addressing: lane = 16 g + p; the group reads rows 8 (g >> 1) + (p >> 2) (+ 4 for the second half of the k block), columns
    // 16 (g & 1) + 4 (p & 3) of its 32-column sub-tile
    const int g4 = lane >> 4, p4 = lane & 15;
    const int frow = 8 * (g4 >> 1) + (p4 >> 2), fcol = 16 * (g4 & 1) + 4 * (p4 & 3);
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    load_rows();
    load(I0{});     // step 0 -> set 0
    load_rows();
    stage(I0{}, 0);
    load(I1{});     // step 1 -> set 1
    load_rows();
    load(I0{});     // step 2 -> set 0
    load_rows();    // tables of step 3
    __syncthreads();
    const int nsteps = b - a;
    auto step = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
        const short* pa = sA + cur * 64 * BM;
        const short* pb = sB + cur * 64 * BN;
        stage(std::integral_constant<int, nx>{}, nx);  // step s + 1: registers (loaded two steps ago) -> LDS[next]
        load(std::integral_constant<int, nx>{});       // step s + 3 -> the registers just staged
        load_rows();                                   // tables of step s + 4
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {  // k16 blocks of the step
            sk_s16x8 fa[TM], fb[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int col = wm * (BM / 2) + tm * 32 + fcol;
                const sk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pa + bfdw_off<BM>(16 * kb + frow, col)));
                const sk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pa + bfdw_off<BM>(16 * kb + frow + 4, col)));
                fa[tm] = (sk_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = wn * (BN / 2) + tn * 32 + fcol;
                const sk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pb + bfdw_off<BN>(16 * kb + frow, col)));
                const sk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pb + bfdw_off<BN>(16 * kb + frow + 4, col)));
                fb[tn] = (sk_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, fa[tm]), __builtin_bit_cast(sk_bf16x8, fb[tn]),
                                                                          acc[tm][tn], 0, 0, 0);
        }
        // everybody has read LDS[cur] and written LDS[cur ^ 1]; the global loads stay in flight across the barrier (no vmcnt drain)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        step(I0{});
        step(I1{});
    }
    if (s < nsteps) step(I0{});
    // partial tile -> slab of this unit (natural [n][j] layout: the reduce kernel reads it coalesced)
    float* slab = slabs + (size_t)u * (BM * BN);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nl = wm * (BM / 2) + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const int jl = wn * (BN / 2) + tn * 32 + (lane & 31);
                slab[nl * BN + jl] = acc[tm][tn][q];
            }
}

// ---------------------------------------------------------------------------------------------
// Split-fp32 weight gradient (fp32 tensors, fp32-grade products at the bf16 matrix rate; see x3_stage in convbf.hip for the arithmetic): the
// decomposition, slabs and ordered reduce of convsk_dw_kernel and its plan (32 rows m per plan step), the transposed-fragment LDS layout of
// convbf_dw_kernel per bf16 PLANE.  A sub-step is 16 rows m = one k16 block: a thread loads 16 bytes = 4 fp32 columns of one row, splits them into
// three 4 x bf16 pieces and stores each to its plane ([16][W] bf16, 64-byte segments XOR-swizzled with the row); fragments by ds_read_b64_tr_b16,
// six MFMAs per fragment pair (small products first), the accumulators flushed into `tot` every SK_CHUNK plan steps as the fp32 kernel does.
// Two workgroups of 4 waves per CU (2 x 48 KB of LDS), not in step with each other: one's split / store phase runs under the other's MFMAs.
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void convx3_dw_kernel(const float* __restrict__ X, const float* __restrict__ dY, const sk_args P, const int K,
                                                           const int ncol, const int nchunk, float* __restrict__ slabs) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int SUB = 16;                          // rows per sub-step
    constexpr int CA = BM / 4, CB = BN / 4;          // 16-byte chunks (4 fp32 columns) per tile row
    constexpr int RPA = 256 / CA, RPB = 256 / CB;    // tile rows covered by one pass of the 256 threads
    constexpr int LA = SUB / RPA, LB = SUB / RPB;    // loads per thread and sub-step
    static_assert(LA >= 1 && LB >= 1, "tile width");
    extern __shared__ __attribute__((aligned(16))) short smem_h[];
    short* sA = smem_h;                           // [2][3][16][BM]
    short* sB = smem_h + 2 * 3 * SUB * BM;        // [2][3][16][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cqa = tid % CA, rra = tid / CA, cqb = tid % CB, rrb = tid / CB;
    const int G = P.G, T = P.T;
    const int bid = blockIdx.x;
    const int u = (bid & 7) * (G >> 3) + (bid >> 3);
    if (u >= T * nchunk) return;
    const int chunk = u / T, tile = u - chunk * T;
    const sk_class& cl = P.cls[0];
    const int Cin = __builtin_amdgcn_readfirstlane(cl.Cin);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)P.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)P.ybytes, 0x00020000);
    const int a = (int)((long)chunk * K / nchunk), b = (int)((long)(chunk + 1) * K / nchunk);  // plan steps of 32 rows
    const int nt = tile / ncol, ct = tile - nt * ncol;
    const int n0 = nt * BM, j0 = ct * BN;
    const int j = j0 + cqb * 4;  // this thread's B columns: 4 consecutive channels of ONE tap
    const int t = j / Cin, c = j - t * Cin;
    // ragged last column tile (taps * Cin not a multiple of BN: the 9 x 64 = 576 columns of L2 in five 128-wide tiles): columns of a tap that does
    // not exist load zeros (the reduce kernel does not store them)
    const unsigned tmask = t < cl.ntaps ? 0u : SK_OOB;
    const unsigned acol = (unsigned)(n0 + cqa * 4) * 4u;
    const unsigned bcol = (unsigned)cl.ashift[t < SDT_MAX_TAPS ? t : SDT_MAX_TAPS - 1] + (unsigned)c * 4u;
    const int sh = 31 - t;  // the row's invalid-tap bit t -> bit 31 of the offset
    int ya[LA];
    int2 xb[LB];
    f32x4 ra[2][LA], rb[2][LB];  // two register sets: a load is issued two sub-steps before its data is split and stored
    int mrow = a * 32;           // first row of the next table read
    auto load_rows = [&]() {
#pragma unroll
        for (int i = 0; i < LA; ++i) ya[i] = ((const int*)P.rowinfo)[4 * (mrow + rra + RPA * i) + 2];
#pragma unroll
        for (int i = 0; i < LB; ++i) xb[i] = ((const int2*)P.rowinfo)[2 * (mrow + rrb + RPB * i)];
        mrow += SUB;
    };
    int left = 2 * (b - a);
    auto load = [&](auto SI) {
        constexpr int S_ = decltype(SI)::value;
        const unsigned off_mask = left > 0 ? 0u : SK_OOB;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const unsigned oa = ((unsigned)ya[i] + acol) | off_mask;  // rows past M carry SK_OOB already
            ra[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)oa, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const unsigned ob = (((unsigned)xb[i].x + bcol) & 0x7fffffffu) | (((unsigned)xb[i].y << sh) & 0x80000000u) | off_mask | tmask;
            rb[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)ob, 0, 0));
        }
        --left;
    };
    auto split_store = [&](short* base, const int plane_stride, const f32x4 v) {  // (as x3_stage of convbf.hip: 4 fp32 -> 3 x 4 bf16, exact)
        unsigned h[2], m[2], l[2];
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2_ x = {v[2 * q], v[2 * q + 1]};
            h[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, sk_bf16x2));
            const f32x2_ r = x - f32x2_{__uint_as_float(h[q] << 16), __uint_as_float(h[q] & 0xffff0000u)};
            m[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sk_bf16x2));
            const f32x2_ t = r - f32x2_{__uint_as_float(m[q] << 16), __uint_as_float(m[q] & 0xffff0000u)};
            l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, sk_bf16x2));
        }
        *(uint2*)base = uint2{h[0], h[1]};
        *(uint2*)(base + plane_stride) = uint2{m[0], m[1]};
        *(uint2*)(base + 2 * plane_stride) = uint2{l[0], l[1]};
    };
    auto stage = [&](auto SI, const int buf) {
        constexpr int S_ = decltype(SI)::value;
#pragma unroll
        for (int i = 0; i < LA; ++i) split_store(sA + buf * 3 * SUB * BM + bfdw_off<BM>(rra + RPA * i, cqa * 4), SUB * BM, ra[S_][i]);
#pragma unroll
        for (int i = 0; i < LB; ++i) split_store(sB + buf * 3 * SUB * BN + bfdw_off<BN>(rrb + RPB * i, cqb * 4), SUB * BN, rb[S_][i]);
    };
    f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][jj][q] = 0.f, tot[i][jj][q] = 0.f;
    // fragment addressing as in convbf_dw_kernel: lane = 16 g + p reads rows 8 (g >> 1) + (p >> 2) (+ 4), columns 16 (g & 1) + 4 (p & 3) of its 32-column block
    const int g4 = lane >> 4, p4 = lane & 15;
    const int frow = 8 * (g4 >> 1) + (p4 >> 2), fcol = 16 * (g4 & 1) + 4 * (p4 & 3);
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    load_rows();
    load(I0{});     // sub-step 0 -> set 0
    load_rows();
    stage(I0{}, 0);
    load(I1{});     // sub-step 1 -> set 1
    load_rows();
    load(I0{});     // sub-step 2 -> set 0
    load_rows();    // tables of sub-step 3
    __syncthreads();
    const int nsub = 2 * (b - a);
    auto step = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // planes hi / mid / lo = 0 / 1 / 2: small products first
        const short* pa = sA + cur * 3 * SUB * BM;
        const short* pb = sB + cur * 3 * SUB * BN;
        sk_s16x8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int col = wm * (BM / 2) + tm * 32 + fcol;
                const sk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pa + pl * SUB * BM + bfdw_off<BM>(frow, col)));
                const sk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pa + pl * SUB * BM + bfdw_off<BM>(frow + 4, col)));
                fa[pl][tm] = (sk_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = wn * (BN / 2) + tn * 32 + fcol;
                const sk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pb + pl * SUB * BN + bfdw_off<BN>(frow, col)));
                const sk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sk_lds_s16x4*)(pb + pl * SUB * BN + bfdw_off<BN>(frow + 4, col)));
                fb[pl][tn] = (sk_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
        stage(std::integral_constant<int, nx>{}, nx);  // sub-step s + 1: registers (loaded two sub-steps ago) -> split -> LDS[next]
        load(std::integral_constant<int, nx>{});       // sub-step s + 3 -> the registers just stored
        load_rows();                                   // tables of sub-step s + 4
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, fa[PA[pr]][tm]), __builtin_bit_cast(sk_bf16x8, fb[PB[pr]][tn]),
                                                                          acc[tm][tn], 0, 0, 0);
        // everybody has read LDS[cur] and written LDS[cur ^ 1]; the global loads stay in flight across the barrier (no vmcnt drain)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    tot[i][jj][q] += acc[i][jj][q];
                    acc[i][jj][q] = 0.f;
                }
    };
    for (int s = 0; s < nsub; s += 2) {  // (nsub is even: a plan step is two sub-steps)
        if (s > 0 && ((a + s / 2) % SK_CHUNK) == 0) flush();  // chunks aligned to the tile's own step index, as in convsk_dw_kernel
        step(I0{});
        step(I1{});
    }
    flush();
    float* slab = slabs + (size_t)u * (BM * BN);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nl = wm * (BM / 2) + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const int jl = wn * (BN / 2) + tn * 32 + (lane & 31);
                slab[nl * BN + jl] = tot[tm][tn][q];
            }
}

// dw[n, wt(t), c] += the slabs of tile (nt, ct).  Q lanes per 4 consecutive columns of a tile row: lane q adds the chunks
// [q * nchunk / Q, (q + 1) * nchunk / Q) in order, then the Q sums are added in order q = 0 .. Q-1 -- a fixed tree, no atomics.  Every
// launch reads the same 33.5 MB of slabs (T * nchunk = 512 units of 64 KB), so the few-tile layers are short of workgroups, not of
// bandwidth: Q = 1 took 17-19 us with 64 / 72 workgroups (1.8 TB/s) against 9 us with 1024; the host picks Q so that the grid has >= 512.
template <int Q>
__global__ __launch_bounds__(256) void dw_sk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, const sk_args P, const int ncol,
                                                           const int nchunk, const int Cout, const int Tw, const int BM, const int BN) {
    const int per_tile = BM * BN / 4 * Q / 256;              // workgroups per tile
    const int tile = blockIdx.x / per_tile;
    const int lt = (blockIdx.x % per_tile) * 256 + threadIdx.x;
    const int e = lt / Q, q = lt % Q;                        // float4 index inside the tile, chunk group
    const int nl = e / (BN / 4), jl = (e % (BN / 4)) * 4;
    const int nt = tile / ncol, ct = tile - nt * ncol;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int c0 = (int)((long)q * nchunk / Q), c1 = (int)((long)(q + 1) * nchunk / Q);
    for (int ch = c0; ch < c1; ++ch) sum += *(const f32x4*)(slabs + ((size_t)ch * P.T + tile) * (size_t)(BM * BN) + nl * BN + jl);
    if constexpr (Q > 1) {  // lanes q = 1 .. Q-1 of the group hand their sums to lane q = 0, which adds them in order
        f32x4 tot = sum;
#pragma unroll
        for (int k = 1; k < Q; ++k) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = __shfl_down(sum[c], k, Q);
            tot += o;
        }
        sum = tot;
        if (q != 0) return;
    }
    const sk_class& cl = P.cls[0];
    const int j = ct * BN + jl, t = j / cl.Cin, c = j - t * cl.Cin;
    const int n = nt * BM + nl;
    if (n < Cout && t < cl.ntaps) {  // (columns of a ragged last tile past the last tap: nothing to store)
        float* d = dw + ((size_t)n * Tw + cl.dyx[t]) * cl.Cin + c;  // dyx[] of a weight-gradient plan holds wt[t]
        *(f32x4*)d += sum;
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: the plan.  Layout of the blob (all int32, offsets in ints from the start of the blob):
//   [0] magic  [1] BM  [2] BN  [3] G  [4] ncls  [5] nnb  [6] T  [7] S  [8] rows (total, padded)  [9] n m-tiles (total)
//   [10] off rowinfo  [11] off tileinfo  [12] off tilecum  [13] off range_tile  [14] off classes  [15] total ints
#define SK_MAGIC 0x534b3033
#define SK_HDR 16
#define SK_CLS_INTS (11 + 3 * SDT_MAX_TAPS)

// esz: bytes per element of X / W (4 fp32, 2 bf16).  A K step is 128 bytes of a row.
static bool sk_supported(const sdt_conv_geom* const* gs, int ncls, int bm, int bn, int esz = 4) {
    for (int c = 0; c < ncls; ++c) {
        const sdt_conv_geom& g = *gs[c];
        if ((g.Cin * esz) % 128 != 0 || g.Cout % bn != 0 || g.Hi >= 32768 || g.Wi >= 32768 || g.ntaps > SDT_MAX_TAPS) return false;
        if (g.Cin != gs[0]->Cin || g.Cout != gs[0]->Cout || g.Tw != gs[0]->Tw || g.B != gs[0]->B || g.Hi != gs[0]->Hi || g.Wi != gs[0]->Wi ||
            g.Hy != gs[0]->Hy || g.Wy != gs[0]->Wy)
            return false;
    }
    return true;
}

// workgroups per CU (experiment switch, sdt_convsk_set_wg_per_cu): 1 = one wave per SIMD, every bubble of a tile switch is exposed
// but the pipelined loop runs at 0.91 of the peak; 2 = two waves per SIMD cover each other's tile switches (0.88 in the loop)
static int g_sk_wpc = 2;
extern "C" int sdt_convsk_set_wg_per_cu(int n) {
    SDT_CHECK_ARG(n == 1 || n == 2, "1 or 2 workgroups per CU");
    g_sk_wpc = n;
    return SDT_OK;
}

// Workgroup slots (of the GPU's 512 two-per-CU slots) left free by plans built afterwards (multiple of 8, up to 256 = half of the GPU -- two PROCESSES
// that share one GPU, the 2-ranks-on-1-GPU tests, then fit side by side exactly: 2 x 256 two-per-CU workgroups or 2 x 128 one-per-CU ones; VERDICT r5
// weak 1: the earlier cap of 248 left 2 x 264 > 512 / 2 x 132 > 256): a persistent launch that fills every slot of the GPU cannot share it
// with another long-lived kernel -- the kernels of a collective (RCCL all-reduce: a few dozen workgroups that live for the whole exchange) take slots,
// the conv workgroups that find none start when the first ones END, and the launch takes twice as long with 6 % of the GPU working
// (tools/debug/comm_emulation.py: 32 such workgroups for 1.2 ms of a step cost 5.5 %, for 3 ms 23 %).  Data-parallel runs therefore plan their BACKWARD
// launches -- the ones a gradient exchange overlaps -- with a reserve (speechdrivestemplates_amd/dp.py).
static int g_sk_reserve = 0;
extern "C" int sdt_convsk_set_reserved_slots(int n) {
    SDT_CHECK_ARG(n >= 0 && n <= 256 && n % 8 == 0, "reserve must be a multiple of 8, at most 256 (half of the GPU)");
    g_sk_reserve = n;
    return SDT_OK;
}
// persistent workgroups of a plan: two-per-CU kernels fill the slots that are not reserved; a one-per-CU workgroup holds a whole CU (two slots)
static int sk_grid(int wpc) { return wpc == 2 ? 512 - g_sk_reserve : (256 - g_sk_reserve / 2) & ~7; }

static void sk_tile_choice(const sdt_conv_geom& g, int& bm, int& bn) {
    if (g.Cout % 128 == 0) bm = 128, bn = 128;
    else if (g_sk_wpc == 1) bm = 256, bn = 64;
    else bm = 128, bn = 64;  // two workgroups per CU: 2 x 55 KB of LDS
}

static int64_t sk_plan_ints(const sdt_conv_geom* const* gs, int ncls, int bm, int bn, int G) {
    int64_t rows = 0, mts = 0;
    for (int c = 0; c < ncls; ++c) {
        const int64_t M = (int64_t)gs[c]->B * gs[c]->Ho * gs[c]->Wo;
        const int64_t nmb = cdiv64(M, bm);
        rows += nmb * bm;
        mts += nmb;
    }
    const int nnb = gs[0]->Cout / bn;
    const int64_t T = mts * nnb;
    return SK_HDR + rows * 4 + mts * 2 + (T + 1) + G + (int64_t)ncls * SK_CLS_INTS;
}

// tile and grid from the workgroups-per-CU setting and the element size.  bf16 operands with ONE workgroup per CU: the bf16-shaped kernel of
// convbf.hip -- 256-row tiles as wide as the layer (256 / 128 / 64 columns), 8 waves; a reserve of n slots leaves n / 2 of its CUs free.
// fp32 operands with ONE workgroup per CU and sdt_convsk_set_f32_split(1): the same kernel in its split-fp32 form (three bf16 planes per operand
// made by the loader, six bf16 MFMAs per fragment pair; 128 x 128 / 256 x 64 tiles: see convx3_launch).
static int g_sk_split = 0;
extern "C" int sdt_convsk_set_f32_split(int on) {
    g_sk_split = on ? 1 : 0;
    return SDT_OK;
}
static int g_dw_wide = 0;  // 1: ragged 128-wide column tiles for the split-fp32 weight gradient (dw_tile); 0: the tile rule of rounds 3-5
extern "C" int sdt_convsk_set_dw_wide_tiles(int on) {
    g_dw_wide = on ? 1 : 0;
    return SDT_OK;
}
static bool is_x3(int esz) { return esz == 4 && g_sk_wpc == 1 && g_sk_split; }
static bool is_bf2(int esz) { return (esz == 2 && g_sk_wpc == 1) || is_x3(esz); }  // "one 8-wave workgroup per CU" plans
static void plan_shape(const sdt_conv_geom& g, int esz, int& bm, int& bn, int& G) {
    if (is_bf2(esz)) {
        G = sk_grid(1);
        bm = 256;
        bn = (g.Cout % 256 == 0 && esz == 2) ? 256 : (g.Cout % 128 == 0 ? 128 : 64);
        if (esz == 4) {  // split-fp32 form: 64 x 32 per wave (128 x 128, or 256 x 64 for the 64-channel outputs)
            bm = bn == 128 ? 128 : 256;
            return;
        }
        // few rows (L5 - L7 at 32 clips: 16960 / 16960 / 8160): 256-row tiles would be fewer than CUs and every one of them cut between several
        // workgroups -- 128 x 128 tiles instead (about one whole tile per CU: no slab hand-off for most of them)
        const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
        if (cdiv64(M, 256) * (g.Cout / bn) < G && g.Cout % 128 == 0) bm = 128, bn = 128;
        return;
    }
    sk_tile_choice(g, bm, bn);
    G = sk_grid(g_sk_wpc);
}

static int dtype_bytes(int dtype) { return dtype == SDT_F32 ? 4 : (dtype == SDT_BF16 ? 2 : 0); }

static int plan_supported(const sdt_conv_geom* geoms, int ncls, int esz) {
    if (!geoms || ncls < 1 || ncls > SK_MAXC || (esz != 4 && esz != 2)) return 0;
    const sdt_conv_geom* gs[SK_MAXC];
    for (int c = 0; c < ncls; ++c) gs[c] = geoms + c;
    int bm, bn, G;
    plan_shape(*gs[0], esz, bm, bn, G);
    return sk_supported(gs, ncls, bm, bn, esz) ? 1 : 0;
}
extern "C" int sdt_convsk_supported(const sdt_conv_geom* geoms, int ncls) { return plan_supported(geoms, ncls, 4); }
extern "C" int sdt_convsk_supported_t(const sdt_conv_geom* geoms, int ncls, int x_dtype) { return plan_supported(geoms, ncls, dtype_bytes(x_dtype)); }

// grid of a plan (number of persistent workgroups): one per CU
extern "C" int sdt_convsk_grid(void) { return sk_grid(g_sk_wpc); }

static int64_t plan_bytes(const sdt_conv_geom* geoms, int ncls, int esz) {
    if (!plan_supported(geoms, ncls, esz)) return -1;
    const sdt_conv_geom* gs[SK_MAXC];
    for (int c = 0; c < ncls; ++c) gs[c] = geoms + c;
    int bm, bn, G;
    plan_shape(*gs[0], esz, bm, bn, G);
    return sk_plan_ints(gs, ncls, bm, bn, G) * 4;
}
extern "C" int64_t sdt_convsk_plan_bytes(const sdt_conv_geom* geoms, int ncls) { return plan_bytes(geoms, ncls, 4); }
extern "C" int64_t sdt_convsk_plan_bytes_t(const sdt_conv_geom* geoms, int ncls, int x_dtype) { return plan_bytes(geoms, ncls, dtype_bytes(x_dtype)); }

// workspace of a launch: slabs + flags + error word (bytes); the caller zero-fills it ONCE after allocation and hands the same
// buffer to every launch of one stream with a strictly increasing epoch (>= 1)
// (slabs: 512 ranges x 128 x 128 fp32 or 256 ranges x 256 x 256 fp32 -- the bf16-shaped kernel's tiles)
#define SK_SLAB_BYTES ((int64_t)256 * 256 * 256 * 4)
extern "C" int64_t sdt_convsk_workspace_bytes(void) { return SK_SLAB_BYTES + (int64_t)512 * 4 + 64; }

// Builds the plan into host memory `out` (sdt_convsk_plan_bytes bytes); the caller copies it to the device once per geometry.
// rows_per_group > 0: statistics group of row m of class c = m / rows_per_group (forward statistics) -- for an input gradient with
// normalisation-backward statistics pass -1: the group is the batch item (groups == B) or 0 (groups == 1), chosen by `bwd_groups`.
// How long the owner of a split tile polls a partner's flag (one poll = a relaxed load + s_sleep 8, ~1 us under load) before it gives
// up: the launch's error word is set (ops.streamk_error_codes(); Trainer raises on it) and the tile is stored as NaN.  The default is
// seconds -- partners publish at the START of their ranges, so a partner that has not arrived by then is not coming (it was never
// dispatched: the GPU is shared with a process that holds its slot).
static int g_sk_korder = 0;
// K order of the 8-wave kernels' tiles (convbf.hip): 0 = tap-major (all channel chunks of a tap, then the next tap), 1 = chunk-major (all live taps of a
// 128-byte channel chunk, then the next chunk: neighbouring taps re-read the previous step's cache lines while the vector L1 still holds them).
// A launch-time setting (kernel argument), not a plan property: the plan counts live steps per tile, which both orders agree on.
extern "C" int sdt_convsk_set_k_order(int order) {
    SDT_CHECK_ARG(order == 0 || order == 1, "0 (tap-major) or 1 (chunk-major)");
    g_sk_korder = order;
    return SDT_OK;
}
static unsigned g_sk_spin_limit = 1u << 22;
extern "C" int sdt_convsk_set_spin_limit(unsigned polls) {
    g_sk_spin_limit = polls;
    return SDT_OK;
}
extern "C" unsigned sdt_convsk_get_spin_limit(void) { return g_sk_spin_limit; }
static int g_sk_ntmajor_bytes = 2 << 20;
static int g_sk_perm_pct = 95, g_sk_perm_pct_bwd = 95;  // image-row-major tile order when it leaves <= this many % of the K steps
#ifdef SDT_TUNING
extern "C" int sdt_convsk_set_ntmajor_bytes(int bytes) {
    g_sk_ntmajor_bytes = bytes;
    return SDT_OK;
}
extern "C" int sdt_convsk_set_perm_pct(int pct) {  // forward plans: pct % 1000, input-gradient plans: pct / 1000
    g_sk_perm_pct = pct % 1000;
    g_sk_perm_pct_bwd = pct / 1000;
    return SDT_OK;
}
#endif
// esz / ysz: bytes per element of X, W / of Y (every byte offset of the plan is in those units; a K step is 128 bytes of an X row)
static int plan_build(const sdt_conv_geom* geoms, int ncls, int rows_per_group, int bwd_groups, void* out, int64_t out_bytes, int esz, int ysz) {
    SDT_CHECK_ARG(plan_supported(geoms, ncls, esz) && (ysz == 4 || ysz == 2), "geometry / element types not supported by this kernel");
    SDT_CHECK_ARG(out != nullptr && out_bytes >= plan_bytes(geoms, ncls, esz), "plan buffer too small");
    const sdt_conv_geom* gs[SK_MAXC];
    for (int c = 0; c < ncls; ++c) gs[c] = geoms + c;
    int bm, bn, G;
    plan_shape(*gs[0], esz, bm, bn, G);
    int* P = (int*)out;
    const int nnb = gs[0]->Cout / bn;
    int64_t rows = 0, mts = 0;
    for (int c = 0; c < ncls; ++c) {
        const int64_t nmb = cdiv64((int64_t)gs[c]->B * gs[c]->Ho * gs[c]->Wo, bm);
        rows += nmb * bm;
        mts += nmb;
    }
    const int64_t T = mts * nnb;
    const int64_t o_row = SK_HDR, o_ti = o_row + rows * 4, o_cum = o_ti + mts * 2, o_rt = o_cum + T + 1, o_cls = o_rt + G;
    int* rowinfo = P + o_row;
    int* tileinfo = P + o_ti;
    int* tilecum = P + o_cum;
    int* range_tile = P + o_rt;
    int* clsp = P + o_cls;
    int64_t row_begin = 0, mt_begin = 0, tile_begin = 0, S = 0;
    tilecum[0] = 0;
    // Tile order.  Default: m-tile major, the n-tiles of an m-tile adjacent (they share the A rows).  With several n-tiles and weights that
    // do not fit an XCD's L2 next to the streaming A rows (> 2 MB: L5-L7), n-tile major: the workgroups of an XCD own consecutive ranges
    // = consecutive tiles, so an XCD then works on ONE n-tile's half of the weights for most of the launch.  (Workgroups of a stream-K launch
    // sit at different K positions of their tiles, so unlike the one-tile-per-workgroup kernel nothing else keeps the weight reads of an XCD
    // together: measured fabric traffic 2*FETCH+WRITE of the 128x128 launches 481 MB against 99 MB algorithmic before this.)
    // (from 64 row tiles up, like the row-major order below: on small launches neither ordering buys anything)
    const bool ntmajor = ncls == 1 && nnb > 1 && (int64_t)gs[0]->Cout * gs[0]->Tw * gs[0]->Cin * esz > (int64_t)g_sk_ntmajor_bytes &&
                         cdiv64((int64_t)gs[0]->B * gs[0]->Ho * gs[0]->Wo, bm) >= 64;
    for (int c = 0; c < ncls; ++c) {
        const sdt_conv_geom& g = *gs[c];
        const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
        const int nmb = (int)cdiv64(M, bm);
        const int nkc = g.Cin * esz / 128;
        int dymin = g.dy[0], dymax = g.dy[0];
        for (int t = 1; t < g.ntaps; ++t) dymin = std::min(dymin, g.dy[t]), dymax = std::max(dymax, g.dy[t]);
        const int nd = dymax - dymin + 1;
        // tap order: by the residue of the input row a tap reads on layers whose weights fit an L2 (see conv.hip), table order otherwise
        const bool rotate = (unsigned)g.Cout * (unsigned)g.Tw * (unsigned)g.Cin * (unsigned)esz <= (3u << 19);  // <= 1.5 MB of weights
        // taps must be sorted by dy for the rotation to be a cyclic shift of the table: true for every geometry ops.py builds
        bool sorted = true;
        for (int t = 1; t < g.ntaps; ++t) sorted = sorted && g.dy[t] >= g.dy[t - 1];
        // Row order of the class.  Natural: (b, oy, ox).  Row-major over oy -- (oy, b, ox) -- when that removes >= 5 % of the K steps
        // (measured: at 3 % the 20-row layers lose more to the scattered reads of a tile than the culled steps give back):
        // a tile then holds output rows of ONE image row (of several items), so its live-tap mask is that image row's valid taps and
        // the taps that fall off the top / bottom edge are culled for the whole tile instead of being multiplied as zeros.  It pays on
        // the short images: the (6, 3) valid-convolution's input gradient (10 image rows, 1 to 5 of the 6 kernel rows valid: -28 % steps),
        // the 3x3 layers on 10-row images (-6 %).  A group's rows stay runs of Wo consecutive rows (the statistics epilogues need >= 32).
        auto decode = [&](int64_t m, bool perm, int& b, int& oy, int& ox) {
            if (perm) {
                const int64_t per = (int64_t)g.B * g.Wo;
                oy = (int)(m / per);
                const int64_t q = m - (int64_t)oy * per;
                b = (int)(q / g.Wo), ox = (int)(q % g.Wo);
            } else {
                ox = (int)(m % g.Wo);
                const int64_t tq = m / g.Wo;
                oy = (int)(tq % g.Ho), b = (int)(tq / g.Ho);
            }
        };
        auto tile_mask = [&](int mt, bool perm) {
            unsigned mask = 0;
            int b, oy, ox;
            for (int rr = 0; rr < bm; ++rr) {
                const int64_t m = (int64_t)mt * bm + rr;
                if (m >= M) break;
                decode(m, perm, b, oy, ox);
                for (int t = 0; t < g.ntaps; ++t)
                    if ((unsigned)(oy * g.sy + g.dy[t]) < (unsigned)g.Hi && (unsigned)(ox * g.sx + g.dx[t]) < (unsigned)g.Wi) mask |= 1u << t;
            }
            return mask;
        };
        bool perm = false;
        if (g.Hi > 1 && g.ntaps > 4 && g.Wo >= 32 && g.Ho > 1 && nmb >= 64) {  // small launches are latency-bound: nothing to gain
            int64_t s_nat = 0, s_perm = 0;
            for (int mt = 0; mt < nmb; ++mt) s_nat += __builtin_popcount(tile_mask(mt, false)), s_perm += __builtin_popcount(tile_mask(mt, true));
            perm = s_perm * 100 <= s_nat * (int64_t)(rows_per_group < 0 ? g_sk_perm_pct_bwd : g_sk_perm_pct);
#ifdef SDT_TUNING
            if (getenv("SDT_SK_PLAN_LOG"))
                fprintf(stderr, "plan B%d %dx%d -> %dx%d Cin %d Cout %d taps %d rpg %d bwd_groups %d: steps natural %ld, row-major %ld -> %s\n", g.B, g.Hi, g.Wi,
                        g.Ho, g.Wo, g.Cin, g.Cout, g.ntaps, rows_per_group, bwd_groups, (long)s_nat, (long)s_perm, perm ? "row-major" : "natural");
#endif
        }
        for (int mt = 0; mt < nmb; ++mt) {
            unsigned mask = 0;
            for (int rr = 0; rr < bm; ++rr) {
                const int64_t m = (int64_t)mt * bm + rr;
                int* ri = rowinfo + (row_begin + m) * 4;
                if (m >= M) {
                    ri[0] = 0;
                    ri[1] = -1;  // every tap invalid
                    ri[2] = (int)SK_OOB;
                    ri[3] = -1;
                    continue;
                }
                int b, oy, ox;
                decode(m, perm, b, oy, ox);
                const int64_t mnat = ((int64_t)b * g.Ho + oy) * g.Wo + ox;
                const int iy0 = oy * g.sy, ix0 = ox * g.sx;
                ri[0] = (int)((uint32_t)((((int64_t)b * g.Hi + iy0) * g.Wi + ix0) * g.Cin * esz));
                unsigned inval = 0x80000000u;  // bit t: tap t reads outside X for this row; bit 31: always set (the "loader off" position)
                ri[2] = (int)((((int64_t)b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout * ysz);  // byte offset of the output row in Y
                ri[3] = rows_per_group > 0 ? (int)(mnat / rows_per_group) : (bwd_groups == 1 ? 0 : b);
                for (int t = 0; t < g.ntaps; ++t) {
                    if ((unsigned)(iy0 + g.dy[t]) < (unsigned)g.Hi && (unsigned)(ix0 + g.dx[t]) < (unsigned)g.Wi) mask |= 1u << t;
                    else inval |= 1u << t;
                }
                ri[1] = (int)inval;
            }
            if (g.Hi == 1 || g.ntaps <= 4) mask = (1u << g.ntaps) - 1u;  // as conv.hip: no culling on short tap lists (a dead tap costs zeros, not wrong results)
            int rot = 0;
            if (rotate && sorted) {
                int b_, oy0, ox_;
                decode((int64_t)mt * bm, perm, b_, oy0, ox_);
                const int b0 = (oy0 * g.sy) % nd;
                // first key in rotated order is 0: taps with (b0 + dy - dymin) mod nd == 0 ... i.e. dy - dymin == (nd - b0) mod nd
                const int want = (nd - b0) % nd;
                rot = g.ntaps;  // none >= want: start from the table's beginning
                for (int t = 0; t < g.ntaps; ++t)
                    if (g.dy[t] - dymin >= want) {
                        rot = t;
                        break;
                    }
                if (rot == g.ntaps) rot = 0;
            }
            // the kernel walks the live taps in the order rot, rot+1, .., ntaps-1, 0, .., rot-1: store the mask rotated accordingly
            unsigned rmask = 0;
            for (int i = 0; i < g.ntaps; ++i)
                if (mask >> ((rot + i) % g.ntaps) & 1u) rmask |= 1u << i;
            tileinfo[(mt_begin + mt) * 2] = (int)rmask;
            tileinfo[(mt_begin + mt) * 2 + 1] = rot;
            const int live = __builtin_popcount(mask) * nkc;
            // a tile without a live K step would never be visited by a range that ends exactly at it, and its outputs (zeros) would
            // stay unwritten: such packs -- k5 s2 input gradients with unreachable trailing rows -- go to the 64x64 kernel
            SDT_CHECK_ARG(live > 0, "a tile of this geometry has no live K step");
            if (ntmajor) {
                tilecum[mt + 1] = live;  // per-m-tile step counts for now: the prefix sums in n-tile-major order follow the loop
            } else {
                for (int nt = 0; nt < nnb; ++nt) {
                    const int64_t tile = tile_begin + (int64_t)mt * nnb + nt;
                    S += live;
                    tilecum[tile + 1] = (int)S;
                }
            }
        }
        if (ntmajor) {
            std::vector<int> live(nmb);
            for (int mt = 0; mt < nmb; ++mt) live[mt] = tilecum[mt + 1];
            for (int nt = 0; nt < nnb; ++nt)
                for (int mt = 0; mt < nmb; ++mt) {
                    S += live[mt];
                    tilecum[(int64_t)nt * nmb + mt + 1] = (int)S;
                }
        }
        int* cp = clsp + c * SK_CLS_INTS;
        cp[0] = g.Hi, cp[1] = g.Wi, cp[2] = g.Cin, cp[3] = g.Cout, cp[4] = g.ntaps, cp[5] = nkc;
        cp[6] = (int)tile_begin, cp[7] = nmb, cp[8] = (int)row_begin, cp[9] = (int)mt_begin, cp[10] = g.Tw;
        for (int t = 0; t < SDT_MAX_TAPS; ++t) {
            const bool on = t < g.ntaps;
            cp[11 + t] = on ? (g.dy[t] * g.Wi + g.dx[t]) * g.Cin * esz : 0;
            cp[11 + SDT_MAX_TAPS + t] = on ? ((g.dy[t] & 0xffff) | (g.dx[t] << 16)) : 0;
            cp[11 + 2 * SDT_MAX_TAPS + t] = on ? g.wt[t] * g.Cin * esz : 0;
        }
        row_begin += (int64_t)nmb * bm;
        mt_begin += nmb;
        tile_begin += (int64_t)nmb * nnb;
    }
    // every range needs at least one step (an empty range between a tile's owner and its last contributor would be waited for and never
    // publish); fp32 launches below 4 steps per range go to the 64x64 kernel of conv.hip (faster there, and the B = 4 fixtures keep their
    // recorded LeakyReLU decisions), bf16 launches have no other kernel and take the persistent one down to one step per range
    // (the bf16-shaped kernel asks for 4 steps per range as well: below that the launch is latency-bound and the round-4 128-row kernel, whose
    // plan the caller builds next, cuts it into twice as many ranges)
    SDT_CHECK_ARG(S >= ((esz == 2 && !is_bf2(esz)) ? 1 : 4) * (int64_t)G && S < (1ll << 31) / 2 / G, "step count out of range for the stream-K split (every range needs work)");
    SDT_CHECK_ARG(T < (1ll << 30), "too many tiles");
    // the bf16-shaped kernel wants at least one 256-row tile per CU: with fewer, every tile is cut between several workgroups and the slab hand-offs
    // of 256 x 256 partial tiles cost more than the K loops (L5 - L7 at 32 clips: 67 / 67 / 32 tiles; the 128-row kernel takes those)
    SDT_CHECK_ARG(!is_bf2(esz) || 2 * T >= G, "too few tiles for one workgroup per CU");
    // first tile of every range: the tile that contains step floor(r * S / G)
    int64_t tile = 0;
    for (int r = 0; r < G; ++r) {
        const int64_t s0 = (int64_t)r * S / G;
        while (tile + 1 < T && tilecum[tile + 1] <= s0) ++tile;
        range_tile[r] = (int)tile;
    }
    // P[3]: grid | workgroups per CU << 16 | (X / W are bf16) << 24 | (Y is bf16) << 25
    P[0] = SK_MAGIC, P[1] = bm, P[2] = bn, P[3] = G | (g_sk_wpc << 16) | ((esz == 2) << 24) | ((ysz == 2) << 25) | ((int)is_x3(esz) << 26), P[4] = ncls, P[5] = nnb | ((int)ntmajor << 16), P[6] = (int)T, P[7] = (int)S, P[8] = (int)rows, P[9] = (int)mts;
    P[10] = (int)o_row, P[11] = (int)o_ti, P[12] = (int)o_cum, P[13] = (int)o_rt, P[14] = (int)o_cls, P[15] = (int)(o_cls + (int64_t)ncls * SK_CLS_INTS);
    return SDT_OK;
}
extern "C" int sdt_convsk_plan_build(const sdt_conv_geom* geoms, int ncls, int rows_per_group, int bwd_groups, void* out, int64_t out_bytes) {
    return plan_build(geoms, ncls, rows_per_group, bwd_groups, out, out_bytes, 4, 4);
}
extern "C" int sdt_convsk_plan_build_t(const sdt_conv_geom* geoms, int ncls, int rows_per_group, int bwd_groups, int x_dtype, int y_dtype, void* out,
                                       int64_t out_bytes) {
    return plan_build(geoms, ncls, rows_per_group, bwd_groups, out, out_bytes, dtype_bytes(x_dtype), dtype_bytes(y_dtype));
}

template <typename TX, typename TY, int BM, int BN, int EPI, int WPC>
static void sk_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, int rpg, const sk_norm_bwd& nb,
                      hipStream_t s) {
    const size_t lds = (size_t)(2 * (BM + BN) * SK_LDP) * 4 + (size_t)BM * 8 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)convsk_kernel<TX, TY, BM, BN, EPI, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((convsk_kernel<TX, TY, BM, BN, EPI, WPC>), dim3(A.G), dim3(256), lds, s, (const TX*)x, (const TX*)w, bias, (TY*)y, A, stats, rpg, nb);
}

// kernel arguments from a plan blob (host copy: header + class tables; device copy: row / tile tables)
static int sk_fill_args(sk_args& A, sk_norm_bwd& nb, const void* plan_host, const void* plan_dev, void* workspace, unsigned epoch,
                        const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes, int64_t ybytes) {
    const int* P = (const int*)plan_host;
    SDT_CHECK_ARG(P[0] == SK_MAGIC, "not a conv plan");
    SDT_CHECK_ARG(xbytes > 0 && wbytes > 0 && ybytes > 0 && xbytes < (1ll << 31) - 65536 && wbytes < (1ll << 31) - 65536 && ybytes < (1ll << 31) - 65536,
                  "tensor sizes out of range");
    A.G = P[3] & 0xffff, A.ncls = P[4], A.nnb = P[5] & 0xffff, A.ntmajor = P[5] >> 16, A.T = P[6], A.S = P[7];
    const int* D = (const int*)plan_dev;
    A.rowinfo = (const int4*)(D + P[10]);
    A.tileinfo = (const int2*)(D + P[11]);
    A.tilecum = D + P[12];
    A.range_tile = D + P[13];
    SDT_CHECK_ARG(P[10] % 4 == 0 && P[11] % 2 == 0, "plan tables misaligned");
    for (int c = 0; c < A.ncls; ++c) {
        const int* cp = P + P[14] + c * SK_CLS_INTS;
        sk_class& k = A.cls[c];
        k.Hi = cp[0], k.Wi = cp[1], k.Cin = cp[2], k.Cout = cp[3], k.ntaps = cp[4], k.nkc = cp[5];
        k.tile_begin = cp[6], k.nmb = cp[7], k.row_begin = cp[8], k.mt_begin = cp[9], k.Tw = cp[10];
        for (int t = 0; t < SDT_MAX_TAPS; ++t) k.ashift[t] = cp[11 + t], k.dyx[t] = cp[11 + SDT_MAX_TAPS + t], k.bshift[t] = cp[11 + 2 * SDT_MAX_TAPS + t];
    }
    for (int c = A.ncls; c < SK_MAXC; ++c) A.cls[c] = A.cls[0];
    A.xbytes = (unsigned)xbytes, A.wbytes = (unsigned)wbytes, A.ybytes = (unsigned)ybytes;
    char* ws = (char*)workspace;
    A.slabs = (float*)ws;
    A.flags = ws ? (unsigned*)(ws + (size_t)SK_SLAB_BYTES) : nullptr;
    A.err = ws ? A.flags + 512 : nullptr;
    A.epoch = epoch;
    A.spin_limit = g_sk_spin_limit;
    A.korder = g_sk_korder;
    nb = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    if (nbw) {
        SDT_CHECK_ARG(nbw->y && nbw->mean && nbw->rstd && nbw->sums, "null pointer in sdt_norm_bwd");
        nb = {nbw->y, nbw->mean, nbw->rstd, nbw->gamma, nbw->beta, nbw->sums, nbw->slope, nbw->groups};
    }
    return SDT_OK;
}

// One launch of the persistent stream-K conv.
//   plan_host : the blob sdt_convsk_plan_build(_t) wrote (host copy: the header and the class tables travel as kernel arguments)
//   plan_dev  : the same blob in device memory (row / tile tables are read from there)
//   workspace : sdt_convsk_workspace_bytes() bytes, zero-filled once.  epoch: any value >= 1 (the flag value of this launch; a flag is lowered
//               again by the workgroup that consumed it, so every flag is zero between launches and the SAME epoch may be passed every time --
//               a launch recorded into a hipGraph replays correctly)
//   stats != NULL: forward statistics as sdt_conv_taps_stats_f32 (plan built with rows_per_group > 0)
//   nbw   != NULL: normalisation-backward statistics as sdt_conv_taps_multi_f32 (plan built with rows_per_group = -1, bwd_groups)
// Element types of x / w / nbw->y and of y: the ones the plan was built for (sdt_convsk_plan_build: fp32; _t: fp32 or bf16).
static int sk_go(const void* x, const void* w, const float* bias, void* y, const void* plan_host, const void* plan_dev, void* workspace,
                 unsigned epoch, double* stats, const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes, int64_t ybytes, void* stream, int want_x,
                 int want_y, int w3 = 0) {
    SDT_CHECK_ARG(x && w && y && plan_host && plan_dev && workspace, "null pointer");
    SDT_CHECK_ARG(epoch >= 1, "epoch must be >= 1");
    SDT_CHECK_ARG(!(stats && nbw), "forward and backward statistics are exclusive");
    SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)workspace | (uintptr_t)plan_dev) % 16) == 0, "operands must be 16-byte aligned");
    sk_args A;
    sk_norm_bwd nb;
    int rc = sk_fill_args(A, nb, plan_host, plan_dev, workspace, epoch, nbw, xbytes, wbytes, ybytes);
    if (rc) return rc;
    const int* P = (const int*)plan_host;
    const int bm = P[1], bn = P[2];
    hipStream_t s = (hipStream_t)stream;
    const int epi = stats ? 1 : (nbw ? 2 : 0);
    const int wpc = (P[3] >> 16) & 0xff;  // workgroups per CU the plan was built for (its grid may leave reserved slots free)
    const int xbf = (P[3] >> 24) & 1, ybf = (P[3] >> 25) & 1;
    SDT_CHECK_ARG(xbf == want_x && ybf == want_y, "the plan was built for other element types than this entry point's");
    SDT_CHECK_ARG((wpc == 1 || wpc == 2) && A.G > 0 && A.G <= 256 * wpc && A.G % 8 == 0, "plan built for an unknown grid");
#define SK_GO(TX_, TY_, BM_, BN_, WPC_)                                                              \
    do {                                                                                              \
        if (epi == 0) sk_launch<TX_, TY_, BM_, BN_, 0, WPC_>(x, w, bias, y, A, stats, 0, nb, s);      \
        else if (epi == 1) sk_launch<TX_, TY_, BM_, BN_, 1, WPC_>(x, w, bias, y, A, stats, 0, nb, s); \
        else sk_launch<TX_, TY_, BM_, BN_, 2, WPC_>(x, w, bias, y, A, stats, 0, nb, s);               \
    } while (0)
    SDT_CHECK_ARG(!w3 || (!xbf && !ybf && ((P[3] >> 26) & 1)), "pre-split weights need a split-fp32 plan (sdt_convsk_set_f32_split)");
    if (!xbf && !ybf && ((P[3] >> 26) & 1)) {
        rc = convx3_launch(x, w, bias, y, A, stats, nb, bm, bn, epi, w3, s);
        SDT_CHECK_ARG(rc == SDT_OK, "plan with a tile shape the split-fp32 kernel is not built for");
    } else if (!xbf && !ybf) {
        if (bm == 128 && bn == 128 && wpc == 1) SK_GO(float, float, 128, 128, 1);
        else if (bm == 128 && bn == 128 && wpc == 2) SK_GO(float, float, 128, 128, 2);
        else if (bm == 256 && bn == 64 && wpc == 1) SK_GO(float, float, 256, 64, 1);
        else if (bm == 128 && bn == 64 && wpc == 2) SK_GO(float, float, 128, 64, 2);
        else SDT_CHECK_ARG(false, "plan with an unknown tile shape");
    } else if (xbf && ybf) {
        if (wpc == 1) {
            rc = convbf2_launch(x, w, bias, y, A, stats, nb, bm, bn, epi, s);
            SDT_CHECK_ARG(rc == SDT_OK, "plan with a tile shape the bf16-shaped kernel is not built for");
        } else if (bm == 128 && bn == 128 && wpc == 2) SK_GO(__bf16, __bf16, 128, 128, 2);
        else if (bm == 128 && bn == 64 && wpc == 2) SK_GO(__bf16, __bf16, 128, 64, 2);
        else SDT_CHECK_ARG(false, "plan with a tile shape the bf16 kernels are not built for");
    } else {
        SDT_CHECK_ARG(false, "mixed element types (bf16 operands, fp32 output or the reverse) are not built");
    }
#undef SK_GO
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
extern "C" int sdt_convsk_f32(const float* x, const float* w, const float* bias, float* y, const void* plan_host, const void* plan_dev,
                              void* workspace, unsigned epoch, double* stats, const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes,
                              int64_t ybytes, void* stream) {
    return sk_go(x, w, bias, y, plan_host, plan_dev, workspace, epoch, stats, nbw, xbytes, wbytes, ybytes, stream, 0, 0);
}
// split-fp32 plan, weights pre-split into three bf16 planes (sdt_wt_desc.planes = 3): wbytes is the size of the fp32 weight tensor they came from
extern "C" int sdt_convsk_f32_w3(const float* x, const void* w3, const float* bias, float* y, const void* plan_host, const void* plan_dev,
                                 void* workspace, unsigned epoch, double* stats, const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes,
                                 int64_t ybytes, void* stream) {
    return sk_go(x, w3, bias, y, plan_host, plan_dev, workspace, epoch, stats, nbw, xbytes, wbytes, ybytes, stream, 0, 0, 1);
}
// bf16 tensors (x, w, y, nbw->y), fp32 bias / statistics / accumulation: the bf16-storage path of BASELINE config 4
extern "C" int sdt_convsk_bf16(const void* x, const void* w, const float* bias, void* y, const void* plan_host, const void* plan_dev,
                               void* workspace, unsigned epoch, double* stats, const sdt_norm_bwd* nbw, int64_t xbytes, int64_t wbytes,
                               int64_t ybytes, void* stream) {
    return sk_go(x, w, bias, y, plan_host, plan_dev, workspace, epoch, stats, nbw, xbytes, wbytes, ybytes, stream, 1, 1);
}

// ---- weight gradient through the stream-K machinery (esz: bytes per element of x / dy: 4 -> convsk_dw_kernel, 32 rows per K step;
// 2 -> convbf_dw_kernel, 64 rows per K step)
// Tile of a weight gradient (rows = output channels, columns = (tap, input channel) pairs): by divisibility.  sdt_convsk_set_dw_wide_tiles(1) lets the
// split-fp32 kernel (convx3_dw_kernel) take 128-wide column tiles with a RAGGED last one where taps * Cin = 64 (mod 128) -- L2: 9 x 64 = 576 columns in
// five tiles, the last half empty, instead of nine 64-wide ones: 279 -> 248 us at 32 clips (profiles/r06_dw_wide_ab.txt).  Off by default: 0.5 % of
// a step, and the re-cut K chunks regroup the sums behind the calibrated margins.  (64 x 256 tiles for the 64-channel layer L1 were measured too:
// 256 registers + spills, 274 -> 279 us; not kept.)
static void dw_tile(const sdt_conv_geom& g, int esz, int& bm, int& bn) {
    const int N = g.ntaps * g.Cin;
    bm = g.Cout % 128 == 0 ? 128 : 64;
    bn = N % 128 == 0 ? 128 : 64;
    if (esz == 4 && g_sk_split && g_sk_wpc == 2 && g_dw_wide && N >= 128 && bn == 64) {
        // ... unless the fewer, wider tiles would leave the K loop (32 rows per step) too short for its G / T chunks of >= 8 steps (dw_supported):
        // the wide rule never takes a launch away from this kernel
        const int64_t K = cdiv64((int64_t)g.B * g.Ho * g.Wo, 32), T = (int64_t)(g.Cout / bm) * cdiv64(N, 128);
        const int G = sk_grid(g_sk_wpc);
        if (T <= G && K >= 8 * (G / T)) bn = 128;
    }
}
static int dw_supported(const sdt_conv_geom* g, int esz) {
    if (!g || (esz != 4 && esz != 2)) return 0;
    if (esz == 2 && g_sk_wpc != 2) return 0;
    const sdt_conv_geom* gs[1] = {g};
    if (!sk_supported(gs, 1, 128, 64, esz)) return 0;  // Cin, sizes
    if (g->Cout % 64 != 0 || (g->ntaps * g->Cin) % 64 != 0 || g->Cin % 64 != 0 || g->ntaps != g->Tw) return 0;
    if (g->osy != 1 || g->osx != 1 || g->ooy != 0 || g->oox != 0 || g->Hy != g->Ho || g->Wy != g->Wo) return 0;  // dense dY
    for (int t = 0; t < g->ntaps; ++t)
        for (int u = 0; u < t; ++u)
            if (g->wt[t] == g->wt[u]) return 0;
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
    const int step = esz == 4 ? 32 : 64;
    int bm, bn;
    dw_tile(*g, esz, bm, bn);
    const int64_t K = cdiv64(M, step), T = (int64_t)(g->Cout / bm) * cdiv64((int64_t)g->ntaps * g->Cin, bn);
    const int G = sk_grid(g_sk_wpc);
    return T <= G && K >= (esz == 4 ? 8 : 4) * (G / T) && T * K < (1ll << 31) / G ? 1 : 0;  // G / T chunks of the K loop per tile, >= 8 (4) steps each
}
extern "C" int sdt_convsk_dw_supported(const sdt_conv_geom* g) { return dw_supported(g, 4); }
extern "C" int sdt_convsk_dw_supported_t(const sdt_conv_geom* g, int dtype) { return dw_supported(g, dtype_bytes(dtype)); }
// plan of a weight gradient: header + per-row table of the forward geometry (rows padded to a multiple of the K step + three extra steps)
static int64_t dw_plan_bytes(const sdt_conv_geom* g, int esz) {
    if (!dw_supported(g, esz)) return -1;
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
    const int step = esz == 4 ? 32 : 64;
    const int64_t rows = (cdiv64(M, step) + (esz == 4 ? 3 : 4)) * step;  // the bf16 kernel reads its row tables one step further ahead
    return (SK_HDR + rows * 4 + SK_CLS_INTS) * 4;
}
extern "C" int64_t sdt_convsk_dw_plan_bytes(const sdt_conv_geom* g) { return dw_plan_bytes(g, 4); }
extern "C" int64_t sdt_convsk_dw_plan_bytes_t(const sdt_conv_geom* g, int dtype) { return dw_plan_bytes(g, dtype_bytes(dtype)); }
extern "C" int64_t sdt_convsk_dw_workspace_bytes(void) { return (int64_t)512 * 128 * 128 * 4; }
static int dw_plan_build(const sdt_conv_geom* gp, void* out, int64_t out_bytes, int esz) {
    SDT_CHECK_ARG(dw_supported(gp, esz), "geometry not supported by the stream-K weight-gradient kernel");
    SDT_CHECK_ARG(out != nullptr && out_bytes >= dw_plan_bytes(gp, esz), "plan buffer too small");
    const sdt_conv_geom& g = *gp;
    int* P = (int*)out;
    const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
    const int step = esz == 4 ? 32 : 64;
    const int64_t K = cdiv64(M, step), rows = (K + (esz == 4 ? 3 : 4)) * step;
    int bm, bn;
    dw_tile(g, esz, bm, bn);
    const int ncol = (int)cdiv64((int64_t)g.ntaps * g.Cin, bn);
    const int64_t T = (int64_t)(g.Cout / bm) * ncol;
    const int G = sk_grid(g_sk_wpc);
    int* rowinfo = P + SK_HDR;
    for (int64_t m = 0; m < rows; ++m) {
        int* ri = rowinfo + m * 4;
        if (m >= M) {
            ri[0] = 0, ri[1] = -1, ri[2] = (int)SK_OOB, ri[3] = -1;
            continue;
        }
        const int ox = (int)(m % g.Wo), tq = (int)(m / g.Wo);
        const int oy = tq % g.Ho, b = tq / g.Ho;
        const int iy0 = oy * g.sy, ix0 = ox * g.sx;
        unsigned inval = 0x80000000u;
        for (int t = 0; t < g.ntaps; ++t)
            if (!((unsigned)(iy0 + g.dy[t]) < (unsigned)g.Hi && (unsigned)(ix0 + g.dx[t]) < (unsigned)g.Wi)) inval |= 1u << t;
        ri[0] = (int)((uint32_t)((((int64_t)b * g.Hi + iy0) * g.Wi + ix0) * g.Cin * esz));
        ri[1] = (int)inval;
        ri[2] = (int)(m * g.Cout * esz);
        ri[3] = 0;
    }
    int* cp = P + SK_HDR + rows * 4;
    cp[0] = g.Hi, cp[1] = g.Wi, cp[2] = g.Cin, cp[3] = g.Cout, cp[4] = g.ntaps, cp[5] = g.Cin * esz / 128;
    cp[6] = 0, cp[7] = 0, cp[8] = 0, cp[9] = 0, cp[10] = g.Tw;
    for (int t = 0; t < SDT_MAX_TAPS; ++t) {
        const bool on = t < g.ntaps;
        cp[11 + t] = on ? (g.dy[t] * g.Wi + g.dx[t]) * g.Cin * esz : 0;
        cp[11 + SDT_MAX_TAPS + t] = on ? g.wt[t] : 0;  // the weight-gradient kernels read wt[t] here
        cp[11 + 2 * SDT_MAX_TAPS + t] = 0;
    }
    P[0] = SK_MAGIC + 1, P[1] = bm, P[2] = bn, P[3] = G | (g_sk_wpc << 16) | ((esz == 2) << 24) | ((int)(esz == 4 && g_sk_split && g_sk_wpc == 2) << 26), P[4] = 1, P[5] = ncol, P[6] = (int)T, P[7] = (int)(T * K), P[8] = (int)rows, P[9] = (int)K;
    P[10] = SK_HDR, P[11] = 0, P[12] = 0, P[13] = 0, P[14] = (int)(SK_HDR + rows * 4), P[15] = (int)(SK_HDR + rows * 4 + SK_CLS_INTS);
    return SDT_OK;
}
extern "C" int sdt_convsk_dw_plan_build(const sdt_conv_geom* gp, void* out, int64_t out_bytes) { return dw_plan_build(gp, out, out_bytes, 4); }
extern "C" int sdt_convsk_dw_plan_build_t(const sdt_conv_geom* gp, int dtype, void* out, int64_t out_bytes) {
    return dw_plan_build(gp, out, out_bytes, dtype_bytes(dtype));
}

// dw (Cout, Tw, Cin) += the weight gradient of the geometry the plan was built for; workspace: sdt_convsk_dw_workspace_bytes() bytes,
// contents irrelevant (every slab that is read has been written by this launch).  Deterministic: fixed split, fixed summation order.
static int dw_go(const void* xv, const void* dyv, float* dw, const void* plan_host, const void* plan_dev, void* workspace, int64_t xbytes,
                 int64_t ybytes, void* stream, int want_bf) {
    const float* x = (const float*)xv;
    const float* dy = (const float*)dyv;
    SDT_CHECK_ARG(x && dy && dw && plan_host && plan_dev && workspace, "null pointer");
    const int* P = (const int*)plan_host;
    SDT_CHECK_ARG(P[0] == SK_MAGIC + 1, "not a weight-gradient plan");
    SDT_CHECK_ARG(((P[3] >> 24) & 1) == want_bf, "the plan was built for another element type than this entry point's");
    SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace | (uintptr_t)plan_dev) % 16) == 0, "operands must be 16-byte aligned");
    SDT_CHECK_ARG(xbytes > 0 && ybytes > 0 && xbytes < (1ll << 31) - 65536 && ybytes < (1ll << 31) - 65536, "tensor sizes out of range");
    sk_args A;
    A.G = P[3] & 0xffff, A.ncls = 1, A.nnb = P[5], A.ntmajor = 0, A.T = P[6], A.S = P[7];
    const int K = P[9], ncol = P[5];
    const int* D = (const int*)plan_dev;
    A.rowinfo = (const int4*)(D + P[10]);
    A.tileinfo = nullptr, A.tilecum = nullptr, A.range_tile = nullptr, A.slabs = nullptr, A.flags = nullptr, A.err = nullptr, A.epoch = 0, A.korder = 0;
    const int* cp = P + P[14];
    sk_class& k = A.cls[0];
    k.Hi = cp[0], k.Wi = cp[1], k.Cin = cp[2], k.Cout = cp[3], k.ntaps = cp[4], k.nkc = cp[5];
    k.tile_begin = 0, k.nmb = 0, k.row_begin = 0, k.mt_begin = 0, k.Tw = cp[10];
    for (int t = 0; t < SDT_MAX_TAPS; ++t) k.ashift[t] = cp[11 + t], k.dyx[t] = cp[11 + SDT_MAX_TAPS + t], k.bshift[t] = 0;
    for (int c = 1; c < SK_MAXC; ++c) A.cls[c] = A.cls[0];
    A.xbytes = (unsigned)xbytes, A.wbytes = 0, A.ybytes = (unsigned)ybytes;
    hipStream_t s = (hipStream_t)stream;
    const int bm = P[1], bn = P[2];
    const size_t lds = (size_t)2 * 32 * (bm + bn) * 4;
    const int nchunk = A.G / A.T;  // T * nchunk <= G workgroups have work (>= 98 % of them on this network's layers)
#define DW_GO(BM_, BN_, WPC_)                                                                                                  \
    do {                                                                                                                       \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            (void)hipFuncSetAttribute((const void*)convsk_dw_kernel<BM_, BN_, WPC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((convsk_dw_kernel<BM_, BN_, WPC_>), dim3(A.G), dim3(256), lds, s, x, dy, A, K, ncol, nchunk, (float*)workspace); \
    } while (0)
    const int wpc = (P[3] >> 16) & 0xff;
    if (want_bf) {
        const size_t ldsb = (size_t)2 * 64 * (bm + bn) * 2;
#define DWB_GO(BM_, BN_)                                                                                                       \
    do {                                                                                                                       \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            (void)hipFuncSetAttribute((const void*)convbf_dw_kernel<BM_, BN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((convbf_dw_kernel<BM_, BN_>), dim3(A.G), dim3(256), ldsb, s, (const __bf16*)xv, (const __bf16*)dyv, A, K, ncol, nchunk, \
                           (float*)workspace);                                                                                \
    } while (0)
        SDT_CHECK_ARG(wpc == 2, "the bf16 weight gradient is built for two workgroups per CU");
        if (bm == 128 && bn == 128) DWB_GO(128, 128);
        else if (bm == 128 && bn == 64) DWB_GO(128, 64);
        else if (bm == 64 && bn == 128) DWB_GO(64, 128);
        else DWB_GO(64, 64);
#undef DWB_GO
    } else if ((P[3] >> 26) & 1) {  // plan built with sdt_convsk_set_f32_split(1): split-fp32 products
        const size_t ldsx = (size_t)2 * 3 * 16 * (bm + bn) * 2;
#define DWX_GO(BM_, BN_)                                                                                                       \
    do {                                                                                                                       \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            (void)hipFuncSetAttribute((const void*)convx3_dw_kernel<BM_, BN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((convx3_dw_kernel<BM_, BN_>), dim3(A.G), dim3(256), ldsx, s, x, dy, A, K, ncol, nchunk, (float*)workspace); \
    } while (0)
        SDT_CHECK_ARG(wpc == 2, "the split-fp32 weight gradient is built for two workgroups per CU");
        if (bm == 128 && bn == 128) DWX_GO(128, 128);
        else if (bm == 128 && bn == 64) DWX_GO(128, 64);
        else if (bm == 64 && bn == 128) DWX_GO(64, 128);
        else DWX_GO(64, 64);
#undef DWX_GO
    } else if (bm == 128 && bn == 128 && wpc == 2) DW_GO(128, 128, 2);
    else if (bm == 128 && bn == 128) DW_GO(128, 128, 1);
    else if (bm == 128 && bn == 64 && wpc == 2) DW_GO(128, 64, 2);
    else if (bm == 128 && bn == 64) DW_GO(128, 64, 1);
    else if (bm == 64 && bn == 128 && wpc == 2) DW_GO(64, 128, 2);
    else if (bm == 64 && bn == 128) DW_GO(64, 128, 1);
    else if (bm == 64 && bn == 64 && wpc == 2) DW_GO(64, 64, 2);
    else DW_GO(64, 64, 1);
#undef DW_GO
    {
        const int blocks1 = A.T * (bm * bn / 4 / 256);  // workgroups at one lane per float4
        const int q = (blocks1 >= 512 || nchunk < 8) ? 1 : (blocks1 >= 256 || nchunk < 16) ? 2 : (blocks1 >= 128 || nchunk < 32) ? 4 : 8;
        if (q == 1) hipLaunchKernelGGL(dw_sk_reduce_kernel<1>, dim3(blocks1), dim3(256), 0, s, (const float*)workspace, dw, A, ncol, nchunk, k.Cout, k.Tw, bm, bn);
        else if (q == 2) hipLaunchKernelGGL(dw_sk_reduce_kernel<2>, dim3(blocks1 * 2), dim3(256), 0, s, (const float*)workspace, dw, A, ncol, nchunk, k.Cout, k.Tw, bm, bn);
        else if (q == 4) hipLaunchKernelGGL(dw_sk_reduce_kernel<4>, dim3(blocks1 * 4), dim3(256), 0, s, (const float*)workspace, dw, A, ncol, nchunk, k.Cout, k.Tw, bm, bn);
        else hipLaunchKernelGGL(dw_sk_reduce_kernel<8>, dim3(blocks1 * 8), dim3(256), 0, s, (const float*)workspace, dw, A, ncol, nchunk, k.Cout, k.Tw, bm, bn);
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
extern "C" int sdt_convsk_dw_f32(const float* x, const float* dy, float* dw, const void* plan_host, const void* plan_dev, void* workspace,
                                 int64_t xbytes, int64_t ybytes, void* stream) {
    return dw_go(x, dy, dw, plan_host, plan_dev, workspace, xbytes, ybytes, stream, 0);
}
// bf16 x / dy, fp32 gradient (ACCUMULATED into dw as above): the bf16-storage path
extern "C" int sdt_convsk_dw_bf16(const void* x, const void* dy, float* dw, const void* plan_host, const void* plan_dev, void* workspace,
                                  int64_t xbytes, int64_t ybytes, void* stream) {
    return dw_go(x, dy, dw, plan_host, plan_dev, workspace, xbytes, ybytes, stream, 1);
}
