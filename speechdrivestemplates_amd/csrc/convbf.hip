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

#define BF_BM 256
#define BF_NT 512
// BF_ABL (tools/debug/r05_bf2_ablation.sh; ablation builds compute WRONG results by design): 1 no global loads in the K loop, 2 no LDS stores,
// 4 no MFMAs, 8 no barrier in the K loop, 16 no fragment reads
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
#else
#define BF_TL(slot, val) do { } while (0)
#endif

// Epilogue of accumulator rows [TMB, TME) x all TN column blocks of one wave: branch-free bf16 stores (pairs of columns as one dword, exchanged
// between neighbouring lanes by DPP) and, EPI 1 / 2, the per-(group, channel) statistics from the fp32 accumulators.  sOut / sGrp: LDS, byte
// offset of each tile row in Y (SK_OOB: none) and its statistics group.  row0: first tile row of the wave, ncol0: first output channel of the wave.
template <int TM, int TN, int EPI, int TMB, int TME>
__device__ __forceinline__ void bf2_epilogue(f32x16 (&acc)[TM][TN], const int* sOut, const int* sGrp, const int row0, const int ncol0, const int lane,
                                             const float* __restrict__ bias, const __amdgpu_buffer_rsrc_t rsY, const int Cout,
                                             double* __restrict__ stats, const sk_norm_bwd& nb, const unsigned ybytes) {
    float yv[(TME - TMB) * TN * 16];
    if constexpr (EPI == 2) {
        // the forward output y of the block below at every position of these rows: all loads before anything else (one exposed latency)
        const __amdgpu_buffer_rsrc_t rsNY = __builtin_amdgcn_make_buffer_rsrc((void*)nb.y, 0, (int)ybytes, 0x00020000);
#pragma unroll
        for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const unsigned nb2 = (unsigned)(ncol0 + tn * 32 + (lane & 31)) * 2u;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int4 o4 = *(const int4*)&sOut[row0 + tm * 32 + 8 * qq + 4 * (lane >> 5)];
                    const int offs[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e] =
                            sk_bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(rsNY, (int)((unsigned)offs[e] + nb2), 0, 0));
                }
            }
    }
#pragma unroll
    for (int tm = TMB; tm < TME; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = ncol0 + tn * 32 + (lane & 31);
            const float bv = bias != nullptr ? bias[n] : 0.f;
            const bool odd = lane & 1;
            const unsigned nb2 = (unsigned)(n & ~1) * 2u;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int4 o4 = *(const int4*)&sOut[row0 + tm * 32 + 8 * qq + 4 * (lane >> 5)];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float v0 = acc[tm][tn][4 * qq + 2 * h] + bv, v1 = acc[tm][tn][4 * qq + 2 * h + 1] + bv;
                    const float give = odd ? v0 : v1;
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
                    sk_bf16x2 pk;
                    pk[0] = (__bf16)(odd ? got : v0);
                    pk[1] = (__bf16)(odd ? v1 : got);
                    const int ro = odd ? (h ? o4.w : o4.y) : (h ? o4.z : o4.x);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rsY, (int)((unsigned)ro + nb2), 0, 0);
                }
            }
            if constexpr (EPI == 1 || EPI == 2) {
                const int rb0 = row0 + tm * 32;
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
                            u = valid ? acc[tm][tn][4 * qq + e] + bv : 0.f;
                            v = u;
                        } else {
                            const float yvv = yv[((tm - TMB) * TN + tn) * 16 + 4 * qq + e];
                            v = (yvv - (second ? mu1 : mu0)) * (second ? rs1 : rs0);
                            u = valid ? acc[tm][tn][4 * qq + e] * act_grad(v * ga + be, nb.slope) : 0.f;
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
                s0 += __shfl_xor(s0, 32, 64);
                q0 += __shfl_xor(q0, 32, 64);
                s1 += __shfl_xor(s1, 32, 64);
                q1 += __shfl_xor(q1, 32, 64);
                double* acc_out = EPI == 1 ? stats : nb.sums;
                if (lane < 32 && gfirst >= 0) {
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
template <int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(BF_NT, 2) void convbf2_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ W, const float* __restrict__ bias,
                                                          __bf16* __restrict__ Y, const sk_args P, double* __restrict__ stats, const sk_norm_bwd nb) {
    constexpr int BM = BF_BM, TM = BM / WGM / 32, TN = BN / WGN / 32, RA = BM / 64, RB = BN / 64, NM = TM * TN, NF = TM + TN;
    constexpr int NSET = BN == 256 ? 1 : 2;  // staging register sets: two when the accumulators leave room (a load then has two steps to land)
    static_assert(WGM * WGN == 8 && TM >= 1 && TN >= 1 && RB >= 1, "wave grid");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                          // [2][BM * LDP]
    float* sB = smem + 2 * BM * SK_LDP;        // [2][BN * LDP]
    int* sOutB = (int*)(sB + 2 * BN * SK_LDP);  // [2][BM] byte offset of the tile's output rows in Y (SK_OOB, negative as int: none)
    int* sGrpB = sOutB + 2 * BM;                // [2][BM] statistics group of each row (EPI 1 / 2)
    int* sFlagOkp = sGrpB + 2 * BM;             // [4]
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
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)P.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)P.ybytes, 0x00020000);

    const int row0 = wm * (TM * 32), col0 = wn * (TN * 32);
    const int fa = (row0 + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int fb = (col0 + (lane & 31)) * SK_LDP + (lane >> 5) * 4;
    const int wofs = r0 * SK_LDP + kv * 4;

    // ---- the walk over the range's tiles
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

    // ---- loader state of the tile whose operands are being fetched
    int nkc = 1, ntaps = 1, rot = 0, kc = 0, left = 0, v_ash = 0, v_bsh = 0;
    unsigned rmask = 0u;
    unsigned abase[RA], inval[RA], bbase[RB];
    f32x4 ra[NSET][RA], rb[NSET][RB];
    // per-tile facts the end phase needs (set by `setup`, which runs BEFORE the previous tile's end phase when tiles are pipelined)
    struct TileFacts {
        int Cout, n0, a, b, tbeg, tend;
    };
    auto load = [&](auto SI) {
        constexpr int S_ = decltype(SI)::value;
        const bool on = left > 0 && rmask != 0u;
        int t = (rmask != 0u ? __builtin_ctz(rmask) : 0) + rot;
        t = t >= ntaps ? t - ntaps : t;
        t = on ? t : 0;
        const int ash = __builtin_amdgcn_readlane(v_ash, t), bsh = __builtin_amdgcn_readlane(v_bsh, t);
        const int cs = kc * 128;  // a K step is 128 bytes of a row
        const int sh = on ? 31 - t : 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const unsigned o = ((abase[i] + (unsigned)ash) & 0x7fffffffu) | ((inval[i] << sh) & 0x80000000u);
#if BF_ABL & 1
            ra[S_][i][0] = __uint_as_float(o + (unsigned)cs);
#else
            ra[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)o, cs, 0));
#endif
        }
        const unsigned bsh_eff = (unsigned)bsh + (on ? 0u : SK_OOB);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#if BF_ABL & 1
            rb[S_][i][0] = __uint_as_float(bbase[i] + bsh_eff + (unsigned)cs);
#else
            rb[S_][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(bbase[i] + bsh_eff), cs, 0));
#endif
        --left;
        const bool wrap = kc + 1 == nkc;
        kc = wrap ? 0 : kc + 1;
        rmask = wrap ? (rmask & (rmask - 1u)) : rmask;
    };
    auto stage = [&](auto SI, const int buf) {
        constexpr int S_ = decltype(SI)::value;
        float* wA = sA + buf * BM * SK_LDP + wofs;
        float* wB = sB + buf * BN * SK_LDP + wofs;
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
    // set-up of tile (c, mt, nt) covering live steps from `pos`: loader state, and the rows' output offsets / groups into sOut / sGrp[sbuf]
    auto setup = [&](const int sbuf) -> TileFacts {
        const sk_class& cl = P.cls[c];
        TileFacts f;
        nkc = __builtin_amdgcn_readfirstlane(cl.nkc);
        ntaps = __builtin_amdgcn_readfirstlane(cl.ntaps);
        f.Cout = __builtin_amdgcn_readfirstlane(cl.Cout);
        f.tbeg = __builtin_amdgcn_readfirstlane(P.tilecum[tile]);
        f.tend = __builtin_amdgcn_readfirstlane(P.tilecum[tile + 1]);
        f.a = pos - f.tbeg;
        f.b = min(s_end, f.tend) - f.tbeg;
        const int2 ti = P.tileinfo[cl.mt_begin + mt];
        const int m0 = cl.row_begin + mt * BM;
        f.n0 = nt * BN;
        v_ash = lane < SDT_MAX_TAPS ? cl.ashift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
        v_bsh = lane < SDT_MAX_TAPS ? cl.bshift[lane < SDT_MAX_TAPS ? lane : 0] : 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int2 ri = ((const int2*)P.rowinfo)[2 * (m0 + r0 + 64 * i)];  // {X byte offset, invalid-tap mask}
            abase[i] = (unsigned)ri.x + (unsigned)kv * 16u;
            inval[i] = (unsigned)ri.y;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) bbase[i] = (unsigned)((f.n0 + r0 + 64 * i) * cl.Tw * cl.Cin) * 2u + (unsigned)kv * 16u;  // W is (N, Tw, Cin) bf16
        if (tid < BM) {
            const int2 ro = ((const int2*)P.rowinfo)[2 * (m0 + tid) + 1];  // {Y byte offset, statistics group}
            sOutB[sbuf * BM + tid] = ro.x;
            sGrpB[sbuf * BM + tid] = ro.y;
        }
        rmask = (unsigned)__builtin_amdgcn_readfirstlane(ti.x);
        rot = __builtin_amdgcn_readfirstlane(ti.y);
        for (int skip = f.a / nkc; skip > 0; --skip) rmask &= rmask - 1;
        kc = f.a - (f.a / nkc) * nkc;
        left = f.b - f.a;
        return f;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x16 acc[TM][TN];
    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
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

    // one K step on LDS buffer `cur`: fragments of k-group 0 are in a0 / b0 on entry (and on exit, for the next step)
    auto step = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value, nx = cur ^ 1;
        typedef std::integral_constant<int, (NSET == 2 ? nx : 0)> SETN;
        const float* pa = sA + cur * BM * SK_LDP + fa;
        const float* pb = sB + cur * BN * SK_LDP + fb;
        BF_READ(a1, b1, pa, pb, 1);
        stage(SETN{}, nx);  // step s + 1: registers -> LDS[next]
        load(SETN{});       // step s + 1 + NSET -> the registers just staged
        BF_MFMAS(a0, b0);
        BF_READ(a0, b0, pa, pb, 2);
        BF_MFMAS(a1, b1);
        BF_READ(a1, b1, pa, pb, 3);
        BF_MFMAS(a0, b0);
        sk_bf_interleave<0, 0, NM, NF, RA + RB>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of LDS[cur] and my writes of LDS[next] are done
#if !(BF_ABL & 8)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        BF_READ(a0, b0, sA + nx * BM * SK_LDP + fa, sB + nx * BN * SK_LDP + fb, 0);
        BF_MFMAS(a1, b1);
#pragma unroll
        for (int q = 0; q < (NM < NF ? NM : NF); ++q) { SK_SGB(0x8, 1); SK_SGB(0x100, 1); }
        if constexpr (NM > NF) SK_SGB(0x8, NM - NF);
        __builtin_amdgcn_sched_barrier(0);
    };

    // LDS hand-over between waves WITHOUT draining the global loads in flight (__syncthreads() waits for vmcnt(0) as well)
#define BF_LDS_BARRIER()                      \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    // ------------------------------------------------------------------ first tile: set-up + the first requests
    int sbuf = 0;
    TileFacts F = setup(0);
    load(I0{});
    for (int seg = 0;; ++seg) {
        BF_TL(0, wall_clock64());
        // ---- pipeline fill: step 0 -> LDS[0]; steps 1 (and 2) -> registers
        BF_LDS_BARRIER();  // the previous tile's LDS tiles are no longer read by anybody
        stage(I0{}, 0);
        if constexpr (NSET == 2) {
            load(I1{});
            load(I0{});
        } else {
            load(I0{});
        }
        BF_LDS_BARRIER();  // LDS[0] and sOut / sGrp[sbuf] are written (the requests just issued stay in flight)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        BF_READ(a0, b0, sA + fa, sB + fb, 0);
        const int nsteps = F.b - F.a;
        BF_TL(1, wall_clock64());
        BF_TL(5, (unsigned long long)nsteps);
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            step(I0{});
            step(I1{});
        }
        if (s < nsteps) step(I0{});
        BF_TL(2, wall_clock64());

        // ---- the tile's facts for the end phase; then the NEXT tile's set-up and first request, before the end phase
        const TileFacts E = F;
        const int ebuf = sbuf;
        const bool owner = E.a == 0, whole = owner && E.b == E.tend - E.tbeg;
        const int ntile_pos = E.tbeg + E.b;
        const bool more = ntile_pos < s_end;
        if (more) {
            pos = ntile_pos;
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
            sbuf ^= 1;
            F = setup(sbuf);  // (writes sOut / sGrp[sbuf]: the other buffer than the one the end phase below reads)
            load(I0{});       // first K step of the next tile: lands under the end phase
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
            if (tid == 0) __hip_atomic_store((gu32*)(P.flags + r), P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            if constexpr (EPI == 2 && TM > 2) {  // the reads of y for half of the rows at a time: 64 instead of 128 registers
                bf2_epilogue<TM, TN, EPI, 0, TM / 2>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes);
                bf2_epilogue<TM, TN, EPI, TM / 2, TM>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes);
            } else {
                bf2_epilogue<TM, TN, EPI, 0, TM>(acc, sOut, sGrp, row0, E.n0 + col0, lane, bias, rsY, E.Cout, stats, nb, P.ybytes);
            }
        }
        BF_TL(4, wall_clock64());
        if (!more) break;
    }
#undef BF_LDS_BARRIER
#undef BF_READ
#undef BF_MFMAS
#undef sFlagOk
}

template <int BN, int WGM, int WGN, int EPI>
static void bf2_launch_one(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, hipStream_t s) {
    const size_t lds = (size_t)(2 * (BF_BM + BN) * SK_LDP) * 4 + (size_t)4 * BF_BM * 4 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)convbf2_kernel<BN, WGM, WGN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((convbf2_kernel<BN, WGM, WGN, EPI>), dim3(A.G), dim3(BF_NT), lds, s, (const __bf16*)x, (const __bf16*)w, bias, (__bf16*)y, A, stats, nb);
}

int convbf2_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, int bn, int epi,
                   hipStream_t s) {
#define BF2_GO(BN_, WGM_, WGN_)                                                              \
    do {                                                                                      \
        if (epi == 0) bf2_launch_one<BN_, WGM_, WGN_, 0>(x, w, bias, y, A, stats, nb, s);     \
        else if (epi == 1) bf2_launch_one<BN_, WGM_, WGN_, 1>(x, w, bias, y, A, stats, nb, s); \
        else bf2_launch_one<BN_, WGM_, WGN_, 2>(x, w, bias, y, A, stats, nb, s);              \
    } while (0)
    if (bn == 256) BF2_GO(256, 2, 4);
    else if (bn == 128) BF2_GO(128, 4, 2);
    else if (bn == 64) BF2_GO(64, 4, 2);
    else return SDT_ERR_ARG;
#undef BF2_GO
    return SDT_OK;
}
