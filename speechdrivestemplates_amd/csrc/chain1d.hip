// The generator's Conv1d stage (U-Net + decoder: generator.py:53-85,96-103 -- sixteen ConvNormRelu('1d') blocks on 64 frames x 256 channels per
// clip) as ONE persistent launch per direction instead of two to three launches per block.
//
// Per block the GEMM is tiny (M = B*T <= 2048 rows, 0.8 GFLOP: 5 us at the fp32 MFMA rate) and a launch of it costs 14.7 us (dispatch 2.3,
// prologue 2.5, first tile 0.9, 6 dependent K steps, epilogue; conv.hip conv1d_small_kernel) plus a 4.5 us reduction / normalisation launch: the
// stage ran 322 us forward and 511 us backward on the main stream of a 7.6 ms step (profiles/r04_bench.json, roofline_conv1d.windows_us).
//
// Here a clip is owned by a CLUSTER of 8 workgroups that the dispatcher places on one XCD (block ids b, b + 8, ..., b + 56 of a 64-block window:
// the same id mod 8).  Workgroup r of the cluster computes output channels [32 r, 32 r + 32) of every block for all frames of its clip:
//   1. wait on the cluster's counter until all 8 members have published the previous block (XCD-local: relaxed agent-scope polls of a counter
//      that lives in that XCD's L2; tools/debug/xcd_barrier_probe.hip: 0.64 us, 2.4 us with a 16 KB exchange),
//   2. read the RAW conv output of the producing block(s) (64 x 256 fp32 = 64 KB, written write-through by the 8 members, read with L1-bypassing
//      loads), normalise each frame over its 256 channels + LeakyReLU on the way into LDS (a workgroup needs every channel of every frame as
//      its K dimension anyway, so it derives the frame statistics itself: no statistics exchange), linear upsampling + skip addition included,
//   3. stream its 32 weight columns (k x Cin x 32 fp32 = 98 KB) through a double-buffered LDS ring under the MFMAs: the 4 waves split the
//      (frames x 32) tile by 32-frame halves and by K,
//   4. reduce the K splits through LDS, publish the raw tile (16-byte write-through stores), drain, bump the counter.
// Backward runs the same machine in reverse order: the gradient of a block's output is gathered on load from the input gradients of its
// consumers (sum of up to two direct consumers + the transposed upsampling of one), the normalisation backward is applied per frame on the
// way into LDS, and the GEMM against the (Cin,taps,Cout) weight mirror yields the block's input gradient.  What the weight-gradient
// launches need -- each block's conv input as consumed and the gradient of its raw output -- is written to HBM by the member that owns the
// frame (frame mod 8), in both directions; those launches stay on the side stream as before.
//
// Correctness never depends on the placement: the hand-off is {16-byte sc1 stores, vmcnt(0), counter} -> {counter poll, sc1 loads}
// (MI355X_MICROARCH.md, "valid forms"), valid across XCDs; same-XCD placement only makes it faster.  A member that never arrives (the GPU
// is shared and the grid is not co-resident) trips the spin limit: the error word is set and the Trainer raises (ops.check_streamk).
#include "common.h"

#define CH_MAXL 20       // blocks per chain
#define CH_C 256         // output channels of every block
#define CH_MAXCIN 320    // input channels of the first block (256 + clip code)
#define CH_MAXT 64       // frames
#define CH_KCMAX 128     // K chunk (floats) streamed per weight buffer
#define CH_WBUF (32 * (CH_KCMAX + 4))
#define CH_ROWS (CH_MAXT + 4)
#define CH_RED 36        // row stride of the K-split reduction tiles

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ch_layer {
    int Ti, To, Cin, k, stride, pad;
    int mode, src_a, src_b, Ta;  // input: 0 plain tensor, 1 act(norm(y[src_a])), 2 upsample(act(norm(y[src_a])), Ta -> Ti) + act(norm(y[src_b]))
    int g_id0, g_id1, g_up;      // backward: blocks whose input gradient is (part of) this block's output gradient; == n: the external gradient
    int KC;                      // K chunk: divides Cin (forward) and 256 (backward uses 128)
    const float* w;              // (256, k, Cin)
    const float* wt;             // (Cin, k, 256)
    float* y;                    // (B, To, 256) raw conv output
    float* x;                    // (B, Ti, Cin) conv input as consumed (mode != 0), or NULL
    float* dy;                   // (B, To, 256) gradient of the raw conv output
    float* dx;                   // (B, Ti, Cin) gradient of the conv input
};

struct ch_args {
    ch_layer L[CH_MAXL];
    int n, B, clip0;   // this launch owns clips [clip0, min(B, clip0 + clusters of the grid))
    float slope, eps;
    const float* x0;   // (B, Ti0, Cin0) input of block 0
    float* zout;       // (B, To_last, 256) act(norm(y[n-1])): the chain's output
    const float* gz;   // backward: gradient of zout
    unsigned* counters;  // one per clip, zero between launches
    unsigned* err;
    unsigned spin_limit;
    int need_dx0;
};

#ifdef SDT_TUNING
// tools/debug/chain_timeline.py: thread 0 of every workgroup stamps the 100 MHz counter at four points of every block:
// ch_dbg_tl[(workgroup * 24 + step) * 4 + slot]: 0 wait over, 1 input staged, 2 tile published, 3 arrived
__device__ unsigned long long* ch_dbg_tl = nullptr;
extern "C" int sdt_debug_set_timeline_chain(void* p) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(ch_dbg_tl), &p, sizeof(p));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define CH_TL(step, slot)                                                                                                         \
    do {                                                                                                                          \
        if (threadIdx.x == 0 && ch_dbg_tl != nullptr) ch_dbg_tl[((size_t)blockIdx.x * 24 + (step)) * 4 + (slot)] = wall_clock64(); \
    } while (0)
// fault injection (tests/test_chain1d_gpu.py::test_chain_lost_member_is_loud): member 3 of this clip's cluster leaves before its first arrival --
// what a workgroup that was never dispatched looks like to the other seven
__device__ int ch_dbg_mute_clip = -1;
extern "C" int sdt_debug_chain_mute_clip(int clip) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(ch_dbg_mute_clip), &clip, sizeof(clip));
    return e == hipSuccess ? SDT_OK : SDT_ERR_LAUNCH;
}
#define CH_MUTED(clip, r) ((clip) == ch_dbg_mute_clip && (r) == 3)
#else
#define CH_TL(step, slot)
#define CH_MUTED(clip, r) false
#endif

__device__ __forceinline__ void ch_src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {  // misc.hip src_index
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

__device__ __forceinline__ f32x4 ch_ld_sc1(const __amdgpu_buffer_rsrc_t rs, int byte_off) {  // L1-bypassing load of data another member published
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ __forceinline__ void ch_st_sc1(const __amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {  // write-through store
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_rsrc(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

// all members of the cluster have bumped the counter `want` times in total; *dead: a wait already timed out (no further waiting, error word set)
__device__ __forceinline__ void ch_wait(gu32* cnt, unsigned want, const ch_args& A, int* dead, int code) {
    if (want != 0u) {
        if (threadIdx.x == 0 && !*dead) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > A.spin_limit) {
                    *dead = 1;
                    __hip_atomic_store((gu32*)A.err, 0x40000000u + (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
}

// A wait of this workgroup timed out (a member of its cluster never arrived): besides the error word, everything downstream of the chain must see it
// at once -- the frames this member owns of `base` (rows x cols fp32, frame t belongs to member (t >> 2) & 7) become NaN, so the loss (forward) or
// the gradients that leave the chain (backward) are NaN in the SAME step, as the stream-K convolution does with a tile whose partner was lost
// (until the Trainer read the error word, up to LOG_INTERVAL optimiser steps applied finite garbage).
__device__ __forceinline__ void ch_poison_rows(float* base, int rows, int cols, int r) {
    const f32x4 nan4 = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    const int nv = cols >> 2;
    for (int i = threadIdx.x; i < rows * nv; i += 256) {
        const int t = i / nv;
        if (((t >> 2) & 7) == r) *(f32x4*)(base + (size_t)t * cols + 4 * (i - t * nv)) = nan4;
    }
}

// publish: every store of this workgroup has left (write-through), then one bump.  The member whose bump completes the launch's total lowers
// the counter again: it is zero between launches (hipGraph replays carry no epoch).
__device__ __forceinline__ void ch_arrive(gu32* cnt, unsigned total) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == total) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ---- weight streaming -----------------------------------------------------------------------
// A workgroup's 32 weight columns are streamed in chunks of KC floats per column.  The stream does not depend on the chain, so the first CH_D
// chunks of a block are requested BEFORE the workgroup waits for its cluster and stages its input (the wait and the staging hide the latency),
// and CH_D chunks stay in flight in registers throughout (measured with one chunk in flight: 2.2 us per chunk whatever the MFMA count).
// Every access of the stream is unconditional (a thread without a vector of the chunk re-reads the column's first floats and parks them in a
// dummy LDS slot; requests past the last chunk re-read the last one): the K loop's chunks are single basic blocks, which is what lets
// sched_group_barrier spread the feeding instructions between the MFMAs.
#define CH_D 4
#define CH_DUMMY (2 * CH_WBUF)  // [256][4] floats behind the two ring slots
struct ch_wstream {
    const float* Wcol;
    int K, KC, nch;
    int goff[4], loff[4];
    bool ok[4];
    f32x4 rg[CH_D][4];
};

__device__ __forceinline__ void ch_w_setup(ch_wstream& S, const float* Wcol, int K, int KC) {
    const int tid = threadIdx.x, kv = KC >> 2, nvec = 32 * kv, WS = KC + 4;
    S.Wcol = Wcol, S.K = K, S.KC = KC, S.nch = K / KC;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + 256 * p;
        S.ok[p] = idx < nvec;
        const int col = idx / kv, v = idx - col * kv;
        S.goff[p] = S.ok[p] ? col * K + 4 * v : 0;
        S.loff[p] = col * WS + 4 * v;
    }
}
template <int SLOT>
__device__ __forceinline__ void ch_w_issue(ch_wstream& S, int c) {
    const int cidx = min(c, S.nch - 1) * S.KC;
#pragma unroll
    for (int p = 0; p < 4; ++p) S.rg[SLOT][p] = *(const f32x4*)(S.Wcol + S.goff[p] + (S.ok[p] ? cidx : 0));
}
template <int SLOT>
__device__ __forceinline__ void ch_w_to_lds(ch_wstream& S, float* wb, int buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) *(f32x4*)(wb + (S.ok[p] ? buf * CH_WBUF + S.loff[p] : CH_DUMMY + 4 * tid)) = S.rg[SLOT][p];
}
__device__ __forceinline__ void ch_w_prefetch(ch_wstream& S) {
    ch_w_issue<0>(S, 0);
    ch_w_issue<1>(S, 1);
    ch_w_issue<2>(S, 2);
    ch_w_issue<3>(S, 3);
}

#define CH_SGB(mask) __builtin_amdgcn_sched_group_barrier(mask, 1, 0)
// tools/debug/r04_chain_ablation.sh (wrong results by design, never in the product library): what the K loop costs without 1 its MFMAs,
// 2 the weight loads of later chunks, 4 the LDS stores of the next chunk
#if defined(CH_ABL) && (CH_ABL & 1)
#define CH_MFMA(C, A, B) asm volatile("" : "+v"(C) : "v"(A), "v"(B))
#else
#define CH_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)
#endif
#if defined(CH_ABL) && (CH_ABL & 2)
#define CH_ABL_LOAD(x)
#else
#define CH_ABL_LOAD(x) x
#endif
#if defined(CH_ABL) && (CH_ABL & 4)
#define CH_ABL_STORE(x)
#else
#define CH_ABL_STORE(x) x
#endif

__device__ __forceinline__ void ch_barrier() {  // LDS traffic of this wave done, then the workgroup barrier (no vmcnt wait: the weight stream stays in flight)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Chunk C of a K loop that is FULLY unrolled (NCH chunks): a loop's back edge made the compiler wait for vmcnt(0) in front of every LDS store of
// the stream -- for the chunk requested one iteration ago instead of CH_D ago (tools/debug/r04_chain_ablation.sh: 0.27 us per chunk).  In
// straight-line code it counts the requests exactly.  Program order = the issue order asked of the scheduler: the chunk's first fragment reads,
// then per group of four MFMAs two fragment reads of a later group and two feeding instructions (LDS stores of chunk C + 1 into the other ring
// slot, then the requests for chunk C + 1 + CH_D into the registers just stored) -- tools/mfma_peak.hip: at one wave per SIMD the matrix pipe
// idles through every feeding instruction that is not issued under an MFMA.
typedef __bf16 ch_bf16x8 __attribute__((ext_vector_type(8)));
// BF: products of bf16-rounded operands on v_mfma_f32_32x32x16_bf16 (SDT_MATH_BF16: the arithmetic of a bf16 run; tensors, LDS tiles and the
// accumulation stay fp32).  Two 4-float fragment reads per lane = the 8 k values of one MFMA, converted (RNE) in registers: 1/16 of the matrix
// pipe's time per product, after which the K loop is bound by its fragment reads and barriers.
template <bool BF, int NJ, int NCH, int C, typename RowFn>
__device__ __forceinline__ void ch_chunk(ch_wstream& S, f32x16& acc, const float* xs, int RS, float* wb, int CK, int m, RowFn rowfn, int koff, int lane) {
    constexpr int NS = (C + 1) % CH_D;
    constexpr int NW = (C + 1 < NCH) ? 4 : 0, NV = (C + 1 + CH_D < NCH) ? 4 : 0, NF = NW + NV;
    constexpr int LEAD = NJ > 1 ? 2 : 1;
    constexpr int PER = (NF + NJ - 1) / NJ > 2 ? (NF + NJ - 1) / NJ : 2;
    const int KC = S.KC, WS = KC + 4, tid = threadIdx.x;
    const int tap = (C * KC) / CK, ci0 = C * KC - tap * CK;
    const float* ap = xs + rowfn(m, tap) * RS + ci0 + koff;
    const float* bp = wb + (C & 1) * CH_WBUF + (lane & 31) * WS + koff;
    const int nb = ((C + 1) & 1) * CH_WBUF;
    const int cnext = (C + 1 + CH_D) * KC;
    f32x4 a[NJ], b[NJ];
#pragma unroll
    for (int j = 0; j < LEAD; ++j) {
        a[j] = *(const f32x4*)(ap + 8 * j);
        b[j] = *(const f32x4*)(bp + 8 * j);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j + LEAD < NJ) {
            a[j + LEAD] = *(const f32x4*)(ap + 8 * (j + LEAD));
            b[j + LEAD] = *(const f32x4*)(bp + 8 * (j + LEAD));
        }
#pragma unroll
        for (int f = (j * PER < NF ? j * PER : NF); f < ((j + 1) * PER < NF ? (j + 1) * PER : NF); ++f) {
            if (f < NW) {
                CH_ABL_STORE(*(f32x4*)(wb + (S.ok[f] ? nb + S.loff[f] : CH_DUMMY + 4 * tid)) = S.rg[NS][f]);
            } else {
                CH_ABL_LOAD(S.rg[NS][f - NW] = *(const f32x4*)(S.Wcol + S.goff[f - NW] + (S.ok[f - NW] ? cnext : 0)));
            }
        }
        if constexpr (!BF) {
            CH_MFMA(acc, a[j][0], b[j][0]);
            CH_MFMA(acc, a[j][1], b[j][1]);
            CH_MFMA(acc, a[j][2], b[j][2]);
            CH_MFMA(acc, a[j][3], b[j][3]);
        } else if ((j & 1) || j + 1 == NJ) {  // a pair of fragment groups (an odd last group alone, zero-padded) = one bf16 MFMA
            const int j0 = (j & 1) ? j - 1 : j;
            ch_bf16x8 av, bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = (__bf16)a[j0][e];
                bv[e] = (__bf16)b[j0][e];
                av[4 + e] = (j0 + 1 < NJ) ? (__bf16)a[j0 + 1 < NJ ? j0 + 1 : j0][e] : (__bf16)0.f;
                bv[4 + e] = (j0 + 1 < NJ) ? (__bf16)b[j0 + 1 < NJ ? j0 + 1 : j0][e] : (__bf16)0.f;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    }
    // the same order for the scheduler (fp32 only: the bf16 loop has an eighth of the MFMAs and is left to the compiler)
#pragma unroll
    for (int j = 0; j < (BF ? 0 : 2 * LEAD); ++j) CH_SGB(0x100);
#pragma unroll
    for (int j = 0; j < (BF ? 0 : NJ); ++j) {
        const int f0 = j * PER < NF ? j * PER : NF, f1 = (j + 1) * PER < NF ? (j + 1) * PER : NF;
        CH_SGB(0x008);
        if (j + LEAD < NJ) CH_SGB(0x100);
        CH_SGB(0x008);
        if (j + LEAD < NJ) CH_SGB(0x100);
        CH_SGB(0x008);
        if (f0 < f1) {
            if (f0 < NW) CH_SGB(0x200);
            else CH_SGB(0x020);
        }
        CH_SGB(0x008);
#pragma unroll
        for (int f = f0 + 1; f < f1; ++f) {
            if (f < NW) CH_SGB(0x200);
            else CH_SGB(0x020);
        }
    }
    ch_barrier();
    if constexpr (C + 1 < NCH) ch_chunk<BF, NJ, NCH, C + 1>(S, acc, xs, RS, wb, CK, m, rowfn, koff, lane);
}

// acc (one 32 x 32 tile per wave) = A (rows from LDS through rowfn) x W[n0 .. n0+32)^T over the stream S (K = ntap * CK floats per column); then
// the K splits are reduced through LDS and rows [0, M) x 32 columns are published to `out` (row stride ldo floats).
// rowfn(m, tap) -> LDS row of xs that multiplies tap `tap` for output row m.
template <bool BF, int NJ, int NCH, typename RowFn>
__device__ __forceinline__ void ch_gemm_store_nj(ch_wstream& S, const float* xs, int RS, float* wb, int CK, int M, RowFn rowfn,
                                                const __amdgpu_buffer_rsrc_t rsOut, int ldo, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MT = M > 32 ? 2 : 1, KS = 4 / MT;
    const int mt = wave % MT, ks = wave / MT;
    const int kper = S.KC / KS;
    int m = mt * 32 + (lane & 31);
    if (m >= M) m = M - 1;  // rows past the end compute a copy of the last row: never stored
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int koff = ks * kper + (lane >> 5) * 4;
    // chunk 0 -> ring slot 0.  Chunk c then reads slot c & 1 while chunk c + 1 is written to the other slot (last read by chunk c - 1, which every
    // wave finished before the barrier that closed it) and chunk c + 1 + CH_D is requested into the registers that held chunk c + 1.
    ch_w_to_lds<0>(S, wb, 0);
    if constexpr (CH_D < NCH) ch_w_issue<0>(S, CH_D);
    ch_barrier();
    ch_chunk<BF, NJ, NCH, 0>(S, acc, xs, RS, wb, CK, m, rowfn, koff, lane);
    // K splits -> LDS (the weight ring is free: the loop ended on a barrier), summed in split order, published
    float* red = wb;
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave * (32 * CH_RED) + (8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * CH_RED + (lane & 31)] = acc[v];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = tid + 256 * p, row = q >> 3, cv = q & 7;
        if (row < M) {
            const float* rp = red + ((row >> 5) * (32 * CH_RED)) + (row & 31) * CH_RED + 4 * cv;
            f32x4 s = *(const f32x4*)rp;
            for (int z = 1; z < KS; ++z) s += *(const f32x4*)(rp + z * MT * (32 * CH_RED));
            ch_st_sc1(rsOut, (row * ldo + n0 + 4 * cv) * 4, s);
        }
    }
    __syncthreads();  // red is the weight ring of the next GEMM
}

// (MFMA groups per chunk, chunks) of the blocks this kernel is built for: 64-frame k3 blocks (8, 6), the 288-channel first block (6, 9), blocks of
// <= 32 frames with k3 (4, 6; 3, 9 with 288 channels) / k4 (4, 8), the input gradient of the first strided block (64 frames, k4: 8, 8); chain_check() admits nothing else.
template <bool BF, typename RowFn>
__device__ __forceinline__ void ch_gemm_store(ch_wstream& S, const float* xs, int RS, float* wb, int CK, int M, RowFn rowfn,
                                             const __amdgpu_buffer_rsrc_t rsOut, int ldo, int n0) {
    const int nj = (S.KC / (M > 32 ? 2 : 4)) >> 3;
    const int key = nj * 16 + S.nch;
    switch (key) {
        case 8 * 16 + 6: ch_gemm_store_nj<BF, 8, 6>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
        case 8 * 16 + 8: ch_gemm_store_nj<BF, 8, 8>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
        case 6 * 16 + 9: ch_gemm_store_nj<BF, 6, 9>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
        case 4 * 16 + 6: ch_gemm_store_nj<BF, 4, 6>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
        case 3 * 16 + 9: ch_gemm_store_nj<BF, 3, 9>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
        default: ch_gemm_store_nj<BF, 4, 8>(S, xs, RS, wb, CK, M, rowfn, rsOut, ldo, n0); break;
    }
}

// ---- frames on load ---------------------------------------------------------------------------
// A wave handles 4 frames per pass: lane = 16 fq + li holds channels 4 (li + 16 q) .. + 3, q = 0..3, of frame t0 + fq, so a frame's sums over
// its 256 channels are 16 in-lane adds + a 16-lane butterfly of 4 DPP steps (no LDS round trips: the first version reduced each frame over the
// whole wave with 12 dependent ds_bpermute per frame and spent 10 us per block on it).
struct ch_row {
    f32x4 v[4];
};
template <int CTRL>
__device__ __forceinline__ float ch_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ch_sum16(float v) {  // every lane of a 16-lane DPP row gets the row's sum
    v = ch_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = ch_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = ch_dpp_add<0x141>(v);  // row_half_mirror
    v = ch_dpp_add<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ ch_row ch_ld_row(const __amdgpu_buffer_rsrc_t rs, int t, int li) {  // out of range: zeros
    ch_row a;
#pragma unroll
    for (int q = 0; q < 4; ++q) a.v[q] = ch_ld_sc1(rs, (t * CH_C + 4 * (li + 16 * q)) * 4);
    return a;
}
__device__ __forceinline__ void ch_row_stats(const ch_row& a, float eps, float& mu, float& rs) {  // norm.hip rownorm_kernel: two passes, biased variance
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += (a.v[q][0] + a.v[q][1]) + (a.v[q][2] + a.v[q][3]);
    mu = ch_sum16(s) * (1.f / CH_C);
    float q2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = a.v[q][e] - mu;
            q2 += d * d;
        }
    rs = 1.f / sqrtf(ch_sum16(q2) * (1.f / CH_C) + eps);
}
__device__ __forceinline__ ch_row ch_norm_row(const ch_row& a, float eps, float slope) {
    float mu, rs;
    ch_row_stats(a, eps, mu, rs);
    ch_row o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[q][e] = act_fwd((a.v[q][e] - mu) * rs, slope);
    return o;
}
__device__ __forceinline__ void ch_put_row(float* dst, const ch_row& a, int li) {  // dst: the frame's first channel (LDS or global)
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 4 * (li + 16 * q)) = a.v[q];
}

template <bool BF>
__global__ __launch_bounds__(256) void chain1d_fwd_kernel(const ch_args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int dead;
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane >> 4, li = lane & 15;
    const int slot = (bid >> 6) * 8 + (bid & 7), clip = A.clip0 + slot, r = (bid >> 3) & 7;  // slot: the cluster's index within this launch
    if (clip >= A.B || CH_MUTED(clip, r)) return;
    float* xs = smem;                                   // [CH_ROWS][Cin + 4]
    float* wb = smem + CH_ROWS * (CH_MAXCIN + 4);       // [2][32][KC + 4]
    gu32* cnt = (gu32*)(A.counters + slot);
    const unsigned total = 8u * (unsigned)(A.n + 1);
    if (tid == 0) dead = 0;
    __syncthreads();
    ch_wstream S;
    for (int l = 0; l < A.n; ++l) {
        const ch_layer& L = A.L[l];
        const int RS = L.Cin + 4;
        ch_w_setup(S, L.w + (size_t)(32 * r) * L.k * L.Cin, L.k * L.Cin, L.KC);
        ch_w_prefetch(S);
        const int after = max(0, (L.To - 1) * L.stride + L.k - 1 - L.pad - (L.Ti - 1));
        // halo frames (zero padding)
        for (int i = tid; i < (L.pad + after) * (L.Cin >> 2); i += 256) {
            const int hr = i / (L.Cin >> 2), c4 = i - hr * (L.Cin >> 2);
            const int row = hr < L.pad ? hr : L.Ti + hr;
            *(f32x4*)(xs + row * RS + 4 * c4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (L.mode == 0) {
            CH_TL(l, 0);
            const float* src = A.x0 + (size_t)clip * L.Ti * L.Cin;
            const int nv = L.Cin >> 2, tot = L.Ti * nv;
            for (int i0 = tid; i0 < tot; i0 += 256 * 6) {
                f32x4 v[6];
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (i0 + 256 * u < tot) v[u] = *(const f32x4*)(src + (size_t)(i0 + 256 * u) * 4);
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int i = i0 + 256 * u;
                    if (i < tot) {
                        const int t = i / nv, c4 = i - t * nv;
                        *(f32x4*)(xs + (t + L.pad) * RS + 4 * c4) = v[u];
                    }
                }
            }
        } else {
            ch_wait(cnt, 8u * (unsigned)l, A, &dead, l);
            CH_TL(l, 0);
            const ch_layer& LA = A.L[L.src_a];
            const __amdgpu_buffer_rsrc_t rsA = ch_rsrc(LA.y + (size_t)clip * LA.To * CH_C, LA.To * CH_C * 4);
            float* xout = L.x ? L.x + (size_t)clip * L.Ti * CH_C : nullptr;
            if (L.mode == 1) {
                const int t0 = wave * 16;  // <= 64 frames: one batch per wave
                if (t0 < L.Ti) {
                    ch_row a[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) a[p] = ch_ld_row(rsA, t0 + 4 * p + fq, li);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int t = t0 + 4 * p + fq;
                        const ch_row z = ch_norm_row(a[p], A.eps, A.slope);
                        if (t < L.Ti) {
                            ch_put_row(xs + (t + L.pad) * RS, z, li);
                            if (xout != nullptr && ((t >> 2) & 7) == r) ch_put_row(xout + (size_t)t * CH_C, z, li);
                        }
                    }
                }
            } else {
                const ch_layer& LB = A.L[L.src_b];
                const __amdgpu_buffer_rsrc_t rsB = ch_rsrc(LB.y + (size_t)clip * LB.To * CH_C, LB.To * CH_C * 4);
                const float sc = (float)L.Ta / (float)L.Ti;
                for (int t0 = wave * 8; t0 < L.Ti; t0 += 32) {
                    ch_row a0[2], a1[2], bs[2];
                    float l1s[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int t = min(t0 + 4 * p + fq, L.Ti - 1);
                        int i0, i1;
                        ch_src_index(sc, t, L.Ta, i0, i1, l1s[p]);
                        a0[p] = ch_ld_row(rsA, i0, li);
                        a1[p] = ch_ld_row(rsA, i1, li);
                        bs[p] = ch_ld_row(rsB, t, li);
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int t = t0 + 4 * p + fq;
                        const ch_row z0 = ch_norm_row(a0[p], A.eps, A.slope), z1 = ch_norm_row(a1[p], A.eps, A.slope);
                        const ch_row zb = ch_norm_row(bs[p], A.eps, A.slope);
                        ch_row u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u.v[q] = (1.f - l1s[p]) * z0.v[q] + l1s[p] * z1.v[q];
                            u.v[q] += zb.v[q];
                        }
                        if (t < L.Ti) {
                            ch_put_row(xs + (t + L.pad) * RS, u, li);
                            if (xout != nullptr && ((t >> 2) & 7) == r) ch_put_row(xout + (size_t)t * CH_C, u, li);
                        }
                    }
                }
            }
        }
        __syncthreads();
        CH_TL(l, 1);
        const int stride = L.stride;
        const __amdgpu_buffer_rsrc_t rsY = ch_rsrc(L.y + (size_t)clip * L.To * CH_C, L.To * CH_C * 4);
        ch_gemm_store<BF>(S, xs, RS, wb, L.Cin, L.To, [stride](int m, int tap) { return m * stride + tap; }, rsY, CH_C, 32 * r);
        CH_TL(l, 2);
        ch_arrive(cnt, total);
        CH_TL(l, 3);
    }
    // the chain's output: act(norm(y[n-1])), each member the frames it owns
    {
        const ch_layer& L = A.L[A.n - 1];
        ch_wait(cnt, 8u * (unsigned)A.n, A, &dead, A.n);
        const __amdgpu_buffer_rsrc_t rsA = ch_rsrc(L.y + (size_t)clip * L.To * CH_C, L.To * CH_C * 4);
        float* zo = A.zout + (size_t)clip * L.To * CH_C;
        for (int t0 = 4 * r + 32 * wave; t0 < L.To; t0 += 128) {  // frames 4 r .. 4 r + 3 of every 32
            const int t = t0 + fq;
            const ch_row z = ch_norm_row(ch_ld_row(rsA, t, li), A.eps, A.slope);
            if (t < L.To) ch_put_row(zo + (size_t)t * CH_C, z, li);
        }
        if (dead) ch_poison_rows(zo, L.To, CH_C, r);
        ch_arrive(cnt, total);
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: blocks in reverse order.  LDS row 0 of the gradient tile is zero (taps that fall outside the output / off the stride grid).
struct ch_bwd_src {
    __amdgpu_buffer_rsrc_t rs0, rs1, rsU;
    float usc;
    int up, TiU;
};

// gradient of the raw output of frames t0 + fq (one pass of a wave): gather the output gradient, normalisation backward (norm.hip rownorm_kernel<BWD>)
template <bool UP, bool TWO>
__device__ __forceinline__ void ch_bwd_frames(const ch_args& A, const ch_layer& L, const ch_bwd_src& G, const float* ysrc, float* dyo, float* ds, int RS,
                                              int t0, int fq, int li, int r, const ch_row* ypre) {
    constexpr int NP = UP ? 1 : 4;  // frames in flight per lane group: a 64-frame block is one batch of loads per wave
    ch_row g[NP], yv[NP], up[UP ? 6 : 1];
    float uw[UP ? 6 : 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int t = t0 + 4 * p + fq;  // past the end: the buffer range returns zeros
        g[p] = ch_ld_row(G.rs0, t, li);  // zero-sized range: zeros
        if constexpr (TWO) {
            const ch_row g1 = ch_ld_row(G.rs1, t, li);
#pragma unroll
            for (int q = 0; q < 4; ++q) g[p].v[q] += g1.v[q];
        }
        if constexpr (!UP) {
            yv[p] = ypre[p];  // requested before the wait for the cluster (the forward launch wrote it: no dependence on this launch's progress)
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                yv[p].v[q] = t < L.To ? *(const f32x4*)(ysrc + (size_t)t * CH_C + 4 * (li + 16 * q)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (UP) {
            const int jlo = (int)floorf(((float)t - 0.5f) / G.usc - 0.5f) - 1;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int j = jlo + s;
                float w = 0.f;
                if (j >= 0 && j < G.TiU) {
                    int i0, i1;
                    float l1;
                    ch_src_index(G.usc, j, L.To, i0, i1, l1);
                    w = (t == i0 ? 1.f - l1 : 0.f) + (t == i1 ? l1 : 0.f);
                }
                uw[s] = w;
                up[s] = ch_ld_row(G.rsU, w != 0.f ? j : 0x100000, li);  // weight 0: out of range, zeros, no traffic
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int t = t0 + 4 * p + fq;
        if constexpr (UP) {
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) g[p].v[q] += uw[s] * up[s].v[q];
        }
        float mu, rs;
        ch_row_stats(yv[p], A.eps, mu, rs);
        ch_row yh, gg;
        float sg = 0.f, sgy = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yh.v[q][e] = (yv[p].v[q][e] - mu) * rs;
                gg.v[q][e] = g[p].v[q][e] * act_grad(yh.v[q][e], A.slope);
                sg += gg.v[q][e];
                sgy += gg.v[q][e] * yh.v[q][e];
            }
        sg = ch_sum16(sg) * (1.f / CH_C);
        sgy = ch_sum16(sgy) * (1.f / CH_C);
        ch_row o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[q][e] = rs * (gg.v[q][e] - sg - yh.v[q][e] * sgy);
        if (t < L.To) {
            ch_put_row(ds + (1 + t) * RS, o, li);
            if (((t >> 2) & 7) == r) ch_put_row(dyo + (size_t)t * CH_C, o, li);
        }
    }
}

template <bool BF>
__global__ __launch_bounds__(256) void chain1d_bwd_kernel(const ch_args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int dead;
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane >> 4, li = lane & 15;
    const int slot = (bid >> 6) * 8 + (bid & 7), clip = A.clip0 + slot, r = (bid >> 3) & 7;  // slot: the cluster's index within this launch
    if (clip >= A.B || CH_MUTED(clip, r)) return;
    const int RS = CH_C + 4;
    float* ds = smem;                                   // [1 + To][260]
    float* wb = smem + CH_ROWS * (CH_MAXCIN + 4);
    gu32* cnt = (gu32*)(A.counters + slot);
    const unsigned total = 8u * (unsigned)A.n;
    if (tid == 0) dead = 0;
    if (tid < 64) *(f32x4*)(ds + 4 * tid) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#ifdef SDT_TUNING
    if (tid == 0 && ch_dbg_tl != nullptr) {  // placement check of tools/debug/chain_timeline.py: the XCD this workgroup runs on
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        ch_dbg_tl[((size_t)blockIdx.x * 24 + 23) * 4] = id & 0xf;
    }
#endif
    ch_wstream S;
    for (int s = 0; s < A.n; ++s) {
        const int l = A.n - 1 - s;
        const ch_layer& L = A.L[l];
        const bool gemm = l > 0 || A.need_dx0;
        if (gemm) {
            ch_w_setup(S, L.wt + (size_t)(32 * r) * L.k * CH_C, L.k * CH_C, CH_KCMAX);
            ch_w_prefetch(S);
        }
        const float* ysrc = L.y + (size_t)clip * L.To * CH_C;
        ch_row ypre[4];
        if (L.g_up < 0) {  // the block's own raw output (HBM by now): in flight across the wait, like the weights
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int t = wave * 16 + 4 * p + fq;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ypre[p].v[q] = t < L.To ? *(const f32x4*)(ysrc + (size_t)t * CH_C + 4 * (li + 16 * q)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        ch_wait(cnt, 8u * (unsigned)s, A, &dead, 64 + l);
        CH_TL(s, 0);
        float* dyo = L.dy + (size_t)clip * L.To * CH_C;
        const bool ext0 = L.g_id0 == A.n;
        const ch_layer& G0 = A.L[(L.g_id0 >= 0 && !ext0) ? L.g_id0 : 0];
        const ch_layer& G1 = A.L[L.g_id1 >= 0 ? L.g_id1 : 0];
        const ch_layer& GU = A.L[L.g_up >= 0 ? L.g_up : 0];
        // consumers read this block's output as their 256-channel input: their input-gradient rows are 256 floats
        ch_bwd_src G;
        G.rs0 = ext0 ? ch_rsrc(A.gz + (size_t)clip * L.To * CH_C, L.To * CH_C * 4)
                     : ch_rsrc(G0.dx + (size_t)clip * G0.Ti * CH_C, (L.g_id0 >= 0 ? G0.Ti * CH_C * 4 : 0));
        G.rs1 = ch_rsrc(G1.dx + (size_t)clip * G1.Ti * CH_C, (L.g_id1 >= 0 ? G1.Ti * CH_C * 4 : 0));
        G.rsU = ch_rsrc(GU.dx + (size_t)clip * GU.Ti * CH_C, (L.g_up >= 0 ? GU.Ti * CH_C * 4 : 0));
        G.up = L.g_up >= 0;
        G.TiU = GU.Ti;
        G.usc = G.up ? (float)L.To / (float)GU.Ti : 1.f;  // upsampling source index scale: in / out
        if (G.up) {
            for (int t0 = wave * 4; t0 < L.To; t0 += 16) ch_bwd_frames<true, true>(A, L, G, ysrc, dyo, ds, RS, t0, fq, li, r, ypre);
        } else if (wave * 16 < L.To) {  // <= 64 frames: one batch per wave
            if (L.g_id1 >= 0) ch_bwd_frames<false, true>(A, L, G, ysrc, dyo, ds, RS, wave * 16, fq, li, r, ypre);
            else ch_bwd_frames<false, false>(A, L, G, ysrc, dyo, ds, RS, wave * 16, fq, li, r, ypre);
        }
        __syncthreads();
        CH_TL(s, 1);
        // ---- input gradient: dx[ti][ci] = sum_{tap, co} dy[(ti + pad - tap) / stride][co] * wt[ci][tap][co]
        if (gemm) {
            const int stride = L.stride, pad = L.pad, To = L.To;
            const __amdgpu_buffer_rsrc_t rsD = ch_rsrc(L.dx + (size_t)clip * L.Ti * L.Cin, L.Ti * L.Cin * 4);
            auto rowfn = [stride, pad, To](int m, int tap) {
                const int num = m + pad - tap;
                if (num < 0) return 0;
                const int to = num / stride;
                return (num - to * stride == 0 && to < To) ? 1 + to : 0;
            };
            ch_gemm_store<BF>(S, ds, RS, wb, CH_C, L.Ti, rowfn, rsD, L.Cin, 32 * r);
            for (int n0 = 32 * r + 256; n0 < L.Cin; n0 += 256) {  // a 288-channel first block: member 0 also owns columns 256 .. 287
                ch_w_setup(S, L.wt + (size_t)n0 * L.k * CH_C, L.k * CH_C, CH_KCMAX);
                ch_w_prefetch(S);
                ch_gemm_store<BF>(S, ds, RS, wb, CH_C, L.Ti, rowfn, rsD, L.Cin, n0);
            }
        }
        CH_TL(s, 2);
        if (l == 0 && dead) {  // (shared flag: uniform after the barriers above) what leaves the chain: block 0's raw-output gradient and input gradient
            ch_poison_rows(L.dy + (size_t)clip * L.To * CH_C, L.To, CH_C, r);
            if (A.need_dx0) ch_poison_rows(L.dx + (size_t)clip * L.Ti * L.Cin, L.Ti, L.Cin, r);
        }
        ch_arrive(cnt, total);
        CH_TL(s, 3);
    }
}

// ---------------------------------------------------------------------------------------------
static bool chain_variant(int nj, int nch) {  // the unrolled K loops ch_gemm_store() dispatches to
    return (nj == 8 && (nch == 6 || nch == 8)) || (nj == 6 && nch == 9) || (nj == 4 && (nch == 6 || nch == 8)) || (nj == 3 && nch == 9);
}

static int chain_check(const sdt_chain1d_layer* Ls, int n, int B, bool pointers = true) {
    SDT_CHECK_ARG(Ls != nullptr && n >= 1 && n <= CH_MAXL && B >= 1, "bad chain");
    for (int l = 0; l < n; ++l) {
        const sdt_chain1d_layer& L = Ls[l];
        SDT_CHECK_ARG(L.Ti >= 1 && L.Ti <= CH_MAXT && L.To >= 1 && L.To <= CH_MAXT, "frames per clip out of range (1..64)");
        SDT_CHECK_ARG(L.Cin % 32 == 0 && L.Cin >= 32 && L.Cin <= CH_MAXCIN, "input channels must be a multiple of 32, at most 320");
        SDT_CHECK_ARG(L.k >= 1 && L.k <= 8 && L.stride >= 1 && L.stride <= 2 && L.pad >= 0 && L.pad < L.k, "kernel / stride / padding out of range");
        SDT_CHECK_ARG(L.To == (L.Ti + 2 * L.pad - L.k) / L.stride + 1, "output length does not match the geometry");
        SDT_CHECK_ARG(L.pad + L.Ti + std::max(0, (L.To - 1) * L.stride + L.k - 1 - L.pad - (L.Ti - 1)) <= CH_ROWS, "padded input exceeds the LDS tile");
        SDT_CHECK_ARG(L.in_mode >= 0 && L.in_mode <= 2, "unknown input mode");
        SDT_CHECK_ARG((L.in_mode == 0) == (l == 0), "exactly block 0 reads the external input");
        if (L.in_mode != 0) {
            SDT_CHECK_ARG(L.Cin == CH_C, "a block fed by the chain has 256 input channels");
            SDT_CHECK_ARG(L.src_a >= 0 && L.src_a < l, "source block must precede its consumer");
            if (L.in_mode == 1) SDT_CHECK_ARG(Ls[L.src_a].To == L.Ti, "source length mismatch");
            if (L.in_mode == 2) {
                SDT_CHECK_ARG(L.src_b >= 0 && L.src_b < l && Ls[L.src_b].To == L.Ti, "skip source mismatch");
                SDT_CHECK_ARG(Ls[L.src_a].To <= L.Ti && 2 * Ls[L.src_a].To >= L.Ti, "upsampling ratio must be in [1, 2]");
            }
        }
        const int KC = L.Cin % 128 == 0 ? 128 : (L.Cin % 96 == 0 ? 96 : (L.Cin % 64 == 0 ? 64 : 32));
        SDT_CHECK_ARG(chain_variant((KC / (L.To > 32 ? 2 : 4)) >> 3, L.k * L.Cin / KC) && chain_variant((128 / (L.Ti > 32 ? 2 : 4)) >> 3, 2 * L.k),
                      "no K loop was built for this block (kernel size 3 or 4; 256 or 288 input channels)");
        SDT_CHECK_ARG(!pointers || (L.w != nullptr && L.y != nullptr), "NULL weight / output");
    }
    return SDT_OK;
}

static int chain_fill(ch_args& A, const sdt_chain1d_layer* Ls, int n, int B, float slope, float eps, unsigned* counters, unsigned* err, bool bwd) {
    A.n = n;
    A.B = B;
    A.slope = slope;
    A.eps = eps;
    A.counters = counters;
    A.err = err;
    // the chain polls with s_sleep(1), the stream-K owner (whose limit this is) with s_sleep(8): the same wall-clock patience needs 8x the polls
    {
        const unsigned long long lim = (unsigned long long)sdt_convsk_get_spin_limit() * 8ull;
        A.spin_limit = lim > 0xffffffffull ? 0xffffffffu : (unsigned)lim;
    }
    for (int l = 0; l < n; ++l) {
        const sdt_chain1d_layer& S = Ls[l];
        ch_layer& L = A.L[l];
        L.Ti = S.Ti, L.To = S.To, L.Cin = S.Cin, L.k = S.k, L.stride = S.stride, L.pad = S.pad;
        L.mode = S.in_mode, L.src_a = S.src_a, L.src_b = S.src_b;
        L.Ta = S.in_mode == 2 ? Ls[S.src_a].To : 0;
        L.g_id0 = L.g_id1 = L.g_up = -1;
        L.KC = S.Cin % 128 == 0 ? 128 : (S.Cin % 96 == 0 ? 96 : (S.Cin % 64 == 0 ? 64 : 32));
        L.w = S.w, L.wt = S.wt, L.y = S.y, L.x = S.x, L.dy = S.dy, L.dx = S.dx;
    }
    if (bwd) {
        // consumers of every block's output, from the forward wiring; the last block's output gradient is external
        A.L[n - 1].g_id0 = n;
        for (int m = 1; m < n; ++m) {
            const sdt_chain1d_layer& S = Ls[m];
            auto add_id = [&](int src) {
                ch_layer& P = A.L[src];
                if (P.g_id0 < 0) P.g_id0 = m;
                else if (P.g_id1 < 0) P.g_id1 = m;
                else return false;
                return true;
            };
            if (S.in_mode == 1) SDT_CHECK_ARG(add_id(S.src_a), "more than two direct consumers of one block");
            if (S.in_mode == 2) {
                SDT_CHECK_ARG(add_id(S.src_b), "more than two direct consumers of one block");
                SDT_CHECK_ARG(A.L[S.src_a].g_up < 0, "more than one upsampling consumer of one block");
                A.L[S.src_a].g_up = m;
            }
        }
        for (int l = 0; l < n; ++l) {
            SDT_CHECK_ARG(A.L[l].g_id0 >= 0 || A.L[l].g_up >= 0, "a block without consumers");
            SDT_CHECK_ARG(A.L[l].g_id0 != n || (A.L[l].g_id1 < 0 && A.L[l].g_up < 0) || l == n - 1, "bad consumer table");
            SDT_CHECK_ARG(Ls[l].dy != nullptr && (l == 0 || (Ls[l].wt != nullptr && Ls[l].dx != nullptr)), "backward needs dy of every block, wt / dx of every block but the first");
        }
    }
    return SDT_OK;
}

static const size_t kChainLds = (size_t)(CH_ROWS * (CH_MAXCIN + 4) + 2 * CH_WBUF + 256 * 4) * 4;

// clips one launch can own: one workgroup per CU (LDS), every cluster co-resident, clusters allotted in windows of 64 block ids (8 clips);
// a larger batch runs as consecutive launches of this many clips each
static int chain_clips_per_launch() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return 8 * (cus / 64);
}

extern "C" int sdt_chain1d_supported(const sdt_chain1d_layer* layers, int nlayers, int B) {
    if (chain_check(layers, nlayers, B, false) != SDT_OK) return 0;
    return chain_clips_per_launch() >= 8 ? 1 : 0;
}

extern "C" int sdt_chain1d_fwd_f32(const sdt_chain1d_layer* layers, int nlayers, const float* x0, float* zout, int B, float slope, float eps,
                                   int math, void* counters, void* err, void* stream) {
    int rc = chain_check(layers, nlayers, B);
    if (rc != SDT_OK) return rc;
    SDT_CHECK_ARG(x0 != nullptr && zout != nullptr && counters != nullptr && err != nullptr, "NULL tensor");
    SDT_CHECK_ARG(math == SDT_MATH_F32 || math == SDT_MATH_BF16, "product arithmetic must be SDT_MATH_F32 or SDT_MATH_BF16");
    ch_args A;
    rc = chain_fill(A, layers, nlayers, B, slope, eps, (unsigned*)counters, (unsigned*)err, false);
    if (rc != SDT_OK) return rc;
    A.x0 = x0;
    A.zout = zout;
    A.gz = nullptr;
    A.need_dx0 = 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)chain1d_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds);
        (void)hipFuncSetAttribute((const void*)chain1d_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds);
        attr_set = true;
    }
    const int per = chain_clips_per_launch();
    SDT_CHECK_ARG(per >= 8, "the device cannot hold one window of clusters");
    for (int c0 = 0; c0 < B; c0 += per) {
        A.clip0 = c0;
        const int nclip = std::min(per, B - c0);
        if (math == SDT_MATH_BF16) hipLaunchKernelGGL(chain1d_fwd_kernel<true>, dim3(64 * ((nclip + 7) / 8)), dim3(256), kChainLds, (hipStream_t)stream, A);
        else hipLaunchKernelGGL(chain1d_fwd_kernel<false>, dim3(64 * ((nclip + 7) / 8)), dim3(256), kChainLds, (hipStream_t)stream, A);
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}

extern "C" int sdt_chain1d_bwd_f32(const sdt_chain1d_layer* layers, int nlayers, const float* gz, int B, float slope, float eps, int need_dx0,
                                   int math, void* counters, void* err, void* stream) {
    int rc = chain_check(layers, nlayers, B);
    if (rc != SDT_OK) return rc;
    SDT_CHECK_ARG(gz != nullptr && counters != nullptr && err != nullptr, "NULL tensor");
    SDT_CHECK_ARG(math == SDT_MATH_F32 || math == SDT_MATH_BF16, "product arithmetic must be SDT_MATH_F32 or SDT_MATH_BF16");
    ch_args A;
    rc = chain_fill(A, layers, nlayers, B, slope, eps, (unsigned*)counters, (unsigned*)err, true);
    if (rc != SDT_OK) return rc;
    A.x0 = nullptr;
    A.zout = nullptr;
    A.gz = gz;
    A.need_dx0 = need_dx0;
    SDT_CHECK_ARG(!need_dx0 || (layers[0].wt != nullptr && layers[0].dx != nullptr), "need_dx0 without wt / dx of block 0");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)chain1d_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds);
        (void)hipFuncSetAttribute((const void*)chain1d_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds);
        attr_set = true;
    }
    const int per = chain_clips_per_launch();
    SDT_CHECK_ARG(per >= 8, "the device cannot hold one window of clusters");
    for (int c0 = 0; c0 < B; c0 += per) {
        A.clip0 = c0;
        const int nclip = std::min(per, B - c0);
        if (math == SDT_MATH_BF16) hipLaunchKernelGGL(chain1d_bwd_kernel<true>, dim3(64 * ((nclip + 7) / 8)), dim3(256), kChainLds, (hipStream_t)stream, A);
        else hipLaunchKernelGGL(chain1d_bwd_kernel<false>, dim3(64 * ((nclip + 7) / 8)), dim3(256), kChainLds, (hipStream_t)stream, A);
    }
    SDT_LAUNCH_CHECK();
    return SDT_OK;
}
