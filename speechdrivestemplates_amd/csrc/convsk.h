// Shared between the persistent stream-K convolution kernels (convsk.hip: fp32 and the round-4 bf16 instantiations; convbf.hip: the
// bf16-shaped kernel of round 5): kernel-argument structs filled from a plan blob (convsk.hip, sdt_convsk_plan_build_t), element-type helpers.
#pragma once
#include "common.h"

#define SK_BK 32
#define SK_LDP 36
#define SK_MAXC 4
#define SK_OOB 0x80000000u
#define SK_CHUNK 8  // K steps per accumulation chunk (256 products)

struct sk_class {
    int Hi, Wi, Cin, Cout, ntaps, nkc;
    int tile_begin, nmb, row_begin, mt_begin;
    int Tw;
    int ashift[SDT_MAX_TAPS];  // ((dy * Wi + dx) * Cin) * 4
    int dyx[SDT_MAX_TAPS];     // (dy & 0xffff) | (dx << 16)
    int bshift[SDT_MAX_TAPS];  // wt * Cin * 4
};

struct sk_args {
    sk_class cls[SK_MAXC];
    int ncls, nnb, T, G;
    int ntmajor;            // one class only: tiles ordered n-tile major (tile = nt * nmb + mt), see plan_build
    int S;                  // total live K steps of the launch
    unsigned xbytes, wbytes, ybytes;
    const int4* rowinfo;    // [rows padded to BM per class]
    const int2* tileinfo;   // [m-tiles of all classes] {live-tap mask rotated by rot, rot}
    const int* tilecum;     // [T + 1]
    const int* range_tile;  // [G] first tile of each range
    float* slabs;           // [G][BM * BN]
    unsigned* flags;        // [G]
    unsigned epoch;
    unsigned* err;          // set to a non-zero code when a spin gives up
    unsigned spin_limit;    // polls of a partner's flag before the owner declares the launch failed (sdt_convsk_set_spin_limit)
    int korder;             // 8-wave kernels (convbf.hip): 0 tap-major K order, 1 chunk-major (sdt_convsk_set_k_order)
};

struct sk_norm_bwd {
    const void* y;          // forward output of the block below, element type TX of the launch
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    double* sums;
    float slope;
    int groups;
};

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sk_bf16x2 __attribute__((ext_vector_type(2)));
// Element type of the operands (TX: X, W and -- EPI 2 -- the forward output y of the block below) and of the output (TY): float, or
// __bf16 for the bf16-storage path (BASELINE config 4): products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, tensors bf16 in HBM.
// A K step is 128 BYTES of a row in both cases (32 fp32 / 64 bf16 channels), so the plan (byte offsets), the loader, the LDS tiles
// ([row][128 B + 16 B pad]) and the 16-byte fragment reads are the same code; a fragment read feeds four fp32 MFMAs (k = 4 j' + e,
// one product per lane) or ONE bf16 MFMA (k = 16 J + 8 (lane >> 5) + e, eight products per lane).
template <typename T> struct sk_is_bf16 { static constexpr bool value = false; };
template <> struct sk_is_bf16<__bf16> { static constexpr bool value = true; };
__device__ __forceinline__ float sk_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

#define SK_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// Scheduling hints of one bf16 K step before its barrier (sched_group_barrier wants literal counts, hence the recursion): a k-group is only
// NM = TM * TN MFMAs of 32 cycles -- 3 NM MFMAs against 3 NF fragment reads, NL LDS stores and NL global loads.  Every MFMA of k-groups 0
// and 1 is followed by its share of the next group's fragment reads, of one half of the stores and of the loads that re-fill the registers
// just stored; every MFMA of k-group 2 by its share of k-group 3's reads.
template <int G2, int Q, int NM, int NF, int NL>
__device__ __forceinline__ void sk_bf_interleave() {
    // the LDS stores of step s+1 and the global loads of step s+3 go with the FIRST k-group: the loads are issued as early in the step as
    // their registers are free (a load lands two steps before it is staged: the L2 latency under load is about one bf16 step)
    constexpr int lo = 0, hi = G2 == 0 ? NL : 0;
    constexpr int nr = (NF + NM - 1 - Q) / NM, nw = (hi - lo + NM - 1 - Q) / NM;
    SK_SGB(0x8, 1);
    if constexpr (nr > 0) SK_SGB(0x100, nr);
    if constexpr (nw > 0) {
        SK_SGB(0x200, nw);
        SK_SGB(0x20, nw);
    }
    if constexpr (Q + 1 < NM) sk_bf_interleave<G2, Q + 1, NM, NF, NL>();
    else if constexpr (G2 + 1 < 3) sk_bf_interleave<G2 + 1, 0, NM, NF, NL>();
}


// convbf.hip: launch of the bf16-shaped kernel (256-row tiles, 8 waves, one workgroup per CU) for a plan built with one workgroup per CU (256 x {256,128,64} or 128 x 128 tiles)
int convbf2_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, int bm, int bn, int epi,
                   hipStream_t s);
// ... and of its split-fp32 form (fp32 tensors; three bf16 planes made by the loader, six bf16 MFMAs per fragment pair): 256 x {128,64} or 128 x 128 tiles
// w3 != 0: `w` holds the weights pre-split into three bf16 planes (sdt_convsk_f32_w3)
int convx3_launch(const void* x, const void* w, const float* bias, void* y, const sk_args& A, double* stats, const sk_norm_bwd& nb, int bm, int bn, int epi,
                  int w3, hipStream_t s);
#ifdef SDT_TUNING
int convbf2_debug_mute_range(int r);  // convbf.hip's copy of the fault injector (device symbols are per translation unit)
#endif
