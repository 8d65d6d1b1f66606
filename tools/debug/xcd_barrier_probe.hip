// Gate of VERDICT r3 item 3 (Conv1d chains as per-clip clusters inside ONE persistent launch): what does a barrier among the 8 workgroups of a
// clip cost when those workgroups sit on one XCD and exchange a layer's activations?  Kill criterion stated there: > 2 us per barrier.
//
// 256 workgroups (one per CU), cluster = the 8 workgroups whose block ids are b, b + 8, ..., b + 56 within a window of 64 consecutive ids (same id
// mod 8 = same XCD by the observed dispatch).  Per iteration every workgroup of a cluster
//   (1) stores PAYLOAD bytes with write-through (sc1) stores, drains them (vmcnt(0)), __syncthreads,
//   (2) lane 0 adds 1 to the cluster's counter (relaxed, agent scope) and polls it with relaxed loads + s_sleep until all 8 have arrived,
//   (3) __syncthreads, then reads its RIGHT NEIGHBOUR's payload with sc1 loads (L1 bypass: no acquire fence needed, MI355X_MICROARCH.md
//       "valid forms") and checks every word.
// Variants: payload 0 (barrier alone), 4 KB, 16 KB (a 64 x 64 fp32 tile: what a workgroup of the design would hand on per layer);
// and the fence form: plain stores + release fence / acquire fence instead of sc1 both sides.
// Build: hipcc --offload-arch=gfx950 -O3 tools/debug/xcd_barrier_probe.hip -o tools/bin/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool FENCE>
__global__ __launch_bounds__(256) void probe(unsigned* counters, unsigned* payload, int words, int iters, long long* cycles, unsigned* bad, unsigned* xcc) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int window = b >> 6, slot = (b >> 3) & 7, xc = b & 7;  // cluster = (window, xc); member = slot
    const int cluster = window * 8 + xc;
    gu32* cnt = (gu32*)(counters + cluster * 32);
    // two payload slots per member, alternating by iteration: a fast member's NEXT store must not land in the slot its left neighbour is still reading
    // (the design's blocks write different tensors, so it has no such hazard; the first version of this probe had it and counted "wrong words")
    unsigned* mine0 = payload + (size_t)((cluster * 8 + slot) * 2) * words;
    const unsigned* right0 = payload + (size_t)((cluster * 8 + ((slot + 1) & 7)) * 2) * words;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    unsigned nbad = 0;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        unsigned* mine = mine0 + (size_t)(it & 1) * words;
        const unsigned* right = right0 + (size_t)(it & 1) * words;
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, words * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)right, 0, words * 4, 0x00020000);
        const unsigned tag = (unsigned)(it + 1) * 0x10001u + (unsigned)slot;
        for (int w = tid * 4; w < words; w += 1024) {
            const u32x4 v = {tag, tag + 1u, tag + 2u, tag + 3u};
            if (FENCE) *(u32x4*)(mine + w) = v;
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsM, w * 4, 0, 16);  // sc1: write-through
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (FENCE) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 8u * (unsigned)(it + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 24)) { atomicAdd(bad, 1000000u); break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const unsigned rtag = (unsigned)(it + 1) * 0x10001u + (unsigned)((slot + 1) & 7);
        for (int w = tid * 4; w < words; w += 1024) {
            u32x4 v;
            if (FENCE) v = *(const u32x4*)(right + w);
            else v = __builtin_amdgcn_raw_buffer_load_b128(rsR, w * 4, 0, 16);  // sc1: L2-served
            nbad += (v[0] != rtag) + (v[1] != rtag + 1u) + (v[2] != rtag + 2u) + (v[3] != rtag + 3u);
        }
        __syncthreads();  // everybody has read before the next iteration overwrites (the neighbour's next store is behind the NEXT barrier anyway)
    }
    const long long t1 = wall_clock64();
    if (tid == 0) cycles[b] = t1 - t0;
    if (nbad) atomicAdd(bad, nbad);
}

template <bool FENCE>
static void run(int words, int iters, unsigned* dcnt, unsigned* dpay, long long* dcyc, unsigned* dbad, unsigned* dxcc) {
    CK(hipMemset(dcnt, 0, 32 * 32 * 4));
    CK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL(probe<FENCE>, dim3(256), dim3(256), 0, 0, dcnt, dpay, words, 3, dcyc, dbad, dxcc);  // warm
    CK(hipMemset(dcnt, 0, 32 * 32 * 4));
    hipLaunchKernelGGL(probe<FENCE>, dim3(256), dim3(256), 0, 0, dcnt, dpay, words, iters, dcyc, dbad, dxcc);
    CK(hipDeviceSynchronize());
    std::vector<long long> cyc(256);
    std::vector<unsigned> xcc(256);
    unsigned bad;
    CK(hipMemcpy(cyc.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(xcc.data(), dxcc, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    long long mx = 0, mn = 1ll << 60;
    for (auto c : cyc) mx = std::max(mx, c), mn = std::min(mn, c);
    int same = 0;
    for (int b = 0; b < 256; ++b) same += xcc[b] == xcc[(b & ~63) | (b & 7)];  // every member on the XCD of the cluster's first member?
    printf("%-34s payload %6d B: %.2f us per iteration (slowest workgroup; fastest %.2f), wrong words %u, members on their cluster's XCD %d/256\n",
           FENCE ? "plain stores + release/acquire" : "sc1 stores + sc1 loads", words * 4, mx / 100.0 / iters, mn / 100.0 / iters, bad, same);
}

int main() {
    unsigned *dcnt, *dpay, *dbad, *dxcc;
    long long* dcyc;
    CK(hipMalloc(&dcnt, 32 * 32 * 4)); CK(hipMalloc(&dpay, (size_t)256 * 2 * 16384)); CK(hipMalloc(&dbad, 4)); CK(hipMalloc(&dxcc, 1024)); CK(hipMalloc(&dcyc, 2048));
    const int iters = 500;
    for (int words : {0, 1024, 4096}) {
        run<false>(words, iters, dcnt, dpay, dcyc, dbad, dxcc);
        run<true>(words, iters, dcnt, dpay, dcyc, dbad, dxcc);
    }
    return 0;
}
