// Kernel laboratory (not part of the product): a PERSISTENT kernel walking a chain of dependent GEMV-like phases with a
// device-side grid barrier (two-level relaxed agent-scope counters, activations in uncached memory), against the same
// chain as separate launches in a hipGraph.  Variant "prefetch": each workgroup puts the weight slice of its first
// virtual block of the NEXT phase in flight before it enters the barrier.
// All spins are bounded.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/mega_lab.hip -o tools/lab/mega_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct MegaArgs {
    int phases, n, spin_limit, prefetch, sc1;
    int pattern[4];
    int *sync;  // per phase 16 lines of 32 ints: [0..13] level-1 counters, [14] level-2, [15] done
    int *err;
    uint32_t *act0, *act1;
    const u32x4 *w;
    size_t wstride;  // u32x4 units per phase
    u64 *stamp;      // [phases][2]: max barrier-exit, max phase end (atomicMax by one lane per WG) -- coarse
};

// `pending` = number of this wave's prefetch loads that may stay in flight across the barrier (issued AFTER its stores:
// vmcnt retires in order, so "at most `pending` outstanding" means every store has landed)
__device__ __forceinline__ void grid_barrier(const MegaArgs &p, int phase, bool pending) {
    if (pending) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 256) {  // the sync wave holds no weight loads: its atomics and polls are not queued behind any
        int *s = p.sync + (size_t)phase * 32 * 32;
        const int grp = blockIdx.x >> 5, ngrp = ((int)gridDim.x + 31) >> 5;
        const int gsz = min(32, (int)gridDim.x - grp * 32);
        if (__hip_atomic_fetch_add(s + grp * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz - 1)
            if (__hip_atomic_fetch_add(s + 30 * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1)
                __hip_atomic_store(s + 31 * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(s + 31 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (++spins > p.spin_limit) { atomicAdd(&p.err[0], 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(320) void mega_kernel(const MegaArgs p) {
    __shared__ uint32_t s_part[4];
    const int tid = threadIdx.x;
    u32x4 wv[10];
    bool have = false;
    for (int k = 0; k < p.phases; ++k) {
        const int nvb = p.pattern[k & 3];
        const uint32_t *in = (k & 1) ? p.act1 : p.act0;
        uint32_t *out = (k & 1) ? p.act0 : p.act1;
        const u32x4 *wk = p.w + (size_t)(k % 40) * p.wstride;
        for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
            if (!have && tid < 256) {
                const u32x4 *wp = wk + (size_t)vb * 256 * 10 + tid;
#pragma unroll
                for (int i = 0; i < 10; ++i) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * 256);
            }
            have = false;
            uint32_t s = 0;
            if (tid < 256) for (int i = tid; i < p.n; i += 256) s += p.sc1 ? __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : in[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            __syncthreads();
            if ((tid & 63) == 0 && tid < 256) s_part[tid >> 6] = s;
            __syncthreads();
            s = s_part[0] + s_part[1] + s_part[2] + s_part[3];
            if (tid == 0 && s != (uint32_t)p.n * (uint32_t)k) atomicAdd(&p.err[1], 1);
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < 10; ++i) x ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
            const uint32_t v = s / (uint32_t)p.n + 1u + (x == 0x9e3779b9u ? 1u : 0u);
            const int per = (p.n + nvb - 1) / nvb;
            if (tid < 256) for (int j = vb * per + tid; j < min(p.n, (vb + 1) * per); j += 256) out[j] = v;
        }
        if (p.prefetch && tid < 256 && k + 1 < p.phases && (int)blockIdx.x < p.pattern[(k + 1) & 3]) {
            const u32x4 *wp = p.w + (size_t)((k + 1) % 40) * p.wstride + (size_t)blockIdx.x * 256 * 10 + tid;
#pragma unroll
            for (int i = 0; i < 10; ++i) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * 256);
            have = true;
        }
        grid_barrier(p, k, have);
        if (tid == 0) p.stamp[(size_t)k * 1024 + blockIdx.x] = wall_clock64();
    }
}

int main() {
    const int LAYERS = 36, n = 1280, phases = LAYERS * 4;
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1e3 / khz;
    const size_t wstride = (size_t)608 * 256 * 10;  // u32x4 per phase (25 MB)
    u32x4 *w; CK(hipMalloc(&w, wstride * 16 * 40)); CK(hipMemset(w, 1, wstride * 16 * 40));
    uint32_t *act[2];
    for (auto &a : act) CK(hipExtMallocWithFlags((void **)&a, n * 4, hipDeviceMallocUncached));
    int *sync, *err; CK(hipMalloc(&sync, (size_t)phases * 32 * 128)); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
    u64 *stamp; CK(hipMalloc(&stamp, (size_t)phases * 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mega_kernel, 320, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, occupancy %d WGs/CU\n", prop.multiProcessorCount, occ);
    for (int sc1 = 0; sc1 < 2; ++sc1)
    for (int prefetch = 0; prefetch < 2; ++prefetch)
    for (int grid : {256, 512, 608, 768}) {
        if (grid > occ * prop.multiProcessorCount) continue;
        MegaArgs a{}; a.phases = phases; a.n = n; a.spin_limit = 200000; a.prefetch = prefetch; a.sc1 = sc1;
        a.pattern[0] = 192; a.pattern[1] = 160; a.pattern[2] = 608; a.pattern[3] = 160;
        a.sync = sync; a.err = err; a.act0 = act[0]; a.act1 = act[1]; a.w = w; a.wstride = wstride; a.stamp = stamp;
        float best = 1e9f;
        std::vector<u64> h(phases);
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(sync, 0, (size_t)phases * 32 * 128, st));
            CK(hipMemsetAsync(act[0], 0, n * 4, st));
            CK(hipMemsetAsync(stamp, 0, (size_t)phases * 1024 * 8, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(mega_kernel, dim3(grid), dim3(320), 0, st, a);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        { std::vector<u64> hs((size_t)phases * 1024); CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
          for (int k = 0; k < phases; ++k) { u64 m = 0; for (int b = 0; b < grid; ++b) m = std::max(m, hs[(size_t)k * 1024 + b]); h[k] = m; } }
        std::vector<uint32_t> ha(n); CK(hipMemcpy(ha.data(), act[phases & 1], n * 4, hipMemcpyDeviceToHost));
        int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost)); CK(hipMemset(err, 0, 8));
        int bad = 0; for (int i = 0; i < n; ++i) bad += ha[i] != (uint32_t)phases;
        double ph[4] = {0, 0, 0, 0};
        for (int k = 4; k < phases; ++k) ph[k & 3] += (double)(h[k] - h[k - 1]) * us;
        printf("sc1 %d prefetch %d grid %3d: kernel %8.1f us = %.2f us/phase | per phase kind (192,160,608,160 vbs): %.2f %.2f %.2f %.2f us | spin overruns %d stale %d final-bad %d\n",
               sc1, prefetch, grid, best * 1e3, best * 1e3 / phases, ph[0] / (LAYERS - 1), ph[1] / (LAYERS - 1), ph[2] / (LAYERS - 1), ph[3] / (LAYERS - 1), herr[0], herr[1], bad);
    }
    return 0;
}
