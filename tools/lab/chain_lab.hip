// Kernel laboratory (not part of the product): how fast can a chain of DEPENDENT small kernels run when the
// dependency is a device-side flag (release/acquire at agent scope) instead of a stream-order kernel boundary?
//   mode 0: one stream, no flags (today's engine: ~1.25 us boundary per kernel)
//   mode 1: one stream, flags on (protocol overhead alone)
//   mode 2/3: kernels alternate over 2/3 captured streams, ordering ONLY through flags; each kernel puts its weight
//             slice in flight before it spins, so launch latency and HBM latency overlap the producer's execution.
// Every spin is bounded (no hang on a mis-ordered dispatch): overrun sets err[0].  err[1] counts stale activation reads.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/chain_lab.hip -o tools/lab/chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct ChainArgs {
    int idx, prev_grid, n, wl, epoch, sleep, two_level;  // n activation words; wl = 16-byte weight loads per thread (<= 10)
    int *flags, *err;
    const uint32_t *in;
    uint32_t *out;
    const u32x4 *w;
    u64 *stamp;  // [kernels][1024 wgs][4]: start, spin exit, end, -
    int spin_limit;
};

template <bool FLAG>
__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs p) {
    __shared__ uint32_t s_part[4];
    const int tid = threadIdx.x;
    u64 *my = p.stamp + ((size_t)p.idx * 1024 + blockIdx.x) * 4;
    if (tid == 0) my[0] = wall_clock64();
    u32x4 wv[10];
    const u32x4 *wp = p.w + (size_t)blockIdx.x * 256 * 10 + tid;
#pragma unroll
    for (int i = 0; i < 10; ++i) wv[i] = __builtin_nontemporal_load(wp + (size_t)(i < p.wl ? i : 0) * 256);
    if constexpr (FLAG) {
        if (p.idx > 0) {
            if (p.two_level) {
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(&p.flags[(p.idx - 1) * 2048 + 2016], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.epoch) {
                        if (++spins > p.spin_limit) { atomicAdd(&p.err[0], 1); break; }
                        for (int z = 0; z < p.sleep; ++z) __builtin_amdgcn_s_sleep(1);
                    }
                }
            } else if (tid < 64) {
                const int nc = (p.prev_grid + 31) >> 5;
                const int want = tid < nc ? min(32, p.prev_grid - tid * 32) * p.epoch : 0;
                int spins = 0;
                while (true) {
                    const int got = tid < nc ? __hip_atomic_load(&p.flags[(p.idx - 1) * 2048 + tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    if (__all(got >= want)) break;
                    if (++spins > p.spin_limit) { if (tid == 0) atomicAdd(&p.err[0], 1); break; }
                    for (int z = 0; z < p.sleep; ++z) __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) my[1] = wall_clock64();
    uint32_t s = 0;
    for (int i = tid; i < p.n; i += 256) s += p.in[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    s = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (tid == 0 && s != (uint32_t)p.n * (uint32_t)p.idx) atomicAdd(&p.err[1], 1);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) x ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    const uint32_t v = s / (uint32_t)p.n + 1u + (x == 0x9e3779b9u ? 1u : 0u);
    const int per = (p.n + gridDim.x - 1) / gridDim.x;
    for (int j = blockIdx.x * per + tid; j < min(p.n, (int)(blockIdx.x + 1) * per); j += 256) p.out[j] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if constexpr (FLAG) {
            int *f = p.flags + p.idx * 2048;
            if (p.two_level) {
                const int grp = blockIdx.x >> 5, ngrp = ((int)gridDim.x + 31) >> 5;
                const int gsz = min(32, (int)gridDim.x - grp * 32);
                if (__hip_atomic_fetch_add(f + grp * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz * p.epoch - 1)
                    if (__hip_atomic_fetch_add(f + 1984, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp * p.epoch - 1)
                        __hip_atomic_store(f + 2016, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(f + (blockIdx.x >> 5) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        my[2] = wall_clock64();
    }
}

__global__ void init_kernel(u64 *stamp, int kernels, int *flags, uint32_t *act, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kernels * 2048; i += gridDim.x * blockDim.x) flags[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) act[i] = 0;
}

int main(int argc, char **argv) {
    const int LAYERS = 36, n = 1280;
    const int pattern[4] = {192, 160, 608, 160};
    const int kernels = LAYERS * 4;
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us_per_tick = 1e3 / khz;
    const size_t wbytes = (size_t)1 << 30;
    u32x4 *w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    const bool uncached = !getenv("LAB_CACHED");
    uint32_t *act[2];
    for (auto &a : act) { if (uncached) CK(hipExtMallocWithFlags((void **)&a, n * 4, hipDeviceMallocUncached)); else CK(hipMalloc(&a, n * 4)); }
    printf("activations %s\n", uncached ? "uncached" : "cached");
    int *flags, *err; CK(hipMalloc(&flags, (size_t)kernels * 2048 * 4)); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
    u64 *stamp; CK(hipMalloc(&stamp, (size_t)kernels * 1024 * 32));
    hipStream_t st[3]; for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join[3]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto &j : join) CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));

    for (int wl : {10})
    for (int two : {0, 1}) for (int slp : {1, 8, 32})
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0 && (two || slp != 1)) continue;
        const int ns = mode < 2 ? 1 : mode;
        const bool flag = mode >= 1;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(init_kernel, dim3(64), dim3(256), 0, st[0], stamp, kernels, flags, act[0], n);
        if (ns > 1) { CK(hipEventRecord(fork, st[0])); for (int s = 1; s < ns; ++s) CK(hipStreamWaitEvent(st[s], fork, 0)); }
        size_t woff = 0;
        for (int k = 0; k < kernels; ++k) {
            const int grid = pattern[k & 3];
            ChainArgs a{};
            a.idx = k; a.prev_grid = k ? pattern[(k - 1) & 3] : 0; a.n = n; a.wl = wl; a.flags = flags; a.err = err;
            a.in = act[k & 1]; a.out = act[(k + 1) & 1]; a.stamp = stamp; a.spin_limit = 20000; a.epoch = 1; a.sleep = slp; a.two_level = two;
            const size_t need = (size_t)grid * 256 * 10;  // u32x4 units
            if (woff + need > wbytes / 16) woff = 0;
            a.w = w + woff; woff += need;
            hipStream_t s = st[k % ns];
            if (flag) hipLaunchKernelGGL(chain_kernel<true>, dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(chain_kernel<false>, dim3(grid), dim3(256), 0, s, a);
        }
        for (int s = 1; s < ns; ++s) { CK(hipEventRecord(join[s], st[s])); CK(hipStreamWaitEvent(st[0], join[s], 0)); }
        CK(hipStreamEndCapture(st[0], &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st[0]));
        CK(hipStreamSynchronize(st[0]));
        const int reps = 20;
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, st[0]));
        CK(hipEventRecord(e1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<u64> hs((size_t)kernels * 1024 * 4); CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<u64> h((size_t)kernels * 4);
        for (int k = 0; k < kernels; ++k) {
            u64 lo = ~0ull, go = 0, hi = 0;
            for (int b = 0; b < pattern[k & 3]; ++b) { const u64 *q = &hs[((size_t)k * 1024 + b) * 4]; lo = std::min(lo, q[0]); go = std::max(go, q[1]); hi = std::max(hi, q[2]); }
            h[4 * k] = lo; h[4 * k + 1] = go; h[4 * k + 2] = hi;
        }
        std::vector<uint32_t> ha(n); CK(hipMemcpy(ha.data(), act[kernels & 1], n * 4, hipMemcpyDeviceToHost));
        int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost)); CK(hipMemset(err, 0, 8));
        int bad = 0; for (int i = 0; i < n; ++i) bad += ha[i] != (uint32_t)kernels;
        const double span = (double)(h[4 * (kernels - 1) + 2] - h[0]) * us_per_tick;
        double body = 0, wait = 0, gap = 0;
        for (int k = 0; k < kernels; ++k) {
            body += (double)(h[4 * k + 2] - h[4 * k + 1]) * us_per_tick;   // spin exit -> end
            wait += (double)(h[4 * k + 1] - h[4 * k]) * us_per_tick;       // start -> spin exit
            if (k) gap += (double)((long long)(h[4 * k + 1] - h[4 * (k - 1) + 2])) * us_per_tick;  // producer end -> consumer go
        }
        printf("two_level %d sleep %2d wl=%2d mode %d (streams %d, flags %d): graph %8.1f us/launch | in-kernel span %8.1f us = %.2f us/kernel | body %.2f  start->go %.2f  prodend->go %.2f us | spin overruns %d stale %d final-bad %d\n",
               two, slp, wl, mode, ns, (int)flag, ms * 1e3 / reps, span, span / kernels, body / kernels, wait / kernels, gap / (kernels - 1), herr[0], herr[1], bad);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
