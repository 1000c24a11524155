// Kernel laboratory (not part of the product): latency of the device-side synchronisation primitives a
// flag-ordered kernel chain would be built from (agent-scope atomics, release-store -> acquire-poll visibility),
// within one XCD and across XCDs.  All loops are bounded.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/prim_lab.hip -o tools/lab/prim_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

// T1: one thread, N dependent operations
__global__ void rt_kernel(int *line, int *sink, u64 *out, int n) {
    if (threadIdx.x != 0) return;
    u64 t0 = wall_clock64();
    int v = 0;
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(line, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1;
    u64 t1 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_load(line + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1;
    u64 t2 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(line + 64 + (v & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 1;
    u64 t3 = wall_clock64();
    for (int i = 0; i < n; ++i) { __hip_atomic_store(line + 128, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    u64 t4 = wall_clock64();
    out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3;
    *sink = v;
}

// T2: ping-pong between workgroup `a` and workgroup `b` of one launch (blockIdx -> XCD is round robin)
__global__ void pingpong_kernel(int *f0, int *f1, int a, int b, int n, u64 *out, int *err) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    u64 t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (me == 0) {
            __hip_atomic_store(f0, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(f1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < i) if (++spins > 100000) { *err = 1; return; }
        } else {
            int spins = 0;
            while (__hip_atomic_load(f0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < i) if (++spins > 100000) { *err = 1; return; }
            __hip_atomic_store(f1, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) out[0] = wall_clock64() - t0;
}

// T3: fan-in: every workgroup adds once to counter[blockIdx / per]; stamps per workgroup
template <int ORDER>
__global__ __launch_bounds__(256) void fanin_kernel(int *counters, int per, u64 *stamp) {
    if (threadIdx.x == 0) {
        stamp[2 * blockIdx.x] = wall_clock64();
        __hip_atomic_fetch_add(counters + (blockIdx.x / per) * 32, 1, ORDER, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[2 * blockIdx.x + 1] = wall_clock64();
    }
}
// T3b: the same with a returning add (the last arriver needs the count)
template <int ORDER>
__global__ __launch_bounds__(256) void fanin_ret_kernel(int *counters, int per, u64 *stamp, int *sink) {
    if (threadIdx.x == 0) {
        stamp[2 * blockIdx.x] = wall_clock64();
        int v = __hip_atomic_fetch_add(counters + (blockIdx.x / per) * 32, 1, ORDER, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0x7fffffff) *sink = 1;
        stamp[2 * blockIdx.x + 1] = wall_clock64();
    }
}

// T4: broadcast: workgroup 0 waits `delay` then release-stores `copies` flags; everyone else polls copy blockIdx % copies.
template <int ST, int LD>
__global__ __launch_bounds__(256) void bcast_kernel(int *flags, int copies, int val, u64 *stamp, int *err) {
    if (threadIdx.x >= 64) return;
    if (blockIdx.x == 0) {
        u64 t = wall_clock64();
        while (wall_clock64() - t < 300) {}  // 3 us: let every poller get resident
        if (threadIdx.x == 0) stamp[0] = wall_clock64();
        if ((int)threadIdx.x < copies) __hip_atomic_store(flags + threadIdx.x * 32, val, ST, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flags + (blockIdx.x % copies) * 32, LD, __HIP_MEMORY_SCOPE_AGENT) != val) {
            if (++spins > 100000) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        stamp[blockIdx.x] = wall_clock64();
    }
}

int main() {
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1e3 / khz;
    int *mem; CK(hipMalloc(&mem, 1 << 20)); CK(hipMemset(mem, 0, 1 << 20));
    u64 *out; CK(hipMalloc(&out, 1 << 20)); int *err = mem + 200000;
    std::vector<u64> h(1 << 17);
    const int n = 200;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rt_kernel, dim3(1), dim3(64), 0, 0, mem, mem + 1000, out, n);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, 32, hipMemcpyDeviceToHost));
    }
    printf("T1 dependent ops: agent fetch_add %.3f us | agent load %.3f us | workgroup fetch_add %.3f us | agent release store %.3f us\n",
           h[0] * us / n, h[1] * us / n, h[2] * us / n, h[3] * us / n);
    for (int b : {1, 2, 4, 8, 16, 9}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(mem, 0, 4096));
            hipLaunchKernelGGL(pingpong_kernel, dim3(b + 1), dim3(64), 0, 0, mem, mem + 64, 0, b, n, out, err);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, 8, hipMemcpyDeviceToHost));
        }
        int he; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
        printf("T2 ping-pong wg0 <-> wg%-2d: one-way %.3f us (err %d)\n", b, h[0] * us / n / 2, he);
    }
    for (int relaxed = 0; relaxed < 2; ++relaxed)
    for (int grid : {160, 608}) for (int per : {100000, 32, 1}) for (int ret = 0; ret < 2; ++ret) {
        double span = 0, lat = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(mem, 0, 1 << 19));
            if (ret && relaxed) hipLaunchKernelGGL(fanin_ret_kernel<__ATOMIC_RELAXED>, dim3(grid), dim3(256), 0, 0, mem, per, out, mem + 150000);
            else if (ret) hipLaunchKernelGGL(fanin_ret_kernel<__ATOMIC_ACQ_REL>, dim3(grid), dim3(256), 0, 0, mem, per, out, mem + 150000);
            else if (relaxed) hipLaunchKernelGGL(fanin_kernel<__ATOMIC_RELAXED>, dim3(grid), dim3(256), 0, 0, mem, per, out);
            else hipLaunchKernelGGL(fanin_kernel<__ATOMIC_RELEASE>, dim3(grid), dim3(256), 0, 0, mem, per, out);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, (size_t)grid * 16, hipMemcpyDeviceToHost));
            u64 lo = ~0ull, hi = 0; lat = 0;
            for (int i = 0; i < grid; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); lat += (h[2 * i + 1] - h[2 * i]) * us; }
            span = (hi - lo) * us; lat /= grid;
        }
        printf("T3 fan-in relaxed %d grid %3d, %6d wgs/counter, returning %d: span %.2f us, mean op %.2f us\n", relaxed, grid, per, ret, span, lat);
    }
    for (int relaxed = 0; relaxed < 2; ++relaxed)
    for (int grid : {160, 608}) for (int copies : {1, 8, 64}) {
        double mean = 0, mx = 0;
        for (int rep = 0; rep < 3; ++rep) {
            const int val = rep + 1 + copies * 10 + grid * 1000 + relaxed * 100000;
            if (relaxed) hipLaunchKernelGGL((bcast_kernel<__ATOMIC_RELAXED, __ATOMIC_RELAXED>), dim3(grid), dim3(256), 0, 0, mem + 65536, copies, val, out, err);
            else hipLaunchKernelGGL((bcast_kernel<__ATOMIC_RELEASE, __ATOMIC_ACQUIRE>), dim3(grid), dim3(256), 0, 0, mem + 65536, copies, val, out, err);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, (size_t)grid * 8, hipMemcpyDeviceToHost));
            mean = 0; mx = 0;
            for (int i = 1; i < grid; ++i) { double d = (double)((long long)(h[i] - h[0])) * us; mean += d; mx = std::max(mx, d); }
            mean /= grid - 1;
        }
        int he; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
        printf("T4 broadcast relaxed %d grid %3d copies %2d: store -> seen mean %.2f us, max %.2f us (err %d)\n", relaxed, grid, copies, mean, mx, he);
    }
    return 0;
}
