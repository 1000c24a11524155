// Kernel laboratory (not part of the product): how long after its start can a wave use its kernel arguments, with a struct
// argument and with flat arguments, with and without -mllvm -amdgpu-kernarg-preload-count=16 (profiles/r02_labs/kernarg_preload_r02.log).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Args { unsigned long long *out; const int *p1; const int *p2; int a, b, c, d; float e, f; const int *p3; int pad[8]; };
__global__ __launch_bounds__(256) void k_struct(const Args p) {
    const unsigned long long t0 = wall_clock64();
    const int v = p.a + p.b;                      // needs the kernarg words
    asm volatile("" :: "s"(v));
    const unsigned long long t1 = wall_clock64();
    const int w = p.p1[v & 3];                    // a dependent global load
    asm volatile("" :: "v"(w));
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { p.out[blockIdx.x * 4] = t0; p.out[blockIdx.x * 4 + 1] = t1; p.out[blockIdx.x * 4 + 2] = t2 + (w & 0); }
}
__global__ __launch_bounds__(256) void k_flat(unsigned long long *out, const int *p1, int a, int b) {
    const unsigned long long t0 = wall_clock64();
    const int v = a + b;
    asm volatile("" :: "s"(v));
    const unsigned long long t1 = wall_clock64();
    const int w = p1[v & 3];
    asm volatile("" :: "v"(w));
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t0; out[blockIdx.x * 4 + 1] = t1; out[blockIdx.x * 4 + 2] = t2 + (w & 0); }
}
int main() {
    const int wgs = 304;
    unsigned long long *out; int *p1; CK(hipMalloc(&out, wgs * 32)); CK(hipMalloc(&p1, 64)); CK(hipMemset(p1, 0, 64));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
    std::vector<unsigned long long> h(wgs * 4);
    for (int variant = 0; variant < 2; ++variant) {
        double d1 = 0, d2 = 0; int n = 0;
        for (int rep = 0; rep < 20; ++rep) {
            Args a{}; a.out = out; a.p1 = p1; a.a = rep; a.b = 1;
            if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(wgs), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL(k_flat, dim3(wgs), dim3(256), 0, 0, out, p1, rep, 1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, wgs * 32, hipMemcpyDeviceToHost));
            if (rep < 4) continue;
            for (int w = 0; w < wgs; ++w) { d1 += (double)(h[w * 4 + 1] - h[w * 4]); d2 += (double)(h[w * 4 + 2] - h[w * 4 + 1]); ++n; }
        }
        printf("%s: wave start -> kernarg words usable %.0f ns, + dependent global load %.0f ns (mean over %d workgroups)\n", variant ? "flat args  " : "struct args", d1 / n * 1e6 / rate, d2 / n * 1e6 / rate, n);
    }
    return 0;
}
