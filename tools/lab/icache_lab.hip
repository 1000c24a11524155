// Kernel laboratory (not part of the product): what does the FIRST pass through a stretch of straight-line code cost?
// One wave per workgroup runs the same N-instruction block twice (a two-iteration loop, same code addresses) and stamps the
// wall clock around each pass; the kernel is launched several times back to back (is the instruction cache kept between
// launches of the same kernel?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int N>
__global__ __launch_bounds__(64) void code_kernel(float *out, unsigned long long *stamps, int passes) {
    float a = threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    unsigned long long t[5];
    t[0] = wall_clock64();
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f);
        }
        asm volatile("s_nop 0" ::: "memory");
        if (p < 4) t[p + 1] = wall_clock64();
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) for (int k = 0; k < 5; ++k) stamps[blockIdx.x * 5 + k] = k <= passes ? t[k] : 0;
}

template <int N>
void run(const char *name) {
    const int wgs = 256;
    float *out; unsigned long long *st; CK(hipMalloc(&out, wgs * 64 * 4)); CK(hipMalloc(&st, wgs * 5 * 8));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int launch = 0; launch < 3; ++launch) {
        hipLaunchKernelGGL(code_kernel<N>, dim3(wgs), dim3(64), 0, 0, out, st, 3);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(wgs * 5); CK(hipMemcpy(h.data(), st, wgs * 5 * 8, hipMemcpyDeviceToHost));
        double p[3] = {0, 0, 0};
        for (int w = 0; w < wgs; ++w) for (int k = 0; k < 3; ++k) p[k] += (double)(h[w * 5 + k + 1] - h[w * 5 + k]);
        printf("  %-10s launch %d: pass 1 %.2f us, pass 2 %.2f us, pass 3 %.2f us (mean over %d waves)\n", name, launch, p[0] / wgs * 1e3 / rate,
               p[1] / wgs * 1e3 / rate, p[2] / wgs * 1e3 / rate, wgs);
    }
    // back-to-back launches of a ONE-pass kernel: the per-launch time the engine would see
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(code_kernel<N>, dim3(wgs), dim3(64), 0, 0, out, st, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(code_kernel<N>, dim3(wgs), dim3(64), 0, 0, out, st, 1); CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(code_kernel<N>, dim3(wgs), dim3(64), 0, 0, out, st, 2); CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize()); float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("  %-10s back-to-back: one pass %.2f us per launch, two passes %.2f us per launch\n", name, ms * 1e3 / 50, ms2 * 1e3 / 50);
    CK(hipFree(out)); CK(hipFree(st));
}

int main() {
    run<256>("1 KiB");
    run<1024>("4 KiB");
    run<4096>("16 KiB");
    run<8192>("32 KiB");
    return 0;
}
