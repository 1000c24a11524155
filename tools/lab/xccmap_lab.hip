// Kernel laboratory (not part of the product), round 5: which XCD does workgroup b of a dispatch run on?
// A decode step's launches have grids of 192 / 32 / 160 / 1216 / 160 ... 4748 / 1 workgroups.  The L2 prefetcher of the AQL route
// (csrc/engine.hip) must bring a workgroup's weights into the L2 of the XCD that workgroup WILL run on: is that (b + c) mod 8 with a
// constant c, or does the round-robin pointer carry over from dispatch to dispatch (grids that are not multiples of 8 would then shift
// everything behind them)?  Records XCC_ID of every workgroup for a sequence of dispatches through hipLaunchKernelGGL on one stream.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/xccmap_lab.hip -o tools/lab/xccmap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void who(int *out, int threads_used) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int)(v & 15u);
    }
}
int main() {
    const int grids[] = {192, 32, 160, 1216, 160, 4748, 1, 192, 32, 160, 1216, 160, 4748, 1, 5, 3, 8, 256};
    const int blocks[] = {256, 256, 256, 256, 512, 256, 1024, 256, 256, 256, 256, 512, 256, 1024, 64, 64, 64, 64};
    const int n = sizeof(grids) / sizeof(int);
    int *buf; CK(hipMalloc(&buf, (size_t)n * 8192 * 4));
    CK(hipMemset(buf, 0xff, (size_t)n * 8192 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(who, dim3(grids[i]), dim3(blocks[i]), 0, st, buf + (size_t)i * 8192, 0);
        CK(hipStreamSynchronize(st));
        std::vector<int> h((size_t)n * 8192);
        CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const int *x = h.data() + (size_t)i * 8192;
            bool rr = true;
            for (int b = 0; b < grids[i]; ++b) rr = rr && x[b] == (x[0] + b) % 8;
            printf("rep %d dispatch %2d grid %4d x %4d: block 0 on XCD %d, round robin over blocks: %s | first 12:", rep, i, grids[i], blocks[i], x[0], rr ? "yes" : "NO");
            for (int b = 0; b < 12 && b < grids[i]; ++b) printf(" %d", x[b]);
            printf("\n");
        }
    }
    return 0;
}
