// Kernel laboratory (not part of the product): the skinny batched-decode matmul (csrc/qmm3.h) at 16/32/64 rows, HIP events.
// Built once per ablation (-DQMM3_ABL=n): tools/lab/run_qmm3_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tiny-llm_amd/csrc/qmm3.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    struct Shape { const char *name; int K, N; } shapes[] = {{"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        const size_t wwords = (size_t)K * N / 8;
        const int copies = 8;
        uint32_t *w, *sb; uint16_t *a; float *partial;
        CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&sb, (size_t)K * G * 4 * copies)); CK(hipMalloc(&a, (size_t)M * N * 2));
        CK(hipMemset(w, 0x5a, wwords * 4 * copies)); CK(hipMemset(sb, 0x3c, (size_t)K * G * 4 * copies)); CK(hipMemset(a, 0x3f, (size_t)M * N * 2));
        const Qmm3Plan pl = qmm3_plan(M, N, K);
        CK(hipMalloc(&partial, pl.partial_bytes));
        const int iters = 24;
        auto run = [&](int i) { Qmm3Args q{}; q.wt = w + (size_t)(i % copies) * wwords; q.sbt = sb + (size_t)(i % copies) * K * G; q.a = a; q.partial = partial; q.M = M; q.N = N; q.K = K; launch_qmm3_bf16(q, 0); };
        for (int i = 0; i < 4; ++i) run(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) run(i); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("  %-8s M=%d (MB%d TW%d LM%d, %d x %d workgroups, lds %zu): %7.1f us  weights %6.1f GB/s  %6.1f TFLOP/s\n", sh.name, M, pl.MB, pl.TW, pl.LM,
               pl.grid_x, pl.slices, pl.lds, us, wwords * 4 / us / 1e3, 2.0 * M * K * N / us / 1e6);
        CK(hipFree(w)); CK(hipFree(sb)); CK(hipFree(a)); CK(hipFree(partial));
    }
    return 0;
}
