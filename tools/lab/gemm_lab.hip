// Kernel laboratory (not part of the product): the W4 prefill GEMM (tl_quantized_matmul, csrc/qmm.hip) at prefill-chunk
// shapes, timed with HIP events.  Built once per ablation (-DQMM_ABL=n when compiling qmm.hip): tools/lab/run_gemm_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/tinyllm_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    struct Shape { const char *name; int K, N; } shapes[] = {{"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_us = 0, total_flop = 0;
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        uint32_t *w; uint16_t *s, *b, *a, *out; void *ws;
        CK(hipMalloc(&w, (size_t)K * N / 2)); CK(hipMalloc(&s, (size_t)K * G * 2)); CK(hipMalloc(&b, (size_t)K * G * 2));
        CK(hipMalloc(&a, (size_t)M * N * 2)); CK(hipMalloc(&out, (size_t)M * K * 2));
        std::vector<uint32_t> hw((size_t)K * N / 8); for (auto &v : hw) v = (uint32_t)rand() * 2654435761u;
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint16_t> hs((size_t)K * G), hb((size_t)K * G), ha((size_t)M * N); for (auto &v : ha) v = (uint16_t)(0x3c00 + (rand() & 0xff));
        for (auto &v : hs) v = (uint16_t)(0x3c00 + (rand() & 0x7f)); for (auto &v : hb) v = (uint16_t)(0xbc00 + (rand() & 0x7f));  // per-group values: a wrong group index shows in the sums
        CK(hipMemcpy(s, hs.data(), hs.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        const size_t wsb = tl_quantized_matmul_workspace_bytes(M, N, K, TL_BF16, 1, 1);
        ws = nullptr; if (wsb) CK(hipMalloc(&ws, wsb));
        auto run = [&]() { if (tl_quantized_matmul(s, b, a, w, out, M, N, K, 128, 4, TL_BF16, 1, 1, ws, wsb, nullptr) != 0) { printf("matmul failed: %s\n", tl_last_error()); exit(1);} };
        for (int i = 0; i < 3; ++i) run();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) run(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, flop = 2.0 * M * K * N;
        std::vector<uint16_t> ho((size_t)M * K), ho2((size_t)M * K); CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0xff, ho.size() * 2)); run(); CK(hipDeviceSynchronize()); CK(hipMemcpy(ho2.data(), out, ho2.size() * 2, hipMemcpyDeviceToHost));
        size_t ndiff = 0; for (size_t i = 0; i < ho.size(); ++i) ndiff += ho[i] != ho2[i];
        if (ndiff) printf("  !! two launches differ in %zu of %zu elements\n", ndiff, ho.size());
        unsigned long long fnv = 1469598103934665603ull; for (uint16_t v : ho) { fnv ^= v; fnv *= 1099511628211ull; }  // same seeds in every build: equal sums = equal bits
        printf("  %-8s M=%d K=%d N=%d split=%d: %8.1f us  %7.1f TFLOP/s  out %016llx\n", sh.name, M, K, N, tl_quantized_matmul_split_k(M, N, K, 1, 1), us, flop / us / 1e6, fnv);
        total_us += us; total_flop += flop;
        CK(hipFree(w)); CK(hipFree(s)); CK(hipFree(b)); CK(hipFree(a)); CK(hipFree(out)); if (ws) CK(hipFree(ws));
    }
    printf("  layer total: %8.1f us  %7.1f TFLOP/s\n", total_us, total_flop / total_us / 1e6);
    return 0;
}
