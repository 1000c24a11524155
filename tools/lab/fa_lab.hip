// Kernel laboratory (not part of the product): the paged FlashAttention kernel (tl_paged_attention, L > 8) on one long
// sequence, timed with HIP events.  Built once per ablation (-DFA_ABL=n when compiling attention.hip): tools/lab/run_fa_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/tinyllm_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 2048, ctx = argc > 2 ? atoi(argv[2]) : 8192;
    const int Hq = 32, Hkv = 8, D = 128, page = 128, pages = ctx / page;
    uint16_t *q, *out, *kp, *vp; int32_t *table, *cl; void *ws;
    CK(hipMalloc(&q, (size_t)Hq * L * D * 2)); CK(hipMalloc(&out, (size_t)Hq * L * D * 2));
    CK(hipMalloc(&kp, (size_t)pages * Hkv * page * D * 2)); CK(hipMalloc(&vp, (size_t)pages * Hkv * page * D * 2));
    std::vector<uint16_t> h((size_t)pages * Hkv * page * D); for (auto &v : h) v = (uint16_t)(0x3c00 + (rand() & 0x1ff));
    CK(hipMemcpy(kp, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vp, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data(), (size_t)Hq * L * D * 2, hipMemcpyHostToDevice));
    std::vector<int32_t> ht(pages); for (int i = 0; i < pages; ++i) ht[i] = (i * 7) % pages;  // 7 and pages are coprime for powers of two
    CK(hipMalloc(&table, pages * 4)); CK(hipMemcpy(table, ht.data(), pages * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&cl, 4)); CK(hipMemcpy(cl, &ctx, 4, hipMemcpyHostToDevice));
    const size_t wsb = tl_paged_attention_workspace_bytes(Hq, L, D, page, pages, Hq, Hkv, ctx);
    ws = nullptr; if (wsb) CK(hipMalloc(&ws, wsb));
    auto run = [&]() { if (tl_paged_attention(q, kp, vp, table, cl, out, Hq, L, D, pages, page, pages, Hq, Hkv, 0.0884f, 1, ctx, TL_BF16, ws, wsb, nullptr) != 0) { printf("failed: %s\n", tl_last_error()); exit(1);} };
    for (int i = 0; i < 2; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 5;
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) run(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double flop = 4.0 * Hq * D * ((double)L * (ctx - L) + (double)L * L / 2);  // causal chunk at the end of the context
    printf("  L=%d ctx=%d: %9.1f us  %7.1f TFLOP/s\n", L, ctx, us, flop / us / 1e6);
    return 0;
}
