// Kernel laboratory (not part of the product): the register-resident batched-decode matmul (csrc/qmm6.h) at the Qwen3-4B projection
// shapes, HIP events, next to the K-sliced skinny matmul + slice reduction (csrc/qmm3.h) on the same inputs.  With -DQMM6_TRACE the
// kernel leaves per-wave wall-clock stamps at its phase boundaries; the lab prints their mean distance from the wave's start.
// usage: qmm6_lab <rows> [1 = the weighted rows of qkv / gate|up / lm_head in fragment order]     build + run: tools/lab/run_qmm6_lab.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../tiny-llm_amd/csrc/qmm3.h"
#include "../../tiny-llm_amd/csrc/qmm6.h"
#include "../../tiny-llm_amd/csrc/qmm7.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

static uint32_t rng_state = 12345u;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    struct Shape { const char *name; int K, N, epi; } shapes[] = {{"qkv", 6144, 2560, EPI_STORE}, {"o", 2560, 4096, EPI_RESIDUAL}, {"gate_up", 19456, 2560, EPI_SWIGLU},
                                                               {"down", 2560, 9728, EPI_RESIDUAL}, {"lm_head", 151936, 2560, EPI_STORE}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        const size_t wwords = (size_t)K * N / 8, swords = (size_t)K * G;
        const int copies = (size_t)wwords * 4 > (64u << 20) ? 3 : 8;
        uint32_t *w, *sb; uint16_t *a, *out, *res, *nw_dev, *outw; float *partial, *ss, *ssout;
        CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&sb, swords * 4 * copies)); CK(hipMalloc(&a, (size_t)(M + 15) / 16 * 16 * N * 2)); CK(hipMemset(a, 0, (size_t)(M + 15) / 16 * 16 * N * 2));
        CK(hipMalloc(&out, (size_t)M * K * 2)); CK(hipMalloc(&res, (size_t)M * K * 2)); CK(hipMalloc(&outw, (size_t)M * K * 2));
        CK(hipMalloc(&nw_dev, (size_t)K * 2)); CK(hipMalloc(&ss, (size_t)M * 160 * 4)); CK(hipMalloc(&ssout, (size_t)M * (K / 16) * 4));
        {
            std::vector<uint32_t> hw(wwords), hs(swords); std::vector<uint16_t> ha((size_t)M * N), hr((size_t)M * K), hn(K);
            for (auto &x : hw) x = rnd();
            for (auto &x : hs) { const float sc = 0.01f + (rnd() >> 8) * (0.01f / 16777216.f), be = ((int)(rnd() >> 8) - 8388608) * (0.1f / 8388608.f); x = (uint32_t)bf16_of(sc) | ((uint32_t)bf16_of(be) << 16); }
            for (auto &x : ha) x = bf16_of(((int)(rnd() >> 8) - 8388608) * (1.0f / 8388608.f));
            for (auto &x : hr) x = bf16_of(((int)(rnd() >> 8) - 8388608) * (1.0f / 8388608.f));
            for (auto &x : hn) x = bf16_of(0.5f + (rnd() >> 8) * (1.0f / 16777216.f));
            for (int c = 0; c < copies; ++c) { CK(hipMemcpy(w + c * wwords, hw.data(), wwords * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sb + c * swords, hs.data(), swords * 4, hipMemcpyHostToDevice)); }
            CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(nw_dev, hn.data(), hn.size() * 2, hipMemcpyHostToDevice));
            std::vector<float> hss((size_t)M * 160, 10.f); CK(hipMemcpy(ss, hss.data(), hss.size() * 4, hipMemcpyHostToDevice));
        }
        const int frag = (argc > 2 ? atoi(argv[2]) : 0) && sh.epi != EPI_RESIDUAL;
        const Qmm6Plan pl = qmm6_plan(M, N, K, frag != 0);
        const Qmm3Plan p3 = qmm3_plan(M, N, K, -1);
        CK(hipMalloc(&partial, std::max<size_t>(p3.partial_bytes, 16)));
        if (!pl.ok) { printf("%-8s rows %d: no plan\n", sh.name, M); continue; }
        const int epi = sh.epi;
        unsigned long long *pb = nullptr;
        const size_t nwaves = (size_t)pl.wgs * pl.row_blocks * QM6_WAVES;
        CK(hipMalloc(&pb, nwaves * 16 * 8)); CK(hipMemset(pb, 0, nwaves * 16 * 8));
        auto mm6 = [&](int i, unsigned long long *prof) {
            Qmm6Args q{}; q.wt = w + (size_t)(i % copies) * wwords; q.sbt = sb + (size_t)(i % copies) * swords; q.a = a; q.out = out; q.M = M; q.N = N; q.K = K; q.eps = 1e-6f;
            if (epi == EPI_RESIDUAL) { q.residual = res; q.norm_out = nw_dev; q.out_w = outw; q.ss_out = ssout; } else { q.ss = ss; q.ss_n = 160; q.a_frag = frag; }
            q.prof = prof;
            if (launch_qmm6_bf16(q, epi, 0) != 0) { printf("qmm6 launch failed\n"); exit(1); } };
        auto mm3 = [&](int i) {
            Qmm3Args q{}; q.wt = w + (size_t)(i % copies) * wwords; q.sbt = sb + (size_t)(i % copies) * swords; q.a = a; q.partial = partial; q.M = M; q.N = N; q.K = K; q.eps = 1e-6f;
            if (launch_qmm3_bf16(q, 0, PRO_NONE, -1) != 0) { printf("qmm3 launch failed\n"); exit(1); }
            if (launch_qmm3_reduce_bf16(partial, p3.slices, M, K, epi, res, out, nullptr, 0) != 0) { printf("reduce failed\n"); exit(1); } };
        const int iters = 24;
        float ms6, ms3;
        for (int i = 0; i < 3; ++i) mm6(i, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) mm6(i, nullptr); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms6, e0, e1));
        for (int i = 0; i < 3; ++i) mm3(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) mm3(i); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms3, e0, e1));
        printf("%-8s rows %2d  qmm6%s <MB %d GPW %2d sets %d> %3d x %d wg, %2d tiles each: %6.2f us   |   qmm3 + reduction (%d slices): %6.2f us\n", sh.name, M, frag ? " (fragment order)" : "", pl.MB, pl.GPW,
               pl.NSETS, pl.wgs, pl.row_blocks, pl.tiles_per_wg, ms6 * 1000.f / iters, p3.slices, ms3 * 1000.f / iters);
        {   // the row-streaming matmul (qmm7.h) on the same inputs: time, and bit equality with the register-resident kernel's output
            const Qmm7Plan p7 = qmm7_plan(M, N, K);
            if (frag && p7.ok && epi != EPI_RESIDUAL) {
                const size_t ob = (size_t)M * (epi == EPI_SWIGLU ? K / 2 : K) * 2;
                std::vector<uint16_t> o6(ob / 2), o7(ob / 2);
                mm6(0, nullptr); CK(hipDeviceSynchronize()); CK(hipMemcpy(o6.data(), out, ob, hipMemcpyDeviceToHost));
                CK(hipMemset(out, 0xff, ob));
                auto mm7 = [&](int i) {
                    Qmm6Args q{}; q.wt = w + (size_t)(i % copies) * wwords; q.sbt = sb + (size_t)(i % copies) * swords; q.a = a; q.out = out; q.M = M; q.N = N; q.K = K; q.eps = 1e-6f;
                    q.ss = ss; q.ss_n = 160; q.a_frag = 1;
                    if (launch_qmm7_bf16(q, epi, 0) != 0) { printf("qmm7 launch failed\n"); exit(1); } };
                mm7(0); CK(hipDeviceSynchronize()); CK(hipMemcpy(o7.data(), out, ob, hipMemcpyDeviceToHost));
                size_t diff = 0; for (size_t i = 0; i < o6.size(); ++i) diff += o6[i] != o7[i];
                float ms7;
                for (int i = 0; i < 3; ++i) mm7(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) mm7(i); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms7, e0, e1));
                printf("         qmm7 <MB %d T %d GPW %d> %3d wg: %6.2f us   (elements that differ from qmm6: %zu of %zu)\n", p7.MB, p7.T, p7.GPW, p7.wgs, ms7 * 1000.f / iters, diff, o6.size());
#ifdef QMM7_TRACE
                {
                    const size_t nw7 = (size_t)p7.wgs * QM7_WAVES;
                    unsigned long long *pb7 = nullptr; CK(hipMalloc(&pb7, nw7 * 16 * 8)); CK(hipMemset(pb7, 0, nw7 * 16 * 8));
                    Qmm6Args q{}; q.wt = w; q.sbt = sb; q.a = a; q.out = out; q.M = M; q.N = N; q.K = K; q.eps = 1e-6f; q.ss = ss; q.ss_n = 160; q.a_frag = 1; q.prof = pb7;
                    launch_qmm7_bf16(q, epi, 0); CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> hp(nw7 * 16); CK(hipMemcpy(hp.data(), pb7, nw7 * 16 * 8, hipMemcpyDeviceToHost));
                    double sum[16] = {0}; size_t cnt[16] = {0}; unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
                    for (size_t i = 0; i < nw7; ++i) if (hp[i * 16]) { tmin = std::min(tmin, hp[i * 16]); smax = std::max(smax, hp[i * 16]); for (int k = 1; k < 14; ++k) { if (hp[i * 16 + k]) { sum[k] += (double)(hp[i * 16 + k] - hp[i * 16]); cnt[k]++; tmax = std::max(tmax, hp[i * 16 + k]); } } }
                    printf("         qmm7 stamps: first wave start -> last stamp %.2f us, start spread %.2f us; mean us since the wave's start (requests out, groups 0.., met, stored):\n        ", (double)(tmax - tmin) * 1000.0 / rate, (double)(smax - tmin) * 1000.0 / rate);
                    for (int k = 1; k < 14; ++k) if (cnt[k]) printf(" %5.2f", sum[k] / cnt[k] * 1000.0 / rate);
                    printf("\n");
                    CK(hipFree(pb7));
                }
#endif
            }
        }
#ifdef QMM6_TRACE
        mm6(0, pb); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hp(nwaves * 16); CK(hipMemcpy(hp.data(), pb, nwaves * 16 * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t i = 0; i < nwaves; ++i) if (hp[i * 16]) { tmin = std::min(tmin, hp[i * 16]); for (int k = 0; k < 14; ++k) tmax = std::max(tmax, hp[i * 16 + k]); }
        double sum[16] = {0}; size_t cnt[16] = {0};
        for (size_t i = 0; i < nwaves; ++i) for (int k = 1; k < 14; ++k) if (hp[i * 16 + k]) { sum[k] += (double)(hp[i * 16 + k] - hp[i * 16]); cnt[k]++; }
        { double sc = 0, wc = 0; for (size_t i = 0; i < nwaves; ++i) { sc += (double)hp[i * 16 + 14]; wc += (double)hp[i * 16 + 15]; }
          printf("         shader clock over the kernel: %.0f MHz\n", wc > 0 ? sc / wc * rate / 1000.0 : 0.0); }
        printf("         first wave start -> last stamp %.2f us; start spread: ", (double)(tmax - tmin) * 1000.0 / rate);
        { unsigned long long smax = 0; for (size_t i = 0; i < nwaves; ++i) if (hp[i * 16]) smax = std::max(smax, hp[i * 16]); printf("%.2f us\n", (double)(smax - tmin) * 1000.0 / rate); }
        printf("         mean us since the wave's start at each stamp (1 requests out, 2 fragments in, 3 group sums, then per tile: mfma, barrier, epilogue):\n        ");
        for (int k = 1; k < 14; ++k) if (cnt[k]) printf(" %5.2f", sum[k] / cnt[k] * 1000.0 / rate);
        printf("\n");
#endif
        CK(hipFree(w)); CK(hipFree(sb)); CK(hipFree(a)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(outw)); CK(hipFree(nw_dev)); CK(hipFree(ss)); CK(hipFree(ssout));
        CK(hipFree(partial)); CK(hipFree(pb));
    }
    return 0;
}
