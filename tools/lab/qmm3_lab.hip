// Kernel laboratory (not part of the product): the skinny batched-decode matmul (csrc/qmm3.h) at 16/32/64 rows, HIP events.
// Runs the one-shot grid (mode 0) and the persistent grid (mode 1) on the same random inputs, compares their reduced bf16
// outputs, and times the matmul alone and matmul + slice reduction.  Built once per ablation (-DQMM3_ABL=n):
// tools/lab/run_qmm3_lab.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../tiny-llm_amd/csrc/qmm3.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

static uint32_t rng_state = 12345u;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static inline float f_of(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const int only_mode = argc > 2 ? atoi(argv[2]) : -1;
    struct Shape { const char *name; int K, N; } shapes[] = {{"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}, {"lm_head", 151936, 2560}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        const size_t wwords = (size_t)K * N / 8, swords = (size_t)K * G;
        const int copies = (size_t)wwords * 4 > (64u << 20) ? 3 : 8;
        uint32_t *w, *sb; uint16_t *a, *out[3], *nw_dev; float *partial, *ss;
        CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&sb, swords * 4 * copies)); CK(hipMalloc(&a, (size_t)M * N * 2));
        for (int i = 0; i < 3; ++i) CK(hipMalloc(&out[i], (size_t)M * K * 2));
        CK(hipMalloc(&nw_dev, (size_t)N * 2)); CK(hipMalloc(&ss, (size_t)M * QM3_SS * 4));
        {
            std::vector<uint32_t> hw(wwords), hs(swords); std::vector<uint16_t> ha((size_t)M * N);
            for (auto &x : hw) x = rnd();
            for (auto &x : hs) { const float sc = 0.01f + (rnd() >> 8) * (0.01f / 16777216.f), be = ((int)(rnd() >> 8) - 8388608) * (0.1f / 8388608.f); x = (uint32_t)bf16_of(sc) | ((uint32_t)bf16_of(be) << 16); }
            for (auto &x : ha) x = bf16_of(((int)(rnd() >> 8) - 8388608) * (1.0f / 8388608.f));
            for (int c = 0; c < copies; ++c) { CK(hipMemcpy(w + c * wwords, hw.data(), wwords * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sb + c * swords, hs.data(), swords * 4, hipMemcpyHostToDevice)); }
            CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
            std::vector<uint16_t> hn(N); for (auto &x : hn) x = bf16_of(0.5f + (rnd() >> 8) * (1.0f / 16777216.f));
            std::vector<float> hss((size_t)M * QM3_SS, 0.f);
            for (int m = 0; m < M; ++m) { double t = 0; for (int n = 0; n < N; ++n) { const float v = f_of(ha[(size_t)m * N + n]); t += (double)v * v; } hss[(size_t)m * QM3_SS] = (float)(t * 0.75); hss[(size_t)m * QM3_SS + 3] = (float)(t * 0.25); }
            CK(hipMemcpy(nw_dev, hn.data(), N * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ss, hss.data(), hss.size() * 4, hipMemcpyHostToDevice));
        }
        size_t pbytes = 0;
        for (int mode = 0; mode < 2; ++mode) pbytes = std::max(pbytes, qmm3_plan(M, N, K, mode).partial_bytes);
        CK(hipMalloc(&partial, pbytes));
        const int pro = argc > 3 ? atoi(argv[3]) : PRO_NONE;
        std::vector<uint16_t> ho[3];
        for (int mode = 0; mode < 2; ++mode) {
            if (only_mode >= 0 && mode != only_mode) continue;
            const Qmm3Plan pl = qmm3_plan(M, N, K, mode);
            const int iters = 24;
            auto mm = [&](int i) { Qmm3Args q{}; q.wt = w + (size_t)(i % copies) * wwords; q.sbt = sb + (size_t)(i % copies) * swords; q.a = a; q.partial = partial; q.M = M; q.N = N; q.K = K; q.norm_w = nw_dev; q.ss = ss; q.eps = 1e-6f;
                                   if (launch_qmm3_bf16(q, 0, pro, mode) != 0) { printf("launch failed\n"); exit(1); } };
            auto red = [&]() { if (launch_qmm3_reduce_bf16(partial, pl.slices, M, K, EPI_STORE, nullptr, out[mode], nullptr, 0) != 0) { printf("reduce failed\n"); exit(1); } };
            CK(hipMemset(partial, 0xff, pbytes));
            for (int i = 0; i < 3; ++i) { mm(i); red(); }
            CK(hipDeviceSynchronize());
            ho[mode].resize((size_t)M * K);
            CK(hipMemcpy(ho[mode].data(), out[mode], (size_t)M * K * 2, hipMemcpyDeviceToHost));
            float ms, ms2;
            CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) mm(i); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) { mm(i); red(); } CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms2, e0, e1));
#if QMM3_ABL & 64
            if (mode == 1) {  // phase stamps of one launch: mean time since the wave's start, over the waves that got that far
                const size_t nw = (size_t)pl.grid_x * QM3_WAVES;
                unsigned long long *pb; CK(hipMalloc(&pb, nw * 16 * 8)); CK(hipMemset(pb, 0, nw * 16 * 8));
                Qmm3Args q{}; q.wt = w; q.sbt = sb; q.a = a; q.partial = partial; q.M = M; q.N = N; q.K = K; q.prof = pb;
                q.norm_w = nw_dev; q.ss = ss; q.eps = 1e-6f;
                launch_qmm3_bf16(q, 0, pro, mode); CK(hipDeviceSynchronize());
                std::vector<unsigned long long> hp(nw * 16); CK(hipMemcpy(hp.data(), pb, nw * 16 * 8, hipMemcpyDeviceToHost));
                int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
                unsigned long long tmin = ~0ull, tmax = 0; for (size_t i = 0; i < nw; ++i) if (hp[i * 16]) { tmin = std::min(tmin, hp[i * 16]); for (int k = 0; k < 16; ++k) tmax = std::max(tmax, hp[i * 16 + k]); }
                for (int cls = 2; cls <= 16; ++cls) {  // waves grouped by how many stamps they took
                    size_t n = 0; double sum[16] = {0};
                    for (size_t i = 0; i < nw; ++i) { int cnt = 0; for (int k = 0; k < 16; ++k) cnt += hp[i * 16 + k] != 0; if (cnt != cls) continue; ++n; for (int k = 1; k < cls; ++k) sum[k] += (double)(hp[i * 16 + k] - hp[i * 16]); }
                    if (!n) continue;
                    printf("    %5zu waves with %2d stamps (us since the wave's start):", n, cls);
                    for (int k = 1; k < cls; ++k) printf(" %.2f", sum[k] / n * 1e3 / rate);
                    printf("\n");
                }
                double s0 = 0; for (size_t i = 0; i < nw; ++i) s0 += (double)(hp[i * 16] - tmin);
                printf("    mean start offset %.2f us, first start -> last stamp %.2f us\n", s0 / nw * 1e3 / rate, (double)(tmax - tmin) * 1e3 / rate);
                CK(hipFree(pb));
            }
#endif
            const double us = ms * 1e3 / iters, us2 = ms2 * 1e3 / iters;
            printf("  %-8s M=%d %s (MB%d TW%d LM%d, %d x %d workgroups, %d tiles/wg, lds %zu): %7.1f us  weights %6.1f GB/s  %6.1f TFLOP/s | + reduce %7.1f us\n",
                   sh.name, M, mode ? "persistent " : "one-shot   ", pl.MB, pl.TW, pl.LM, pl.grid_x, pl.slices, pl.tiles_per_wg, pl.lds, us,
                   wwords * 4 / us / 1e3, 2.0 * M * K * N / us / 1e6, us2);
        }
        for (int mode = 1; mode < 2; ++mode) {
            if (ho[0].empty() || ho[mode].empty()) continue;
            double maxd = 0, maxv = 0; size_t diff = 0, bad = 0;
            for (size_t i = 0; i < ho[0].size(); ++i) {
                const float x = f_of(ho[0][i]), y = f_of(ho[mode][i]);
                if (!(x == x) || !(y == y)) ++bad;
                if (ho[0][i] != ho[mode][i]) ++diff;
                maxd = std::max(maxd, (double)fabsf(x - y)); maxv = std::max(maxv, (double)fabsf(x));
            }
            printf("  %-8s M=%d one-shot vs %s: %zu of %zu bf16 outputs differ, max |diff| %.5f (max |value| %.3f), NaNs %zu\n", sh.name, M,
                   "persistent", diff, ho[0].size(), maxd, maxv, bad);
        }
        CK(hipFree(w)); CK(hipFree(sb)); CK(hipFree(a)); CK(hipFree(partial)); for (int i = 0; i < 3; ++i) CK(hipFree(out[i])); CK(hipFree(nw_dev)); CK(hipFree(ss));
    }
    return 0;
}
