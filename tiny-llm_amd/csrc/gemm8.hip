// Launcher of the plain bf16 prefill GEMM (gemm8.h) and the one-time W4 -> bf16 weight expansion it runs on.
#include "gemm8.h"

#include <algorithm>

namespace tl {

// one thread = one packed word = 8 consecutive weights of a row: bf16(q * s + beta), the tile GEMM's own weight rounding
__global__ __launch_bounds__(256) void dequant_w4_bf16_kernel(const uint32_t *__restrict__ weight, const uint16_t *__restrict__ scales,
                                                              const uint16_t *__restrict__ biases, uint16_t *__restrict__ out, size_t words, int words_per_row) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    const size_t row = i / words_per_row;
    const int j = (int)(i - row * words_per_row);
    const int G = words_per_row >> 4;  // groups of 128 = 16 words
    const float s = BF16::to_float(scales[row * G + (j >> 4)]), be = BF16::to_float(biases[row * G + (j >> 4)]);
    const uint32_t w = weight[i];
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = __builtin_fmaf((float)((w >> (8 * e)) & 0xfu), s, be);
        const float hi = __builtin_fmaf((float)((w >> (8 * e + 4)) & 0xfu), s, be);
        r[e] = BF16::pack2(lo, hi);
    }
    *reinterpret_cast<u32x4 *>(out + i * 8) = r;
}

int dequant_w4_to_bf16(const uint32_t *weight, const uint16_t *scales, const uint16_t *biases, uint16_t *out, int rows, int cols, hipStream_t st) {
    if (!weight || !scales || !biases || !out || rows <= 0 || cols <= 0 || cols % 128 != 0) return -1;
    const size_t words = (size_t)rows * (cols / 8);
    hipLaunchKernelGGL(dequant_w4_bf16_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, weight, scales, biases, out, words, cols / 8);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// whole 64-wide reduction steps; rows and columns are ragged-safe (buffer range checks, masked stores); the byte offsets of a lane stay
// inside 32 bits
bool gemm8_applicable(int M, int N, int K) {
    return M >= 1 && N >= 4 && K >= 2 * G8_BK && K % G8_BK == 0 && N % 4 == 0 && (size_t)(M + 256) * K * 2 < (1ull << 31) && (size_t)(N + 256) * K * 2 < (1ull << 31) &&
           (size_t)M * N * 2 < (1ull << 31);  // (a lane stores four consecutive columns; the output's byte offsets stay inside 31 bits too)
}

int qmm3_num_cus();  // qmm3.hip
// The tile whose grid costs the fewest (rounds over the CUs) x (outputs per tile); the narrower tiles feed their MFMAs with more LDS reads
// per flop, priced at 3 % / 8 %.  2,048 x 6,144: 256 x 192 = 256 tiles, one round, instead of 192 on 256 x 256; 4,096 x 2,560: 256 x 160 =
// 256 tiles instead of 160.
Gemm8Plan gemm8_plan(int M, int N, int K) {
    Gemm8Plan best{};
    if (!gemm8_applicable(M, N, K)) return best;
    const int cus = std::max(1, qmm3_num_cus());
#ifndef G8_LAB_CANDS
#define G8_LAB_CANDS 5
#endif
    const struct { int bm, bn; double eff; } cand[5] = {{256, 256, 1.0}, {256, 192, 0.97}, {256, 160, 0.92}, {128, 160, 0.80}, {128, 256, 0.86}};
    double best_cost = 0.0;
    for (int ci = 0; ci < G8_LAB_CANDS; ++ci) {
        const auto &c = cand[ci];
        const int tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const double cost = (double)((tiles + cus - 1) / cus) * c.bm * c.bn / c.eff;
        if (!best.ok || cost < best_cost) best = Gemm8Plan{c.bm, c.bn, tiles, true}, best_cost = cost;
    }
    return best;
}

template <typename KernelT>
static int launch8(KernelT kern, const Gemm8Args &a, const Gemm8Plan &pl, hipStream_t st) {
    const size_t lds = gemm8_lds_bytes(pl.BM, pl.BN);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // one workgroup per CU at most: a workgroup walks tiles b, b + grid, ... with its reduction steps running on across them (gemm8_body.inc)
    // (slots, not tiles: gemm8_body.inc deals blocks of 4 x 8 tiles to the XCDs; a ragged block's empty slots are skipped inside)
    const int slots = (((a.M + pl.BM - 1) / pl.BM + 3) / 4 * (((a.N + pl.BN - 1) / pl.BN + 7) / 8) + 7) / 8 * 256;
    hipLaunchKernelGGL(kern, dim3(std::min(slots, std::max(8, qmm3_num_cus() / 8 * 8))), dim3(G8_WAVES * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int EPI>
static int launch8_tile(const Gemm8Args &a, const Gemm8Plan &pl, hipStream_t st) {
    if (pl.BM == 256 && pl.BN == 256) return launch8(gemm8_kernel_256x256<EPI>, a, pl, st);
    if (pl.BM == 256 && pl.BN == 192) return launch8(gemm8_kernel_256x192<EPI>, a, pl, st);
    if (pl.BM == 256 && pl.BN == 160) return launch8(gemm8_kernel_256x160<EPI>, a, pl, st);
    if (pl.BM == 128 && pl.BN == 160) return launch8(gemm8_kernel_128x160<EPI>, a, pl, st);
    if (pl.BM == 128 && pl.BN == 256) return launch8(gemm8_kernel_128x256<EPI>, a, pl, st);
    return -1;
}

int launch_gemm8_bf16(const Gemm8Args &args, int epi, hipStream_t st) {
    if (!args.a || !args.w || !args.out) return -1;
    const Gemm8Plan pl = gemm8_plan(args.M, args.N, args.K);
    if (!pl.ok) return -1;
    if (epi == EPI_RESIDUAL && !args.residual) return -1;
    if (epi == EPI_STORE) return launch8_tile<EPI_STORE>(args, pl, st);
    if (epi == EPI_RESIDUAL) return launch8_tile<EPI_RESIDUAL>(args, pl, st);
    if (epi == EPI_SWIGLU) return launch8_tile<EPI_SWIGLU>(args, pl, st);
    return -1;
}

#ifdef G8_TRACE
int gemm8_trace_read(unsigned long long *dst, int n) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g8_trace), (size_t)n * 8) == hipSuccess ? 0 : -1; }
#endif

}  // namespace tl
