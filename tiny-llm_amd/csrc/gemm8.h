// Plain bf16 GEMM for prefill chunks over a bf16 copy of the weights:  C[M, N] = A[M, K] . W[N, K]^T  (+ residual / SwiGLU), fp32 accumulate.
//
// Why (round 6).  The W4 prefill GEMM (qmm.hip) dequantises in its main loop: 128 x 128 tiles, a barrier per 64-wide step, VALU and matrix
// pipe alternating -- 750 TFLOP/s for a 2,048-row layer, 30 % of the dense bf16 peak, for three rounds.  The reference's tile GEMM rounds the
// dequantised weights to bf16 BEFORE the product (quantized_matmul.metal:96-249: `w = T(q * s + beta)` into threadgroup memory, then bf16
// simdgroup MMA with fp32 accumulation), so a bf16 copy of the weights, made once, IS that kernel's B operand: 288 GB of HBM hold the 8 GB of
// a Qwen3-4B without noticing, a 2,048-row chunk re-reads them at 6 % of its time, and the main loop is left with nothing but the two
// operand streams and the matrix pipe.
//
// Structure (guides/cdna_hip_programming.md, "canonical CDNA GEMM" and the glds notes):
//   * 256 x 256 output tile per workgroup, 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave = 8 x 4 accumulator tiles of
//     v_mfma_f32_16x16x32_bf16 (128 registers); ONE workgroup per CU;
//   * both operands are K-major ([rows][K]): a 64-wide K step of either tile is 256 rows x 128 bytes = 32 KB, brought in by LDS-DMA
//     (buffer_load ... lds, 16 bytes per lane, 8 rows per instruction, no registers): 8 instructions per lane and step; the LDS image is
//     lane-linear, the XOR swizzle that makes the fragment reads conflict-poor is applied to the SOURCE chunk (chunk p of row r holds
//     global chunk p ^ ((r >> 1) & 7)); rows past M / N read zeros through the buffer resource's range check;
//   * two LDS stages (128 KB), ONE raw s_barrier per 64-wide step (a __syncthreads() would drain the DMA), the fragments of the two 32-wide
//     k-steps double-buffered in registers: a stage is refilled (tile kt + 2) as soon as every wave has read its second half, while the
//     matrix pipe works on it;
//   * tiles are dealt so that the workgroups of one XCD share one 256-row band of A (its L2 holds the band; the weight panels stream).
#pragma once
#include "common.h"
#include "qmv.h"

namespace tl {

struct Gemm8Args {
    const uint16_t *a;         // [M, K] bf16
    const uint16_t *w;         // [N, K] bf16 (the dequantised weights, row = output feature)
    uint16_t *out;             // [M, N]  (EPI_SWIGLU: [M, N / 2], weight rows interleaved gate_i, up_i)
    const uint16_t *residual;  // EPI_RESIDUAL [M, N]
    int M, N, K;
};

constexpr int G8_BK = 64, G8_WAVES = 8;
#ifdef G8_TRACE  // tools/lab/gemm8_lab only: wall-clock stamps of wave 0 of workgroup G8_TRACE behind every step's barrier (and around a tile's epilogue)
__device__ unsigned long long g8_trace[2048];
#define G8_STAMP(code) do { if (blockIdx.x == (G8_TRACE) && tid == 0 && g8_n < 2040) { g8_trace[1 + g8_n++] = (wall_clock64() << 8) | (unsigned)(code); g8_trace[0] = g8_n; } } while (0)
#else
#define G8_STAMP(code) do { } while (0)
#endif
// The tile is a parameter of the kernel body (gemm8_body.inc): WM x WN waves, TM x TN accumulator tiles (16 x 16) per wave -> BM = 16 WM TM
// rows, BN = 16 WN TN columns.  256 x 256 = (2, 4, 8, 4);  256 x 192 = (2, 4, 8, 3) and 256 x 160 = (4, 2, 4, 5) exist because 2,048- and 4,096-row chunks against 2,560 /
//   6,144 output columns leave a quarter to a third of the 256 CUs idle on 256 x 256 tiles (gemm8_plan picks by rounds x tile size).
constexpr size_t gemm8_lds_bytes(int BM, int BN) { return (size_t)2 * (BM + BN) * G8_BK * 2; }

#define G8_NAME gemm8_kernel_256x256
#define G8_WM 2
#define G8_WN 4
#define G8_TM 8
#define G8_TN 4
#include "gemm8_body.inc"
#undef G8_NAME
#undef G8_WM
#undef G8_WN
#undef G8_TM
#undef G8_TN
#define G8_NAME gemm8_kernel_256x192
#define G8_WM 2
#define G8_WN 4
#define G8_TM 8
#define G8_TN 3
#include "gemm8_body.inc"
#undef G8_NAME
#undef G8_WM
#undef G8_WN
#undef G8_TM
#undef G8_TN
#define G8_NAME gemm8_kernel_256x160
#define G8_WM 4
#define G8_WN 2
#define G8_TM 4
#define G8_TN 5
#include "gemm8_body.inc"
#undef G8_NAME
#undef G8_WM
#undef G8_WN
#undef G8_TM
#undef G8_TN
#define G8_NAME gemm8_kernel_128x160
#define G8_WM 4
#define G8_WN 2
#define G8_TM 2
#define G8_TN 5
#include "gemm8_body.inc"
#undef G8_NAME
#undef G8_WM
#undef G8_WN
#undef G8_TM
#undef G8_TN
#define G8_NAME gemm8_kernel_128x256
#define G8_WM 2
#define G8_WN 4
#define G8_TM 4
#define G8_TN 4
#include "gemm8_body.inc"
#undef G8_NAME
#undef G8_WM
#undef G8_WN
#undef G8_TM
#undef G8_TN

// gemm8.hip
struct Gemm8Plan {
    int BM, BN, tiles;
    bool ok;
};
Gemm8Plan gemm8_plan(int M, int N, int K);
bool gemm8_applicable(int M, int N, int K);
int launch_gemm8_bf16(const Gemm8Args &args, int epi, hipStream_t st);  // -1: not applicable, nothing launched
// W4 (checkpoint layout) -> bf16 [rows, cols]: bf16(q * s + beta) per element, the reference tile GEMM's own rounding (quantized_matmul.metal:142-160)
#ifdef G8_TRACE
int gemm8_trace_read(unsigned long long *dst, int n);
#endif
int dequant_w4_to_bf16(const uint32_t *weight, const uint16_t *scales, const uint16_t *biases, uint16_t *out, int rows, int cols, hipStream_t st);

}  // namespace tl
