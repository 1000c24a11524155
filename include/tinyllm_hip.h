/*
 * tinyllm_hip.h — C ABI of the MI355X (gfx950) replacement for the reference's
 * native extension `tiny_llm_ext_ref` (reference: src/extensions_ref/bindings.cpp:11-65,
 * C++ declarations src/extensions_ref/src/tiny_llm_ext.h:10-141).
 *
 * Every entry point takes raw DEVICE pointers, plain integer sizes, a dtype tag
 * and a hipStream_t (passed as void*).  No torch / MLX types cross this
 * boundary.  All functions return 0 on success and a negative tl_status on
 * failure; tl_last_error() returns a thread-local, human-readable message (the
 * Python binding turns it into RuntimeError exactly where the reference's C++
 * throws std::runtime_error).
 *
 * Kernels are launched asynchronously on `stream`; nothing here synchronises.
 * Inputs are borrowed, outputs are caller-allocated — except
 * tl_paged_cache_update, which writes in place (the reference aliases the
 * output to the `pages` buffer, paged_attention.cpp:46-49).
 *
 * Dimension naming follows the reference's quantized matmul:
 *   a:[M,N] activations, b:[K,N/8] packed uint32 weights, out:[M,K]
 *   (N = in-features / reduction, K = out-features; quantized_matmul.cpp:125-127).
 */
#ifndef TINYLLM_HIP_H
#define TINYLLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tl_dtype {
    TL_F32 = 0,
    TL_F16 = 1,
    TL_BF16 = 2,
} tl_dtype;

typedef enum tl_status {
    TL_OK = 0,
    TL_ERR_INVALID = -1,     /* argument validation failed (reference: std::runtime_error) */
    TL_ERR_UNSUPPORTED = -2, /* valid request, no kernel for it */
    TL_ERR_HIP = -3,         /* HIP runtime error at launch */
} tl_status;

/* Thread-local message for the last failing call on this thread. */
const char *tl_last_error(void);

/* ABI version; bumped when a signature changes. */
int tl_abi_version(void);

/* Replaces load_library(path) (src/extensions_ref/src/utils.cpp:9-14). The HIP
 * code objects are embedded in this shared library, so this only checks that a
 * gfx950 device is visible.  Returns TL_OK or TL_ERR_HIP. */
int tl_load_library(const char *path);

/* ---- W4A16 (group 128) matmul: replaces quantized_matmul --------------------
 * reference: quantized_matmul.cpp:14-80 (validation), :111-240 (dispatch),
 * kernels quantized_matmul.metal:8-56 (vanilla), :96-293 (tile GEMM, split-K),
 * :441-538 (decode matvec).
 *   scales,biases : [K, N/128] dtype (f16|bf16)
 *   a             : [M, N] dtype, row-contiguous
 *   b             : [K, N/8] uint32, nibble i of word j = element 8j+i
 *   out           : [M, K] dtype
 * Dispatch mirrors the reference: use_simdgroup && M<=8 -> GEMV;
 * use_simdgroup && use_split_k && policy>1 -> split-K MFMA GEMM (partials in
 * dtype, workspace required); use_simdgroup -> MFMA GEMM; else one-thread-per-
 * output kernel.  `workspace` may be NULL when
 * tl_quantized_matmul_workspace_bytes(...) == 0. */
int tl_quantized_matmul(const void *scales, const void *biases, const void *a, const uint32_t *b, void *out, int M,
                        int N, int K, int group_size, int bits, tl_dtype dtype, int use_simdgroup, int use_split_k,
                        void *workspace, size_t workspace_bytes, void *stream);
size_t tl_quantized_matmul_workspace_bytes(int M, int N, int K, tl_dtype dtype, int use_simdgroup, int use_split_k);
/* split-K factor the dispatch would use (1 = no split); exposed for tests
 * (reference policy quantized_matmul.cpp:138-151, re-derived for 256 CUs). */
int tl_quantized_matmul_split_k(int M, int N, int K, int use_simdgroup, int use_split_k);

/* ---- replaces quantized_embedding (quantized_matmul.cpp:82-101,:242-273) ----
 * indices [tokens] int32|uint32, weight [V, dim/8] uint32, scales/biases
 * [V, dim/128] dtype, out [tokens, dim] dtype. */
int tl_quantized_embedding(const void *indices, int indices_unsigned, const void *scales, const void *biases,
                           const uint32_t *weight, void *out, int tokens, int dim, int vocab, int group_size, int bits,
                           tl_dtype dtype, void *stream);

/* ---- replaces rms_norm (week2_kernels.cpp:36-42,104-125) -------------------
 * out = T(x * rsqrt(mean(x^2)+eps) * w), fp32 inside, one rounding. */
int tl_rms_norm(const void *x, const void *weight, void *out, int rows, int dim, float eps, tl_dtype dtype,
                void *stream);

/* ---- replaces rope (week2_kernels.cpp:44-55,127-156) ------------------------
 * x,out [B,L,H,D]; offsets [B] int32; rotates the first `dims` of D. */
int tl_rope(const void *x, const int32_t *offsets, void *out, int B, int L, int H, int D, int dims, float base,
            int traditional, tl_dtype dtype, void *stream);

/* ---- replaces swiglu (week2_kernels.cpp:57-63,158-174) ---------------------- */
int tl_swiglu(const void *gate, const void *up, void *out, size_t size, tl_dtype dtype, void *stream);

/* ---- replaces decode_attention (week2_kernels.cpp:65-84,176-211) -----------
 * q,out [q_rows=B*Hq, L, D]; k,v [B*Hkv, S, D]; mask fp32 [q_rows, L, S] when
 * has_mask; causal rule: key position > S - L + query position is skipped. */
int tl_decode_attention(const void *q, const void *k, const void *v, const float *mask, void *out, int q_rows, int L,
                        int S, int D, int num_heads, int num_kv_heads, float scale, int is_causal, int has_mask,
                        tl_dtype dtype, void *stream);

/* ---- replaces mx.gather_qmm as the reference uses it (grouped_expert_linear, src/tiny_llm_ref/moe.py:7-36;
 * SURVEY.md §8f row 3): out[m,:] = a[m,:] @ dequant(b[expert_ids[m]])^T, one expert per activation row.
 * b [E,K,N/8] u32, scales,biases [E,K,N/128], a [M,N], expert_ids [M] int32 ON THE DEVICE (ids outside [0,E) are
 * clamped), out [M,K].  group_size 128, bits 4, f16/bf16 like tl_quantized_matmul. */
int tl_gather_quantized_matvec(const void *scales, const void *biases, const void *a, const uint32_t *b,
                               const int32_t *expert_ids, void *out, int M, int N, int K, int num_experts,
                               int group_size, int bits, tl_dtype dtype, void *stream);

/* ---- replaces paged_cache_update (paged_attention.cpp:14-70) ---------------
 * pages [P,H,page_size,D] (written IN PLACE), values [1,H,length,D]. */
int tl_paged_cache_update(void *pages, const void *values, int num_pages, int heads, int page_size, int head_dim,
                          int length, int page_id, int start, tl_dtype dtype, void *stream);

/* ---- replaces paged_attention (paged_attention.cpp:77-225) ------------------
 * q,out [N=B*Hq, L, D]; key_pages,value_pages [P,Hkv,page_size,D];
 * block_table [B,max_pages] int32 (-1 = unused), context_lens [B] int32.
 * L<=8 -> split-context decode kernel (+ merge); L>8 bf16 D==128 -> MFMA
 * FlashAttention (the context is also split, + merge, when few query rows
 * meet a long context); L>8 f32 -> scalar tile kernel.
 * max_context_hint: upper bound of context_lens known to the host (<=0: use
 * max_pages*page_size); it only sizes the context split, never correctness.
 * workspace: tl_paged_attention_workspace_bytes(...) bytes, may be NULL if 0. */
int tl_paged_attention(const void *q, const void *key_pages, const void *value_pages, const int32_t *block_table,
                       const int32_t *context_lens, void *out, int N, int L, int D, int num_pages, int page_size,
                       int max_pages, int num_heads, int num_kv_heads, float scale, int is_causal,
                       int max_context_hint, tl_dtype dtype, void *workspace, size_t workspace_bytes, void *stream);
size_t tl_paged_attention_workspace_bytes(int N, int L, int D, int page_size, int max_pages, int num_heads,
                                          int num_kv_heads, int max_context_hint);
/* test / lab hook: waves per workgroup of the bf16 FlashAttention prefill kernel -- 8 (default: one workgroup per CU, the K/V tile
 * double-buffered and shared by 8 (head, query block) items; pages of 64+ tokens, chunks of 64+ rows) or 4 (its twin, bit-identical
 * results).  Returns the previous value; any other argument only reads it. */
int tl_paged_attention_waves(int waves);

/* ===== FP8 (E4M3) KV pages -- SURVEY.md section 8f row 4, "quantized-KV" ========
 * NO reference interface is replaced: the reference lists quantised KV caches as
 * not covered (README.md:134-135).  These are the quantised twins of the two
 * paged entry points above (paged_attention.cpp:14-70, :77-225), for the page
 * format oracle/kv_fp8.py states and csrc/kv8.h implements:
 *   pages  [P,H,page_size,128] uint8  -- OCP FP8 E4M3 codes, the bf16 layout at one byte per element
 *   scales [P,H,page_size]     float  -- one power of two per (page, head, slot) row: the smallest with amax / s <= 448
 * A dequantised value code * s is exactly a bfloat16 value; attention over FP8
 * pages is tl_paged_attention's arithmetic over the dequantised rows.  bfloat16
 * values / queries / outputs, head dimension 128 only. */
/* values [rows,128] bf16 -> codes [rows,128] + scales [rows]; and back (out [rows,128] bf16, exact) */
int tl_kv_fp8_quantize_rows(const void *values, void *codes, float *scales, long rows, int head_dim, void *stream);
int tl_kv_fp8_dequantize_rows(const void *codes, const float *scales, void *out, long rows, int head_dim, void *stream);
/* pages / page_scales written IN PLACE at (page_id, start .. start + length); values [1,H,length,128] bf16 */
int tl_paged_cache_update_fp8(void *pages, float *page_scales, const void *values, int num_pages, int heads, int page_size,
                              int head_dim, int length, int page_id, int start, void *stream);
/* as tl_paged_attention (same routing by L, same workspace rule); q, out bf16 */
int tl_paged_attention_fp8(const void *q, const void *key_pages, const float *key_scales, const void *value_pages,
                           const float *value_scales, const int32_t *block_table, const int32_t *context_lens, void *out, int N,
                           int L, int D, int num_pages, int page_size, int max_pages, int num_heads, int num_kv_heads, float scale,
                           int is_causal, int max_context_hint, void *workspace, size_t workspace_bytes, void *stream);

/* ===== fused decode fast path (not visible through the reference API) ========
 * One Qwen3 decode step = the layer loop of Qwen3ModelWeek2/3.__call__
 * (qwen3_week2.py:357-392, qwen3_week3.py:320-338) for L=1, with the
 * reference's op boundaries kept as bf16 rounding points inside fused kernels.
 * See include/tinyllm_engine.h. */

#ifdef __cplusplus
}
#endif
#endif /* TINYLLM_HIP_H */
