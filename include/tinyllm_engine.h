/*
 * tinyllm_engine.h — C ABI of the fused Qwen3 W4A16 forward ("decode engine").
 *
 * The reference runs one token through ~690 separately dispatched MLX primitives
 * (layer loop: src/tiny_llm_ref/qwen3_week2.py:357-392, qwen3_week3.py:320-338;
 * per-layer ops: qwen3_week3.py:55-121,141-147,181-187).  On an MI355X the whole
 * 382 us/token budget would be spent in launch gaps, so the same arithmetic is
 * issued here as 5 kernels per layer, captured once in a hipGraph and replayed:
 *
 *   1. RMSNorm -> fused QKV W4 GEMV                 (input_layernorm + wq|wk|wv)
 *   2. q/k-RMSNorm + RoPE + paged KV append + split-context GQA attention
 *      (+ a merge kernel only when the context is split)
 *   3. wo W4 GEMV + residual add
 *   4. RMSNorm -> gate|up W4 GEMV -> SwiGLU         (post_attention_layernorm + MLP in)
 *   5. w_down W4 GEMV + residual add
 *   final: RMSNorm -> lm_head W4 GEMV -> argmax -> next-token embedding gather.
 *
 * Every reference op boundary is kept as a bf16 rounding point inside the fused
 * kernels, so logits track the op-by-op path (tests/test_engine_gpu.py).
 *
 * State that changes every token (token ids, context lengths, step counter) lives
 * in device memory and is advanced by the last kernel of the step, so a captured
 * step replays without host involvement; the host only appends a page id to a
 * block-table row when a sequence crosses a page boundary.
 *
 * The multi-token path (chunked prefill, reference Request.try_prefill
 * src/tiny_llm_ref/batch.py:48-76) uses the MFMA W4 GEMM and the paged
 * FlashAttention kernel of tinyllm_hip.h with the same fused weight layout.
 *
 * KV storage is the reference's paged layout, one pool per layer:
 * key/value pages [P, Hkv, page_size, D] bf16 (paged_kv_cache.py:21-242), block
 * table [max_batch, max_pages_per_seq] int32 (-1 = unused), context_lens
 * [max_batch] int32 (kv_cache.py:210-224).  Pools are owned by the engine and
 * sized once at creation (288 GB of HBM: no growth path on the hot loop).
 *
 * All functions return 0 / negative tl_status; message via tl_last_error().
 * Pointers named *_dev are device pointers; everything else is host memory.
 */
#ifndef TINYLLM_ENGINE_H
#define TINYLLM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tl_engine tl_engine;

/* One W4 (group 128) linear: packed [rows, in/8] uint32, scales/biases [rows, in/128] bf16. */
typedef struct tl_w4 {
    const uint32_t *weight_dev;
    const void *scales_dev;
    const void *biases_dev;
    int rows; /* out features */
    int cols; /* in features  */
} tl_w4;

/* Per-layer weights in the engine's fused layout (built by the host mirror,
 * tiny_llm_hip/engine.py, from the mlx_lm-shaped checkpoint object):
 *   wqkv  rows = [q (Hq*D) | k (Hkv*D) | v (Hkv*D)]                      cols = hidden
 *   wo    rows = hidden                                                  cols = Hq*D
 *   wgu   rows interleaved: 2i = gate_proj row i, 2i+1 = up_proj row i   cols = hidden
 *   wdown rows = hidden                                                  cols = intermediate */
typedef struct tl_layer_weights {
    tl_w4 wqkv, wo, wgu, wdown;
    const void *input_norm_dev; /* [hidden] bf16 */
    const void *post_norm_dev;  /* [hidden] bf16 */
    const void *q_norm_dev;     /* [D] bf16 */
    const void *k_norm_dev;     /* [D] bf16 */
} tl_layer_weights;

typedef struct tl_engine_config {
    int hidden_size, num_layers, num_heads, num_kv_heads, head_dim, intermediate_size, vocab_size;
    float rope_theta, rms_norm_eps;
    int page_size;         /* tokens per KV page (reference default 128, qwen3_week3.py:222) */
    int num_pages;         /* physical pages per layer pool */
    int max_batch;         /* sequence slots */
    int max_pages_per_seq; /* block-table width */
    int max_prefill_rows;  /* largest prefill chunk (rows of the activation workspace).  From 1,792 rows the engine also keeps the layer
                              matrices as bf16 (rows x cols x 2 bytes each: 7.3 GB at Qwen3-4B) for the plain bf16 GEMM of large chunks */
} tl_engine_config;

/* Counters mirroring the pool statistics the reference serving bench prints
 * (paged_kv_cache.py:36-40, benches/bench.py:546-556). */
typedef struct tl_engine_stats {
    int pages_in_use, pages_free, peak_pages_in_use;
    long page_allocations, reused_page_allocations;
    long decode_steps, graph_captures, graph_replays, prefill_tokens;
    size_t kv_bytes, workspace_bytes;
    long graph_cache_flushes; /* times the cache of captured decode graphs (48 plans) was emptied */
    long aql_steps;           /* decode steps replayed as AQL packets on the engine's own HSA queue (TL_AQL=1; 0 on the hipGraph route) */
} tl_engine_stats;

/* A Qwen3-MoE layer (reference: src/tiny_llm_ref/qwen3_week3.py:209-214, 258-272 builds a Moe block for it; moe.py:39-89 is the
 * block: router softmax in fp32 -> top_k experts -> gathered gate / up / down W4 products -> probability-weighted sum).
 * router [E, hidden]; experts stacked on a leading axis: gate / up [E, I, hidden/8] words, down [E, hidden, I/8] words, scales /
 * biases [E, rows, cols/128] bf16.  W4, group 128.  Borrowed memory, like every weight. */
typedef struct tl_moe_weights {
    tl_w4 router;
    const uint32_t *gate_dev, *up_dev, *down_dev;
    const void *gate_scales_dev, *gate_biases_dev, *up_scales_dev, *up_biases_dev, *down_scales_dev, *down_biases_dev;
    int num_experts;        /* E <= 1024 */
    int experts_per_token;  /* top_k <= 16 */
    int intermediate_size;  /* I (moe_intermediate_size), a multiple of 128 */
    int norm_topk_prob;     /* renormalise the selected probabilities (Qwen3-MoE: true) */
} tl_moe_weights;

/* embed: the quantized embedding table [vocab, hidden]; lm_head: NULL for tied
 * embeddings (reference qwen3_week3.py:314-318).  Weight memory is borrowed and
 * must outlive the engine.  `stream` is the hipStream_t all work is issued on;
 * NULL makes the engine create and own a non-blocking stream (the legacy default
 * stream cannot be graph-captured) after a device-wide synchronise. */
int tl_engine_create(const tl_engine_config *cfg, const tl_layer_weights *layers, const tl_w4 *embed,
                     const void *final_norm_dev, const tl_w4 *lm_head, void *stream, tl_engine **out);
/* The same engine with its KV pages in another format ("quantized-KV"; NO reference interface is
 * replaced -- the reference lists quantised KV caches as not covered, README.md:134-135).  TL_KV_FP8_E4M3 (head_dim 128 only):
 * every K / V row is stored as 128 OCP FP8 E4M3 codes + one power-of-two float32 scale (include/tinyllm_hip.h, csrc/kv8.h,
 * oracle/kv_fp8.py): 132 bytes per row instead of 256.  Rows are quantised where they are written (prefill's qkv_post launch, the
 * decode attention's append) and every attention kernel reads the codes; the token being decoded is quantised before it is attended
 * to.  The model's arithmetic is otherwise unchanged: attention over the dequantised rows, which are exact bfloat16 values.
 * tl_engine_create == tl_engine_create_kv(..., TL_KV_BF16, ...). */
enum { TL_KV_BF16 = 0, TL_KV_FP8_E4M3 = 1 };
int tl_engine_create_kv(const tl_engine_config *cfg, const tl_layer_weights *layers, const tl_w4 *embed, const void *final_norm_dev,
                        const tl_w4 *lm_head, void *stream, int kv_format, tl_engine **out);
/* TL_KV_BF16 or TL_KV_FP8_E4M3 */
int tl_engine_kv_format(const tl_engine *e);
/* Make `layer` a mixture-of-experts layer (after tl_engine_create, before the first prefill / decode of the engine).  The layer's
 * dense wgu / wdown may be null in tl_engine_create's `layers` then; a layer with neither is an error at its first use.  The MoE
 * MLP runs as the reference's op sequence inside the captured step: RMSNorm, router GEMV, route kernel, gathered gate and up
 * GEMVs (one launch each, an expert per row), SiLU x up, gathered down GEMV, weighted sum + residual. */
int tl_engine_set_moe_layer(tl_engine *e, int layer, const tl_moe_weights *w);
/* Test / lab hook: switch a route that has an A/B twin, per engine, before its first prefill / decode (an error afterwards: captured
 * steps hold the routes they were captured with).  value 0 = off, anything else = on.  Options (defaults in brackets):
 *   "qmm3" [1] rows > 8 of a decode step on the K-sliced skinny matmul (0: the prefill GEMM's op sequence);  "qmm6" [1] the
 *   register-resident batched matmul (0: the K-sliced one everywhere);  "qmm7" [1] the row-streaming matmul for gate|up / qkv (0: qmm6);
 *   "gemm8" [1] prefill chunks of 1,792 rows and more through the plain bf16 GEMM over the bf16 weight copy (0: the W4 GEMM at every size);
 *   "attn_qkv_partials" [1] the decode attention adds the qkv slice planes itself;  "lmhead_tile_max" [1] per-tile maxima from the lm_head
 *   GEMV;  "gemm_fused_epilogue" [1] residual / SwiGLU inside the prefill GEMM;  "prefill_reduce_norm" [1] the split-K residual
 *   reduction of a small prefill chunk also writes the RMSNorm behind it;  "aql_fences" [0] HIP's agent-scope fences back on every
 *   packet of the AQL route.  Not part of the reference's surface: product code never calls it. */
int tl_engine_set_option(tl_engine *e, const char *name, int value);
void tl_engine_destroy(tl_engine *e);
/* Block the host until everything enqueued on the engine stream has finished. */
int tl_engine_synchronize(tl_engine *e);

/* Sequence slots ------------------------------------------------------------
 * begin: claim slot (must be free), context 0.  Fails with TL_ERR_INVALID when
 *        the slot is live.
 * reserve: make sure pages exist for `total_tokens` tokens of context (allocates
 *        from the free list, appends to the slot's block-table row); fails
 *        without side effects when the pool or the table width is exhausted
 *        (transactional like TinyKvPagedCache._append_chunk, paged_kv_cache.py:271-312).
 * release: return the slot's pages to the free list, context 0, row = -1.
 * rewind: drop the last n tokens of the slot (reference rewind(), paged_kv_cache.py:414-434). */
int tl_engine_begin(tl_engine *e, int slot);
int tl_engine_reserve(tl_engine *e, int slot, int total_tokens);
int tl_engine_release(tl_engine *e, int slot);
int tl_engine_rewind(tl_engine *e, int slot, int n);
int tl_engine_context_len(const tl_engine *e, int slot);  /* host mirror, <0 if slot free */
/* Adopt the sequence of slot `src` into the free slot `dst` (reference BatchingKvCache.add_request,
 * kv_cache.py:226-238): block-table row, context length and pending token change hands, no K/V bytes move. */
int tl_engine_move(tl_engine *e, int src, int dst);

/* Chunked prefill of ONE slot: runs `n` tokens (host int32 ids) at positions
 * [context, context+n) through the multi-token path, appends their K/V, and, when
 * want_logits != 0, computes the last row's logits + greedy token which becomes
 * the slot's pending input token for the next decode step.  n <= max_prefill_rows. */
int tl_engine_prefill(tl_engine *e, int slot, const int32_t *tokens, int n, int want_logits);

/* The same for up to 16 slots in ONE pass (continuous batching: several admitted prompts prefilled together -- reference
 * batch.py:48-76 prefills one request per turn; here the projections run once over the concatenated rows, at the row count
 * the MFMA GEMM is efficient at, while RoPE / KV append / paged attention stay per sequence).  tokens = the chunks back to
 * back (sum of lens <= max_prefill_rows), lens[i] tokens for slots[i] at positions [context_i, context_i + lens[i]);
 * want_logits[i] != 0: that chunk ends its prompt -> last-row logits + greedy token become the slot's pending token.
 * All-or-nothing: slot states, page counts and row totals are checked before anything is reserved. */
int tl_engine_prefill_packed(tl_engine *e, int n_seqs, const int *slots, const int32_t *tokens, const int *lens,
                             const int *want_logits);

/* Prefix sharing (reference KvPrefixGenerator fork/restore, agent/branching.py:42-208): make the
 * free slot `dst` a second sequence with the same tokens as the live slot `src`.  Full KV pages are shared (reference
 * counted, never rewritten), a partially filled tail page is copied; the pending input token is copied too.  Afterwards
 * both slots decode, rewind and release independently. */
int tl_engine_fork(tl_engine *e, int src, int dst);

/* Speculative verification (reference speculative_generate, generate.py:84-322: one target call over the pending token
 * plus the draft's proposals, logits_to_keep = all rows).  Appends n (1..8) tokens to the slot exactly like a prefill chunk
 * and returns in out_ids[i] the greedy token that follows tokens[0..i].  Nothing is recorded as generated; the caller
 * rewinds the rejected suffix with tl_engine_rewind and sets the next input with tl_engine_set_token.  Synchronises. */
int tl_engine_verify(tl_engine *e, int slot, const int32_t *tokens, int n, int32_t *out_ids);

/* Set the pending input token of a slot explicitly (e.g. sampled on the host). */
int tl_engine_set_token(tl_engine *e, int slot, int32_t token);

/* Run `steps` greedy decode steps over the live slots [0, batch): each step feeds
 * every slot's pending token at position context_len, appends K/V, and leaves the
 * argmax token as the next pending token.  Generated ids are appended to an
 * on-device ring [capacity, max_batch] readable via tl_engine_read_tokens.
 * The step is captured in a hipGraph on first use (re-captured when the
 * attention split bucket or batch changes); use_graph = 0 launches eagerly.
 * Pages are reserved on demand (TL_ERR_INVALID if the pool is exhausted).
 * Synchronisation depends on the replay route (tl_engine_replay_route): on the
 * default "aql" route the call waits for the stream before its first captured
 * step and RETURNS DRAINED -- every step of the call has run (the engine's HSA
 * queue is not the stream, so nothing stream-ordered may follow undrained steps;
 * the host waits ~50 us spinning, then blocked on the completion signal).  On the
 * "hipgraph" route (TL_AQL=0, or a plan the AQL route does not take) and with
 * use_graph = 0 the call only enqueues on the engine stream and does not
 * synchronise: a caller that overlaps host work with decode steps (a draft
 * model free-running beside its target) wants that route.
 * Must be called with the engine's own device current (the one current at
 * tl_engine_create): anything else is TL_ERR_INVALID. */
int tl_engine_decode(tl_engine *e, int batch, int steps, int use_graph);

/* Copy the ids produced by the last `count` decode steps for `slot` to host
 * (synchronises the stream). */
int tl_engine_read_tokens(tl_engine *e, int slot, int count, int32_t *out);

/* Pending token ids of slots [0, count) to host memory, after synchronising the stream. */
int tl_engine_read_pending(tl_engine *e, int count, int32_t *out);

/* Device pointer to the most recent logits, [rows, vocab] bf16 (decode: rows =
 * batch of the last step; prefill with want_logits: 1 row). */
const void *tl_engine_logits_dev(const tl_engine *e);
/* Stream-ordered device-to-device copy of the first `rows` logits rows into dst_dev ([rows, vocab] bf16). */
int tl_engine_copy_logits(tl_engine *e, void *dst_dev, int rows);
/* Device pointer to the pending token ids [max_batch] int32. */
const int32_t *tl_engine_tokens_dev(const tl_engine *e);

int tl_engine_get_stats(const tl_engine *e, tl_engine_stats *out);

/* How captured decode steps are replayed: "aql" -- hand-written AQL dispatch packets on the engine's own HSA queue, no cache
 * maintenance between the launches of a step (the default; csrc/aql.h) -- or "hipgraph: <why the AQL route is not available>"
 * (hipGraphLaunch; also TL_AQL=0).  The string lives until the calling thread's next call. */
const char *tl_engine_replay_route(const tl_engine *e);

/* Algorithmic HBM bytes of ONE decode step at the current state: all W4 weights
 * streamed once + K/V of every live context. */
size_t tl_engine_step_bytes(const tl_engine *e, int batch);

/* Measurement aid: runs ONE real decode step eagerly (same kernels, same state update as
 * tl_engine_decode(e, batch, 1, 0)) with every kernel stamping the device wall clock at its first
 * workgroup's start and its last wave's end.  kernel_us[k] / launches[k] are summed per kind:
 *   0 qkv GEMV, 1 wo GEMV, 2 gate|up GEMV, 3 w_down GEMV, 4 lm_head GEMV, 5 attention, 6 attention merge,
 *   7 step end (argmax + embed).
 * gemv_bytes[k] = algorithmic W4 bytes those launches stream (packed nibbles + bf16 scales and biases).
 * span_us = first start to last end of the step (includes the small reduce kernels this mode inserts
 * between launches, so it is NOT the production step time).  Synchronises the stream. */
typedef struct tl_step_profile {
    double kernel_us[8];
    int launches[8];
    double gemv_bytes[5];
    double span_us;
    int clock_khz;
    int n_splits;
} tl_step_profile;
int tl_engine_profile_step(tl_engine *e, int batch, tl_step_profile *out);

/* Test aid for the AQL replay route's invariant (csrc/aql.h: inside a replayed step no cache is written back or invalidated between
 * the launches, which is correct because every address one launch hands to a later launch of the step is WRITTEN ONCE PER STEP and
 * read only after it).  Runs ONE real decode step eagerly -- the kernels and the state update of tl_engine_decode(e, batch, 1, 0) --
 * with the per-layer hand-over buffers poisoned (every element a NaN) at the start and a checker behind every launch that compares
 * the hand-over regions (the shared activations of the arena; the per-layer buffers) with a shadow copy, element by 2-byte element:
 *   double_writes      elements a launch changed that an earlier launch of the step had already written (0 = the invariant holds for
 *                      this plan); first_launch / first_kind (tl_step_profile's kinds) / first_region (0 shared, 1 per-layer) /
 *                      first_offset (element index inside the region) name one offence of the earliest offending launch;
 *   written_once_plan  1 when the plan uses the per-layer buffers throughout -- the only plans the engine replays as AQL packets;
 *   a value read before the step wrote it is a NaN: the caller checks the logits (finite, and equal to an unchecked engine's).
 * A rewrite of an element with the value it already holds is not seen (and cannot be read stale).  Synchronises. */
typedef struct tl_step_check {
    int launches, written_once_plan, n_splits;
    long double_writes, elements_written;
    int first_launch, first_kind, first_region;
    long first_offset;
} tl_step_check;
int tl_engine_check_step(tl_engine *e, int batch, tl_step_check *out);

/* ===== kernel-level entry points of the decode path ===========================================
 * The launch code the engine runs per projection and per layer, on caller-owned device buffers: what the operator
 * microbenches time (reference benches/bench_week2_operators.py:355-358, bench_week3_attention.py:74-77) and what the
 * parity tests drive at the real Qwen3-4B shapes (tests/test_decode_kernels_gpu.py).  Not used by the engine itself. */

/* A W4 matrix re-packed into the decode layout ([rows/16][cols/128][64 lanes][4 words], csrc/qmv3.h).  `w` stays
 * borrowed (the packed-dot fallback reads the checkpoint layout); rows % 16 == 0, cols % 128 == 0.  Stream ordered. */
typedef struct tl_tiled_w4 tl_tiled_w4;
int tl_tiled_w4_create(const tl_w4 *w, void *stream, tl_tiled_w4 **out);
void tl_tiled_w4_destroy(tl_tiled_w4 *t);

/* Which kernel a projection ran: 1 = fused MFMA GEMV (qmv3: p = MR, KS, CW, LM, workgroups), 2 = skinny MFMA matmul +
 * slice reduction (qmm3: p = MB, TW, LM, slices, tile groups), 3 = packed-dot GEMV fallback, 4 = prefill GEMM path,
 * 5 = register-resident batched matmul (qmm6: p = MB, groups per wave, weight sets, row blocks, workgroups),
 * 6 = row-streaming batched matmul (qmm7: p = MB, groups per wave, tiles per workgroup, row blocks, workgroups). */
typedef struct tl_linear_info {
    int kernel;
    int launches;
    int rows_per_pass; /* activation rows per launch (the GEMV splits M when the staged rows exceed LDS) */
    int p[5]; /* GEMV: MR, KS, CW, LM, workgroups.  Skinny matmul: MB (16-row blocks), TW (tiles per wave; 0 = the persistent
                 grid, one workgroup per CU), LM (groups per slice), slices, workgroups */
} tl_linear_info;

/* out = epilogue(prologue(a) @ W^T) over M (1..64) bf16 rows, exactly as one projection of a decode step:
 *   prologue 0 none | 1 RMSNorm(a, norm_w, eps) rounded to bf16;  epilogue 0 store | 1 residual + bf16(acc) |
 *   2 SwiGLU over interleaved (gate_i, up_i) rows -> out [M, rows/2].
 *   kernel 0 = the routing of a single projection by M and matrix size, 1 = force the fused GEMV (M <= 8), 2 = force the skinny matmul
 *   (grid chosen by shape as the engine does), 3 / 4 = the skinny matmul on its one-shot / persistent grid, 5 = the register-resident
 *   matmul of a batched decode step (csrc/qmm6.h; prologue 0, or 3 through tl_decode_linear_ex), 6 = the row-streaming matmul
 *   (csrc/qmm7.h; prologue 3 with epilogue 0 or 2 through tl_decode_linear_ex -- bit-identical to kernel 5 on the same inputs).
 * The engine uses the pairs (1,0) qkv / lm_head, (0,1) wo / w_down, (1,2) gate|up, (0,0) -- and at 5..64 rows, through kernel 5,
 * (3,0) qkv / lm_head, (0,1) + ss_out + out_w for wo / w_down, (3,2) gate|up: rows travel weighted between the projections. */
size_t tl_decode_linear_workspace_bytes(int M, int rows, int cols);
int tl_decode_linear(const tl_tiled_w4 *w, const void *a_dev, void *out_dev, int M, int prologue, int epilogue,
                     const void *norm_w_dev, const void *residual_dev, float eps, int kernel, void *workspace_dev,
                     size_t workspace_bytes, void *stream, tl_linear_info *info);

/* The routes of the fused GEMV that only a whole engine step reached before round 4 -- the kernels BASELINE configs[1] times
 * (csrc/qmv3.h): the wo projection of ONE row forming its input row from the decode-attention split partials, the gate|up
 * projection over rows its producer left weighted, and the producer / consumer hand-over of RMSNorm sums of squares.  The
 * reference tests its matvec per shape against the dequantised product (tests_refsol/test_week_2_day_3.py:89-118); these
 * entry points let tests/test_decode_kernels_gpu.py do the same for exactly those instantiations.
 *   prologue 2 (with epilogue 1, M = 1, kernel 1): `a_dev` is not read; the activation row is the merge of
 *     merge_ws_dev [cols / 128 heads][n_splits][128 + 4] fp32 (128 value sums, running max in log2 units, running sum, 2 pad;
 *     n_splits 2 / 4 / 8): per column  bf16(sum_s v_s 2^(m_s - max m) / sum_s l_s 2^(m_s - max m)),  zero where the sum is zero.
 *   prologue 3 (with epilogue 2, M <= 8, kernel 1): a_dev holds bf16(x * norm_weight) (what a producer's out_w_dev holds) and
 *     ss_in_dev the sums of squares of x; out = SwiGLU(bf16(rsqrt(mean x^2 + eps) * (a @ W^T))).
 *   ss_in_dev [M][ss_in_n] (prologues 1 and 3): partial sums of squares of each row, added instead of re-derived (the GEMV: any
 *     multiple of 4 up to 256 partials, and so does the skinny matmul).
 *   epilogue 1 through the GEMV: ss_out_dev [M][rows / 16] receives the sum of squares of every 16 stored bf16 outputs;
 *     norm_out_dev [rows] + out_w_dev [M][rows]: also store bf16(out * norm_out).
 *   kernel 5 (M <= 64): prologue 3 with epilogue 0 or 2 (ss_in_dev required), prologue 0 with epilogue 0 or 1; epilogue 1 takes
 *     ss_out_dev / norm_out_dev + out_w_dev as above. */
typedef struct tl_linear_ex {
    const float *merge_ws_dev;
    int n_splits;
    const float *ss_in_dev;
    int ss_in_n;
    float *ss_out_dev;
    const void *norm_out_dev;
    void *out_w_dev;
    int fragment_order; /* kernels 2-5: the weighted rows on either side -- a_dev with prologue 3 (kernel 5), out_w_dev -- lie in the
                           batched step's FRAGMENT ORDER instead of row-major: [16-row block][128-column group][k-step t 0..3]
                           [lane = r + 16 c][8 elements] = row 16 block + r, columns 128 g + 32 c + 8 t .. + 7 (rows padded to 16: out_w_dev
                           holds ceil16(M) x rows elements).  The engine hands its rows over that way (every load of the consumer is one
                           contiguous 1 KiB). */
} tl_linear_ex;
int tl_decode_linear_ex(const tl_tiled_w4 *w, const void *a_dev, void *out_dev, int M, int prologue, int epilogue,
                        const void *norm_w_dev, const void *residual_dev, float eps, int kernel, void *workspace_dev,
                        size_t workspace_bytes, void *stream, const tl_linear_ex *ex, tl_linear_info *info);

/* The prefill projection of LARGE chunks (csrc/gemm8.h; the engine takes it from 1,792 rows): the reference's tile GEMM rounds the dequantised
 * weights to bf16 before the product (quantized_matmul.metal:96-249), so the weights are expanded ONCE --
 *   tl_prefill_weights_bf16: out_dev [rows, cols] bf16 = bf16(q * scale + bias) per element --
 * and the product is a plain bf16 GEMM with fp32 accumulation over the whole reduction (the unsplit tile kernel's arithmetic),
 *   tl_prefill_matmul_bf16: out [M, rows] = epilogue(a [M, cols] @ w_bf16^T); epilogue 0 store | 1 residual + bf16(acc) | 2 SwiGLU over
 *   interleaved (gate_i, up_i) weight rows -> out [M, rows / 2]; cols a multiple of 64, rows even.  Stream ordered. */
int tl_prefill_weights_bf16(const tl_w4 *w, void *out_dev, void *stream);
int tl_prefill_matmul_bf16(const void *a_dev, const void *w_bf16_dev, void *out_dev, int M, int rows, int cols, int epilogue,
                           const void *residual_dev, void *stream);

/* The attention launch of one decode layer: q/k-RMSNorm + RoPE at position context_lens[b] + append of the new K/V row to
 * the pages (IN PLACE) + GQA attention over context_lens[b] + 1 tokens (+ the merge launch when the context is split).
 *   qkv [batch, (Hq + 2 Hkv) D] bf16, pages [P, Hkv, page_size, D] bf16, block_table [batch, max_pages] int32,
 *   context_lens [batch] int32 (tokens already cached), out [batch, Hq D] bf16.
 * max_context: host upper bound of context_lens (sizes the context split exactly as tl_engine_decode does). */
/* Host-only (no device, no launch): the plans the decode path picks.  tl_decode_gemv_plan: MFMA GEMV of M rows against a
 * [rows, cols] W4 matrix -> out5 = {activation rows per workgroup, reduction split, waves, groups per wave, workgroups}; returns 1
 * when the MFMA GEMV takes the shape (0: the packed-dot GEMV would).  tl_decode_attention_plan: `batch` sequences whose longest
 * holds max_context tokens before this step -> out3 = {windows per sequence, tokens per window, query heads per workgroup}, at head size
 * 128 on 128-token pages. */
int tl_decode_gemv_plan(int M, int rows, int cols, int *out5);
/* 1 when the library holds a fused-GEMV kernel for (rows per workgroup, reduction split, waves, groups per wave): a plan is only
 * ever "taken" (tl_decode_gemv_plan returns 1) for such a combination; anything else decodes through the packed-dot GEMV. */
int tl_decode_gemv_variant_compiled(int MR, int KS, int CW, int LM);
/* The register-resident matmul of a batched decode step (csrc/qmm6.h): M rows against [rows, cols] -> out6 = {16-row blocks per
 * workgroup, quantisation groups per wave, weight sets, row blocks, workgroups per row block, tiles per workgroup}; returns 1 when
 * the kernel takes the shape.  tl_decode_batched_variant_compiled: 1 when the library holds a kernel for (row blocks, groups per wave). */
int tl_decode_batched_plan(int M, int rows, int cols, int *out6);
int tl_decode_batched_variant_compiled(int MB, int GPW);
/* The row-streaming matmul of a batched decode step (csrc/qmm7.h; gate|up and qkv where its plan exists): M rows against [rows, cols]
 * -> out4 = {16-row blocks, tiles per workgroup, quantisation groups per wave, workgroups}; returns 1 when the kernel takes the shape.
 * tl_decode_streaming_variant_compiled: 1 when the library holds the kernels for (tiles per workgroup, groups per wave) -- each pair
 * exists for 1 .. 4 row blocks and the store / SwiGLU epilogues. */
int tl_decode_streaming_plan(int M, int rows, int cols, int *out4);
int tl_decode_streaming_variant_compiled(int T, int GPW);
int tl_decode_attention_plan(int batch, int max_context, int num_heads, int num_kv_heads, int *out3);

typedef struct tl_attention_info {
    int n_splits, tokens_per_split, heads_per_workgroup;
    int launches;
} tl_attention_info;
size_t tl_decode_attention_fused_workspace_bytes(int batch, int num_heads, int head_dim);
int tl_decode_attention_fused(const void *qkv_dev, const void *q_norm_dev, const void *k_norm_dev, void *key_pages_dev,
                              void *value_pages_dev, const int32_t *block_table_dev, const int32_t *context_lens_dev,
                              void *out_dev, int batch, int num_heads, int num_kv_heads, int head_dim, int page_size,
                              int max_pages, float rope_theta, float eps, int max_context, void *workspace_dev,
                              size_t workspace_bytes, void *stream, tl_attention_info *info);
/* ... over FP8 (E4M3) pages: key_pages / value_pages [P, Hkv, page, 128] uint8 + key_scales / value_scales [P, Hkv, page] float32
 * (tl_engine_create_kv, include/tinyllm_hip.h "FP8 KV pages"); the appended row is quantised, head_dim 128 */
int tl_decode_attention_fused_fp8(const void *qkv_dev, const void *q_norm_dev, const void *k_norm_dev, void *key_pages_dev,
                                  float *key_scales_dev, void *value_pages_dev, float *value_scales_dev,
                                  const int32_t *block_table_dev, const int32_t *context_lens_dev, void *out_dev, int batch,
                                  int num_heads, int num_kv_heads, int head_dim, int page_size, int max_pages, float rope_theta,
                                  float eps, int max_context, void *workspace_dev, size_t workspace_bytes, void *stream,
                                  tl_attention_info *info);

#ifdef __cplusplus
}
#endif
#endif /* TINYLLM_ENGINE_H */
