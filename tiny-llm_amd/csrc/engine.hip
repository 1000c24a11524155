// Decode engine: host side of include/tinyllm_engine.h.
//   * page allocator + slot table (host mirrors of block_table / context_lens), transactional reserve
//     (reference semantics: TinyKvPagedPool / TinyKvPagedCache, src/tiny_llm_ref/paged_kv_cache.py:21-443)
//   * one fused decode step = 5 launches per layer (+ merge when the context is split) + 2 at the end,
//     captured into a hipGraph per (batch, n_splits) and replayed
//   * multi-token prefill on the MFMA W4 GEMM + paged FlashAttention operators of tinyllm_hip.h
#include <algorithm>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/tinyllm_engine.h"
#include "common.h"
#include "engine_kernels.h"
#include "qmv.h"
#include "qmv3.h"
#include "qmm3.h"
#include "qmm6.h"
#include "qmm7.h"
#include "gemm8.h"
#include "attn_mfma.h"
#include "aql.h"

#include <dlfcn.h>
#include <memory>

namespace tl {

#define TL_TRY(expr)                  \
    do {                              \
        const int rc__ = (expr);      \
        if (rc__ != TL_OK) return rc__; \
    } while (0)

#define TL_HIP(expr)                                                                           \
    do {                                                                                       \
        const hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) return fail(TL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace tl

using namespace tl;

struct tl_engine {
    tl_engine_config cfg{};
    std::vector<tl_layer_weights> layers;
    tl_w4 embed{}, lm_head{};
    const void *final_norm = nullptr;
    hipStream_t stream = nullptr;
    bool owns_stream = false;

    // device memory (one arena for state + activations, one for KV)
    char *arena = nullptr;
    size_t arena_bytes = 0;
    uint16_t *kpool = nullptr, *vpool = nullptr;  // [layers][P, Hkv, page, D]
    size_t layer_pool_elems = 0, kv_bytes = 0;
    // FP8 pages (tl_engine_create_kv, kv8.h): the pools hold one byte per element and the rows' scales live beside them
    int kv_format = TL_KV_BF16;
    float *kscale_pool = nullptr, *vscale_pool = nullptr;  // [layers][P, Hkv, page]
    size_t layer_scale_elems = 0;
    size_t kv_elem_bytes() const { return kv_format == TL_KV_FP8_E4M3 ? 1 : 2; }
    float *layer_ks(int l) const { return kscale_pool ? kscale_pool + (size_t)l * layer_scale_elems : nullptr; }
    float *layer_vs(int l) const { return vscale_pool ? vscale_pool + (size_t)l * layer_scale_elems : nullptr; }
    void *splitk_ws = nullptr;
    size_t splitk_ws_bytes = 0;
    // decode-path copy of every W4 matrix in the tiled MFMA layout (qmv3.h); keyed by the checkpoint pointer
    struct Tiled {
        uint32_t *wt = nullptr, *sbt = nullptr;
    };
    std::map<const uint32_t *, Tiled> tiled;
    size_t tiled_bytes = 0;
    // Prefill chunks of GEMM8_MIN_ROWS (1,536) rows and more (an engine created with max_prefill_rows that large): the layer matrices once more as
    // bf16 -- bf16(q * s + beta), the B operand the reference's tile GEMM forms in threadgroup memory (quantized_matmul.metal:96-249) -- for
    // the plain bf16 GEMM of gemm8.h (256 x 256 tiles by LDS-DMA, no dequantisation in the loop).  7.3 GB at Qwen3-4B, of 288.
    std::map<const uint32_t *, uint16_t *> bf16w;
    size_t bf16w_bytes = 0;
    bool use_gemm8 = true;  // tl_engine_set_option "gemm8" = 0: every chunk through the W4 GEMM (qmm.hip), the twin
    bool fuse_reduce_norm = true;  // "prefill_reduce_norm" = 0: the split-K residual reduction and the RMSNorm behind it as two launches, the twin

    int32_t *block_table = nullptr, *context_lens = nullptr, *tokens = nullptr, *live = nullptr, *produced = nullptr,
            *ring = nullptr, *scratch_ctx = nullptr, *prefill_tokens = nullptr;
    uint16_t *x = nullptr, *h = nullptr, *xn = nullptr, *qkv = nullptr, *q_t = nullptr, *attn_t = nullptr,
             *attn = nullptr, *gu = nullptr, *act = nullptr, *tmp = nullptr, *logits = nullptr;
    float *attn_ws = nullptr;
    int last_attn_launches = 0;
    // lm_head GEMV of a 1-4-row decode step: per 16-logit tile (max, lowest index) pairs for step_end_kernel (qmv3.h tile_max);
    // tl_engine_set_option "lmhead_tile_max" = 0: step_end reads the logits row again
    f32x2 *lm_tile_max = nullptr;            // [8][vocab / 16]
    bool lm_tile_max_on = true, want_tile_max = false;
    int tile_max_rows = 0;                   // rows of the last lm_head launch that left pairs (0 = none)
    float *ss_x = nullptr, *ss_h = nullptr;  // [max_batch][QM3_SS] partial sums of squares of the rows of x / h (qmm3.h)
    // At 5 .. 64 decode rows the qkv projection's slice reduction is not launched; the decode-attention kernel adds the fp32 slice
    // partials itself (engine_kernels.h, QP).  Measured in round 3 (profiles/r03_labs/batched_decode_status.jsonl): 5 / 8 / 16 / 64
    // sequences 1.87 -> 1.81, 1.91 -> 1.89, 2.08 -> 2.03, 3.63 -> 3.52 ms per step.  tl_engine_set_option "attn_qkv_partials" = 0 launches the reduction.
    bool attn_qkv_partials = true;
    // Single-row decode with the context split 2 / 4 / 8 ways: no merge launch behind the attention kernel -- the wo GEMV forms the
    // merged row from the split partials while it stages it (qmv3.h, PRO_ATTN_MERGE).  Measured in round 3
    // (profiles/r03_labs/wo_merges_attn_ab_after_dpp.jsonl): 4 windows 1.023 -> 0.993 ms per token, 8 windows 1.033 -> 1.009 (the
    // merging GEMV costs +0.9 / +2.0 us per layer, the merge launch cost 0.6 us + a boundary); 64-token windows stay the best
    // (128-token windows: 1.022).  Other split counts, more sequences or another head size keep the merge launch (wo_merge_applicable).
    bool wo_merges_attn = true;
    // The 1-4-row GEMVs add the partial sums of squares their producer left instead of re-deriving them (a row without partials --
    // the first step after a MoE layer, a packed-dot fallback -- is still re-derived inside the kernel: qmv3.h, ss_given):
    // qkv -0.44 us, gate|up -1.1 us per layer (abl_lab, bit 8)
    bool gemv_producer_ss = true;
    // The wo GEMV of 1-4 decode rows also leaves h * post_attention_layernorm (bf16) and the gate|up GEMV stages THAT row and
    // multiplies its sums by the row's 1 / rms at the end (qmv3.h, PRO_RMS_WEIGHTED): the 1,216 workgroups of gate|up no longer
    // fetch the norm weights and normalise the whole row each.  tools/lab/trace_lab, back to back: gate|up 7.82 -> 7.08 us, wo
    // 3.91 -> 4.06; the qkv and lm_head GEMVs gain nothing from it and keep the fused RMSNorm (weighted_rows_apply decides by shape).
    bool gemv_weighted_rows = true;
    bool fuse_norm = true;                   // the skinny matmul normalises its own slice whenever its producer left sums of squares
    int32_t *verify_ids = nullptr;  // greedy ids of the rows of the last tl_engine_verify
    int qmm3_min_rows = 5;  // rows from which a projection uses the K-sliced skinny matmul instead of the GEMV (TL_QMM3_MIN_M)
    bool use_qmm3 = true;   // tl_engine_set_option "qmm3" = 0: rows > 8 go through the prefill GEMM path instead
    // 5 .. 64 rows: the register-resident matmul (qmm6.h) takes every projection whose plan fits; rows travel WEIGHTED between the
    // projections (qkv <- w_down / the embedding, gate|up <- wo).  tl_engine_set_option "qmm6" = 0: the K-sliced skinny matmul as before.
    bool use_qmm6 = true;
    // ... and, where its plan exists (round 6: gate|up and qkv of a 2,560-wide model), the row-streaming matmul (qmm7.h) instead of the
    // register-resident one: the rows' arrival overlaps the walk, a step costs by its 16-row blocks (3 included).  force_qmm7: the
    // kernel-level entry point asked for it by name (an error where it does not apply).
    bool use_qmm7 = true, force_qmm7 = false;
    int attn_rq = 0;             // query heads per decode-attention workgroup; 0 = by context (TL_ATTN_RQ at create: 1 or 4)
    int attn_rq1_ctx = 4096;     // contexts up to this many tokens use one query head per workgroup
    int attn_rq1_batch = 2;      // ... and up to this many sequences; at 4 the re-read windows cost 261 vs 180 us
    int attn_min_tokens = 64;    // tokens per attention workgroup before the context is split (TL_ATTN_MIN_TOKENS)
    int attn_wg_cap = 0;         // most attention workgroups per launch; 0 = by sequences (pick_decode_splits)
    bool attn_min_tokens_auto = true;  // ... or more, by context and sequences (pick_decode_splits)
    bool attn_mfma = true;       // TL_ATTN_MFMA=0: the GQA-group walk on the VALU (attn_decode_fused_kernel), the A/B twin of attn_mfma.h
    int attn_max_splits = 64;    // most context splits per sequence (TL_ATTN_MAX_SPLITS, a power of two <= 256)
    int attn_max_splits_gqa = 32;  // ... when a workgroup takes a whole GQA group (TL_ATTN_MAX_SPLITS sets both)
    tl_linear_info *linfo = nullptr;    // kernel-level entry points: which kernel a projection ran
    int force_linear = 0;               // kernel-level entry points: 1 = fused GEMV, 2 = skinny matmul
    int qmm3_mode = -1;                 // skinny matmul grid: -1 by shape (qmm3_plan), 0 one-shot, 1 persistent
    bool gemm_fused_epilogue = true;    // tl_engine_set_option "gemm_fused_epilogue" = 0: residual / SwiGLU of the prefill GEMM as separate launches
    size_t attn_ws_bytes = 0;
    int rows_cap = 0;
    int ring_cap = 4096;

    // host mirrors
    std::vector<std::vector<int>> slot_pages;
    std::vector<int> slot_ctx;
    std::vector<char> slot_live;
    std::vector<int> slot_produced;
    std::vector<int> free_pages;
    std::vector<char> page_was_used;
    std::vector<int> page_refs;  // sequences holding each page (prefix sharing after tl_engine_fork); 0 = free
    tl_engine_stats stats{};

    bool warmed = false;
    std::map<std::pair<int, long>, hipGraphExec_t> graphs;  // (batch, n_splits << 32 | tokens_per_split)
    // AQL replay (aql.h; TL_AQL=1 at create): a captured step also becomes a program of hand-written dispatch packets on the engine's own
    // HSA queue; a plan without a program (a kernel outside the device-only code objects, a node that is not a kernel) stays on hipGraphLaunch
    bool aql_on = false;
    int device = 0;                  // the HIP device the engine was created on: its buffers, its stream, its AQL runtime
    AqlRuntime *aql_rt = nullptr;    // the runtime of that device (aql.h: one per device, process-wide)
    std::unique_ptr<AqlQueue> aql_queue;
    std::map<std::pair<int, long>, std::unique_ptr<AqlProgram>> aql_programs;
    AqlFences aql_fences;
    std::string aql_why;  // why the last plan got no program
    // Per-layer decode activations (common.h, "activations between the launches of a decode step"): with the AQL route every value a launch
    // hands to a later launch of the SAME step lives at an address written once per step -- layer l's x / h / weighted h / qkv / attention
    // rows and partials / SwiGLU rows / sums of squares have their own buffers (1-4 rows: the fused-GEMV route; 36 x ~0.3 MB at Qwen3-4B)
    struct LayerAct {
        uint16_t *x_out, *h, *xn, *qkv, *attn, *act;
        float *ss_x_out, *ss_h, *attn_ws;
        // 5 .. 64 rows (round 5): x_out weighted for the NEXT RMSNorm, and the fp32 slice planes of the layer's K-sliced matmuls
        // (wo at 17-32 rows, w_down at every row count): written once per step like everything else here
        uint16_t *xw;
        float *planes[2];
    };
    std::vector<LayerAct> layer_act;
    char *layer_act_mem = nullptr;
    int layer_act_rows = 0;       // rows the per-layer buffers hold (0: none)
    size_t layer_ws_bytes = 0;    // attention partials per layer
    size_t layer_plane_bytes[2] = {0, 0};  // slice planes per layer: [0] wo, [1] w_down
    // the K-sliced matmul writes its fp32 planes here instead of splitk_ws while a per-layer batched step is enqueued (engine_linear)
    float *planes_now = nullptr;
    size_t planes_now_bytes = 0;
    bool step_written_once = false;  // the last enqueued step used the per-layer buffers throughout (enqueue_step)
    // tl_engine_check_step (test-only): the hand-over regions -- [0] the shared activations of the arena, [1] the per-layer buffers -- with
    // a shadow copy and one "written this step" byte per 2-byte element each, alive only inside the call
    struct WrittenOnceCheck {
        bool on = false;
        char *region[2] = {nullptr, nullptr};
        size_t bytes[2] = {0, 0};
        uint32_t *shadow[2] = {nullptr, nullptr};
        uint8_t *written[2] = {nullptr, nullptr};
        unsigned long long *report = nullptr;
    } check;
    size_t arena_act_off = 0;        // where the activations start inside the arena (behind the state words)
    size_t layer_act_bytes = 0;      // size of layer_act_mem
    // Qwen3-MoE layers (tl_engine_set_moe_layer): router + stacked experts instead of the dense gate|up / w_down of that layer
    std::vector<tl_moe_weights> moe;  // per layer; num_experts == 0: dense
    int moe_k_max = 0, moe_e_max = 0, moe_i_max = 0;
    char *moe_ws = nullptr;  // one allocation: router logits, ids, scores, gate / up / act rows, expert outputs
    size_t moe_ws_bytes = 0;
    uint16_t *moe_logits = nullptr, *moe_scores = nullptr, *moe_gate = nullptr, *moe_up = nullptr, *moe_act = nullptr, *moe_y = nullptr;
    int32_t *moe_ids = nullptr;
    bool is_moe(int l) const { return l < (int)moe.size() && moe[l].num_experts > 0; }
    float2 *rope_table = nullptr, *rope_cur = nullptr;
    int rope_positions = 0;
    int logits_rows = 0;

    int qkv_dim() const { return (cfg.num_heads + 2 * cfg.num_kv_heads) * cfg.head_dim; }
    int q_dim() const { return cfg.num_heads * cfg.head_dim; }
    const tl_w4 &head() const { return lm_head.weight_dev ? lm_head : embed; }
    uint16_t *layer_k(int l) const { return (uint16_t *)((char *)kpool + (size_t)l * layer_pool_elems * kv_elem_bytes()); }
    uint16_t *layer_v(int l) const { return (uint16_t *)((char *)vpool + (size_t)l * layer_pool_elems * kv_elem_bytes()); }
};

static int aql_drain(tl_engine *e);
static std::string library_dir();

namespace tl {

// ---- profile step bookkeeping -------------------------------------------------------------------------
// kinds: 0..4 GEMV (qkv, o, gate_up, down, lm_head), 5 attention, 6 merge, 7 step end
struct ProfCtx {
    prof_t *buf = nullptr;    // per-workgroup (start, end) pairs of the launch in flight
    prof_t *pairs = nullptr;  // [n][2] reduced (start, end) per launch
    std::vector<int> kinds;
    int cap = 0;
};

static void prof_after(tl_engine *e, ProfCtx *pc, int kind, int n_wg);

// ---- small launch helpers ----------------------------------------------------------------------
static int poke(tl_engine *e, std::vector<std::pair<int32_t *, int32_t>> &items) {
    for (size_t i = 0; i < items.size(); i += 8) {
        PokeArgs a{};
        a.n = (int)std::min<size_t>(8, items.size() - i);
        for (int j = 0; j < a.n; ++j) {
            a.addr[j] = items[i + j].first;
            a.value[j] = items[i + j].second;
        }
        hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(64), 0, e->stream, a);
    }
    items.clear();
    TL_CHECK_LAUNCH("engine poke");
    return TL_OK;
}

static int check_w4(const tl_w4 &w, int rows, int cols, const char *name) {
    if (!w.weight_dev || !w.scales_dev || !w.biases_dev)
        return fail(TL_ERR_INVALID, std::string("engine: null weight pointer in ") + name);
    if (w.rows != rows || w.cols != cols)
        return fail(TL_ERR_INVALID, std::string("engine: unexpected shape for ") + name + " (got " +
                                        std::to_string(w.rows) + "x" + std::to_string(w.cols) + ", want " +
                                        std::to_string(rows) + "x" + std::to_string(cols) + ")");
    if ((uintptr_t)w.weight_dev % 16 != 0)
        return fail(TL_ERR_INVALID, std::string("engine: weight not 16-byte aligned: ") + name);
    return TL_OK;
}

// GEMV with fused prologue/epilogue over M <= 8 rows; splits the rows when the activation tile exceeds LDS.
// ss_in / ss_in_n: partial sums of squares of the rows of `a` ([M][ss_in_n]) when its producer left them; ss_out: where an
// EPI_RESIDUAL GEMV leaves those of `out` ([M][rows / 16]); *ss_out_n = partials per row actually written (0 = none).
static int engine_qmv(tl_engine *e, const tl_w4 &w, const uint16_t *a, uint16_t *out, int M, int pro, int epi,
                      const void *norm_w, const uint16_t *residual, ProfCtx *pc = nullptr, int kind = 0,
                      const float *ss_in = nullptr, int ss_in_n = 0, float *ss_out = nullptr, int *ss_out_n = nullptr,
                      const void *norm_out = nullptr, uint16_t *out_w = nullptr) {
    // norm_out / out_w (EPI_RESIDUAL): also leave out * norm_out for a PRO_RMS_WEIGHTED consumer; the caller has checked
    // (weighted_rows_apply) that the MFMA GEMV takes all rows in one pass -- anything else is an error, not a silent fallback
    if (ss_out_n) *ss_out_n = 0;
    bool all_emitted = ss_out != nullptr && epi == EPI_RESIDUAL && e->gemv_producer_ss;
    int step = std::min(M, 8);  // both GEMV kernels hold at most 8 activation rows (MR <= 8): more rows go in passes of 8
    const bool has_tiled = e->tiled.count(w.weight_dev) != 0;
    auto fits = [&](int rows) {
        return (has_tiled && qmv3_plan(rows, w.cols, w.rows).ok) || qmv_plan(rows, w.cols, w.rows).lds <= 150 * 1024;
    };
    while (!fits(step) && step > 1) step = (step + 1) / 2;
    const int out_cols = epi == EPI_SWIGLU ? w.rows / 2 : w.rows;
    for (int m0 = 0; m0 < M; m0 += step) {
        QmvArgs args{};
        args.scales = (const uint16_t *)w.scales_dev;
        args.biases = (const uint16_t *)w.biases_dev;
        args.b = w.weight_dev;
        args.a = a + (size_t)m0 * w.cols;
        args.out = out + (size_t)m0 * out_cols;
        args.norm_w = (const uint16_t *)norm_w;
        args.residual = residual ? residual + (size_t)m0 * w.rows : nullptr;
        args.eps = e->cfg.rms_norm_eps;
        args.M = std::min(step, M - m0);
        args.N = w.cols;
        args.K = w.rows;
        args.prof = pc ? pc->buf : nullptr;
        const auto tiled = e->tiled.find(w.weight_dev);
        const Qmv3Plan p3 = qmv3_plan(args.M, args.N, args.K);
        if (tiled != e->tiled.end() && p3.ok) {
            Qmv3Args a3{};
            a3.wt = tiled->second.wt;
            a3.sbt = tiled->second.sbt;
            a3.a = args.a;
            a3.out = args.out;
            a3.norm_w = args.norm_w;
            a3.residual = args.residual;
            a3.eps = args.eps;
            a3.M = args.M;
            a3.N = args.N;
            a3.K = args.K;
            a3.prof = args.prof;
            if (e->gemv_producer_ss) {
                if ((pro == PRO_RMSNORM || pro == PRO_RMS_WEIGHTED) && ss_in && ss_in_n > 0) a3.ss_in = ss_in + (size_t)m0 * ss_in_n, a3.ss_n = ss_in_n;
                if (epi == EPI_RESIDUAL && ss_out) a3.ss_out = ss_out + (size_t)m0 * (w.rows / 16);
            }
            if (e->want_tile_max && e->lm_tile_max && epi == EPI_STORE && step == M && w.rows % 16 == 0) {
                a3.tile_max = e->lm_tile_max;
                e->tile_max_rows = M;
            }
            if (out_w) {
                TL_REQUIRE(epi == EPI_RESIDUAL && norm_out && step == M, "engine: weighted rows need the residual epilogue and one pass");
                a3.norm_out = (const uint16_t *)norm_out;
                a3.out_w = out_w;
            }
            if (launch_qmv3_bf16(a3, pro, epi, e->stream) != 0)
                return fail(TL_ERR_UNSUPPORTED, "engine: MFMA GEMV launch failed");
            if (pc) prof_after(e, pc, kind, p3.blocks);
            if (e->linfo) {
                tl_linear_info &li = *e->linfo;
                li.kernel = li.kernel == 0 || li.kernel == 1 ? 1 : li.kernel;
                li.launches += 1;
                li.rows_per_pass = step;
                li.p[0] = p3.MR, li.p[1] = p3.KS, li.p[2] = p3.CW, li.p[3] = p3.LM, li.p[4] = p3.blocks;
            }
            continue;
        }
        all_emitted = false;  // the packed-dot fallback leaves no partials
        TL_REQUIRE(out_w == nullptr && pro != PRO_RMS_WEIGHTED, "engine: weighted rows are a route of the MFMA GEMV only");
        if (launch_qmv_fused_bf16(args, pro, epi, e->stream) != 0)
            return fail(TL_ERR_UNSUPPORTED, "engine: no GEMV configuration for this shape");
        if (pc) prof_after(e, pc, kind, qmv_plan(args.M, args.N, args.K).blocks);
        if (e->linfo) {
            e->linfo->kernel = 3;  // the packed-dot fallback ran (at least once)
            e->linfo->launches += 1;
            e->linfo->rows_per_pass = step;
        }
    }
    TL_CHECK_LAUNCH("engine gemv");
    if (ss_out_n && all_emitted && w.rows % 16 == 0) *ss_out_n = w.rows / 16;
    return TL_OK;
}

// The split-K / skinny-matmul workspace is sized ONCE in tl_engine_create for every shape the engine can launch
// (instantiated graphs hold its address, and a capture cannot synchronise or allocate): a request beyond it is an error.
static int ensure_splitk(tl_engine *e, size_t bytes) {
    if (bytes <= e->splitk_ws_bytes) return TL_OK;
    return fail(TL_ERR_INVALID, "engine: matmul workspace too small for this shape (sized at tl_engine_create: " +
                                    std::to_string(e->splitk_ws_bytes) + " bytes, need " + std::to_string(bytes) + ")");
}

// Reference-semantics GEMM over the checkpoint layout (weights rounded to bf16 first): tl_quantized_matmul.
static int engine_qmm(tl_engine *e, const tl_w4 &w, const uint16_t *a, uint16_t *out, int M) {
    const size_t need = tl_quantized_matmul_workspace_bytes(M, w.cols, w.rows, TL_BF16, 1, 1);
    TL_TRY(ensure_splitk(e, need));
    return tl_quantized_matmul(w.scales_dev, w.biases_dev, a, w.weight_dev, out, M, w.cols, w.rows, 128, 4, TL_BF16, 1, 1,
                               e->splitk_ws, e->splitk_ws_bytes, e->stream);
}

// out = epilogue(a @ W^T) for any number of rows (chunked prefill, batches above 64): the reference's own op sequence --
// W4 MFMA GEMM over the checkpoint layout (quantize.py:54-65 routes rows > 8 to the matmul path, whose tile kernel rounds
// the dequantised weights to bf16 first), then SwiGLU / residual as separate launches.
// From this many rows a chunk's projections run on the plain bf16 GEMM (gemm8.h).  In the lab (back-to-back launches on one weight matrix, which then sits in the
// 256-MB Infinity Cache) gemm8 wins from ~1,500 rows; in the ENGINE every layer streams its own 202 MB of bf16 weights from HBM and the grid counts in
// whole 256-row bands, measured at the end of round 6 (chunked prefill of 6,144 / 8,192 tokens, gemm8 / W4 GEMM, tokens/s): 1,536-row chunks 63.0k / 71.7k,
// 2,048 83.8k / 75.6k, 3,072 77.7k / 78.0k, 4,096 100.5k / 75.9k -- so from 7 bands (until then the constant was 1,536: 12 % slower at exactly that size).
constexpr int GEMM8_MIN_ROWS = 1792;
static bool gemm8_wins(int M, int out_features) {
    (void)out_features;
    return M >= GEMM8_MIN_ROWS;
}
// norm_w / norm_out / norm_done: the RMSNorm that follows an EPI_RESIDUAL projection, taken along by its split-K reduction pass where there is one
// (small chunks on the W4 GEMM); *norm_done says whether norm_out was written -- the caller launches tl_rms_norm otherwise
static int engine_gemm(tl_engine *e, const tl_w4 &w, const uint16_t *a, uint16_t *out, int M, int epi,
                       const uint16_t *residual, const void *norm_w = nullptr, uint16_t *norm_out = nullptr, bool *norm_done = nullptr) {
    if (norm_done) *norm_done = false;
    if (e->use_gemm8 && gemm8_wins(M, w.rows)) {
        const auto wb = e->bf16w.find(w.weight_dev);
        if (wb != e->bf16w.end() && gemm8_applicable(M, w.rows, w.cols)) {
            Gemm8Args g{};
            g.a = a, g.w = wb->second, g.out = out, g.residual = residual, g.M = M, g.N = w.rows, g.K = w.cols;
            if (launch_gemm8_bf16(g, epi, e->stream) != 0) return fail(TL_ERR_UNSUPPORTED, "engine: bf16 GEMM launch failed");
            TL_CHECK_LAUNCH("engine bf16 matmul");
            return TL_OK;
        }
    }
    if (epi != EPI_STORE && M > 8 && e->gemm_fused_epilogue) {  // residual / SwiGLU inside the GEMM store or its split-K reduction
        const size_t need = tl_quantized_matmul_workspace_bytes(M, w.cols, w.rows, TL_BF16, 1, 1);
        TL_TRY(ensure_splitk(e, need));
        TL_TRY(qmm_bf16_epilogue(w.scales_dev, w.biases_dev, a, w.weight_dev, out, M, w.cols, w.rows, epi, residual, e->splitk_ws,
                                 e->splitk_ws_bytes, e->stream, e->fuse_reduce_norm ? (const uint16_t *)norm_w : nullptr, norm_out, e->cfg.rms_norm_eps,
                                 norm_done));
        TL_CHECK_LAUNCH("engine matmul");
        return TL_OK;
    }
    uint16_t *plain = epi == EPI_STORE ? out : (epi == EPI_SWIGLU ? e->gu : e->tmp);
    TL_TRY(engine_qmm(e, w, a, plain, M));
    if (epi == EPI_SWIGLU) {
        const long n4 = (long)M * (w.rows / 2) / 4;
        hipLaunchKernelGGL(swiglu_interleaved_kernel, dim3(ceil_div(n4, 256)), dim3(256), 0, e->stream, plain, out, n4);
    } else if (epi == EPI_RESIDUAL) {
        const long n8 = (long)M * w.rows / 8;
        hipLaunchKernelGGL(residual_add_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, e->stream, residual, plain, out, n8);
    }
    TL_CHECK_LAUNCH("engine matmul");
    return TL_OK;
}

// Does the register-resident matmul (qmm6.h) take this projection at M rows?
static bool qmm6_takes(const tl_engine *e, const tl_w4 &w, int M) {
    return e->use_qmm6 && e->force_linear == 0 && M >= e->qmm3_min_rows && M <= 64 && e->tiled.count(w.weight_dev) != 0 && qmm6_plan(M, w.cols, w.rows).ok;
}
// The weighted rows of a batched step (5 .. 64 rows) travel in fragment order (qmm6.h) from 9 rows -- and from 5 where the row-streaming
// matmul (qmm7.h) is the consumer: ONE answer per (engine, batch) for every producer and consumer of a step.
static bool rows_travel_in_fragment_order(const tl_engine *e, int batch) {
    if (batch > 8) return true;
    if (!e->use_qmm7 || e->layers.empty() || !e->layers[0].wgu.weight_dev) return false;
    return qmm7_plan(batch, e->layers[0].wgu.cols, e->layers[0].wgu.rows).ok && qmm7_plan(batch, e->layers[0].wqkv.cols, e->layers[0].wqkv.rows).ok;
}
// One projection through qmm6: `a` plain rows, or (ss_in given) WEIGHTED rows whose 1 / rms scales the result.  EPI_RESIDUAL: ss_out
// receives rows / 16 partial sums of squares per row, out_w the rows weighted for the next RMSNorm (norm_out).
static int engine_qmm6(tl_engine *e, const tl_w4 &w, const uint16_t *a, uint16_t *out, int M, int epi, const uint16_t *residual,
                       ProfCtx *pc, int kind, const float *ss_in, int ss_in_n, float *ss_out, int *ss_out_n, const void *norm_out,
                       uint16_t *out_w, bool frag = false, int out_w_frag = -1) {
    // frag: the weighted rows on either side (`a` with ss_in, `out_w`) lie in fragment order (qmm6.h) -- the engine's own hand-over;
    // out_w_frag >= 0 decides for out_w alone (the kernel-level entry point)
    if (ss_out_n) *ss_out_n = 0;
    const auto tiled = e->tiled.find(w.weight_dev);
    TL_REQUIRE(tiled != e->tiled.end(), "engine: the register-resident matmul needs the tiled weights");
    const bool a_frag = frag && ss_in != nullptr && epi != EPI_RESIDUAL;
    const Qmm7Plan p7 = (e->use_qmm7 || e->force_qmm7) && a_frag && out_w == nullptr ? qmm7_plan(M, w.cols, w.rows) : Qmm7Plan{};
    TL_REQUIRE(p7.ok || !e->force_qmm7, "engine: the row-streaming matmul takes weighted rows in fragment order (store / SwiGLU) at the shapes of qmm7_plan");
    const Qmm6Plan pl = qmm6_plan(M, w.cols, w.rows, a_frag);
    TL_REQUIRE(p7.ok || pl.ok, "engine: the register-resident matmul does not cover this shape");
    Qmm6Args q{};
    q.wt = tiled->second.wt;
    q.sbt = tiled->second.sbt;
    q.a = a;
    q.out = out;
    q.residual = residual;
    q.norm_out = (const uint16_t *)norm_out;
    q.out_w = out_w;
    q.ss = ss_in;
    q.ss_n = ss_in ? ss_in_n : 0;
    q.ss_out = epi == EPI_RESIDUAL ? ss_out : nullptr;
    q.eps = e->cfg.rms_norm_eps;
    q.M = M;
    q.N = w.cols;
    q.K = w.rows;
    q.prof = pc ? pc->buf : nullptr;
    q.a_frag = a_frag;
    q.out_w_frag = (out_w_frag >= 0 ? out_w_frag != 0 : frag) && out_w != nullptr;
    int n_wg = 0;
    if (p7.ok) {
        if (launch_qmm7_bf16(q, epi, e->stream, &n_wg) != 0) return fail(TL_ERR_UNSUPPORTED, "engine: row-streaming matmul launch failed");
        if (pc) prof_after(e, pc, kind, n_wg);
        if (e->linfo) {
            tl_linear_info &li = *e->linfo;
            li.kernel = 6;
            li.launches += 1;
            li.rows_per_pass = p7.MB * 16;
            li.p[0] = p7.MB, li.p[1] = p7.GPW, li.p[2] = p7.T, li.p[3] = p7.row_blocks, li.p[4] = n_wg;
        }
        TL_CHECK_LAUNCH("engine row-streaming matmul");
        return TL_OK;
    }
    if (launch_qmm6_bf16(q, epi, e->stream, &n_wg) != 0) return fail(TL_ERR_UNSUPPORTED, "engine: register-resident matmul launch failed");
    if (pc) prof_after(e, pc, kind, n_wg);
    if (ss_out_n && q.ss_out) *ss_out_n = w.rows / 16;
    if (e->linfo) {
        tl_linear_info &li = *e->linfo;
        li.kernel = 5;
        li.launches += 1;
        li.rows_per_pass = pl.MB * 16;
        li.p[0] = pl.MB, li.p[1] = pl.GPW, li.p[2] = pl.NSETS, li.p[3] = pl.row_blocks, li.p[4] = n_wg;
    }
    TL_CHECK_LAUNCH("engine register-resident matmul");
    return TL_OK;
}

// One projection of the decode step over `M` activation rows.  Up to 4 rows: the fused MFMA GEMV (weights streamed once,
// RMSNorm / residual / SwiGLU inside).  5 .. 64 rows: the skinny matmul (qmm3.h) for every projection -- at 8 rows the GEMV
// re-stages all rows in every workgroup (qkv 10.3 us against 4.8 + reduction; profiles/r02_labs/batched_rows_routing.log).  More rows, or option "qmm3" = 0: the
// reference's own op sequence -- RMSNorm kernel, W4 MFMA GEMM (quantize.py:54-65 routes rows > 8 to the matmul path),
// then SwiGLU / residual kernels.
// ss_in: partial sums of squares of the rows of `a` when its producer emitted them (fused RMSNorm of the skinny matmul),
// else nullptr.  ss_out / *ss_emitted: where the slice reduction should leave the partials of `out`, and whether it did.
// keep (EPI_STORE only): the caller's consumer adds the slices itself -- the reduction launch is skipped and *keep says where the
// fp32 planes are; `out` is then NOT written.  Left empty (partial == nullptr) when another kernel took the projection.
// rows that engine_linear hands to the GEMV before it considers anything else
static bool gemv_takes_rows(const tl_engine *e, int M) {
    return e->force_linear == 1 || (e->force_linear != 2 && (M < e->qmm3_min_rows || (M <= 8 && !e->use_qmm3)));
}
// Can `producer` (EPI_RESIDUAL) leave its rows weighted for `consumer` (PRO_RMS_WEIGHTED)?  Both must be single-pass MFMA GEMVs,
// the producer must leave the sums of squares, and the consumer's row must sit in its register chunks (qmv3.h, reg_path).
static bool weighted_rows_apply(const tl_engine *e, const tl_w4 &producer, const tl_w4 &consumer, int M) {
    if (!e->gemv_weighted_rows || !e->gemv_producer_ss || !gemv_takes_rows(e, M) || M > 8) return false;
    if (e->tiled.count(producer.weight_dev) == 0 || e->tiled.count(consumer.weight_dev) == 0) return false;
    const Qmv3Plan pp = qmv3_plan(M, producer.cols, producer.rows), pcn = qmv3_plan(M, consumer.cols, consumer.rows);
    if (!pp.ok || !pcn.ok || producer.rows != consumer.cols) return false;
    return qmv3_takes_weighted_rows(pcn, consumer.cols, producer.rows / 16);
}
// Does engine_linear send M rows of this projection to the K-sliced skinny matmul (qmm3.h)?  ONE predicate for the router below and for
// every caller that plans around its answer (enqueue_step decides from it whether a producer will leave weighted rows).
static bool takes_skinny_matmul(const tl_engine *e, const tl_w4 &w, int M) {
    return !gemv_takes_rows(e, M) && e->use_qmm3 && M <= 64 && e->tiled.count(w.weight_dev) != 0 && qmm3_plan(M, w.cols, w.rows, e->qmm3_mode).ok;
}
struct KeptPartials {
    const float *partial = nullptr;
    int slices = 0;
    long plane = 0;  // elements between slices (= rows * output columns)
};
static int engine_linear(tl_engine *e, const tl_w4 &w, const uint16_t *a, uint16_t *out, int M, int pro, int epi,
                         const void *norm_w, const uint16_t *residual, ProfCtx *pc, int kind, const float *ss_in_any = nullptr,
                         float *ss_out = nullptr, bool *ss_emitted = nullptr, KeptPartials *keep = nullptr, int ss_in_n = QM3_SS,
                         int *ss_out_n = nullptr, const void *norm_out = nullptr, uint16_t *out_w = nullptr, bool out_w_frag = false) {
    // ss_in_any holds ss_in_n partials per row; the skinny matmul reads exactly QM3_SS of them, the GEMV any number.
    // *ss_out_n (when asked for) = partials per row left in ss_out: QM3_SS by the slice reduction, rows / 16 by a GEMV, 0 = none
    if (ss_emitted) *ss_emitted = false;
    if (ss_out_n) *ss_out_n = 0;
    if (keep) *keep = KeptPartials{};
    const float *ss_in = qmm3_takes_ss(ss_in_n) ? ss_in_any : nullptr;
    if (gemv_takes_rows(e, M))
        return engine_qmv(e, w, a, out, M, pro, epi, norm_w, residual, pc, kind, ss_in_any, ss_in_n, ss_out, ss_out_n, norm_out, out_w);
    TL_REQUIRE(pro != PRO_RMS_WEIGHTED && (out_w == nullptr || (epi == EPI_RESIDUAL && norm_out != nullptr)),
               "engine: the skinny matmul leaves weighted rows behind a residual epilogue only, and takes none");
    const tl_engine_config &c = e->cfg;
    const uint16_t *in = a;
    // qmm3_min_rows .. 64 rows (batched decode): K-sliced skinny MFMA matmul over the tiled weights, then the slice
    // reduction with the epilogue.  RMSNorm runs as its own launch (a slice does not see the whole row).
    const auto tiled = e->tiled.find(w.weight_dev);
    const Qmm3Plan p3 = qmm3_plan(M, w.cols, w.rows, e->qmm3_mode);
    if (takes_skinny_matmul(e, w, M)) {
        const bool fused_norm = pro == PRO_RMSNORM && ss_in != nullptr && e->fuse_norm;
        if (pro == PRO_RMSNORM && !fused_norm) {
            TL_TRY(tl_rms_norm(a, norm_w, e->xn, M, w.cols, c.rms_norm_eps, TL_BF16, e->stream));
            in = e->xn;
        }
        if (e->planes_now) TL_REQUIRE(p3.partial_bytes <= e->planes_now_bytes, "engine: per-layer slice planes too small for this shape");
        else TL_TRY(ensure_splitk(e, p3.partial_bytes));
        Qmm3Args q{};
        q.wt = tiled->second.wt;
        q.sbt = tiled->second.sbt;
        q.a = in;
        q.partial = e->planes_now ? e->planes_now : (float *)e->splitk_ws;
        q.M = M;
        q.N = w.cols;
        q.K = w.rows;
        q.prof = pc ? pc->buf : nullptr;
        q.norm_w = (const uint16_t *)norm_w;
        q.ss = ss_in;
        q.ss_n = ss_in_n;
        q.eps = c.rms_norm_eps;
        if (launch_qmm3_bf16(q, e->stream, fused_norm ? PRO_RMSNORM : PRO_NONE, e->qmm3_mode) != 0)
            return fail(TL_ERR_UNSUPPORTED, "engine: skinny matmul launch failed");
        if (pc) prof_after(e, pc, kind, p3.persistent ? p3.grid_x : p3.grid_x * p3.slices);
        float *ss_dst = (ss_out && e->fuse_norm && qmm3_reduce_can_emit_ss(epi, w.rows)) ? ss_out : nullptr;
        const bool kept = keep != nullptr && epi == EPI_STORE && ss_dst == nullptr;
        if (kept) {
            keep->partial = q.partial;
            keep->slices = p3.slices;
            keep->plane = (long)M * w.rows;
        } else {
            int reduce_wg = 0;
            if (launch_qmm3_reduce_bf16(q.partial, p3.slices, M, w.rows, epi, residual, out, q.prof, e->stream, ss_dst, &reduce_wg,
                                        out_w ? (const uint16_t *)norm_out : nullptr, out_w, out_w_frag ? 1 : 0) != 0)
                return fail(TL_ERR_UNSUPPORTED, "engine: skinny matmul reduction launch failed");
            if (pc) prof_after(e, pc, kind, reduce_wg);
        }
        if (ss_emitted) *ss_emitted = ss_dst != nullptr;
        if (ss_out_n) *ss_out_n = ss_dst != nullptr ? QM3_SS : 0;
        TL_CHECK_LAUNCH("engine skinny matmul");
        if (e->linfo) {
            tl_linear_info &li = *e->linfo;
            li.kernel = 2;
            li.launches += (kept ? 1 : 2) + (pro == PRO_RMSNORM && !fused_norm ? 1 : 0);
            li.rows_per_pass = M;
            li.p[0] = p3.MB, li.p[1] = p3.persistent ? 0 : p3.TW, li.p[2] = p3.LM, li.p[3] = p3.slices;
            li.p[4] = p3.persistent ? p3.grid_x : p3.grid_x * p3.slices;
        }
        return TL_OK;
    }
    if (e->force_linear == 2) return fail(TL_ERR_UNSUPPORTED, "engine: the skinny matmul does not cover this shape");
    TL_REQUIRE(out_w == nullptr, "engine: no kernel leaves weighted rows for this shape");
    if (M <= 8) return engine_qmv(e, w, a, out, M, pro, epi, norm_w, residual, pc, kind, ss_in_any, ss_in_n, ss_out, ss_out_n);
    if (pro == PRO_RMSNORM) {
        TL_TRY(tl_rms_norm(a, norm_w, e->xn, M, w.cols, c.rms_norm_eps, TL_BF16, e->stream));
        in = e->xn;
    }
    if (e->linfo) e->linfo->kernel = 4;
    return engine_gemm(e, w, in, out, M, epi, residual);
}

// Context split of the decode attention: power-of-two bucket >= context, fixed windows of C tokens per workgroup.
struct SplitPlan {
    int n_splits, tokens_per_split;
    int rq;  // query heads per workgroup
    long key() const { return ((long)rq << 40) | ((long)n_splits << 24) | (long)tokens_per_split; }
};
// Measured on MI355X (profiles/README.md, profiles/r02_labs): a decode-attention workgroup is bound by its dependent latency
// chain and by how many L2 misses ONE CU keeps in flight (~32 KiB), not by chip bandwidth.  Few sequences and short
// contexts: one query head and a 64-token window (32 KiB of K/V) per workgroup, partials merged by the wo GEMV (2 / 4 / 8
// windows of one sequence) or by a second launch.  Many sequences or long contexts: one workgroup per GQA group walking 64-token
// stages, so that the K/V window is read from HBM once.  (A "wide" one-head kernel -- the whole 64..512-token window in one
// workgroup, no merge -- was built in round 2, lost on every context it was meant for (8.2 us per layer against 2.9 + 1.3) and was
// removed in round 3 together with the last-arriver in-kernel merge, which measured neutral.)
// the attention plan's lab knobs (environment, read when an engine -- or the standalone operator's stand-in for one -- is set up)
static void read_attention_knobs(tl_engine *e) {
    if (const char *q = getenv("TL_ATTN_MFMA")) e->attn_mfma = atoi(q) != 0;
    if (const char *q = getenv("TL_ATTN_RQ")) e->attn_rq = atoi(q) <= 0 ? 0 : (atoi(q) == 1 ? 1 : AD_RQ);
    if (const char *q = getenv("TL_ATTN_MAX_SPLITS")) e->attn_max_splits = e->attn_max_splits_gqa = std::min(256, std::max(1, atoi(q)));
    if (const char *q = getenv("TL_ATTN_MIN_TOKENS")) e->attn_min_tokens = std::max(64, atoi(q)), e->attn_min_tokens_auto = false;
}
static SplitPlan pick_decode_splits(const tl_engine *e, int batch, int max_ctx) {
    const int rep = e->cfg.num_heads / e->cfg.num_kv_heads;
    int rq = e->attn_rq;
    // two sequences re-read twice the windows: the GQA-group walk takes over at a quarter of the context (round 3, 2 sequences at 1,500 tokens
    // one head per workgroup 1.326 against 1.383 ms per step, at 3,000 tokens 1.527 against 1.465)
    // (round 4, with the group walk on the matrix cores: 2 sequences at 1,500 tokens 1.46 -> 1.38 ms per step, at 700 a tie; one sequence
    // at 700 / 1,500 / 3,000 tokens 1.05 / 1.11 / 1.24 for one head per workgroup against 1.13 / 1.21 / 1.24)
    if (rq <= 0) rq = (max_ctx <= (batch <= 1 ? e->attn_rq1_ctx : e->attn_rq1_ctx / 4) && batch <= e->attn_rq1_batch) ? 1 : AD_RQ;
    if (rq != 1) rq = AD_RQ;
    int bucket = 64;
    while (bucket < max_ctx) bucket *= 2;
    const int chunks = (rep + rq - 1) / rq;
    const int base = std::max(1, batch * e->cfg.num_kv_heads * chunks);
    int s = 1;
    // Windows grow with the context (round 3, tools/decode_ab.py; profiles/r03_labs/attention_window_size_*.jsonl).  Up to 512 tokens of
    // context 64-token windows are the best at 1-4 sequences (one sequence at 200 tokens: 128-token windows 1.022 against 0.993 ms).
    // Beyond, 256-token windows: one sequence 700 / 1,500 / 3,000 tokens 1.078 -> 1.060, 1.218 -> 1.143, 1.386 -> 1.246 ms per step (the wo
    // GEMV still merges the 4 / 8 windows up to 2k tokens, 16 stand where 64 stood at 4k), two sequences 1.310 -> 1.245, 1.510 -> 1.361,
    // 1.666 -> 1.540, four sequences at 1,000 tokens 1.762 -> 1.610.  Two exceptions at 257..512 tokens, both measured: one sequence
    // 128-token windows (4 instead of 8 partials for the wo GEMV: 1.018 -> 1.014), 5-8 sequences 128 (1.882 -> 1.824 at 8).
    // TL_ATTN_MIN_TOKENS pins one size (lab).
    // Batched steps (5+ sequences) up to 1,024 tokens of context, re-measured at the end of round 6 on the round's kernels (matrix-core walk, shared prologue;
    // tools/lab/ab_attn_splits_batched.sh, profiles/r06_labs/README.md section 9): ONE workgroup per CU at most and windows of 128+ tokens -- the merge launch
    // and the second dependent trip of a short window cost more than a longer walk once every sequence's KV head has a workgroup: 5 / 8 / 12 / 16 / 23 sequences
    // at ~600 tokens 1.47 / 1.52 / 1.56 / 1.59 / 2.04 -> 1.39 / 1.42 / 1.53 / 1.57 / 1.87 ms per step, 9-23 at ~300 tokens -3 ... -6 %.
    const bool batched_short = e->attn_min_tokens_auto && batch >= 5 && bucket <= 1024;
    int min_tokens = e->attn_min_tokens;
    if (e->attn_min_tokens_auto) {
        if (bucket > 512 && batch <= 4) min_tokens = 256;
        else if (bucket == 512 && batch == 1) min_tokens = 128;
        else if (batched_short) min_tokens = 128;
    }
    // few sequences: split for latency (up to 2048 short-lived workgroups); many sequences: the chip is already full, longer
    // windows amortise the per-workgroup prologue and skip the merge launch (measured at 16 and 64 sequences)
    // (5-7 sequences beyond 1,024 tokens too: 4 windows on 160-224 workgroups against 8 on 320-448 -- 5 / 6 / 7 sequences at 2,000 tokens 1.64 / 1.67 / 1.71 ->
    // 1.57 / 1.63 / 1.70 ms per step, at 8,000 2.45 / 2.57 / 2.65 -> 2.26 / 2.43 / 2.62; 8 and 16 sequences keep 512: 8 x 4,000 2.11 against 2.19)
    // (... and 8-16 sequences, the round's last sweep at 1,500 / 3,000 / 6,000 tokens: 10 sequences 1.80 / 2.17 / 2.96 -> 1.72 / 2.03 / 2.74 ms per step, 12: 1.85 / 2.27 / 3.08 ->
    // 1.78 / 2.15 / 3.00, 14: -2 %, 8 and 16 within 1 % either way; from 17 sequences no split at all, below)
    const bool few_long = e->attn_min_tokens_auto && batch >= 5 && batch <= 16 && bucket > 1024;
    // (2-4 sequences on the GQA-group walk likewise, end of round 6: one workgroup per CU -- 3 / 4 sequences at 4,500 tokens 1.80 / 1.97 -> 1.66 / 1.81 ms per step,
    // at 8,000 2.02 / 2.29 -> 1.86 / 2.10, two sequences at 4,500 / 8,000 1.42 / 1.59 -> 1.36 / 1.55, 4 x 32,000 4.55 -> 4.41; one sequence: 8 KV heads x 32 windows already)
    const bool few_gqa = e->attn_min_tokens_auto && batch >= 2 && batch <= 4 && rq == AD_RQ;
    const int wg_cap = e->attn_wg_cap > 0 ? e->attn_wg_cap : (batch <= 4 ? (few_gqa ? 256 : 2048) : ((batched_short || few_long) ? 256 : 512));
    // a whole GQA group per workgroup (long contexts / several sequences): at most 32 windows -- one workgroup per CU for one sequence;
    // measured at 8k 666 -> 680 tok/s against 64 windows, 32k unchanged (round 3)
    int max_splits = rq == AD_RQ ? e->attn_max_splits_gqa : e->attn_max_splits;
    // (64 windows = two workgroups per CU for the matrix-core walk at 32k: 1.931 / 1.896 -> 1.898 / 1.866 ms per step in one same-box
    // A/B, 1.874 -> 1.891 in the next: within the noise, not taken)
    // Many sequences at short contexts: one window per sequence and NO merge launch (round 4, same-box A/B at ~190 / ~660 tokens,
    // profiles/r04_labs/README.md): 12 / 16 sequences 1.623 -> 1.603 / 1.661 -> 1.634 ms per step at ~190 tokens (at ~660 the split stays:
    // 16 sequences 1.885 against 2.000), 24 / 32 sequences 2.15 -> 2.03 / 2.215 -> 2.07 at ~190 and 32 sequences 2.56 -> 2.495 at ~660.
    // The merge launch and its boundary cost ~2.5 us per layer; the unsplit walk of a few stages costs less once every CU has a workgroup.
    // (from 5 sequences at up to 256 tokens since the end of round 6: 5 / 7 / 9 / 10 sequences at ~150 tokens 1.30 / 1.35 / 1.44 / 1.45 -> 1.26 / 1.28 / 1.31 / 1.33 ms per step)
    // (from 17 sequences up to 8,192 tokens too, end of round 6: 160+ workgroups walking whole contexts beat twice as many on halves + a merge launch -- 20 / 23 / 24 / 32
    // sequences at 1,500 tokens 2.49 / 2.56 / 2.63 / 2.82 -> 2.25 / 2.36 / 2.40 / 2.70 ms per step, 24 at 2,500 / 4,000 / 8,000 3.10 / 4.20 / 6.44 -> 2.91 / 3.98 / 6.12;
    // 16 sequences at 2,000 want their 4 windows: 2.16 against 2.29)
    if (e->attn_min_tokens_auto && ((batch >= 24 && bucket <= 1024) || (batch >= 5 && bucket <= 256) || (batch >= 17 && bucket <= 8192))) max_splits = 1;  // (17-19 sequences measured last: 1,500 / 3,000 / 6,000 tokens 2.44 / 3.19 / 4.85 -> 2.20 / 2.87 / 4.13 ms per step)
    while (s * 2 <= bucket / min_tokens && s * 2 * base <= wg_cap && s * 2 <= max_splits) s *= 2;  // >= min_tokens per workgroup
    // Windows sized to the context, not to its power-of-two bucket: a workgroup walks its whole window in 64-token stages
    // whether or not the tokens exist, so a 33k context on a 64k bucket spent half of every window on masked loads (r02:
    // 63 us per layer in the step against 44 us for the same kernel on an exactly filled bucket).  The split COUNT stays a
    // power of two (the kernels decode blockIdx with shifts); the plan (and its captured graph) changes every 64 * s tokens.
    const int per_split = ((max_ctx + s - 1) / s + 63) / 64 * 64;
    return SplitPlan{s, std::max(64, std::min(per_split, bucket / s)), rq};
}

// head_dim 128 with a whole GQA group per workgroup is the only shape that takes qkv slice partials (batched decode of 5+ rows)
static bool attn_takes_qkv_partials(int head_dim, int rq) { return head_dim == 128 && rq == AD_RQ; }
template <int VD, bool SP, bool IP = false>
static void launch_attn_decode_sp(const AttnDecodeArgs &a, dim3 grid, hipStream_t st, int rq) {
    const size_t lds = (size_t)16 * rq * (16 * VD + 2) * sizeof(float);
    if constexpr (VD == 8) {
        if (a.key_scales != nullptr) {  // FP8 pages (kv8.h)
            if (a.qkv_partial != nullptr && rq == AD_RQ) {
                const size_t staged = (size_t)(2 + AD_RQ) * 16 * VD * sizeof(uint16_t);
                hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, AD_RQ, SP, IP, true, true>), grid, dim3(256), lds + staged, st, a);
            } else if (rq == 1) hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, 1, SP, IP, false, true>), grid, dim3(256), lds, st, a);
            else hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, AD_RQ, SP, IP, false, true>), grid, dim3(256), lds, st, a);
            return;
        }
        if (a.qkv_partial != nullptr && rq == AD_RQ) {
            const size_t staged = (size_t)(2 + AD_RQ) * 16 * VD * sizeof(uint16_t);
            hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, AD_RQ, SP, IP, true>), grid, dim3(256), lds + staged, st, a);
            return;
        }
    }
    if (rq == 1) hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, 1, SP, IP>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((attn_decode_fused_kernel<VD, 4, AD_RQ, SP, IP>), grid, dim3(256), lds, st, a);
}
template <int VD>
static void launch_attn_decode(const AttnDecodeArgs &a, dim3 grid, hipStream_t st, int rq, bool mfma) {
    if constexpr (VD == 8) {
        // a whole GQA group per workgroup: the walk on the matrix cores (attn_mfma.h) from 128-token windows -- a stage is 4 waves x 32
        // tokens, and on 64-token windows half of the waves idle (round 4, same-box A/B: 4 / 8 sequences at ~230 tokens 1.445 / 1.539
        // against 1.416 / 1.527 ms per step for the VALU walk; 4 sequences at 2k 1.691 against 1.729, one at 8k / 32k 1.256 / 1.90
        // against 1.35 / 2.13)
        if (mfma && a.tokens_per_split >= 128 && attn_decode_mfma_applicable(a, 16 * VD, rq)) {
            launch_attn_decode_mfma(a, grid, st);
            return;
        }
    }
    const bool single_page = a.tokens_per_split <= a.page_size && a.page_size % a.tokens_per_split == 0;
    // every 64-token stage of a window inside one page: windows are multiples of 64 tokens, pages a power of two >= 64
    const bool stage_page = !single_page && a.page_shift >= 6 && a.tokens_per_split % 64 == 0;
    if (single_page) launch_attn_decode_sp<VD, true>(a, grid, st, rq);
    else if (stage_page) launch_attn_decode_sp<VD, false, true>(a, grid, st, rq);
    else launch_attn_decode_sp<VD, false>(a, grid, st, rq);
}

// Can the wo GEMV of ONE decode row take the attention split partials instead of the merged row (qmv3.hip,
// launch_qmv3_attn_merge_bf16: the instantiated plans)?
static bool wo_merge_applicable(const tl_engine *e, const tl_w4 &wo, int batch, const SplitPlan &sp) {
    if (!e->wo_merges_attn || batch != 1 || e->force_linear != 0) return false;
    if (sp.n_splits != 2 && sp.n_splits != 4 && sp.n_splits != 8) return false;
    if (e->cfg.head_dim != 128 || wo.cols != e->cfg.num_heads * 128 || e->tiled.count(wo.weight_dev) == 0) return false;
    if (1 >= e->qmm3_min_rows && e->use_qmm3) return false;  // a single row would not take the GEMV
    const Qmv3Plan pl = qmv3_plan(1, wo.cols, wo.rows);
    const bool shape = (pl.KS == 2 && pl.CW == 4) || (pl.KS == 4 && pl.CW == 4) || (pl.KS == 8 && pl.CW == 8);
    return pl.ok && pl.MR == 1 && shape && wo.cols / 8 <= 2 * pl.CW * 64;
}
// h = x + merge(attention partials) @ wo^T for one row.
static int engine_wo_merge(tl_engine *e, const tl_w4 &wo, const uint16_t *residual, uint16_t *out, int n_splits, ProfCtx *pc,
                           float *ss_out = nullptr, int *ss_out_n = nullptr, const void *norm_out = nullptr, uint16_t *out_w = nullptr) {
    if (ss_out_n) *ss_out_n = 0;
    const auto tiled = e->tiled.find(wo.weight_dev);
    Qmv3Args a3{};
    a3.wt = tiled->second.wt;
    a3.sbt = tiled->second.sbt;
    a3.a = nullptr;
    a3.out = out;
    a3.residual = residual;
    a3.eps = e->cfg.rms_norm_eps;
    a3.M = 1;
    a3.N = wo.cols;
    a3.K = wo.rows;
    a3.prof = pc ? pc->buf : nullptr;
    a3.merge_ws = e->attn_ws;
    if (ss_out && e->gemv_producer_ss && wo.rows % 16 == 0) {
        a3.ss_out = ss_out;
        if (ss_out_n) *ss_out_n = wo.rows / 16;
    }
    if (out_w) a3.norm_out = (const uint16_t *)norm_out, a3.out_w = out_w;
    if (launch_qmv3_attn_merge_bf16(a3, n_splits, e->stream) != 0)
        return fail(TL_ERR_UNSUPPORTED, "engine: no wo GEMV that merges the attention partials for this shape");
    if (pc) prof_after(e, pc, 1, qmv3_plan(1, wo.cols, wo.rows).blocks);
    TL_CHECK_LAUNCH("engine wo gemv with merge");
    return TL_OK;
}

// q/k-norm + RoPE + KV append + decode attention of one layer over slots [0, batch) (+ the merge launch when the context
// is split).  qkv [batch, (Hq + 2 Hkv) D] -> out [batch, Hq D]; partials in e->attn_ws.
static int engine_attention(tl_engine *e, const uint16_t *qkv, const void *q_norm, const void *k_norm, uint16_t *key_pages,
                            uint16_t *value_pages, uint16_t *out, int batch, const SplitPlan &sp, ProfCtx *pc,
                            const KeptPartials *qkv_parts = nullptr, const tl_w4 *merging_wo = nullptr, bool *merge_left = nullptr,
                            float *key_scales = nullptr, float *value_scales = nullptr) {
    if (merge_left) *merge_left = false;
    const tl_engine_config &c = e->cfg;
    const int D = c.head_dim;
    const int n_splits = sp.n_splits;
    const int rep = c.num_heads / c.num_kv_heads;
    const int chunks = (rep + sp.rq - 1) / sp.rq;
    AttnDecodeArgs a{};
    a.qkv = qkv;
    a.q_norm_w = (const uint16_t *)q_norm;
    a.k_norm_w = (const uint16_t *)k_norm;
    a.key_pages = key_pages;
    a.value_pages = value_pages;
    a.key_scales = key_scales;
    a.value_scales = value_scales;
    TL_REQUIRE(key_scales == nullptr || D == 128, "engine: FP8 pages need head_dim 128");
    a.block_table = e->block_table;
    a.context_lens = e->context_lens;
    a.out = out;
    a.ws = e->attn_ws;
    a.page_size = c.page_size;
    a.max_pages = c.max_pages_per_seq;
    a.num_heads = c.num_heads;
    a.num_kv_heads = c.num_kv_heads;
    a.scale = 1.0f / sqrtf((float)D);
    a.eps = c.rms_norm_eps;
    a.rope_base = c.rope_theta;
    a.n_splits = n_splits;
    a.n_row_chunks = chunks;
    a.tokens_per_split = sp.tokens_per_split;
    a.split_shift = 0;
    while ((1 << a.split_shift) < n_splits) ++a.split_shift;
    a.rep = rep;
    a.page_shift = -1;
    for (int sh = 0; sh < 30; ++sh)
        if ((1 << sh) == c.page_size) a.page_shift = sh;
    a.rope_cur = e->rope_cur;
    a.prof = pc ? pc->buf : nullptr;
    if (qkv_parts && qkv_parts->partial) {
        TL_REQUIRE(attn_takes_qkv_partials(D, sp.rq), "engine: this decode-attention plan does not read qkv slice partials");
        a.qkv_partial = qkv_parts->partial;
        a.qkv_slices = qkv_parts->slices;
        a.qkv_plane = qkv_parts->plane;
    }
    TL_REQUIRE((size_t)batch * c.num_heads * n_splits * (D + ATTN_WS_PAD) * sizeof(float) <= e->attn_ws_bytes || n_splits == 1,
               "engine: attention workspace too small for this split plan");
    const dim3 grid(n_splits * chunks, c.num_kv_heads, batch);
    switch (D) {
        case 128: launch_attn_decode<8>(a, grid, e->stream, sp.rq, e->attn_mfma); break;
        case 64: launch_attn_decode<4>(a, grid, e->stream, sp.rq, false); break;
        case 32: launch_attn_decode<2>(a, grid, e->stream, sp.rq, false); break;
        default: return fail(TL_ERR_UNSUPPORTED, "engine: head_dim must be 32, 64 or 128");
    }
    if (pc) prof_after(e, pc, 5, (int)(grid.x * grid.y * grid.z));
    // the consumer (the wo GEMV of a single row) merges the partials itself: no merge launch, `out` is not written
    const bool leave_merge = merging_wo != nullptr && merge_left != nullptr && wo_merge_applicable(e, *merging_wo, batch, sp);
    if (leave_merge) *merge_left = true;
    e->last_attn_launches = 1 + ((n_splits > 1 && !leave_merge) ? 1 : 0);
    if (n_splits > 1 && !leave_merge) {
        const dim3 mg(batch * c.num_heads), mb(128);
        prof_t *pb = pc ? pc->buf : nullptr;
        int merge_wg = batch * c.num_heads;
        switch (n_splits) {
            case 2: hipLaunchKernelGGL(attn_merge_kernel<2>, mg, mb, 0, e->stream, e->attn_ws, out, D, pb); break;
            case 4: hipLaunchKernelGGL(attn_merge_kernel<4>, mg, mb, 0, e->stream, e->attn_ws, out, D, pb); break;
            case 8: hipLaunchKernelGGL(attn_merge_kernel<8>, mg, mb, 0, e->stream, e->attn_ws, out, D, pb); break;
            default:  // 16 and more splits: a row's partials spread over D / 32 workgroups x 8 split groups
                merge_wg *= (D + 31) / 32;
                hipLaunchKernelGGL(attn_merge_cols_kernel, dim3(batch * c.num_heads, (D + 31) / 32), dim3(256), 0, e->stream, e->attn_ws,
                                   out, D, n_splits, pb);
                break;
        }
        if (pc) prof_after(e, pc, 6, merge_wg);
    }
    TL_CHECK_LAUNCH("engine attention");
    return TL_OK;
}

// The MLP of a Qwen3-MoE layer over `rows` activation rows: xn = RMSNorm(h) is given; out = h + sum_j score_j * expert_j(xn)
// (reference: moe.py:69-89 inside qwen3_week3.py:204-205).  Every launch reads device-resident ids: graph-capturable.
static int engine_moe_mlp(tl_engine *e, int l, const uint16_t *xn, const uint16_t *h, uint16_t *out, int rows, ProfCtx *pc) {
    const tl_moe_weights &m = e->moe[l];
    const int D = e->cfg.hidden_size, E = m.num_experts, k = m.experts_per_token, I = m.intermediate_size;
    TL_REQUIRE(e->moe_ws != nullptr && rows <= e->rows_cap && (long)rows * k <= 65535, "engine: MoE workspace missing or too many expert rows");
    // router logits [rows, E] through the reference-semantics matmul (quantized_linear of the router, moe.py:44)
    TL_TRY(engine_qmm(e, m.router, xn, e->moe_logits, rows));
    hipLaunchKernelGGL(moe_route_kernel, dim3(rows), dim3(256), 0, e->stream, e->moe_logits, E, k, m.norm_topk_prob, e->moe_ids, e->moe_scores);
    const int er = rows * k;  // expert rows: token-major, the token's top_k experts in descending probability
    TL_TRY(gather_qmv_bf16(m.gate_scales_dev, m.gate_biases_dev, xn, m.gate_dev, e->moe_ids, e->moe_gate, er, D, I, E, k, e->stream));
    TL_TRY(gather_qmv_bf16(m.up_scales_dev, m.up_biases_dev, xn, m.up_dev, e->moe_ids, e->moe_up, er, D, I, E, k, e->stream));
    const long n8 = (long)er * I / 8;
    hipLaunchKernelGGL(moe_silu_mul_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, e->stream, e->moe_gate, e->moe_up, e->moe_act, n8);
    TL_TRY(gather_qmv_bf16(m.down_scales_dev, m.down_biases_dev, e->moe_act, m.down_dev, e->moe_ids, e->moe_y, er, I, D, E, 1, e->stream));
    hipLaunchKernelGGL(moe_combine_kernel, dim3(rows), dim3(256), 0, e->stream, e->moe_y, e->moe_scores, h, out, D, k);
    (void)pc;  // the MoE launches carry no in-kernel stamps: tl_engine_profile_step leaves them out of its kinds
    TL_CHECK_LAUNCH("engine MoE layer");
    return TL_OK;
}

// One fused decode step over slots [0, batch).
static int enqueue_step(tl_engine *e, int batch, SplitPlan sp, ProfCtx *pc = nullptr) {
    const tl_engine_config &c = e->cfg;
    // x enters the step from the embedding gather (embed_slots_kernel / the previous step's step_end_kernel), which leaves the
    // per-row partial sums of squares in ss_x; every slice reduction that rewrites x or h refreshes them (or says it did not)
    int x_ss = QM3_SS;  // partials per row in ss_x (0 = none): QM3_SS from the embedding kernels, then whatever the last writer of x left
    // 5 .. 64 rows on the register-resident matmul: xn holds x weighted by the NEXT RMSNorm's weight whenever xw is set (written by
    // the w_down epilogue of the previous layer, or by one pointwise launch ahead of layer 0)
    bool xw = false;
    // where the residual stream and its sums of squares stand (the shared buffers, or the last layer's own in per-layer mode)
    uint16_t *x_cur = e->x;
    float *ssx_cur = e->ss_x;
    // per-layer hand-over buffers: the fused-GEMV rows of a dense model whose attention partials fit the per-layer workspace
    const bool ws_fits = sp.n_splits == 1 || (size_t)batch * c.num_heads * sp.n_splits * (c.head_dim + ATTN_WS_PAD) * sizeof(float) <= e->layer_ws_bytes;
    bool per_layer = e->layer_act_rows > 0 && batch <= e->layer_act_rows && gemv_takes_rows(e, batch) && ws_fits;
    // ... and the rows of the batched-matmul step (5 .. 64): every projection on the register-resident or the K-sliced matmul with its
    // hand-over through weighted rows (the branch below), slice planes inside the per-layer ones
    bool per_layer_b = e->layer_act_rows > 0 && batch <= e->layer_act_rows && !gemv_takes_rows(e, batch) && ws_fits && e->force_linear == 0;
    for (int l = 0; l < c.num_layers; ++l) {
        if (e->is_moe(l)) per_layer = per_layer_b = false;
        const tl_layer_weights &w = e->layers[l];
        if (per_layer_b) {  // every condition of the batched branch below, known ahead: no layer may fall out of it half way through a step
            const bool wo_ok = (qmm6_takes(e, w.wo, batch) && qmm3_takes_ss(w.wo.rows / 16)) ||
                               (takes_skinny_matmul(e, w.wo, batch) && e->fuse_norm && qmm3_reduce_can_emit_ss(EPI_RESIDUAL, w.wo.rows));
            const bool down_ok = (takes_skinny_matmul(e, w.wdown, batch) && e->fuse_norm && qmm3_reduce_can_emit_ss(EPI_RESIDUAL, w.wdown.rows)) ||
                                 (qmm6_takes(e, w.wdown, batch) && qmm3_takes_ss(w.wdown.rows / 16));
            if (!(w.wgu.weight_dev && qmm6_takes(e, w.wqkv, batch) && qmm6_takes(e, w.wgu, batch) && qmm6_takes(e, e->head(), batch) && wo_ok && down_ok))
                per_layer_b = false;
        }
    }
    // only a step whose hand-overs all live at addresses written once per step may be replayed without cache maintenance (tl_engine_decode)
    e->step_written_once = per_layer || per_layer_b;
    // the residual stream of the batched branch: where x, x weighted for the next RMSNorm and its sums of squares stand
    uint16_t *bx = e->x, *bxw = e->xn;
    float *bssx = e->ss_x;
    for (int l = 0; l < c.num_layers; ++l) {
        const tl_layer_weights &w = e->layers[l];
        // the sliced matmul as the producer of weighted rows (its reduction writes them): wo here, w_down below
        auto sliced_leaves_weighted = [&](const tl_w4 &m) {  // (the router's own predicate + what its reduction needs to emit the hand-over)
            return takes_skinny_matmul(e, m, batch) && e->fuse_norm && qmm3_reduce_can_emit_ss(EPI_RESIDUAL, m.rows);
        };
        const bool wo6_ok = !e->is_moe(l) && qmm6_takes(e, w.wo, batch) && qmm3_takes_ss(w.wo.rows / 16);
        if (!e->is_moe(l) && w.wgu.weight_dev != nullptr && qmm6_takes(e, w.wgu, batch) && x_ss > 0 && qmm3_takes_ss(x_ss) &&
            (wo6_ok || sliced_leaves_weighted(w.wo))) {
            // gate|up and lm_head are the register-resident kernel's at every row count; qkv and wo where it measured ahead on a fast AND
            // on a slow box (profiles/r04_labs/README.md: its per-workgroup copy of the rows rides the L2 -> CU path, the part of the chip
            // that differs most between boxes): wo up to 16 and from 33 rows (qkv: below).  Both kinds of producer leave x / h weighted AND plain.
            // the weighted rows travel in fragment order from 9 rows (same-box A/B, profiles/r04_labs/README.md: 16 / 32 / 64 sequences
            // -1.5 / -4 / -1.3 % per step; at 8 sequences +2 %: row-major there)
            // (round 6: from 5 rows wherever the row-streaming matmul takes the layer's gate|up -- it reads nothing else)
            const bool frag = rows_travel_in_fragment_order(e, batch);
            // qkv on this kernel at every row count since round 5 (rows in fragment order from 9 rows): same-box A/B at 128-token contexts,
            // two alternating rounds, 24 / 32 / 48 / 64 sequences 1.91 / 1.95 / 2.56 / 2.69 -> 1.87 / 1.89 / 2.50 / 2.59 ms per step
            // (profiles/r05_labs/batched_qkv_on_qmm6_ab.log; round 4 had measured -1 ... -2.9 % on a fast box and left 17-64 rows on the sliced
            // matmul, whose slices the attention kernel adds -- that route is now the one behind option "qmm6" = 0 only)
            const bool qkv6 = qmm6_takes(e, w.wqkv, batch);
            // wo on the register-resident kernel at EVERY row count (round 6: its planner now deals 16-row blocks to more workgroups where the rows are long --
            // 17-32 rows 6.9 us against 8.4-9.1 for the sliced matmul + reduction that took them until then, 33-48 rows 10.1 -> 7.1; qmm6.h, qmm6_plan)
            const bool wo6 = wo6_ok;
            // this layer's hand-over buffers: the shared ones, or -- the AQL route's per-layer mode -- its own (written once per step)
            uint16_t *hb = e->h, *hwb = e->xn, *qkvb = e->qkv, *attnb = e->attn, *actb = e->act, *x_out = e->x, *xw_out = e->xn;
            float *sshb = e->ss_h, *ssx_out = e->ss_x;
            float *const ws_shared = e->attn_ws;
            if (per_layer_b) {
                const tl_engine::LayerAct &la = e->layer_act[l];
                hb = la.h, hwb = la.xn, qkvb = la.qkv, attnb = la.attn, actb = la.act, x_out = la.x_out, xw_out = la.xw, sshb = la.ss_h, ssx_out = la.ss_x_out;
                e->attn_ws = la.attn_ws;  // engine_attention reads the member
            }
            auto planes = [&](int which) {  // the K-sliced matmul's fp32 planes of the NEXT engine_linear call
                e->planes_now = per_layer_b ? e->layer_act[l].planes[which] : nullptr;
                e->planes_now_bytes = per_layer_b ? e->layer_plane_bytes[which] : 0;
            };
            auto run_layer_b = [&]() -> int {
            KeptPartials parts;
            if (qkv6) {
                if (!xw) {
                    const long n8 = (long)batch * c.hidden_size / 8;
                    hipLaunchKernelGGL(weight_rows_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, e->stream, bx, (const uint16_t *)w.input_norm_dev,
                                       e->xn, n8, c.hidden_size / 8, frag ? 1 : 0);
                    bxw = e->xn;
                }
                TL_TRY(engine_qmm6(e, w.wqkv, bxw, qkvb, batch, EPI_STORE, nullptr, pc, 0, bssx, x_ss, nullptr, nullptr, nullptr, nullptr, frag));
            } else {
                const bool keep_qkv = e->attn_qkv_partials && attn_takes_qkv_partials(c.head_dim, sp.rq);
                TL_TRY(engine_linear(e, w.wqkv, bx, qkvb, batch, PRO_RMSNORM, EPI_STORE, w.input_norm_dev, nullptr, pc, 0, bssx, nullptr, nullptr,
                                     keep_qkv ? &parts : nullptr, x_ss));
            }
            bool merged = false;
            TL_TRY(engine_attention(e, qkvb, w.q_norm_dev, w.k_norm_dev, e->layer_k(l), e->layer_v(l), attnb, batch, sp, pc, &parts, &w.wo, &merged, e->layer_ks(l), e->layer_vs(l)));
            TL_REQUIRE(!merged, "engine: a batched step left its attention windows unmerged");
            int h_ss = 0;
            if (wo6) TL_TRY(engine_qmm6(e, w.wo, attnb, hb, batch, EPI_RESIDUAL, bx, pc, 1, nullptr, 0, sshb, &h_ss, w.post_norm_dev, hwb, frag));
            else {
                planes(0);
                TL_TRY(engine_linear(e, w.wo, attnb, hb, batch, PRO_NONE, EPI_RESIDUAL, nullptr, bx, pc, 1, nullptr, sshb, nullptr, nullptr, QM3_SS, &h_ss,
                                     w.post_norm_dev, hwb, frag));
            }
            TL_REQUIRE(h_ss > 0 && qmm3_takes_ss(h_ss), "engine: the wo projection left no sums of squares for its weighted rows");
            TL_TRY(engine_qmm6(e, w.wgu, hwb, actb, batch, EPI_SWIGLU, nullptr, pc, 2, sshb, h_ss, nullptr, nullptr, nullptr, nullptr, frag));
            // the rows w_down leaves are weighted for their next reader: the next layer's input norm, or the final norm ahead of lm_head
            const void *next_norm = l + 1 < c.num_layers ? e->layers[l + 1].input_norm_dev : e->final_norm;
            // w_down: 76 groups against 160 tiles -- every workgroup of the register-resident kernel would pull 311 KB of rows for ONE tile
            // (measured 9.0 us at 8 rows, 18.9 at 64, against 6.5 / 13.0 for the K-sliced matmul + reduction): the sliced kernel keeps
            // it wherever its plan exists, and its reduction leaves the weighted rows
            planes(1);
            if (sliced_leaves_weighted(w.wdown)) {
                TL_TRY(engine_linear(e, w.wdown, actb, x_out, batch, PRO_NONE, EPI_RESIDUAL, nullptr, hb, pc, 3, nullptr, ssx_out, nullptr, nullptr,
                                     QM3_SS, &x_ss, next_norm, xw_out, frag));
                xw = x_ss > 0;
            } else if (qmm6_takes(e, w.wdown, batch) && qmm3_takes_ss(w.wdown.rows / 16)) {
                TL_TRY(engine_qmm6(e, w.wdown, actb, x_out, batch, EPI_RESIDUAL, hb, pc, 3, nullptr, 0, ssx_out, &x_ss, next_norm, xw_out, frag));
                xw = true;
            } else {
                TL_TRY(engine_linear(e, w.wdown, actb, x_out, batch, PRO_NONE, EPI_RESIDUAL, nullptr, hb, pc, 3, nullptr, ssx_out, nullptr, nullptr, QM3_SS, &x_ss));
                xw = false;
            }
            return TL_OK;
            };
            const int rc_b = run_layer_b();
            e->attn_ws = ws_shared;
            e->planes_now = nullptr, e->planes_now_bytes = 0;
            TL_TRY(rc_b);
            bx = x_out, bxw = xw_out, bssx = ssx_out;
            x_cur = bx, ssx_cur = bssx;
            continue;
        }
        // (a layer outside the batched branch reads and writes the shared buffers: a per-layer batched step has none -- per_layer_b above.
        // The pre-check restates the branch's conditions; should the two ever diverge -- the row's sums of squares missing, say -- the step
        // must not be captured as "written once": it would be replayed without cache maintenance over buffers written several times)
        TL_REQUIRE(!per_layer_b, "engine: a layer fell out of the batched branch of a step planned on the per-layer buffers");
        xw = false;
        // this layer's hand-over buffers: the shared ones, or -- per-layer mode -- its own (written once per step)
        uint16_t *x_in = x_cur, *x_out = e->x, *hb = e->h, *xnb = e->xn, *qkvb = e->qkv, *attnb = e->attn, *actb = e->act;
        float *ssx_in = ssx_cur, *ssx_out = e->ss_x, *sshb = e->ss_h;
        float *const ws_shared = e->attn_ws;
        if (per_layer) {
            const tl_engine::LayerAct &la = e->layer_act[l];
            x_out = la.x_out, hb = la.h, xnb = la.xn, qkvb = la.qkv, attnb = la.attn, actb = la.act, ssx_out = la.ss_x_out, sshb = la.ss_h;
            e->attn_ws = la.attn_ws;  // engine_attention / engine_wo_merge read the member
        }
        auto restore_ws = [&]() { e->attn_ws = ws_shared; };
        auto run_layer = [&]() -> int {
        KeptPartials qkv_parts;
        const bool keep_qkv = e->attn_qkv_partials && attn_takes_qkv_partials(c.head_dim, sp.rq);
        TL_TRY(engine_linear(e, w.wqkv, x_in, qkvb, batch, PRO_RMSNORM, EPI_STORE, w.input_norm_dev, nullptr, pc, 0,
                             x_ss ? ssx_in : nullptr, nullptr, nullptr, keep_qkv ? &qkv_parts : nullptr, x_ss));
        bool merge_left = false;
        TL_TRY(engine_attention(e, qkvb, w.q_norm_dev, w.k_norm_dev, e->layer_k(l), e->layer_v(l), attnb, batch, sp, pc, &qkv_parts,
                                &w.wo, &merge_left, e->layer_ks(l), e->layer_vs(l)));
        int h_ss = 0;
        if (e->is_moe(l)) {  // wo + residual, then the MoE MLP as its own launches (no producer-side sums for the next layer)
            if (merge_left) TL_TRY(engine_wo_merge(e, w.wo, x_in, hb, sp.n_splits, pc, nullptr, nullptr));
            else TL_TRY(engine_linear(e, w.wo, attnb, hb, batch, PRO_NONE, EPI_RESIDUAL, nullptr, x_in, pc, 1));
            TL_TRY(tl_rms_norm(hb, w.post_norm_dev, xnb, batch, c.hidden_size, c.rms_norm_eps, TL_BF16, e->stream));
            TL_TRY(engine_moe_mlp(e, l, xnb, hb, x_out, batch, pc));
            x_ss = 0;
            return TL_OK;
        }
        TL_REQUIRE(w.wgu.weight_dev != nullptr, "engine: a layer has neither a dense MLP nor experts (tl_engine_set_moe_layer)");
        // h leaves the wo GEMV twice when the gate|up GEMV can take it weighted: as the residual stream and, in xn, times the
        // post-attention norm weight
        const bool weighted = weighted_rows_apply(e, w.wo, w.wgu, batch);
        uint16_t *hw = weighted ? xnb : nullptr;
        if (merge_left) TL_TRY(engine_wo_merge(e, w.wo, x_in, hb, sp.n_splits, pc, sshb, &h_ss, w.post_norm_dev, hw));
        else TL_TRY(engine_linear(e, w.wo, attnb, hb, batch, PRO_NONE, EPI_RESIDUAL, nullptr, x_in, pc, 1, nullptr, sshb, nullptr, nullptr, QM3_SS, &h_ss, w.post_norm_dev, hw));
        if (weighted) {
            TL_REQUIRE(h_ss > 0, "engine: the wo GEMV left no sums of squares for its weighted rows");
            TL_TRY(engine_linear(e, w.wgu, xnb, actb, batch, PRO_RMS_WEIGHTED, EPI_SWIGLU, nullptr, nullptr, pc, 2, sshb, nullptr, nullptr, nullptr, h_ss));
        } else
        TL_TRY(engine_linear(e, w.wgu, hb, actb, batch, PRO_RMSNORM, EPI_SWIGLU, w.post_norm_dev, nullptr, pc, 2,
                             h_ss ? sshb : nullptr, nullptr, nullptr, nullptr, h_ss));
        TL_TRY(engine_linear(e, w.wdown, actb, x_out, batch, PRO_NONE, EPI_RESIDUAL, nullptr, hb, pc, 3, nullptr, ssx_out, nullptr, nullptr, QM3_SS, &x_ss));
        return TL_OK;
        };
        const int layer_rc = run_layer();
        restore_ws();
        TL_TRY(layer_rc);
        x_cur = x_out;
        ssx_cur = ssx_out;
    }
    e->want_tile_max = e->lm_tile_max_on;
    e->tile_max_rows = 0;
    const int head_rc = xw && qmm6_takes(e, e->head(), batch) && qmm3_takes_ss(x_ss)
                            ? engine_qmm6(e, e->head(), bxw, e->logits, batch, EPI_STORE, nullptr, pc, 4, bssx, x_ss, nullptr, nullptr, nullptr, nullptr, rows_travel_in_fragment_order(e, batch))
                            : engine_linear(e, e->head(), x_cur, e->logits, batch, PRO_RMSNORM, EPI_STORE, e->final_norm, nullptr, pc, 4,
                                            x_ss ? ssx_cur : nullptr, nullptr, nullptr, nullptr, x_ss);
    e->want_tile_max = false;
    TL_TRY(head_rc);
    StepEndArgs s{};
    if (e->tile_max_rows == batch) s.tile_max = e->lm_tile_max, s.tiles = e->head().rows / 16;
    s.logits = e->logits;
    s.vocab = c.vocab_size;
    s.slot0 = 0;
    s.tokens = e->tokens;
    s.context_lens = e->context_lens;
    s.live = e->live;
    s.produced = e->produced;
    s.ring = e->ring;
    s.ring_cap = e->ring_cap;
    s.advance = 1;
    s.emb_w = e->embed.weight_dev;
    s.emb_s = (const uint16_t *)e->embed.scales_dev;
    s.emb_b = (const uint16_t *)e->embed.biases_dev;
    s.x = e->x;
    s.hidden = c.hidden_size;
    s.rope_table = e->rope_table;
    s.rope_cur = e->rope_cur;
    s.rope_positions = e->rope_positions;
    s.rope_half = c.head_dim / 2;
    s.ss_out = e->ss_x;
    s.prof = pc ? pc->buf : nullptr;
    hipLaunchKernelGGL(step_end_kernel, dim3(batch), dim3(1024), 0, e->stream, s);
    if (pc) prof_after(e, pc, 7, batch);
    TL_CHECK_LAUNCH("engine step end");
    return TL_OK;
}

static void prof_after(tl_engine *e, ProfCtx *pc, int kind, int n_wg) {
    const int idx = (int)pc->kinds.size();
    if (idx >= pc->cap) return;
    hipLaunchKernelGGL(prof_reduce_kernel, dim3(1), dim3(1024), 0, e->stream, pc->buf, n_wg, pc->pairs + 2 * (size_t)idx);
    pc->kinds.push_back(kind);
    if (e->check.on)  // tl_engine_check_step: what this launch (and any unstamped one ahead of it) stored, against "written once per step"
        for (int rg = 0; rg < 2; ++rg)
            if (e->check.bytes[rg])
                hipLaunchKernelGGL(written_once_check_kernel, dim3(1024), dim3(256), 0, e->stream, (const uint32_t *)e->check.region[rg], e->check.shadow[rg],
                                   e->check.written[rg], e->check.bytes[rg] / 4, idx, rg, e->check.report);
}

// ---- page ownership: a page is shared by every sequence forked from a common prefix and returns to the free list when
// the last holder lets go of it
static int take_page(tl_engine *e) {
    const int id = e->free_pages.back();
    e->free_pages.pop_back();
    e->page_refs[id] = 1;
    e->stats.page_allocations++;
    if (e->page_was_used[id]) e->stats.reused_page_allocations++;
    e->page_was_used[id] = 1;
    e->stats.pages_in_use++;
    e->stats.peak_pages_in_use = std::max(e->stats.peak_pages_in_use, e->stats.pages_in_use);
    return id;
}
static void drop_page(tl_engine *e, int id) {
    if (--e->page_refs[id] == 0) {
        e->free_pages.push_back(id);
        e->stats.pages_in_use--;
    }
}
// K and V rows of one page, every layer (device to device, stream ordered)
static int copy_page(tl_engine *e, int from, int to) {
    const tl_engine_config &c = e->cfg;
    const size_t page_bytes = (size_t)c.num_kv_heads * c.page_size * c.head_dim * e->kv_elem_bytes();
    const size_t page_rows = (size_t)c.num_kv_heads * c.page_size;  // FP8 pages: one scale per row
    for (int l = 0; l < c.num_layers; ++l) {
        TL_HIP(hipMemcpyAsync((char *)e->layer_k(l) + (size_t)to * page_bytes, (char *)e->layer_k(l) + (size_t)from * page_bytes, page_bytes,
                              hipMemcpyDeviceToDevice, e->stream));
        TL_HIP(hipMemcpyAsync((char *)e->layer_v(l) + (size_t)to * page_bytes, (char *)e->layer_v(l) + (size_t)from * page_bytes, page_bytes,
                              hipMemcpyDeviceToDevice, e->stream));
        if (e->kv_format == TL_KV_FP8_E4M3) {
            TL_HIP(hipMemcpyAsync(e->layer_ks(l) + (size_t)to * page_rows, e->layer_ks(l) + (size_t)from * page_rows, page_rows * 4,
                                  hipMemcpyDeviceToDevice, e->stream));
            TL_HIP(hipMemcpyAsync(e->layer_vs(l) + (size_t)to * page_rows, e->layer_vs(l) + (size_t)from * page_rows, page_rows * 4,
                                  hipMemcpyDeviceToDevice, e->stream));
        }
    }
    return TL_OK;
}

static int reserve_locked(tl_engine *e, int slot, int total_tokens,
                          std::vector<std::pair<int32_t *, int32_t>> &pokes) {
    const tl_engine_config &c = e->cfg;
    const int need = (total_tokens + c.page_size - 1) / c.page_size;
    auto &pages = e->slot_pages[slot];
    const int have = (int)pages.size();
    if (need <= have) return TL_OK;
    if (need > c.max_pages_per_seq)
        return fail(TL_ERR_INVALID, "engine: sequence exceeds max_pages_per_seq * page_size tokens");
    if (need - have > (int)e->free_pages.size())
        return fail(TL_ERR_INVALID, "engine: KV page pool exhausted");
    for (int j = have; j < need; ++j) {
        const int id = take_page(e);
        pages.push_back(id);
        pokes.emplace_back(e->block_table + (size_t)slot * c.max_pages_per_seq + j, id);
    }
    return TL_OK;
}

// Pages for ONE more token in every live slot of [0, batch): all or nothing.  The totals are checked before anything is
// mutated, so a failing step leaves the host mirrors and the device block table exactly as they were (a half-applied
// reservation would leave a slot that owns a page on the host and -1 on the device: silently dropped K/V).
static int reserve_step_locked(tl_engine *e, int batch, std::vector<std::pair<int32_t *, int32_t>> &pokes, int *max_ctx) {
    const tl_engine_config &c = e->cfg;
    size_t extra = 0;
    for (int b = 0; b < batch; ++b) {
        if (!e->slot_live[b]) continue;
        const int need = (e->slot_ctx[b] + 1 + c.page_size - 1) / c.page_size;
        if (need > c.max_pages_per_seq)
            return fail(TL_ERR_INVALID, "engine: sequence exceeds max_pages_per_seq * page_size tokens");
        if (need > (int)e->slot_pages[b].size()) extra += (size_t)need - e->slot_pages[b].size();
    }
    if (extra > e->free_pages.size()) return fail(TL_ERR_INVALID, "engine: KV page pool exhausted");
    for (int b = 0; b < batch; ++b) {
        if (!e->slot_live[b]) continue;
        TL_TRY(reserve_locked(e, b, e->slot_ctx[b] + 1, pokes));  // cannot fail after the checks above
        *max_ctx = std::max(*max_ctx, e->slot_ctx[b] + 1);
    }
    return TL_OK;
}

static int slot_check(const tl_engine *e, int slot, bool must_be_live) {
    if (!e) return fail(TL_ERR_INVALID, "engine: null engine");
    if (slot < 0 || slot >= e->cfg.max_batch) return fail(TL_ERR_INVALID, "engine: slot out of range");
    if (must_be_live && !e->slot_live[slot]) return fail(TL_ERR_INVALID, "engine: slot holds no sequence");
    return TL_OK;
}

}  // namespace tl

// ================================================================================================
extern "C" int tl_engine_create(const tl_engine_config *cfg, const tl_layer_weights *layers, const tl_w4 *embed,
                                const void *final_norm_dev, const tl_w4 *lm_head, void *stream, tl_engine **out) {
    return tl_engine_create_kv(cfg, layers, embed, final_norm_dev, lm_head, stream, TL_KV_BF16, out);
}

extern "C" int tl_engine_create_kv(const tl_engine_config *cfg, const tl_layer_weights *layers, const tl_w4 *embed,
                                   const void *final_norm_dev, const tl_w4 *lm_head, void *stream, int kv_format, tl_engine **out) {
    TL_REQUIRE(cfg && layers && embed && final_norm_dev && out, "engine_create: null argument");
    const tl_engine_config &c = *cfg;
    TL_REQUIRE(kv_format == TL_KV_BF16 || kv_format == TL_KV_FP8_E4M3, "engine_create: kv_format must be TL_KV_BF16 or TL_KV_FP8_E4M3");
    TL_REQUIRE(kv_format == TL_KV_BF16 || c.head_dim == 128, "engine_create: FP8 KV pages need head_dim 128");
    TL_REQUIRE(c.num_layers > 0 && c.hidden_size > 0 && c.hidden_size % 128 == 0, "engine_create: hidden_size must be a positive multiple of 128");
    TL_REQUIRE(c.num_heads > 0 && c.num_kv_heads > 0 && c.num_heads % c.num_kv_heads == 0,
               "engine_create: num_heads must be divisible by num_kv_heads");
    TL_REQUIRE(c.head_dim == 32 || c.head_dim == 64 || c.head_dim == 128, "engine_create: head_dim must be 32, 64 or 128");
    TL_REQUIRE((c.num_heads * c.head_dim) % 128 == 0 && c.intermediate_size % 128 == 0,
               "engine_create: projection widths must be multiples of the quantization group (128)");
    TL_REQUIRE(c.page_size > 0 && c.num_pages > 0 && c.max_batch > 0 && c.max_batch <= 256 && c.max_pages_per_seq > 0,
               "engine_create: need page_size, num_pages, max_pages_per_seq > 0 and 1 <= max_batch <= 256");
    TL_REQUIRE(c.max_prefill_rows > 0, "engine_create: max_prefill_rows must be positive");
    const int qkv_dim = (c.num_heads + 2 * c.num_kv_heads) * c.head_dim;
    const int q_dim = c.num_heads * c.head_dim;
    for (int l = 0; l < c.num_layers; ++l) {
        TL_TRY(check_w4(layers[l].wqkv, qkv_dim, c.hidden_size, "wqkv"));
        TL_TRY(check_w4(layers[l].wo, c.hidden_size, q_dim, "wo"));
        // a layer without a dense MLP (both null) must get its experts through tl_engine_set_moe_layer before the first step
        if (layers[l].wgu.weight_dev != nullptr || layers[l].wdown.weight_dev != nullptr) {
            TL_TRY(check_w4(layers[l].wgu, 2 * c.intermediate_size, c.hidden_size, "wgu"));
            TL_TRY(check_w4(layers[l].wdown, c.hidden_size, c.intermediate_size, "wdown"));
        }
        TL_REQUIRE(layers[l].input_norm_dev && layers[l].post_norm_dev && layers[l].q_norm_dev && layers[l].k_norm_dev,
                   "engine_create: null norm weight");
    }
    TL_TRY(check_w4(*embed, c.vocab_size, c.hidden_size, "embed_tokens"));
    if (lm_head) TL_TRY(check_w4(*lm_head, c.vocab_size, c.hidden_size, "lm_head"));

    auto *e = new tl_engine();
    e->cfg = c;
    e->kv_format = kv_format;
    if (hipGetDevice(&e->device) != hipSuccess) {
        delete e;
        return fail(TL_ERR_HIP, "engine_create: hipGetDevice failed");
    }
    e->layers.assign(layers, layers + c.num_layers);
    e->embed = *embed;
    if (lm_head) e->lm_head = *lm_head;
    e->final_norm = final_norm_dev;
    e->stream = (hipStream_t)stream;
    if (!e->stream) {
        // The legacy default stream cannot be captured into a graph: own a non-blocking stream instead, and
        // make sure everything the caller enqueued before (weight uploads / re-packing) has finished.
        if (hipDeviceSynchronize() != hipSuccess ||
            hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
            delete e;
            return fail(TL_ERR_HIP, "engine_create: could not create the engine stream");
        }
        e->owns_stream = true;
    }
    e->rows_cap = std::max(c.max_prefill_rows, c.max_batch);

    // ---- arena layout
    const size_t R = (size_t)e->rows_cap;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t o_bt = carve((size_t)c.max_batch * c.max_pages_per_seq * 4);
    const size_t o_ctx = carve((size_t)c.max_batch * 4);
    const size_t o_tok = carve((size_t)c.max_batch * 4);
    const size_t o_live = carve((size_t)c.max_batch * 4);
    const size_t o_prod = carve((size_t)c.max_batch * 4);
    const size_t o_ring = carve((size_t)c.max_batch * e->ring_cap * 4);
    const size_t o_sctx = carve(64);
    const size_t o_ptok = carve(R * 4);
    const size_t o_x = carve(R * c.hidden_size * 2);
    e->arena_act_off = o_x;
    const size_t o_h = carve(R * c.hidden_size * 2);
    const size_t o_xn = carve((R + 15) / 16 * 16 * c.hidden_size * 2);  // weighted rows of a batched step lie in 16-row fragment blocks (qmm6.h)
    const size_t o_tmp = carve(R * c.hidden_size * 2);
    const size_t o_qkv = carve(R * qkv_dim * 2);
    const size_t o_qt = carve(R * q_dim * 2);
    const size_t o_at = carve(R * q_dim * 2);
    const size_t o_attn = carve(R * q_dim * 2);
    const size_t o_gu = carve(R * 2 * c.intermediate_size * 2);
    const size_t o_act = carve(R * c.intermediate_size * 2);
    const size_t o_log = carve((size_t)std::max(c.max_batch, 8) * c.vocab_size * 2);  // decode rows, or 8 verification rows
    const size_t o_vid = carve(8 * 4);
    const size_t o_tmax = carve((size_t)8 * (c.vocab_size / 16 + 1) * sizeof(f32x2));
    // per-row partial sums of squares: QM3_SS per row from the skinny-matmul reduction / the embedding kernels, one per 16-row
    // tile (hidden / 16) from the 1-4-row GEMVs
    const size_t ss_per_row = (size_t)std::max(QM3_SS, c.hidden_size / 16 + 1);
    const size_t o_ssx = carve((size_t)c.max_batch * ss_per_row * 4);
    const size_t o_ssh = carve((size_t)c.max_batch * ss_per_row * 4);
    // attention partials: decode (batch*Hq rows x 64 splits) or the L<=8 operator path during short prefills
    // decode partials: at most 64 splits per row with many sequences, at most 256 split-rows per head with few (pick_decode_splits)
    const size_t ws_row = (size_t)c.head_dim + ATTN_WS_PAD;
    e->attn_ws_bytes = std::max((size_t)std::max(c.max_batch * 64, 4 * 256) * c.num_heads * ws_row * 4, (size_t)c.num_heads * 8 * 64 * ws_row * 4);
    for (int L = 1; L <= c.max_prefill_rows; ++L)  // the paged attention operator may split the context for any chunk length
        e->attn_ws_bytes = std::max(e->attn_ws_bytes, tl_paged_attention_workspace_bytes(c.num_heads, L, c.head_dim, c.page_size,
                                                                                         c.max_pages_per_seq, c.num_heads,
                                                                                         c.num_kv_heads, 0));
    const size_t o_ws = carve(e->attn_ws_bytes);
    e->arena_bytes = off;

    auto cleanup_fail = [&](const std::string &msg) {
        if (e->arena) (void)hipFree(e->arena);
        if (e->kpool) (void)hipFree(e->kpool);
        if (e->vpool) (void)hipFree(e->vpool);
        if (e->kscale_pool) (void)hipFree(e->kscale_pool);
        if (e->vscale_pool) (void)hipFree(e->vscale_pool);
        if (e->rope_table) (void)hipFree(e->rope_table);
        if (e->rope_cur) (void)hipFree(e->rope_cur);
        if (e->splitk_ws) (void)hipFree(e->splitk_ws);
        for (auto &kv : e->tiled) {
            (void)hipFree(kv.second.wt);
            (void)hipFree(kv.second.sbt);
        }
        for (auto &kv : e->bf16w) (void)hipFree(kv.second);
        if (e->owns_stream) (void)hipStreamDestroy(e->stream);
        delete e;
        return fail(TL_ERR_HIP, msg);
    };
    if (hipMalloc((void **)&e->arena, e->arena_bytes) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(arena) failed");
    e->layer_pool_elems = (size_t)c.num_pages * c.num_kv_heads * c.page_size * c.head_dim;
    const size_t pool_bytes = e->layer_pool_elems * e->kv_elem_bytes() * c.num_layers;
    if (hipMalloc((void **)&e->kpool, pool_bytes) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(key pages) failed");
    if (hipMalloc((void **)&e->vpool, pool_bytes) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(value pages) failed");
    e->kv_bytes = 2 * pool_bytes;
    if (kv_format == TL_KV_FP8_E4M3) {
        // one float32 scale per (page, kv head, slot) row; zero like the codes (a zero scale times a zero code is the zero the bf16 pool holds)
        e->layer_scale_elems = (size_t)c.num_pages * c.num_kv_heads * c.page_size;
        const size_t scale_bytes = e->layer_scale_elems * 4 * c.num_layers;
        if (hipMalloc((void **)&e->kscale_pool, scale_bytes) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(key scales) failed");
        if (hipMalloc((void **)&e->vscale_pool, scale_bytes) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(value scales) failed");
        if (hipMemsetAsync(e->kscale_pool, 0, scale_bytes, e->stream) != hipSuccess ||
            hipMemsetAsync(e->vscale_pool, 0, scale_bytes, e->stream) != hipSuccess)
            return cleanup_fail("engine_create: memset(KV scales) failed");
        e->kv_bytes += 2 * scale_bytes;
    }
    // Masked (out-of-context) token slots are still loaded and multiplied by a zero weight in the decode kernel,
    // so the pools must never hold NaN/Inf bit patterns: start from zeros (kernels only ever write finite values).
    if (hipMemsetAsync(e->kpool, 0, pool_bytes, e->stream) != hipSuccess ||
        hipMemsetAsync(e->vpool, 0, pool_bytes, e->stream) != hipSuccess)
        return cleanup_fail("engine_create: memset(KV pools) failed");

    char *A = e->arena;
    e->block_table = (int32_t *)(A + o_bt);
    e->context_lens = (int32_t *)(A + o_ctx);
    e->tokens = (int32_t *)(A + o_tok);
    e->live = (int32_t *)(A + o_live);
    e->produced = (int32_t *)(A + o_prod);
    e->ring = (int32_t *)(A + o_ring);
    e->scratch_ctx = (int32_t *)(A + o_sctx);
    e->prefill_tokens = (int32_t *)(A + o_ptok);
    e->x = (uint16_t *)(A + o_x);
    e->h = (uint16_t *)(A + o_h);
    e->xn = (uint16_t *)(A + o_xn);
    e->tmp = (uint16_t *)(A + o_tmp);
    e->qkv = (uint16_t *)(A + o_qkv);
    e->q_t = (uint16_t *)(A + o_qt);
    e->attn_t = (uint16_t *)(A + o_at);
    e->attn = (uint16_t *)(A + o_attn);
    e->gu = (uint16_t *)(A + o_gu);
    e->act = (uint16_t *)(A + o_act);
    e->logits = (uint16_t *)(A + o_log);
    e->attn_ws = (float *)(A + o_ws);
    e->verify_ids = (int32_t *)(A + o_vid);
    e->lm_tile_max = (f32x2 *)(A + o_tmax);
    e->ss_x = (float *)(A + o_ssx);
    e->ss_h = (float *)(A + o_ssh);
    if (const char *q = getenv("TL_QMM3_MIN_M")) e->qmm3_min_rows = std::max(1, atoi(q));
    read_attention_knobs(e);
    // Decode steps replay as AQL packets on the engine's own HSA queue (aql.h) unless TL_AQL=0: the same captured step, without the
    // cache maintenance HIP puts between its launches (0.987 -> 0.929 ms per token at Qwen3-4B, logits bit-identical).  Where the route is
    // not available (no code objects next to the library, no HSA agent for the device) the engine stays on hipGraphLaunch and says why
    // in tl_engine_replay_route(); TL_AQL=1 makes that an error instead.
    const char *aql_env = getenv("TL_AQL");
    if (aql_env == nullptr || atoi(aql_env) != 0) {
        {
            AqlRuntime &rt = AqlRuntime::for_device(e->device);
            e->aql_rt = &rt;
            std::string why;
            e->aql_queue = std::make_unique<AqlQueue>();
            if (!rt.ensure_loaded(library_dir()) || !e->aql_queue->create(rt, why)) {
                e->aql_why = rt.ok() ? why : rt.why();
                e->aql_queue.reset();
                if (aql_env != nullptr) {
                    const std::string msg = "engine_create: TL_AQL=1 but the AQL route is not available: " + e->aql_why;
                    tl_engine_destroy(e);
                    return fail(TL_ERR_UNSUPPORTED, msg);
                }
            }
        }
        if (e->aql_queue) {
            e->aql_on = true;
            // the route's code objects are compiled with TL_COHERENT (common.h): no cache maintenance between the launches of a step.
            // (tl_engine_set_option "aql_fences" = 1 puts HIP's agent-scope fences back on every packet: the A/B of what the maintenance costs)
            e->aql_fences.inner_acquire = e->aql_fences.inner_release = HSA_FENCE_SCOPE_NONE;
            // per-layer decode activations (1-4 rows: the fused-GEMV step; 5-64 rows since round 5: the batched-matmul step): every hand-over
            // address of a step is written once per step.  The attention partials hold 16 windows per row (at most 1,024 row-windows): a plan
            // beyond that keeps the shared buffers and the graph route.
            {
                const int rows = std::min(c.max_batch, 64);
                const size_t ssr = (size_t)std::max(QM3_SS, c.hidden_size / 16 + 1);
                // (weighted rows lie in 16-row fragment blocks, qmm6.h: a partly filled last block spans all 16 row slots -- sized like the shared xn;
                // with `rows` itself a batch of 17-24 on a 24-slot engine wrote its block past xn into xw and past xw into qkv: dead bytes at that
                // moment by the order of the launches, but a second write per step to addresses the replay route reads without cache maintenance)
                const size_t b_xf = align_up((size_t)((rows + 15) / 16 * 16) * c.hidden_size * 2, 256);
                const size_t b_x = align_up((size_t)rows * c.hidden_size * 2, 256), b_qkv = align_up((size_t)rows * qkv_dim * 2, 256),
                             b_attn = align_up((size_t)rows * q_dim * 2, 256), b_act = align_up((size_t)rows * c.intermediate_size * 2, 256),
                             b_ss = align_up((size_t)rows * ssr * 4, 256),
                             b_ws = align_up((size_t)std::min(rows * 64, std::max(4 * 64, std::min(rows * 16, 1024))) * c.num_heads * ws_row * 4, 256);
                // slice planes of the sliced matmuls a batched step can take (wo, w_down), the largest over 5 .. rows rows
                size_t b_pl[2] = {0, 0};
                for (int M = std::min(5, e->qmm3_min_rows); M <= rows; ++M) {
                    const Qmm3Plan pw = qmm3_plan(M, q_dim, c.hidden_size, -1), pd = qmm3_plan(M, c.intermediate_size, c.hidden_size, -1);
                    if (pw.ok) b_pl[0] = std::max(b_pl[0], align_up(pw.partial_bytes, 256));
                    if (pd.ok) b_pl[1] = std::max(b_pl[1], align_up(pd.partial_bytes, 256));
                }
                const size_t per_layer = 2 * b_x + 2 * b_xf + b_qkv + b_attn + b_act + 2 * b_ss + b_ws + b_pl[0] + b_pl[1];
                if (hipMalloc((void **)&e->layer_act_mem, per_layer * c.num_layers) != hipSuccess ||
                    hipMemsetAsync(e->layer_act_mem, 0, per_layer * c.num_layers, e->stream) != hipSuccess) {
                    tl_engine_destroy(e);
                    return fail(TL_ERR_HIP, "engine_create: hipMalloc(per-layer decode activations) failed");
                }
                e->layer_act.resize(c.num_layers);
                for (int l = 0; l < c.num_layers; ++l) {
                    char *m = e->layer_act_mem + (size_t)l * per_layer;
                    tl_engine::LayerAct &a = e->layer_act[l];
                    a.x_out = (uint16_t *)m, m += b_x;
                    a.h = (uint16_t *)m, m += b_x;
                    a.xn = (uint16_t *)m, m += b_xf;
                    a.xw = (uint16_t *)m, m += b_xf;
                    a.qkv = (uint16_t *)m, m += b_qkv;
                    a.attn = (uint16_t *)m, m += b_attn;
                    a.act = (uint16_t *)m, m += b_act;
                    a.ss_x_out = (float *)m, m += b_ss;
                    a.ss_h = (float *)m, m += b_ss;
                    a.attn_ws = (float *)m, m += b_ws;
                    a.planes[0] = (float *)m, m += b_pl[0];
                    a.planes[1] = (float *)m;
                }
                e->layer_plane_bytes[0] = b_pl[0], e->layer_plane_bytes[1] = b_pl[1];
                e->layer_act_bytes = per_layer * c.num_layers;
                e->layer_act_rows = rows;
                e->layer_ws_bytes = b_ws;
            }
        }
    }

    // state words: zero everything up to the activations, then the block table to -1
    if (hipMemsetAsync(e->arena, 0, o_x, e->stream) != hipSuccess) return cleanup_fail("engine_create: memset failed");
    const int bt_n = c.max_batch * c.max_pages_per_seq;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(ceil_div(bt_n, 256)), dim3(256), 0, e->stream, e->block_table, -1, bt_n);
    if (hipGetLastError() != hipSuccess) return cleanup_fail("engine_create: block-table init failed");

    // RoPE table for every position a sequence can reach
    {
        const long max_pos = std::min<long>((long)c.max_pages_per_seq * c.page_size + 1, 1 << 20);
        e->rope_positions = (int)max_pos;
        const int half = c.head_dim / 2;
        if (hipMalloc((void **)&e->rope_table, (size_t)max_pos * half * sizeof(float2)) != hipSuccess)
            return cleanup_fail("engine_create: hipMalloc(rope table) failed");
        hipLaunchKernelGGL(rope_table_kernel, dim3(ceil_div(max_pos * half, 256)), dim3(256), 0, e->stream, e->rope_table,
                           (int)max_pos, half, c.rope_theta);
        if (hipGetLastError() != hipSuccess) return cleanup_fail("engine_create: rope table kernel failed");
        if (hipMalloc((void **)&e->rope_cur, (size_t)c.max_batch * half * sizeof(float2)) != hipSuccess ||
            hipMemsetAsync(e->rope_cur, 0, (size_t)c.max_batch * half * sizeof(float2), e->stream) != hipSuccess)
            return cleanup_fail("engine_create: hipMalloc(rope state) failed");
    }

    // decode-path weight copies in the tiled MFMA layout
    {
        auto add_tiled = [&](const tl_w4 &w) -> bool {
            if (w.rows % 16 != 0 || w.cols % 128 != 0 || e->tiled.count(w.weight_dev)) return true;
            tl_engine::Tiled t;
            const size_t wbytes = (size_t)w.rows * w.cols / 2, sbytes = (size_t)w.rows * (w.cols / 128) * 4;
            if (hipMalloc((void **)&t.wt, wbytes + 16384) != hipSuccess) return false;  // + slack: fixed-length wave slices
            if (hipMalloc((void **)&t.sbt, sbytes + 1024) != hipSuccess) {
                (void)hipFree(t.wt);
                return false;
            }
            if (repack_w4_tiled(w.weight_dev, (const uint16_t *)w.scales_dev, (const uint16_t *)w.biases_dev, t.wt, t.sbt, w.rows,
                                w.cols, e->stream) != 0) {
                (void)hipFree(t.wt);
                (void)hipFree(t.sbt);
                return false;
            }
            e->tiled[w.weight_dev] = t;
            e->tiled_bytes += wbytes + sbytes;
            return true;
        };
        bool ok = true;
        for (const auto &l : e->layers) {
            ok = ok && add_tiled(l.wqkv) && add_tiled(l.wo);
            if (l.wgu.weight_dev) ok = ok && add_tiled(l.wgu) && add_tiled(l.wdown);
        }
        ok = ok && add_tiled(e->head());
        if (!ok) return cleanup_fail("engine_create: hipMalloc(tiled weights) failed");
    }
    if (c.max_prefill_rows >= GEMM8_MIN_ROWS) {  // the caller asked for chunks the plain bf16 GEMM takes: expand the layer matrices once
        auto add_bf16 = [&](const tl_w4 &w) -> bool {
            if (!w.weight_dev || w.cols % 128 != 0 || e->bf16w.count(w.weight_dev) || !gemm8_applicable(c.max_prefill_rows, w.rows, w.cols)) return true;
            uint16_t *wb = nullptr;
            const size_t bytes = (size_t)w.rows * w.cols * 2;
            if (hipMalloc((void **)&wb, bytes) != hipSuccess) return false;
            if (dequant_w4_to_bf16(w.weight_dev, (const uint16_t *)w.scales_dev, (const uint16_t *)w.biases_dev, wb, w.rows, w.cols, e->stream) != 0) {
                (void)hipFree(wb);
                return false;
            }
            e->bf16w[w.weight_dev] = wb;
            e->bf16w_bytes += bytes;
            return true;
        };
        bool ok = true;
        for (const auto &l : e->layers) {
            ok = ok && add_bf16(l.wqkv) && add_bf16(l.wo);
            if (l.wgu.weight_dev) ok = ok && add_bf16(l.wgu) && add_bf16(l.wdown);
        }
        if (!ok) {
            for (auto &kv : e->bf16w) (void)hipFree(kv.second);
            e->bf16w.clear();
            return cleanup_fail("engine_create: hipMalloc(bf16 weights for the prefill GEMM) failed");
        }
    }

    {
        // Workspace of every matmul the engine can launch, allocated once: fp32 slice partials of the skinny matmul
        // (qmm3_min_rows .. 64 decode rows, any of the five matrices) and split-K partials of the prefill GEMM (9 ..
        // rows_cap rows).  Graphs captured later hold this address, so it is never reallocated.
        size_t need = 0;
        const tl_layer_weights *dense = &e->layers[0];  // the first layer that has a dense MLP sizes the MLP workspaces
        for (const auto &l : e->layers)
            if (l.wgu.weight_dev) {
                dense = &l;
                break;
            }
        const tl_w4 *mats[5] = {&e->layers[0].wqkv, &e->layers[0].wo, &dense->wgu, &dense->wdown, &e->head()};
        for (const tl_w4 *w : mats) {
            if (!w->weight_dev) continue;
            for (int M = 1; M <= std::min(64, e->rows_cap); ++M) {
                const Qmm3Plan p3 = qmm3_plan(M, w->cols, w->rows);
                if (p3.ok) need = std::max(need, p3.partial_bytes);
            }
            for (int M = 9; M <= e->rows_cap; ++M)
                need = std::max(need, tl_quantized_matmul_workspace_bytes(M, w->cols, w->rows, TL_BF16, 1, 1));
        }
        if (need > 0) {
            if (hipMalloc(&e->splitk_ws, need) != hipSuccess) return cleanup_fail("engine_create: hipMalloc(matmul workspace) failed");
            e->splitk_ws_bytes = need;
        }
    }

    e->moe.assign(c.num_layers, tl_moe_weights{});
    e->slot_pages.assign(c.max_batch, {});
    e->slot_ctx.assign(c.max_batch, 0);
    e->slot_live.assign(c.max_batch, 0);
    e->slot_produced.assign(c.max_batch, 0);
    e->free_pages.resize(c.num_pages);
    for (int i = 0; i < c.num_pages; ++i) e->free_pages[i] = c.num_pages - 1 - i;  // pop_back hands out 0,1,2,...
    e->page_was_used.assign(c.num_pages, 0);
    e->page_refs.assign(c.num_pages, 0);
    e->stats.pages_free = c.num_pages;
    e->stats.kv_bytes = e->kv_bytes;
    e->stats.workspace_bytes = e->arena_bytes + e->tiled_bytes + e->bf16w_bytes;
    *out = e;
    return TL_OK;
}

extern "C" int tl_engine_set_moe_layer(tl_engine *e, int layer, const tl_moe_weights *w) {
    TL_REQUIRE(e && w, "engine_set_moe_layer: null argument");
    const tl_engine_config &c = e->cfg;
    TL_REQUIRE(layer >= 0 && layer < c.num_layers, "engine_set_moe_layer: layer out of range");
    TL_REQUIRE(e->graphs.empty() && e->stats.decode_steps == 0 && e->stats.prefill_tokens == 0,
               "engine_set_moe_layer: call it before the first prefill / decode");
    TL_REQUIRE(w->num_experts > 0 && w->num_experts <= 1024 && w->experts_per_token > 0 && w->experts_per_token <= 16 &&
                   w->experts_per_token <= w->num_experts,
               "engine_set_moe_layer: need 1 <= experts_per_token <= min(16, num_experts) and num_experts <= 1024");
    TL_REQUIRE(w->intermediate_size > 0 && w->intermediate_size % 128 == 0, "engine_set_moe_layer: intermediate_size must be a positive multiple of 128");
    TL_TRY(check_w4(w->router, w->num_experts, c.hidden_size, "moe router"));
    TL_REQUIRE(w->gate_dev && w->up_dev && w->down_dev && w->gate_scales_dev && w->gate_biases_dev && w->up_scales_dev &&
                   w->up_biases_dev && w->down_scales_dev && w->down_biases_dev,
               "engine_set_moe_layer: null expert tensor");
    TL_REQUIRE((long)e->rows_cap * w->experts_per_token <= 65535, "engine_set_moe_layer: max_prefill_rows x experts_per_token must stay below 65536 (one grouped launch)");
    // the router's matmul must fit the workspace sized at tl_engine_create (it does for every E <= the widest projection)
    for (int M : {1, 8, 9, e->rows_cap})
        TL_REQUIRE(tl_quantized_matmul_workspace_bytes(std::min(M, e->rows_cap), c.hidden_size, w->num_experts, TL_BF16, 1, 1) <= e->splitk_ws_bytes,
                   "engine_set_moe_layer: the router matmul does not fit the engine's matmul workspace");
    const int k = std::max(e->moe_k_max, w->experts_per_token), E = std::max(e->moe_e_max, w->num_experts),
              I = std::max(e->moe_i_max, w->intermediate_size);
    if (k != e->moe_k_max || E != e->moe_e_max || I != e->moe_i_max) {  // (re)size the workspace: nothing captured holds it yet
        const size_t R = (size_t)e->rows_cap;
        size_t off = 0;
        auto carve = [&](size_t bytes) {
            const size_t at = off;
            off = align_up(off + bytes, 256);
            return at;
        };
        const size_t o_log = carve(R * E * 2), o_ids = carve(R * k * 4), o_sc = carve(R * k * 2), o_g = carve(R * k * I * 2),
                     o_u = carve(R * k * I * 2), o_a = carve(R * k * I * 2), o_y = carve(R * k * c.hidden_size * 2);
        TL_HIP(hipStreamSynchronize(e->stream));
        if (e->moe_ws) (void)hipFree(e->moe_ws);
        e->moe_ws = nullptr;
        TL_HIP(hipMalloc((void **)&e->moe_ws, off));
        e->moe_ws_bytes = off;
        char *A = e->moe_ws;
        e->moe_logits = (uint16_t *)(A + o_log);
        e->moe_ids = (int32_t *)(A + o_ids);
        e->moe_scores = (uint16_t *)(A + o_sc);
        e->moe_gate = (uint16_t *)(A + o_g);
        e->moe_up = (uint16_t *)(A + o_u);
        e->moe_act = (uint16_t *)(A + o_a);
        e->moe_y = (uint16_t *)(A + o_y);
        e->moe_k_max = k, e->moe_e_max = E, e->moe_i_max = I;
        e->stats.workspace_bytes = e->arena_bytes + e->tiled_bytes + e->moe_ws_bytes;
    }
    e->moe[layer] = *w;
    // a model with a sparse layer never takes the per-layer decode buffers (enqueue_step: its steps stay on the shared buffers and the
    // hipGraph route): give the 17 MB per layer back (nothing captured holds them yet -- checked above)
    if (e->layer_act_mem) {
        TL_HIP(hipStreamSynchronize(e->stream));
        (void)hipFree(e->layer_act_mem);
        e->layer_act_mem = nullptr;
        e->layer_act.clear();
        e->layer_act_rows = 0, e->layer_act_bytes = 0, e->layer_ws_bytes = 0;
        e->layer_plane_bytes[0] = e->layer_plane_bytes[1] = 0;
    }
    return TL_OK;
}

// Test / lab hook (header): routes that have an A/B twin, switched per engine before its first step.
extern "C" int tl_engine_set_option(tl_engine *e, const char *name, int value) {
    TL_REQUIRE(e && name, "engine_set_option: null argument");
    TL_REQUIRE(e->graphs.empty() && e->stats.decode_steps == 0 && e->stats.prefill_tokens == 0,
               "engine_set_option: call it before the first prefill / decode (captured steps hold the routes they were captured with)");
    const std::string n = name;
    const bool on = value != 0;
    if (n == "qmm3") e->use_qmm3 = on;
    else if (n == "qmm6") e->use_qmm6 = on;
    else if (n == "qmm7") e->use_qmm7 = on;
    else if (n == "attn_qkv_partials") e->attn_qkv_partials = on;
    else if (n == "lmhead_tile_max") e->lm_tile_max_on = on;
    else if (n == "gemm_fused_epilogue") e->gemm_fused_epilogue = on;
    else if (n == "gemm8") e->use_gemm8 = on;
    else if (n == "prefill_reduce_norm") e->fuse_reduce_norm = on;
    else if (n == "aql_fences") e->aql_fences.inner_acquire = e->aql_fences.inner_release = on ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
    else return fail(TL_ERR_INVALID, "engine_set_option: unknown option '" + n + "' (qmm3, qmm6, qmm7, gemm8, prefill_reduce_norm, attn_qkv_partials, lmhead_tile_max, gemm_fused_epilogue, aql_fences)");
    return TL_OK;
}

extern "C" void tl_engine_destroy(tl_engine *e) {
    if (!e) return;
    (void)aql_drain(e);
    (void)hipStreamSynchronize(e->stream);
    e->aql_programs.clear();
    e->aql_queue.reset();
    if (e->layer_act_mem) (void)hipFree(e->layer_act_mem);
    if (e->moe_ws) (void)hipFree(e->moe_ws);
    for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    if (e->arena) (void)hipFree(e->arena);
    if (e->kpool) (void)hipFree(e->kpool);
    if (e->vpool) (void)hipFree(e->vpool);
    if (e->kscale_pool) (void)hipFree(e->kscale_pool);
    if (e->vscale_pool) (void)hipFree(e->vscale_pool);
    if (e->splitk_ws) (void)hipFree(e->splitk_ws);
    if (e->rope_table) (void)hipFree(e->rope_table);
    if (e->rope_cur) (void)hipFree(e->rope_cur);
    for (auto &kv : e->tiled) {
        (void)hipFree(kv.second.wt);
        (void)hipFree(kv.second.sbt);
    }
    for (auto &kv : e->bf16w) (void)hipFree(kv.second);
    if (e->owns_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int tl_engine_kv_format(const tl_engine *e) { return e ? e->kv_format : TL_KV_BF16; }

extern "C" const char *tl_engine_replay_route(const tl_engine *e) {
    static thread_local std::string text;
    if (!e) return "";
    text = e->aql_on ? std::string("aql") : ("hipgraph" + (e->aql_why.empty() ? std::string() : ": " + e->aql_why));
    return text.c_str();
}

extern "C" int tl_engine_synchronize(tl_engine *e) {
    TL_REQUIRE(e, "engine_synchronize: null engine");
    TL_HIP(hipStreamSynchronize(e->stream));
    return TL_OK;
}

extern "C" int tl_engine_begin(tl_engine *e, int slot) {
    TL_TRY(slot_check(e, slot, false));
    TL_REQUIRE(!e->slot_live[slot], "engine_begin: slot already holds a sequence (release it first)");
    e->slot_live[slot] = 1;
    e->slot_ctx[slot] = 0;
    e->slot_produced[slot] = 0;
    std::vector<std::pair<int32_t *, int32_t>> pk;
    pk.emplace_back(e->live + slot, 1);
    pk.emplace_back(e->context_lens + slot, 0);
    pk.emplace_back(e->produced + slot, 0);
    pk.emplace_back(e->tokens + slot, 0);
    return poke(e, pk);
}

extern "C" int tl_engine_reserve(tl_engine *e, int slot, int total_tokens) {
    TL_TRY(slot_check(e, slot, true));
    TL_REQUIRE(total_tokens >= 0, "engine_reserve: total_tokens must be nonnegative");
    std::vector<std::pair<int32_t *, int32_t>> pk;
    TL_TRY(reserve_locked(e, slot, total_tokens, pk));
    e->stats.pages_free = (int)e->free_pages.size();
    return poke(e, pk);
}

extern "C" int tl_engine_release(tl_engine *e, int slot) {
    TL_TRY(slot_check(e, slot, true));
    std::vector<std::pair<int32_t *, int32_t>> pk;
    auto &pages = e->slot_pages[slot];
    for (size_t j = 0; j < pages.size(); ++j) {
        drop_page(e, pages[j]);
        pk.emplace_back(e->block_table + (size_t)slot * e->cfg.max_pages_per_seq + j, -1);
    }
    pages.clear();
    e->slot_live[slot] = 0;
    e->slot_ctx[slot] = 0;
    pk.emplace_back(e->live + slot, 0);
    pk.emplace_back(e->context_lens + slot, 0);
    pk.emplace_back(e->tokens + slot, 0);
    e->stats.pages_free = (int)e->free_pages.size();
    return poke(e, pk);
}

extern "C" int tl_engine_rewind(tl_engine *e, int slot, int n) {
    TL_TRY(slot_check(e, slot, true));
    TL_REQUIRE(n >= 0 && n <= e->slot_ctx[slot], "engine_rewind: cannot rewind past the start of the sequence");
    const int ctx = e->slot_ctx[slot] - n;
    const int keep = (ctx + e->cfg.page_size - 1) / e->cfg.page_size;
    std::vector<std::pair<int32_t *, int32_t>> pk;
    auto &pages = e->slot_pages[slot];
    {
        // the copy of a shared tail page needs one free page; pages this rewind itself returns count.  Checked before
        // anything is mutated, so a failing rewind leaves host mirrors and device tables untouched.
        const bool cow = keep > 0 && ctx % e->cfg.page_size != 0 && keep <= (int)pages.size() && e->page_refs[pages[keep - 1]] > 1;
        size_t will_free = 0;
        for (int j = keep; j < (int)pages.size(); ++j) will_free += e->page_refs[pages[j]] == 1 ? 1 : 0;
        TL_REQUIRE(!cow || e->free_pages.size() + will_free >= 1, "engine_rewind: KV page pool exhausted (copy of a shared tail page)");
    }
    while ((int)pages.size() > keep) {
        drop_page(e, pages.back());
        pk.emplace_back(e->block_table + (size_t)slot * e->cfg.max_pages_per_seq + (pages.size() - 1), -1);
        pages.pop_back();
    }
    // the next append lands in the tail page: if a fork shares it, give this sequence its own copy first
    if (keep > 0 && ctx % e->cfg.page_size != 0 && e->page_refs[pages[keep - 1]] > 1) {
        const int old_id = pages[keep - 1];  // a free page exists: checked above
        const int fresh = take_page(e);
        TL_TRY(copy_page(e, old_id, fresh));
        drop_page(e, old_id);
        pages[keep - 1] = fresh;
        pk.emplace_back(e->block_table + (size_t)slot * e->cfg.max_pages_per_seq + (keep - 1), fresh);
    }
    e->slot_ctx[slot] = ctx;
    pk.emplace_back(e->context_lens + slot, ctx);
    e->stats.pages_free = (int)e->free_pages.size();
    return poke(e, pk);
}

// Fork: slot `dst` becomes a second sequence with the same prefix as `src` (reference KvPrefixGenerator fork / restore on
// dense caches, agent/branching.py:42-208; here on the page pool).  Full pages are shared by reference count -- they are
// never written again -- and a partially filled tail page is copied, so both sequences can append independently.
extern "C" int tl_engine_fork(tl_engine *e, int src, int dst) {
    TL_TRY(slot_check(e, src, true));
    TL_TRY(slot_check(e, dst, false));
    TL_REQUIRE(src != dst, "engine_fork: source and destination are the same slot");
    TL_REQUIRE(!e->slot_live[dst], "engine_fork: destination slot already holds a sequence");
    const tl_engine_config &c = e->cfg;
    const int ctx = e->slot_ctx[src];
    const int full = ctx / c.page_size;
    const bool partial = ctx % c.page_size != 0;
    TL_REQUIRE(!partial || !e->free_pages.empty(), "engine_fork: KV page pool exhausted");
    const auto &from = e->slot_pages[src];
    auto &to = e->slot_pages[dst];
    to.clear();
    std::vector<std::pair<int32_t *, int32_t>> pk;
    for (int j = 0; j < full; ++j) {
        e->page_refs[from[j]]++;
        to.push_back(from[j]);
        pk.emplace_back(e->block_table + (size_t)dst * c.max_pages_per_seq + j, from[j]);
    }
    if (partial) {
        const int fresh = take_page(e);
        TL_TRY(copy_page(e, from[full], fresh));
        to.push_back(fresh);
        pk.emplace_back(e->block_table + (size_t)dst * c.max_pages_per_seq + full, fresh);
    }
    e->slot_live[dst] = 1;
    e->slot_ctx[dst] = ctx;
    e->slot_produced[dst] = 0;
    pk.emplace_back(e->live + dst, 1);
    pk.emplace_back(e->context_lens + dst, ctx);
    pk.emplace_back(e->produced + dst, 0);
    TL_TRY(poke(e, pk));
    // the pending input token travels on the device
    TL_HIP(hipMemcpyAsync(e->tokens + dst, e->tokens + src, sizeof(int32_t), hipMemcpyDeviceToDevice, e->stream));
    e->stats.pages_free = (int)e->free_pages.size();
    return TL_OK;
}

// Move a (prefilled) sequence from slot `src` to the free slot `dst`: the reference prefills a request in its own
// cache and then adopts it into a batch slot (BatchingKvCache.add_request, kv_cache.py:226-238); here only the
// block-table row, context length and pending token change hands — no K/V byte moves.
extern "C" int tl_engine_move(tl_engine *e, int src, int dst) {
    TL_TRY(slot_check(e, src, true));
    TL_TRY(slot_check(e, dst, false));
    TL_REQUIRE(src != dst, "engine_move: source and destination are the same slot");
    TL_REQUIRE(!e->slot_live[dst], "engine_move: destination slot already holds a sequence");
    const int W = e->cfg.max_pages_per_seq;
    std::vector<std::pair<int32_t *, int32_t>> pk;
    auto &pages = e->slot_pages[src];
    for (size_t j = 0; j < pages.size(); ++j) {
        pk.emplace_back(e->block_table + (size_t)dst * W + j, pages[j]);
        pk.emplace_back(e->block_table + (size_t)src * W + j, -1);
    }
    pk.emplace_back(e->context_lens + dst, e->slot_ctx[src]);
    pk.emplace_back(e->context_lens + src, 0);
    pk.emplace_back(e->live + dst, 1);
    pk.emplace_back(e->live + src, 0);
    pk.emplace_back(e->produced + dst, 0);
    TL_TRY(poke(e, pk));
    TL_HIP(hipMemcpyAsync(e->tokens + dst, e->tokens + src, 4, hipMemcpyDeviceToDevice, e->stream));
    e->slot_pages[dst] = std::move(e->slot_pages[src]);
    e->slot_pages[src].clear();
    e->slot_ctx[dst] = e->slot_ctx[src];
    e->slot_ctx[src] = 0;
    e->slot_live[dst] = 1;
    e->slot_live[src] = 0;
    e->slot_produced[dst] = 0;
    e->slot_produced[src] = 0;
    return TL_OK;
}

// Pending token ids of slots [0, count) after synchronising the stream (one copy per decode step instead of one
// ring read per slot).
extern "C" int tl_engine_read_pending(tl_engine *e, int count, int32_t *out) {
    TL_REQUIRE(e && out, "engine_read_pending: null argument");
    TL_REQUIRE(count > 0 && count <= e->cfg.max_batch, "engine_read_pending: count out of range");
    TL_HIP(hipStreamSynchronize(e->stream));
    TL_HIP(hipMemcpy(out, e->tokens, (size_t)count * 4, hipMemcpyDeviceToHost));
    return TL_OK;
}

extern "C" int tl_engine_context_len(const tl_engine *e, int slot) {
    if (!e || slot < 0 || slot >= e->cfg.max_batch || !e->slot_live[slot]) return -1;
    return e->slot_ctx[slot];
}

extern "C" int tl_engine_set_token(tl_engine *e, int slot, int32_t token) {
    TL_TRY(slot_check(e, slot, true));
    TL_REQUIRE(token >= 0 && token < e->cfg.vocab_size, "engine_set_token: token id out of range");
    std::vector<std::pair<int32_t *, int32_t>> pk;
    pk.emplace_back(e->tokens + slot, token);
    return poke(e, pk);
}

// logits_mode: 0 = none, 1 = last row (greedy id recorded as the slot's pending token), 2 = every row (n <= 8: greedy ids
// land in e->verify_ids, nothing is recorded; speculative verification)
// the paged attention operator over layer l's pages, by the engine's page format
static int engine_paged_attention(tl_engine *e, int l, const uint16_t *q_t, const int32_t *block_row, const int32_t *ctx_dev, uint16_t *attn_t,
                                  int n, int ctx_hint) {
    const tl_engine_config &c = e->cfg;
    const int D = c.head_dim, Hq = c.num_heads, Hkv = c.num_kv_heads;
    if (e->kv_format == TL_KV_FP8_E4M3)
        return tl_paged_attention_fp8(q_t, e->layer_k(l), e->layer_ks(l), e->layer_v(l), e->layer_vs(l), block_row, ctx_dev, attn_t, Hq, n, D,
                                      c.num_pages, c.page_size, c.max_pages_per_seq, Hq, Hkv, 1.0f / sqrtf((float)D), 1, ctx_hint, e->attn_ws,
                                      e->attn_ws_bytes, e->stream);
    return tl_paged_attention(q_t, e->layer_k(l), e->layer_v(l), block_row, ctx_dev, attn_t, Hq, n, D, c.num_pages, c.page_size,
                              c.max_pages_per_seq, Hq, Hkv, 1.0f / sqrtf((float)D), 1, ctx_hint, TL_BF16, e->attn_ws, e->attn_ws_bytes, e->stream);
}
static void launch_qkv_post(tl_engine *e, int l, QkvPostArgs &q, int n) {
    q.key_scales = e->layer_ks(l);
    q.value_scales = e->layer_vs(l);
    switch (e->cfg.head_dim) {
        case 128:
            if (e->kv_format == TL_KV_FP8_E4M3) hipLaunchKernelGGL((qkv_post_kernel<8, true>), dim3(n), dim3(256), 0, e->stream, q);
            else hipLaunchKernelGGL((qkv_post_kernel<8>), dim3(n), dim3(256), 0, e->stream, q);
            break;
        case 64: hipLaunchKernelGGL((qkv_post_kernel<4>), dim3(n), dim3(256), 0, e->stream, q); break;
        default: hipLaunchKernelGGL((qkv_post_kernel<2>), dim3(n), dim3(256), 0, e->stream, q); break;
    }
}

static int prefill_impl(tl_engine *e, int slot, const int32_t *tokens, int n, int logits_mode) {
    const int want_logits = logits_mode == 1;
    TL_TRY(slot_check(e, slot, true));
    TL_REQUIRE(tokens && n > 0, "engine_prefill: need at least one token");
    TL_REQUIRE(n <= e->cfg.max_prefill_rows, "engine_prefill: chunk exceeds max_prefill_rows");
    const tl_engine_config &c = e->cfg;
    TL_REQUIRE(n <= 8 || c.head_dim == 128, "engine_prefill: chunks longer than 8 tokens need head_dim 128 (bf16 FlashAttention)");
    for (int i = 0; i < n; ++i) TL_REQUIRE(tokens[i] >= 0 && tokens[i] < c.vocab_size, "engine_prefill: token id out of range");
    const int start = e->slot_ctx[slot];
    std::vector<std::pair<int32_t *, int32_t>> pk;
    TL_TRY(reserve_locked(e, slot, start + n, pk));
    e->stats.pages_free = (int)e->free_pages.size();
    pk.emplace_back(e->scratch_ctx, start + n);
    TL_TRY(poke(e, pk));
    TL_HIP(hipMemcpyAsync(e->prefill_tokens, tokens, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));

    const int D = c.head_dim, Hq = c.num_heads, Hkv = c.num_kv_heads;
    const int32_t *block_row = e->block_table + (size_t)slot * c.max_pages_per_seq;
    TL_TRY(tl_quantized_embedding(e->prefill_tokens, 0, e->embed.scales_dev, e->embed.biases_dev, e->embed.weight_dev, e->x,
                                  n, c.hidden_size, c.vocab_size, 128, 4, TL_BF16, e->stream));
    const size_t attn_ws_need = tl_paged_attention_workspace_bytes(Hq, n, D, c.page_size, c.max_pages_per_seq, Hq, Hkv, start + n);
    TL_REQUIRE(attn_ws_need <= e->attn_ws_bytes, "engine_prefill: attention workspace too small");
    bool x_normed = false;  // the previous layer's w_down reduction left this layer's normalised rows in xn (engine_gemm)
    for (int l = 0; l < c.num_layers; ++l) {
        const tl_layer_weights &w = e->layers[l];
        if (!x_normed) TL_TRY(tl_rms_norm(e->x, w.input_norm_dev, e->xn, n, c.hidden_size, c.rms_norm_eps, TL_BF16, e->stream));
        x_normed = false;
        TL_TRY(engine_gemm(e, w.wqkv, e->xn, e->qkv, n, EPI_STORE, nullptr));
        QkvPostArgs q{};
        q.qkv = e->qkv;
        q.q_norm_w = (const uint16_t *)w.q_norm_dev;
        q.k_norm_w = (const uint16_t *)w.k_norm_dev;
        q.q_t = e->q_t;
        q.key_pages = e->layer_k(l);
        q.value_pages = e->layer_v(l);
        q.block_row = block_row;
        q.T = n;
        q.start = start;
        q.page_size = c.page_size;
        q.max_pages = c.max_pages_per_seq;
        q.num_heads = Hq;
        q.num_kv_heads = Hkv;
        q.eps = c.rms_norm_eps;
        q.rope_base = c.rope_theta;
        launch_qkv_post(e, l, q, n);
        TL_CHECK_LAUNCH("engine qkv_post");
        TL_TRY(engine_paged_attention(e, l, e->q_t, block_row, e->scratch_ctx, e->attn_t, n, start + n));
        {
            const long total = (long)Hq * n * (D / 8);
            hipLaunchKernelGGL(heads_to_rows_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, e->stream, e->attn_t, e->attn, Hq, n, D);
        }
        bool h_normed = false;
        TL_TRY(engine_gemm(e, w.wo, e->attn, e->h, n, EPI_RESIDUAL, e->x, w.post_norm_dev, e->xn, &h_normed));
        if (!h_normed) TL_TRY(tl_rms_norm(e->h, w.post_norm_dev, e->xn, n, c.hidden_size, c.rms_norm_eps, TL_BF16, e->stream));
        if (e->is_moe(l)) {
            TL_TRY(engine_moe_mlp(e, l, e->xn, e->h, e->x, n, nullptr));
        } else {
            TL_REQUIRE(w.wgu.weight_dev != nullptr, "engine: a layer has neither a dense MLP nor experts (tl_engine_set_moe_layer)");
            TL_TRY(engine_gemm(e, w.wgu, e->xn, e->act, n, EPI_SWIGLU, nullptr));
            TL_TRY(engine_gemm(e, w.wdown, e->act, e->x, n, EPI_RESIDUAL, e->h, l + 1 < c.num_layers ? e->layers[l + 1].input_norm_dev : nullptr, e->xn,
                               &x_normed));
        }
        TL_CHECK_LAUNCH("engine prefill layer");
    }
    e->slot_ctx[slot] = start + n;
    pk.emplace_back(e->context_lens + slot, start + n);
    TL_TRY(poke(e, pk));
    e->stats.prefill_tokens += n;
    if (logits_mode == 2) {
        TL_TRY(engine_qmv(e, e->head(), e->x, e->logits, n, PRO_RMSNORM, EPI_STORE, e->final_norm, nullptr));
        e->logits_rows = n;
        hipLaunchKernelGGL(argmax_rows_kernel, dim3(n), dim3(1024), 0, e->stream, e->logits, c.vocab_size, e->verify_ids);
        TL_CHECK_LAUNCH("engine verify argmax");
    }
    if (want_logits) {
        // logits_to_keep = 1 (reference qwen3_week3.py:331-336): last row only
        const uint16_t *last = e->x + (size_t)(n - 1) * c.hidden_size;
        TL_TRY(engine_qmv(e, e->head(), last, e->logits, 1, PRO_RMSNORM, EPI_STORE, e->final_norm, nullptr));
        e->logits_rows = 1;
        StepEndArgs s{};
        s.logits = e->logits;
        s.vocab = c.vocab_size;
        s.slot0 = slot;
        s.tokens = e->tokens;
        s.context_lens = e->context_lens;
        s.live = e->live;
        s.produced = e->produced;
        s.ring = e->ring;
        s.ring_cap = e->ring_cap;
        s.advance = 0;
        s.emb_w = e->embed.weight_dev;
        s.emb_s = (const uint16_t *)e->embed.scales_dev;
        s.emb_b = (const uint16_t *)e->embed.biases_dev;
        s.x = e->h;  // scratch: the prefill activations in x[0..n) must stay intact; decode re-embeds from tokens
        s.hidden = c.hidden_size;
        s.rope_table = e->rope_table;
        s.rope_cur = e->rope_cur;
        s.rope_positions = e->rope_positions;
        s.rope_half = c.head_dim / 2;
        hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(1024), 0, e->stream, s);
        TL_CHECK_LAUNCH("engine prefill argmax");
        e->slot_produced[slot] += 1;
    }
    return TL_OK;
}

// Several sequences' chunks in ONE pass of the multi-token path.  The projections (97 % of the prefill FLOPs) run once over the
// concatenated rows -- a 2,048-row GEMM instead of several 300-row ones -- while RoPE / KV append, the paged FlashAttention
// and the head transpose stay per sequence (each has its own block-table row, start position and causal mask).
static int prefill_packed_impl(tl_engine *e, int n_seqs, const int *slots, const int32_t *tokens, const int *lens, const int *want_logits) {
    const tl_engine_config &c = e->cfg;
    TL_REQUIRE(e && slots && tokens && lens && want_logits, "engine_prefill_packed: null argument");
    TL_REQUIRE(n_seqs >= 1 && n_seqs <= 16, "engine_prefill_packed: between 1 and 16 sequences per call");
    TL_REQUIRE(c.head_dim == 128, "engine_prefill_packed: head_dim 128 (bf16 FlashAttention)");
    int total = 0;
    size_t extra_pages = 0;
    for (int i = 0; i < n_seqs; ++i) {
        TL_TRY(slot_check(e, slots[i], true));
        TL_REQUIRE(lens[i] > 0, "engine_prefill_packed: every sequence needs at least one token");
        for (int j = 0; j < i; ++j) TL_REQUIRE(slots[j] != slots[i], "engine_prefill_packed: a slot appears twice");
        const int need = (e->slot_ctx[slots[i]] + lens[i] + c.page_size - 1) / c.page_size;
        TL_REQUIRE(need <= c.max_pages_per_seq, "engine: sequence exceeds max_pages_per_seq * page_size tokens");
        if (need > (int)e->slot_pages[slots[i]].size()) extra_pages += (size_t)need - e->slot_pages[slots[i]].size();
        total += lens[i];
    }
    TL_REQUIRE(total <= c.max_prefill_rows, "engine_prefill_packed: the chunks together exceed max_prefill_rows");
    {
        int wanted = 0;
        for (int i = 0; i < n_seqs; ++i) wanted += want_logits[i] ? 1 : 0;
        TL_REQUIRE(wanted <= std::max(c.max_batch, 8), "engine_prefill_packed: more prompts end in this pass than the logits buffer has rows (max(max_batch, 8))");
    }
    TL_REQUIRE(extra_pages <= e->free_pages.size(), "engine: KV page pool exhausted");  // checked before anything is mutated
    for (int i = 0; i < total; ++i) TL_REQUIRE(tokens[i] >= 0 && tokens[i] < c.vocab_size, "engine_prefill_packed: token id out of range");

    std::vector<std::pair<int32_t *, int32_t>> pk;
    std::vector<int> start(n_seqs), row0(n_seqs);
    int rows = 0;
    for (int i = 0; i < n_seqs; ++i) {
        start[i] = e->slot_ctx[slots[i]];
        row0[i] = rows;
        rows += lens[i];
        TL_TRY(reserve_locked(e, slots[i], start[i] + lens[i], pk));  // cannot fail after the checks above
        pk.emplace_back(e->scratch_ctx + i, start[i] + lens[i]);
    }
    e->stats.pages_free = (int)e->free_pages.size();
    TL_TRY(poke(e, pk));
    pk.clear();
    TL_HIP(hipMemcpyAsync(e->prefill_tokens, tokens, (size_t)total * 4, hipMemcpyHostToDevice, e->stream));

    const int D = c.head_dim, Hq = c.num_heads, Hkv = c.num_kv_heads;
    TL_TRY(tl_quantized_embedding(e->prefill_tokens, 0, e->embed.scales_dev, e->embed.biases_dev, e->embed.weight_dev, e->x,
                                  total, c.hidden_size, c.vocab_size, 128, 4, TL_BF16, e->stream));
    for (int i = 0; i < n_seqs; ++i)
        TL_REQUIRE(tl_paged_attention_workspace_bytes(Hq, lens[i], D, c.page_size, c.max_pages_per_seq, Hq, Hkv, start[i] + lens[i]) <=
                       e->attn_ws_bytes, "engine_prefill_packed: attention workspace too small");
    bool x_normed = false;  // the previous layer's w_down reduction left this layer's normalised rows in xn (engine_gemm)
    for (int l = 0; l < c.num_layers; ++l) {
        const tl_layer_weights &w = e->layers[l];
        if (!x_normed) TL_TRY(tl_rms_norm(e->x, w.input_norm_dev, e->xn, total, c.hidden_size, c.rms_norm_eps, TL_BF16, e->stream));
        x_normed = false;
        TL_TRY(engine_gemm(e, w.wqkv, e->xn, e->qkv, total, EPI_STORE, nullptr));
        for (int i = 0; i < n_seqs; ++i) {
            const int n = lens[i];
            const int32_t *block_row = e->block_table + (size_t)slots[i] * c.max_pages_per_seq;
            uint16_t *q_t = e->q_t + (size_t)row0[i] * Hq * D;        // this sequence's [Hq][n][D] block
            uint16_t *attn_t = e->attn_t + (size_t)row0[i] * Hq * D;
            QkvPostArgs q{};
            q.qkv = e->qkv + (size_t)row0[i] * (Hq + 2 * Hkv) * D;
            q.q_norm_w = (const uint16_t *)w.q_norm_dev;
            q.k_norm_w = (const uint16_t *)w.k_norm_dev;
            q.q_t = q_t;
            q.key_pages = e->layer_k(l);
            q.value_pages = e->layer_v(l);
            q.block_row = block_row;
            q.T = n;
            q.start = start[i];
            q.page_size = c.page_size;
            q.max_pages = c.max_pages_per_seq;
            q.num_heads = Hq;
            q.num_kv_heads = Hkv;
            q.eps = c.rms_norm_eps;
            q.rope_base = c.rope_theta;
            launch_qkv_post(e, l, q, n);
            TL_CHECK_LAUNCH("engine qkv_post");
            TL_TRY(engine_paged_attention(e, l, q_t, block_row, e->scratch_ctx + i, attn_t, n, start[i] + n));
            const long items = (long)Hq * n * (D / 8);
            hipLaunchKernelGGL(heads_to_rows_kernel, dim3(ceil_div(items, 256)), dim3(256), 0, e->stream, attn_t,
                               e->attn + (size_t)row0[i] * Hq * D, Hq, n, D);
        }
        bool h_normed = false;
        TL_TRY(engine_gemm(e, w.wo, e->attn, e->h, total, EPI_RESIDUAL, e->x, w.post_norm_dev, e->xn, &h_normed));
        if (!h_normed) TL_TRY(tl_rms_norm(e->h, w.post_norm_dev, e->xn, total, c.hidden_size, c.rms_norm_eps, TL_BF16, e->stream));
        if (e->is_moe(l)) {
            TL_TRY(engine_moe_mlp(e, l, e->xn, e->h, e->x, total, nullptr));
        } else {
            TL_REQUIRE(w.wgu.weight_dev != nullptr, "engine: a layer has neither a dense MLP nor experts (tl_engine_set_moe_layer)");
            TL_TRY(engine_gemm(e, w.wgu, e->xn, e->act, total, EPI_SWIGLU, nullptr));
            TL_TRY(engine_gemm(e, w.wdown, e->act, e->x, total, EPI_RESIDUAL, e->h, l + 1 < c.num_layers ? e->layers[l + 1].input_norm_dev : nullptr, e->xn,
                               &x_normed));
        }
        TL_CHECK_LAUNCH("engine packed prefill layer");
    }
    int n_logits = 0;
    for (int i = 0; i < n_seqs; ++i) {
        e->slot_ctx[slots[i]] = start[i] + lens[i];
        pk.emplace_back(e->context_lens + slots[i], start[i] + lens[i]);
        if (want_logits[i]) {  // last rows side by side in xn: one lm_head pass over them (logits_to_keep = 1, qwen3_week3.py:331-336)
            TL_HIP(hipMemcpyAsync(e->xn + (size_t)n_logits * c.hidden_size, e->x + (size_t)(row0[i] + lens[i] - 1) * c.hidden_size,
                                  (size_t)c.hidden_size * 2, hipMemcpyDeviceToDevice, e->stream));
            ++n_logits;
        }
    }
    TL_TRY(poke(e, pk));
    e->stats.prefill_tokens += total;
    if (n_logits > 0) {
        TL_TRY(engine_qmv(e, e->head(), e->xn, e->logits, n_logits, PRO_RMSNORM, EPI_STORE, e->final_norm, nullptr));
        e->logits_rows = n_logits;
        int j = 0;
        for (int i = 0; i < n_seqs; ++i) {
            if (!want_logits[i]) continue;
            StepEndArgs s{};
            s.logits = e->logits + (size_t)j * c.vocab_size;
            s.vocab = c.vocab_size;
            s.slot0 = slots[i];
            s.tokens = e->tokens;
            s.context_lens = e->context_lens;
            s.live = e->live;
            s.produced = e->produced;
            s.ring = e->ring;
            s.ring_cap = e->ring_cap;
            s.advance = 0;
            s.emb_w = e->embed.weight_dev;
            s.emb_s = (const uint16_t *)e->embed.scales_dev;
            s.emb_b = (const uint16_t *)e->embed.biases_dev;
            s.x = e->h;  // scratch, as in prefill_impl
            s.hidden = c.hidden_size;
            s.rope_table = e->rope_table;
            s.rope_cur = e->rope_cur;
            s.rope_positions = e->rope_positions;
            s.rope_half = c.head_dim / 2;
            hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(1024), 0, e->stream, s);
            TL_CHECK_LAUNCH("engine packed prefill argmax");
            e->slot_produced[slots[i]] += 1;
            ++j;
        }
    }
    return TL_OK;
}

extern "C" int tl_engine_prefill_packed(tl_engine *e, int n_seqs, const int *slots, const int32_t *tokens, const int *lens,
                                        const int *want_logits) {
    TL_REQUIRE(e, "engine_prefill_packed: null engine");
    return prefill_packed_impl(e, n_seqs, slots, tokens, lens, want_logits);
}

extern "C" int tl_engine_prefill(tl_engine *e, int slot, const int32_t *tokens, int n, int want_logits) {
    return prefill_impl(e, slot, tokens, n, want_logits ? 1 : 0);
}

extern "C" int tl_engine_verify(tl_engine *e, int slot, const int32_t *tokens, int n, int32_t *out_ids) {
    TL_REQUIRE(e && out_ids, "engine_verify: null argument");
    TL_REQUIRE(n >= 1 && n <= 8, "engine_verify: between 1 and 8 tokens per call (the paged decode kernel's query rows)");
    TL_TRY(prefill_impl(e, slot, tokens, n, 2));
    TL_HIP(hipMemcpyAsync(out_ids, e->verify_ids, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
    TL_HIP(hipStreamSynchronize(e->stream));
    return TL_OK;
}

// directory of this shared library: the device-only code objects of the AQL route lie next to it
static std::string library_dir() {
    Dl_info info{};
    if (dladdr((const void *)&library_dir, &info) == 0 || !info.dli_fname) return ".";
    const std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

// everything the AQL queue holds has run: the stream may be used again (and the host may read what the steps wrote)
static int aql_drain(tl_engine *e) {
    if (e->aql_queue && e->aql_queue->busy()) {
        std::string why;
        if (!e->aql_queue->wait(30.0, why)) return fail(TL_ERR_HIP, "engine: " + why);
    }
    return TL_OK;
}

extern "C" int tl_engine_decode(tl_engine *e, int batch, int steps, int use_graph) {
    TL_REQUIRE(e, "engine_decode: null engine");
    TL_REQUIRE(batch > 0 && batch <= e->cfg.max_batch, "engine_decode: batch out of range");
    TL_REQUIRE(steps >= 0, "engine_decode: steps must be nonnegative");
    if (steps == 0) return TL_OK;
    {  // launches go to the CURRENT device's copy of a kernel, the AQL packets to the engine's own agent: both must be the engine's device
        int dev_now = -1;
        TL_HIP(hipGetDevice(&dev_now));
        TL_REQUIRE(dev_now == e->device, "engine_decode: the current HIP device is not the device this engine was created on");
    }
    const tl_engine_config &c = e->cfg;
    // input activations of the first step come from the pending token ids
    hipLaunchKernelGGL(embed_slots_kernel, dim3(batch), dim3(256), 0, e->stream, e->tokens, e->embed.weight_dev,
                       (const uint16_t *)e->embed.scales_dev, (const uint16_t *)e->embed.biases_dev, e->x, c.hidden_size,
                       c.vocab_size, e->context_lens, e->rope_table, e->rope_cur, e->rope_positions, c.head_dim / 2, e->ss_x);
    TL_CHECK_LAUNCH("engine embed");
    std::vector<std::pair<int32_t *, int32_t>> pk;
    bool on_queue = false;  // steps of this call are in flight on the AQL queue (the stream is idle and must stay so until they are drained)
    for (int s = 0; s < steps; ++s) {
        int max_ctx = 1;
        const int rrc = reserve_step_locked(e, batch, pk, &max_ctx);
        if (rrc != TL_OK) {
            (void)aql_drain(e);
            return rrc;
        }
        if (!pk.empty()) {
            e->stats.pages_free = (int)e->free_pages.size();
            if (on_queue) {  // a page id changes: the poke is a stream launch and must land between the steps
                TL_TRY(aql_drain(e));
                on_queue = false;
            }
            TL_TRY(poke(e, pk));
        }
        const SplitPlan sp = pick_decode_splits(e, batch, max_ctx);
        if (use_graph && e->warmed) {
            const auto key = std::make_pair(batch, sp.key());
            auto it = e->graphs.find(key);
            if (it == e->graphs.end()) {
                // The split plan (and with it the key) changes every 64 * n_splits tokens of context: a long run would keep one
                // ~220-node executable graph per plan and row bucket for ever.  Plans are visited in order of growing context, so
                // when the cache is full the old ones are dead: drop them all (a live plan is re-captured once, ~0.3 ms).
                if (on_queue) {
                    TL_TRY(aql_drain(e));
                    on_queue = false;
                }
                if (e->graphs.size() >= 48) {
                    TL_HIP(hipStreamSynchronize(e->stream));
                    for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
                    e->graphs.clear();
                    e->aql_programs.clear();
                    e->stats.graph_cache_flushes++;
                }
                hipGraph_t graph = nullptr;
                TL_HIP(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
                const int rc = enqueue_step(e, batch, sp);
                const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
                if (rc != TL_OK) {
                    if (graph) (void)hipGraphDestroy(graph);
                    return rc;
                }
                if (ce != hipSuccess) return fail(TL_ERR_HIP, std::string("engine_decode: graph capture failed: ") + hipGetErrorString(ce));
                hipGraphExec_t exec = nullptr;
                const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (ie == hipSuccess && e->aql_on && !e->step_written_once) e->aql_why = "a hand-over of this plan lives in a shared buffer (written more than once per step)";
                if (ie == hipSuccess && e->aql_on && e->step_written_once) {  // the same nodes as packet templates (aql.h); a plan that cannot be built keeps the graph route
                    auto prog = std::make_unique<AqlProgram>();
                    if (aql_program_from_graph(*e->aql_rt, graph, e->stream, *prog, e->aql_why) == 0) e->aql_programs[key] = std::move(prog);
                }
                (void)hipGraphDestroy(graph);
                if (ie != hipSuccess) return fail(TL_ERR_HIP, std::string("engine_decode: graph instantiate failed: ") + hipGetErrorString(ie));
                it = e->graphs.emplace(key, exec).first;
                e->stats.graph_captures++;
            }
            const auto prog = e->aql_on ? e->aql_programs.find(key) : e->aql_programs.end();
            if (prog != e->aql_programs.end()) {
                bool first = false;
                if (!on_queue) {  // hand-over stream -> queue: everything enqueued so far (embedding gather, pokes, earlier steps) has run
                    TL_HIP(hipStreamSynchronize(e->stream));
                    on_queue = first = true;
                }
                std::string why;
                if (!e->aql_queue->submit(*prog->second, e->aql_fences, first, s + 1 == steps, why)) {
                    (void)aql_drain(e);
                    return fail(TL_ERR_HIP, "engine_decode: " + why);
                }
                e->stats.aql_steps++;
            } else {
                if (on_queue) {
                    TL_TRY(aql_drain(e));
                    on_queue = false;
                }
                TL_HIP(hipGraphLaunch(it->second, e->stream));
            }
            e->stats.graph_replays++;
        } else {
            if (on_queue) {
                TL_TRY(aql_drain(e));
                on_queue = false;
            }
            TL_TRY(enqueue_step(e, batch, sp));
            e->warmed = true;
        }
        for (int b = 0; b < batch; ++b) {
            if (!e->slot_live[b]) continue;
            e->slot_ctx[b] += 1;
            e->slot_produced[b] += 1;
        }
        e->stats.decode_steps++;
    }
    // the queue is not the stream: what follows this call (reads, prefills, the next call's embedding gather) is stream-ordered
    if (on_queue) TL_TRY(aql_drain(e));
    e->logits_rows = batch;
    return TL_OK;
}

extern "C" int tl_engine_read_tokens(tl_engine *e, int slot, int count, int32_t *out) {
    TL_TRY(slot_check(e, slot, false));
    TL_REQUIRE(out && count >= 0 && count <= e->ring_cap, "engine_read_tokens: bad count");
    TL_REQUIRE(count <= e->slot_produced[slot], "engine_read_tokens: fewer ids have been produced");
    TL_HIP(hipStreamSynchronize(e->stream));
    std::vector<int32_t> ring(e->ring_cap);
    TL_HIP(hipMemcpy(ring.data(), e->ring + (size_t)slot * e->ring_cap, (size_t)e->ring_cap * 4, hipMemcpyDeviceToHost));
    const int produced = e->slot_produced[slot];
    for (int i = 0; i < count; ++i) out[i] = ring[(produced - count + i) % e->ring_cap];
    return TL_OK;
}

extern "C" const void *tl_engine_logits_dev(const tl_engine *e) { return e ? e->logits : nullptr; }
extern "C" int tl_engine_copy_logits(tl_engine *e, void *dst_dev, int rows) {
    TL_REQUIRE(e && dst_dev, "engine_copy_logits: null argument");
    TL_REQUIRE(rows > 0 && rows <= e->cfg.max_batch, "engine_copy_logits: rows out of range");
    TL_HIP(hipMemcpyAsync(dst_dev, e->logits, (size_t)rows * e->cfg.vocab_size * 2, hipMemcpyDeviceToDevice, e->stream));
    return TL_OK;
}
extern "C" const int32_t *tl_engine_tokens_dev(const tl_engine *e) { return e ? e->tokens : nullptr; }

extern "C" int tl_engine_get_stats(const tl_engine *e, tl_engine_stats *out) {
    TL_REQUIRE(e && out, "engine_get_stats: null argument");
    *out = e->stats;
    out->pages_free = (int)e->free_pages.size();
    return TL_OK;
}

extern "C" size_t tl_engine_step_bytes(const tl_engine *e, int batch) {
    if (!e) return 0;
    const tl_engine_config &c = e->cfg;
    auto w4_bytes = [](const tl_w4 &w) { return (size_t)w.rows * w.cols / 2 + (size_t)w.rows * (w.cols / 128) * 4; };
    size_t total = 0;
    for (const auto &l : e->layers) total += w4_bytes(l.wqkv) + w4_bytes(l.wo) + w4_bytes(l.wgu) + w4_bytes(l.wdown);
    total += w4_bytes(e->head());
    // K and V rows of a cached token, all layers: 2 bytes per element, or (FP8 pages) one byte per element + a 4-byte scale per row
    const size_t kv_per_token = e->kv_format == TL_KV_FP8_E4M3 ? (size_t)2 * c.num_layers * c.num_kv_heads * (c.head_dim + 4)
                                                               : (size_t)2 * c.num_layers * c.num_kv_heads * c.head_dim * 2;
    for (int b = 0; b < batch && b < c.max_batch; ++b)
        if (e->slot_live[b]) total += kv_per_token * (size_t)e->slot_ctx[b];
    return total;
}

// One REAL decode step (state advances exactly like tl_engine_decode(e, batch, 1, 0)) launched eagerly with
// in-kernel wall-clock stamps; see tl_step_profile in the header.
extern "C" int tl_engine_profile_step(tl_engine *e, int batch, tl_step_profile *out) {
    TL_REQUIRE(e && out, "engine_profile_step: null argument");
    TL_REQUIRE(batch > 0 && batch <= e->cfg.max_batch, "engine_profile_step: batch out of range");
    const tl_engine_config &c = e->cfg;
    int rate_khz = 0;
    int dev = 0;
    TL_HIP(hipGetDevice(&dev));
    TL_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev));
    TL_REQUIRE(rate_khz > 0, "engine_profile_step: device reports no wall clock rate");
    // largest grid of the step: the lm_head GEMV (4 rows per workgroup at worst) or the attention grid
    const int max_wg = std::max(c.vocab_size / 4 + 64, 64 * 4 * c.num_kv_heads * batch) + 1024;
    ProfCtx pc;
    pc.cap = c.num_layers * 12 + 8;  // up to 11 launches per layer at 5 .. 64 rows (four skinny matmuls + reductions, attention, merge, norms)
    TL_HIP(hipStreamSynchronize(e->stream));
    TL_HIP(hipMalloc((void **)&pc.buf, (size_t)max_wg * 2 * sizeof(prof_t)));
    TL_HIP(hipMalloc((void **)&pc.pairs, (size_t)pc.cap * 2 * sizeof(prof_t)));
    auto cleanup = [&]() {
        (void)hipFree(pc.buf);
        (void)hipFree(pc.pairs);
    };
    if (hipMemsetAsync(pc.buf, 0, (size_t)max_wg * 2 * sizeof(prof_t), e->stream) != hipSuccess) {
        cleanup();
        return fail(TL_ERR_HIP, "engine_profile_step: memset failed");
    }
    hipLaunchKernelGGL(embed_slots_kernel, dim3(batch), dim3(256), 0, e->stream, e->tokens, e->embed.weight_dev,
                       (const uint16_t *)e->embed.scales_dev, (const uint16_t *)e->embed.biases_dev, e->x, c.hidden_size,
                       c.vocab_size, e->context_lens, e->rope_table, e->rope_cur, e->rope_positions, c.head_dim / 2, e->ss_x);
    std::vector<std::pair<int32_t *, int32_t>> pk;
    int max_ctx = 1;
    int rc = reserve_step_locked(e, batch, pk, &max_ctx);
    if (rc != TL_OK) {
        cleanup();
        return rc;
    }
    rc = pk.empty() ? TL_OK : poke(e, pk);
    const SplitPlan sp = pick_decode_splits(e, batch, max_ctx);
    const int n_splits = sp.n_splits;
    if (rc == TL_OK) rc = enqueue_step(e, batch, sp, &pc);
    if (rc != TL_OK) {
        (void)hipStreamSynchronize(e->stream);
        cleanup();
        return rc;
    }
    e->warmed = true;
    for (int b = 0; b < batch; ++b) {
        if (!e->slot_live[b]) continue;
        e->slot_ctx[b] += 1;
        e->slot_produced[b] += 1;
    }
    e->stats.decode_steps++;
    e->logits_rows = batch;
    std::vector<prof_t> pairs(pc.kinds.size() * 2);
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(pairs.data(), pc.pairs, pairs.size() * sizeof(prof_t), hipMemcpyDeviceToHost);
    cleanup();
    if (he != hipSuccess) return fail(TL_ERR_HIP, std::string("engine_profile_step: ") + hipGetErrorString(he));

    *out = tl_step_profile{};
    out->clock_khz = rate_khz;
    out->n_splits = n_splits;
    const double us_per_tick = 1e3 / (double)rate_khz;
    prof_t first = ~0ull, last = 0;
    for (size_t i = 0; i < pc.kinds.size(); ++i) {
        const prof_t t0 = pairs[2 * i], t1 = pairs[2 * i + 1];
        const double us = (t1 > t0 ? (double)(t1 - t0) : 0.0) * us_per_tick;
        first = std::min(first, t0);
        last = std::max(last, t1);
        out->kernel_us[pc.kinds[i]] += us;
        out->launches[pc.kinds[i]] += 1;
    }
    out->span_us = (last > first ? (double)(last - first) : 0.0) * us_per_tick;
    auto w4_bytes = [](const tl_w4 &w) { return (double)w.rows * w.cols / 2 + (double)w.rows * (w.cols / 128) * 4; };
    for (const auto &l : e->layers) {
        out->gemv_bytes[0] += w4_bytes(l.wqkv);
        out->gemv_bytes[1] += w4_bytes(l.wo);
        out->gemv_bytes[2] += w4_bytes(l.wgu);
        out->gemv_bytes[3] += w4_bytes(l.wdown);
    }
    out->gemv_bytes[4] = w4_bytes(e->head());
    return TL_OK;
}


// One REAL decode step, launched eagerly like tl_engine_profile_step, with the written-once checker behind every launch (header).
extern "C" int tl_engine_check_step(tl_engine *e, int batch, tl_step_check *out) {
    TL_REQUIRE(e && out, "engine_check_step: null argument");
    TL_REQUIRE(batch > 0 && batch <= e->cfg.max_batch, "engine_check_step: batch out of range");
    const tl_engine_config &c = e->cfg;
    *out = tl_step_check{};
    out->first_launch = -1, out->first_kind = -1, out->first_region = -1, out->first_offset = -1;
    TL_TRY(aql_drain(e));
    TL_HIP(hipStreamSynchronize(e->stream));
    tl_engine::WrittenOnceCheck &ck = e->check;
    ck = tl_engine::WrittenOnceCheck{};
    ck.region[0] = e->arena + e->arena_act_off, ck.bytes[0] = (e->arena_bytes - e->arena_act_off) / 4 * 4;
    ck.region[1] = e->layer_act_mem, ck.bytes[1] = e->layer_act_mem ? e->layer_act_bytes / 4 * 4 : 0;
    const int max_wg = std::max(c.vocab_size / 4 + 64, 64 * 4 * c.num_kv_heads * batch) + 1024;
    ProfCtx pc;
    pc.cap = c.num_layers * 12 + 8;
    auto cleanup = [&]() {
        for (int rg = 0; rg < 2; ++rg) {
            if (ck.shadow[rg]) (void)hipFree(ck.shadow[rg]);
            if (ck.written[rg]) (void)hipFree(ck.written[rg]);
        }
        if (ck.report) (void)hipFree(ck.report);
        if (pc.buf) (void)hipFree(pc.buf);
        if (pc.pairs) (void)hipFree(pc.pairs);
        ck = tl_engine::WrittenOnceCheck{};
    };
    auto bail = [&](int code, const std::string &msg) {
        (void)hipStreamSynchronize(e->stream);
        cleanup();
        return fail(code, msg);
    };
    bool ok = hipMalloc((void **)&pc.buf, (size_t)max_wg * 2 * sizeof(prof_t)) == hipSuccess && hipMalloc((void **)&pc.pairs, (size_t)pc.cap * 2 * sizeof(prof_t)) == hipSuccess &&
              hipMalloc((void **)&ck.report, 8 * sizeof(unsigned long long)) == hipSuccess;
    for (int rg = 0; rg < 2 && ok; ++rg)
        if (ck.bytes[rg]) ok = hipMalloc((void **)&ck.shadow[rg], ck.bytes[rg]) == hipSuccess && hipMalloc((void **)&ck.written[rg], ck.bytes[rg] / 2) == hipSuccess;
    if (!ok) return bail(TL_ERR_HIP, "engine_check_step: hipMalloc of the shadow buffers failed");
    // the per-layer buffers start the step POISONED (every 16-bit and 32-bit pattern a NaN): a value read before this step wrote it
    // reaches the logits as NaN; the shared activations carry state between steps (x, its sums of squares) and keep their contents
    const unsigned long long report0[8] = {0, ~0ull, 0, 0, 0, 0, 0, 0};
    ok = hipMemsetAsync(pc.buf, 0, (size_t)max_wg * 2 * sizeof(prof_t), e->stream) == hipSuccess &&
         hipMemcpyAsync(ck.report, report0, sizeof(report0), hipMemcpyHostToDevice, e->stream) == hipSuccess;
    if (ok && ck.bytes[1]) ok = hipMemsetAsync(ck.region[1], 0xff, ck.bytes[1], e->stream) == hipSuccess;
    if (!ok) return bail(TL_ERR_HIP, "engine_check_step: initialisation failed");
    hipLaunchKernelGGL(embed_slots_kernel, dim3(batch), dim3(256), 0, e->stream, e->tokens, e->embed.weight_dev,
                       (const uint16_t *)e->embed.scales_dev, (const uint16_t *)e->embed.biases_dev, e->x, c.hidden_size,
                       c.vocab_size, e->context_lens, e->rope_table, e->rope_cur, e->rope_positions, c.head_dim / 2, e->ss_x);
    // the shadows = the regions as the step finds them; nothing written yet
    for (int rg = 0; rg < 2 && ok; ++rg)
        if (ck.bytes[rg]) ok = hipMemcpyAsync(ck.shadow[rg], ck.region[rg], ck.bytes[rg], hipMemcpyDeviceToDevice, e->stream) == hipSuccess &&
                               hipMemsetAsync(ck.written[rg], 0, ck.bytes[rg] / 2, e->stream) == hipSuccess;
    if (!ok) return bail(TL_ERR_HIP, "engine_check_step: shadow copy failed");
    std::vector<std::pair<int32_t *, int32_t>> pk;
    int max_ctx = 1;
    int rc = reserve_step_locked(e, batch, pk, &max_ctx);
    if (rc == TL_OK && !pk.empty()) rc = poke(e, pk);
    const SplitPlan sp = pick_decode_splits(e, batch, max_ctx);
    if (rc == TL_OK) {
        ck.on = true;
        rc = enqueue_step(e, batch, sp, &pc);
        ck.on = false;
    }
    if (rc != TL_OK) {
        (void)hipStreamSynchronize(e->stream);
        cleanup();
        return rc;
    }
    e->warmed = true;
    for (int b = 0; b < batch; ++b) {
        if (!e->slot_live[b]) continue;
        e->slot_ctx[b] += 1;
        e->slot_produced[b] += 1;
    }
    e->stats.decode_steps++;
    e->logits_rows = batch;
    unsigned long long report[8] = {0};
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(report, ck.report, sizeof(report), hipMemcpyDeviceToHost);
    const std::vector<int> kinds = pc.kinds;
    cleanup();
    if (he != hipSuccess) return fail(TL_ERR_HIP, std::string("engine_check_step: ") + hipGetErrorString(he));
    out->launches = (int)kinds.size();
    out->written_once_plan = e->step_written_once ? 1 : 0;
    out->n_splits = sp.n_splits;
    out->double_writes = (long)report[0];
    out->elements_written = (long)report[3];
    if (report[0]) {
        out->first_launch = (int)report[1];
        out->first_kind = report[1] < kinds.size() ? kinds[report[1]] : -1;
        out->first_region = (int)report[2];
        out->first_offset = (long)report[4];
    }
    return TL_OK;
}

// ================================================================================================
// Kernel-level entry points of the decode path (include/tinyllm_engine.h, last section): the SAME launch code the engine
// runs per projection / per layer, on caller-owned buffers.  Used by the operator microbenches and by the parity tests at
// the real Qwen3-4B shapes.
struct tl_tiled_w4 {
    tl_w4 w{};
    tl_engine::Tiled t{};
};

extern "C" int tl_tiled_w4_create(const tl_w4 *w, void *stream, tl_tiled_w4 **out) {
    TL_REQUIRE(w && out, "tiled_w4_create: null argument");
    TL_TRY(check_w4(*w, w->rows, w->cols, "tiled_w4_create"));
    TL_REQUIRE(w->rows > 0 && w->rows % 16 == 0 && w->cols > 0 && w->cols % 128 == 0,
               "tiled_w4_create: rows must be a multiple of 16 and cols a multiple of 128");
    auto *t = new tl_tiled_w4();
    t->w = *w;
    const size_t wbytes = (size_t)w->rows * w->cols / 2, sbytes = (size_t)w->rows * (w->cols / 128) * 4;
    if (hipMalloc((void **)&t->t.wt, wbytes + 16384) != hipSuccess) {
        delete t;
        return fail(TL_ERR_HIP, "tiled_w4_create: hipMalloc failed");
    }
    if (hipMalloc((void **)&t->t.sbt, sbytes + 1024) != hipSuccess) {
        (void)hipFree(t->t.wt);
        delete t;
        return fail(TL_ERR_HIP, "tiled_w4_create: hipMalloc failed");
    }
    if (repack_w4_tiled(w->weight_dev, (const uint16_t *)w->scales_dev, (const uint16_t *)w->biases_dev, t->t.wt, t->t.sbt, w->rows,
                        w->cols, (hipStream_t)stream) != 0) {
        (void)hipFree(t->t.wt);
        (void)hipFree(t->t.sbt);
        delete t;
        return fail(TL_ERR_HIP, "tiled_w4_create: repack launch failed");
    }
    *out = t;
    return TL_OK;
}

extern "C" void tl_tiled_w4_destroy(tl_tiled_w4 *t) {
    if (!t) return;
    (void)hipFree(t->t.wt);
    (void)hipFree(t->t.sbt);
    delete t;
}

extern "C" size_t tl_decode_linear_workspace_bytes(int M, int rows, int cols) {
    if (M <= 0 || rows <= 0 || cols <= 0) return 0;
    size_t need = align_up((size_t)((M + 15) / 16 * 16) * cols * 2, 256);  // RMSNorm output ahead of the skinny matmul / rows in fragment order (kernel 5)
    size_t partial = 0;
    for (int mode = 0; mode < 2; ++mode) {  // either grid of the skinny matmul (kernel 3 / 4 pin one)
        const Qmm3Plan p3 = qmm3_plan(std::min(M, 64), cols, rows, mode);
        if (p3.ok) partial = std::max(partial, p3.partial_bytes);
    }
    return need + partial;
}

static int decode_linear_impl(const tl_tiled_w4 *w, const void *a_dev, void *out_dev, int M, int prologue, int epilogue,
                              const void *norm_w_dev, const void *residual_dev, float eps, int kernel, void *workspace_dev,
                              size_t workspace_bytes, void *stream, const tl_linear_ex *ex, tl_linear_info *info) {
    TL_REQUIRE(w && out_dev, "decode_linear: null argument");
    TL_REQUIRE(M >= 1 && M <= 64, "decode_linear: between 1 and 64 activation rows");
    TL_REQUIRE(prologue == PRO_NONE || prologue == PRO_RMSNORM || (ex && (prologue == PRO_ATTN_MERGE || prologue == PRO_RMS_WEIGHTED)),
               "decode_linear: prologue is 0 (none) or 1 (RMSNorm); tl_decode_linear_ex also takes 2 (merge of attention partials) and 3 (weighted rows)");
    TL_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_RESIDUAL || epilogue == EPI_SWIGLU,
               "decode_linear: epilogue is 0 (store), 1 (residual add) or 2 (SwiGLU over interleaved rows)");
    TL_REQUIRE(prologue == PRO_ATTN_MERGE || a_dev, "decode_linear: null activation rows");
    TL_REQUIRE(prologue != PRO_RMSNORM || norm_w_dev, "decode_linear: the RMSNorm prologue needs its weight");
    TL_REQUIRE(epilogue != EPI_RESIDUAL || residual_dev, "decode_linear: the residual epilogue needs the residual rows");
    TL_REQUIRE(kernel >= 0 && kernel <= 6,
               "decode_linear: kernel is 0 (engine routing), 1 (fused GEMV), 2 (skinny matmul), 3 / 4 (its one-shot / persistent grid), 5 (register-resident matmul), 6 (row-streaming matmul)");
    TL_REQUIRE(epilogue != EPI_SWIGLU || w->w.rows % 2 == 0, "decode_linear: SwiGLU needs an even number of weight rows");
    // the engine's own fused variants: RMSNorm+store (qkv, lm_head), residual (wo, w_down), RMSNorm+SwiGLU (gate|up), plain;
    // through tl_decode_linear_ex also: merged attention partials + residual (wo of one row), weighted rows + SwiGLU (gate|up)
    TL_REQUIRE((prologue == PRO_NONE && epilogue != EPI_SWIGLU) || (prologue == PRO_RMSNORM && epilogue != EPI_RESIDUAL) ||
                   (prologue == PRO_ATTN_MERGE && epilogue == EPI_RESIDUAL) || (prologue == PRO_RMS_WEIGHTED && epilogue == EPI_SWIGLU) ||
                   ((kernel == 5 || kernel == 6) && prologue == PRO_RMS_WEIGHTED && epilogue == EPI_STORE),
               "decode_linear: no fused variant for this prologue / epilogue pair");
    const size_t need = tl_decode_linear_workspace_bytes(M, w->w.rows, w->w.cols);
    TL_REQUIRE(workspace_dev && workspace_bytes >= need, "decode_linear: workspace is missing or too small");
    tl_engine e;  // only the fields the projection code reads
    e.cfg.rms_norm_eps = eps;
    e.stream = (hipStream_t)stream;
    e.tiled[w->w.weight_dev] = w->t;
    e.xn = (uint16_t *)workspace_dev;
    const size_t xn_bytes = align_up((size_t)((M + 15) / 16 * 16) * w->w.cols * 2, 256);
    e.splitk_ws = (char *)workspace_dev + xn_bytes;
    e.splitk_ws_bytes = workspace_bytes - xn_bytes;
    e.force_linear = kernel >= 2 && kernel <= 4 ? 2 : (kernel >= 5 ? 0 : kernel);
    e.use_qmm7 = false, e.force_qmm7 = kernel == 6;  // 5 and 6 name their kernel
    e.qmm3_mode = kernel == 3 ? 0 : (kernel == 4 ? 1 : -1);
    tl_linear_info li{};
    e.linfo = &li;
    if (const char *q = getenv("TL_QMM3_MIN_M")) e.qmm3_min_rows = std::max(1, atoi(q));
    int rc = TL_OK;
    auto done = [&](int code) {
        e.splitk_ws = nullptr;  // borrowed
        e.tiled.clear();
        if (info) *info = li;
        return code;
    };
    if (kernel == 6 && (prologue != PRO_RMS_WEIGHTED || epilogue == EPI_RESIDUAL))
        return done(fail(TL_ERR_INVALID, "decode_linear: the row-streaming matmul takes weighted rows with ss_in (prologue 3) and stores or applies SwiGLU (epilogue 0 / 2)"));
    if (kernel == 5 || kernel == 6) {  // qmm6.h / qmm7.h: plain rows (qmm6 only), or weighted rows with their partial sums of squares
        const bool weighted = prologue == PRO_RMS_WEIGHTED;
        if (prologue == PRO_RMSNORM || prologue == PRO_ATTN_MERGE)
            return done(fail(TL_ERR_INVALID, "decode_linear: the register-resident matmul takes plain rows (prologue 0) or weighted rows with ss_in (prologue 3)"));
        if (weighted && (!ex || !ex->ss_in_dev || !qmm3_takes_ss(ex->ss_in_n)))
            return done(fail(TL_ERR_INVALID, "decode_linear_ex: weighted rows need ss_in (a multiple of 4, at most 256 partials per row)"));
        if (ex && ((ex->out_w_dev != nullptr) != (ex->norm_out_dev != nullptr) || ((ex->out_w_dev || ex->ss_out_dev) && epilogue != EPI_RESIDUAL)))
            return done(fail(TL_ERR_INVALID, "decode_linear_ex: ss_out / (norm_out, out_w) belong to the residual epilogue; norm_out and out_w come together"));
        if (kernel == 5 && !qmm6_plan(M, w->w.cols, w->w.rows).ok)
            return done(fail(TL_ERR_UNSUPPORTED, "decode_linear: the register-resident matmul does not cover this shape"));
        if (kernel == 6 && !qmm7_plan(M, w->w.cols, w->w.rows).ok)
            return done(fail(TL_ERR_UNSUPPORTED, "decode_linear: the row-streaming matmul does not cover this shape"));
        // weighted rows enter the kernel in fragment order (qmm6.h): as the caller left them (ex->fragment_order), or re-ordered here
        const uint16_t *a6 = (const uint16_t *)a_dev;
        if (weighted && !ex->fragment_order) {
            const long n8 = (long)M * w->w.cols / 8;
            hipLaunchKernelGGL(weight_rows_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, e.stream, a6, (const uint16_t *)nullptr, e.xn, n8, w->w.cols / 8, 1);
            a6 = e.xn;
        }
        int ss_n6 = 0;
        rc = engine_qmm6(&e, w->w, a6, (uint16_t *)out_dev, M, epilogue, (const uint16_t *)residual_dev, nullptr, 0,
                         weighted ? ex->ss_in_dev : nullptr, weighted ? ex->ss_in_n : 0, ex ? ex->ss_out_dev : nullptr, &ss_n6,
                         ex ? ex->norm_out_dev : nullptr, ex ? (uint16_t *)ex->out_w_dev : nullptr, weighted, ex ? (ex->fragment_order ? 1 : 0) : 0);
        return done(rc);
    }
    if (!ex) {
        rc = engine_linear(&e, w->w, (const uint16_t *)a_dev, (uint16_t *)out_dev, M, prologue, epilogue, norm_w_dev,
                           (const uint16_t *)residual_dev, nullptr, 0);
        return done(rc);
    }
    // ---- the routes only the engine could reach before round 4 (qmv3.h: PRO_ATTN_MERGE, PRO_RMS_WEIGHTED, ss_in / ss_out, out_w)
    const bool skinny_forced = kernel >= 2 && kernel <= 4;  // its slice reduction also leaves weighted rows (not the GEMV's 16-row sums of squares)
    const bool gemv_only = prologue == PRO_ATTN_MERGE || prologue == PRO_RMS_WEIGHTED || ex->ss_out_dev || (ex->out_w_dev && !skinny_forced);
    if (gemv_only && !(kernel == 1 || (kernel == 0 && M < e.qmm3_min_rows)))
        return done(fail(TL_ERR_INVALID, "decode_linear_ex: merged partials, weighted rows, ss_out and out_w are routes of the fused GEMV (kernel 1, or 0 with fewer than 5 rows)"));
    if ((ex->out_w_dev != nullptr) != (ex->norm_out_dev != nullptr) || (ex->out_w_dev && epilogue != EPI_RESIDUAL) ||
        (ex->ss_out_dev && epilogue != EPI_RESIDUAL))
        return done(fail(TL_ERR_INVALID, "decode_linear_ex: ss_out / (norm_out, out_w) belong to the residual epilogue; norm_out and out_w come together"));
    if (ex->ss_in_dev && (ex->ss_in_n <= 0 || !(prologue == PRO_RMSNORM || prologue == PRO_RMS_WEIGHTED)))
        return done(fail(TL_ERR_INVALID, "decode_linear_ex: ss_in needs ss_in_n > 0 and a normalising prologue (1 or 3)"));
    int ss_n = 0;
    if (prologue == PRO_ATTN_MERGE) {
        if (M != 1 || !ex->merge_ws_dev) return done(fail(TL_ERR_INVALID, "decode_linear_ex: the merging prologue takes ONE row and the split partials (merge_ws_dev)"));
        e.attn_ws = const_cast<float *>(ex->merge_ws_dev);
        rc = engine_wo_merge(&e, w->w, (const uint16_t *)residual_dev, (uint16_t *)out_dev, ex->n_splits, nullptr, ex->ss_out_dev, &ss_n,
                             ex->norm_out_dev, (uint16_t *)ex->out_w_dev);
        e.attn_ws = nullptr;  // borrowed
        if (rc == TL_OK) {
            const Qmv3Plan pl = qmv3_plan(1, w->w.cols, w->w.rows);
            li.kernel = 1, li.launches = 1, li.rows_per_pass = 1;
            li.p[0] = pl.MR, li.p[1] = pl.KS, li.p[2] = pl.CW, li.p[3] = pl.LM, li.p[4] = pl.blocks;
        }
        return done(rc);
    }
    if (prologue == PRO_RMS_WEIGHTED) {
        const Qmv3Plan pl = qmv3_plan(std::min(M, 8), w->w.cols, w->w.rows);
        if (!ex->ss_in_dev || M > 8 || !qmv3_takes_weighted_rows(pl, w->w.cols, ex->ss_in_n))
            return done(fail(TL_ERR_INVALID, "decode_linear_ex: weighted rows need ss_in (a multiple of 4, at most 256 partials per row), at most 8 rows and a row that fits the staging registers"));
    }
    if (gemv_only || kernel == 1 || (kernel == 0 && M < e.qmm3_min_rows)) {
        if (M > 8) return done(fail(TL_ERR_INVALID, "decode_linear_ex: the fused GEMV takes at most 8 rows"));
        rc = engine_qmv(&e, w->w, (const uint16_t *)a_dev, (uint16_t *)out_dev, M, prologue, epilogue, norm_w_dev,
                        (const uint16_t *)residual_dev, nullptr, 0, ex->ss_in_dev, ex->ss_in_n, ex->ss_out_dev, &ss_n, ex->norm_out_dev,
                        (uint16_t *)ex->out_w_dev);
        if (rc == TL_OK && ex->ss_out_dev && ss_n != w->w.rows / 16)
            rc = fail(TL_ERR_UNSUPPORTED, "decode_linear_ex: the GEMV that ran left no sums of squares (packed-dot fallback or several passes)");
        return done(rc);
    }
    // skinny matmul with its fused RMSNorm (any multiple of 4 up to 256 partials per row)
    if (ex->ss_in_dev && !qmm3_takes_ss(ex->ss_in_n))
        return done(fail(TL_ERR_INVALID, "decode_linear_ex: the skinny matmul reads a multiple of 4, at most 256, partial sums of squares per row"));
    rc = engine_linear(&e, w->w, (const uint16_t *)a_dev, (uint16_t *)out_dev, M, prologue, epilogue, norm_w_dev,
                       (const uint16_t *)residual_dev, nullptr, 0, ex->ss_in_dev, nullptr, nullptr, nullptr, ex->ss_in_dev ? ex->ss_in_n : 0,
                       nullptr, ex->norm_out_dev, (uint16_t *)ex->out_w_dev, ex->fragment_order != 0);
    return done(rc);
}

extern "C" int tl_decode_linear(const tl_tiled_w4 *w, const void *a_dev, void *out_dev, int M, int prologue, int epilogue,
                                const void *norm_w_dev, const void *residual_dev, float eps, int kernel, void *workspace_dev,
                                size_t workspace_bytes, void *stream, tl_linear_info *info) {
    return decode_linear_impl(w, a_dev, out_dev, M, prologue, epilogue, norm_w_dev, residual_dev, eps, kernel, workspace_dev,
                              workspace_bytes, stream, nullptr, info);
}

extern "C" int tl_decode_linear_ex(const tl_tiled_w4 *w, const void *a_dev, void *out_dev, int M, int prologue, int epilogue,
                                   const void *norm_w_dev, const void *residual_dev, float eps, int kernel, void *workspace_dev,
                                   size_t workspace_bytes, void *stream, const tl_linear_ex *ex, tl_linear_info *info) {
    TL_REQUIRE(ex, "decode_linear_ex: null extension block (use tl_decode_linear)");
    return decode_linear_impl(w, a_dev, out_dev, M, prologue, epilogue, norm_w_dev, residual_dev, eps, kernel, workspace_dev,
                              workspace_bytes, stream, ex, info);
}

// The prefill projection of large chunks on caller buffers (header): W4 -> bf16 expansion, then the plain bf16 GEMM of gemm8.h.
extern "C" int tl_prefill_weights_bf16(const tl_w4 *w, void *out_dev, void *stream) {
    TL_REQUIRE(w && out_dev, "prefill_weights_bf16: null argument");
    TL_TRY(check_w4(*w, w->rows, w->cols, "prefill_weights_bf16"));
    TL_REQUIRE(w->cols % 128 == 0, "prefill_weights_bf16: columns must be a multiple of the quantisation group (128)");
    if (dequant_w4_to_bf16(w->weight_dev, (const uint16_t *)w->scales_dev, (const uint16_t *)w->biases_dev, (uint16_t *)out_dev, w->rows, w->cols, (hipStream_t)stream) != 0)
        return fail(TL_ERR_HIP, "prefill_weights_bf16: launch failed");
    return TL_OK;
}
extern "C" int tl_prefill_matmul_bf16(const void *a_dev, const void *w_bf16_dev, void *out_dev, int M, int rows, int cols, int epilogue,
                                      const void *residual_dev, void *stream) {
    TL_REQUIRE(a_dev && w_bf16_dev && out_dev, "prefill_matmul_bf16: null argument");
    TL_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_RESIDUAL || epilogue == EPI_SWIGLU, "prefill_matmul_bf16: epilogue is 0 (store), 1 (residual add) or 2 (SwiGLU over interleaved rows)");
    TL_REQUIRE(epilogue != EPI_RESIDUAL || residual_dev, "prefill_matmul_bf16: the residual epilogue needs the residual rows");
    TL_REQUIRE(gemm8_applicable(M, rows, cols), "prefill_matmul_bf16: needs M >= 1, an even number of weight rows and columns in whole 64-wide steps");
    Gemm8Args g{};
    g.a = (const uint16_t *)a_dev, g.w = (const uint16_t *)w_bf16_dev, g.out = (uint16_t *)out_dev, g.residual = (const uint16_t *)residual_dev;
    g.M = M, g.N = rows, g.K = cols;
    if (launch_gemm8_bf16(g, epilogue, (hipStream_t)stream) != 0) return fail(TL_ERR_UNSUPPORTED, "prefill_matmul_bf16: launch failed");
    TL_CHECK_LAUNCH("prefill_matmul_bf16");
    return TL_OK;
}

// (cos, sin) of each row's position = its context length, the same expression as rope_table_kernel
__global__ __launch_bounds__(64) void rope_rows_kernel(const int32_t *__restrict__ context_lens, float2 *__restrict__ rope_cur,
                                                       int half, float base) {
    const int b = blockIdx.x;
    const int pos = context_lens[b];
    for (int item = threadIdx.x; item < half; item += 64) {
        const float fp = -(float)item / (float)half;
        const float angle = (float)pos * exp2f(fp * log2f(base));
        float sn, cs;
        sincosf(angle, &sn, &cs);
        rope_cur[(long)b * half + item] = make_float2(cs, sn);
    }
}

extern "C" size_t tl_decode_attention_fused_workspace_bytes(int batch, int num_heads, int head_dim) {
    if (batch <= 0 || num_heads <= 0 || head_dim <= 0) return 0;
    return align_up((size_t)batch * (head_dim / 2) * sizeof(float2), 256) +
           (size_t)batch * num_heads * 256 * (head_dim + ATTN_WS_PAD) * sizeof(float);
}

static int decode_attention_fused_impl(const void *qkv_dev, const void *q_norm_dev, const void *k_norm_dev, void *key_pages_dev,
                                       float *key_scales_dev, void *value_pages_dev, float *value_scales_dev,
                                       const int32_t *block_table_dev, const int32_t *context_lens_dev, void *out_dev, int batch,
                                       int num_heads, int num_kv_heads, int head_dim, int page_size, int max_pages, float rope_theta,
                                       float eps, int max_context, void *workspace_dev, size_t workspace_bytes, void *stream,
                                       tl_attention_info *info) {
    TL_REQUIRE(qkv_dev && q_norm_dev && k_norm_dev && key_pages_dev && value_pages_dev && block_table_dev && context_lens_dev &&
                   out_dev, "decode_attention_fused: null pointer");
    TL_REQUIRE(batch >= 1 && batch <= 256, "decode_attention_fused: between 1 and 256 sequences");
    TL_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
               "decode_attention_fused: num_heads must be divisible by num_kv_heads");
    TL_REQUIRE(head_dim == 32 || head_dim == 64 || head_dim == 128, "decode_attention_fused: head_dim must be 32, 64 or 128");
    TL_REQUIRE(page_size > 0 && max_pages > 0 && max_context >= 0, "decode_attention_fused: bad page geometry");
    const size_t need = tl_decode_attention_fused_workspace_bytes(batch, num_heads, head_dim);
    TL_REQUIRE(workspace_dev && workspace_bytes >= need, "decode_attention_fused: workspace is missing or too small");
    tl_engine e;
    e.cfg.num_heads = num_heads;
    e.cfg.num_kv_heads = num_kv_heads;
    e.cfg.head_dim = head_dim;
    e.cfg.page_size = page_size;
    e.cfg.max_pages_per_seq = max_pages;
    e.cfg.rope_theta = rope_theta;
    e.cfg.rms_norm_eps = eps;
    e.stream = (hipStream_t)stream;
    e.block_table = const_cast<int32_t *>(block_table_dev);
    e.context_lens = const_cast<int32_t *>(context_lens_dev);
    e.rope_cur = (float2 *)workspace_dev;
    const size_t rc_bytes = align_up((size_t)batch * (head_dim / 2) * sizeof(float2), 256);
    e.attn_ws = (float *)((char *)workspace_dev + rc_bytes);
    e.attn_ws_bytes = workspace_bytes - rc_bytes;
    read_attention_knobs(&e);
    hipLaunchKernelGGL(rope_rows_kernel, dim3(batch), dim3(64), 0, e.stream, context_lens_dev, e.rope_cur, head_dim / 2, rope_theta);
    TL_CHECK_LAUNCH("decode_attention_fused rope");
    const SplitPlan sp = pick_decode_splits(&e, batch, std::max(1, max_context + 1));
    const int rc = engine_attention(&e, (const uint16_t *)qkv_dev, q_norm_dev, k_norm_dev, (uint16_t *)key_pages_dev,
                                    (uint16_t *)value_pages_dev, (uint16_t *)out_dev, batch, sp, nullptr, nullptr, nullptr, nullptr,
                                    key_scales_dev, value_scales_dev);
    e.rope_cur = nullptr;  // borrowed
    if (info) {
        info->n_splits = sp.n_splits;
        info->tokens_per_split = sp.tokens_per_split;
        info->heads_per_workgroup = sp.rq;
        info->launches = e.last_attn_launches;
    }
    return rc;
}

extern "C" int tl_decode_attention_fused(const void *qkv_dev, const void *q_norm_dev, const void *k_norm_dev,
                                         void *key_pages_dev, void *value_pages_dev, const int32_t *block_table_dev,
                                         const int32_t *context_lens_dev, void *out_dev, int batch, int num_heads,
                                         int num_kv_heads, int head_dim, int page_size, int max_pages, float rope_theta,
                                         float eps, int max_context, void *workspace_dev, size_t workspace_bytes,
                                         void *stream, tl_attention_info *info) {
    return decode_attention_fused_impl(qkv_dev, q_norm_dev, k_norm_dev, key_pages_dev, nullptr, value_pages_dev, nullptr, block_table_dev,
                                       context_lens_dev, out_dev, batch, num_heads, num_kv_heads, head_dim, page_size, max_pages, rope_theta,
                                       eps, max_context, workspace_dev, workspace_bytes, stream, info);
}

extern "C" int tl_decode_attention_fused_fp8(const void *qkv_dev, const void *q_norm_dev, const void *k_norm_dev, void *key_pages_dev,
                                             float *key_scales_dev, void *value_pages_dev, float *value_scales_dev,
                                             const int32_t *block_table_dev, const int32_t *context_lens_dev, void *out_dev, int batch,
                                             int num_heads, int num_kv_heads, int head_dim, int page_size, int max_pages,
                                             float rope_theta, float eps, int max_context, void *workspace_dev, size_t workspace_bytes,
                                             void *stream, tl_attention_info *info) {
    TL_REQUIRE(key_scales_dev && value_scales_dev, "decode_attention_fused_fp8: null scale pointer");
    TL_REQUIRE(head_dim == 128, "decode_attention_fused_fp8: FP8 pages need head_dim 128");
    return decode_attention_fused_impl(qkv_dev, q_norm_dev, k_norm_dev, key_pages_dev, key_scales_dev, value_pages_dev, value_scales_dev,
                                       block_table_dev, context_lens_dev, out_dev, batch, num_heads, num_kv_heads, head_dim, page_size,
                                       max_pages, rope_theta, eps, max_context, workspace_dev, workspace_bytes, stream, info);
}

// ---- host-only: the plans the decode path would pick (no device, no launch): what the CPU tests and a binding's dry run read -------
extern "C" int tl_decode_gemv_plan(int M, int rows, int cols, int *out5) {
    if (!out5 || M < 1 || rows <= 0 || cols <= 0) return 0;
    const Qmv3Plan pl = qmv3_plan(M, cols, rows);
    out5[0] = pl.MR, out5[1] = pl.KS, out5[2] = pl.CW, out5[3] = pl.LM, out5[4] = pl.blocks;
    return pl.ok ? 1 : 0;
}
extern "C" int tl_decode_gemv_variant_compiled(int MR, int KS, int CW, int LM) { return qmv3_variant_in_table(MR, KS, CW, LM) ? 1 : 0; }
extern "C" int tl_decode_batched_plan(int M, int rows, int cols, int *out6) {
    if (!out6 || M < 1 || rows <= 0 || cols <= 0) return 0;
    const Qmm6Plan pl = qmm6_plan(M, cols, rows);
    out6[0] = pl.MB, out6[1] = pl.GPW, out6[2] = pl.NSETS, out6[3] = pl.row_blocks, out6[4] = pl.wgs, out6[5] = pl.tiles_per_wg;
    return pl.ok ? 1 : 0;
}
extern "C" int tl_decode_batched_variant_compiled(int MB, int GPW) { return qmm6_variant_in_table(MB, GPW) ? 1 : 0; }
extern "C" int tl_decode_streaming_plan(int M, int rows, int cols, int *out4) {
    if (!out4 || M < 1 || rows <= 0 || cols <= 0) return 0;
    const Qmm7Plan pl = qmm7_plan(M, cols, rows);
    out4[0] = pl.MB, out4[1] = pl.T, out4[2] = pl.GPW, out4[3] = pl.wgs;
    return pl.ok ? 1 : 0;
}
extern "C" int tl_decode_streaming_variant_compiled(int T, int GPW) { return qmm7_variant_in_table(T, GPW) ? 1 : 0; }
extern "C" int tl_decode_attention_plan(int batch, int max_context, int num_heads, int num_kv_heads, int *out3) {
    if (!out3 || batch < 1 || max_context < 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0) return 0;
    tl_engine e;
    e.cfg.num_heads = num_heads;
    e.cfg.num_kv_heads = num_kv_heads;
    e.cfg.head_dim = 128;  // the plan of the Qwen3 head size on 128-token pages
    e.cfg.page_size = 128;
    read_attention_knobs(&e);
    const SplitPlan sp = pick_decode_splits(&e, batch, std::max(1, max_context + 1));
    out3[0] = sp.n_splits, out3[1] = sp.tokens_per_split, out3[2] = sp.rq;
    return 1;
}
