// Device kernels private to the decode engine (engine.hip).  See include/tinyllm_engine.h for the step
// structure; every kernel here keeps the reference's op boundaries as bf16 rounding points.
// (Kernels are `static`: attn_mfma.hip includes this header for the argument block and the prologue helpers of the decode attention
// and is compiled with its own MFMA flag; internal linkage keeps the two objects from defining the same symbols.)
#pragma once
#include <type_traits>
#include "common.h"
#include "kv8.h"

namespace tl {

constexpr float ENG_LOG2E = 1.44269504089f;
constexpr int AD_RQ = 4;  // most query heads of one GQA group handled per workgroup

// ---------------------------------------------------------------------------------------------
// Shared prologue: RMSNorm over one head row (D = 16*VD, lane t holds dims [t*VD, t*VD+VD)) rounded to bf16
// (FastRMSNorm, week2_kernels.metal:41-47), then non-traditional RoPE at `pos` rounded to bf16
// (FastRoPE, week2_kernels.metal:86-104; pair (d, d+D/2) lives in lane t^8).  Same expression order as
// rms_norm_kernel / rope_kernel (pointwise.hip) so the fused and op-by-op paths agree.
// ---------------------------------------------------------------------------------------------
template <int VD>
__device__ __forceinline__ void load_row(const uint16_t *src, float (&f)[VD]) {
    uint16_t raw[VD];
    if constexpr (VD == 8) {
        *reinterpret_cast<uint4 *>(raw) = *reinterpret_cast<const uint4 *>(src);
    } else if constexpr (VD == 4) {
        *reinterpret_cast<uint2 *>(raw) = *reinterpret_cast<const uint2 *>(src);
    } else if constexpr (VD == 2) {
        *reinterpret_cast<uint32_t *>(raw) = *reinterpret_cast<const uint32_t *>(src);
    } else {
#pragma unroll
        for (int i = 0; i < VD; ++i) raw[i] = src[i];
    }
#pragma unroll
    for (int i = 0; i < VD; ++i) f[i] = BF16::to_float(raw[i]);
}

template <int VD>
__device__ __forceinline__ void store_row(uint16_t *dst, const float (&f)[VD]) {
    uint16_t raw[VD];
#pragma unroll
    for (int i = 0; i < VD; ++i) raw[i] = BF16::from_float(f[i]);
    if constexpr (VD == 8) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(raw);
    } else if constexpr (VD == 4) {
        *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(raw);
    } else if constexpr (VD == 2) {
        *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(raw);
    } else {
#pragma unroll
        for (int i = 0; i < VD; ++i) dst[i] = raw[i];
    }
}

template <int VD>
__device__ __forceinline__ void rope_factors(int t, int pos, float base, float (&cs)[VD], float (&sn)[VD]) {
    constexpr int half = 8 * VD;
    const float lb = log2f(base);
#pragma unroll
    for (int i = 0; i < VD; ++i) {
        const int item = (t & 7) * VD + i;
        const float fp = -(float)item / (float)half;
        const float angle = (float)pos * exp2f(fp * lb);
        sincosf(angle, &sn[i], &cs[i]);
    }
}

template <int VD>
__device__ __forceinline__ void head_norm_rope(const uint16_t *src, const uint16_t *w, int t, float eps,
                                               const float (&cs)[VD], const float (&sn)[VD], float (&out)[VD]) {
    constexpr int D = 16 * VD;
    float f[VD], g[VD];
    load_row<VD>(src + t * VD, f);
    load_row<VD>(w + t * VD, g);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VD; ++i) ss += f[i] * f[i];
    ss = group16_sum(ss);
    const float inv = rsqrtf(ss / (float)D + eps);
#pragma unroll
    for (int i = 0; i < VD; ++i) {
        const float n = bf16_round(f[i] * inv * g[i]);
        const float partner = dpp_row_ror<8>(n);  // lane t ^ 8 of the 16-lane group
        const float r = (t < 8) ? (n * cs[i] - partner * sn[i]) : (n * cs[i] + partner * sn[i]);
        out[i] = bf16_round(r);
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE table: (cos, sin) of position * base^(-item / half) for every position the engine can hold, computed on the
// device with the SAME expression as rope_kernel (pointwise.hip) / the reference fast kernel
// (week2_kernels.metal:86-104), so table and on-the-fly values are bit-identical.  Decode then does no trig at all.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void rope_table_kernel(float2 *__restrict__ table, int max_pos, int half, float base) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)max_pos * half) return;
    const int pos = (int)(idx / half);
    const int item = (int)(idx - (long)pos * half);
    const float fp = -(float)item / (float)half;
    const float angle = (float)pos * exp2f(fp * log2f(base));
    float s, c;
    sincosf(angle, &s, &c);
    table[idx] = make_float2(c, s);
}

template <int VD>
__device__ __forceinline__ void rope_from_table(const float2 *__restrict__ row, int t, float (&cs)[VD], float (&sn)[VD]) {
    const float2 *p = row + (t & 7) * VD;
#pragma unroll
    for (int i = 0; i < VD; ++i) {
        const float2 v = p[i];
        cs[i] = v.x;
        sn[i] = v.y;
    }
}

// all-reduce over an aligned group of 16 lanes with DPP row rotations (4 VALU ops, no LDS crossbar traffic)
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float group16_allsum(float v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Decode attention with fused q/k-norm + RoPE + paged KV append (L = 1).
//   grid = (n_splits * n_row_chunks, Hkv, batch); workgroup = 16 groups x 16 lanes.
//   A workgroup owns one KV head, up to AD_RQ query heads of its GQA group and the FIXED token window
//   [split * C, split * C + C) of the context (C = tokens_per_split, chosen by the host from a power-of-two bucket,
//   so no address depends on the device-side context length — only the masks do).  A 16-lane group reads one
//   token's K row and V row as 16 x 16 B (coalesced 256 B); U tokens per group are in flight, and the next
//   window's page ids and K/V are requested before the current one is reduced.
//   Latency structure (decode attention at short context is pure latency): one round trip for
//   {context length, page ids, the q/k/v row, norm weights}, one dependent round trip for {K/V rows, RoPE table row,
//   the append slot}; every load is unconditional from a clamped address (hipcc waits at the join of any divergent
//   branch that contains a load).
//   The token being decoded never round-trips through HBM: its K/V come from registers (split 0) and are written to
//   the page by one group.  n_splits == 1 writes the output row, otherwise (m, l, acc) partials for the merge kernel (or the merging wo GEMV, qmv3.h).
//   reference semantics: paged_attention.metal:108-248 (decode), paged_cache_update :82-106,
//   qwen3_week3.py:63-86 for the op order.
// ---------------------------------------------------------------------------------------------
// row of split partials in global memory: D value sums + (max, sum) + 2 floats of padding, so that rows start on 16 bytes (the wo GEMV
// that merges them reads them with 16-byte loads); the LDS rows inside the kernels keep D + 2
constexpr int ATTN_WS_PAD = 4;
struct AttnDecodeArgs {
    const uint16_t *qkv;  // [batch, (Hq + 2 Hkv) * D]
    const uint16_t *q_norm_w, *k_norm_w;
    uint16_t *key_pages, *value_pages;  // this layer's pools [P, Hkv, page, D]
    const int32_t *block_table;         // [max_batch, max_pages]
    const int32_t *context_lens;        // [max_batch] tokens already cached (= position of the new token)
    uint16_t *out;                      // [batch, Hq * D]
    float *ws;                          // [batch * Hq, n_splits, D + 4]: D value sums, running max, running sum, 2 pad (16-byte rows)
    const float2 *rope_cur;             // [max_batch, D / 2] (cos, sin) of each slot's NEXT position, kept by step_end
    int page_size, max_pages, num_heads, num_kv_heads;
    float scale, eps, rope_base;
    int n_splits, n_row_chunks, tokens_per_split;
    int split_shift;  // log2(n_splits): the split count is a power of two
    int rep;          // query heads per KV head
    int page_shift;   // log2(page_size), or -1 when the page size is not a power of two (then: integer division)
    prof_t *prof;
    // attn_decode_fused_kernel<.., QP = true> (TL_ATTN_QKV_PARTIALS=1): the qkv projection ran as the K-sliced skinny matmul and
    // its slice reduction was NOT launched -- the rows arrive as qkv_slices fp32 planes [slice][batch][(Hq + 2 Hkv) D]
    // (plane stride qkv_plane elements), added here in slice order and rounded once, exactly as qmm3_reduce_kernel does.
    const float *qkv_partial;
    int qkv_slices;
    long qkv_plane;
    // KV8 instantiations (FP8 pages, kv8.h): key_pages / value_pages hold E4M3 codes, one byte per element of the same layout, and these
    // the rows' power-of-two scales [P, Hkv, page] -- folded into the scores and the softmax weights of the walk
    float *key_scales, *value_scales;
};

// a K or V chunk of a row as it travels from the page to the walk: 8 bf16 values, or (KV8) 8 codes + the row's scale
template <int VD>
struct Kv8Row {
    u32x2 c;
    float s;
};

// Scalar (wave-uniform) 32-bit load through the scalar cache, and the wait that makes its result usable.  The address must
// be wave-uniform.  Words read this way were written by EARLIER launches (scalar caches are invalidated at kernel start).
__device__ __forceinline__ void sload_i32(const int32_t *ptr, int &dst) {
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(dst) : "s"(ptr));
}
__device__ __forceinline__ void sload_wait(int &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void sload_wait(int &a, int &b2) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b2)); }
__device__ __forceinline__ void sload_wait(int &a, int &b2, int &c2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b2), "+s"(c2));
}
__device__ __forceinline__ void sload_wait(int &a, int &b2, int &c2, int &d2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b2), "+s"(c2), "+s"(d2));
}
__device__ __forceinline__ void sload_wait(int &a, int &b2, int &c2, int &d2, int &e2, int &f2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b2), "+s"(c2), "+s"(d2), "+s"(e2), "+s"(f2));
}

template <int VD>
struct alignas(2 * VD) RawRow {
    uint16_t v[VD];
};
template <int VD>
__device__ __forceinline__ void load_raw(const uint16_t *src, RawRow<VD> &r) {
    if constexpr (VD == 8) {
        *reinterpret_cast<uint4 *>(r.v) = *reinterpret_cast<const uint4 *>(src);
    } else if constexpr (VD == 4) {
        *reinterpret_cast<uint2 *>(r.v) = *reinterpret_cast<const uint2 *>(src);
    } else {
        *reinterpret_cast<uint32_t *>(r.v) = *reinterpret_cast<const uint32_t *>(src);
    }
}

// a row another launch of this step has written (the qkv projection's output): common.h, TL_COHERENT
template <int VD>
__device__ __forceinline__ void load_raw_act(const uint16_t *src, RawRow<VD> &r) {
    if constexpr (VD == 8) {
        *reinterpret_cast<u32x4 *>(r.v) = act_load(reinterpret_cast<const u32x4 *>(src));  // (clang vector types: HIP's uint4 is a class)
    } else if constexpr (VD == 4) {
        *reinterpret_cast<u32x2 *>(r.v) = act_load(reinterpret_cast<const u32x2 *>(src));
    } else {
        *reinterpret_cast<uint32_t *>(r.v) = act_load(reinterpret_cast<const uint32_t *>(src));
    }
}

template <int VD>
__device__ __forceinline__ void store_raw(uint16_t *dst, const RawRow<VD> &r) {
    if constexpr (VD == 8) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(r.v);
    } else if constexpr (VD == 4) {
        *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(r.v);
    } else {
        *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(r.v);
    }
}

// RQ = query heads of the GQA group handled per workgroup (1, 2 or 4): fewer heads per workgroup shorten the dependent
// VALU chain of a step at the price of re-reading the K/V window from L2 once per extra workgroup
// SP = the workgroup's token window lies inside ONE page (tokens_per_split divides page_size): the page id is then a
// scalar load that returns long before the vector round trip, and the K/V rows no longer wait for it
// IP = the window spans several pages, but every 16 U-token stage of the walk lies inside one (page size a power of two >= 16 U,
// windows start on stage boundaries): the stage's page id is one scalar word fetched a stage ahead, and a row address is
// (uniform row base of the stage) + (a lane offset fixed for the kernel) -- no per-row page lookups (2 U vector loads per lane and
// stage) and no per-row 64-bit address chains.  The long-context plan (512-token windows over 128-token pages) runs this way.
// QP = the new token's q / k / v rows are read as fp32 slice partials of the skinny matmul (AttnDecodeArgs::qkv_partial) instead
// of bf16 rows: the (2 + RQ) D values a workgroup needs are 4-column chunks shared out over its threads (one global round
// trip, all slices of a chunk in flight together), summed in slice order, rounded to bf16 and handed to every 16-lane group
// through LDS -- the qkv projection's slice-reduction launch (a dependent phase of ~3.3 us per layer) is gone.
// KV8 = FP8 pages (head dimension 128): a lane requests 8 bytes of codes per row chunk and the row's scale; scores and weights take the
// scales as one multiply per (token, head) -- bit for bit this kernel's arithmetic over the dequantised rows (kv8.h).  The token being
// decoded is quantised in registers: this step attends to what the page will hold.
template <int VD, int U, int RQ, bool SP, bool IP = false, bool QP = false, bool KV8 = false>
static __global__ __launch_bounds__(256) void attn_decode_fused_kernel(const AttnDecodeArgs p) {
    static_assert(!KV8 || VD == 8, "FP8 pages: head dimension 128");
    using KVRow = typename std::conditional<KV8, Kv8Row<VD>, RawRow<VD>>::type;
    constexpr int D = 16 * VD;
    constexpr int STRIDE = D + 2;
    extern __shared__ __attribute__((aligned(16))) float psm[];  // [16][RQ][STRIDE] (+ QP: [(2 + RQ)][D] bf16 staged rows)
    const prof_t prof_t0 = prof_begin(p.prof);
    // wave-uniform indices stay on the scalar unit (readfirstlane: hipcc otherwise parks blockIdx in VGPRs after the
    // profiling branch and emulates every division below on the VALU, in front of the first load)
    const int bx = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const int split = bx & (p.n_splits - 1);
    const int chunk = bx >> p.split_shift;
    const int kvh = __builtin_amdgcn_readfirstlane(blockIdx.y);
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.z);
    const int Hq = p.num_heads, Hkv = p.num_kv_heads;
    const int rep = p.rep;
    // (IP implies a power-of-two page: no division branch, which would be a control-flow join inside the stage loop)
    auto page_of = [&](int tok) { return (IP || p.page_shift >= 0) ? (tok >> p.page_shift) : tok / p.page_size; };
    const int g = threadIdx.x >> 4;
    const int t = threadIdx.x & 15;
    const int32_t *brow = p.block_table + (long)b * p.max_pages;
    const uint16_t *row = p.qkv + (long)b * (Hq + 2 * Hkv) * D;
    const float scale_log2 = p.scale * ENG_LOG2E;
    const int C = p.tokens_per_split;
    const int t_begin = split * C;
    const int n_it = (C + 16 * U - 1) / (16 * U);

    // ---- round trip 1: everything whose address is known at launch ---------------------------------------------
    // Wave-uniform words (context length, first page id, and with SP the window's page id) come through the scalar cache:
    // explicit s_load, because hipcc turns such loads into "vector load + wait + readfirstlane" at the top of the kernel.
    // They are waited for (lgkmcnt) only after the vector loads of this round trip have been issued.
    int ctx, first_page, pid_s = 0;
    sload_i32(p.context_lens + b, ctx);
    sload_i32(brow, first_page);
    if constexpr (SP) sload_i32(brow + min(page_of(t_begin), p.max_pages - 1), pid_s);
    int pid[U], pid_next[U];
    int pg_nxt = 0, pg_new = 0;  // IP: page id of the next stage / of the one after it (scalar)
    if constexpr (IP) {
        sload_i32(brow + min(page_of(t_begin), p.max_pages - 1), pid_s);
        sload_i32(brow + min(page_of(t_begin + 16 * U), p.max_pages - 1), pg_nxt);
    }
    if constexpr (!SP && !IP) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tok = t_begin + u * 16 + g;
            pid[u] = brow[min(page_of(tok), p.max_pages - 1)];
            pid_next[u] = brow[min(page_of(tok + 16 * U), p.max_pages - 1)];
        }
    }
    RawRow<VD> kraw_new, vraw_new, qraw[RQ], qw, kw;
    if constexpr (!QP) {
        load_raw_act<VD>(row + (long)(Hq + kvh) * D + t * VD, kraw_new);
        load_raw_act<VD>(row + (long)(Hq + Hkv + kvh) * D + t * VD, vraw_new);
    }
    load_raw<VD>(p.q_norm_w + t * VD, qw);
    load_raw<VD>(p.k_norm_w + t * VD, kw);
    // QP: staged rows 0 = k, 1 = v, 2 + r = query head r of this workgroup; chunk c of the workgroup = 4 columns of one row
    constexpr int QP_ROWS = 2 + RQ;
    constexpr int QP_CHUNKS = QP_ROWS * D / 4;
    constexpr int QP_PER = (QP_CHUNKS + 255) / 256;  // chunks per thread
    constexpr int QP_INFLIGHT = 4;                   // slices of a chunk in flight together
    f32x4 qp_x[QP ? QP_PER : 1][QP ? QP_INFLIGHT : 1];
    auto qp_col = [&](int ch) {  // first column (inside a [batch, (Hq + 2 Hkv) D] plane row) of chunk ch
        const int prow = ch / (D / 4);
        const int c4 = ch - prow * (D / 4);
        const int head = prow == 0 ? Hq + kvh : (prow == 1 ? Hq + Hkv + kvh : kvh * rep + min(chunk * RQ + (prow - 2), rep - 1));
        return (long)head * D + c4 * 4;
    };
    if constexpr (QP) {
        const float *prow_base = p.qkv_partial + (long)b * (Hq + 2 * Hkv) * D;
#pragma unroll
        for (int j = 0; j < QP_PER; ++j) {
            const int ch = min((int)threadIdx.x + j * 256, QP_CHUNKS - 1);
            const float *src = prow_base + qp_col(ch);
#pragma unroll
            for (int s = 0; s < QP_INFLIGHT; ++s)
                qp_x[j][s] = *reinterpret_cast<const f32x4 *>(src + (long)min(s, p.qkv_slices - 1) * p.qkv_plane);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            const int hq = min(chunk * RQ + r, rep - 1);
            load_raw_act<VD>(row + (long)(kvh * rep + hq) * D + t * VD, qraw[r]);
        }
    }

    float cs[VD], sn[VD];
    rope_from_table<VD>(p.rope_cur + (long)b * (D / 2), t, cs, sn);
    __builtin_amdgcn_sched_barrier(0);
    sload_wait(ctx, first_page, pid_s, pg_nxt);
    const bool live = first_page >= 0;  // a sequence always owns its first page; idle slots have an all -1 row and produce zeros
    if constexpr (SP) {
#pragma unroll
        for (int u = 0; u < U; ++u) pid[u] = pid_next[u] = pid_s;
    }

    // ---- round trip 2: addresses that depend on the page ids (K/V rows) or on the context length (append slot) ------
    const int wp = page_of(ctx);
    const int wslot = ctx - wp * p.page_size;
    int wpage;
    sload_i32(brow + min(wp, p.max_pages - 1), wpage);  // waited for at the very end of the kernel
    KVRow kr[U], vr[U], kr_next[U], vr_next[U];
    bool ok[U];
    // (prow = the row's index in the page layout, off = the lane's element offset in it)
    auto load_kv = [&](long prow, long off, KVRow &kk, KVRow &vv) {
        if constexpr (KV8) {
            kk.c = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint8_t *>(p.key_pages) + off);
            vv.c = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint8_t *>(p.value_pages) + off);
            kk.s = p.key_scales[prow];
            vv.s = p.value_scales[prow];
        } else {
            load_raw<VD>(p.key_pages + off, kk);
            load_raw<VD>(p.value_pages + off, vv);
        }
    };
    auto issue_kv = [&](int base, const int (&ids)[U], KVRow (&kk)[U], KVRow (&vv)[U], bool (&valid)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tok = base + u * 16 + g;
            const int lp = page_of(tok);
            const int slot = tok - lp * p.page_size;
            valid[u] = tok < ctx && tok < t_begin + C && lp < p.max_pages && ids[u] >= 0;
            const long prow = ((long)max(ids[u], 0) * Hkv + kvh) * p.page_size + slot;
            load_kv(prow, prow * D + t * VD, kk[u], vv[u]);
        }
    };
    const int lane_row = g * D + t * VD;  // IP: element offset of this lane's 16 bytes inside a stage's first 16 rows
    auto issue_kv_stage = [&](int base, int pg, KVRow (&kk)[U], KVRow (&vv)[U], bool (&valid)[U]) {
        const int lp = page_of(base);  // uniform
        const bool page_ok = lp < p.max_pages && pg >= 0;
        const long prow0 = ((long)max(pg, 0) * Hkv + kvh) * p.page_size + (base - lp * p.page_size);  // uniform
        const long rowbase = prow0 * D;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tok = base + u * 16 + g;
            valid[u] = tok < ctx && tok < t_begin + C && page_ok;
            load_kv(prow0 + g + u * 16, rowbase + lane_row + u * 16 * D, kk[u], vv[u]);
        }
    };
    if constexpr (IP) issue_kv_stage(t_begin, pid_s, kr, vr, ok);
    else issue_kv(t_begin, pid, kr, vr, ok);

    if constexpr (QP) {
        // slice partials -> bf16 rows in LDS -> every 16-lane group's registers (the K/V rows requested above stay in flight:
        // the barrier waits for LDS traffic only)
        uint16_t *qs = reinterpret_cast<uint16_t *>(psm + 16 * RQ * STRIDE);
        const float *prow_base = p.qkv_partial + (long)b * (Hq + 2 * Hkv) * D;
#pragma unroll
        for (int j = 0; j < QP_PER; ++j) {
            const int chu = (int)threadIdx.x + j * 256;
            const int ch = min(chu, QP_CHUNKS - 1);
            f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < QP_INFLIGHT; ++s)
                if (s < p.qkv_slices) {  // uniform
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) acc4[e2] += qp_x[j][s][e2];
                }
            for (int s = QP_INFLIGHT; s < p.qkv_slices; ++s) {  // more than QP_INFLIGHT slices (not planned for qkv shapes): dependent loads
                const f32x4 x4 = *reinterpret_cast<const f32x4 *>(prow_base + qp_col(ch) + (long)s * p.qkv_plane);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) acc4[e2] += x4[e2];
            }
            if (chu < QP_CHUNKS) {
                uint2 packed;
                packed.x = BF16::pack2(acc4[0], acc4[1]);  // round-to-nearest-even, as BF16::from_float in qmm3_reduce_kernel
                packed.y = BF16::pack2(acc4[2], acc4[3]);
                *reinterpret_cast<uint2 *>(qs + ch * 4) = packed;
            }
        }
        __syncthreads();
        load_raw<VD>(qs + 0 * D + t * VD, kraw_new);
        load_raw<VD>(qs + 1 * D + t * VD, vraw_new);
#pragma unroll
        for (int r = 0; r < RQ; ++r) load_raw<VD>(qs + (2 + r) * D + t * VD, qraw[r]);
    }

    // ---- prologue math while the K/V rows are in flight ---------------------------------------------------------------
    auto norm_rope = [&](const RawRow<VD> &x, const RawRow<VD> &w, float (&out)[VD]) {
        float f[VD];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            f[i] = BF16::to_float(x.v[i]);
            ss += f[i] * f[i];
        }
        ss = group16_allsum(ss);
        const float inv = rsqrtf(ss / (float)D + p.eps);
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            const float n = bf16_round(f[i] * inv * BF16::to_float(w.v[i]));
            const float partner = row_ror<8>(n);  // lane t ^ 8 of the same 16-lane group
            const float r2 = (t < 8) ? (n * cs[i] - partner * sn[i]) : (n * cs[i] + partner * sn[i]);
            out[i] = bf16_round(r2);
        }
    };
    float k_new[VD], v_new[VD];
    norm_rope(kraw_new, kw, k_new);
#pragma unroll
    for (int i = 0; i < VD; ++i) v_new[i] = BF16::to_float(vraw_new.v[i]);
    // KV8: the new rows as the page will hold them (every 16-lane group holds the whole row: the same codes and scale in each)
    u32x2 k_new_c = u32x2{0u, 0u}, v_new_c = u32x2{0u, 0u};
    float k_new_s = 1.f, v_new_s = 1.f;
    if constexpr (KV8) {
        float kd[8], vd[8];
        kv8_quantize_row16(k_new, k_new_c, k_new_s, kd);
        kv8_quantize_row16(v_new, v_new_c, v_new_s, vd);
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            k_new[i] = kd[i];
            v_new[i] = vd[i];
        }
    }
    float qv[RQ][VD], acc[RQ][VD], m[RQ], l[RQ];
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
        float qn[VD];
        norm_rope(qraw[r], qw, qn);
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            qv[r][i] = qn[i] * scale_log2;
            acc[r][i] = 0.f;
        }
        m[r] = -1e30f;
        l[r] = 0.f;
    }

    // ---- walk the window: request the next batch, then reduce the current one ------------------------------------------
    // Two register sets for the K/V rows, used alternately (the loop is unrolled by two through this lambda): handing the
    // prefetched rows over by copying them cost 96 v_mov per stage.
    bool ok_next[U];
    // The next stage's K/V rows are requested first -- UNCONDITIONALLY (the last stage requests itself again: rows that have just been
    // read).  A branch around those loads is a control-flow join, and at a join hipcc's wait bookkeeping loses the issue order of pending
    // loads: the first use of THIS stage's rows then waited for the rows just requested for the next one (round 4, found in the ISA:
    // vmcnt(7) .. vmcnt(0) behind the prefetch) -- a stage cost one memory round trip whatever was prefetched.
    auto walk_stage = [&](int it, KVRow(&kc)[U], KVRow(&vc)[U], bool(&okc)[U], KVRow(&kn)[U], KVRow(&vn)[U], bool(&okn)[U]) {
        const bool more = it + 1 < n_it;
        const int nx = min(it + 1, n_it - 1);
        if constexpr (IP) {
            issue_kv_stage(t_begin + nx * 16 * U, pg_nxt, kn, vn, okn);  // (last stage: any valid rows; the result is not used)
            if (more) sload_i32(brow + min(page_of(t_begin + (it + 2) * 16 * U), p.max_pages - 1), pg_new);  // scalar: waited for at the end of this stage
        } else {
            issue_kv(t_begin + nx * 16 * U, pid_next, kn, vn, okn);
            if constexpr (!SP) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    pid_next[u] = brow[min(page_of(t_begin + (it + 2) * 16 * U + u * 16 + g), p.max_pages - 1)];
            }
        }
        // V rows of masked tokens are ZERO, not "whatever the page holds times a zero weight": 0 x NaN is NaN, and a recycled or
        // caller-provided page may hold anything behind the context
        float vf[U][VD];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (KV8) {
                float raw[8];
                kv8_unpack8(vc[u].c, raw);
#pragma unroll
                for (int i = 0; i < VD; ++i) vf[u][i] = (okc[u] && live) ? raw[i] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < VD; ++i) vf[u][i] = (okc[u] && live) ? BF16::to_float(vc[u].v[i]) : 0.f;
            }
        }
        float sc[RQ][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[VD];
            if constexpr (KV8) {
                kv8_unpack8(kc[u].c, kf);
            } else {
#pragma unroll
                for (int i = 0; i < VD; ++i) kf[i] = BF16::to_float(kc[u].v[i]);
            }
#pragma unroll
            for (int r = 0; r < RQ; ++r) {
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < VD; ++i) part += qv[r][i] * kf[i];
                sc[r][u] = part;
            }
        }
        // the RQ*U partial dot products go through the four DPP rotations TOGETHER, step by step: reduced one value at a time,
        // each add waits for the one before it (a DPP operand needs two wait states behind the VALU write: the compiler filled
        // them with 128 s_nop per stage)
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int u = 0; u < U; ++u) sc[r][u] += row_ror<8>(sc[r][u]);
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int u = 0; u < U; ++u) sc[r][u] += row_ror<4>(sc[r][u]);
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int u = 0; u < U; ++u) sc[r][u] += row_ror<2>(sc[r][u]);
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                sc[r][u] += row_ror<1>(sc[r][u]);
                if constexpr (KV8) sc[r][u] = (okc[u] && live) ? sc[r][u] * kc[u].s : -1e30f;  // the K row's scale: a power of two
                else sc[r][u] = (okc[u] && live) ? sc[r][u] : -1e30f;
            }
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            float nm = m[r];
#pragma unroll
            for (int u = 0; u < U; ++u) nm = fmaxf(nm, sc[r][u]);
            const float of = exp2_hw(m[r] - nm);
            m[r] = nm;
            l[r] *= of;
#pragma unroll
            for (int i = 0; i < VD; ++i) acc[r][i] *= of;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pw = (okc[u] && live) ? exp2_hw(sc[r][u] - nm) : 0.f;
                l[r] += pw;
                float pwv = pw;
                if constexpr (KV8) pwv = (okc[u] && live) ? pw * vc[u].s : 0.f;  // the V row's scale rides on the weight
#pragma unroll
                for (int i = 0; i < VD; ++i) acc[r][i] += pwv * vf[u][i];
            }
        }
        if constexpr (IP) {
            if (more) {  // scalar loads only inside: both sides of this join carry the same pending vector loads
                sload_wait(pg_new);
                pg_nxt = pg_new;
            }
        }
    };
    for (int it = 0; it < n_it; it += 2) {
        walk_stage(it, kr, vr, ok, kr_next, vr_next, ok_next);
        if (it + 1 < n_it) walk_stage(it + 1, kr_next, vr_next, ok_next, kr, vr, ok);
    }
    // the token being decoded (position ctx), straight from registers
    if (split == 0) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < VD; ++i) part += qv[r][i] * k_new[i];
            const float score = group16_allsum(part);
            if (live && g == 0) {
                const float nm = fmaxf(m[r], score);
                const float of = exp2_hw(m[r] - nm);
                const float sf = exp2_hw(score - nm);
                l[r] = l[r] * of + sf;
#pragma unroll
                for (int i = 0; i < VD; ++i) acc[r][i] = acc[r][i] * of + sf * v_new[i];
                m[r] = nm;
            }
        }
    }

    // merge the 16 groups through LDS
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
        float *dst = psm + ((long)g * RQ + r) * STRIDE;
#pragma unroll
        for (int i = 0; i < VD; ++i) dst[t * VD + i] = acc[r][i];
        if (t == 0) {
            dst[D] = m[r];
            dst[D + 1] = l[r];
        }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < RQ * D; item += 256) {
        const int r = item / D;
        const int d = item - r * D;
        const int hq = chunk * RQ + r;
        if (hq >= rep) continue;
        float gm = -1e30f;
#pragma unroll
        for (int j = 0; j < 16; ++j) gm = fmaxf(gm, psm[((long)j * RQ + r) * STRIDE + D]);
        float gl = 0.f, vs = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *src = psm + ((long)j * RQ + r) * STRIDE;
            const float f = exp2_hw(src[D] - gm);
            gl += src[D + 1] * f;
            vs += src[d] * f;
        }
        const long orow = (long)b * Hq + kvh * rep + hq;
        if (p.n_splits == 1) {
            act_store(&p.out[orow * D + d], BF16::from_float(gl == 0.f ? 0.f : vs / gl));
        } else {
            float *w = p.ws + (orow * p.n_splits + split) * (D + ATTN_WS_PAD);
            act_store(&w[d], vs);
            if (d == 0) {
                act_store(&w[D], gm);
                act_store(&w[D + 1], gl);
            }
        }
    }
    // append the new token's K (normed + roped) and V to the slot's page: last, so the page-id lookup that depends on
    // the context length never sits on the critical path
    sload_wait(wpage);
    if (live && wp < p.max_pages && wpage >= 0 && split == 0 && chunk == 0 && g == 0) {
        const long prow = ((long)wpage * Hkv + kvh) * p.page_size + wslot;
        const long off = prow * D + t * VD;
        if constexpr (KV8) {
            *reinterpret_cast<u32x2 *>(reinterpret_cast<uint8_t *>(p.key_pages) + off) = k_new_c;
            *reinterpret_cast<u32x2 *>(reinterpret_cast<uint8_t *>(p.value_pages) + off) = v_new_c;
            if (t == 0) {
                p.key_scales[prow] = k_new_s;
                p.value_scales[prow] = v_new_s;
            }
        } else {
            store_row<VD>(p.key_pages + off, k_new);
            store_raw<VD>(p.value_pages + off, vraw_new);
        }
    }
    prof_end(p.prof, prof_t0);
}

// partials [rows, NS, D+2] -> out [rows, D]; all loads of a thread are independent and issued together
template <int NS>
static __global__ __launch_bounds__(128) void attn_merge_kernel(const float *__restrict__ ws, uint16_t *__restrict__ out,
                                                         int D, prof_t *prof) {
    const prof_t prof_t0 = prof_begin(prof);
    const long orow = blockIdx.x;
    const int stride = D + ATTN_WS_PAD;
    const float *base = ws + orow * NS * stride;
    const int d = threadIdx.x < D ? threadIdx.x : 0;
    float ms[NS], ls[NS], vs[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        ms[s2] = act_load(base + s2 * stride + D);
        ls[s2] = act_load(base + s2 * stride + D + 1);
        vs[s2] = act_load(base + s2 * stride + d);
    }
    float gm = -1e30f;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) gm = fmaxf(gm, ms[s2]);
    float gl = 0.f, acc = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const float f = exp2_hw(ms[s2] - gm);
        gl += ls[s2] * f;
        acc += vs[s2] * f;
    }
    if ((int)threadIdx.x < D) act_store(&out[orow * D + d], BF16::from_float(gl == 0.f ? 0.f : acc / gl));
    prof_end(prof, prof_t0);
}

// The merge for 16 and more splits (contexts from ~1k tokens): grid = (rows, D / 32), block = 32 output columns x 8 split
// groups.  One workgroup per row with every split's loads issued by 128 threads took 8 us at 64 splits (r02 bench, 8k and
// 32k contexts: 288 us per step): only batch x heads CUs took part, each pulling 33 KB through its ~32 KiB-in-flight limit.
// Here a row's partials are read by 4 workgroups x 8 split groups, 8 loads in flight per thread; every group folds its
// splits (s = sg, sg + 8, ...) with the online-softmax update in index order, and the 8 groups meet in LDS in group order:
// deterministic.
static __global__ __launch_bounds__(256) void attn_merge_cols_kernel(const float *__restrict__ ws, uint16_t *__restrict__ out, int D,
                                                              int n_splits, prof_t *prof) {
    __shared__ float s_m[8][32], s_l[8][32], s_a[8][32];
    const prof_t prof_t0 = prof_begin(prof);
    const long orow = blockIdx.x;
    const int c = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int d = min((int)blockIdx.y * 32 + c, D - 1);
    const int stride = D + ATTN_WS_PAD;
    const float *base = ws + orow * n_splits * stride;
    float m = -1e30f, l = 0.f, acc = 0.f;
    for (int s0 = sg; s0 < n_splits; s0 += 64) {
        float ms[8], ls[8], vs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float *row = base + (size_t)min(s0 + 8 * j, n_splits - 1) * stride;
            ms[j] = act_load(row + D);
            ls[j] = act_load(row + D + 1);
            vs[j] = act_load(row + d);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (s0 + 8 * j < n_splits) {  // uniform per split group; no load inside
                const float nm = fmaxf(m, ms[j]);
                const float f0 = exp2_hw(m - nm), f1 = exp2_hw(ms[j] - nm);
                l = l * f0 + ls[j] * f1;
                acc = acc * f0 + vs[j] * f1;
                m = nm;
            }
        }
    }
    s_m[sg][c] = m;
    s_l[sg][c] = l;
    s_a[sg][c] = acc;
    __syncthreads();
    if (sg == 0) {
        float gm = s_m[0][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) gm = fmaxf(gm, s_m[j][c]);
        float gl = 0.f, ga = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = exp2_hw(s_m[j][c] - gm);
            gl += s_l[j][c] * f;
            ga += s_a[j][c] * f;
        }
        if ((int)blockIdx.y * 32 + c < D) act_store(&out[orow * D + d], BF16::from_float(gl == 0.f ? 0.f : ga / gl));
    }
    prof_end(prof, prof_t0);
}

// ---------------------------------------------------------------------------------------------
// End of step: greedy argmax over bf16 logits (first maximum wins, like argmax), record the id, advance
// the slot, and dequantize the id's embedding row into the next step's input activation.
//   grid = rows; block = 1024.  Block i reads logits row i and serves slot slot0 + i.
//   reference: mx.argmax(logits[:, -1]) (benches/bench.py:234-243) + QuantizedEmbedding (embedding.py:38-54).
// ---------------------------------------------------------------------------------------------
struct StepEndArgs {
    const uint16_t *logits;  // [rows, vocab]
    int vocab;
    int slot0;
    int32_t *tokens;        // [max_batch] pending input token per slot
    int32_t *context_lens;  // [max_batch]
    const int32_t *live;    // [max_batch] 1 = slot holds a sequence
    int32_t *produced;      // [max_batch] number of ids recorded so far
    int32_t *ring;          // [max_batch, ring_cap]
    int ring_cap;
    int advance;  // 1: context_lens[slot] += 1 (decode); 0: prefill sets it on the host side
    // next-step embedding
    const uint32_t *emb_w;
    const uint16_t *emb_s, *emb_b;
    uint16_t *x;  // [max_batch, hidden], row = slot
    int hidden;
    // RoPE factors of the slot's next position (read by the next step's attention kernels)
    const float2 *rope_table;
    float2 *rope_cur;
    int rope_positions, rope_half;
    // optional: [max_batch][8] partial sums of squares of the embedded row (entry 0; the rest zero) for the fused RMSNorm of
    // the next step's skinny QKV matmul (qmm3.h)
    float *ss_out;
    prof_t *prof;
    // optional: per 16-logit tile (largest bf16 logit, lowest index holding it), left by the lm_head GEMV's epilogue (qmv3.h tile_max):
    // [rows][tiles] pairs; the greedy id is then picked from `tiles` pairs and the logits row is not read again
    const f32x2 *tile_max;
    int tiles;
};

static __global__ __launch_bounds__(1024) void step_end_kernel(const StepEndArgs p) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_token, s_ctx;
    const prof_t prof_t0 = prof_begin(p.prof);
    const int i = blockIdx.x;
    const int slot = p.slot0 + i;
    const uint16_t *lg = p.logits + (long)i * p.vocab;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    int vec_end = ((uintptr_t)lg % 16 == 0) ? (p.vocab & ~7) : 0;
    int scalar_from = vec_end;
    if (p.tile_max) {  // uniform: 9,496 pairs instead of 151,936 logits; same rule (strictly greater wins, the lower index on a tie)
        const f32x2 *tm = p.tile_max + (long)i * p.tiles;
        constexpr int TM_NB = 10;
        for (int t0 = threadIdx.x; t0 < p.tiles; t0 += 1024 * TM_NB) {
            f32x2 pr[TM_NB];
#pragma unroll
            for (int j = 0; j < TM_NB; ++j) pr[j] = act_load(tm + min(t0 + j * 1024, p.tiles - 1));  // (the lm_head launch of this step wrote them)
#pragma unroll
            for (int j = 0; j < TM_NB; ++j) {
                if (t0 + j * 1024 >= p.tiles) continue;
                const float v = pr[j][0];
                const int idx = pr[j][1] < 1.0e30f ? (int)pr[j][1] : 0x7fffffff;
                if (v > best || (v == best && idx < best_i)) {
                    best = v;
                    best_i = idx;
                }
            }
        }
        vec_end = 0;
        scalar_from = p.vocab;  // nothing of the row itself is read
    }
    // ONE workgroup reads the whole row (304 KB at Qwen3's vocabulary): a plain loop is a chain of 19 dependent L2 round trips
    // (12 us in the step profile).  The loads go out ten 16-byte chunks at a time, from clamped addresses, and are looked at after:
    // two round trips.
    constexpr int SE_NB = 10;
    for (int c0 = threadIdx.x * 8; c0 < vec_end; c0 += 1024 * 8 * SE_NB) {
        u32x4 rawv[SE_NB];
#pragma unroll
        for (int j = 0; j < SE_NB; ++j) rawv[j] = act_load(reinterpret_cast<const u32x4 *>(lg + min(c0 + j * 8192, vec_end - 8)));
#pragma unroll
        for (int j = 0; j < SE_NB; ++j) {
            const int c = c0 + j * 8192;
            if (c >= vec_end) continue;
            uint16_t raw[8];
            *reinterpret_cast<u32x4 *>(raw) = rawv[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = BF16::to_float(raw[e]);
                if (v > best) {  // strictly greater: the earliest index of a tie stays
                    best = v;
                    best_i = c + e;
                }
            }
        }
    }
    for (int c = scalar_from + threadIdx.x; c < p.vocab; c += 1024) {
        const float v = BF16::to_float(act_load(lg + c));
        if (v > best || (v == best && c < best_i)) {
            best = v;
            best_i = c;
        }
    }
    {   // wave-wide (maximum, lowest index that holds it) by DPP rotations instead of twelve ds_bpermute round trips: indices are
        // below 2^24, exact as floats, so the lowest index is -max(-index) over the lanes that hold the maximum
        const float m = wave_max(best);
        const float cand = (best == m && best_i != 0x7fffffff) ? (float)best_i : 3.0e38f;
        const float lowest = -wave_max(-cand);
        best = m;
        best_i = lowest < 1.0e30f ? (int)lowest : 0x7fffffff;
    }
    if ((threadIdx.x & 63) == 0) {
        s_val[threadIdx.x >> 6] = best;
        s_idx[threadIdx.x >> 6] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = s_val[0];
        int bi = s_idx[0];
        for (int w = 1; w < 16; ++w) {
            if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) {
                bv = s_val[w];
                bi = s_idx[w];
            }
        }
        if (bi < 0 || bi >= p.vocab) bi = 0;  // all-NaN / -inf row
        s_token = bi;
        int ctx_now = p.context_lens[slot];
        if (p.live[slot]) {
            p.tokens[slot] = bi;
            const int n = p.produced[slot];
            p.ring[(long)slot * p.ring_cap + (n % p.ring_cap)] = bi;
            p.produced[slot] = n + 1;
            if (p.advance) p.context_lens[slot] = ++ctx_now;
        }
        s_ctx = ctx_now;
    }
    __syncthreads();
    if ((int)threadIdx.x < p.rope_half) {
        const int pos = min(s_ctx, p.rope_positions - 1);  // the slot's NEXT position
        p.rope_cur[(long)slot * p.rope_half + threadIdx.x] = p.rope_table[(long)pos * p.rope_half + threadIdx.x];
    }
    const int token = s_token;
    const int words = p.hidden / 8;
    const int groups = p.hidden / 128;
    float sumsq = 0.f;
    for (int w = threadIdx.x; w < words; w += 1024) {
        const uint32_t packed = p.emb_w[(long)token * words + w];
        const float scale = BF16::to_float(p.emb_s[(long)token * groups + w / 16]);
        const float bias = BF16::to_float(p.emb_b[(long)token * groups + w / 16]);
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = BF16::from_float((float)((packed >> (4 * e)) & 0xfu) * scale + bias);
            const float v = BF16::to_float(o[e]);
            sumsq += v * v;
        }
        *reinterpret_cast<uint4 *>(p.x + (long)slot * p.hidden + w * 8) = *reinterpret_cast<const uint4 *>(o);
    }
    if (p.ss_out) {  // uniform.  s_val is free again: its last readers ran before the barrier above
        const float ws = wave_sum(sumsq);
        if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = ws;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int w = 0; w < 16; ++w) tot += s_val[w];
            p.ss_out[(long)slot * 8] = tot;
            for (int i = 1; i < 8; ++i) p.ss_out[(long)slot * 8 + i] = 0.f;
        }
    }
    prof_end(p.prof, prof_t0);
}

// Greedy id of every logits row (speculative verification): same tie rule as step_end_kernel (first maximum wins).
//   grid = rows, block = 1024
static __global__ __launch_bounds__(1024) void argmax_rows_kernel(const uint16_t *__restrict__ logits, int vocab,
                                                           int32_t *__restrict__ ids) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    const uint16_t *lg = logits + (long)blockIdx.x * vocab;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int c = threadIdx.x; c < vocab; c += 1024) {  // ascending per thread: the earliest index of a tie stays
        const float v = BF16::to_float(lg[c]);
        if (v > best) {
            best = v;
            best_i = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) {
            best = ov;
            best_i = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_val[threadIdx.x >> 6] = best;
        s_idx[threadIdx.x >> 6] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = s_val[0];
        int bi = s_idx[0];
        for (int w = 1; w < 16; ++w) {
            if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) {
                bv = s_val[w];
                bi = s_idx[w];
            }
        }
        ids[blockIdx.x] = (bi < 0 || bi >= vocab) ? 0 : bi;
    }
}

// tl_engine_check_step (test-only): what did the launch before this one store?  Compares a region of hand-over buffers word by word with
// the shadow copy taken after the previous launch; a 2-byte element that changed was written by that launch -- a second write of the
// step to the same element is counted (the AQL replay route reads these addresses without cache maintenance: "written once per step"
// is what makes that correct, csrc/aql.h) -- and the shadow follows.  report: [0] double writes, [1] first offending launch (min),
// [2] region of the first offence, [3] elements written, [4 .. 5] 64-bit element offset of one offence of the first offending launch.
static __global__ __launch_bounds__(256) void written_once_check_kernel(const uint32_t *__restrict__ region, uint32_t *__restrict__ shadow,
                                                                         uint8_t *__restrict__ written, size_t words, int launch_idx, int region_id,
                                                                         unsigned long long *report) {
    unsigned long long doubles = 0, fresh = 0, at = ~0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        const uint32_t cur = region[i], old = shadow[i];
        if (cur == old) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (((cur ^ old) >> (16 * h)) & 0xffffu) {
                if (written[2 * i + h]) ++doubles, at = 2 * i + h;
                else written[2 * i + h] = 1, ++fresh;
            }
        }
        shadow[i] = cur;
    }
    if (fresh) atomicAdd(&report[3], fresh);
    if (doubles) {
        atomicAdd(&report[0], doubles);
        const unsigned long long before = atomicMin(&report[1], (unsigned long long)launch_idx);
        if ((unsigned long long)launch_idx <= before) report[2] = (unsigned long long)region_id, report[4] = at;
    }
}

// (start, end) of one instrumented launch: min over workgroup starts, max over ends; clears the buffer.
static __global__ __launch_bounds__(1024) void prof_reduce_kernel(prof_t *buf, int n_wg, prof_t *out_pair) {
    __shared__ prof_t s_min[16], s_max[16];
    prof_t lo = ~0ull, hi = 0ull;
    for (int i = threadIdx.x; i < n_wg; i += 1024) {
        lo = min(lo, buf[2 * i]);
        hi = max(hi, buf[2 * i + 1]);
        buf[2 * i] = 0;
        buf[2 * i + 1] = 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (prof_t)__shfl_xor((unsigned long long)lo, o, 64));
        hi = max(hi, (prof_t)__shfl_xor((unsigned long long)hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lo;
        s_max[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lo = min(lo, s_min[w]);
            hi = max(hi, s_max[w]);
        }
        out_pair[0] = lo;
        out_pair[1] = hi;
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-token (prefill) glue kernels
// ---------------------------------------------------------------------------------------------
// qkv [T, (Hq+2Hkv)*D] -> q_t [Hq, T, D] (normed + roped) and K/V appended to the slot's pages at
// positions start + r.  grid = T; block = 256 (16 groups of 16 lanes; a group owns one head at a time).
struct QkvPostArgs {
    const uint16_t *qkv;
    const uint16_t *q_norm_w, *k_norm_w;
    uint16_t *q_t;
    uint16_t *key_pages, *value_pages;
    const int32_t *block_row;  // this slot's block-table row [max_pages]
    int T, start, page_size, max_pages, num_heads, num_kv_heads;
    float eps, rope_base;
    float *key_scales, *value_scales;  // KV8: the row scales of FP8 pages (kv8.h); key_pages / value_pages then hold bytes
};

template <int VD, bool KV8 = false>
static __global__ __launch_bounds__(256) void qkv_post_kernel(const QkvPostArgs p) {
    static_assert(!KV8 || VD == 8, "FP8 pages: head dimension 128");
    constexpr int D = 16 * VD;
    const int r = blockIdx.x;
    const int g = threadIdx.x >> 4;
    const int t = threadIdx.x & 15;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads;
    const int pos = p.start + r;
    const uint16_t *row = p.qkv + (long)r * (Hq + 2 * Hkv) * D;
    float cs[VD], sn[VD];
    rope_factors<VD>(t, pos, p.rope_base, cs, sn);
    const int lp = pos / p.page_size;
    const int slot = pos - lp * p.page_size;
    const int page_id = lp < p.max_pages ? p.block_row[lp] : -1;
    for (int h = g; h < Hq + 2 * Hkv; h += 16) {
        float v[VD];
        if (h < Hq) {
            head_norm_rope<VD>(row + (long)h * D, p.q_norm_w, t, p.eps, cs, sn, v);
            store_row<VD>(p.q_t + ((long)h * p.T + r) * D + t * VD, v);
        } else if (h < Hq + Hkv) {
            head_norm_rope<VD>(row + (long)h * D, p.k_norm_w, t, p.eps, cs, sn, v);
            const long prow = ((long)max(page_id, 0) * Hkv + (h - Hq)) * p.page_size + slot;
            if constexpr (KV8) {
                u32x2 codes;
                float sc, deq[8];
                kv8_quantize_row16(v, codes, sc, deq);
                if (page_id >= 0) {
                    *reinterpret_cast<u32x2 *>(reinterpret_cast<uint8_t *>(p.key_pages) + prow * D + t * VD) = codes;
                    if (t == 0) p.key_scales[prow] = sc;
                }
            } else if (page_id >= 0)
                store_row<VD>(p.key_pages + prow * D + t * VD, v);
        } else {
            load_row<VD>(row + (long)h * D + t * VD, v);
            const long prow = ((long)max(page_id, 0) * Hkv + (h - Hq - Hkv)) * p.page_size + slot;
            if constexpr (KV8) {
                u32x2 codes;
                float sc, deq[8];
                kv8_quantize_row16(v, codes, sc, deq);
                if (page_id >= 0) {
                    *reinterpret_cast<u32x2 *>(reinterpret_cast<uint8_t *>(p.value_pages) + prow * D + t * VD) = codes;
                    if (t == 0) p.value_scales[prow] = sc;
                }
            } else if (page_id >= 0)
                store_row<VD>(p.value_pages + prow * D + t * VD, v);
        }
    }
}

// [H, T, D] -> [T, H*D], 16 B per thread (D % 8 == 0)
static __global__ __launch_bounds__(256) void heads_to_rows_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out,
                                                            int H, int T, int D) {
    const int vpr = D / 8;
    const long total = (long)H * T * vpr;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int v = (int)(idx % vpr);
    const int h = (int)((idx / vpr) % H);
    const int t = (int)(idx / ((long)vpr * H));
    *reinterpret_cast<uint4 *>(out + ((long)t * H + h) * D + v * 8) =
        *reinterpret_cast<const uint4 *>(in + ((long)h * T + t) * D + v * 8);
}

// out = bf16(a + b), n % 8 == 0
static __global__ __launch_bounds__(256) void residual_add_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                           uint16_t *__restrict__ out, long n8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    uint16_t x[8], y[8], o[8];
    *reinterpret_cast<uint4 *>(x) = reinterpret_cast<const uint4 *>(a)[idx];
    *reinterpret_cast<uint4 *>(y) = reinterpret_cast<const uint4 *>(b)[idx];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = BF16::from_float(BF16::to_float(x[e]) + BF16::to_float(y[e]));
    reinterpret_cast<uint4 *>(out)[idx] = *reinterpret_cast<const uint4 *>(o);
}

// out[row, :] = bf16(x[row, :] * w) (w == nullptr: a copy): rows weighted by an RMSNorm weight for a consumer that applies 1 / rms to
// its sums (qmm6.h); one launch per batched step, ahead of layer 0 (every later layer gets its rows weighted by the w_down epilogue).
// Chunks of 8 columns; frag != 0: written in qmm6.h's fragment order (a chunk is one 16-byte run there too).
static __global__ __launch_bounds__(256) void weight_rows_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w,
                                                          uint16_t *__restrict__ out, long n8, int chunks_per_row, int frag) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    uint16_t a[8], g[8], o[8];
    *reinterpret_cast<uint4 *>(a) = reinterpret_cast<const uint4 *>(x)[idx];
    const int row = (int)(idx / chunks_per_row), ch = (int)(idx - (long)row * chunks_per_row);
    if (w) {
        *reinterpret_cast<uint4 *>(g) = reinterpret_cast<const uint4 *>(w)[ch];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = BF16::from_float(BF16::to_float(a[e]) * BF16::to_float(g[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e];
    }
    uint16_t *dst = out + idx * 8;
    if (frag) {  // qmm6_frag_offset(row, 8 ch, 8 chunks_per_row), spelled out (this header does not see qmm6.h)
        const int G = chunks_per_row >> 4, gcol = ch >> 4, k8 = ch & 15;  // k8: 8-column run inside the group; c = k8 >> 2, t = k8 & 3
        dst = out + ((((size_t)(row >> 4) * G + gcol) * 4 + (k8 & 3)) * 64 + (size_t)(16 * (k8 >> 2) + (row & 15))) * 8;
    }
    act_store16(dst, *reinterpret_cast<const u32x4 *>(o));  // (read by the qkv projection of layer 0 in the same step)
}

// gu [T, 2I] with (gate_i, up_i) interleaved -> act [T, I] = bf16(silu(gate) * up)   (I % 4 == 0)
static __global__ __launch_bounds__(256) void swiglu_interleaved_kernel(const uint16_t *__restrict__ gu,
                                                                 uint16_t *__restrict__ act, long n4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    uint16_t x[8], o[4];
    *reinterpret_cast<uint4 *>(x) = reinterpret_cast<const uint4 *>(gu)[idx];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gt = BF16::to_float(x[2 * e]);
        const float up = BF16::to_float(x[2 * e + 1]);
        o[e] = BF16::from_float((gt / (1.0f + expf(-gt))) * up);
    }
    reinterpret_cast<uint2 *>(act)[idx] = *reinterpret_cast<const uint2 *>(o);
}

// tokens[max_batch] -> x[slot, hidden] for slots [0, batch): one block per slot
static __global__ __launch_bounds__(256) void embed_slots_kernel(const int32_t *__restrict__ tokens,
                                                          const uint32_t *__restrict__ emb_w,
                                                          const uint16_t *__restrict__ emb_s,
                                                          const uint16_t *__restrict__ emb_b, uint16_t *__restrict__ x,
                                                          int hidden, int vocab, const int32_t *__restrict__ context_lens,
                                                          const float2 *__restrict__ rope_table,
                                                          float2 *__restrict__ rope_cur, int rope_positions,
                                                          int rope_half, float *__restrict__ ss_out) {
    __shared__ float wave_ss[4];
    const int slot = blockIdx.x;
    int token = tokens[slot];
    token = token < 0 ? 0 : (token >= vocab ? vocab - 1 : token);
    const int words = hidden / 8;
    const int groups = hidden / 128;
    float sumsq = 0.f;
    for (int w = threadIdx.x; w < words; w += 256) {
        const uint32_t packed = emb_w[(long)token * words + w];
        const float scale = BF16::to_float(emb_s[(long)token * groups + w / 16]);
        const float bias = BF16::to_float(emb_b[(long)token * groups + w / 16]);
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = BF16::from_float((float)((packed >> (4 * e)) & 0xfu) * scale + bias);
            const float v = BF16::to_float(o[e]);
            sumsq += v * v;
        }
        *reinterpret_cast<uint4 *>(x + (long)slot * hidden + w * 8) = *reinterpret_cast<const uint4 *>(o);
    }
    if (ss_out) {  // partial sums of squares of the row for the fused RMSNorm of the skinny QKV matmul (entry 0; rest zero)
        const float ws = wave_sum(sumsq);
        if ((threadIdx.x & 63) == 0) wave_ss[threadIdx.x >> 6] = ws;
        __syncthreads();
        if (threadIdx.x == 0) {
            ss_out[(long)slot * 8] = (wave_ss[0] + wave_ss[1]) + (wave_ss[2] + wave_ss[3]);
            for (int i = 1; i < 8; ++i) ss_out[(long)slot * 8 + i] = 0.f;
        }
    }
    // whatever changed the slot's context on the host side (prefill, rewind, move), the step starts from fresh factors
    if ((int)threadIdx.x < rope_half) {
        const int pos = min(context_lens[slot], rope_positions - 1);
        rope_cur[(long)slot * rope_half + threadIdx.x] = rope_table[(long)pos * rope_half + threadIdx.x];
    }
}

// host-by-value writes of a few int32 words (block-table entries, context lengths, tokens) in stream order
struct PokeArgs {
    int32_t *addr[8];
    int32_t value[8];
    int n;
};
static __global__ void poke_kernel(const PokeArgs p) {
    if ((int)threadIdx.x < p.n) *p.addr[threadIdx.x] = p.value[threadIdx.x];
}
static __global__ void fill_i32_kernel(int32_t *dst, int32_t value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// ---- Qwen3-MoE layers inside the engine (reference: src/tiny_llm_ref/moe.py:39-89, qwen3_week3.py:258-272) ----------------------
// Router: softmax over the E router logits of a row in fp32, rounded to bf16 (the activations' dtype, moe.py:45-46), the top_k
// largest probabilities in descending order (equal probabilities: the lower expert index first) and their scores, renormalised over
// the selection when `norm` (bf16 arithmetic as the reference's arrays: the sum rounded once, the quotient rounded once).
// One workgroup per activation row; E <= 1024, top_k <= 16.
static __global__ __launch_bounds__(256) void moe_route_kernel(const uint16_t *__restrict__ logits, int E, int top_k, int norm,
                                                        int32_t *__restrict__ ids, uint16_t *__restrict__ scores) {
    __shared__ float red[4];
    __shared__ int red_i[4];
    __shared__ uint16_t probs[1024];
    __shared__ uint16_t picked[16];
    const int row = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint16_t *lr = logits + (long)row * E;
    float mx = -INFINITY;
    for (int i = tid; i < E; i += 256) mx = fmaxf(mx, BF16::to_float(lr[i]));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < E; i += 256) sum += expf(BF16::to_float(lr[i]) - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = tid; i < E; i += 256) probs[i] = BF16::from_float(expf(BF16::to_float(lr[i]) - mx) / sum);
    __syncthreads();
    for (int j = 0; j < top_k; ++j) {  // top_k rounds of a workgroup-wide arg-max; a picked expert leaves the pool
        float best = -1.f;
        int best_i = 0x7fffffff;
        for (int i = tid; i < E; i += 256) {
            const float v = BF16::to_float(probs[i]);
            if (probs[i] != 0xffffu && (v > best || (v == best && i < best_i))) best = v, best_i = i;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(best_i, off);
            if (ov > best || (ov == best && oi < best_i)) best = ov, best_i = oi;
        }
        if (lane == 0) red[wave] = best, red_i[wave] = best_i;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red[w] > best || (red[w] == best && red_i[w] < best_i)) best = red[w], best_i = red_i[w];
            if (best_i < 0 || best_i >= E) {  // NaN / Inf router logits leave no candidate: take the lowest expert still in the pool
                best_i = 0;               // (a finite, in-range id: the grouped GEMVs index expert weights with it inside a captured graph)
                for (int i = 0; i < E; ++i)
                    if (probs[i] != 0xffffu) {
                        best_i = i;
                        break;
                    }
            }
            ids[(long)row * top_k + j] = best_i;
            picked[j] = probs[best_i];
            probs[best_i] = 0xffffu;  // a NaN pattern no probability takes: out of the pool
        }
        __syncthreads();
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int j = 0; j < top_k; ++j) tot += BF16::to_float(picked[j]);
        const float den = BF16::to_float(BF16::from_float(tot));
        for (int j = 0; j < top_k; ++j)
            scores[(long)row * top_k + j] = norm ? BF16::from_float(BF16::to_float(picked[j]) / den) : picked[j];
    }
}

// act = bf16(bf16(silu(gate)) * up), 8 elements per thread (moe.py:83-84: silu(gate) * up on bf16 arrays, two roundings)
static __global__ __launch_bounds__(256) void moe_silu_mul_kernel(const uint16_t *__restrict__ gate, const uint16_t *__restrict__ up,
                                                           uint16_t *__restrict__ act, long n8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    uint16_t g[8], u[8], o[8];
    *reinterpret_cast<uint4 *>(g) = reinterpret_cast<const uint4 *>(gate)[idx];
    *reinterpret_cast<uint4 *>(u) = reinterpret_cast<const uint4 *>(up)[idx];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gt = BF16::to_float(g[e]);
        o[e] = BF16::from_float(bf16_round(gt / (1.0f + expf(-gt))) * BF16::to_float(u[e]));
    }
    reinterpret_cast<uint4 *>(act)[idx] = *reinterpret_cast<const uint4 *>(o);
}

// out[m] = bf16(h[m] + bf16(sum_j bf16(y[m, j] * score[m, j])))   (moe.py:89 then the layer's residual add, qwen3_week3.py:204-205);
// the sum over the top_k expert rows accumulates in fp32 in expert order.  grid = rows, D % 8 == 0.
static __global__ __launch_bounds__(256) void moe_combine_kernel(const uint16_t *__restrict__ y, const uint16_t *__restrict__ scores,
                                                          const uint16_t *__restrict__ h, uint16_t *__restrict__ out, int D,
                                                          int top_k) {
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < D / 8; c += 256) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int j = 0; j < top_k; ++j) {
            const float sc = BF16::to_float(scores[(long)m * top_k + j]);
            uint16_t v[8];
            *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(y + ((long)m * top_k + j) * D + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf16_round(BF16::to_float(v[e]) * sc);
        }
        uint16_t r[8], o[8];
        *reinterpret_cast<uint4 *>(r) = *reinterpret_cast<const uint4 *>(h + (long)m * D + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = BF16::from_float(BF16::to_float(r[e]) + bf16_round(acc[e]));
        *reinterpret_cast<uint4 *>(out + (long)m * D + c * 8) = *reinterpret_cast<const uint4 *>(o);
    }
}

}  // namespace tl
