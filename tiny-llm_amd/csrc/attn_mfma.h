// Decode attention of a whole GQA group on the matrix cores (head_dim 128, pages a power of two >= 32 tokens).
//
// attn_decode_fused_kernel (engine_kernels.h) reduces a window on the VALU: 570 instructions per wave for 16 tokens x 4 query heads
// (dot products by FMA, 64 DPP adds to close them, bf16 -> fp32 conversions of every K / V element).  One sequence at a long context
// hides that behind the K/V round trips; 8-64 sequences do not: with every CU holding two workgroups the walk is VALU bound, and
// attention was the largest phase of a batched step (round 4: 27 us per layer at 64 sequences of ~320 tokens, 84 MB of K/V = 14 us at
// 6 TB/s).  Here the same walk runs as two matrix products per 32 tokens and wave:
//   S^T [token, head]    = K [token, 128] . Q^T [128, head]       4 x v_mfma_f32_16x16x32_bf16 per 16 tokens (heads = 4 of 16 columns)
//   O^T [dim, head]     += V^T [dim, token] . P^T [token, head]    8 x (hi + lo) per 32 tokens
// The transposed score tile IS the B operand of the second product: lane (head n = lane % 16, g = lane / 16) holds the scores of tokens
// 4 g .. 4 g + 3 of each 16-token tile, and a B operand wants 8 consecutive reduction indices per lane -- the two tiles' 4 + 4 tokens,
// in that order; V's rows are loaded in the same order, so no value crosses lanes between the products.  Softmax statistics are per
// lane (head = column), the running maximum is closed over the four 16-lane rows with two v_permlane swaps per 32 tokens.
//   K rows are read from HBM coalesced (a 16-lane row reads one token's 256 bytes) and turned into A operands (lane = token, 8
//   consecutive dims) through a wave-private LDS image with 272-byte rows: 16-byte reads of 16 rows at one column touch all 64 banks.
//   (Loading A operands straight from HBM -- 16 rows x 64 B per instruction -- is the access pattern qmv3.h measured at ~120 ns per
//   instruction to ISSUE.)  LDS traffic of one wave is ordered: no barrier inside the walk.
//   V rows are read the same way (lane (c, g): dims 8 c .. 8 c + 7 of its 8 tokens); the A operand of output tile j takes element j of
//   the eight chunks: 4 v_perm_b32.  Output tile j, row m <-> dim 8 m + j, so D's four rows of a lane are dims 32 g + 8 i + j.
//   P is split into bf16 hi + lo parts (two MFMAs): the weights keep ~16 mantissa bits, l is summed in fp32 as before.
// Everything around the walk is attn_decode_fused_kernel's: the same two round trips (scalar context length / page ids, the new token's
// q / k / v rows or fp32 slice partials; then K/V), q/k-norm + RoPE, the token being decoded from registers, (m, l, acc) partials or the
// output row, the KV append.  Merge slots in LDS: 0 = the token being decoded, 1 + w = wave w's part of the window.
//   reference semantics: paged_attention.metal:108-248 (decode), paged_cache_update :82-106, qwen3_week3.py:63-86 for the op order.
#pragma once
#include "engine_kernels.h"

namespace tl {

constexpr int AM_KROW = 128 * 2 + 16;    // bytes of a K row in LDS
constexpr int AM_KWAVE = 32 * AM_KROW;   // a wave's 32 rows
constexpr int AM_NSLOT = 5;
constexpr int AM_OFF_K = AM_NSLOT * AD_RQ * (128 + 2) * 4;  // 10,400: a multiple of 16
constexpr int AM_OFF_Q = AM_OFF_K + 4 * AM_KWAVE;
constexpr int AM_OFF_QP = AM_OFF_Q + AD_RQ * 128 * 2;
static_assert(AM_OFF_K % 16 == 0 && AM_OFF_Q % 16 == 0 && AM_OFF_QP % 16 == 0, "16-byte LDS rows");
constexpr size_t attn_mfma_lds_bytes(bool qp) { return (size_t)AM_OFF_QP + (qp ? (2 + AD_RQ) * 128 * 2 : 0); }

// KV8 = FP8 pages (kv8.h): a lane requests 8 bytes of codes per row instead of 16 of bf16, and the scales of its 8 tokens as two
// 16-byte words per pool.  Codes become bf16 UNSCALED on their way into the K image / the V operands (exact); the K row's scale
// multiplies its score, the V row's its softmax weight -- powers of two, so this is bit for bit the walk over the dequantised rows.
template <bool QP, bool KV8 = false>
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(const AttnDecodeArgs p) {
    using KVC = typename std::conditional<KV8, u32x2, u32x4>::type;  // a lane's chunk of a K / V row as requested
    constexpr int VD = 8, D = 128, RQ = AD_RQ, STRIDE = D + 2, WT = 32, ST = 4 * WT;  // tokens per wave / workgroup and stage
    extern __shared__ __attribute__((aligned(16))) float psm[];  // [AM_NSLOT][RQ][STRIDE] | K rows | q rows | (QP) staged rows
    char *lds = reinterpret_cast<char *>(psm);
    const prof_t prof_t0 = prof_begin(p.prof);
    const int bx = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const int split = bx & (p.n_splits - 1);
    const int chunk = bx >> p.split_shift;
    const int kvh = __builtin_amdgcn_readfirstlane(blockIdx.y);
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.z);
    const int Hq = p.num_heads, Hkv = p.num_kv_heads;
    const int rep = p.rep;
    auto page_of = [&](int tok) { return tok >> p.page_shift; };
    const int g16 = threadIdx.x >> 4;  // 16-lane group of the workgroup (prologue / epilogue roles, as in attn_decode_fused_kernel)
    const int t = threadIdx.x & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int32_t *brow = p.block_table + (long)b * p.max_pages;
    const uint16_t *row = p.qkv + (long)b * (Hq + 2 * Hkv) * D;
    const float scale_log2 = p.scale * ENG_LOG2E;
    const int C = p.tokens_per_split;
    const int t_begin = split * C;
    const int n_it = (C + ST - 1) / ST;
    auto wave_base = [&](int it) { return t_begin + it * ST + w * WT; };  // first token of this wave's 32 in stage it (inside one page)

    // ---- round trip 1 ------------------------------------------------------------------------------------------------------
    int ctx, first_page, pg_cur = 0, pg_nxt = 0, pg_new = 0;
    sload_i32(p.context_lens + b, ctx);
    sload_i32(brow, first_page);
    sload_i32(brow + min(page_of(wave_base(0)), p.max_pages - 1), pg_cur);
    sload_i32(brow + min(page_of(wave_base(min(1, n_it - 1))), p.max_pages - 1), pg_nxt);
    // the prologue's query rows are shared out over the waves (round 6): wave w norms and rotates query head w of the group (and the new K row, which
    // every wave needs for its head's score against the token being decoded).  Until then every wave did all RQ heads and three waves' results were
    // dropped: 720 VALU instructions per wave in front of a walk of 170 per stage -- with two workgroups per CU the prologue WAS the launch at short contexts.
    static_assert(RQ == 4, "one query head of the GQA group per wave");
    RawRow<VD> kraw_new, vraw_new, qraw_w, qw, kw;
    if constexpr (!QP) {
        load_raw_act<VD>(row + (long)(Hq + kvh) * D + t * VD, kraw_new);
        load_raw_act<VD>(row + (long)(Hq + Hkv + kvh) * D + t * VD, vraw_new);
    }
    load_raw<VD>(p.q_norm_w + t * VD, qw);
    load_raw<VD>(p.k_norm_w + t * VD, kw);
    constexpr int QP_ROWS = 2 + RQ;
    constexpr int QP_CHUNKS = QP_ROWS * D / 4;
    constexpr int QP_PER = (QP_CHUNKS + 255) / 256;
    constexpr int QP_INFLIGHT = 4;
    f32x4 qp_x[QP ? QP_PER : 1][QP ? QP_INFLIGHT : 1];
    auto qp_col = [&](int ch) {
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
        const int hq = min(chunk * RQ + w, rep - 1);
        load_raw_act<VD>(row + (long)(kvh * rep + hq) * D + t * VD, qraw_w);
    }
    float cs[VD], sn[VD];
    rope_from_table<VD>(p.rope_cur + (long)b * (D / 2), t, cs, sn);
    __builtin_amdgcn_sched_barrier(0);
    sload_wait(ctx, first_page, pg_cur, pg_nxt);
    const bool live = first_page >= 0;

    // ---- round trip 2: K/V rows of the first stage, the append slot -------------------------------------------------------
    const int wp = page_of(ctx);
    const int wslot = ctx - wp * p.page_size;
    int wpage;
    sload_i32(brow + min(wp, p.max_pages - 1), wpage);
    // lane (c, g) reads dims 8 c .. 8 c + 7 of rows 4 g + e (e < 4) and 16 + 4 g + e - 4 of the wave's 32: the order of the second
    // product's reduction index
    // Buffer loads: the stage's row base is a scalar resource, a lane's offset never changes -- no 64-bit address arithmetic on the
    // VALU, and no address registers for the allocator to recycle as load destinations (which made the next stage's requests wait for
    // this stage's V rows: a write-after-write on the recycled register).
    const int lane_bytes = ((4 * g) * D + c * VD) * 2;
    // The resource ends behind the last VISIBLE row of the wave's 32 (context length, window end, an unmapped page: none): rows past it
    // come back as zeros from the range check -- a masked token's weight is 0, but 0 x NaN is NaN, and a recycled or caller-provided
    // page may hold anything behind the context (scalar arithmetic only: nothing is added to the walk)
    // KV8: the same rows at one byte per element, and the scales of the lane's 8 tokens (rows 4 g .. 4 g + 3 and 16 + 4 g .. + 3 of the wave's 32)
    struct Scales {
        f32x4 lo, hi;
    };
    auto issue_rows = [&](const uint16_t *pool, const float *scales, int tb, int pg, KVC(&rows)[8], Scales &sc) {
        const long prow0 = ((long)max(pg, 0) * Hkv + kvh) * p.page_size + (tb & (p.page_size - 1));  // uniform
        const int end = (pg >= 0 && page_of(tb) < p.max_pages && live) ? min(ctx, t_begin + C) : 0;
        const int visible = min(max(end - tb, 0), WT);
        if constexpr (KV8) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(reinterpret_cast<const uint8_t *>(pool) + prow0 * D), 0, visible * D, 0x00020000);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                rows[e] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes / 2 + ((e < 4) ? e : 12 + e) * D, 0, 0));
            const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(scales + prow0), 0, visible * 4, 0x00020000);
            sc.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rss, 16 * g, 0, 0));
            sc.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rss, 64 + 16 * g, 0, 0));
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(pool + prow0 * D), 0, visible * D * 2, 0x00020000);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                rows[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes + ((e < 4) ? e : 12 + e) * D * 2, 0, 0));
        }
    };
    KVC kr[8], va[8], vb[8];
    Scales ksa, ksb, vsa, vsb;  // KV8: the scales travel with the V register sets (a stage's K scales are used after the next stage's K rows are requested)
    issue_rows(p.key_pages, p.key_scales, wave_base(0), pg_cur, kr, ksa);
    issue_rows(p.value_pages, p.value_scales, wave_base(0), pg_cur, va, vsa);

    if constexpr (QP) {
        uint16_t *qs = reinterpret_cast<uint16_t *>(lds + AM_OFF_QP);
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
            for (int s = QP_INFLIGHT; s < p.qkv_slices; ++s) {
                const f32x4 x4 = *reinterpret_cast<const f32x4 *>(prow_base + qp_col(ch) + (long)s * p.qkv_plane);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) acc4[e2] += x4[e2];
            }
            if (chu < QP_CHUNKS) {
                uint2 packed;
                packed.x = BF16::pack2(acc4[0], acc4[1]);
                packed.y = BF16::pack2(acc4[2], acc4[3]);
                *reinterpret_cast<uint2 *>(qs + ch * 4) = packed;
            }
        }
        __syncthreads();
        load_raw<VD>(qs + 0 * D + t * VD, kraw_new);
        load_raw<VD>(qs + 1 * D + t * VD, vraw_new);
        load_raw<VD>(qs + (2 + w) * D + t * VD, qraw_w);
    }

    // ---- prologue math while the K/V rows are in flight ---------------------------------------------------------------
    auto norm_rope = [&](const RawRow<VD> &x, const RawRow<VD> &wn, float(&out)[VD]) {
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
            const float n = bf16_round(f[i] * inv * BF16::to_float(wn.v[i]));
            const float partner = row_ror<8>(n);
            const float r2 = (t < 8) ? (n * cs[i] - partner * sn[i]) : (n * cs[i] + partner * sn[i]);
            out[i] = bf16_round(r2);
        }
    };
    float k_new[VD];
    norm_rope(kraw_new, kw, k_new);
    // KV8: the new rows as the page will hold them; this step attends to the dequantised values
    float v_new[VD];
#pragma unroll
    for (int i = 0; i < VD; ++i) v_new[i] = BF16::to_float(vraw_new.v[i]);
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
    uint16_t *qrows = reinterpret_cast<uint16_t *>(lds + AM_OFF_Q);
    const bool with_new = split == 0 && live;
    {
        const int r = w;  // this wave's query head
        float qn[VD];
        norm_rope(qraw_w, qw, qn);
        // merge slot 0: the token being decoded (position ctx), straight from registers: (m, l, acc) = (score, 1, v)
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < VD; ++i) part += (qn[i] * scale_log2) * k_new[i];
        const float score = group16_allsum(part);
        if ((threadIdx.x & 63) < 16) {  // the wave's first 16-lane group holds the row like every other: it writes
            store_row<VD>(qrows + r * D + t * VD, qn);  // (bf16 values: exact)
            float *dst = psm + (long)r * STRIDE;
#pragma unroll
            for (int i = 0; i < VD; ++i) dst[t * VD + i] = with_new ? v_new[i] : 0.f;
            if (t == 0) {
                dst[D] = with_new ? score : -1e30f;
                dst[D + 1] = with_new ? 1.f : 0.f;
            }
        }
    }
    __syncthreads();
    // B operands of the first product: lane (head n = c, g), k-step j: dims 32 g + 8 j .. + 7 of head n (columns >= RQ: zero)
    u32x4 qb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(qrows + min(c, RQ - 1) * D + 32 * g + 8 * j);
        qb[j] = c < RQ ? v : u32x4{0u, 0u, 0u, 0u};
    }

    // ---- walk the window ------------------------------------------------------------------------------------------------------
    // One register set for K (a stage's rows go to LDS as soon as they arrive, and the next stage's are requested into the same
    // registers), two for V (used alternately: the loop is unrolled by two through this lambda).  Every request stands outside any
    // branch (a stage past the window's end requests the last stage again): at a control-flow join hipcc's wait bookkeeping loses the
    // issue order of pending loads (engine_kernels.h, attn_decode_fused_kernel).
    char *kl = lds + AM_OFF_K + w * AM_KWAVE;
    f32x4 o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    auto walk_stage = [&](int it, KVC(&vraw)[8], KVC(&vn)[8], Scales &ksc, Scales &vsc, Scales &ksn, Scales &vsn) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u32x4 kk;
            if constexpr (KV8) kk = kv8_to_bf16x8(kr[e], 1.0f);
            else kk = kr[e];
            *reinterpret_cast<u32x4 *>(kl + (4 * g + ((e < 4) ? e : 12 + e)) * AM_KROW + 16 * c) = kk;
        }
        const int nx = min(it + 1, n_it - 1);
        issue_rows(p.key_pages, p.key_scales, wave_base(nx), pg_nxt, kr, ksn);
        issue_rows(p.value_pages, p.value_scales, wave_base(nx), pg_nxt, vn, vsn);
        sload_i32(brow + min(page_of(wave_base(min(it + 2, n_it - 1))), p.max_pages - 1), pg_new);  // waited for at the end of this stage
        // S^T = K . Q^T for the two 16-token tiles
        f32x4 s[2];
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            s[ab] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 ka = *reinterpret_cast<const u32x4 *>(kl + (16 * ab + c) * AM_KROW + 64 * g + 16 * j);
                s[ab] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka), __builtin_bit_cast(bf16x8_t, qb[j]), s[ab], 0,
                                                                0, 0);
            }
        }
        // lane (head c, g): tokens tb + 16 ab + 4 g + i.  A masked score is far below the initial maximum (-1e30): its weight is 0
        // without a second select.
        const int tb = wave_base(it);
        const bool page_ok = page_of(tb) < p.max_pages && pg_cur >= 0 && live;
        const int limit = page_ok ? min(ctx, t_begin + C) : 0;
        float sv[8];
        float tm = -3e38f;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tok = tb + 16 * ab + 4 * g + i;
                if constexpr (KV8) sv[4 * ab + i] = tok < limit ? s[ab][i] * scale_log2 * (ab ? ksc.hi[i] : ksc.lo[i]) : -3e38f;
                else sv[4 * ab + i] = tok < limit ? s[ab][i] * scale_log2 : -3e38f;
                tm = fmaxf(tm, sv[4 * ab + i]);
            }
        tm = fmaxf(tm, lane_xor16(tm, lane));
        tm = fmaxf(tm, lane_xor32(tm, lane));
        const float nm = fmaxf(m_run, tm);
        const float of = exp2_hw(m_run - nm);
        m_run = nm;
        float pw[8], psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pw[e] = exp2_hw(sv[e] - nm);
            psum += pw[e];
        }
        l_run = l_run * of + psum;
        u32x4 vc[8];  // the V chunks as bf16
        if constexpr (KV8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pw[e] *= (e < 4) ? vsc.lo[e & 3] : vsc.hi[e & 3];  // the V row's scale rides on the weight (a masked row: 0 x 0)
                vc[e] = kv8_to_bf16x8(vraw[e], 1.0f);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) vc[e] = vraw[e];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][i] *= of;
        // B operands of the second product: the lane's 8 weights in reduction order, as bf16 hi + lo
        u32x4 ph, pl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ph[q] = BF16::pack2(pw[2 * q], pw[2 * q + 1]);
            const float r0 = pw[2 * q] - __uint_as_float(ph[q] << 16), r1 = pw[2 * q + 1] - __uint_as_float(ph[q] & 0xffff0000u);
            pl[q] = BF16::pack2(r0, r1);
        }
        // O^T += V^T . P^T: the A operand of output tile j is element j of the lane's eight 16-byte chunks
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
            u32x4 a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = __builtin_amdgcn_perm(vc[2 * q + 1][j >> 1], vc[2 * q][j >> 1], sel);
            }
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, ph), o[j], 0, 0, 0);
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, pl), o[j], 0, 0, 0);
        }
        sload_wait(pg_new);
        pg_cur = pg_nxt;
        pg_nxt = pg_new;
    };
    // (the odd tail leaves the loop instead of skipping to its latch: the back edge carries ONE order of pending requests)
    for (int it = 0;; it += 2) {
        walk_stage(it, va, vb, ksa, vsa, ksb, vsb);
        if (it + 1 >= n_it) break;
        walk_stage(it + 1, vb, va, ksb, vsb, ksa, vsa);
        if (it + 2 >= n_it) break;
    }

    // ---- the wave's part -> merge slot 1 + w; merge; append ---------------------------------------------------------------
    float l_tot = l_run + lane_xor16(l_run, lane);
    l_tot += lane_xor32(l_tot, lane);
    if (c < RQ) {
        float *dst = psm + ((long)(1 + w) * RQ + c) * STRIDE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                *reinterpret_cast<float2 *>(dst + 32 * g + 8 * i + j) = make_float2(o[j][i], o[j + 1][i]);
        if (g == 0) {
            dst[D] = m_run;
            dst[D + 1] = l_tot;
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
        for (int j = 0; j < AM_NSLOT; ++j) gm = fmaxf(gm, psm[((long)j * RQ + r) * STRIDE + D]);
        float gl = 0.f, vs = 0.f;
#pragma unroll
        for (int j = 0; j < AM_NSLOT; ++j) {
            const float *src = psm + ((long)j * RQ + r) * STRIDE;
            const float f = exp2_hw(src[D] - gm);
            gl += src[D + 1] * f;
            vs += src[d] * f;
        }
        const long orow = (long)b * Hq + kvh * rep + hq;
        if (p.n_splits == 1) {
            act_store(&p.out[orow * D + d], BF16::from_float(gl == 0.f ? 0.f : vs / gl));
        } else {
            float *wsr = p.ws + (orow * p.n_splits + split) * (D + ATTN_WS_PAD);
            act_store(&wsr[d], vs);
            if (d == 0) {
                act_store(&wsr[D], gm);
                act_store(&wsr[D + 1], gl);
            }
        }
    }
    sload_wait(wpage);
    if (live && wp < p.max_pages && wpage >= 0 && split == 0 && chunk == 0 && g16 == 0) {
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

// launcher (attn_mfma.hip, compiled with the VGPR form of MFMA: the softmax reads and rescales every accumulator on the VALU)
bool attn_decode_mfma_applicable(const AttnDecodeArgs &a, int head_dim, int rq);
void launch_attn_decode_mfma(const AttnDecodeArgs &a, dim3 grid, hipStream_t st);

}  // namespace tl
