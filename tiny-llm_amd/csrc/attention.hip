// Attention kernels for gfx950: dense decode attention, paged-KV scatter,
// split-context paged decode attention (+merge) and paged FlashAttention on
// bf16 MFMA.  Reference: week2_kernels.metal:119-235, paged_attention.metal:82-506,
// host dispatch week2_kernels.cpp:176-211 and paged_attention.cpp:129-225.
#include <type_traits>

#include "common.h"
#include "kv8.h"

namespace tl {

constexpr float LOG2E = 1.44269504089f;

// ---------------------------------------------------------------------------
// Dense decode attention (any dtype, D <= 256).  One workgroup per query row;
// 16 groups of 16 lanes each walk the context with their own online softmax,
// then merge through LDS.  Same visibility rule as the reference kernel
// (position > S - L + query_position is skipped, week2_kernels.metal:165).
// ---------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void decode_attention_kernel(const typename TT::storage *__restrict__ q,
                                                               const typename TT::storage *__restrict__ k,
                                                               const typename TT::storage *__restrict__ v,
                                                               const float *__restrict__ mask,
                                                               typename TT::storage *__restrict__ out, int q_rows,
                                                               int L, int S, int D, int num_heads, int num_kv_heads,
                                                               float scale, int is_causal, int has_mask) {
    constexpr int MAXV = 16;  // D <= 256
    extern __shared__ __attribute__((aligned(16))) float dsm[];  // [16][D] acc + [16] m + [16] l
    const int query_index = blockIdx.x;
    const int query_row = query_index / L;
    const int qp = query_index - query_row * L;
    const int batch = query_row / num_heads;
    const int qh = query_row - batch * num_heads;
    const int kvh = qh / (num_heads / num_kv_heads);
    const long kv_row = (long)batch * num_kv_heads + kvh;
    const int g = threadIdx.x >> 4;
    const int t = threadIdx.x & 15;
    const int nv = (D + 15) >> 4;

    float qv[MAXV], acc[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int d = t + 16 * i;
        qv[i] = (i < nv && d < D) ? TT::to_float(q[(long)query_index * D + d]) * scale : 0.f;
        acc[i] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    const int last_visible = is_causal ? (S - L + qp) : (S - 1);
    for (int pos = g; pos < S; pos += 16) {
        if (pos > last_visible) break;
        const typename TT::storage *kp = k + (kv_row * S + pos) * D;
        const typename TT::storage *vp = v + (kv_row * S + pos) * D;
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int d = t + 16 * i;
            if (i < nv && d < D) part += qv[i] * TT::to_float(kp[d]);
        }
        float score = group16_sum(part);
        if (has_mask) score += mask[(long)query_index * S + pos];
        const float nm = fmaxf(m, score);
        const float of = __expf(m - nm);
        const float sf = __expf(score - nm);
        l = l * of + sf;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int d = t + 16 * i;
            if (i < nv && d < D) acc[i] = acc[i] * of + sf * TT::to_float(vp[d]);
        }
        m = nm;
    }
    float *pacc = dsm;
    float *pm = dsm + 16 * D;
    float *pl = pm + 16;
    if (t == 0) {
        pm[g] = m;
        pl[g] = l;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int d = t + 16 * i;
        if (i < nv && d < D) pacc[g * D + d] = acc[i];
    }
    __syncthreads();
    float gm = -1e30f;
#pragma unroll
    for (int j = 0; j < 16; ++j) gm = fmaxf(gm, pm[j]);
    float gl = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) gl += pl[j] * __expf(pm[j] - gm);
    for (int d = threadIdx.x; d < D; d += 256) {
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) vs += pacc[j * D + d] * __expf(pm[j] - gm);
        out[(long)query_index * D + d] = TT::from_float(gl == 0.f ? 0.f : vs / gl);
    }
}

// ---------------------------------------------------------------------------
// Paged KV scatter: values [1,H,len,D] -> pages[page_id,h,start+t,:], 16 B per
// thread.  reference: paged_attention.metal:82-106 (in-place on `pages`).
// ---------------------------------------------------------------------------
template <int BYTES>
__global__ __launch_bounds__(256) void paged_cache_update_kernel(const char *__restrict__ values,
                                                                 char *__restrict__ pages, int heads, int length,
                                                                 long row_bytes, int page_size, int page_id,
                                                                 int start) {
    // one "row" = one token's D elements for one head = row_bytes bytes
    const long vec_per_row = row_bytes / BYTES;
    const long total = (long)heads * length * vec_per_row;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long rowi = idx / vec_per_row;
    const long within = idx - rowi * vec_per_row;
    const int head = (int)(rowi / length);
    const int token = (int)(rowi - (long)head * length);
    const long dst = (((long)page_id * heads + head) * page_size + start + token) * row_bytes + within * BYTES;
    const long src = rowi * row_bytes + within * BYTES;
    if constexpr (BYTES == 16) {
        *reinterpret_cast<uint4 *>(pages + dst) = *reinterpret_cast<const uint4 *>(values + src);
    } else if constexpr (BYTES == 4) {
        *reinterpret_cast<uint32_t *>(pages + dst) = *reinterpret_cast<const uint32_t *>(values + src);
    } else {
        *reinterpret_cast<uint16_t *>(pages + dst) = *reinterpret_cast<const uint16_t *>(values + src);
    }
}

// ---------------------------------------------------------------------------
// FP8 (E4M3) pages (kv8.h; no reference counterpart): the quantising twin of the scatter above -- rows of 128 bf16 values
// [H, len, 128] -> codes pages[page_id, h, start + t, :] + one scale per row -- and the inverse.  One aligned group of 16 lanes per
// row (8 values per lane), the row's largest magnitude by DPP.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv8_quantize_rows_kernel(const uint16_t *__restrict__ values, uint8_t *__restrict__ pages,
                                                                float *__restrict__ scales, long rows, int length, int page_size,
                                                                long page_row0, int start) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int t = threadIdx.x & 15;
    const long r = min(row, rows - 1);  // every lane takes part in the row reduction
    const u32x4 raw = *reinterpret_cast<const u32x4 *>(values + r * 128 + t * 8);
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[2 * i] = __uint_as_float(raw[i] << 16);
        x[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
    }
    u32x2 codes;
    float sc, deq[8];
    kv8_quantize_row16(x, codes, sc, deq);
    if (row >= rows) return;
    const long h = r / length, tt = r - h * length;
    const long dst = page_row0 + h * page_size + start + tt;
    *reinterpret_cast<u32x2 *>(pages + dst * 128 + t * 8) = codes;
    if (t == 0) scales[dst] = sc;
}

__global__ __launch_bounds__(256) void kv8_dequantize_rows_kernel(const uint8_t *__restrict__ codes, const float *__restrict__ scales,
                                                                  uint16_t *__restrict__ out, long rows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // a chunk of 8 elements
    if (i >= rows * 16) return;
    const u32x2 c = *reinterpret_cast<const u32x2 *>(codes + i * 8);
    *reinterpret_cast<u32x4 *>(out + i * 8) = kv8_to_bf16x8(c, scales[i >> 4]);
}

// ---------------------------------------------------------------------------
// Paged decode attention, split over the context (flash-decoding).
//   grid = (n_splits * n_row_chunks, Hkv, B); workgroup = 16 groups x 16 lanes.
//   A workgroup owns one KV head, RQ=4 query rows of that head's GQA group
//   (row r -> q head r / L, query position r % L) and one slice of the context,
//   so every K/V byte is fetched once per 4 query heads (the reference fetches
//   it once per query head, paged_attention.metal:108-248).  A 16-lane group
//   reads one token's K (and V) row as 16 x 16 B = fully coalesced 256 B.
//   Result: either final output (n_splits == 1) or (m, l, acc) partials that
//   paged_merge_kernel combines.
// ---------------------------------------------------------------------------
constexpr int PD_RQ = 4;

// KV8 (bf16 queries, D = 128): the pages hold E4M3 codes, one byte per element of the same layout, with one power-of-two float32
// scale per (page, kv head, slot) row in key_scales / value_scales (kv8.h); the row scales are folded into the score and into the
// softmax weight -- bit for bit the arithmetic of this kernel over the dequantised rows.
template <typename TT, int VD, bool VEC, bool KV8 = false>
__global__ __launch_bounds__(256) void paged_decode_kernel(
    const typename TT::storage *__restrict__ q, const typename TT::storage *__restrict__ key_pages,
    const typename TT::storage *__restrict__ value_pages, const int32_t *__restrict__ block_table,
    const int32_t *__restrict__ context_lens, typename TT::storage *__restrict__ out, float *__restrict__ ws, int L,
    int D, int page_size, int max_pages, int num_heads, int num_kv_heads, float scale, int is_causal, int n_splits,
    int n_row_chunks, const float *__restrict__ key_scales = nullptr, const float *__restrict__ value_scales = nullptr) {
    static_assert(!KV8 || (VD == 8 && VEC), "FP8 pages: head dimension 128");
    using S = typename TT::storage;
    extern __shared__ __attribute__((aligned(16))) float psm[];  // [16][RQ][D+2]
    const int split = blockIdx.x % n_splits;
    const int chunk = blockIdx.x / n_splits;
    const int kvh = blockIdx.y;
    const int b = blockIdx.z;
    const int rep = num_heads / num_kv_heads;
    const int R = rep * L;
    const int g = threadIdx.x >> 4;
    const int t = threadIdx.x & 15;
    const int ctx = context_lens[b];
    const float scale_log2 = scale * LOG2E;
    const int stride = D + 2;

    // query rows of this chunk
    int rq_n[PD_RQ], rq_qp[PD_RQ], rq_vis[PD_RQ];
    bool rq_ok[PD_RQ];
    float qv[PD_RQ][VD], acc[PD_RQ][VD], m[PD_RQ], l[PD_RQ];
    int max_vis = 0;
#pragma unroll
    for (int r = 0; r < PD_RQ; ++r) {
        const int rr = chunk * PD_RQ + r;
        rq_ok[r] = rr < R;
        const int hq = rq_ok[r] ? rr / L : 0;
        rq_qp[r] = rq_ok[r] ? rr - hq * L : 0;
        rq_n[r] = b * num_heads + kvh * rep + hq;
        int vis = is_causal ? min(max(ctx - L + rq_qp[r] + 1, 0), ctx) : ctx;
        rq_vis[r] = rq_ok[r] ? vis : 0;
        max_vis = max(max_vis, rq_vis[r]);
        m[r] = -1e30f;
        l[r] = 0.f;
        const S *qp = q + ((long)rq_n[r] * L + rq_qp[r]) * D;
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            const int d = VEC ? t * VD + i : t + 16 * i;
            qv[r][i] = (rq_ok[r] && d < D) ? TT::to_float(qp[d]) * scale_log2 : 0.f;
            acc[r][i] = 0.f;
        }
    }
    // context slice of this split (multiples of 16 tokens so groups stay aligned)
    const int per = ((max_vis + n_splits - 1) / n_splits + 15) & ~15;
    const int t_begin = split * per;
    const int t_end = min(t_begin + per, max_vis);

    for (int tok = t_begin + g; tok < t_end; tok += 16) {
        const int lp = tok / page_size;
        const int slot = tok - lp * page_size;
        const int page_id = lp < max_pages ? block_table[(long)b * max_pages + lp] : -1;
        if (page_id < 0) continue;
        const long off = (((long)page_id * num_kv_heads + kvh) * page_size + slot) * D;
        float kf[VD], vf[VD];
        float k_s = 1.f, v_s = 1.f;  // KV8: the rows' scales
        if constexpr (KV8) {
            const u32x2 kc = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint8_t *>(key_pages) + off + t * VD);
            const u32x2 vc = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint8_t *>(value_pages) + off + t * VD);
            k_s = key_scales[off / D];
            v_s = value_scales[off / D];
            kv8_unpack8(kc, kf);
            kv8_unpack8(vc, vf);
        } else if constexpr (VEC) {
            S kr[VD], vr[VD];
            constexpr int BYTES = VD * sizeof(S);
            if constexpr (BYTES == 16) {
                *reinterpret_cast<uint4 *>(kr) = *reinterpret_cast<const uint4 *>(key_pages + off + t * VD);
                *reinterpret_cast<uint4 *>(vr) = *reinterpret_cast<const uint4 *>(value_pages + off + t * VD);
            } else if constexpr (BYTES == 32) {
                reinterpret_cast<uint4 *>(kr)[0] = reinterpret_cast<const uint4 *>(key_pages + off + t * VD)[0];
                reinterpret_cast<uint4 *>(kr)[1] = reinterpret_cast<const uint4 *>(key_pages + off + t * VD)[1];
                reinterpret_cast<uint4 *>(vr)[0] = reinterpret_cast<const uint4 *>(value_pages + off + t * VD)[0];
                reinterpret_cast<uint4 *>(vr)[1] = reinterpret_cast<const uint4 *>(value_pages + off + t * VD)[1];
            } else {
                *reinterpret_cast<uint2 *>(kr) = *reinterpret_cast<const uint2 *>(key_pages + off + t * VD);
                *reinterpret_cast<uint2 *>(vr) = *reinterpret_cast<const uint2 *>(value_pages + off + t * VD);
            }
#pragma unroll
            for (int i = 0; i < VD; ++i) {
                kf[i] = TT::to_float(kr[i]);
                vf[i] = TT::to_float(vr[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VD; ++i) {
                const int d = t + 16 * i;
                kf[i] = d < D ? TT::to_float(key_pages[off + d]) : 0.f;
                vf[i] = d < D ? TT::to_float(value_pages[off + d]) : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < PD_RQ; ++r) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < VD; ++i) part += qv[r][i] * kf[i];
            const float score = KV8 ? group16_sum(part) * k_s : group16_sum(part);
            if (tok < rq_vis[r]) {
                const float nm = fmaxf(m[r], score);
                const float of = exp2_hw(m[r] - nm);
                const float sf = exp2_hw(score - nm);
                l[r] = l[r] * of + sf;
                const float sfv = KV8 ? sf * v_s : sf;
#pragma unroll
                for (int i = 0; i < VD; ++i) acc[r][i] = acc[r][i] * of + sfv * vf[i];
                m[r] = nm;
            }
        }
    }

    // merge the 16 groups
#pragma unroll
    for (int r = 0; r < PD_RQ; ++r) {
        float *dst = psm + ((long)g * PD_RQ + r) * stride;
#pragma unroll
        for (int i = 0; i < VD; ++i) {
            const int d = VEC ? t * VD + i : t + 16 * i;
            if (d < D) dst[d] = acc[r][i];
        }
        if (t == 0) {
            dst[D] = m[r];
            dst[D + 1] = l[r];
        }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < PD_RQ * D; item += 256) {
        const int r = item / D;
        const int d = item - r * D;
        const int rr = chunk * PD_RQ + r;
        if (rr >= R) continue;
        const int hq_o = rr / L;
        const int qp_o = rr - hq_o * L;
        float gm = -1e30f;
#pragma unroll
        for (int j = 0; j < 16; ++j) gm = fmaxf(gm, psm[((long)j * PD_RQ + r) * stride + D]);
        float gl = 0.f, vs = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *src = psm + ((long)j * PD_RQ + r) * stride;
            const float f = exp2_hw(src[D] - gm);
            gl += src[D + 1] * f;
            vs += src[d] * f;
        }
        const long orow = ((long)b * num_heads + kvh * rep + hq_o) * L + qp_o;
        if (n_splits == 1) {
            out[orow * D + d] = TT::from_float(gl == 0.f ? 0.f : vs / gl);
        } else {
            float *w = ws + (orow * n_splits + split) * stride;
            w[d] = vs;
            if (d == 0) {
                w[D] = gm;
                w[D + 1] = gl;
            }
        }
    }
}

template <typename TT>
__global__ __launch_bounds__(128) void paged_merge_kernel(const float *__restrict__ ws,
                                                          typename TT::storage *__restrict__ out, int D,
                                                          int n_splits) {
    const long orow = blockIdx.x;
    const int stride = D + 2;
    const float *base = ws + orow * n_splits * stride;
    float gm = -1e30f;
    for (int s = 0; s < n_splits; ++s) gm = fmaxf(gm, base[s * stride + D]);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float gl = 0.f, vs = 0.f;
        for (int s = 0; s < n_splits; ++s) {
            const float f = exp2_hw(base[s * stride + D] - gm);
            gl += base[s * stride + D + 1] * f;
            vs += base[s * stride + d] * f;
        }
        out[orow * D + d] = TT::from_float(gl == 0.f ? 0.f : vs / gl);
    }
}

// ---------------------------------------------------------------------------
// Paged FlashAttention-2 on bf16 MFMA (D = 128, L > 8).
//   reference: paged_attention_mma_bf16_d128 (paged_attention.metal:250-506),
//   BQ=64/BK=32 with 8x8 simdgroup MMA.  gfx950 design:
//   * workgroup = 4 waves; each wave owns a 32-row query block of ONE query
//     head; the 4 waves of a workgroup are 4 (head, q-block) items of the same
//     KV head, so a K/V tile staged in LDS is shared by the whole GQA group.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: a lane then holds 16 scores of
//     ONE query row, so the softmax row reduction is in-register plus a single
//     cross-half exchange, and P^T is already in B-operand layout for
//     O^T = V^T P^T (the k index of that product is permuted consistently:
//     k = 16s + 4h + (j&3) + 8(j>>2)).
//   * K tile row-major + XOR swizzle (ds_read_b128 conflict-free), V tile
//     stored transposed [dim][token] with a 36-element row so the A fragment is
//     two ds_read_b64.
//   * next tile's global loads are issued before the current tile's MFMAs.
//   fp32 scores/softmax, P rounded to bf16 before PV, causal tile skipping and
//   zero output for empty rows, as the reference.
// ---------------------------------------------------------------------------
#ifndef FA_ABL
#define FA_ABL 0  // tools/lab/fa_lab only: 1 no softmax arithmetic, 2 no PV MFMA, 4 no QK MFMA, 8 no LDS staging stores, 16 no K/V reloads
#endif
constexpr int FA_BK = 64;            // tokens staged per barrier phase (two 32-token MFMA sub-tiles)
constexpr int FA_SUB = FA_BK / 32;
constexpr int FA_VROW = 128 + 32;    // bf16 elements per row of the row-major V tile: rows 320 B apart put the 4 rows x 16 dims a
                                     // 16-lane group gathers with ds_read_b64_tr_b16 on distinct banks (16 r + 8 g + 2 q + {0, 1})

// ONEPAGE: the page size is a power of two >= FA_BK, so a 64-token stage lies inside ONE page.  Its page id is then a single
// wave-uniform word and every K/V address of the stage is (uniform base of the page's rows) + (a per-thread offset fixed for
// the whole kernel): the per-chunk page lookups (8 vector loads per thread and stage) and the 64-bit address chains behind them
// (~20 VALU instructions per chunk, as much as the softmax of the stage) disappear.
// QR = 32-row query blocks per wave (round 3): with two, every K and V fragment read from LDS feeds two MFMAs and a workgroup
// covers 4 heads x 64 query rows per K/V tile it stages -- half the staging, half the fragment reads per flop; one wave per SIMD.
// KV8: the pages hold E4M3 codes (one byte per element of the same layout) and key_scales / value_scales one power-of-two float32 per
// (page, kv head, slot) row (kv8.h).  A thread requests 8 bytes per chunk instead of 16, plus its rows' scales, and converts to the bf16
// values a bf16 page would hold (exact) when it stores the chunk into the LDS tiles: everything behind the staging is unchanged.
// NW = waves per workgroup.  4: two workgroups per CU, a single-buffered K/V tile, two barriers per stage, the next stage's rows requested
// into registers behind the second one.  8 (round 6, ONEPAGE only): ONE workgroup per CU whose 8 waves -- the 4 query heads of the KV head x two
// consecutive 32-row query blocks -- share the staged tile (half the requests and half the staging stores per flop), the tile DOUBLE-buffered
// in LDS: per stage ONE barrier, then the rows of stage s + 2 are requested (two register sets: a stage's rows have a whole stage AND its
// compute to arrive, not one), stage s is computed from buffer s & 1, and the rows of stage s + 1 are stored into the other buffer.  The
// arithmetic of a (head, query block) is the same instruction sequence in both: the results are bit-identical.
// wave-uniform 32-bit load through the scalar cache (its own counter: it neither waits for nor delays the vector loads in flight)
__device__ __forceinline__ void fa_sload_i32(const int32_t *ptr, int &dst) { asm volatile("s_load_dword %0, %1, 0x0" : "=s"(dst) : "s"(ptr)); }
__device__ __forceinline__ void fa_sload_wait(int &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
constexpr int FA_KS_ELEMS = FA_BK * 128, FA_VS_ELEMS = FA_BK * FA_VROW;
constexpr size_t fa_pipe_lds_bytes() { return (size_t)2 * (FA_KS_ELEMS + FA_VS_ELEMS) * 2 + 2 * FA_BK * sizeof(int); }
template <bool ONEPAGE, int QR = 1, bool KV8 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, (QR == 1 && NW == 4) ? 2 : 1) void paged_fa_bf16_d128_kernel(
    const uint16_t *__restrict__ q, const void *__restrict__ key_pages_v, const void *__restrict__ value_pages_v,
    const int32_t *__restrict__ block_table, const int32_t *__restrict__ context_lens, uint16_t *__restrict__ out,
    float *__restrict__ ws, int n_splits, int L, int page_size, int page_shift, int max_pages, int num_heads, int num_kv_heads,
    float scale, int is_causal, int xcd_remap, const float *__restrict__ key_scales = nullptr,
    const float *__restrict__ value_scales = nullptr) {
    constexpr int D = 128;
    using KVE = typename std::conditional<KV8, uint8_t, uint16_t>::type;  // a page element
    using KVC = typename std::conditional<KV8, u32x2, u32x4>::type;        // a chunk of 8 of them
    const KVE *__restrict__ key_pages = reinterpret_cast<const KVE *>(key_pages_v);
    const KVE *__restrict__ value_pages = reinterpret_cast<const KVE *>(value_pages_v);
    constexpr bool PIPE = NW == 8;
    static_assert(NW == 4 || (NW == 8 && ONEPAGE && QR == 1), "8 waves: pages of 64+ tokens, one row block per wave");
    constexpr int NT = 64 * NW;                  // threads
    constexpr int CPT = FA_BK * 16 / NT;         // 16-byte K (and V) chunks per thread and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_dyn[];  // PIPE: K[2] | V[2] | tile_page[2]
    __shared__ __attribute__((aligned(16))) uint16_t ks_one[PIPE ? 8 : FA_KS_ELEMS];  // [token][dim] swizzled
    __shared__ __attribute__((aligned(16))) uint16_t vs_one[PIPE ? 8 : FA_VS_ELEMS];  // [token][dim] as it lies in the page; read transposed
    __shared__ int tile_page_one[PIPE ? 2 : 2 * FA_BK];
    uint16_t *const ks = PIPE ? reinterpret_cast<uint16_t *>(fa_dyn) : ks_one;
    uint16_t *const vs = PIPE ? reinterpret_cast<uint16_t *>(fa_dyn) + 2 * FA_KS_ELEMS : vs_one;
    int *const tile_page = PIPE ? reinterpret_cast<int *>(fa_dyn + (size_t)2 * (FA_KS_ELEMS + FA_VS_ELEMS) * 2) : tile_page_one;  // [2][FA_BK]

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int l32 = lane & 31;
    const int h = lane >> 5;
    // XCD-aware order (xcd_remap): workgroups are dealt to the 8 XCDs round-robin in launch order, so in launch order the K/V
    // pages of ONE kv head are streamed through all 8 L2s (4 MiB each; one head's K + V at 8k tokens is 4 MiB).  Remapped, the
    // workgroups an XCD receives are one contiguous range of the (kv head, x) order: with 8 kv heads, one head per XCD.
    int bx = blockIdx.x, kvh = blockIdx.y;
    if (xcd_remap) {
        const int gx = gridDim.x, T = gx * (int)gridDim.y;
        const int l = blockIdx.x + gx * blockIdx.y;
        const int c = l & 7;
        const int t = c * (T >> 3) + min(c, T & 7) + (l >> 3);
        kvh = t / gx;
        bx = t - kvh * gx;
    }
    const int b = blockIdx.z;
    const int rep = num_heads / num_kv_heads;
    const int QB = (L + 32 * QR - 1) / (32 * QR);  // query blocks of 32 QR rows: one per wave
    const int items = rep * QB;
    // bx = item block * n_splits + context split: with few query rows (chunked prefill of a long prompt) the KV
    // range is cut into n_splits pieces, one workgroup each, merged by paged_merge_kernel (flash-decoding style)
    const int split = bx % n_splits;
    // causal chunks: the LAST query blocks see the most tokens -- they go first (launch order = dispatch order), the light ones fill the tail
    const int item_block = is_causal ? ((int)gridDim.x / n_splits - 1 - bx / n_splits) : bx / n_splits;
    const int item = item_block * NW + wave;
    const bool wave_live = item < items;
    const int hq = wave_live ? item % rep : 0;
    const int qb = wave_live ? item / rep : 0;
    const int n = b * num_heads + kvh * rep + hq;
    const int ctx = context_lens[b];
    const float scale_log2 = scale * LOG2E;
    int qrow[QR];
    bool q_valid[QR];
    // Q^T fragments (B operand): lane (qrow, half h), step s -> dims 16s + 8h .. +8
    u32x4 qf[QR][8];
    f32x16 o[QR][4];
    float run_max[QR], run_sum[QR];
#pragma unroll
    for (int rb = 0; rb < QR; ++rb) {
        qrow[rb] = (qb * QR + rb) * 32 + l32;
        q_valid[rb] = wave_live && qrow[rb] < L;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (q_valid[rb]) {
                qf[rb][s] = *reinterpret_cast<const u32x4 *>(q + ((long)n * L + qrow[rb]) * D + 16 * s + 8 * h);
            } else {
                qf[rb][s] = u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[rb][db][r] = 0.f;
        run_max[rb] = -INFINITY;
        run_sum[rb] = 0.f;
    }

    // tile range: block-level limit = max over this block's items
    const int total_tiles = (ctx + 31) / 32;
    int my_tiles = total_tiles;
    if (is_causal) {
        const int last_query = min((qb + 1) * 32 * QR, L) - 1;
        const int last_key = last_query + (ctx - L);
        my_tiles = min(max((last_key + 1 + 31) / 32, 0), total_tiles);
    }
    if (!wave_live) my_tiles = 0;
    int blk_tiles;
    {
        // items of a block differ at most in q-block; the last wave_live item has the largest limit
        const int last_item = min(item_block * NW + NW - 1, items - 1);
        const int last_qb = last_item / rep;
        if (is_causal) {
            const int lq = min((last_qb + 1) * 32 * QR, L) - 1;
            const int lk = lq + (ctx - L);
            blk_tiles = min(max((lk + 1 + 31) / 32, 0), total_tiles);
        } else {
            blk_tiles = total_tiles;
        }
    }

    // staging: thread -> (token = c/16, chunk = c%16) for c = tid, tid+256
    // A STAGE is FA_BK tokens (FA_SUB sub-tiles of 32): one pair of barriers and one global round trip per stage.  With
    // 32-token stages the MFMA work of a stage (~0.5 us) could not cover the latency of the next stage's rows.
    struct Stg {  // a stage's rows on their way from the page to the LDS tiles
        KVC k[CPT], v[CPT];
        float ksc[KV8 ? CPT : 1], vsc[KV8 ? CPT : 1];  // KV8: the scale of each chunk's row
        bool ok[CPT];
        int pg[CPT];        // the page each chunk's token lives on (-1: none); ONEPAGE: pg[0] for all
        bool full;          // ONEPAGE: every row of the stage is a live token (no zeroing at the store)
    };
    Stg sa;
    // page ids travel one stage ahead of the K/V rows they address: a stage's loads are then ONE global round trip behind
    // the MFMAs of the previous stage instead of two dependent ones (block table, then rows)
    // K and V chunk c = tid + 256 i  ->  (token c / 16, 16-byte chunk c % 16): coalesced rows, b128 stores into the swizzled K tile
    // and into the row-major V tile (round 3: the PV fragments are gathered by ds_read_b64_tr_b16; until then V was stored
    // transposed with 16 ds_write_b32 per thread and stage).
    int pid_reg[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) sa.ok[i] = false;
    sa.full = false;
    // page_shift = log2(page_size), or -1 (integer division, ~30 VALU ops each, 16 of them per stage and thread)
    auto logical_page = [&](int tok) { return page_shift >= 0 ? (tok >> page_shift) : tok / page_size; };
    // The id is NOT touched here (no "in ? id : -1"): any use of the loaded word right behind the load makes the compiler wait
    // for it on the spot -- and, loads returning in issue order, for the K/V rows of the next stage requested just before it,
    // i.e. the stage prefetch stopped overlapping the MFMAs (r02: found in the ISA as vmcnt(0) behind load_pids).  Whether the
    // token has a page at all is kept as a flag computed from the token index alone and applied where the id is used.
    bool pid_in[CPT];
    auto page_of_token = [&](int tok, bool &in) {
        const int lp = logical_page(tok);
        in = tok < ctx && lp < max_pages;
        return block_table[(long)b * max_pages + (in ? lp : 0)];  // unconditional load from a clamped address
    };
    int page_next = -1;  // ONEPAGE: the (uniform) page id of the stage whose rows are requested next
    auto load_pids = [&](int stage) {
        if constexpr (ONEPAGE) {
            const int lp = (stage * FA_BK) >> page_shift;
            const int id = block_table[(long)b * max_pages + min(lp, max_pages - 1)];  // same address in every lane
            page_next = id;                                                              // made uniform where it is used
            pid_in[0] = lp < max_pages;
        } else {
#pragma unroll
            for (int i = 0; i < CPT; ++i) pid_reg[i] = page_of_token(stage * FA_BK + ((tid + i * NT) >> 4), pid_in[i]);
        }
    };
    auto stage_load = [&](int stage, Stg &g, int page_arg = -1) {
        if constexpr (ONEPAGE) {
            const int page = PIPE ? page_arg : (pid_in[0] ? __builtin_amdgcn_readfirstlane(page_next) : -1);
            const int slot0 = (stage * FA_BK) & (page_size - 1);
            const long base = (((long)max(page, 0) * num_kv_heads + kvh) * page_size + slot0) * D;  // uniform
            const KVE *kbase = key_pages + base;
            const KVE *vbase = value_pages + base;
            g.full = page >= 0 && (stage + 1) * FA_BK <= ctx;  // uniform: nothing to zero when the store comes
            g.pg[0] = page;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + i * NT;  // token c / 16, chunk c % 16: the stage's 64 K rows are contiguous
                g.k[i] = *reinterpret_cast<const KVC *>(kbase + (size_t)c * 8);
                g.ok[i] = page >= 0 && stage * FA_BK + (c >> 4) < ctx;  // rows past the context are zeroed in LDS as before
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) g.v[i] = *reinterpret_cast<const KVC *>(vbase + (size_t)(tid + i * NT) * 8);
            if constexpr (KV8) {
                const long row0 = base / D;  // uniform
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    g.ksc[i] = key_scales[row0 + ((tid + i * NT) >> 4)];
                    g.vsc[i] = value_scales[row0 + ((tid + i * NT) >> 4)];
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            const int tok_in = c >> 4;
            const int ch = c & 15;
            const int tok = stage * FA_BK + tok_in;
            const int lp = logical_page(tok);
            const int slot = tok - lp * page_size;
            const int page_id = pid_in[i] ? pid_reg[i] : -1;
            // unconditional loads from a clamped address (a divergent branch around a load makes hipcc wait for it at the
            // join, i.e. before the MFMAs it should overlap); rows of unused pages are zeroed when they are stored to LDS
            const long off = (((long)max(page_id, 0) * num_kv_heads + kvh) * page_size + slot) * D + ch * 8;
            g.k[i] = *reinterpret_cast<const KVC *>(key_pages + off);
            g.v[i] = *reinterpret_cast<const KVC *>(value_pages + off);
            if constexpr (KV8) {
                g.ksc[i] = key_scales[off / D];
                g.vsc[i] = value_scales[off / D];
            }
            g.ok[i] = page_id >= 0;
            g.pg[i] = page_id;
        }
    };

    // a stage's rows into K / V tile `buf` (PIPE: 0 / 1; else 0) and its page ids into tile_page[tp]
    auto stage_store = [&](const Stg &g, int buf, int tp) {
        uint16_t *ksb = ks + buf * FA_KS_ELEMS, *vsb = vs + buf * FA_VS_ELEMS;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            const int tok_in = c >> 4;
            const int ch = c & 15;
            u32x4 kk;
            if constexpr (KV8) kk = kv8_to_bf16x8(g.k[i], g.ksc[i]);
            else kk = g.k[i];
            if (!(ONEPAGE && g.full) && !g.ok[i]) kk = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4 *>(&ksb[tok_in * D + ((ch ^ (tok_in & 15)) * 8)]) = kk;
            if (ch == 0) tile_page[tp * FA_BK + tok_in] = ONEPAGE ? (g.ok[i] ? g.pg[0] : -1) : g.pg[i];  // read behind the next barrier
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {  // V rows as they lie in the page: one b128 store per chunk (the PV fragments are read transposed)
            const int c = tid + i * NT;
            u32x4 vv;
            if constexpr (KV8) vv = kv8_to_bf16x8(g.v[i], g.vsc[i]);
            else vv = g.v[i];
            if (!(ONEPAGE && g.full) && !g.ok[i]) vv = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4 *>(&vsb[(c >> 4) * FA_VROW + (c & 15) * 8]) = vv;
        }
    };

    // this lane's corner of the [4 tokens][16 dims] block its 16-lane group gathers (group = (dim half g, token half h))
    typedef short v4s16 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s16 lds_v4s16;
    uint16_t *vtr = vs + (4 * h + ((lane & 15) >> 2)) * FA_VROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    // splits are whole stages; tile counts (total / mine / the block's) stay in 32-token units
    const int total_stages = (total_tiles + FA_SUB - 1) / FA_SUB;
    const int blk_stages = (blk_tiles + FA_SUB - 1) / FA_SUB;
    const int stages_per_split = (total_stages + n_splits - 1) / n_splits;
    const int stage_begin = split * stages_per_split;
    const int stage_end = min(stage_begin + stages_per_split, blk_stages);
    // one stage's two products and its softmax, from K / V tile `buf` and the page ids in tile_page[tp]
    auto compute_stage = [&](int stage, int buf, int tp) {
        const uint16_t *ksb = ks + buf * FA_KS_ELEMS;
        const uint16_t *vtrb = vtr + buf * FA_VS_ELEMS;
        const int *tpb = tile_page + tp * FA_BK;
        static_assert(FA_SUB == 2, "a stage is two 32-token sub-tiles");
        const int tile0 = stage * FA_SUB;
        if (tile0 >= my_tiles) return;          // wave-uniform: this wave's rows see nothing in this stage
        const bool two = tile0 + 1 < my_tiles;  // ... or only its first 32 tokens (the causal diagonal)

        // S^T = K Q^T for BOTH sub-tiles first (round 6): 16 independent-of-the-softmax MFMAs in a row, then ONE softmax update per 64
        // tokens -- one running-maximum step, one rescale of the output tile, two cross-half exchanges instead of four -- then the 16 MFMAs
        // of the second product.  (Until then: product, softmax, product per 32 tokens.)  A K fragment feeds the MFMA of every row block.
        f32x16 sacc[FA_SUB][QR];
#pragma unroll
        for (int sub = 0; sub < FA_SUB; ++sub) {
#pragma unroll
            for (int rb = 0; rb < QR; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[sub][rb][r] = 0.f;
            if (sub == 1 && !two) continue;
            const int tb = sub * 32;  // token offset of the sub-tile inside the stage
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int ch = (2 * s + h) ^ (l32 & 15);
                const u32x4 kf = *reinterpret_cast<const u32x4 *>(&ksb[(tb + l32) * D + ch * 8]);
#pragma unroll
                for (int rb = 0; rb < QR; ++rb) {
                    if constexpr (FA_ABL & 4) sacc[sub][rb][s] += __uint_as_float((kf[0] ^ qf[rb][s][1]) & 0x3f800000u);
                    else
                    sacc[sub][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                                             __builtin_bit_cast(bf16x8_t, qf[rb][s]), sacc[sub][rb], 0, 0, 0);
                }
            }
        }
        u32x4 pf[FA_SUB][QR][2];
#pragma unroll
        for (int rb = 0; rb < QR; ++rb) {
        // mask + scale ; lane holds tokens (r&3)+8(r>>2)+4h of each sub-tile for query row qrow[rb].
        // Interior stages (wave-uniform test: every token is inside the context, on a live page, and at or below the causal diagonal of
        // the row block's FIRST query row) need no per-element test: most stages of a long context are interior.
        float tmax = -INFINITY;
        const int last_tok = tile0 * 32 + FA_BK - 1;
        const bool interior = two && page_shift >= 5 && last_tok < ctx && tpb[0] >= 0 && tpb[32] >= 0 &&
                              (!is_causal || last_tok <= (qb * QR + rb) * 32 + (ctx - L));
        if (interior) {  // raw scores here; the scale goes into the exponent's FMA below
#pragma unroll
            for (int sub = 0; sub < FA_SUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[sub][rb][r]);
            tmax *= scale_log2;  // scale > 0: max and scaling commute
        } else {
#pragma unroll
            for (int sub = 0; sub < FA_SUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int in_stage = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int tok = tile0 * 32 + in_stage;
                    bool valid = (sub == 0 || two) && q_valid[rb] && tok < ctx && tpb[in_stage] >= 0;
                    if (is_causal) valid = valid && tok <= qrow[rb] + (ctx - L);
                    sacc[sub][rb][r] = valid ? sacc[sub][rb][r] * scale_log2 : -INFINITY;
                    tmax = fmaxf(tmax, sacc[sub][rb][r]);
                }
        }
        if constexpr (FA_ABL & 1) {
            run_sum[rb] += sacc[0][rb][0] + sacc[1][rb][0];
        } else {
        tmax = fmaxf(tmax, lane_xor32(tmax, lane));  // v_permlane32_swap: one VALU instruction (a ds_bpermute is a trip through the LDS crossbar)
        const float new_max = fmaxf(run_max[rb], tmax);
        float prev_scale, tsum = 0.f;
        if (interior) {  // every score is finite: exp2f(-inf) of the first stage's running maximum is the wanted 0
            prev_scale = exp2_hw(run_max[rb] - new_max);
#pragma unroll
            for (int sub = 0; sub < FA_SUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[sub][rb][r] = exp2_hw(fmaf(sacc[sub][rb][r], scale_log2, -new_max));
                    tsum += sacc[sub][rb][r];
                }
        } else {
            const bool finite_row = q_valid[rb] && new_max != -INFINITY;
            prev_scale = (run_max[rb] == -INFINITY || !finite_row) ? 0.f : exp2_hw(run_max[rb] - new_max);
#pragma unroll
            for (int sub = 0; sub < FA_SUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = (sacc[sub][rb][r] == -INFINITY || !finite_row) ? 0.f : exp2_hw(sacc[sub][rb][r] - new_max);
                    sacc[sub][rb][r] = p;
                    tsum += p;
                }
        }
        tsum += lane_xor32(tsum, lane);
        run_max[rb] = new_max;
        run_sum[rb] = prev_scale * run_sum[rb] + tsum;
        if (!__all(prev_scale == 1.0f)) {  // once the running maxima have settled the rescale is the identity for the whole wave
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[rb][db][r] *= prev_scale;
        }
        }  // FA_ABL & 1

        // P^T fragments (B operand), step s uses regs 8s..8s+7; rounded to bf16 before the product like the reference
        // (paged_attention.metal:439-444)
#pragma unroll
        for (int sub = 0; sub < FA_SUB; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) pf[sub][rb][s][e] = BF16::pack2(sacc[sub][rb][8 * s + 2 * e], sacc[sub][rb][8 * s + 2 * e + 1]);
        }  // row block

        // O^T += V^T P^T: a V fragment (gathered transposed from the row-major tile) feeds every row block
#pragma unroll
        for (int sub = 0; sub < FA_SUB; ++sub) {
            if (sub == 1 && !two) continue;  // (its weights are all zero)
            const int tb = sub * 32;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // A fragment of V^T: lane (dim db*32 + l32, half h) needs tokens tb + 16 s + 4 h + {0..3} and + 8 + {0..3} of its dim.
                    // ds_read_b64_tr_b16: in a 16-lane group lane c hands in the address of [row c >> 2][4 dims from 4 (c & 3)] of a
                    // [4 tokens][16 dims] block and receives the block's column c (tools/lab/tr_probe.hip) -- V stays row-major.
                    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16 *)(vtrb + (tb + 16 * s) * FA_VROW + db * 32));
                    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16 *)(vtrb + (tb + 16 * s + 8) * FA_VROW + db * 32));
                    const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
                    const u32x4 vf = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
                    for (int rb = 0; rb < QR; ++rb) {
                        if constexpr (FA_ABL & 2) o[rb][db][s] += __uint_as_float((vf[0] ^ pf[sub][rb][s][1]) & 0x3f800000u);
                        else
                        o[rb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                                            __builtin_bit_cast(bf16x8_t, pf[sub][rb][s]), o[rb][db], 0, 0, 0);
                    }
                }
            }
        }
    };

    if constexpr (!PIPE) {
        if (stage_begin < stage_end) {
            load_pids(stage_begin);
            stage_load(stage_begin, sa);
            if (stage_begin + 1 < stage_end) load_pids(stage_begin + 1);
        }
        for (int stage = stage_begin; stage < stage_end; ++stage) {
            __syncthreads();  // previous stage's LDS reads are complete
            if (!(FA_ABL & 8) || stage == stage_begin) stage_store(sa, 0, stage & 1);
            __syncthreads();
            if (stage + 1 < stage_end && !(FA_ABL & 16)) {
                stage_load(stage + 1, sa);
                if (stage + 2 < stage_end) load_pids(stage + 2);
            }
            compute_stage(stage, 0, stage & 1);
        }
    } else if (stage_begin < stage_end) {
        // page ids by scalar loads, one stage ahead of the rows they address; every row request stands outside any branch (a stage past the
        // end requests the last one again: the rows are not stored)
        const int32_t *brow = block_table + (long)b * max_pages;
        const int last = stage_end - 1;
        auto page_slot = [&](int stage) { return brow + min((stage * FA_BK) >> page_shift, max_pages - 1); };
        auto page_live = [&](int stage, int id) { return ((stage * FA_BK) >> page_shift) < max_pages ? id : -1; };
        int pg0, pg1, pg_n;
        fa_sload_i32(page_slot(stage_begin), pg0);
        fa_sload_i32(page_slot(min(stage_begin + 1, last)), pg1);
        fa_sload_i32(page_slot(min(stage_begin + 2, last)), pg_n);
        fa_sload_wait(pg0);
        fa_sload_wait(pg1);
        fa_sload_wait(pg_n);
        stage_load(stage_begin, sa, page_live(stage_begin, pg0));
        stage_store(sa, 0, 0);
        stage_load(min(stage_begin + 1, last), sa, page_live(min(stage_begin + 1, last), pg1));
        // iteration of stage s (r = s - stage_begin): the registers hold the rows of s + 1 (requested one iteration ago): they go into the
        // other tile, the rows of s + 2 are requested into the same registers, stage s is computed
        for (int stage = stage_begin, r = 0; stage < stage_end; ++stage, ++r) {
            __syncthreads();  // stage s is in tile r & 1; everyone is done reading the other tile
            if (stage + 1 < stage_end && !(FA_ABL & 8)) stage_store(sa, (r + 1) & 1, (r + 1) & 1);
            const int s2 = min(stage + 2, last);
            if (!(FA_ABL & 16)) stage_load(s2, sa, page_live(s2, pg_n));
            fa_sload_i32(page_slot(min(stage + 3, last)), pg_n);  // waited for at the end of this iteration
            compute_stage(stage, r & 1, r & 1);
            fa_sload_wait(pg_n);
        }
    }

#pragma unroll
    for (int rb = 0; rb < QR; ++rb) {
        if (!q_valid[rb]) continue;
        if (n_splits > 1) {
            // un-normalised partial: D values, running max (log2 domain), running sum
            float *w = ws + (((long)n * L + qrow[rb]) * n_splits + split) * (D + 2);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[db * 32 + 8 * rg + 4 * h + e] = o[rb][db][4 * rg + e];
            if (h == 0) {
                w[D] = run_max[rb] == -INFINITY ? -1e30f : run_max[rb];
                w[D + 1] = run_sum[rb];
            }
            continue;
        }
        uint16_t *orow = out + ((long)n * L + qrow[rb]) * D;
        const float inv = run_sum[rb] == 0.f ? 0.f : 1.0f / run_sum[rb];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                // regs 4rg..4rg+3 -> dims db*32 + 8rg + 4h + 0..3
                u32x2 pk;
                pk[0] = BF16::pack2(o[rb][db][4 * rg + 0] * inv, o[rb][db][4 * rg + 1] * inv);
                pk[1] = BF16::pack2(o[rb][db][4 * rg + 2] * inv, o[rb][db][4 * rg + 3] * inv);
                *reinterpret_cast<u32x2 *>(orow + db * 32 + 8 * rg + 4 * h) = pk;
            }
        }
    }
}

// context splits of the MFMA prefill kernel: only when the (head, query block) items alone leave most CUs idle, and never
// below 256 tokens per split.  `wg_target`: 512 workgroups of 4 waves (two per CU) or 256 of 8.
static int pick_fa_splits(int B, int Hkv, int item_blocks, int max_ctx, int wg_target = 512) {
    const int base = std::max(1, B * Hkv * item_blocks);
    int s = wg_target / base;
    s = std::min(s, std::max(1, max_ctx / 256));
    s = std::min(s, 32);
    return std::max(s, 1);
}

static int pick_splits(int B, int Hkv, int row_chunks, int max_ctx) {
    const int base = std::max(1, B * Hkv * row_chunks);
    int s = (512 + base - 1) / base;
    const int by_len = std::max(1, max_ctx / 64);
    s = std::min(s, by_len);
    s = std::min(s, 64);
    return std::max(s, 1);
}

}  // namespace tl

using namespace tl;

extern "C" int tl_decode_attention(const void *q, const void *k, const void *v, const float *mask, void *out,
                                   int q_rows, int L, int S, int D, int num_heads, int num_kv_heads, float scale,
                                   int is_causal, int has_mask, tl_dtype dtype, void *stream) {
    TL_REQUIRE(q && k && v && out, "decode_attention: null pointer");
    TL_REQUIRE(D > 0 && D <= 256 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 &&
                   q_rows % num_heads == 0,
               "decode_attention: incompatible attention shapes");
    TL_REQUIRE(!has_mask || mask, "decode_attention: mask must have shape [B*Hq,L,S]");
    if (q_rows == 0 || L == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(16 * D + 32) * sizeof(float);
    const dim3 grid(q_rows * L), block(256);
#define DA_LAUNCH(TT)                                                                                            \
    hipLaunchKernelGGL((decode_attention_kernel<TT>), grid, block, lds, st, (const typename TT::storage *)q,     \
                       (const typename TT::storage *)k, (const typename TT::storage *)v, mask,                   \
                       (typename TT::storage *)out, q_rows, L, S, D, num_heads, num_kv_heads, scale, is_causal, \
                       has_mask)
    switch (dtype) {
        case TL_F32: DA_LAUNCH(F32); break;
        case TL_F16: DA_LAUNCH(F16); break;
        case TL_BF16: DA_LAUNCH(BF16); break;
        default: return fail(TL_ERR_INVALID, "decode_attention: expected float32, float16, or bfloat16");
    }
#undef DA_LAUNCH
    TL_CHECK_LAUNCH("decode_attention");
    return TL_OK;
}

extern "C" int tl_paged_cache_update(void *pages, const void *values, int num_pages, int heads, int page_size,
                                     int head_dim, int length, int page_id, int start, tl_dtype dtype, void *stream) {
    TL_REQUIRE(dtype == TL_F32 || dtype == TL_BF16,
               "paged_cache_update: pages and values must have the same float32 or bfloat16 dtype");
    TL_REQUIRE(pages && values, "paged_cache_update: null pointer");
    TL_REQUIRE(heads > 0 && page_size > 0 && head_dim > 0 && length >= 0,
               "paged_cache_update: expected pages [P, H, page_size, D] and values [1, H, length, D]");
    TL_REQUIRE(page_id >= 0 && page_id < num_pages && start >= 0 && start + length <= page_size,
               "paged_cache_update: destination slice is outside page storage");
    if (length == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const long esz = dtype == TL_F32 ? 4 : 2;
    const long row_bytes = (long)head_dim * esz;
    const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)pages % 16 == 0) && ((uintptr_t)values % 16 == 0);
    if (a16) {
        const long total = (long)heads * length * (row_bytes / 16);
        hipLaunchKernelGGL((paged_cache_update_kernel<16>), dim3(ceil_div(total, 256)), dim3(256), 0, st,
                           (const char *)values, (char *)pages, heads, length, row_bytes, page_size, page_id, start);
    } else if (esz == 4) {
        const long total = (long)heads * length * (row_bytes / 4);
        hipLaunchKernelGGL((paged_cache_update_kernel<4>), dim3(ceil_div(total, 256)), dim3(256), 0, st,
                           (const char *)values, (char *)pages, heads, length, row_bytes, page_size, page_id, start);
    } else {
        const long total = (long)heads * length * (row_bytes / 2);
        hipLaunchKernelGGL((paged_cache_update_kernel<2>), dim3(ceil_div(total, 256)), dim3(256), 0, st,
                           (const char *)values, (char *)pages, heads, length, row_bytes, page_size, page_id, start);
    }
    TL_CHECK_LAUNCH("paged_cache_update");
    return TL_OK;
}

static bool paged_uses_fa(int L, int D, tl_dtype dtype) { return L > 8 && dtype == TL_BF16 && D == 128; }
// waves per workgroup of the FlashAttention prefill kernel: 8 (default) takes pages of 64+ tokens and chunks of 64+ query rows; 4 is the
// twin (and what everything else runs on).  tl_paged_attention_waves: test / lab hook, returns the previous value.
static int g_fa_waves = 8;
extern "C" int tl_paged_attention_waves(int waves) {
    const int before = g_fa_waves;
    if (waves == 4 || waves == 8) g_fa_waves = waves;
    return before;
}

extern "C" size_t tl_paged_attention_workspace_bytes(int N, int L, int D, int page_size, int max_pages, int num_heads,
                                                     int num_kv_heads, int max_context_hint) {
    if (N <= 0 || L <= 0 || num_heads <= 0 || num_kv_heads <= 0) return 0;
    const int B = N / num_heads;
    const int rep = num_heads / num_kv_heads;
    const int row_chunks = (rep * L + PD_RQ - 1) / PD_RQ;
    const int max_ctx = max_context_hint > 0 ? max_context_hint : max_pages * page_size;
    int splits = pick_splits(B, num_kv_heads, row_chunks, max_ctx);
    if (L > 8 && D == 128)  // the bf16 prefill kernel's own split rule (an fp32 call of this shape needs no more)
        splits = std::max(splits, pick_fa_splits(B, num_kv_heads, (rep * ((L + 31) / 32) + 3) / 4, max_ctx));
    if (splits <= 1) return 0;
    return (size_t)N * L * splits * (D + 2) * sizeof(float);
}

// key_scales != nullptr: FP8 pages (bf16 queries, head dimension 128)
static int paged_attention_impl(const void *q, const void *key_pages, const void *value_pages, const float *key_scales,
                                const float *value_scales, const int32_t *block_table, const int32_t *context_lens, void *out, int N,
                                int L, int D, int num_pages, int page_size, int max_pages, int num_heads, int num_kv_heads, float scale,
                                int is_causal, int max_context_hint, tl_dtype dtype, void *workspace, size_t workspace_bytes,
                                void *stream) {
    const bool kv8 = key_scales != nullptr;
    TL_REQUIRE(dtype == TL_F32 || dtype == TL_BF16,
               "paged_attention: q, key_pages, and value_pages must have the same float32 or bfloat16 dtype");
    TL_REQUIRE(q && key_pages && value_pages && block_table && context_lens && out, "paged_attention: null pointer");
    TL_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
               "paged_attention: num_heads must be divisible by num_kv_heads");
    TL_REQUIRE(N % num_heads == 0, "paged_attention: q.shape[0] must be divisible by num_heads");
    TL_REQUIRE(page_size > 0 && max_pages > 0 && num_pages > 0 && L > 0,
               "paged_attention: page tensors must be 4D [P, H_kv, page_size, D]");
    TL_REQUIRE(D > 0 && D <= 128, "paged_attention: head dimension must be at most 128");
    if (N == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const int B = N / num_heads;
    const int rep = num_heads / num_kv_heads;

    if (L > 8 && dtype == TL_BF16) {
        if (D != 128) return fail(TL_ERR_UNSUPPORTED, "paged_attention: bfloat16 prefill requires head dimension 128");
        // the FlashAttention tiles take the maximum of the RAW scores and scale afterwards: only valid for a positive scale.  The
        // decode and float32 kernels scale before they compare and take any float, like the reference's paged_attention.
        TL_REQUIRE(scale > 0.f, "paged_attention: the bfloat16 FlashAttention path needs a positive scale");
        // (the kernel is templated on QR, 32-row query blocks per wave; two of them -- every K / V fragment read feeds two MFMAs, half
        // the K/V staging per flop -- were measured in round 3 and lose at one wave per SIMD: 392 -> 573 us at 2,048 x 8,192,
        // profiles/r03_labs/prefill_fa_two_row_blocks.log; only QR = 1 is instantiated)
        constexpr int qr = 1;
        const int items = rep * ((L + 32 * qr - 1) / (32 * qr));
        int page_shift = -1;
        for (int sh = 0; sh < 30; ++sh)
            if ((1 << sh) == page_size) page_shift = sh;
        // 8 waves sharing a double-buffered tile (one workgroup per CU): 8 consecutive (head, query block) items, pages of 64+ tokens
        const bool w8 = g_fa_waves == 8 && page_shift >= 6 && L >= 64;
        const int nw = w8 ? 8 : 4;
        const int item_blocks = (items + nw - 1) / nw;
        const int max_ctx_fa = max_context_hint > 0 ? max_context_hint : max_pages * page_size;
        int fa_splits = pick_fa_splits(B, num_kv_heads, item_blocks, max_ctx_fa, w8 ? 256 : 512);
        const size_t fa_need = fa_splits > 1 ? (size_t)N * L * fa_splits * (D + 2) * sizeof(float) : 0;
        if (fa_need > 0 && (!workspace || workspace_bytes < fa_need)) fa_splits = 1;  // no workspace: one pass, still correct
        const dim3 grid(item_blocks * fa_splits, num_kv_heads, B);
        const int fa_xcd_remap = 1;  // one KV head per XCD (round 3 A/B: +2 % at 8k prefill)
#define FA_LAUNCH(ONEP, QRv, K8)                                                                                                \
        hipLaunchKernelGGL((paged_fa_bf16_d128_kernel<ONEP, QRv, K8>), grid, dim3(256), 0, st, (const uint16_t *)q, key_pages,  \
                           value_pages, block_table, context_lens, (uint16_t *)out, (float *)workspace, fa_splits, L, page_size, \
                           page_shift, max_pages, num_heads, num_kv_heads, scale, is_causal, fa_xcd_remap, key_scales,            \
                           value_scales)
#define FA_LAUNCH8(K8)                                                                                                           \
        do {                                                                                                                         \
            static bool attr_set_dev[64] = {};  /* per HIP device: a function attribute belongs to the device's code object */      \
            int dev_ = 0;                                                                                                            \
            (void)hipGetDevice(&dev_);                                                                                               \
            bool &attr_set = attr_set_dev[dev_ & 63];                                                                                \
            if (!attr_set) {                                                                                                         \
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&paged_fa_bf16_d128_kernel<true, 1, K8, 8>),                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)fa_pipe_lds_bytes()) != hipSuccess)         \
                    return fail(TL_ERR_HIP, "paged_attention: cannot size the prefill kernel's LDS");                               \
                attr_set = true;                                                                                                     \
            }                                                                                                                        \
            hipLaunchKernelGGL((paged_fa_bf16_d128_kernel<true, 1, K8, 8>), grid, dim3(512), fa_pipe_lds_bytes(), st,                \
                               (const uint16_t *)q, key_pages, value_pages, block_table, context_lens, (uint16_t *)out,             \
                               (float *)workspace, fa_splits, L, page_size, page_shift, max_pages, num_heads, num_kv_heads, scale,   \
                               is_causal, fa_xcd_remap, key_scales, value_scales);                                                   \
        } while (0)
        if (w8) {
            if (kv8) FA_LAUNCH8(true);
            else FA_LAUNCH8(false);
        } else if (kv8) {
            if (page_shift >= 6) FA_LAUNCH(true, 1, true);
            else FA_LAUNCH(false, 1, true);
        } else if (page_shift >= 6) FA_LAUNCH(true, 1, false);  // a 64-token stage never straddles pages
        else FA_LAUNCH(false, 1, false);
#undef FA_LAUNCH
#undef FA_LAUNCH8
        TL_CHECK_LAUNCH("paged_attention(prefill)");
        if (fa_splits > 1) {
            hipLaunchKernelGGL((paged_merge_kernel<BF16>), dim3(N * L), dim3(128), 0, st, (const float *)workspace,
                               (uint16_t *)out, D, fa_splits);
            TL_CHECK_LAUNCH("paged_attention(prefill merge)");
        }
        return TL_OK;
    }

    // decode (L <= 8) and the fp32 prefill fallback share the split-context kernel
    const int row_chunks = (rep * L + PD_RQ - 1) / PD_RQ;
    const int max_ctx = max_context_hint > 0 ? max_context_hint : max_pages * page_size;
    const int splits = pick_splits(B, num_kv_heads, row_chunks, max_ctx);
    const size_t need = splits > 1 ? (size_t)N * L * splits * (D + 2) * sizeof(float) : 0;
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(TL_ERR_INVALID, "paged_attention: workspace is missing or too small");
    const dim3 grid(splits * row_chunks, num_kv_heads, B), block(256);
    const size_t lds = (size_t)16 * PD_RQ * (D + 2) * sizeof(float);
    float *ws = (float *)workspace;
#define PD_LAUNCH(TT, VDv, VECv)                                                                                    \
    hipLaunchKernelGGL((paged_decode_kernel<TT, VDv, VECv>), grid, block, lds, st, (const typename TT::storage *)q, \
                       (const typename TT::storage *)key_pages, (const typename TT::storage *)value_pages,         \
                       block_table, context_lens, (typename TT::storage *)out, ws, L, D, page_size, max_pages,     \
                       num_heads, num_kv_heads, scale, is_causal, splits, row_chunks)
    if (kv8) {
        hipLaunchKernelGGL((paged_decode_kernel<BF16, 8, true, true>), grid, block, lds, st, (const uint16_t *)q,
                           (const uint16_t *)key_pages, (const uint16_t *)value_pages, block_table, context_lens, (uint16_t *)out, ws,
                           L, D, page_size, max_pages, num_heads, num_kv_heads, scale, is_causal, splits, row_chunks, key_scales,
                           value_scales);
    } else if (dtype == TL_BF16) {
        if (D == 128) PD_LAUNCH(BF16, 8, true);
        else if (D == 64) PD_LAUNCH(BF16, 4, true);
        else PD_LAUNCH(BF16, 8, false);
    } else {
        if (D == 128) PD_LAUNCH(F32, 8, true);
        else if (D == 64) PD_LAUNCH(F32, 4, true);
        else PD_LAUNCH(F32, 8, false);
    }
#undef PD_LAUNCH
    TL_CHECK_LAUNCH("paged_attention(decode)");
    if (splits > 1) {
        if (dtype == TL_BF16) {
            hipLaunchKernelGGL((paged_merge_kernel<BF16>), dim3(N * L), dim3(128), 0, st, ws, (uint16_t *)out, D, splits);
        } else {
            hipLaunchKernelGGL((paged_merge_kernel<F32>), dim3(N * L), dim3(128), 0, st, ws, (float *)out, D, splits);
        }
        TL_CHECK_LAUNCH("paged_attention(merge)");
    }
    (void)paged_uses_fa;
    return TL_OK;
}

extern "C" int tl_paged_attention(const void *q, const void *key_pages, const void *value_pages,
                                  const int32_t *block_table, const int32_t *context_lens, void *out, int N, int L,
                                  int D, int num_pages, int page_size, int max_pages, int num_heads, int num_kv_heads,
                                  float scale, int is_causal, int max_context_hint, tl_dtype dtype, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    return paged_attention_impl(q, key_pages, value_pages, nullptr, nullptr, block_table, context_lens, out, N, L, D, num_pages, page_size,
                                max_pages, num_heads, num_kv_heads, scale, is_causal, max_context_hint, dtype, workspace, workspace_bytes,
                                stream);
}

// ---- FP8 (E4M3) KV pages: the quantised twins of paged_cache_update / paged_attention (kv8.h, include/tinyllm_hip.h) ----------------
extern "C" int tl_kv_fp8_quantize_rows(const void *values, void *codes, float *scales, long rows, int head_dim, void *stream) {
    TL_REQUIRE(values && codes && scales, "kv_fp8_quantize_rows: null pointer");
    TL_REQUIRE(head_dim == 128 && rows >= 0 && rows < (1L << 30), "kv_fp8_quantize_rows: fewer than 2^30 rows of 128 bfloat16 values");
    if (rows == 0) return TL_OK;
    hipLaunchKernelGGL(kv8_quantize_rows_kernel, dim3(ceil_div(rows, 16)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)values,
                       (uint8_t *)codes, scales, rows, (int)rows, (int)rows, 0L, 0);
    TL_CHECK_LAUNCH("kv_fp8_quantize_rows");
    return TL_OK;
}

extern "C" int tl_kv_fp8_dequantize_rows(const void *codes, const float *scales, void *out, long rows, int head_dim, void *stream) {
    TL_REQUIRE(codes && scales && out, "kv_fp8_dequantize_rows: null pointer");
    TL_REQUIRE(head_dim == 128 && rows >= 0, "kv_fp8_dequantize_rows: rows of 128 codes");
    if (rows == 0) return TL_OK;
    hipLaunchKernelGGL(kv8_dequantize_rows_kernel, dim3(ceil_div(rows * 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)codes, scales, (uint16_t *)out, rows);
    TL_CHECK_LAUNCH("kv_fp8_dequantize_rows");
    return TL_OK;
}

extern "C" int tl_paged_cache_update_fp8(void *pages, float *page_scales, const void *values, int num_pages, int heads, int page_size,
                                         int head_dim, int length, int page_id, int start, void *stream) {
    TL_REQUIRE(pages && page_scales && values, "paged_cache_update_fp8: null pointer");
    TL_REQUIRE(head_dim == 128, "paged_cache_update_fp8: FP8 pages need head dimension 128");
    TL_REQUIRE(heads > 0 && page_size > 0 && length >= 0,
               "paged_cache_update_fp8: expected pages [P, H, page_size, 128] and bfloat16 values [1, H, length, 128]");
    TL_REQUIRE(page_id >= 0 && page_id < num_pages && start >= 0 && start + length <= page_size,
               "paged_cache_update_fp8: destination slice is outside page storage");
    if (length == 0) return TL_OK;
    const long rows = (long)heads * length;
    hipLaunchKernelGGL(kv8_quantize_rows_kernel, dim3(ceil_div(rows, 16)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)values,
                       (uint8_t *)pages, page_scales, rows, length, page_size, (long)page_id * heads * page_size, start);
    TL_CHECK_LAUNCH("paged_cache_update_fp8");
    return TL_OK;
}

extern "C" int tl_paged_attention_fp8(const void *q, const void *key_pages, const float *key_scales, const void *value_pages,
                                      const float *value_scales, const int32_t *block_table, const int32_t *context_lens, void *out,
                                      int N, int L, int D, int num_pages, int page_size, int max_pages, int num_heads, int num_kv_heads,
                                      float scale, int is_causal, int max_context_hint, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    TL_REQUIRE(key_scales && value_scales, "paged_attention_fp8: null scale pointer");
    TL_REQUIRE(D == 128, "paged_attention_fp8: FP8 pages need head dimension 128");
    return paged_attention_impl(q, key_pages, value_pages, key_scales, value_scales, block_table, context_lens, out, N, L, D, num_pages,
                                page_size, max_pages, num_heads, num_kv_heads, scale, is_causal, max_context_hint, TL_BF16, workspace,
                                workspace_bytes, stream);
}
