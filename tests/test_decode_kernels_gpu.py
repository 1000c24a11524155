"""GPU parity of the kernels the decode engine ACTUALLY launches, at the real Qwen3-4B shapes and BASELINE contexts.

The operator tests (test_ops_gpu.py) drive the public reference API; the engine, however, decodes through its own
kernels: `qmv3_kernel` (fused MFMA GEMV over the tiled weights), `qmm3_kernel` + slice reduction (5..64 rows) and
`attn_decode_fused_kernel` (+ merge).  These tests call exactly that launch code through the
kernel-level C entry points (`tl_decode_linear`, `tl_decode_attention_fused`, include/tinyllm_engine.h) and compare with
the numpy oracle on the same seeded inputs:

  * the five Qwen3-4B projections (6144x2560 qkv, 2560x4096 wo, 19456x2560 interleaved gate|up, 2560x9728 w_down,
    151936x2560 lm_head) x rows {1,2,4,8} (GEMV) and {5,9,16,33,64} (skinny matmul) x the fused variant the engine uses for
    that projection (RMSNorm prologue, residual / SwiGLU epilogue) and the plain one, asserting WHICH kernel and which
    template instantiation ran (no silent fallback to the packed-dot GEMV);
  * decode attention at contexts 1 .. 32,768, page 128, Hq32/Hkv8/D128 (the reference's bench_week3_attention.py:74-77
    shape), 1 and 4 sequences, the wide one-head kernel, the split kernel + merge, and the >64-split merge;
  * the MFMA FlashAttention operator at a 2048-row chunk over an 8k cached context (sampled query rows).

Reference tests mirrored: tests_refsol/test_week_2_day_3.py:89-118 (per-shape matvec vs the dequantised product),
test_week_3_day_4.py:118-245 (paged decode), test_week_3_day_5.py:23-61 (paged FlashAttention).
Tolerance (stated per assert): outputs are bf16; the oracle accumulates in float64 and rounds once, the kernels
accumulate in fp32 in another order, so a value may land on the neighbouring bf16 (1 ulp); an op with two roundings
(residual, SwiGLU) may move 2.  `abs_floor` covers outputs that are small against their own partial sums.
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import assert_bf16_close, assert_within, bf16_ulp, log_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS = 1e-6

PRO_NONE, PRO_RMSNORM, PRO_ATTN_MERGE, PRO_RMS_WEIGHTED = 0, 1, 2, 3
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU = 0, 1, 2

# name: (weight rows K, reduction N, prologue, epilogue) -- the fused pair the engine uses for that projection
PROJECTIONS = {
    "qkv": (6144, 2560, PRO_RMSNORM, EPI_STORE),
    "wo": (2560, 4096, PRO_NONE, EPI_RESIDUAL),
    "gate_up": (19456, 2560, PRO_RMSNORM, EPI_SWIGLU),
    "down": (2560, 9728, PRO_NONE, EPI_RESIDUAL),
    "lm_head": (151936, 2560, PRO_RMSNORM, EPI_STORE),
}
# qmv3 template parameters the planner picks at ONE row (MR, KS, CW, LM): what bench.py's roofline kernel is
GEMV_PLAN_M1 = {"qkv": (1, 2, 4, 10), "wo": (1, 4, 4, 8), "gate_up": (1, 4, 4, 5), "down": (1, 8, 8, 10),
                "lm_head": (1, 2, 4, 10)}
# ... and at 2 and 4 rows where they differ from one row by more than MR (round 3, tools/lab/plan_lab: gate|up keeps the finer cut up to
# 4 rows, w_down takes 16 waves of 5 groups at 3-4 rows)
GEMV_PLAN_ROWS = {("gate_up", 2): (2, 4, 4, 5), ("gate_up", 4): (4, 4, 4, 5), ("down", 2): (2, 8, 8, 10), ("down", 4): (4, 16, 16, 5),
                  ("qkv", 2): (2, 2, 4, 10), ("qkv", 4): (4, 2, 4, 10), ("wo", 2): (2, 4, 4, 8), ("wo", 4): (4, 4, 4, 8)}
MAX_ROWS = 64


@pytest.fixture(scope="module")
def ext():
    import tiny_llm_ext_hip

    tiny_llm_ext_hip.load_library(".")
    return tiny_llm_ext_hip


def _bf16_host(t: torch.Tensor) -> np.ndarray:
    return t.float().cpu().numpy()


class _Projection:
    """Seeded W4 matrix of one Qwen3-4B projection (product quantiser on N(0, 0.02) weights, built on the GPU), the
    activation / residual rows, and the oracle's results for every fused variant (all MAX_ROWS rows, computed once)."""

    def __init__(self, ext, name):
        from tiny_llm_hip.synthetic import quantize

        K, N, pro, epi = PROJECTIONS[name]
        self.name, self.K, self.N, self.pro, self.epi = name, K, N, pro, epi
        gen = torch.Generator(device=DEV)
        gen.manual_seed(1000 + len(name) * 7 + K % 97)
        w = (torch.randn((K, N), generator=gen, device=DEV, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        packed, scales, biases = quantize(w)
        del w
        self.a = torch.randn((MAX_ROWS, N), generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
        self.norm_w = (1.0 + 0.05 * torch.randn((N,), generator=gen, device=DEV, dtype=torch.float32)).to(torch.bfloat16)
        self.residual = torch.randn((MAX_ROWS, K), generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
        self.tiled = ext.TiledW4(packed, scales, biases)
        # oracle, float64 accumulation, one rounding per reference op
        hp = packed.cpu().numpy().view(np.uint32)
        hs, hb = _bf16_host(scales), _bf16_host(biases)
        self.packed_host, self.scales_host, self.biases_host = hp, hs, hb
        self._w4_dev = (packed, scales, biases)
        a, nw, res = _bf16_host(self.a), _bf16_host(self.norm_w), _bf16_host(self.residual)
        # one pass over the weights for both activation sets (unpacking 389 M nibbles dominates the lm_head oracle)
        stacked = np.concatenate([a, O.rms_norm_fast(a, nw, EPS)], axis=0) if pro == PRO_RMSNORM else a
        both = O.quantized_matmul(hs, hb, stacked, hp, "bf16")
        plain = both[:MAX_ROWS]
        self.scale = float(np.sqrt(np.mean(plain.astype(np.float64) ** 2)))
        # Allowance per element: the kernels accumulate in fp32 in another order than the oracle's float64, so every bf16
        # ROUNDING of an accumulated value may land on the neighbouring bf16 (1 ulp of THAT value), and what a later op makes
        # of it follows from that op's derivative.  floor = fp32 accumulation noise on sums of rms `scale`.
        floor = 2e-4 * max(1.0, self.scale)
        self.want = {(PRO_NONE, EPI_STORE): plain}
        self.allowed = {(PRO_NONE, EPI_STORE): bf16_ulp(plain) + floor}
        if pro == PRO_RMSNORM:
            normed = both[MAX_ROWS:]
            if epi == EPI_SWIGLU:  # rows interleaved: even = gate_i, odd = up_i (the engine's fused gate|up weight)
                g, u = normed[:, 0::2].astype(np.float64), normed[:, 1::2].astype(np.float64)
                out = O.swiglu(normed[:, 0::2], normed[:, 1::2])
                self.want[(pro, epi)] = out
                # d silu / dg is within [-0.1, 1.1]; the gate and the up value are each rounded once before the product
                self.allowed[(pro, epi)] = (1.1 * (bf16_ulp(g) + floor) * np.abs(u) + (bf16_ulp(u) + floor) * np.abs(g / (1 + np.exp(-g)))
                                            + bf16_ulp(out))
            else:
                self.want[(pro, epi)] = normed
                self.allowed[(pro, epi)] = bf16_ulp(normed) + floor
        else:
            out = O.bf16(res + plain)
            self.want[(pro, epi)] = out
            # bf16(residual + bf16(acc)): one ulp of the projection's value, then one ulp of the sum
            self.allowed[(pro, epi)] = bf16_ulp(plain) + floor + bf16_ulp(out)

    def squared_dot(self, a: np.ndarray) -> np.ndarray:
        """sum_n (a[m, n] W[k, n])^2 in float64 (on the device: W is up to 19,456 x 2,560), W = the dequantised weights."""
        packed, scales, biases = self._w4_dev
        shifts = torch.arange(0, 32, 4, device=DEV, dtype=torch.int32)
        q = ((packed.to(torch.int32).unsqueeze(-1) >> shifts) & 0xF).reshape(self.K, self.N).double()
        w = q * scales.double().repeat_interleave(128, dim=1) + biases.double().repeat_interleave(128, dim=1)
        a2 = torch.from_numpy(np.asarray(a, dtype=np.float64)).to(DEV) ** 2
        return (a2 @ (w * w).T).cpu().numpy()

    def max_abs_weight_per_output(self) -> np.ndarray:
        """max_n |W[k, n]| of the dequantised weights, [K] (on the device)."""
        packed, scales, biases = self._w4_dev
        shifts = torch.arange(0, 32, 4, device=DEV, dtype=torch.int32)
        out = torch.empty((self.K,), dtype=torch.float32, device=DEV)
        step = 16384
        for k0 in range(0, self.K, step):
            q = ((packed[k0:k0 + step].to(torch.int32).unsqueeze(-1) >> shifts) & 0xF).reshape(-1, self.N).float()
            w = q * scales[k0:k0 + step].float().repeat_interleave(128, dim=1) + biases[k0:k0 + step].float().repeat_interleave(128, dim=1)
            out[k0:k0 + step] = w.abs().amax(dim=1)
        return out.cpu().numpy().astype(np.float64)

    def run(self, ext, M, variant, kernel):
        pro, epi = variant
        return ext.decode_linear(self.tiled, self.a[:M].contiguous(), prologue=pro, epilogue=epi,
                                 norm_weight=self.norm_w if pro == PRO_RMSNORM else None,
                                 residual=self.residual[:M].contiguous() if epi == EPI_RESIDUAL else None, eps=EPS,
                                 kernel=kernel)


_cache = {}


@pytest.fixture()
def projection(ext, request):
    name = request.param
    if name not in _cache:  # built once per session: the lm_head oracle alone unpacks 389 M nibbles
        _cache[name] = _Projection(ext, name)
    return _cache[name]


def _variants(p):
    return [(PRO_NONE, EPI_STORE), (p.pro, p.epi)]


def _check(p, got, M, variant, what):
    assert_within(_bf16_host(got), p.want[variant][:M], p.allowed[variant][:M], what=what)


def _check_with_flips(p, got, M, variant, what):
    """For the fused-RMSNorm routes that take 1 / rms from partial sums of squares: the kernel's fp32 1 / rms may differ from the
    oracle's in the last bit, which turns the bf16 rounding of a staged element that lies on a rounding boundary (about one element in
    30,000: now and then ONE per row).  Every output of that row then moves by ulp(a_n) |W_kn| -- visible only where the output itself
    is tiny (seen on the device: lm_head at 33 rows, 441 of 5,013,888 outputs, all in one row, by up to 6e-4).  Two tiers: the plain
    allowance for at least 98 % of every row, and `2 flips of the row's largest staged element against the output's largest weight`
    on top of it for the rest."""
    g, want, allowed = _bf16_host(got).astype(np.float64), p.want[variant][:M].astype(np.float64), p.allowed[variant][:M]
    excess = np.abs(g - want) - allowed
    bad = ~(excess <= 0)
    if not bad.any():
        return
    normed = O.rms_norm_fast(_bf16_host(p.a[:M]), _bf16_host(p.norm_w), EPS).astype(np.float64)
    wmax = p.max_abs_weight_per_output()  # [K]
    flip = 2.0 * bf16_ulp(np.abs(normed).max(axis=1))[:, None] * wmax[None, :]
    if variant[1] == EPI_SWIGLU:  # the shift of a gate / up pre-activation through the product's derivative (d silu / dg within [-0.1, 1.1])
        pre = O.quantized_matmul(p.scales_host, p.biases_host, normed.astype(np.float32), p.packed_host, "bf16").astype(np.float64)
        gpre, upre = pre[:, 0::2], pre[:, 1::2]
        flip = 1.1 * flip[:, 0::2] * np.abs(upre) + flip[:, 1::2] * np.abs(gpre / (1 + np.exp(-gpre)))
    assert (bad.mean(axis=1) <= 0.02).all(), f"{what}: rows with more than 2 % of their outputs outside the plain allowance: {bad.mean(axis=1).max():.4f}"
    assert (excess <= flip).all(), f"{what}: {int((excess > flip).sum())} elements outside even the flipped-element allowance; worst excess {excess.max():.3g}"
    log_parity({"what": "fused_norm_flip_tier_used", "case": what[:120], "elements": int(bad.sum()), "rows": int(bad.any(axis=1).sum()), "worst_excess": float(excess.max())})


@pytest.mark.parametrize("M", [1, 2, 4, 8])
@pytest.mark.parametrize("projection", list(PROJECTIONS), indirect=True)
def test_fused_gemv_at_qwen3_4b_shapes(ext, projection, M):
    """qmv3_kernel, the kernel bench.py's roofline names, on every 4B projection: plain and with the engine's fused
    prologue / epilogue; the MFMA kernel itself must have run, with the planner's documented instantiation at one row."""
    p = projection
    for variant in _variants(p):
        got, info = p.run(ext, M, variant, kernel=1)
        what = f"qmv3 {p.name} M={M} variant={variant} {info}"
        assert info["kernel"] == 1, f"{what}: fell back to {info['kernel_name']}"
        if M == 1:
            assert tuple(info["p"][:4]) == GEMV_PLAN_M1[p.name], what
        if (p.name, M) in GEMV_PLAN_ROWS and info["rows_per_pass"] == M:
            assert tuple(info["p"][:4]) == GEMV_PLAN_ROWS[(p.name, M)], what
        if info["rows_per_pass"] == M:
            assert info["launches"] == 1, what
        _check(p, got, M, variant, what)


# The skinny matmul has two grids (csrc/qmm3.h): kernel 3 = one workgroup per (tile group, slice), kernel 4 = persistent (one
# workgroup per CU, 8-group slices); the planner's choice by shape (qmm3_prefers_persistent, from the r02 lab) for the real
# matrices, by 16-row blocks MB = 1 (<= 16 rows), 2 (<= 32), 4 (<= 64) -- and 3 at 33-48 rows on the persistent grid (round 6):
PERSISTENT_FROM_MB = {"qkv": None, "wo": None, "gate_up": 1, "down": 2, "lm_head": 2}


@pytest.mark.parametrize("M", [5, 9, 16, 17, 32, 33, 48, 49, 64])
@pytest.mark.parametrize("projection", list(PROJECTIONS), indirect=True)
def test_skinny_matmul_at_qwen3_4b_shapes(ext, projection, M):
    """qmm3_kernel / qmm3p_kernel + qmm3_reduce_kernel (batched decode, 5..64 rows): fp32 slice partials summed in slice order,
    the engine's epilogues applied by the reduction; includes the 64-row lm_head whose partials are the largest workspace.
    Both grids are held against the oracle for every projection and row count; the grid the planner picks is asserted."""
    p = projection
    MB = 1 if M <= 16 else (2 if M <= 32 else 4)
    for grid in (3, 4):
        for variant in _variants(p):
            got, info = p.run(ext, M, variant, kernel=grid)
            what = f"qmm3 grid {grid} {p.name} M={M} variant={variant} {info}"
            assert info["kernel"] == 2 and info["p"][0] == (3 if grid == 4 and 32 < M <= 48 else MB), what
            assert (info["p"][1] == 0) == (grid == 4), f"{what}: p[1] = tiles per wave, 0 on the persistent grid"
            if grid == 4:
                assert info["p"][2] in (4, 8) and info["p"][4] <= 256, f"{what}: at most one workgroup per CU"
            _check(p, got, M, variant, what)
    _, info = p.run(ext, M, (p.pro, p.epi), kernel=2)
    want_persistent = PERSISTENT_FROM_MB[p.name] is not None and MB >= PERSISTENT_FROM_MB[p.name] and not (p.name == "lm_head" and MB == 1)
    assert info["kernel"] == 2 and (info["p"][1] == 0) == want_persistent, f"planner's grid for {p.name} at M={M}: {info}"
    if M > 8:  # the engine's own routing sends more than 8 rows here as well
        _, info = p.run(ext, M, (p.pro, p.epi), kernel=0)
        assert info["kernel"] == 2, f"routing at M={M}: {info}"


@pytest.mark.parametrize("M", [5, 16, 33, 64])
@pytest.mark.parametrize("name", ["qkv", "gate_up", "lm_head"])
@pytest.mark.parametrize("partials", [8, 160])
def test_skinny_matmul_normalises_with_the_producers_sums_of_squares(ext, name, M, partials):
    """PRO_RMSNORM of the skinny matmul from `partials` sums of squares per row -- 8 (the embedding kernels and the slice reduction:
    the engine's own hand-over at 5-64 rows) or 160 (hidden / 16: what a GEMV producer leaves) -- against the oracle with the plain
    allowance; the RMSNorm must be fused (2 launches: matmul + slice reduction)."""
    if name not in _cache:
        _cache[name] = _Projection(ext, name)
    p = _cache[name]
    x = p.a[:M].double()
    if partials == 8:
        ss = torch.zeros((M, 8), dtype=torch.float32, device=DEV)
        ss[:, 0] = (x * x).sum(dim=1).float()
    else:
        ss = (x * x).reshape(M, p.N // 16, 16).sum(dim=2).float().contiguous()
    got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_RMSNORM, epilogue=p.epi, norm_weight=p.norm_w,
                                  eps=EPS, kernel=2, ss_in=ss)
    what = f"skinny matmul, fused RMSNorm from {partials} partials, {name} M={M} {info}"
    assert info["kernel"] == 2 and info["launches"] == 2, f"{what}: the RMSNorm must be fused"
    _check_with_flips(p, got, M, (PRO_RMSNORM, p.epi), what)


@pytest.mark.parametrize("projection", list(PROJECTIONS), indirect=True)
def test_engine_routing_between_gemv_and_skinny_matmul(ext, projection):
    """1..4 rows: fused GEMV everywhere.  5 rows and more: the skinny matmul for every projection (csrc/engine.hip
    engine_linear); it agrees with the oracle at 5 and 8 rows."""
    p = projection
    variant = (p.pro, p.epi)
    for M in (1, 3, 4):
        _, info = p.run(ext, M, variant, kernel=0)
        assert info["kernel"] == 1, f"{p.name} M={M}: {info}"
    for M in (5, 8):
        got, info = p.run(ext, M, variant, kernel=0)
        assert info["kernel"] == 2, f"{p.name} M={M}: {info}"
        _check(p, got, M, variant, f"routing {p.name} M={M}")


# ---------------------------------------------------------------------------------------------------------------------
# The instantiations BASELINE configs[1] TIMES (bench.py at a 128-token prompt: 4 attention windows): the wo GEMV that forms its
# input row from the split partials, qmv3_kernel<1,4,4,PRO_ATTN_MERGE,EPI_RESIDUAL,8,NS>, and the gate|up GEMV over rows its
# producer left weighted, qmv3_kernel<1,4,4,PRO_RMS_WEIGHTED,EPI_SWIGLU,5> -- each against the numpy oracle at the real shapes,
# through tl_decode_linear_ex, the way the reference tests its matvec per shape (tests_refsol/test_week_2_day_3.py:89-118).
# ---------------------------------------------------------------------------------------------------------------------
PARTIAL_ROW = 128 + 4  # floats per (head, split): 128 value sums, running max (log2 units), running sum, 2 pad


def _split_partials(rng, heads, n_splits):
    """What attn_decode_fused_kernel leaves per (head, split): (sum_j p_j v_j, max score, sum_j p_j) with p_j = 2^(score_j - max).
    Built from random scores / values of 40 tokens per window; head 3 has an EMPTY last window (max -1e30, sums 0: a window past the
    context), head 5 has no token at all (every window empty: the merged row must be zero there)."""
    ws = np.zeros((heads, n_splits, PARTIAL_ROW), dtype=np.float32)
    for h in range(heads):
        for s2 in range(n_splits):
            empty = (h == 3 and s2 == n_splits - 1) or h == 5
            if empty:
                ws[h, s2, 128] = -1e30
                continue
            scores = rng.standard_normal(40) * 4.0 + rng.standard_normal() * 6.0
            vals = O.bf16(rng.standard_normal((40, 128), dtype=np.float32))
            m = scores.max()
            pj = np.exp2(scores - m)
            ws[h, s2, :128] = (pj[:, None] * vals).sum(axis=0)
            ws[h, s2, 128] = m
            ws[h, s2, 129] = pj.sum()
    return ws


def _merged_row(ws):
    """attn_merge_kernel's definition in float64 (engine_kernels.h; reference: the online-softmax merge of
    paged_attention.metal:214-236): out = sum_s v_s f_s / sum_s l_s f_s, f_s = 2^(m_s - max m); zero where the denominator is zero."""
    w = ws.astype(np.float64)
    m, l, v = w[:, :, 128], w[:, :, 129], w[:, :, :128]
    f = np.exp2(m - m.max(axis=1, keepdims=True))
    den = (l * f).sum(axis=1)
    num = (v * f[:, :, None]).sum(axis=1)
    out = np.where(den[:, None] == 0.0, 0.0, num / np.where(den == 0.0, 1.0, den)[:, None])
    return O.bf16(out.reshape(1, -1).astype(np.float32))


@pytest.mark.parametrize("n_splits", [2, 4, 8])
@pytest.mark.parametrize("projection", ["wo"], indirect=True)
def test_wo_gemv_that_merges_the_attention_windows_against_the_oracle(ext, projection, n_splits):
    """h = x + bf16(merge(partials) @ wo^T), the sums of squares of h per 16-row tile and h * post_attention_layernorm: the launch
    bench.py's single stream runs 36 times per token (PRO_ATTN_MERGE; NS = 4 at the bench's 128..256-token context)."""
    p = projection
    rng = np.random.default_rng(40 + n_splits)
    ws = _split_partials(rng, p.N // 128, n_splits)
    row = _merged_row(ws)
    assert not row[0, 5 * 128:6 * 128].any() and row[0, 3 * 128:4 * 128].any()
    hp = p.packed_host
    plain = O.quantized_matmul(p.scales_host, p.biases_host, row, hp, "bf16")
    res = _bf16_host(p.residual[:1])
    want = O.bf16(res + plain)
    # allowance: one ulp of the projection's value + one of the sum (as for the plain wo variant) + the merged row itself: an element
    # whose float64 quotient lies within fp32 noise of a bf16 rounding boundary may land on the other side in the kernel (fp32
    # division, hardware exp2) -- 16 such flips of the largest |a| ulp against the largest |w| bound it
    flips = 16 * float(bf16_ulp(np.abs(row)).max()) * float(np.abs(O.dequantize_weights(hp[:64], p.scales_host[:64], p.biases_host[:64], dtype="bf16")).max())
    allowed = bf16_ulp(plain) + 2e-4 * max(1.0, float(np.sqrt(np.mean(plain.astype(np.float64) ** 2)))) + bf16_ulp(want) + flips
    norm_out = (1.0 + 0.05 * torch.randn((p.K,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))).to(torch.bfloat16)
    got, info = ext.decode_linear(p.tiled, None, prologue=PRO_ATTN_MERGE, epilogue=EPI_RESIDUAL, residual=p.residual[:1].contiguous(),
                                  eps=EPS, kernel=1, merge_partials=torch.from_numpy(ws).to(DEV), want_ss_out=True, norm_out=norm_out)
    what = f"merging wo GEMV, {n_splits} windows {info['p']}"
    assert info["kernel"] == 1 and tuple(info["p"][:4]) == GEMV_PLAN_M1["wo"] and info["launches"] == 1, what
    assert_within(_bf16_host(got), want, allowed, what=what)
    # the hand-over to the consumer: partial sums of squares of the STORED bf16 values, and the weighted row -- both functions of `got`
    g64 = got.double()
    ss_want = (g64 * g64).reshape(1, p.K // 16, 16).sum(dim=2)
    assert torch.allclose(info["ss_out"].double(), ss_want, rtol=1e-5, atol=1e-9), f"{what}: ss_out is not the sum of squares of the stored outputs"
    assert torch.equal(info["out_w"], (got.float() * norm_out.float()).to(torch.bfloat16)), f"{what}: out_w is not bf16(out * norm_out)"
    log_parity({"what": "wo_gemv_attn_merge", "n_splits": n_splits, "max_abs_err": float(np.abs(_bf16_host(got) - want).max()),
                "max_allowed": float(np.max(allowed)), "p": info["p"]})


def _weighted_rows_case(p, M):
    """Rows as the wo GEMV hands them to gate|up: a = bf16(x * w_norm) and per-16-column sums of squares of x (160 partials)."""
    x = p.a[:M].float()
    a_w = (x * p.norm_w.float()).to(torch.bfloat16).contiguous()
    ss = (x.double() ** 2).reshape(M, p.N // 16, 16).sum(dim=2).float().contiguous()
    return a_w, ss


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("projection", ["gate_up"], indirect=True)
def test_gate_up_gemv_over_weighted_rows_against_the_oracle(ext, projection, M):
    """SwiGLU(bf16(inv * (bf16(x w) @ W^T))) against the reference's order SwiGLU(bf16(bf16(x inv w) @ W^T)) (FastRMSNorm then the
    matvec, week2_kernels.metal:41-47 + quantized_matmul.metal:441-538).  The staged element is rounded once on both sides -- of
    x w here, of x inv w there -- so the two pre-activations differ by a sum of N independent rounding differences:
    sigma^2 = 2 (u^2 / 12) sum_n (a_n W_kn)^2 with u <= 2^-7 relative; 6 sigma is allowed on top of the plain variant's allowance."""
    p = projection
    a_w, ss = _weighted_rows_case(p, M)
    got, info = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_SWIGLU, eps=EPS, kernel=1, ss_in=ss)
    what = f"weighted-row gate|up GEMV M={M} {info['p']}"
    assert info["kernel"] == 1 and info["launches"] == 1 and tuple(info["p"][:4]) == (M, 4, 4, 5), what
    normed = O.rms_norm_fast(_bf16_host(p.a[:M]), _bf16_host(p.norm_w), EPS)
    pre = O.quantized_matmul(p.scales_host, p.biases_host, normed, p.packed_host, "bf16")
    sq = p.squared_dot(normed)  # sum_n (a_n W_kn)^2, [M, K]
    sigma = np.sqrt(2.0 / 12.0) * 2.0 ** -7 * np.sqrt(sq)
    g, u = pre[:, 0::2].astype(np.float64), pre[:, 1::2].astype(np.float64)
    dg, du = 6.0 * sigma[:, 0::2], 6.0 * sigma[:, 1::2]
    want = O.swiglu(pre[:, 0::2], pre[:, 1::2])
    floor = 2e-4 * max(1.0, p.scale)
    allowed = 1.1 * (bf16_ulp(g) + floor + dg) * np.abs(u) + (bf16_ulp(u) + floor + du) * np.abs(g / (1 + np.exp(-g))) + bf16_ulp(want)
    assert_within(_bf16_host(got), want, allowed, what=what)
    # and it is as close to the UNROUNDED product (activations not rounded at all) as the reference's order is: both carry one rounding
    err = np.abs(_bf16_host(got) - want)
    log_parity({"what": "gate_up_gemv_weighted_rows", "M": M, "max_abs_err": float(err.max()), "max_allowed": float(allowed.max()),
                "share_of_allowance_max": float((err / allowed).max()), "p": info["p"]})


@pytest.mark.parametrize("name,M", [("qkv", 1), ("qkv", 4), ("lm_head", 1), ("gate_up", 2)])
def test_rmsnorm_gemv_with_the_producers_sums_of_squares(ext, name, M):
    """PRO_RMSNORM with ss_in (the engine's default for qkv and lm_head: w_down's epilogue leaves 160 partials per row): the same
    values as the self-derived norm, held against the oracle with the plain allowance."""
    if name not in _cache:
        _cache[name] = _Projection(ext, name)
    p = _cache[name]
    _, ss = _weighted_rows_case(p, M)
    got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_RMSNORM, epilogue=p.epi, norm_weight=p.norm_w, eps=EPS,
                                  kernel=1, ss_in=ss)
    what = f"RMSNorm GEMV with producer partials {name} M={M} {info['p']}"
    assert info["kernel"] == 1 and info["launches"] == 1, what
    _check_with_flips(p, got, M, (PRO_RMSNORM, p.epi), what)


@pytest.mark.parametrize("name,M", [("wo", 1), ("wo", 4), ("down", 1), ("down", 2), ("down", 4)])
def test_residual_gemv_leaves_sums_of_squares_and_the_weighted_row(ext, name, M):
    """EPI_RESIDUAL with ss_out (+ out_w for wo): the producer side of the hand-over, for the plain (non-merging) GEMV."""
    if name not in _cache:
        _cache[name] = _Projection(ext, name)
    p = _cache[name]
    norm_out = (1.0 + 0.05 * torch.randn((p.K,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))).to(torch.bfloat16)
    got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_NONE, epilogue=EPI_RESIDUAL,
                                  residual=p.residual[:M].contiguous(), eps=EPS, kernel=1, want_ss_out=True,
                                  norm_out=norm_out if name == "wo" else None)
    what = f"residual GEMV with hand-over {name} M={M} {info['p']}"
    assert info["kernel"] == 1 and info["launches"] == 1, what
    _check(p, got, M, (PRO_NONE, EPI_RESIDUAL), what)
    g64 = got.double()
    assert torch.allclose(info["ss_out"].double(), (g64 * g64).reshape(M, p.K // 16, 16).sum(dim=2), rtol=1e-5, atol=1e-9), what
    if name == "wo":
        assert torch.equal(info["out_w"], (got.float() * norm_out.float()).to(torch.bfloat16)), what


# ---------------------------------------------------------------------------------------------------------------------
# decode attention of the engine (q/k-norm + RoPE + KV append + paged GQA attention) at BASELINE contexts
# ---------------------------------------------------------------------------------------------------------------------
HQ, HKV, D, PAGE = 32, 8, 128, 128
THETA = 1e6


def _attention_case(rng, ctxs, extra_pages=2):
    """Scattered physical pages (reference construction test_week_3_day_5.py:25-37), random bf16 K/V, one qkv row per
    sequence; ctx = tokens already cached (an idle slot has ctx 0 and an all -1 block-table row)."""
    B = len(ctxs)
    need = [(c + 1 + PAGE - 1) // PAGE if c >= 0 else 0 for c in ctxs]
    P = sum(need) + extra_pages
    ids = list(rng.permutation(P))
    max_pages = max(max(need), 1) + 1
    table = -np.ones((B, max_pages), dtype=np.int32)
    for b in range(B):
        for j in range(need[b]):
            table[b, j] = ids.pop()
    kp = O.bf16(rng.standard_normal((P, HKV, PAGE, D), dtype=np.float32))
    vp = O.bf16(rng.standard_normal((P, HKV, PAGE, D), dtype=np.float32))
    qkv = O.bf16(rng.standard_normal((B, (HQ + 2 * HKV) * D), dtype=np.float32))
    qn = O.bf16(1.0 + 0.1 * rng.standard_normal((D,), dtype=np.float32))
    kn = O.bf16(1.0 + 0.1 * rng.standard_normal((D,), dtype=np.float32))
    return kp, vp, table, np.asarray([max(c, 0) for c in ctxs], dtype=np.int32), qkv, qn, kn


def _attention_oracle(kp, vp, table, ctx, qkv, qn, kn, idle):
    """Reference op order (qwen3_week3.py:63-86): q/k RMSNorm -> RoPE at offset ctx -> paged_cache_update -> paged_attention
    over ctx + 1 tokens."""
    B = len(ctx)
    kp, vp = kp.copy(), vp.copy()
    rows = qkv.reshape(B, HQ + 2 * HKV, D)
    q = O.rms_norm_fast(rows[:, :HQ], qn, EPS)[:, None]             # [B, L=1, Hq, D]
    k = O.rms_norm_fast(rows[:, HQ:HQ + HKV], kn, EPS)[:, None]
    v = rows[:, HQ + HKV:][:, None]
    q = O.rope(q, ctx, D, THETA, False, "bf16")
    k = O.rope(k, ctx, D, THETA, False, "bf16")
    lens = ctx.copy()
    for b in range(B):
        if idle[b]:
            lens[b] = 0
            continue
        pid, slot = int(table[b, ctx[b] // PAGE]), int(ctx[b] % PAGE)
        kp[pid, :, slot, :] = k[b, 0]
        vp[pid, :, slot, :] = v[b, 0]
        lens[b] = ctx[b] + 1
    out = O.paged_attention(q.transpose(0, 2, 1, 3).reshape(B * HQ, 1, D), kp, vp, table, lens, D ** -0.5, True, HKV, HQ)
    return out.reshape(B, HQ * D), kp, vp


def _run_attention(ext, case, max_context):
    kp, vp, table, ctx, qkv, qn, kn = case
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.bfloat16)
    kpd, vpd = t(kp), t(vp)
    out, info = ext.decode_attention_fused(t(qkv), t(qn), t(kn), kpd, vpd, torch.from_numpy(table).to(DEV),
                                           torch.from_numpy(ctx).to(DEV), num_heads=HQ, num_kv_heads=HKV, rope_theta=THETA,
                                           eps=EPS, max_context=max_context)
    torch.cuda.synchronize()
    return _bf16_host(out), _bf16_host(kpd), _bf16_host(vpd), info


def _check_attention(case, got, kp_after, vp_after, idle, what):
    want, kp_want, vp_want = _attention_oracle(*case, idle)
    # outputs are softmax-weighted means of N(0,1) values (|out| ~ 0.05 .. 1): 1 bf16 ulp + fp32 / exp2 accumulation noise
    assert_bf16_close(got, want, ulps=1.0, abs_floor=1.5e-3, what=what)
    for b, is_idle in enumerate(idle):
        if is_idle:
            assert not got[b].any(), f"{what}: idle slot {b} must produce zeros"
    np.testing.assert_array_equal(vp_after, vp_want, err_msg=f"{what}: value pages (appended V row, nothing else touched)")
    kp_before, table, ctx = case[0], case[2], case[3]
    touched = np.zeros(kp_before.shape[:3], dtype=bool)  # [P, Hkv, slot]
    for b, is_idle in enumerate(idle):
        if not is_idle:
            touched[int(table[b, ctx[b] // PAGE]), :, int(ctx[b] % PAGE)] = True
    np.testing.assert_array_equal(kp_after[~touched], kp_before[~touched], err_msg=f"{what}: key pages outside the appended rows")
    # the appended K row went through RMSNorm and RoPE in fp32 on both sides (two bf16 roundings): a one-ulp difference of a
    # normalised value moves the rotated pair by up to 2^-8 of the LARGER partner, whatever the size of the result
    assert_bf16_close(kp_after[touched], kp_want[touched], ulps=1.0, abs_floor=2.0 ** -6, what=f"{what}: appended K row (norm + RoPE)")


@pytest.mark.parametrize("ctx", [0, 1, 63, 64, 127, 128, 255, 256, 300, 511, 512, 1000, 3000, 4095])
def test_decode_attention_contexts_up_to_4k(ext, ctx, monkeypatch):
    """One sequence, the plan the engine picks: one query head per workgroup and windows of a quarter of the context's power-of-two
    bucket, between 64 and 256 tokens (round 3: 64-token windows up to 256 tokens of context, 8 windows at 2k, 16 at 4k), partials
    merged by a second launch at this kernel-level entry point (inside the engine the wo GEMV merges 2 / 4 / 8 windows itself)."""
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MIN_TOKENS"):
        monkeypatch.delenv(name, raising=False)
    rng = np.random.default_rng(1000 + ctx)
    case = _attention_case(rng, [ctx])
    got, kpa, vpa, info = _run_attention(ext, case, ctx)
    what = f"ctx={ctx} {info}"
    assert info["heads_per_workgroup"] == 1, what
    assert info["n_splits"] * info["tokens_per_split"] >= ctx + 1, what
    bucket = 64
    while bucket < ctx + 1:
        bucket *= 2
    window = max(64, min(256, bucket // 4))
    assert info["n_splits"] == max(1, bucket // window) and info["tokens_per_split"] <= window, what
    assert info["launches"] == (1 if info["n_splits"] == 1 else 2), what
    _check_attention(case, got, kpa, vpa, [False], what)
    log_parity({"what": "decode_attention", "ctx": ctx, "mode": "default", **info})


@pytest.mark.parametrize("ctxs", [[8191], [8192, 5000, 129, -1], [32767], [32768, 1, 700, 20000]])
@pytest.mark.parametrize("mode", ["default", "splits256", "legacy_rq1", "valu_walk"])
def test_decode_attention_long_contexts(ext, ctxs, mode, monkeypatch):
    """BASELINE configs 3 and 5 (8k and 32k cached tokens, page 128), 1 and 4 sequences (ragged, one idle slot written as
    -1): the default plan (one workgroup per GQA group walking its window on the matrix cores, csrc/attn_mfma.h, split + merge), 256
    context splits (128-token windows; attn_merge_cols_kernel over many groups), the one-head split kernel, and the GQA-group walk on
    the VALU (TL_ATTN_MFMA=0: attn_decode_fused_kernel in 64-token stages, the route of rounds 2-3 and still the one for 64-token
    windows)."""
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MFMA"):
        monkeypatch.delenv(name, raising=False)
    if mode == "splits256":
        monkeypatch.setenv("TL_ATTN_MAX_SPLITS", "256")
    elif mode == "legacy_rq1":
        monkeypatch.setenv("TL_ATTN_RQ", "1")
    elif mode == "valu_walk":
        monkeypatch.setenv("TL_ATTN_MFMA", "0")
    idle = [c < 0 for c in ctxs]
    rng = np.random.default_rng(abs(sum(ctxs)) + len(mode))
    case = _attention_case(rng, ctxs)
    got, kpa, vpa, info = _run_attention(ext, case, max(ctxs))
    what = f"ctxs={ctxs} mode={mode} {info}"
    if mode == "splits256" and max(ctxs) >= 32767 and len(ctxs) == 1:
        assert info["n_splits"] > 64, f"{what}: expected the many-split merge"
    _check_attention(case, got, kpa, vpa, idle, what)
    log_parity({"what": "decode_attention_long", "ctxs": ctxs, "mode": mode, **info})


@pytest.mark.parametrize("ctxs,windows", [
    ([150, 129, 64, 1, 255], 1),                                         # 5+ sequences up to 256 tokens: one window each, no merge launch
    ([300, 511, 257, 2, 130, 400, 64, 333, 500], 2),                     # 9 sequences at 257..512: windows of 128+ tokens on at most 256 workgroups
    ([2000, 1030, 5, 1500, 1999, 700], 4),                               # 5-7 sequences beyond 1,024 tokens: 4 windows
    ([1500, 1100, 3000, 700, 64, 129, 2047, 2048, 1, 900] * 2, 1),       # 20 sequences up to 8,192 tokens: whole contexts, one workgroup per (sequence, KV head)
    ([8000] + [200, 1300, 77, 4100, 640, 31, 2500] * 3 + [129, -1], 1),  # 24 slots, one at 8,000 tokens, one idle
])
def test_decode_attention_many_sequences_follow_the_remeasured_plan(ext, ctxs, windows, monkeypatch):
    """The batched plans of the end of round 6 (engine.hip plan_splits; profiles/r06_labs/README.md section 9) at the kernel-level entry point, ragged
    contexts, against the oracle: longer windows than the round-4 plan walked (up to a whole 8,000-token context in one workgroup)."""
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MIN_TOKENS", "TL_ATTN_MFMA"):
        monkeypatch.delenv(name, raising=False)
    idle = [c < 0 for c in ctxs]
    rng = np.random.default_rng(abs(sum(ctxs)) + len(ctxs))
    case = _attention_case(rng, ctxs)
    got, kpa, vpa, info = _run_attention(ext, case, max(ctxs))
    what = f"ctxs={ctxs} {info}"
    assert info["n_splits"] == windows and info["heads_per_workgroup"] == 4, what
    assert info["launches"] == (1 if windows == 1 else 2), what
    _check_attention(case, got, kpa, vpa, idle, what)
    log_parity({"what": "decode_attention_many", "ctxs": ctxs, **info})


def test_flash_attention_2048_row_chunk_over_8k_context(ext):
    """The paged MFMA FlashAttention operator as chunked prefill uses it at BASELINE config 3: the last 2048-row chunk of an
    8,192-token prompt (context 8,192 incl. the chunk), 32/8 heads, page 128.  Sampled query rows (first / last rows,
    32- and 64-row tile edges, random interior rows), every head, against the oracle with P rounded to bf16 before PV
    (paged_attention.metal:439-444)."""
    rng = np.random.default_rng(2048)
    L, ctx = 2048, 8192
    need = ctx // PAGE
    P = need + 3
    table = np.asarray([rng.permutation(P)[:need]], dtype=np.int32)
    kp = O.bf16(rng.standard_normal((P, HKV, PAGE, D), dtype=np.float32))
    vp = O.bf16(rng.standard_normal((P, HKV, PAGE, D), dtype=np.float32))
    q = O.bf16(rng.standard_normal((HQ, L, D), dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.bfloat16)
    got = ext.paged_attention(t(q), t(kp), t(vp), torch.from_numpy(table).to(DEV),
                              torch.tensor([ctx], dtype=torch.int32, device=DEV), D ** -0.5, True, num_kv_heads=HKV,
                              num_heads=HQ, max_context_hint=ctx)
    got = _bf16_host(got)
    rows = sorted({0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 1023, 1024, 2046, 2047, *rng.integers(0, L, size=10).tolist()})
    for r in rows:
        vis = ctx - L + r + 1  # causal visibility of chunk row r (paged_attention.metal:158-160)
        want = O.paged_attention(q[:, r:r + 1], kp, vp, table, np.asarray([vis], dtype=np.int32), D ** -0.5, True, HKV, HQ,
                                 "bf16", round_p=True)
        assert_bf16_close(got[:, r], want[:, 0], ulps=1.0, abs_floor=1.5e-3, what=f"FA row {r} (sees {vis} tokens)")


@pytest.mark.parametrize("ctxs,mode", [([100], "default"), ([700], "default"), ([8191], "default"), ([8192, 5000, 129, -1], "default"),
                                       ([8191], "valu_walk"), ([300, 17, 2000], "valu_walk"), ([300, 17, 2000], "default")])
def test_decode_attention_ignores_what_pages_hold_behind_the_context(ext, ctxs, mode, monkeypatch):
    """A masked token has weight 0 -- and 0 x NaN is NaN: rows behind a sequence's context (the rest of its last page, pages it does not
    own, a recycled page) must not reach the output whatever they hold.  Every K/V row no sequence can see is NaN here (the row the kernel
    appends included: it is written, not read); outputs and appended rows must equal the run on clean pages bit for bit
    (reference: paged_attention.metal:158-160 reads visible rows only)."""
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MFMA"):
        monkeypatch.delenv(name, raising=False)
    if mode == "valu_walk":
        monkeypatch.setenv("TL_ATTN_MFMA", "0")
    idle = [c < 0 for c in ctxs]
    rng = np.random.default_rng(77 + abs(sum(ctxs)))
    case = _attention_case(rng, ctxs)
    kp, vp, table, ctx = case[0], case[1], case[2], case[3]
    visible = np.zeros(kp.shape[:1] + kp.shape[2:3], dtype=bool)  # [P, slot]
    for b, is_idle in enumerate(idle):
        if not is_idle:
            for tok in range(int(ctx[b])):
                visible[int(table[b, tok // PAGE]), tok % PAGE] = True
    kp_nan, vp_nan = kp.copy(), vp.copy()
    kp_nan[~visible[:, None, :].repeat(HKV, axis=1)] = np.nan
    vp_nan[~visible[:, None, :].repeat(HKV, axis=1)] = np.nan
    want, kpa_w, vpa_w, info = _run_attention(ext, case, max(ctxs))
    got, kpa, vpa, _ = _run_attention(ext, (kp_nan, vp_nan) + tuple(case[2:]), max(ctxs))
    what = f"ctxs={ctxs} mode={mode} {info}"
    assert np.isfinite(got).all(), f"{what}: a value behind the context reached the output"
    np.testing.assert_array_equal(got, want, err_msg=what)
    for b, is_idle in enumerate(idle):
        if not is_idle:
            pid, slot = int(table[b, ctx[b] // PAGE]), int(ctx[b] % PAGE)
            np.testing.assert_array_equal(kpa[pid, :, slot], kpa_w[pid, :, slot], err_msg=f"{what}: appended K row")
            np.testing.assert_array_equal(vpa[pid, :, slot], vpa_w[pid, :, slot], err_msg=f"{what}: appended V row")
