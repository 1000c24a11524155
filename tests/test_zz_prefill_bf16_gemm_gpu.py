"""GPU parity of the plain bf16 prefill GEMM (csrc/gemm8.h, round 6) that large prefill chunks run on.

The reference's tile GEMM (quantized_matmul_simdgroup_w4a16_g128: src/extensions_ref/src/quantized_matmul.metal:96-249) rounds every
dequantised weight to bf16 -- T(q * scale + bias) -- before a bf16 MMA with fp32 accumulation over the whole reduction.  The engine forms
that B operand once (tl_prefill_weights_bf16) and multiplies chunks of 1,792 rows and more against it (kernel-level cases from 1,536) with a 256-wide-tile GEMM whose
operands both arrive by LDS-DMA (tl_prefill_matmul_bf16).  Held here

  * the expansion against the numpy oracle's dequantisation: BIT-identical;
  * the product at the Qwen3-4B shapes, 1,536- to 4,096-row chunks (every tile shape of the planner) and ragged row counts, against oracle.quantized_matmul_tile (split_k = 1: weights
    rounded first, one rounding of the fp32 sum) on sampled rows -- one bf16 step of the oracle's value + 2^-20 of the absolute sum for the
    other summation order -- for the store, residual and SwiGLU epilogues;
  * the engine: a 4,096-token prompt prefilled in one chunk through this GEMM against the same prompt through the W4 GEMM (engine option
    "gemm8" = 0) and against the float64 truth with the band every model-level test uses.
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import TINY_CFG, assert_within, bf16_ulp, check_against_truth, log_parity, to_mlx_shaped, ulp_of, w4_abs_dot

pytestmark = pytest.mark.gpu

DEV = "cuda"
SHAPES = {"qkv": (6144, 2560), "wo": (2560, 4096), "gate_up": (19456, 2560), "down": (2560, 9728)}  # weight rows (outputs), columns (reduction)


@pytest.fixture(scope="module")
def ext(built_libs):
    import tiny_llm_ext_hip as e

    return e


def _bf16_host(t):
    return O.from_bf16_bits(t.view(torch.int16).cpu().numpy().view(np.uint16))


_cache = {}


def _matrix(ext, name):
    if name not in _cache:
        from tiny_llm_hip.synthetic import quantize

        rows, cols = SHAPES[name]
        gen = torch.Generator(device=DEV)
        gen.manual_seed(4000 + rows % 89 + cols % 97)
        w = (torch.randn((rows, cols), generator=gen, device=DEV, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        packed, scales, biases = quantize(w)
        wb = ext.prefill_weights_bf16(packed, scales, biases)
        _cache[name] = (packed, scales, biases, wb)
    return _cache[name]


@pytest.mark.parametrize("name", list(SHAPES))
def test_the_bf16_weight_copy_is_the_oracles_dequantisation(ext, name):
    packed, scales, biases, wb = _matrix(ext, name)
    rows = np.linspace(0, SHAPES[name][0] - 1, 97).astype(np.int64)
    want = O.dequantize_weights(packed.cpu().numpy().view(np.uint32)[rows], _bf16_host(scales)[rows], _bf16_host(biases)[rows], 128, 4, "bf16")
    got = _bf16_host(wb[torch.from_numpy(rows).to(DEV)])
    assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} of {got.size} weights differ from bf16(q * s + beta)"


@pytest.mark.parametrize("M", [1536, 2048, 3072, 4096, 4099])
@pytest.mark.parametrize("name", list(SHAPES))
def test_product_against_the_tile_oracle(ext, name, M):
    packed, scales, biases, wb = _matrix(ext, name)
    rows_w, cols = SHAPES[name]
    gen = torch.Generator(device=DEV)
    gen.manual_seed(M + rows_w)
    a = torch.randn((M, cols), generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    epi = {"qkv": 0, "wo": 1, "gate_up": 2, "down": 1}[name]
    residual = torch.randn((M, rows_w), generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16) if epi == 1 else None
    got = ext.prefill_matmul_bf16(a, wb, epilogue=epi, residual=residual)
    plain = ext.prefill_matmul_bf16(a, wb, epilogue=0) if epi != 0 else got
    assert torch.isfinite(got.float()).all()
    # sampled rows (the product is row-independent): first / last rows of the first and last 256-row tile, the ragged tail, a spread between
    pick = sorted(set([0, 1, 127, 128, 255, 256, 257, M - 257, M - 256, M - 2, M - 1] + [int(x) for x in np.linspace(300, M - 300, 21)]))
    idx = torch.tensor(pick, device=DEV)
    hp, hs, hb = packed.cpu().numpy().view(np.uint32), _bf16_host(scales), _bf16_host(biases)
    ah = _bf16_host(a[idx])
    want_plain = O.quantized_matmul_tile(hs, hb, ah, hp, "bf16", split_k=1)
    floor = 2.0 ** -20 * w4_abs_dot(ah, hp, hs, hb, "bf16")
    what = f"gemm8 {name} M={M}"
    assert_within(_bf16_host(plain[idx]), want_plain, ulp_of(want_plain, "bf16") + floor, what=what + " [store]")
    if epi == 1:
        rh = _bf16_host(residual[idx])
        want = O.bf16(rh + want_plain)
        assert_within(_bf16_host(got[idx]), want, ulp_of(want_plain, "bf16") + floor + bf16_ulp(want), what=what + " [residual]")
    elif epi == 2:
        g, u = want_plain[:, 0::2].astype(np.float64), want_plain[:, 1::2].astype(np.float64)
        fg, fu = floor[:, 0::2], floor[:, 1::2]
        want = O.swiglu(want_plain[:, 0::2], want_plain[:, 1::2])
        allowed = 1.1 * (bf16_ulp(g) + fg) * np.abs(u) + (bf16_ulp(u) + fu) * np.abs(g / (1 + np.exp(-g))) + bf16_ulp(want)
        assert_within(_bf16_host(got[idx]), want, allowed, what=what + " [SwiGLU]")
    err = np.abs(_bf16_host(plain[idx]).astype(np.float64) - want_plain)
    log_parity({"what": "gemm8_vs_tile_oracle", "name": name, "M": M, "max_abs_err": float(err.max()), "bit_identical_share": float((err == 0).mean())})


def test_engine_prefill_of_a_4096_token_chunk_against_the_w4_gemm_and_the_truth():
    """One 4,096-row chunk through the bf16 GEMM (max_prefill_rows >= 1,792 makes the engine keep the bf16 copy) against the same chunk through
    the W4 GEMM (option "gemm8" = 0) -- two summation orders of the same rounded operands: within the band of two HIP paths -- and both
    against the float64 truth of a shorter prompt's last row (TINY model: the truth of 4,096 tokens would take minutes on the CPU, so the
    truth check runs at 3,100 tokens, still one gemm8 chunk)."""
    from tiny_llm_hip.engine import DecodeEngine

    weights = O.make_qwen3_weights(TINY_CFG, seed=11, sigma=0.05)
    model = to_mlx_shaped(TINY_CFG, weights)
    rng = np.random.default_rng(5)
    runs = {}
    for n_tok in (4096, 3100):
        prompt = [int(t) for t in rng.integers(1, TINY_CFG["vocab_size"], size=n_tok)]
        for flag in (1, 0):
            eng = DecodeEngine(model, page_size=128, num_pages=n_tok // 128 + 3, max_batch=1, max_prefill_rows=4096, options={"gemm8": flag})
            try:
                eng.begin(0)
                eng.prefill(0, prompt, chunk=4096)
                first = eng.logits(1)[0].float().cpu().numpy()
                eng.decode(2, batch=1)
                runs[(n_tok, flag)] = (first, eng.logits(1)[0].float().cpu().numpy(), eng.read_tokens(0, 3), prompt)
                eng.release(0)
            finally:
                eng.close()
    for n_tok in (4096, 3100):
        a, b = runs[(n_tok, 1)], runs[(n_tok, 0)]
        scale = float(np.abs(b[0]).max())
        assert float(np.abs(a[0] - b[0]).max()) <= 0.05 * max(1.0, scale), f"{n_tok} tokens: prefill logits of the two GEMMs differ by {np.abs(a[0] - b[0]).max()}"
    prompt = runs[(3100, 1)][3]
    ref, truth = O.OracleQwen3(TINY_CFG, weights), O.TruthQwen3(TINY_CFG, weights)
    want, exact = ref.forward(prompt)[0, -1][None], truth.forward(prompt)[0, -1][None]
    rec = check_against_truth(runs[(3100, 1)][0][None], want, exact, what="engine prefill, one 3,100-row chunk through the bf16 GEMM")
    log_parity({"what": "gemm8_engine_prefill_vs_truth", **rec})


@pytest.mark.parametrize("name,M", [("qkv", 4096), ("wo", 2048), ("gate_up", 1536), ("down", 4096), ("gate_up", 6000)])
def test_repeated_launches_under_memory_load_are_bit_identical(ext, name, M):
    """Race screen for the LDS-DMA pipeline (a stage is refilled while the matrix pipe still works on fragments read from it; the only
    ordering is the counted wait + the one barrier per step): the kernel is deterministic, so ANY difference between two launches on the
    same inputs is a stage read before its DMA landed or overwritten before it was read.  60 launches per shape (every tile shape of the
    planner: 256 x 192, 128 x 160, 256 x 256, 256 x 160, and a ragged 6,000 rows), half of them racing a copy kernel on a second stream
    that keeps HBM and the L2s busy, against the first launch -- and the first launch against torch's fp32 product on sampled rows."""
    packed, scales, biases, wb = _matrix(ext, name)
    rows_w, cols = SHAPES[name]
    gen = torch.Generator(device=DEV)
    gen.manual_seed(17 * M + rows_w)
    a = torch.randn((M, cols), generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    first = ext.prefill_matmul_bf16(a, wb, epilogue=0)
    idx = torch.tensor(sorted(set([0, M - 1] + [int(x) for x in np.linspace(1, M - 2, 29)])), device=DEV)
    want = (a[idx].float() @ wb.float().T)
    assert torch.allclose(first[idx].float(), want, rtol=2 ** -7, atol=2e-3), f"{name} M={M}: first launch off by {float((first[idx].float() - want).abs().max())}"
    side = torch.cuda.Stream()
    junk = torch.empty((256 * 1024 * 1024,), dtype=torch.uint8, device=DEV)
    for i in range(60):
        if i % 2:
            with torch.cuda.stream(side):
                junk.copy_(junk.flip(0) if i % 4 == 1 else junk)
        again = ext.prefill_matmul_bf16(a, wb, epilogue=0)
        if not torch.equal(again, first):
            bad = (again != first)
            raise AssertionError(f"{name} M={M}: launch {i} differs from the first in {int(bad.sum())} elements "
                                 f"(first at flat index {int(bad.flatten().nonzero()[0])})")
    torch.cuda.synchronize()
