"""The FP8 (E4M3) KV-page oracle (`oracle/kv_fp8.py`) pinned to third-party arithmetic: PyTorch's
`torch.float8_e4m3fn` casts (OCP FP8, round to nearest even).  The reference has no quantised cache (README.md:134-135), so the
number format is what can be pinned; the cache semantics are stated in the oracle's header."""
import numpy as np
import pytest
import torch

from oracle import kv_fp8
from oracle import tiny_oracle as O


def _torch_encode(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def _torch_decode(codes):
    return torch.from_numpy(np.asarray(codes, np.uint8)).view(torch.float8_e4m3fn).to(torch.float32).numpy()


def test_decode_every_code_equals_torch():
    codes = np.arange(256, dtype=np.uint8)
    mine, ref = kv_fp8.decode_e4m3(codes), _torch_decode(codes)
    assert np.array_equal(np.isnan(mine), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(mine[ok].view(np.uint32), ref[ok].view(np.uint32))
    assert kv_fp8.decode_e4m3(np.uint8(0x7E)) == 448.0 and kv_fp8.decode_e4m3(np.uint8(0x01)) == 2.0 ** -9


def test_encode_equals_torch_over_values_ties_and_subnormals():
    rng = np.random.default_rng(0)
    finite = kv_fp8.decode_e4m3(np.arange(256, dtype=np.uint8))
    finite = np.sort(finite[~np.isnan(finite)])
    mids = (finite[:-1].astype(np.float64) + finite[1:].astype(np.float64)) / 2  # every tie
    cases = np.concatenate([
        finite, mids.astype(np.float32), np.nextafter(mids.astype(np.float32), np.float32(np.inf)),
        np.nextafter(mids.astype(np.float32), np.float32(-np.inf)),
        rng.uniform(-448, 448, 20000).astype(np.float32), (rng.standard_normal(20000) * 0.01).astype(np.float32),
        (rng.standard_normal(20000) * 2.0 ** -8).astype(np.float32), np.float32([0.0, -0.0, 448.0, -448.0, 2.0 ** -10, 2.0 ** -9, 3 * 2.0 ** -10])])
    mine, ref = kv_fp8.encode_e4m3(cases), _torch_encode(cases)
    assert np.array_equal(mine, ref)
    # every finite code is a fixed point
    codes = np.arange(256, dtype=np.uint8)
    codes = codes[(codes & 0x7F) != 0x7F]
    assert np.array_equal(kv_fp8.encode_e4m3(kv_fp8.decode_e4m3(codes)), codes)


def test_row_scale_is_the_smallest_power_of_two_that_fits():
    rng = np.random.default_rng(1)
    amax = np.concatenate([np.float32([448.0, 448.0 * 2, 447.9, 448.1, 1.75, 1.7500001, 1.0, 3.5, 2.0 ** -20, 3e38, 1e-30]),
                           np.exp(rng.uniform(-30, 30, 5000)).astype(np.float32)])
    s = kv_fp8.row_scale(amax)
    m, e = np.frexp(s)
    assert np.all(m == 0.5)  # powers of two
    big = amax > 2.0 ** (kv_fp8.SCALE_EXP_MIN - 127 + 9)
    assert np.all(amax[big].astype(np.float64) / s[big] <= 448.0)
    assert np.all(amax[big].astype(np.float64) / (s[big].astype(np.float64) / 2) > 448.0)


@pytest.mark.parametrize("sigma", [1e-3, 0.05, 1.0, 30.0])
def test_round_trip_rows_are_bf16_values_within_one_e4m3_step(sigma):
    rng = np.random.default_rng(2)
    x = O.bf16(rng.standard_normal((3, 5, 7, 128)).astype(np.float32) * sigma)
    x[0, 0, 0] = 0.0  # an all-zero row
    codes, s = kv_fp8.quantize_rows(x)
    assert codes.dtype == np.uint8 and s.shape == x.shape[:-1] and not np.any((codes & 0x7F) == 0x7F)
    y = kv_fp8.dequantize_rows(codes, s)
    assert np.array_equal(O.bf16(y), y)  # exactly representable in bf16
    amax = np.max(np.abs(x), axis=-1, keepdims=True)
    # relative step of E4M3 is 2^-3 (half of it after rounding); values far below the row maximum fall into the subnormal grid
    assert np.all(np.abs(y - x) <= np.maximum(np.abs(x) * 2.0 ** -4, s[..., None] * 2.0 ** -10) + 0)
    assert np.all(np.max(np.abs(y), axis=-1, keepdims=True) <= amax * (1 + 2.0 ** -4))
    assert np.all(y[0, 0, 0] == 0)
    # idempotent: a dequantised row quantises to itself
    codes2, s2 = kv_fp8.quantize_rows(y)
    assert np.array_equal(kv_fp8.dequantize_rows(codes2, s2), y)


def test_paged_update_and_attention_over_dequantised_pages():
    rng = np.random.default_rng(3)
    P, H, page, D = 6, 2, 16, 128
    kp = np.zeros((P, H, page, D), np.uint8)
    ks = np.zeros((P, H, page), np.float32)
    vals = O.bf16(rng.standard_normal((1, H, 11, D)).astype(np.float32))
    kv_fp8.paged_cache_update(kp, ks, vals, 4, 3)
    deq = kv_fp8.dequantize_pages(kp, ks)
    assert np.array_equal(deq[4, :, 3:14], kv_fp8.round_trip(vals[0]))
    assert not deq[:4].any() and not deq[5].any() and not deq[4, :, :3].any() and not deq[4, :, 14:].any()


def test_oracle_model_with_a_quantised_cache_stays_near_the_bf16_one():
    cfg = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
               intermediate_size=512, vocab_size=512, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True)
    w = O.make_fast_w4_weights(cfg, seed=5)
    toks = np.random.default_rng(4).integers(0, 512, 24)
    a, b = O.OracleQwen3(cfg, w), O.OracleQwen3(cfg, w, kv_format="fp8")
    la = [a.forward(toks[:16])] + [a.forward(toks[16 + i:17 + i]) for i in range(4)]
    lb = [b.forward(toks[:16])] + [b.forward(toks[16 + i:17 + i]) for i in range(4)]
    truth = O.TruthQwen3(cfg, w)
    lt = [truth.forward(toks[:16])] + [truth.forward(toks[16 + i:17 + i]) for i in range(4)]
    ea = max(np.max(np.abs(x - t)) for x, t in zip(la, lt))
    eb = max(np.max(np.abs(x - t)) for x, t in zip(lb, lt))
    assert any(not np.array_equal(x, y) for x, y in zip(la, lb))  # the quantised cache is really in the loop
    assert eb <= 6 * ea + 0.05, (ea, eb)
