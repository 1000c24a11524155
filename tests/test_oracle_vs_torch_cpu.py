"""CPU tier: the oracle's ARITHMETIC against live PyTorch -- an implementation the oracle shares no code with.

tests/golden/torch_vectors.npz pins the fp32 operators; this file pins what the round-1 review found unpinned: the bf16 / f16
rounding emulation itself (bit for bit against torch's casts), the W4 matmul definition, the tile-GEMM rounding order
(weights rounded to T before the products), the split-K partial rounding, P rounded to bf16 before P.V in the FlashAttention
branch, the 16-bit pointwise paths, and the two restatements of mx.quantize (numpy here, torch in the product's synthetic
checkpoints / the facade).  Where both sides accumulate in different orders the statement is "the oracle's value is one of
the two T-neighbours of the float64 result" (checked through torch.nextafter); where nothing but one rounding is involved the
statement is bit equality.  Still not MLX: the reference's own outputs cannot be produced here.
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O

T = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def t64(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


def rounds_to_a_neighbour(got, exact64, dtype):
    """got (float32 container of T values) is the T value just below or just above exact64 (or equal to its rounding)."""
    got = torch.from_numpy(np.ascontiguousarray(got)).to(torch.float64)
    exact = t64(exact64)
    near = exact.to(T[dtype])
    lo = torch.minimum(near, torch.nextafter(near, torch.full_like(near, -float("inf")))).to(torch.float64)
    hi = torch.maximum(near, torch.nextafter(near, torch.full_like(near, float("inf")))).to(torch.float64)
    inside = (got >= lo) & (got <= hi)
    assert bool(inside.all()), f"{int((~inside).sum())} values are not a {dtype} neighbour of the float64 result"
    return float((got == near.to(torch.float64)).double().mean())


def test_bf16_and_f16_rounding_emulation_is_torch_bit_for_bit():
    rng = np.random.default_rng(0)
    vals = [rng.standard_normal(200_000).astype(np.float32) * s for s in (1e-30, 1e-3, 1.0, 77.0, 3e30)]
    # exact ties (bit 15 set, bits 0..14 clear) with even and odd kept mantissas, +-0, the largest finite values, subnormals
    base = rng.integers(0, 2**16, size=50_000).astype(np.uint32) << 16
    ties = (base | 0x8000).view(np.float32)
    ties = ties[np.isfinite(ties)]
    edge = np.array([0.0, -0.0, 3.3895314e38, -3.3895314e38, 3.4028235e38, 1e-45, -1e-45, 1.1754944e-38, 65504.0, 65520.0, 65519.9],
                    dtype=np.float32)
    x = np.concatenate(vals + [ties, np.nextafter(ties, np.float32(np.inf)), np.nextafter(ties, np.float32(-np.inf)), edge])
    with np.errstate(over="ignore"):
        want_bf = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
        want_h = torch.from_numpy(x).to(torch.float16).to(torch.float32).numpy()
        got_bf, got_h = O.bf16(x), O.f16(x)
    np.testing.assert_array_equal(got_bf.view(np.uint32), want_bf.view(np.uint32))
    np.testing.assert_array_equal(got_h.view(np.uint32), want_h.view(np.uint32))
    assert np.isnan(O.bf16(np.array([np.nan], np.float32))[0])
    bits = O.bf16_bits(x)
    np.testing.assert_array_equal(O.from_bf16_bits(bits).view(np.uint32), want_bf.view(np.uint32))
    np.testing.assert_array_equal(bits, torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_two_restatements_of_mx_quantize_agree_bit_for_bit(dtype):
    """oracle.quantize_affine (numpy) and tiny_llm_hip.synthetic.quantize (torch; also what the facade's mx.quantize calls)
    were written separately from the same description: packed words, scales and biases are identical,
    on ordinary groups and on the degenerate ones (all-equal group, all-zero group, one outlier, negative-dominant)."""
    from tiny_llm_hip.synthetic import quantize as torch_quantize

    rng = np.random.default_rng(1)
    w = rng.standard_normal((24, 512)).astype(np.float32) * 0.07
    w[0, :128] = 0.25
    w[1, 128:256] = 0.0
    w[2, 256:384] = 0.0
    w[2, 300] = 3.0
    w[3] = -np.abs(w[3]) - 0.5
    w[4, :128] = np.linspace(-1, 2, 128)
    w = O.cast(w, dtype)
    packed, scales, biases = O.quantize_affine(w, dtype=dtype)
    tp, ts, tb = torch_quantize(torch.from_numpy(w).to(T[dtype]))
    np.testing.assert_array_equal(np.asarray(packed, np.uint32), tp.numpy().view(np.uint32))
    np.testing.assert_array_equal(np.asarray(scales, np.float32), ts.float().numpy())
    np.testing.assert_array_equal(np.asarray(biases, np.float32), tb.float().numpy())
    # round trip: inside the grid the nearest code point is taken (error <= |scale| / 2); the end of the range OPPOSITE the
    # bias edge may be clipped by up to one step, because the scale is re-derived so that the bias edge is an exact multiple
    # of it (edge / round(edge / scale)) and 15 steps of the new scale can fall short of the far end
    deq = O.dequantize_weights(packed, scales, biases, dtype="f32")
    step = np.repeat(np.abs(scales), 128, axis=-1)
    err = np.abs(deq - w)
    assert np.all(err <= 1.07 * step + 1e-6)
    codes = O.unpack_codes(packed)
    inside = (codes > 0) & (codes < 15)
    assert np.all(err[inside] <= 0.57 * step[inside] + 1e-6)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_dequantisation_and_matmul_definition_against_torch(dtype):
    rng = np.random.default_rng(2)
    w = O.cast(rng.standard_normal((96, 384)).astype(np.float32) * 0.05, dtype)
    packed, scales, biases = O.quantize_affine(w, dtype=dtype)
    x = O.cast(rng.standard_normal((5, 384)).astype(np.float32), dtype)
    # torch: codes by shifts on int64, dequantisation in float64
    words = torch.from_numpy(np.asarray(packed, np.uint32).astype(np.int64))
    codes = torch.stack([(words >> (4 * i)) & 0xF for i in range(8)], dim=-1).reshape(96, 384).to(torch.float64)
    s64 = t64(scales).repeat_interleave(128, dim=-1)
    b64 = t64(biases).repeat_interleave(128, dim=-1)
    dense64 = codes * s64 + b64
    # (a) readable dequantisation: ONE rounding of the fp32 q*s+b -> bit equality with torch's own fp32 arithmetic + cast
    want = (codes.float() * s64.float() + b64.float()).to(T[dtype]).float().numpy()
    np.testing.assert_array_equal(O.dequantize_weights(packed, scales, biases, dtype=dtype), want)
    # (b) matvec / matmul definition: unrounded weights, one cast of the sum
    exact = t64(x) @ dense64.T
    share = rounds_to_a_neighbour(O.quantized_matmul(scales, biases, x, packed, dtype), exact.numpy(), dtype)
    assert share > 0.97  # float64 on both sides: they differ only where the float32 staging of the sum straddles a tie
    np.testing.assert_allclose(O.quantized_matmul(scales, biases, x, packed, dtype, raw=True), exact.numpy(), rtol=1e-12, atol=1e-12)
    # (c) tile GEMM: every weight rounded to T BEFORE the products (quantized_matmul.metal:186-191)
    dense_t = torch.from_numpy(want).to(torch.float64)
    exact_tile = t64(x) @ dense_t.T
    assert rounds_to_a_neighbour(O.quantized_matmul_tile(scales, biases, x, packed, dtype), exact_tile.numpy(), dtype) > 0.97
    assert float((exact_tile - exact).abs().max()) > 0  # the two definitions really differ on this input
    # (d) split-K: partial sums of the K slices are STORED in T, then added (metal:251-293)
    for split in (2, 3):
        parts = [(t64(x)[:, i * 384 // split:(i + 1) * 384 // split] @ dense_t[:, i * 384 // split:(i + 1) * 384 // split].T)
                 .float().to(T[dtype]).to(torch.float64) for i in range(split)]
        got = O.quantized_matmul_tile(scales, biases, x, packed, dtype, split_k=split)
        assert rounds_to_a_neighbour(got, sum(parts).numpy(), dtype) > 0.97
    # (e) embedding rows are the dequantised rows
    idx = np.array([[3, 95], [0, 3]])
    np.testing.assert_array_equal(O.quantized_embedding(idx, scales, biases, packed, dtype), want[idx.reshape(-1)].reshape(2, 2, 384))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_sixteen_bit_pointwise_paths_against_torch(dtype):
    rng = np.random.default_rng(3)
    x = O.cast(rng.standard_normal((7, 256)).astype(np.float32) * 2.0, dtype)
    w = O.cast(1.0 + 0.1 * rng.standard_normal(256).astype(np.float32), dtype)
    x64, w64 = t64(x), t64(w)
    # fused RMSNorm: one rounding of x * rsqrt(mean x^2 + eps) * w
    exact = x64 * torch.rsqrt((x64 ** 2).mean(dim=-1, keepdim=True) + 1e-6) * w64
    assert rounds_to_a_neighbour(O.rms_norm_fast(x, w, 1e-6, dtype), exact.numpy(), dtype) > 0.95
    # readable RMSNorm: normalise, round, multiply by the rounded weight, round (layer_norm.py:10-15)
    n = (x64 * torch.rsqrt((x64 ** 2).mean(dim=-1, keepdim=True) + 1e-6)).float().to(T[dtype]).to(torch.float64)
    assert rounds_to_a_neighbour(O.rms_norm_readable(x, w, 1e-6, dtype), (n * w64).numpy(), dtype) > 0.9
    # SwiGLU: fp32 silu(g) * u, one rounding
    g = O.cast(rng.standard_normal((5, 96)).astype(np.float32) * 3.0, dtype)
    u = O.cast(rng.standard_normal((5, 96)).astype(np.float32), dtype)
    exact = torch.nn.functional.silu(t64(g)) * t64(u)
    assert rounds_to_a_neighbour(O.swiglu(g, u, dtype), exact.numpy(), dtype) > 0.97
    # RoPE on [B, L, H, D], both pairings, per-row offsets, partial dims: rotate in float64, one rounding
    q = O.cast(rng.standard_normal((2, 3, 2, 64)).astype(np.float32), dtype)
    offsets = np.array([0, 37])
    for traditional, dims in ((False, 64), (True, 64), (False, 32)):
        half = dims // 2
        pos = t64(offsets)[:, None] + torch.arange(3, dtype=torch.float64)[None, :]
        inv = torch.pow(torch.tensor(10000.0, dtype=torch.float64), -torch.arange(half, dtype=torch.float64) / half)
        ang = pos[:, :, None, None] * inv
        c, s = torch.cos(ang), torch.sin(ang)
        q64 = t64(q)
        out = q64.clone()
        if traditional:
            re, im = q64[..., 0:dims:2], q64[..., 1:dims:2]
            out[..., 0:dims:2], out[..., 1:dims:2] = re * c - im * s, im * c + re * s
        else:
            re, im = q64[..., :half], q64[..., half:dims]
            out[..., :half], out[..., half:dims] = re * c - im * s, im * c + re * s
        got = O.rope(q, offsets, dims, 10000.0, traditional, dtype)
        # the kernels (and the oracle) take the angle in fp32: the rotated value may sit a few fp32 ulps of the angle away
        near = out.float().to(T[dtype]).float().numpy()
        ulp = np.abs(near) * 2.0 ** (-7 if dtype == "bf16" else -10) + 1e-6
        assert np.all(np.abs(got - near) <= ulp) and float(np.mean(got == near)) > 0.9
        np.testing.assert_array_equal(got[..., dims:], q[..., dims:])  # tail dims are copied


@pytest.mark.parametrize("round_p", [False, True])
def test_paged_attention_and_the_rounding_of_p_against_torch(round_p):
    """paged_attention over scattered pages == dense softmax(q k^T) v on the gathered rows (float64 in torch); with round_p
    (the MFMA FlashAttention branch, paged_attention.metal:439-444) the UNNORMALISED probabilities exp(s - max) are rounded to
    bf16 before P.V while the row sum stays fp32."""
    rng = np.random.default_rng(4)
    Hq, Hkv, D, page, P, L = 4, 2, 64, 16, 7, (12 if round_p else 2)
    kp = O.bf16(rng.standard_normal((P, Hkv, page, D)).astype(np.float32))
    vp = O.bf16(rng.standard_normal((P, Hkv, page, D)).astype(np.float32))
    q = O.bf16(rng.standard_normal((Hq, L, D)).astype(np.float32))
    table = np.array([[5, 1, 3, -1]], dtype=np.int32)
    ctx = np.array([41], dtype=np.int32)
    scale = D ** -0.5
    got = O.paged_attention(q, kp, vp, table, ctx, scale, True, Hkv, Hq, "bf16", round_p=round_p)
    k = torch.cat([t64(kp[p]) for p in (5, 1, 3)], dim=1)[:, :41]  # [Hkv, S, D]
    v = torch.cat([t64(vp[p]) for p in (5, 1, 3)], dim=1)[:, :41]
    out = []
    for h in range(Hq):
        s = (t64(q[h]) @ k[h // 2].T) * scale  # [L, S]
        keep = torch.ones((L, 41), dtype=torch.bool).tril(diagonal=41 - L)
        s = s.masked_fill(~keep, float("-inf"))
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        p = e.float().to(torch.bfloat16).to(torch.float64) if round_p else e
        out.append((p @ v[h // 2]) / e.sum(dim=-1, keepdim=True))
    exact = torch.stack(out)
    near = exact.float().to(torch.bfloat16).float().numpy()
    ulp = np.abs(near) * 2.0 ** -7 + 1e-6
    assert np.all(np.abs(got - near) <= ulp), float(np.abs(got - near).max())
    assert float(np.mean(got == near)) > 0.9
    if round_p:  # and the rounding of P is visible: the unrounded formula lands elsewhere on some outputs
        plain = O.paged_attention(q, kp, vp, table, ctx, scale, True, Hkv, Hq, "bf16", round_p=False)
        assert np.any(plain != got)
