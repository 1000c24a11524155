"""Test-only helpers shared by the GPU parity tests: move oracle (numpy) checkpoints into the mlx_lm-shaped
object tree the product models and the decode engine consume (attribute names as read by
reference qwen3_week2.py:288-350)."""

from types import SimpleNamespace

import numpy as np
import torch

TINY_CFG = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                intermediate_size=512, vocab_size=1024, rope_theta=1000000, rms_norm_eps=1e-6,
                max_position_embeddings=4096, tie_word_embeddings=True)


def _bf16(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device, torch.bfloat16)


def _qlayer(t, device):
    packed, scales, biases = t
    return SimpleNamespace(weight=torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)).to(device),
                           scales=_bf16(scales, device), biases=_bf16(biases, device), group_size=128, bits=4)


def to_mlx_shaped(cfg: dict, w: dict, device: str = "cuda") -> SimpleNamespace:
    layers = []
    for lw in w["layers"]:
        layers.append(SimpleNamespace(
            self_attn=SimpleNamespace(
                q_proj=_qlayer(lw["q"], device), k_proj=_qlayer(lw["k"], device), v_proj=_qlayer(lw["v"], device),
                o_proj=_qlayer(lw["o"], device), q_norm=SimpleNamespace(weight=_bf16(lw["q_norm"], device)),
                k_norm=SimpleNamespace(weight=_bf16(lw["k_norm"], device))),
            mlp=SimpleNamespace(gate_proj=_qlayer(lw["gate"], device), up_proj=_qlayer(lw["up"], device),
                                down_proj=_qlayer(lw["down"], device)),
            input_layernorm=SimpleNamespace(weight=_bf16(lw["input_norm"], device)),
            post_attention_layernorm=SimpleNamespace(weight=_bf16(lw["post_norm"], device))))
    model = SimpleNamespace(embed_tokens=_qlayer(w["embed"], device), layers=layers,
                            norm=SimpleNamespace(weight=_bf16(w["norm"], device)))
    out = SimpleNamespace(args=SimpleNamespace(**cfg), model=model)
    if "lm_head" in w:
        out.lm_head = _qlayer(w["lm_head"], device)
    return out


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# Real Qwen3-4B shapes (pinned in the reference's benchmark JSON) and ground-truth based tolerances
# ---------------------------------------------------------------------------------------------------------------------
QWEN4B_CFG = dict(hidden_size=2560, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                  intermediate_size=9728, vocab_size=151936, rope_theta=1000000, rms_norm_eps=1e-6,
                  max_position_embeddings=40960, tie_word_embeddings=True)

# A model-level parity test bounds the HIP path's distance from the float64 ground truth (oracle.TruthQwen3 /
# c_oracle.CTruthQwen3) by this multiple of the bf16 oracle's own distance from it, plus one bf16 ulp of the largest
# logit (both sides are bf16-rounded, so each max-error is itself quantised to half an ulp).
TRUTH_FACTOR = 1.5


def bf16_ulp(x) -> np.ndarray:
    """Spacing of bfloat16 at |x| (8 significant bits)."""
    x = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -126)
    return 2.0 ** (np.floor(np.log2(x)) - 7)


def assert_bf16_close(got, want, ulps: float = 2.0, abs_floor: float = 0.0, what: str = ""):
    """|got - want| <= ulps * ulp_bf16(want) + abs_floor elementwise.  `want` is the oracle's bf16 result (float64
    accumulation, one rounding); a kernel that accumulates in fp32 in another order lands within one ulp of it except
    where the value is small against its own partial sums, which `abs_floor` (stated by the caller) covers."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    allowed = ulps * bf16_ulp(want) + abs_floor
    bad = np.abs(got - want) > allowed
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(got - want) - allowed), got.shape)
        raise AssertionError(f"{what}: {int(bad.sum())} of {got.size} elements outside {ulps} bf16 ulp + {abs_floor:g}; "
                             f"worst at {i}: got {got[i]!r}, want {want[i]!r}")


def ulp_of(x, dtype: str) -> np.ndarray:
    """Spacing of `dtype` ("bf16" / "f16" / "f32") at |x| (f16 subnormals: the fixed spacing 2^-24)."""
    x = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -126)
    e = np.floor(np.log2(x))
    if dtype == "bf16":
        return 2.0 ** (e - 7)
    if dtype == "f16":
        return 2.0 ** (np.maximum(e, -14) - 10)
    return 2.0 ** (e - 23)


def assert_rounded_close(got, want, dtype: str, ulps: float = 1.0, floor=0.0, what: str = ""):
    """Per-element operator tolerance (replaces a whole-tensor atol scaled by the largest output, which hid relative errors on
    small outputs): |got - want| <= ulps * ulp_dtype(want) + floor.  `want` is the oracle's result -- float64 accumulation, ONE
    rounding to `dtype`; the kernel accumulates in fp32 in another order and rounds once, so it lands within one ulp of `want`
    wherever the value is not small against the partial sums it was formed from.  That case is `floor`, which the caller
    derives from the magnitudes that were summed (e.g. 2^-20 * sum |a_k w_k| = 16 fp32 steps of the absolute sum)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert_within(got, want, ulps * ulp_of(want, dtype) + np.asarray(floor, dtype=np.float64), what)


def w4_abs_dot(a, packed, scales, biases, dtype: str) -> np.ndarray:
    """sum_k |a[m, k]| |w[n, k]| in float64 for W4 weights: the absolute sum a dot product's fp32 accumulation error scales with."""
    from oracle import tiny_oracle as O

    w = np.abs(np.asarray(O.dequantize_weights(packed, scales, biases, dtype=dtype), dtype=np.float64))
    return np.abs(np.asarray(a, dtype=np.float64)) @ w.T


def assert_within(got, want, allowed, what: str = ""):
    """|got - want| <= allowed elementwise (allowed: an array derived by the caller from the magnitudes that were rounded)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    allowed = np.broadcast_to(np.asarray(allowed, dtype=np.float64), want.shape)
    excess = np.abs(got - want) - allowed
    bad = ~(excess <= 0)  # NaN counts as bad
    if bad.any():
        i = np.unravel_index(np.nanargmax(np.where(np.isnan(excess), np.inf, excess)), got.shape)
        raise AssertionError(f"{what}: {int(bad.sum())} of {got.size} elements outside the allowance; worst at {i}: "
                             f"got {got[i]!r}, want {want[i]!r}, allowed {allowed[i]:.3g}")


def log_parity(record: dict) -> None:
    """Append one measured-parity record to gpurun_out/parity_numbers.jsonl (scratch; summaries are copied to profiles/)."""
    import json
    from pathlib import Path

    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "parity_numbers.jsonl", "a") as f:
            f.write(json.dumps(record) + "\n")
    except OSError:
        pass


# The root-mean-square error over all logits does not hang on one of 151,936 x steps values the way the maximum does: where both sides
# run the SAME arithmetic per row (decode rows: the matvec form on both sides) the HIP path's rms error must stay within this multiple
# of the bf16 checker's (measured in round 5 over every decode route at the Qwen3-4B shapes: 0.955 .. 1.06).
RMS_FACTOR_DECODE = 1.10


def check_against_truth(got, oracle, truth, what: str, factor: float = TRUTH_FACTOR, rms_factor: float | None = None) -> dict:
    """got / oracle: bf16 logits [steps, vocab] of the HIP path and of the bf16 oracle; truth: float64 logits.
    Asserts max|got - truth| <= factor * max|oracle - truth| + one bf16 ulp of the largest logit (and rms <= rms_factor * the oracle's
    rms error when given: a tighter band on the stabler statistic), logs the numbers."""
    got, oracle, truth = (np.asarray(a, dtype=np.float64) for a in (got, oracle, truth))
    e_hip = float(np.abs(got - truth).max())
    e_orc = float(np.abs(oracle - truth).max())
    rms_hip = float(np.sqrt(np.mean((got - truth) ** 2)))
    rms_orc = float(np.sqrt(np.mean((oracle - truth) ** 2)))
    ulp = float(bf16_ulp(np.abs(truth).max()))
    rec = {"what": what, "max_abs_hip_vs_truth": e_hip, "max_abs_oracle_vs_truth": e_orc,
           "max_abs_hip_vs_oracle": float(np.abs(got - oracle).max()), "rms_hip_vs_truth": rms_hip,
           "rms_oracle_vs_truth": rms_orc, "max_abs_logit": float(np.abs(truth).max()), "bf16_ulp_at_max": ulp}
    log_parity(rec)
    assert e_hip <= factor * e_orc + ulp, f"{what}: HIP is {e_hip:.4g} from the float64 truth, the bf16 oracle {e_orc:.4g}"
    assert rms_hip <= factor * rms_orc + 1e-6, f"{what}: rms error {rms_hip:.4g} (HIP) vs {rms_orc:.4g} (oracle)"
    if rms_factor is not None:
        assert rms_hip <= rms_factor * rms_orc + 1e-6, (f"{what}: rms error {rms_hip:.4g} (HIP) is {rms_hip / rms_orc:.3f} x the bf16 checker's "
                                                        f"{rms_orc:.4g}; the band for rows on the same arithmetic is {rms_factor}")
    rec["rms_ratio"] = rms_hip / rms_orc if rms_orc > 0 else None
    return rec


def oracle_weights_from_model(model) -> dict:
    """mlx_lm-shaped torch model (any device) -> the oracle's weight dict with raw uint16 bf16 bits (c_oracle accepts
    those directly; tiny_oracle functions need from_bf16_bits first)."""
    def w4(layer):
        return (layer.weight.cpu().numpy().view(np.uint32), layer.scales.view(torch.int16).cpu().numpy().view(np.uint16),
                layer.biases.view(torch.int16).cpu().numpy().view(np.uint16))

    def norm(t):
        return t.to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)

    layers = []
    for layer in model.model.layers:
        a, m = layer.self_attn, layer.mlp
        layers.append(dict(q=w4(a.q_proj), k=w4(a.k_proj), v=w4(a.v_proj), o=w4(a.o_proj), gate=w4(m.gate_proj),
                           up=w4(m.up_proj), down=w4(m.down_proj), q_norm=norm(a.q_norm.weight),
                           k_norm=norm(a.k_norm.weight), input_norm=norm(layer.input_layernorm.weight),
                           post_norm=norm(layer.post_attention_layernorm.weight)))
    out = dict(embed=w4(model.model.embed_tokens), layers=layers, norm=norm(model.model.norm.weight))
    if hasattr(model, "lm_head"):
        out["lm_head"] = w4(model.lm_head)
    return out
