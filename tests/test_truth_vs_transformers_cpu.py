"""CPU tier: the float64 ground truth (oracle.TruthQwen3) against a THIRD-PARTY implementation of the architecture:
Hugging Face `transformers`' Qwen3ForCausalLM (the model family mlx_lm.models.qwen3 ports), run in float64 on the same
dequantised weights -- prefill logits of every position and KV-cached decode steps.  Every model-level tolerance in this
repository is derived from TruthQwen3; this test is what ties that truth to code the builder did not write:
wiring (pre-norm residual block, per-head q/k RMSNorm, non-traditional RoPE over the whole head, GQA by head grouping,
lower-right causal mask, SwiGLU, tied / untied output projection) and constants.  The residue (< 1e-6 on O(3) logits) is
transformers keeping its rotary table in float32.

Second part: Qwen3-MoE.  The facade's `mlx_lm` MoE model (the oracle of tests/facade_model_cases.py), run with float32
parameters so that nothing is rounded to bf16, against transformers' Qwen3MoeForCausalLM in float64 on the same checkpoint:
router softmax over all experts, top-k, renormalisation, SwiGLU experts, probability-weighted sum, dense `mlp_only_layers`.

`transformers` is part of this image (no download: the models are built from configs); skipped where it is missing.
"""

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import TINY_CFG
from oracle import tiny_oracle as O

transformers = pytest.importorskip("transformers")
ROOT = Path(__file__).resolve().parent.parent


def dense64(triple) -> torch.Tensor:
    """The checkpoint's stored weights q * scale + bias, exact in float64 (any leading expert axis is kept)."""
    packed, scales, biases = (np.asarray(a) for a in triple)
    q = O.unpack_codes(packed.reshape(-1, packed.shape[-1])).astype(np.float64).reshape(*packed.shape[:-1], -1)
    return torch.from_numpy(q * np.repeat(scales.astype(np.float64), 128, axis=-1) + np.repeat(biases.astype(np.float64), 128, axis=-1))


def vec64(a) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def attention_and_norm_tensors(w: dict) -> dict:
    sd = {"model.embed_tokens.weight": dense64(w["embed"]), "model.norm.weight": vec64(w["norm"]),
          "lm_head.weight": dense64(w.get("lm_head") or w["embed"])}
    for i, lw in enumerate(w["layers"]):
        base = f"model.layers.{i}."
        for name, key in (("self_attn.q_proj", "q"), ("self_attn.k_proj", "k"), ("self_attn.v_proj", "v"), ("self_attn.o_proj", "o")):
            sd[base + name + ".weight"] = dense64(lw[key])
        for name, key in (("self_attn.q_norm", "q_norm"), ("self_attn.k_norm", "k_norm"), ("input_layernorm", "input_norm"),
                          ("post_attention_layernorm", "post_norm")):
            sd[base + name + ".weight"] = vec64(lw[key])
    return sd


def hf_common(cfg: dict) -> dict:
    return dict(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"], rms_norm_eps=cfg["rms_norm_eps"],
                rope_theta=cfg["rope_theta"], max_position_embeddings=cfg["max_position_embeddings"],
                tie_word_embeddings=cfg["tie_word_embeddings"], attention_bias=False, attention_dropout=0.0, use_sliding_window=False)


def load_exactly(model, tensors: dict) -> None:
    missing, unexpected = model.load_state_dict(tensors, strict=False)
    assert not unexpected, unexpected
    assert all(name == "lm_head.weight" for name in missing), missing  # tied models may not list the head


@pytest.mark.parametrize("overrides,seed", [
    (dict(), 3),
    (dict(hidden_size=384, num_attention_heads=3, num_key_value_heads=1, intermediate_size=640, num_hidden_layers=3,
          tie_word_embeddings=False, rope_theta=10000), 12),
])
def test_float64_truth_agrees_with_transformers_qwen3(overrides, seed):
    cfg = dict(TINY_CFG, **overrides)
    w = O.make_qwen3_weights(cfg, seed=seed, sigma=0.05)
    hf_cfg = transformers.Qwen3Config(**hf_common(cfg))
    hf_cfg._attn_implementation = "eager"
    model = transformers.Qwen3ForCausalLM(hf_cfg).double().eval()
    tensors = attention_and_norm_tensors(w)
    for i, lw in enumerate(w["layers"]):
        for name, key in (("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
            tensors[f"model.layers.{i}.{name}.weight"] = dense64(lw[key])
    load_exactly(model, tensors)

    prompt = [int(t) for t in np.random.default_rng(seed).integers(1, cfg["vocab_size"], size=29)]
    truth, oracle = O.TruthQwen3(cfg, w), O.OracleQwen3(cfg, w)
    with torch.no_grad():
        out = model(torch.tensor([prompt]), use_cache=True)
    want = truth.forward(prompt, logits_to_keep=None)[0]
    worst = float(np.abs(out.logits[0].numpy() - want).max())
    e_oracle = float(np.abs(oracle.forward(prompt)[0, -1] - want[-1]).max())
    past, tok = out.past_key_values, int(np.argmax(want[-1]))
    for _ in range(4):  # KV-cached decode steps, teacher-forced on the truth's greedy ids
        with torch.no_grad():
            out = model(torch.tensor([[tok]]), past_key_values=past, use_cache=True)
        past = out.past_key_values
        row = truth.forward([tok])[0, -1]
        worst = max(worst, float(np.abs(out.logits[0, -1].numpy() - row).max()))
        e_oracle = max(e_oracle, float(np.abs(oracle.forward([tok])[0, -1] - row).max()))
        tok = int(np.argmax(row))
    print(f"max |truth - transformers| = {worst:.3e};  for scale: max |bf16 oracle - truth| = {e_oracle:.3e}")
    assert worst < 5e-6, worst
    assert 1e-3 < e_oracle < 0.2  # the bf16 pipeline sits four orders of magnitude further out: E of DESIGN.md §2.1


def test_product_week1_model_on_the_host_against_transformers():
    """BASELINE config 1's model (tiny_llm_hip.Qwen3ModelWeek1: dense bf16 weights from the W4 checkpoint, plain torch ops, runs on
    host tensors without the extension) against transformers in float64: the product's readable path is a bf16 pipeline, so it
    sits one bf16-pipeline error E away (E = the bf16 oracle's own distance from the same reference), not further."""
    for extra in (ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
        if str(extra) not in sys.path:
            sys.path.insert(0, str(extra))
    from helpers import to_mlx_shaped
    from tiny_llm_hip import Qwen3ModelWeek1

    cfg = dict(TINY_CFG)
    w = O.make_qwen3_weights(cfg, seed=3, sigma=0.05)
    hf_cfg = transformers.Qwen3Config(**hf_common(cfg))
    hf_cfg._attn_implementation = "eager"
    model = transformers.Qwen3ForCausalLM(hf_cfg).double().eval()
    tensors = attention_and_norm_tensors(w)
    for i, lw in enumerate(w["layers"]):
        for name, key in (("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
            tensors[f"model.layers.{i}.{name}.weight"] = dense64(lw[key])
    load_exactly(model, tensors)
    prompt = [int(t) for t in np.random.default_rng(21).integers(1, cfg["vocab_size"], size=25)]
    with torch.no_grad():
        want = model(torch.tensor([prompt])).logits[0].numpy()
        got = Qwen3ModelWeek1(to_mlx_shaped(cfg, w, device="cpu"))(torch.tensor([prompt], dtype=torch.int32))[0].float().numpy()
    e_oracle = float(np.abs(O.OracleQwen3(cfg, w).forward(prompt, logits_to_keep=None)[0] - want).max())
    e_week1 = float(np.abs(got - want).max())
    print(f"max |Week-1 model - transformers| = {e_week1:.3e}; max |bf16 oracle - transformers| = {e_oracle:.3e}")
    assert e_week1 <= 2.0 * e_oracle + 2.0 ** -7, (e_week1, e_oracle)
    assert np.array_equal(np.argmax(got, axis=-1), np.argmax(want, axis=-1)) or \
        np.all(np.take_along_axis(want, np.argmax(want, -1)[:, None], -1)[:, 0] - np.take_along_axis(want, np.argmax(got, -1)[:, None], -1)[:, 0] <= 2 * e_week1)


def hf_moe_model(cfg: dict, w: dict):
    """transformers' Qwen3MoeForCausalLM in float64 holding exactly the checkpoint's stored weights (dense mlp_only_layers, router +
    stacked experts on the sparse layers)."""
    hf_cfg = transformers.Qwen3MoeConfig(**hf_common(cfg), num_experts=cfg["num_experts"], num_experts_per_tok=cfg["num_experts_per_tok"],
                                         moe_intermediate_size=cfg["moe_intermediate_size"], norm_topk_prob=cfg["norm_topk_prob"],
                                         decoder_sparse_step=cfg["decoder_sparse_step"], mlp_only_layers=cfg["mlp_only_layers"],
                                         output_router_logits=False)
    hf_cfg._attn_implementation = "eager"
    hf_cfg._experts_implementation = "eager"  # the module's own per-expert loop (the grouped-GEMM kernel path has no float64)
    model = transformers.Qwen3MoeForCausalLM(hf_cfg).double().eval()
    model.config._experts_implementation = "eager"
    tensors = attention_and_norm_tensors(w)
    for i, lw in enumerate(w["layers"]):
        base = f"model.layers.{i}.mlp."
        if "moe" in lw:
            tensors[base + "gate.weight"] = dense64(lw["moe"]["router"])
            tensors[base + "experts.gate_up_proj"] = torch.cat([dense64(lw["moe"]["gate_proj"]), dense64(lw["moe"]["up_proj"])], dim=1)
            tensors[base + "experts.down_proj"] = dense64(lw["moe"]["down_proj"])
        else:
            for name, key in (("gate_proj", "gate"), ("up_proj", "up"), ("down_proj", "down")):
                tensors[base + name + ".weight"] = dense64(lw[key])
    load_exactly(model, tensors)
    return model


def test_facade_moe_model_agrees_with_transformers_qwen3_moe(tmp_path):
    from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights, write_checkpoint

    for extra in (ROOT / "tiny-llm_amd" / "compat", ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
        if str(extra) not in sys.path:
            sys.path.insert(0, str(extra))
    from mlx_lm.models.qwen3 import Model
    from tiny_llm_hip.loader import load_weights

    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    w = make_moe_weights(cfg, seed=5)
    model = hf_moe_model(cfg, w)

    # the SAME checkpoint through the product's loader into the facade's mlx_lm model, parameters widened to float32
    tree = load_weights(write_checkpoint(tmp_path / "moe", cfg, w), device="cpu")

    def widen(node):
        for name, value in vars(node).items():
            if isinstance(value, torch.Tensor) and value.is_floating_point():
                setattr(node, name, value.float())
            elif isinstance(value, list):
                for item in value:
                    if hasattr(item, "__dict__"):
                        widen(item)
            elif hasattr(value, "__dict__") and not isinstance(value, torch.Tensor):
                widen(value)

    widen(tree)
    facade = Model.from_checkpoint(tree)
    assert [type(layer.mlp).__name__ for layer in facade.layers] == ["MLP", "Qwen3MoeSparseMoeBlock", "Qwen3MoeSparseMoeBlock"]
    prompt = torch.tensor([[int(t) for t in np.random.default_rng(8).integers(1, cfg["vocab_size"], size=17)]])
    with torch.no_grad():
        want = model(prompt).logits[0]
    got = facade(prompt.to(torch.int32))[0]
    assert got.dtype == torch.float32
    worst = float((got.double() - want).abs().max())
    print(f"max |facade MoE model (float32) - transformers Qwen3-MoE (float64)| = {worst:.3e}")
    assert worst < 2e-4, worst  # float32 accumulation over 3 layers on O(3) logits; a routing or wiring difference is O(1)


@pytest.mark.parametrize("seed", [5, 21])
def test_float64_truth_with_moe_layers_agrees_with_transformers_qwen3_moe(seed):
    """oracle.TruthQwen3 on a checkpoint with Qwen3-MoE sparse layers (round 5: the ground truth of tests/test_engine_moe_gpu.py) against
    transformers' Qwen3MoeForCausalLM in float64 on the same stored weights, every prompt position and KV-cached decode steps: router
    softmax over all experts, top-k, renormalised scores, SwiGLU experts, weighted sum, dense layer 0 (reference: moe.py:39-89 inside
    qwen3_week3.py:209-214).  The bf16 oracle with the same layers sits one bf16 pipeline error from it."""
    from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights

    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    w = make_moe_weights(cfg, seed=seed)
    model = hf_moe_model(cfg, w)
    rng = np.random.default_rng(seed)
    prompt = [int(t) for t in rng.integers(1, cfg["vocab_size"], size=19)]
    steps = [int(t) for t in rng.integers(1, cfg["vocab_size"], size=4)]
    with torch.no_grad():
        want = model(torch.tensor([prompt + steps])).logits[0].numpy()
    truth = O.TruthQwen3(cfg, w)
    got = [truth.forward(prompt, logits_to_keep=None)[0]]
    for t in steps:
        got.append(truth.forward([t])[0])
    got = np.concatenate(got)
    worst = float(np.abs(got - want).max())
    print(f"max |TruthQwen3 with MoE layers - transformers Qwen3-MoE (float64)| = {worst:.3e} on logits up to {np.abs(want).max():.2f}")
    assert worst < 5e-6, worst  # (transformers keeps its rotary table in float32; a routing or wiring difference is O(1))
    assert all(m["same_choice"].all() for calls in truth.moe_margins.values() for m in calls)
    oracle = O.OracleQwen3(cfg, w)
    rows = [oracle.forward(prompt, logits_to_keep=None)[0]] + [oracle.forward([t])[0] for t in steps]
    err = float(np.abs(np.concatenate(rows) - want).max())
    assert 1e-3 < err < 0.15, f"the bf16 oracle with MoE layers is {err:.4f} from transformers' float64 logits"
