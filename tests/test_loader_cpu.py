"""Checkpoint loader (tiny_llm_hip/loader.py, the mlx_lm.load replacement of SURVEY.md §8f row 1) on CPU tensors."""
import json

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import TINY_CFG
from checkpoint_fixture import write_checkpoint


@pytest.fixture(scope="module")
def weights():
    return O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)


def _loader():
    import importlib.util, pathlib, sys

    path = pathlib.Path(__file__).resolve().parents[1] / "tiny-llm_amd" / "tiny_llm_hip" / "loader.py"
    spec = importlib.util.spec_from_file_location("tl_loader_under_test", path)  # no GPU extension import on the way
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("shards", [1, 3])
def test_load_reproduces_every_tensor(tmp_path, weights, shards):
    loader = _loader()
    ckpt = write_checkpoint(tmp_path / "ckpt", TINY_CFG, weights, shards=shards)
    model, tok = loader.load(str(ckpt), device="cpu")
    a = model.args
    assert (a.hidden_size, a.num_hidden_layers, a.head_dim, a.vocab_size) == (256, 2, 128, 1024) and a.tie_word_embeddings
    assert not hasattr(model, "lm_head") and len(model.model.layers) == 2

    def same(layer, triple):
        packed, scales, biases = triple
        assert layer.weight.dtype == torch.int32 and layer.scales.dtype == torch.bfloat16
        assert layer.group_size == 128 and layer.bits == 4
        np.testing.assert_array_equal(layer.weight.numpy().view(np.uint32), np.asarray(packed, dtype=np.uint32))
        np.testing.assert_array_equal(layer.scales.float().numpy(), np.asarray(scales, dtype=np.float32))
        np.testing.assert_array_equal(layer.biases.float().numpy(), np.asarray(biases, dtype=np.float32))

    same(model.model.embed_tokens, weights["embed"])
    for layer, lw in zip(model.model.layers, weights["layers"]):
        for attr, key in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("o_proj", "o")):
            same(getattr(layer.self_attn, attr), lw[key])
        for attr, key in (("gate_proj", "gate"), ("up_proj", "up"), ("down_proj", "down")):
            same(getattr(layer.mlp, attr), lw[key])
        np.testing.assert_array_equal(layer.self_attn.q_norm.weight.float().numpy(), np.asarray(lw["q_norm"], np.float32))
        np.testing.assert_array_equal(layer.input_layernorm.weight.float().numpy(), np.asarray(lw["input_norm"], np.float32))
    np.testing.assert_array_equal(model.model.norm.weight.float().numpy(), np.asarray(weights["norm"], np.float32))

    # tokenizer wrapper: the surface the generation loops use
    ids = tok.encode("hello tiny llm", add_special_tokens=False)
    assert ids == [2, 4, 5] and tok.eos_token_id == 0 and tok.eos_token_ids == {0}
    assert tok.get_vocab()["world"] == 3
    detok = tok.detokenizer
    detok.reset()
    pieces = []
    for t in ids:
        detok.add_token(t)
        pieces.append(detok.last_segment)
    assert "".join(pieces) == detok.text == tok.decode(ids)
    prompt = tok.apply_chat_template([{"role": "user", "content": "hello"}], tokenize=False, add_generation_prompt=True)
    assert prompt.startswith("user hello") and prompt.endswith("assistant")

    # the scheduler builds a private detokenizer per request exactly like the reference
    # (batch.py:23: tokenizer.detokenizer.__class__(tokenizer._tokenizer)): the loader's wrapper must carry both
    from types import SimpleNamespace

    from tiny_llm_hip.batch import Request

    fake_model = SimpleNamespace(create_kv_cache=lambda: [])
    req = Request(fake_model, tok, "hello tiny llm", prefill_max_step=2, device="cpu")
    assert req.prefill_tokens.tolist() == [2, 4, 5] and req.eos_token_id == 0
    assert req.detokenizer is not tok.detokenizer
    for t in ids:
        req.detokenizer.add_token(t)
    assert req.detokenizer.text == tok.decode(ids)


@pytest.mark.parametrize("mutate,message", [
    (lambda cfg, path: cfg["quantization"].update(group_size=64), "group_size=64"),
    (lambda cfg, path: cfg["quantization"].update(bits=8), "bits=8"),
    (lambda cfg, path: cfg.pop("quantization"), "not quantized"),
    (lambda cfg, path: cfg.update(num_hidden_layers=3), "model.layers.2"),
    (lambda cfg, path: cfg.update(intermediate_size=1024), "do not describe"),
    (lambda cfg, path: cfg.pop("rms_norm_eps"), "lacks"),
    # scaled RoPE variants are refused even when a converter has hoisted rope_theta to the top level (round-5 review)
    (lambda cfg, path: cfg.update(rope_theta=1e6, rope_parameters={"rope_theta": 1e6, "rope_type": "yarn", "factor": 4.0}), "rope_type='yarn'"),
    (lambda cfg, path: cfg.update(rope_theta=1e6, rope_scaling={"type": "linear", "factor": 2.0}), "rope_scaling"),
])
def test_load_rejects_what_the_hot_path_cannot_run(tmp_path, weights, mutate, message):
    loader = _loader()
    ckpt = write_checkpoint(tmp_path / "ckpt", TINY_CFG, weights)
    cfg = json.loads((ckpt / "config.json").read_text())
    mutate(cfg, ckpt)
    (ckpt / "config.json").write_text(json.dumps(cfg))
    with pytest.raises((ValueError, KeyError), match=message):
        loader.load_weights(ckpt, device="cpu")


def test_untied_head_and_missing_directory(tmp_path):
    loader = _loader()
    cfg = dict(TINY_CFG, tie_word_embeddings=False)
    w = O.make_qwen3_weights(cfg, seed=9, sigma=0.05)
    assert "lm_head" in w
    model = loader.load_weights(write_checkpoint(tmp_path / "untied", cfg, w), device="cpu")
    np.testing.assert_array_equal(model.lm_head.weight.numpy().view(np.uint32), np.asarray(w["lm_head"][0], np.uint32))
    with pytest.raises(FileNotFoundError, match="neither a checkpoint directory"):
        loader.resolve_model_dir(str(tmp_path / "nope"))


def test_load_reads_a_qwen3_moe_checkpoint(tmp_path):
    """Sparse layers come back as `mlp.gate` + `mlp.switch_mlp.{gate,up,down}_proj` with a leading expert axis (the tree
    reference qwen3_week3.py:258-272 builds its Moe blocks from); dense layers (mlp_only_layers) keep the dense MLP."""
    from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights

    loader = _loader()
    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    w = make_moe_weights(cfg, seed=5)
    ckpt = write_checkpoint(tmp_path / "moe", cfg, w, shards=2)
    assert json.loads((ckpt / "config.json").read_text())["model_type"] == "qwen3_moe"
    model, _ = loader.load(str(ckpt), device="cpu")
    a = model.args
    assert (a.num_experts, a.num_experts_per_tok, a.moe_intermediate_size, a.norm_topk_prob) == (4, 2, 256, True)
    assert hasattr(model.model.layers[0].mlp, "gate_proj") and not hasattr(model.model.layers[0].mlp, "switch_mlp")
    for layer, lw in zip(model.model.layers[1:], w["layers"][1:]):
        assert not hasattr(layer.mlp, "gate_proj")
        for got, want in ((layer.mlp.gate, lw["moe"]["router"]), (layer.mlp.switch_mlp.gate_proj, lw["moe"]["gate_proj"]),
                          (layer.mlp.switch_mlp.up_proj, lw["moe"]["up_proj"]), (layer.mlp.switch_mlp.down_proj, lw["moe"]["down_proj"])):
            np.testing.assert_array_equal(got.weight.numpy().view(np.uint32), np.asarray(want[0], dtype=np.uint32))
            np.testing.assert_array_equal(got.scales.float().numpy(), np.asarray(want[1], dtype=np.float32))
            np.testing.assert_array_equal(got.biases.float().numpy(), np.asarray(want[2], dtype=np.float32))
        assert tuple(layer.mlp.switch_mlp.down_proj.weight.shape) == (4, 256, 256 // 8)

    # a config that announces experts but not their width is refused by name
    broken = json.loads((ckpt / "config.json").read_text())
    del broken["moe_intermediate_size"]
    (ckpt / "config.json").write_text(json.dumps(broken))
    with pytest.raises(ValueError, match="moe_intermediate_size"):
        loader.load_weights(ckpt, device="cpu")


def test_course_models_on_loaded_checkpoints_against_the_facade_mlx_lm_model(built_libs):
    """Loader -> Week 1 / Week 2 / Week 3 (dense and Qwen3-MoE) models -> logits, against the facade's mlx_lm model on the
    same tensors, in the pattern of the reference's checkpoint-dependent tests (tests/facade_model_cases.py).  The numpy
    oracle answers the C ABI in this container; the same cases run on the HIP kernels from tests/test_zz_facade_models_gpu.py."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=str(root / "tests"))
    proc = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "facade_model_cases.py"), "-p", "no:cacheprovider",
                           "-p", "refsol_oracle_plugin", "-q", "--tb=short"], cwd=root, env=env, capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0 and "4 passed" in proc.stdout, proc.stdout[-3000:]
