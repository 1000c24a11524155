"""The loader against a checkpoint directory WRITTEN BY A THIRD PARTY: Hugging Face transformers builds a tiny Qwen3ForCausalLM and
saves it itself (`save_pretrained`: its tensor names, its config.json keys, its sharded index), `tools/convert_hf_to_mlx4bit.py` turns
that directory into the MLX 4-bit layout (what `mlx_lm.convert -q` produces on a Mac), and `tiny_llm_hip.loader.load_weights` reads
it.  Until round 5 the loader had only seen directories written by this repository's own fixture (tests/checkpoint_fixture.py) --
names typed by hand.  No real `Qwen/Qwen3-4B-MLX-4bit` is on disk and there is no network; this pins everything about the format
that does not need one: names, config keys, shard index, dtypes, shapes, the quantised triples (reference: main.py:96-110 loads
through mlx_lm; quantize.py:29-46 reads weight / scales / biases / group_size / bits of every layer)."""
import json
import sys
from pathlib import Path

import pytest
import torch

transformers = pytest.importorskip("transformers")
safetensors = pytest.importorskip("safetensors")

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

CFG = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=512,
           vocab_size=640, rms_norm_eps=1e-6, rope_theta=500000.0, max_position_embeddings=4096)  # (not the loader's default theta)


def _hf_checkpoint(path: Path, tie: bool):
    torch.manual_seed(11)
    cfg = transformers.Qwen3Config(**CFG, tie_word_embeddings=tie, torch_dtype="bfloat16")
    model = transformers.Qwen3ForCausalLM(cfg).to(torch.bfloat16)
    model.save_pretrained(str(path), max_shard_size="400KB", safe_serialization=True)  # small shards: transformers writes its own index
    return model


def _dequant(layer) -> torch.Tensor:
    words = layer.weight.to(torch.int64) & 0xFFFFFFFF
    codes = torch.stack([(words >> (4 * i)) & 0xF for i in range(8)], dim=-1).reshape(words.shape[0], -1).to(torch.float32)
    s = layer.scales.to(torch.float32).repeat_interleave(128, dim=1)
    b = layer.biases.to(torch.float32).repeat_interleave(128, dim=1)
    return codes * s + b


@pytest.mark.parametrize("tie", [True, False])
def test_loader_reads_what_transformers_wrote_and_the_converter_quantised(tmp_path, tie):
    from convert_hf_to_mlx4bit import convert
    from tiny_llm_hip.loader import load_weights
    from tiny_llm_hip.synthetic import quantize

    hf_dir, mlx_dir = tmp_path / "hf", tmp_path / "mlx4"
    model = _hf_checkpoint(hf_dir, tie)
    assert (hf_dir / "model.safetensors.index.json").is_file(), "transformers was asked for several shards"
    report = convert(hf_dir, mlx_dir, shards=3)
    state = {k: v for k, v in model.state_dict().items()}
    linear_names = [k for k in state if k.endswith(("_proj.weight", "embed_tokens.weight")) or (k == "lm_head.weight" and not tie)]
    assert sorted(report["quantized"]) == sorted(linear_names)
    assert set(report["skipped"]) <= ({"lm_head.weight"} if tie else set())  # (transformers itself leaves a tied head out of the file)

    loaded = load_weights(mlx_dir, device="cpu")
    hf_config = json.loads((hf_dir / "config.json").read_text())
    for key in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim", "intermediate_size",
                "vocab_size", "rms_norm_eps", "tie_word_embeddings", "max_position_embeddings"):
        assert getattr(loaded.args, key) == hf_config[key], key
    # transformers >= 5 nests the RoPE base ({"rope_parameters": {"rope_theta": ...}}); older versions (and the MLX exports made from them)
    # carry it at the top level: the loader takes either, the converter writes both
    hf_theta = hf_config["rope_theta"] if "rope_theta" in hf_config else hf_config["rope_parameters"]["rope_theta"]
    assert hf_theta == 500000.0 and loaded.args.rope_theta == hf_theta
    assert json.loads((mlx_dir / "config.json").read_text())["rope_theta"] == hf_theta
    assert loaded.args.quantization == {"group_size": 128, "bits": 4}

    def check_linear(layer, name):
        w = state[name].to(torch.bfloat16)
        words, scales, biases = quantize(w)
        assert torch.equal(layer.weight, words) and torch.equal(layer.scales, scales) and torch.equal(layer.biases, biases), name
        assert layer.group_size == 128 and layer.bits == 4
        err = (_dequant(layer) - w.to(torch.float32)).abs()
        # sanity of the triple itself (the packing is pinned above; mx.quantize snaps the group's edge to a multiple of the scale, which can
        # shrink the step by up to 1/16 and leave the far end of the range one step short): never more than ~a step, a quarter on average
        step = layer.scales.to(torch.float32).abs().repeat_interleave(128, dim=1)
        assert bool((err <= 1.25 * step + 1e-6).all()) and float((err / step).mean()) < 0.3, name

    check_linear(loaded.model.embed_tokens, "model.embed_tokens.weight")
    for i, layer in enumerate(loaded.model.layers):
        for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
            check_linear(getattr(layer.self_attn, proj), f"model.layers.{i}.self_attn.{proj}.weight")
        for proj in ("gate_proj", "up_proj", "down_proj"):
            check_linear(getattr(layer.mlp, proj), f"model.layers.{i}.mlp.{proj}.weight")
        for got, name in ((layer.self_attn.q_norm, "self_attn.q_norm"), (layer.self_attn.k_norm, "self_attn.k_norm"),
                          (layer.input_layernorm, "input_layernorm"), (layer.post_attention_layernorm, "post_attention_layernorm")):
            assert torch.equal(got.weight, state[f"model.layers.{i}.{name}.weight"].to(torch.bfloat16)), name
    assert torch.equal(loaded.model.norm.weight, state["model.norm.weight"].to(torch.bfloat16))
    if tie:
        assert not hasattr(loaded, "lm_head")
    else:
        check_linear(loaded.lm_head, "lm_head.weight")

    # every tensor transformers wrote is accounted for: quantised, copied, or the tied head
    from safetensors import safe_open

    written = set()
    for f in sorted(hf_dir.glob("*.safetensors")):
        with safe_open(str(f), framework="pt") as h:
            written |= set(h.keys())
    assert written == set(report["quantized"]) | set(report["copied"]) | set(report["skipped"])


def test_converter_refuses_what_it_cannot_represent(tmp_path):
    import subprocess

    from convert_hf_to_mlx4bit import convert

    hf_dir = tmp_path / "hf"
    _hf_checkpoint(hf_dir, True)
    # the command line, in a process of its own (no test path set-up behind it)
    done = subprocess.run([sys.executable, str(ROOT / "tools" / "convert_hf_to_mlx4bit.py"), str(hf_dir), str(tmp_path / "cli")],
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "quantized 15 matrices" in done.stdout and (tmp_path / "cli" / "model.safetensors").is_file()
    cfg = json.loads((hf_dir / "config.json").read_text())
    (hf_dir / "config.json").write_text(json.dumps(dict(cfg, quantization={"group_size": 64, "bits": 4})))
    with pytest.raises(ValueError, match="already quantized"):
        convert(hf_dir, tmp_path / "out")
    (hf_dir / "config.json").write_text(json.dumps(dict(cfg, num_experts=8)))
    with pytest.raises(ValueError, match="MoE"):
        convert(hf_dir, tmp_path / "out2")


@pytest.mark.parametrize("tie", [True, False])
def test_loaded_tree_computes_what_transformers_computes_on_the_same_files(tmp_path, tie):
    """End to end on the host, two independent readers of ONE converted directory: (a) the loader's attribute tree -> the float64 truth
    model of this repository (oracle.TruthQwen3, itself pinned to transformers on builder-made weights); (b) the files' tensors taken BY
    NAME, dequantised, and loaded by transformers' own `load_state_dict` into its Qwen3ForCausalLM in float64.  A projection landing
    in the wrong attribute, a norm on the wrong layer, a transposed matrix or a mis-read config field separates the two."""
    import numpy as np
    from safetensors import safe_open

    from convert_hf_to_mlx4bit import convert
    from oracle import tiny_oracle as O
    from test_truth_vs_transformers_cpu import dense64, load_exactly
    from tiny_llm_hip.loader import load_weights

    hf_dir, mlx_dir = tmp_path / "hf", tmp_path / "mlx4"
    _hf_checkpoint(hf_dir, tie)
    convert(hf_dir, mlx_dir, shards=2)

    # (a) through the loader
    loaded = load_weights(mlx_dir, device="cpu")
    cfg = {k: getattr(loaded.args, k) for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
                                                "intermediate_size", "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings",
                                                "max_position_embeddings")}

    def triple(layer):
        return (layer.weight.numpy().view(np.uint32), layer.scales.float().numpy(), layer.biases.float().numpy())

    def vec(n):
        return n.weight.float().numpy()

    w = dict(embed=triple(loaded.model.embed_tokens), norm=vec(loaded.model.norm), layers=[])
    for layer in loaded.model.layers:
        a, m = layer.self_attn, layer.mlp
        w["layers"].append(dict(q=triple(a.q_proj), k=triple(a.k_proj), v=triple(a.v_proj), o=triple(a.o_proj), gate=triple(m.gate_proj),
                                up=triple(m.up_proj), down=triple(m.down_proj), q_norm=vec(a.q_norm), k_norm=vec(a.k_norm),
                                input_norm=vec(layer.input_layernorm), post_norm=vec(layer.post_attention_layernorm)))
    if not tie:
        w["lm_head"] = triple(loaded.lm_head)
    truth = O.TruthQwen3(cfg, w)

    # (b) by name, through transformers
    stored = {}
    for f in sorted(mlx_dir.glob("*.safetensors")):
        with safe_open(str(f), framework="pt") as h:
            for key in h.keys():
                stored[key] = h.get_tensor(key)
    tensors = {}
    for name, t in stored.items():
        if name.endswith((".scales", ".biases")):
            continue
        base = name[: -len(".weight")]
        if base + ".scales" in stored:
            tensors[name] = dense64((t.view(torch.int32).numpy().view(np.uint32), stored[base + ".scales"].float().numpy(),
                                     stored[base + ".biases"].float().numpy()))
        else:
            tensors[name] = t.to(torch.float64)
    hf_cfg = transformers.Qwen3Config(**CFG, tie_word_embeddings=tie, attention_bias=False)
    hf_cfg._attn_implementation = "eager"
    model = transformers.Qwen3ForCausalLM(hf_cfg).double().eval()
    load_exactly(model, tensors)

    prompt = [int(t) for t in np.random.default_rng(5).integers(1, CFG["vocab_size"], size=23)]
    with torch.no_grad():
        out = model(torch.tensor([prompt]), use_cache=True)
    want = truth.forward(prompt, logits_to_keep=None)[0]
    worst = float(np.abs(out.logits[0].numpy() - want).max())
    past, tok = out.past_key_values, int(np.argmax(want[-1]))
    for _ in range(3):
        with torch.no_grad():
            out = model(torch.tensor([[tok]]), past_key_values=past, use_cache=True)
        past = out.past_key_values
        row = truth.forward([tok])[0, -1]
        worst = max(worst, float(np.abs(out.logits[0, -1].numpy() - row).max()))
        tok = int(np.argmax(row))
    assert worst < 5e-6, worst
