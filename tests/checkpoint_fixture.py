"""Writes a tiny MLX-format 4-bit Qwen3 checkpoint directory (config.json, model.safetensors, tokenizer files) from oracle
weights, for the loader tests.  Tensor names and dtypes follow the mlx-community 4-bit exports the reference loads through
mlx_lm (uint32 packed words, bf16 scales / biases / norms, {"quantization": {"group_size": 128, "bits": 4}})."""
import json
from pathlib import Path

import numpy as np
import torch


def _u32(packed: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)).view(torch.uint32)


def _bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)


def write_checkpoint(path: Path, cfg: dict, w: dict, *, group_size: int = 128, bits: int = 4, shards: int = 1,
                     vocab_words=None) -> Path:
    from safetensors.torch import save_file

    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    tensors = {}

    def put(prefix, triple):
        packed, scales, biases = triple
        tensors[f"{prefix}.weight"], tensors[f"{prefix}.scales"], tensors[f"{prefix}.biases"] = _u32(packed), _bf16(scales), _bf16(biases)

    put("model.embed_tokens", w["embed"])
    for i, lw in enumerate(w["layers"]):
        base = f"model.layers.{i}"
        for name, key in (("self_attn.q_proj", "q"), ("self_attn.k_proj", "k"), ("self_attn.v_proj", "v"),
                          ("self_attn.o_proj", "o"), ("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
            put(f"{base}.{name}", lw[key])
        tensors[f"{base}.self_attn.q_norm.weight"] = _bf16(lw["q_norm"])
        tensors[f"{base}.self_attn.k_norm.weight"] = _bf16(lw["k_norm"])
        tensors[f"{base}.input_layernorm.weight"] = _bf16(lw["input_norm"])
        tensors[f"{base}.post_attention_layernorm.weight"] = _bf16(lw["post_norm"])
    tensors["model.norm.weight"] = _bf16(w["norm"])
    if "lm_head" in w:
        put("lm_head", w["lm_head"])
    config = dict(cfg, model_type="qwen3", quantization={"group_size": group_size, "bits": bits})
    (path / "config.json").write_text(json.dumps(config))
    if shards == 1:
        save_file(tensors, str(path / "model.safetensors"), metadata={"format": "mlx"})
    else:
        names = sorted(tensors)
        weight_map = {}
        for s in range(shards):
            part = {n: tensors[n] for n in names[s::shards]}
            fname = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
            save_file(part, str(path / fname), metadata={"format": "mlx"})
            weight_map.update({n: fname for n in part})
        (path / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": weight_map}))

    # a word-level tokenizer is enough to exercise the wrapper (ids 0.. = <eos>, then the words)
    from tokenizers import Tokenizer, models, pre_tokenizers

    words = list(vocab_words or ["hello", "world", "tiny", "llm", "on", "mi355x", "!", "\\n"])
    vocab = {"<eos>": 0, "<unk>": 1, **{wd: i + 2 for i, wd in enumerate(words)}}
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.save(str(path / "tokenizer.json"))
    (path / "tokenizer_config.json").write_text(json.dumps({
        "tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "<eos>", "unk_token": "<unk>",
        "chat_template": "{% for m in messages %}{{ m['role'] }} {{ m['content'] }} {% endfor %}{% if add_generation_prompt %}assistant{% endif %}"}))
    (path / "generation_config.json").write_text(json.dumps({"eos_token_id": [0]}))
    return path


def write_hf_cache_snapshot(hf_home: Path, repo_id: str, cfg: dict, w: dict, **kwargs) -> Path:
    """The same checkpoint laid out as a snapshot of `repo_id` in a Hugging Face cache rooted at `hf_home`
    (hub/models--ORG--NAME/{refs/main, snapshots/<revision>/...}): what `huggingface_hub.snapshot_download(repo_id,
    local_files_only=True)` resolves, i.e. how the reference finds "Qwen/Qwen3-0.6B-MLX-4bit" (tests_refsol/utils.py:118-125,
    benches/bench.py:601-609)."""
    revision = "0" * 40
    repo = Path(hf_home) / "hub" / ("models--" + repo_id.replace("/", "--"))
    (repo / "refs").mkdir(parents=True, exist_ok=True)
    (repo / "refs" / "main").write_text(revision)
    return write_checkpoint(repo / "snapshots" / revision, cfg, w, **kwargs)
