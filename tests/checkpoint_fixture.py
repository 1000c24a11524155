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
                          ("self_attn.o_proj", "o")):
            put(f"{base}.{name}", lw[key])
        if "moe" in lw:  # Qwen3-MoE layer: router + stacked experts [E, out, in] (mlx_lm SwitchGLU tensor names)
            put(f"{base}.mlp.gate", lw["moe"]["router"])
            for name in ("gate_proj", "up_proj", "down_proj"):
                put(f"{base}.mlp.switch_mlp.{name}", lw["moe"][name])
        else:
            for name, key in (("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
                put(f"{base}.{name}", lw[key])
        tensors[f"{base}.self_attn.q_norm.weight"] = _bf16(lw["q_norm"])
        tensors[f"{base}.self_attn.k_norm.weight"] = _bf16(lw["k_norm"])
        tensors[f"{base}.input_layernorm.weight"] = _bf16(lw["input_norm"])
        tensors[f"{base}.post_attention_layernorm.weight"] = _bf16(lw["post_norm"])
    tensors["model.norm.weight"] = _bf16(w["norm"])
    if "lm_head" in w:
        put("lm_head", w["lm_head"])
    config = dict(cfg, model_type="qwen3_moe" if cfg.get("num_experts") else "qwen3",
                  quantization={"group_size": group_size, "bits": bits})
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


# Synthetic stand-ins for the checkpoints the reference's tests and scripts look up by repository name (no download is
# possible here): small Qwen3-shaped W4 models, (repo id, config overrides on helpers.TINY_CFG, seed).
STAND_INS = [
    ("Qwen/Qwen3-0.6B-MLX-4bit", dict(), 11),
    ("Qwen/Qwen3-1.7B-MLX-4bit", dict(hidden_size=384, num_attention_heads=3, num_key_value_heads=1, intermediate_size=640,
                                      num_hidden_layers=3, tie_word_embeddings=False), 12),
    ("Qwen/Qwen3-4B-MLX-4bit", dict(hidden_size=512, num_attention_heads=8, num_key_value_heads=2, intermediate_size=768), 13),
]
# ... and one whose greedy generation TERMINATES (the reference's main.py / batch-main.py generate until <eos>, which a
# random model practically never emits): an untied output projection whose <eos> row is 6x larger than the others
EOS_FRIENDLY = ("Qwen/Qwen3-8B-MLX-4bit", dict(tie_word_embeddings=False), 23)


def write_stand_in_checkpoints(hf_home: Path, eos_friendly: bool = False) -> None:
    from helpers import TINY_CFG
    from oracle import tiny_oracle as O

    words = [f"w{i}" for i in range(200)]
    for repo_id, overrides, seed in STAND_INS:
        cfg = dict(TINY_CFG, **overrides)
        write_hf_cache_snapshot(hf_home, repo_id, cfg, O.make_qwen3_weights(cfg, seed=seed, sigma=0.05), vocab_words=words)
    if eos_friendly:
        repo_id, overrides, seed = EOS_FRIENDLY
        cfg = dict(TINY_CFG, **overrides)
        w = O.make_qwen3_weights(cfg, seed=seed, sigma=0.05)
        rng = np.random.default_rng(seed)
        head = O.bf16(rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"])).astype(np.float32) * 0.05)
        head[0] = O.bf16(head[0] * 6.0)
        w["lm_head"] = O.quantize_affine(head)
        write_hf_cache_snapshot(hf_home, repo_id, cfg, w, vocab_words=words)


MOE_CFG_OVERRIDES = dict(num_experts=4, num_experts_per_tok=2, moe_intermediate_size=256, norm_topk_prob=True,
                         decoder_sparse_step=1, mlp_only_layers=[0], num_hidden_layers=3)


def make_moe_weights(cfg: dict, seed: int = 0, sigma: float = 0.05) -> dict:
    """make_qwen3_weights plus, on every sparse layer (reference is_qwen3_moe_sparse_layer), a W4 router [E, hidden] and
    stacked W4 experts gate/up [E, moe_inter, hidden], down [E, hidden, moe_inter]."""
    from oracle import tiny_oracle as O

    w = O.make_qwen3_weights(cfg, seed=seed, sigma=sigma)
    rng = np.random.default_rng(seed + 1000)
    E, hs, mi = cfg["num_experts"], cfg["hidden_size"], cfg["moe_intermediate_size"]

    def stack(out_dim, in_dim):
        parts = [O.quantize_affine(O.bf16(rng.standard_normal((out_dim, in_dim), dtype=np.float32) * sigma)) for _ in range(E)]
        return tuple(np.stack([p[j] for p in parts]) for j in range(3))

    for i, lw in enumerate(w["layers"]):
        if i in cfg.get("mlp_only_layers", []) or (i + 1) % cfg.get("decoder_sparse_step", 1) != 0:
            continue
        for key in ("gate", "up", "down"):
            lw.pop(key)
        lw["moe"] = dict(router=O.quantize_affine(O.bf16(rng.standard_normal((E, hs), dtype=np.float32) * 0.5)),
                         gate_proj=stack(mi, hs), up_proj=stack(mi, hs), down_proj=stack(hs, mi))
    return w
