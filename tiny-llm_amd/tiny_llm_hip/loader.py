"""Checkpoint + tokenizer loading without mlx_lm (reference: ``mlx_lm.load`` as used by main.py:96-110, batch-main.py:62-75,
benches/bench.py and ``QuantizedWeights.from_mlx_layer``, src/tiny_llm_ref/quantize.py:29-46).

``load(path)`` reads an MLX-format 4-bit checkpoint directory (``config.json``, ``model.safetensors`` or a sharded index,
tokenizer files) and returns the same two objects the reference gets from ``mlx_lm.load``:

* a model object whose attribute tree is the one the reference walks (``model.args``, ``model.model.embed_tokens``,
  ``model.model.layers[i].self_attn.q_proj.{weight,scales,biases,group_size,bits}`` ...), with torch tensors on ``device``
  (packed weights as int32 views of the stored uint32 words, scales / biases / norms in the stored 16-bit float type);
* a tokenizer wrapper with the surface the generation loops use (``encode``, ``eos_token_id``, ``get_vocab``,
  ``detokenizer`` with ``reset / add_token / last_segment / text / finalize``, ``apply_chat_template``).

Qwen3-MoE checkpoints (``num_experts`` in config.json) load too: sparse layers carry ``mlp.gate`` (router) and
``mlp.switch_mlp.{gate,up,down}_proj`` with a leading expert axis, the tree Qwen3ModelWeek3 builds its Moe blocks from
(reference qwen3_week3.py:258-272).

Only what the hot path supports is accepted: affine 4-bit weights in groups of 128 (quantize.py:103-121);
anything else raises ``ValueError`` naming the offending field.  No network access: names are resolved in the local
Hugging Face cache only.
"""

from __future__ import annotations

import json
from pathlib import Path
from types import SimpleNamespace

import torch

__all__ = ["load", "load_weights", "resolve_model_dir", "TokenizerWrapper"]

_LINEARS = {
    "self_attn": ("q_proj", "k_proj", "v_proj", "o_proj"),
    "mlp": ("gate_proj", "up_proj", "down_proj"),
}
_CONFIG_KEYS = ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "intermediate_size",
                "vocab_size", "rms_norm_eps")


def resolve_model_dir(name_or_path: str) -> Path:
    """A directory that holds config.json: the path itself, or the snapshot of a repo id in the local HF cache."""
    p = Path(name_or_path).expanduser()
    if (p / "config.json").is_file():
        return p
    try:
        from huggingface_hub import snapshot_download

        return Path(snapshot_download(name_or_path, local_files_only=True))
    except Exception as exc:  # no network here: a missing snapshot is a user-facing error, not a download trigger
        raise FileNotFoundError(
            f"{name_or_path!r} is neither a checkpoint directory nor a repo in the local Hugging Face cache") from exc


class _TensorSource:
    """Name -> tensor over one or several safetensors files (``model.safetensors.index.json`` lists the shards)."""

    def __init__(self, model_dir: Path):
        from safetensors import safe_open

        index = model_dir / "model.safetensors.index.json"
        if index.is_file():
            weight_map = json.loads(index.read_text())["weight_map"]
            files = sorted(set(weight_map.values()))
        else:
            files = sorted(f.name for f in model_dir.glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors file in {model_dir}")
        self._open = {f: safe_open(str(model_dir / f), framework="pt") for f in files}
        self._where = {}
        for f, handle in self._open.items():
            for key in handle.keys():
                self._where[key] = f

    def __contains__(self, name: str) -> bool:
        return name in self._where

    def get(self, name: str) -> torch.Tensor:
        if name not in self._where:
            raise KeyError(f"checkpoint has no tensor named {name!r}")
        return self._open[self._where[name]].get_tensor(name)


def _quantization(config: dict) -> tuple[int, int]:
    q = config.get("quantization") or config.get("quantization_config")
    if not isinstance(q, dict):
        raise ValueError("checkpoint is not quantized (config.json has no 'quantization' section): the hot path is W4A16")
    group_size, bits = int(q.get("group_size", 0)), int(q.get("bits", 0))
    if bits != 4:
        raise ValueError(f"unsupported quantization bits={bits}: only 4-bit weights are supported")
    if group_size != 128:
        raise ValueError(f"unsupported quantization group_size={group_size}: only groups of 128 are supported")
    return group_size, bits


def load_weights(model_dir: str | Path, device: str = "cuda") -> SimpleNamespace:
    """The mlx_lm-shaped model object of an MLX 4-bit Qwen3 checkpoint directory (see module docstring)."""
    model_dir = Path(model_dir)
    config = json.loads((model_dir / "config.json").read_text())
    missing = [k for k in _CONFIG_KEYS if k not in config]
    if missing:
        raise ValueError(f"config.json lacks {missing}")
    group_size, bits = _quantization(config)
    config.setdefault("head_dim", config["hidden_size"] // config["num_attention_heads"])
    # Only plain RoPE is what the hot path computes (reference qwen3_week3.py:230: theta alone): a scaled variant -- transformers >= 5's
    # {"rope_parameters": {"rope_type": "yarn" | "linear" | ...}} or the legacy {"rope_scaling": {...}} -- is refused whether or not a
    # top-level rope_theta is present (a converter may have hoisted it), instead of loading silently as plain RoPE
    rp = config.get("rope_parameters")
    if isinstance(rp, dict) and rp.get("rope_type", "default") not in ("default", None):
        raise ValueError(f"unsupported rope_parameters.rope_type={rp.get('rope_type')!r}: only plain RoPE is supported")
    rs = config.get("rope_scaling")
    if isinstance(rs, dict) and (rs.get("rope_type") or rs.get("type") or "default") != "default":
        raise ValueError(f"unsupported rope_scaling={rs!r}: only plain RoPE is supported")
    if "rope_theta" not in config and isinstance(rp, dict) and "rope_theta" in rp:
        config["rope_theta"] = rp["rope_theta"]  # transformers >= 5 nests it; mlx_lm's ModelArgs (and the reference) read the top-level key
    config.setdefault("rope_theta", 1000000)
    config.setdefault("tie_word_embeddings", True)
    config.setdefault("max_position_embeddings", 40960)
    src = _TensorSource(model_dir)

    def linear(prefix: str, out_dim: int, in_dim: int) -> SimpleNamespace:
        w, s, b = (src.get(f"{prefix}.{part}") for part in ("weight", "scales", "biases"))
        if w.dtype not in (torch.uint32, torch.int32):
            raise ValueError(f"{prefix}.weight: expected packed uint32 words, found {w.dtype}")
        if s.dtype not in (torch.bfloat16, torch.float16) or b.dtype != s.dtype:
            raise ValueError(f"{prefix}: scales/biases must share one 16-bit float type, found {s.dtype}/{b.dtype}")
        if tuple(w.shape) != (out_dim, in_dim * bits // 32) or tuple(s.shape) != (out_dim, in_dim // group_size) \
                or tuple(b.shape) != tuple(s.shape):
            raise ValueError(f"{prefix}: shapes {tuple(w.shape)}, {tuple(s.shape)}, {tuple(b.shape)} do not describe a "
                             f"[{out_dim}, {in_dim}] matrix in {bits}-bit groups of {group_size}")
        return SimpleNamespace(weight=w.view(torch.int32).to(device), scales=s.to(device), biases=b.to(device),
                               group_size=group_size, bits=bits)

    def norm(name: str, n: int) -> SimpleNamespace:
        w = src.get(name)
        if tuple(w.shape) != (n,):
            raise ValueError(f"{name}: expected shape ({n},), found {tuple(w.shape)}")
        return SimpleNamespace(weight=w.to(device))

    def experts(prefix: str, count: int, out_dim: int, in_dim: int) -> SimpleNamespace:
        """One stacked expert projection of a Qwen3-MoE layer (mlx_lm SwitchLinear): weight [E, out, in/8] u32,
        scales / biases [E, out, in/128] (what grouped_expert_linear consumes, reference moe.py:7-36)."""
        w, s, b = (src.get(f"{prefix}.{part}") for part in ("weight", "scales", "biases"))
        if w.dtype not in (torch.uint32, torch.int32) or s.dtype not in (torch.bfloat16, torch.float16) or b.dtype != s.dtype:
            raise ValueError(f"{prefix}: expected packed uint32 words with 16-bit float scales / biases, found "
                             f"{w.dtype}/{s.dtype}/{b.dtype}")
        if tuple(w.shape) != (count, out_dim, in_dim * bits // 32) or tuple(s.shape) != (count, out_dim, in_dim // group_size) \
                or tuple(b.shape) != tuple(s.shape):
            raise ValueError(f"{prefix}: shapes {tuple(w.shape)}, {tuple(s.shape)}, {tuple(b.shape)} do not describe {count} "
                             f"[{out_dim}, {in_dim}] expert matrices in {bits}-bit groups of {group_size}")
        return SimpleNamespace(weight=w.view(torch.int32).to(device), scales=s.to(device), biases=b.to(device),
                               group_size=group_size, bits=bits)

    hs, inter = config["hidden_size"], config["intermediate_size"]
    hq, hkv, hd = config["num_attention_heads"], config["num_key_value_heads"], config["head_dim"]
    dims = {"q_proj": (hq * hd, hs), "k_proj": (hkv * hd, hs), "v_proj": (hkv * hd, hs), "o_proj": (hs, hq * hd),
            "gate_proj": (inter, hs), "up_proj": (inter, hs), "down_proj": (hs, inter)}
    n_experts = int(config.get("num_experts", 0) or 0)
    if n_experts > 0:
        # Qwen3-MoE (`model_type: qwen3_moe`, e.g. Qwen3-30B-A3B): the fields Qwen3ModelWeek3 reads (reference
        # qwen3_week3.py:208-215,258-272) must be present; the router and the experts are W4 like every other matrix
        for key in ("num_experts_per_tok", "moe_intermediate_size"):
            if key not in config:
                raise ValueError(f"config.json has num_experts={n_experts} but lacks {key!r}")
        config.setdefault("norm_topk_prob", False)
        config.setdefault("decoder_sparse_step", 1)
        config.setdefault("mlp_only_layers", [])
    layers = []
    for i in range(config["num_hidden_layers"]):
        base = f"model.layers.{i}"
        blocks = {}
        sparse = (n_experts > 0 and i not in config["mlp_only_layers"] and (i + 1) % config["decoder_sparse_step"] == 0)
        for block, names in _LINEARS.items():
            if block == "mlp" and sparse:
                moe_inter = config["moe_intermediate_size"]
                blocks[block] = SimpleNamespace(
                    gate=linear(f"{base}.mlp.gate", n_experts, hs),
                    switch_mlp=SimpleNamespace(gate_proj=experts(f"{base}.mlp.switch_mlp.gate_proj", n_experts, moe_inter, hs),
                                               up_proj=experts(f"{base}.mlp.switch_mlp.up_proj", n_experts, moe_inter, hs),
                                               down_proj=experts(f"{base}.mlp.switch_mlp.down_proj", n_experts, hs, moe_inter)))
                continue
            blocks[block] = SimpleNamespace(**{n: linear(f"{base}.{block}.{n}", *dims[n]) for n in names})
        blocks["self_attn"].q_norm = norm(f"{base}.self_attn.q_norm.weight", hd)
        blocks["self_attn"].k_norm = norm(f"{base}.self_attn.k_norm.weight", hd)
        layers.append(SimpleNamespace(self_attn=blocks["self_attn"], mlp=blocks["mlp"],
                                      input_layernorm=norm(f"{base}.input_layernorm.weight", hs),
                                      post_attention_layernorm=norm(f"{base}.post_attention_layernorm.weight", hs)))
    inner = SimpleNamespace(embed_tokens=linear("model.embed_tokens", config["vocab_size"], hs), layers=layers,
                            norm=norm("model.norm.weight", hs))
    out = SimpleNamespace(args=SimpleNamespace(**config), model=inner)
    if not config["tie_word_embeddings"]:
        out.lm_head = linear("lm_head", config["vocab_size"], hs)
    elif "lm_head.weight" in src:
        raise ValueError("checkpoint ties the embeddings but also carries lm_head tensors")
    return out


class _Detokenizer:
    """Incremental detokenizer with the mlx_lm surface the loops use: text so far, and the segment added since the last
    read (re-decodes the running id list; fine for a CLI, no hot path goes through it)."""

    def __init__(self, tokenizer_or_decode):
        # mlx_lm builds its detokenizers from the tokenizer itself (reference batch.py:23:
        # ``tokenizer.detokenizer.__class__(tokenizer._tokenizer)``); a bare decode callable is accepted too
        if callable(tokenizer_or_decode) and not hasattr(tokenizer_or_decode, "decode"):
            self._decode = tokenizer_or_decode
        else:
            tok = tokenizer_or_decode
            self._decode = lambda ids: tok.decode(list(ids), skip_special_tokens=False)
        self.reset()

    def reset(self):
        self.tokens: list[int] = []
        self._emitted = 0

    def add_token(self, token: int):
        self.tokens.append(int(token))

    def finalize(self):
        pass

    @property
    def text(self) -> str:
        return self._decode(self.tokens)

    @property
    def last_segment(self) -> str:
        text = self.text
        if text.endswith("\ufffd"):  # an incomplete multi-byte character: wait for its remaining tokens
            return ""
        segment = text[self._emitted:]
        self._emitted = len(text)
        return segment


class TokenizerWrapper:
    """The tokenizer surface of ``mlx_lm.tokenizer_utils.TokenizerWrapper`` that the reference touches."""

    def __init__(self, hf_tokenizer, eos_token_ids=None):
        self._tok = hf_tokenizer
        self._tokenizer = hf_tokenizer  # the attribute name mlx_lm's wrapper exposes (read by reference batch.py:23)
        ids = eos_token_ids if eos_token_ids is not None else [hf_tokenizer.eos_token_id]
        self.eos_token_ids = {int(t) for t in (ids if isinstance(ids, (list, tuple, set)) else [ids]) if t is not None}
        self._detok = _Detokenizer(hf_tokenizer)

    @property
    def eos_token_id(self):
        return self._tok.eos_token_id

    @property
    def vocab_size(self) -> int:
        return int(self._tok.vocab_size)

    def __getattr__(self, name):
        # like mlx_lm's wrapper, anything else is answered by the wrapped Hugging Face tokenizer
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._tok, name)

    def encode(self, text: str, add_special_tokens: bool = False) -> list[int]:
        return list(self._tok.encode(text, add_special_tokens=add_special_tokens))

    def decode(self, ids, **kwargs) -> str:
        return self._tok.decode(list(ids), **kwargs)

    def get_vocab(self) -> dict:
        return dict(self._tok.get_vocab())

    def apply_chat_template(self, messages, **kwargs):
        return self._tok.apply_chat_template(messages, **kwargs)

    @property
    def detokenizer(self) -> _Detokenizer:
        return self._detok


def load(name_or_path: str, device: str = "cuda"):
    """(model, tokenizer) like ``mlx_lm.load``; see the module docstring."""
    model_dir = resolve_model_dir(name_or_path)
    model = load_weights(model_dir, device=device)
    from transformers import AutoTokenizer

    hf_tok = AutoTokenizer.from_pretrained(str(model_dir), local_files_only=True)
    eos = None
    gen_cfg = model_dir / "generation_config.json"
    if gen_cfg.is_file():
        eos = json.loads(gen_cfg.read_text()).get("eos_token_id")
    return model, TokenizerWrapper(hf_tok, eos)
