#!/usr/bin/env python3
"""Hugging Face Qwen3 checkpoint (bf16 / fp16 / fp32 safetensors, as `transformers` writes it) -> the MLX-format 4-bit directory the
reference loads through `mlx_lm.load` and this repository through `tiny_llm_hip.loader.load` -- what `mlx_lm.convert -q --q-bits 4
--q-group-size 128` produces on a Mac, for a machine that has no MLX.

    python tools/convert_hf_to_mlx4bit.py <hf checkpoint dir> <output dir> [--group-size 128] [--shards N]

Every `nn.Linear` and the embedding table are quantised (group-wise affine 4-bit in MLX's packing: `tiny_llm_hip.synthetic.quantize`,
a restatement of `mx.quantize`; reference quantize.py:103-121 reads exactly these triples); RMSNorm weights are copied in bf16; tensor
names stay transformers' own, plus `.scales` / `.biases`; config.json gains `{"quantization": {"group_size": 128, "bits": 4}}`;
tokenizer and generation files are copied.  Dense Qwen3 only (a Qwen3-MoE export stacks its experts under `switch_mlp`: not done here).
Host-only; no network."""
from __future__ import annotations

import argparse
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    sys.path.insert(0, str(p))

QUANTIZED_SUFFIXES = ("_proj.weight", "embed_tokens.weight", "lm_head.weight")


def _tensors(model_dir: Path):
    """(name, tensor) of every tensor of a single-file or sharded safetensors checkpoint, in the files' own order."""
    from safetensors import safe_open

    index = model_dir / "model.safetensors.index.json"
    files = sorted(set(json.loads(index.read_text())["weight_map"].values())) if index.is_file() else ["model.safetensors"]
    for name in files:
        with safe_open(str(model_dir / name), framework="pt") as f:
            for key in f.keys():
                yield key, f.get_tensor(key)


def convert(src: Path, dst: Path, group_size: int = 128, shards: int = 1) -> dict:
    import torch
    from safetensors.torch import save_file
    from tiny_llm_hip.synthetic import quantize

    src, dst = Path(src), Path(dst)
    config = json.loads((src / "config.json").read_text())
    if config.get("num_experts"):
        raise ValueError("Qwen3-MoE checkpoints are not converted here (their experts are stacked under switch_mlp by mlx_lm)")
    if config.get("quantization") or config.get("quantization_config"):
        raise ValueError("the source checkpoint is already quantized")
    dst.mkdir(parents=True, exist_ok=True)
    out, report = {}, {"quantized": [], "copied": [], "skipped": []}
    for name, t in _tensors(src):
        if name == "lm_head.weight" and config.get("tie_word_embeddings", True):
            report["skipped"].append(name)  # tied: the embedding table is the head (reference qwen3_week3.py:314-318)
            continue
        if t.dim() == 2 and name.endswith(QUANTIZED_SUFFIXES):
            if t.shape[1] % group_size:
                raise ValueError(f"{name}: {t.shape[1]} input features are not a multiple of the group size {group_size}")
            words, scales, biases = quantize(t.to(torch.bfloat16), group_size=group_size, bits=4)
            base = name[: -len(".weight")]
            out[name] = words.contiguous().view(torch.uint32)
            out[base + ".scales"], out[base + ".biases"] = scales.contiguous(), biases.contiguous()
            report["quantized"].append(name)
        else:
            out[name] = t.to(torch.bfloat16).contiguous() if t.is_floating_point() else t.contiguous()
            report["copied"].append(name)
    config = dict(config, quantization={"group_size": group_size, "bits": 4})
    rp = config.get("rope_parameters")
    if isinstance(rp, dict) and rp.get("rope_type", "default") not in ("default", None):
        raise ValueError(f"rope_parameters.rope_type={rp.get('rope_type')!r}: the loader (and the reference, qwen3_week3.py:230) compute plain RoPE only")
    rs = config.get("rope_scaling")
    if isinstance(rs, dict) and (rs.get("rope_type") or rs.get("type") or "default") != "default":
        raise ValueError(f"rope_scaling={rs!r}: the loader (and the reference) compute plain RoPE only")
    if "rope_theta" not in config and isinstance(rp, dict) and "rope_theta" in rp:
        config["rope_theta"] = rp["rope_theta"]  # transformers >= 5 nests it; mlx_lm's ModelArgs (the reference's loader) reads the top-level key
    (dst / "config.json").write_text(json.dumps(config, indent=1))
    if shards <= 1:
        save_file(out, str(dst / "model.safetensors"), metadata={"format": "mlx"})
    else:
        names, weight_map = sorted(out), {}
        for s in range(shards):
            part = {n: out[n] for n in names[s::shards]}
            fname = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
            save_file(part, str(dst / fname), metadata={"format": "mlx"})
            weight_map.update({n: fname for n in part})
        (dst / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": weight_map}))
    for extra in src.iterdir():  # tokenizer, chat template, generation config
        if extra.is_file() and extra.suffix in (".json", ".txt", ".model", ".jinja") and extra.name not in ("config.json", "model.safetensors.index.json"):
            shutil.copy2(extra, dst / extra.name)
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src", type=Path)
    ap.add_argument("dst", type=Path)
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--shards", type=int, default=1)
    args = ap.parse_args()
    report = convert(args.src, args.dst, args.group_size, args.shards)
    print(f"quantized {len(report['quantized'])} matrices, copied {len(report['copied'])} tensors, skipped {report['skipped']} -> {args.dst}")


if __name__ == "__main__":
    main()
