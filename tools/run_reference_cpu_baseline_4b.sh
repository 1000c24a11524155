#!/bin/bash
# The reference's OWN code on the host CPU at the Qwen3-4B shape (BASELINE north_star: "the MLX CPU path timed on the host cores
# as the reported baseline" -- MLX itself cannot be installed here; this is the closest thing that can run): the reference's
# unmodified benches/bench.py driving the reference's unmodified tiny_llm_ref sources (REFSOL_REFERENCE_SOURCES=1), `mlx` being the
# torch facade, on the only checkpoint the reference allows on `--device cpu`: Week 2 `kv-cache` (dense bf16 weights dequantised
# from the W4 checkpoint, readable operators, KV cache; reference benches/bench.py:158-169).  No extension kernel and no oracle
# take part.  Synthetic Qwen3-4B-shaped W4 checkpoint, reference acceptance shape scaled down (prompt 128, 17 new tokens).
# Output: profiles/r02_labs/reference_cpu_baseline_qwen3_4b.txt      (~10 min, ~25 GB of host memory)
set -u
cd "$(dirname "$0")/.."
export HF_HOME=$(mktemp -d) HF_HUB_OFFLINE=1
python - <<'PY'
import os, sys
sys.path[:0] = [".", "tests", "tiny-llm_amd", "tiny-llm_amd/extensions_hip"]
from pathlib import Path
import numpy as np, torch
from checkpoint_fixture import write_hf_cache_snapshot
from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3
cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
tree = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cpu")
t = lambda l: (l.weight.numpy().view(np.uint32), l.scales.float().numpy(), l.biases.float().numpy())
n = lambda l: l.weight.float().numpy()
layers = [dict(q=t(L.self_attn.q_proj), k=t(L.self_attn.k_proj), v=t(L.self_attn.v_proj), o=t(L.self_attn.o_proj),
               gate=t(L.mlp.gate_proj), up=t(L.mlp.up_proj), down=t(L.mlp.down_proj), q_norm=n(L.self_attn.q_norm),
               k_norm=n(L.self_attn.k_norm), input_norm=n(L.input_layernorm), post_norm=n(L.post_attention_layernorm))
          for L in tree.model.layers]
w = dict(embed=t(tree.model.embed_tokens), layers=layers, norm=n(tree.model.norm))
write_hf_cache_snapshot(Path(os.environ["HF_HOME"]), "Qwen/Qwen3-4B-MLX-4bit", cfg, w, vocab_words=[f"w{i}" for i in range(2000)])
PY
OUT=profiles/r02_labs/reference_cpu_baseline_qwen3_4b.txt
C="--model qwen3-4b --num-seqs 1 --min-input-len 128 --max-input-len 128 --min-output-len 17 --max-output-len 17 --warmup 1 --seed 0 --device cpu --solution ref --loader week2 --week2-checkpoint kv-cache --prefill-logits last"
{
echo "# /root/reference/benches/bench.py + /root/reference/src/tiny_llm_ref (both unmodified), mlx = torch facade, host CPU: $(nproc) cores, torch threads default"
echo "# Qwen3-4B-shaped synthetic W4 checkpoint; the only CPU-runnable reference path (Week 2 kv-cache: dense bf16 weights, readable ops)"
echo; echo "\$ REFSOL_REFERENCE_SOURCES=1 python benches/bench.py $C"
REFSOL_REFERENCE_SOURCES=1 timeout 3000 python tests/run_reference_script.py benches/bench.py $C 2>&1 | grep -v "it/s\]" | tail -8
} > $OUT 2>&1
rm -rf "$HF_HOME"
cat $OUT
