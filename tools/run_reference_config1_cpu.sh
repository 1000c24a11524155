#!/bin/bash
# BASELINE config 1 ("Qwen3-0.6B Week-1 greedy decode on mlx.core CPU stream: plumbing, no GPU") with the reference's OWN
# benches/bench.py, unmodified, through the import facade: Qwen3-0.6B-SHAPED synthetic W4 checkpoint (hidden 1024, 28 layers,
# vocab 151,936) in a throw-away Hugging Face cache, `--loader week1 --device cpu` (the course's readable model: dense bf16
# weights, no KV cache, plain torch ops on host tensors -- no extension kernel, the oracle is not involved) and
# `--solution mlx --device cpu` (the facade's mlx_lm model, fp32 torch).  Output: profiles/r02_labs/reference_bench_config1_through_facade.txt
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
cfg = dict(QWEN3_CONFIGS["qwen3-0.6b"])
tree = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cpu")
t = lambda l: (l.weight.numpy().view(np.uint32), l.scales.float().numpy(), l.biases.float().numpy())
n = lambda l: l.weight.float().numpy()
layers = [dict(q=t(L.self_attn.q_proj), k=t(L.self_attn.k_proj), v=t(L.self_attn.v_proj), o=t(L.self_attn.o_proj),
               gate=t(L.mlp.gate_proj), up=t(L.mlp.up_proj), down=t(L.mlp.down_proj), q_norm=n(L.self_attn.q_norm),
               k_norm=n(L.self_attn.k_norm), input_norm=n(L.input_layernorm), post_norm=n(L.post_attention_layernorm))
          for L in tree.model.layers]
w = dict(embed=t(tree.model.embed_tokens), layers=layers, norm=n(tree.model.norm))
write_hf_cache_snapshot(Path(os.environ["HF_HOME"]), "Qwen/Qwen3-0.6B-MLX-4bit", cfg, w, vocab_words=[f"w{i}" for i in range(2000)])
PY
OUT=profiles/r02_labs/reference_bench_config1_through_facade.txt
C="--model qwen3-0.6b --num-seqs 2 --min-input-len 32 --max-input-len 32 --min-output-len 16 --max-output-len 16 --warmup 1 --device cpu"
{
echo "# /root/reference/benches/bench.py (unmodified) through tiny-llm_amd/compat, host CPU ($(nproc) cores), Qwen3-0.6B-shaped synthetic W4 checkpoint"
for s in "--solution ref --loader week1" "--solution ref --loader week2 --week2-checkpoint kv-cache" "--solution mlx"; do
  echo; echo "\$ python benches/bench.py $C $s"
  timeout 1500 python tests/run_reference_script.py benches/bench.py $C $s 2>&1 | grep -v "it/s\]" | tail -8
done
} > $OUT 2>&1
rm -rf "$HF_HOME"
cat $OUT
