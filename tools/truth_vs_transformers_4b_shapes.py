#!/usr/bin/env python3
"""One-off evidence (too slow for the test tier: ~3 min, ~15 GB of host memory): the float64 ground truth (oracle.TruthQwen3)
against Hugging Face transformers' Qwen3ForCausalLM in float64 at the REAL Qwen3-4B shapes (hidden 2560, 32 / 8 heads of 128,
intermediate 9728, vocabulary 151,936, rope theta 1e6), two layers, random W4 weights: prefill logits of every position and
KV-cached decode steps.  tests/test_truth_vs_transformers_cpu.py holds the same statement at small shapes.

    python tools/truth_vs_transformers_4b_shapes.py > profiles/r02_labs/truth_vs_transformers_qwen3_4b_shapes.txt
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch
import transformers

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tiny-llm_amd"), str(ROOT / "tiny-llm_amd" / "extensions_hip")]
from oracle import tiny_oracle as O  # noqa: E402
from test_truth_vs_transformers_cpu import attention_and_norm_tensors, dense64, hf_common, load_exactly  # noqa: E402
from tiny_llm_hip.synthetic import QWEN3_CONFIGS  # noqa: E402

cfg = dict(QWEN3_CONFIGS["qwen3-4b"], num_hidden_layers=2)
t0 = time.time()
w = O.make_qwen3_weights(cfg, seed=1, sigma=0.02)
hf_cfg = transformers.Qwen3Config(**hf_common(cfg))
hf_cfg._attn_implementation = "eager"
reference = transformers.Qwen3ForCausalLM(hf_cfg).double().eval()
tensors = attention_and_norm_tensors(w)
for i, lw in enumerate(w["layers"]):
    for name, key in (("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
        tensors[f"model.layers.{i}.{name}.weight"] = dense64(lw[key])
load_exactly(reference, tensors)
del tensors
print(f"# transformers {transformers.__version__}, torch {torch.__version__}; config {cfg}")
print(f"# weights + model built in {time.time() - t0:.0f} s")
prompt = [int(t) for t in np.random.default_rng(2).integers(1, cfg["vocab_size"], size=24)]
truth, oracle = O.TruthQwen3(cfg, w), O.OracleQwen3(cfg, w)
with torch.no_grad():
    out = reference(torch.tensor([prompt]), use_cache=True)
want = truth.forward(prompt, logits_to_keep=None)[0]
print(f"prefill, {len(prompt)} positions: max |truth - transformers| = {np.abs(out.logits[0].numpy() - want).max():.3e}  (max |logit| {np.abs(want).max():.2f})")
print(f"  for scale, last position: max |bf16 oracle - truth| = {np.abs(oracle.forward(prompt)[0, -1] - want[-1]).max():.3e}")
past, tok = out.past_key_values, int(np.argmax(want[-1]))
for step in range(4):
    with torch.no_grad():
        out = reference(torch.tensor([[tok]]), past_key_values=past, use_cache=True)
    past = out.past_key_values
    row = truth.forward([tok])[0, -1]
    print(f"decode step {step}: max |truth - transformers| = {np.abs(out.logits[0, -1].numpy() - row).max():.3e}; "
          f"max |bf16 oracle - truth| = {np.abs(oracle.forward([tok])[0, -1] - row).max():.3e}; same greedy id: "
          f"{int(np.argmax(row)) == int(out.logits[0, -1].argmax())}")
    tok = int(np.argmax(row))
