#!/usr/bin/env python3
"""Lab (CPU only): top-2 margins of the float64 truth and the bf16 C port's distance from it on PEAKED synthetic checkpoints with a
permuted head (tiny_llm_hip/synthetic.py head_permutation), for a list of (embed_sigma, residual_gain) -- to pick the recipe whose
margin stands 5-50 x above the rounding error at 36 layers.  usage: peaked_recipe_probe.py LAYERS ES:RG [ES:RG ...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    sys.path.insert(0, str(p))
import random

from bench import build_prompt, host_weights
from oracle import c_oracle
from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

layers = int(sys.argv[1])
cfg = dict(QWEN3_CONFIGS["qwen3-4b"], num_hidden_layers=layers)
for spec in sys.argv[2:]:
    es, rg = (float(x) for x in spec.split(":"))
    t0 = time.time()
    model = synthetic_qwen3(cfg, seed=7, sigma=0.02, device="cpu", embed_sigma=es, residual_gain=rg, head_permutation=(48271, 11))
    w = host_weights(model)
    w["lm_head"] = (model.lm_head.weight.numpy().view(np.uint32), model.lm_head.scales.view(__import__("torch").int16).numpy().view(np.uint16),
                    model.lm_head.biases.view(__import__("torch").int16).numpy().view(np.uint16))
    c2 = dict(cfg, tie_word_embeddings=False)
    prompt = build_prompt(random.Random(4321), 8, cfg["vocab_size"])
    tru = c_oracle.CTruthQwen3(c2, w, max_ctx=32)
    orc = c_oracle.COracleQwen3(c2, w, max_ctx=32)
    for t in prompt:
        tid, tl = tru.step(t)
        _, ol = orc.step(t)
    ids, margins, errs, tops = [tid], [], [], []
    for _ in range(8):
        top2 = np.partition(tl, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        tops.append(float(top2[1]))
        errs.append(float(np.abs(ol.astype(np.float64) - tl).max()))
        tid, tl = tru.step(ids[-1])
        _, ol = orc.step(ids[-1])
        ids.append(tid)
    print(f"es {es} rg {rg} layers {layers}: ids {ids} distinct {len(set(ids))} margins min {min(margins):.3f} max {max(margins):.3f} top {np.mean(tops):.2f} "
          f"oracle err max {max(errs):.4f} ratio {min(margins) / max(errs):.1f} ({time.time() - t0:.0f} s)", flush=True)
    tru.close(); orc.close()
