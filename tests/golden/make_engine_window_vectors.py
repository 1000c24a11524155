"""Generates tests/golden/engine_window_vectors.npz: the float64 truth (oracle.TruthQwen3) and the bf16 oracle's distance from it
(oracle.OracleQwen3) for decode steps behind prompts of 100 / 130 / 300 / 1,200 / 2,500 tokens on a model with Qwen3-4B LAYER shapes
(hidden 2560, 32 + 8 heads of 128, intermediate 9728; 2 layers; vocabulary cut to 8,192 so that the committed logits stay small).

Why a committed fixture: these contexts put the fused engine on 2 / 4 / 4 / 8 / 16 attention windows (130 tokens: the four 64-token windows of bench.py's own
128-token prompt) -- the plans in which the wo GEMV
merges the split partials itself (qmv3.h PRO_ATTN_MERGE, NS = 2 / 4 / 8) and the column-parallel merge launch (16) -- and the
truth of a 2,500-token prompt costs minutes of host time per run on the GPU box.  The weights come from numpy's seeded generator
(oracle.make_qwen3_weights: bit-reproducible on any host), so tests/test_zz_engine_windows_vs_truth_gpu.py rebuilds the same
checkpoint on the device box, checks the committed checksum of its packed words, and compares the engine's logits with the truth
stored here through helpers.check_against_truth (max|HIP - truth| <= 1.5 max|oracle - truth| + one bf16 step).

Run from the repository root:  python tests/golden/make_engine_window_vectors.py
(about 10 minutes on 8 cores; writes tests/golden/engine_window_vectors.npz)."""

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import tiny_oracle as O  # noqa: E402

CFG = dict(hidden_size=2560, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
           intermediate_size=9728, vocab_size=8192, rope_theta=1000000, rms_norm_eps=1e-6, max_position_embeddings=40960,
           tie_word_embeddings=True)
SEED, SIGMA = 31, 0.02
PROMPTS = (100, 130, 300, 1200, 2500)  # 2 / 4 (64-token) / 4 (128-token) / 8 / 16 attention windows in the decode steps behind them
STEPS = 4  # decode steps fed with the truth's own greedy ids; logits rows kept = STEPS + 1 (the row behind the prompt first)


def checksum(weights) -> np.ndarray:
    """One uint64 per W4 matrix in a fixed order: the wrapped sum of its packed words (any changed nibble changes it)."""
    mats = [weights["embed"]] + [lw[k] for lw in weights["layers"] for k in ("q", "k", "v", "o", "gate", "up", "down")]
    return np.asarray([np.asarray(m[0], dtype=np.uint32).astype(np.uint64).sum() for m in mats], dtype=np.uint64)


def main() -> None:
    t0 = time.time()
    weights = O.make_qwen3_weights(CFG, seed=SEED, sigma=SIGMA)
    print(f"weights {time.time() - t0:.0f} s", flush=True)
    out = {"checksum": checksum(weights), "prompts": np.asarray(PROMPTS), "steps": np.asarray(STEPS),
           "seed": np.asarray(SEED), "sigma": np.asarray(SIGMA)}
    for n in PROMPTS:
        rng = np.random.default_rng(1000 + n)
        prompt = rng.integers(16, CFG["vocab_size"], size=n).astype(np.int32)
        truth, oracle = O.TruthQwen3(CFG, weights), O.OracleQwen3(CFG, weights)
        rows_t = [truth.forward(prompt)[0, -1]]
        rows_o = [oracle.forward(prompt)[0, -1]]
        fed = []
        for _ in range(STEPS):
            tok = int(np.argmax(rows_t[-1]))
            fed.append(tok)
            rows_t.append(truth.forward([tok])[0, -1])
            rows_o.append(oracle.forward([tok])[0, -1])
        rows_t, rows_o = np.stack(rows_t), np.stack(rows_o).astype(np.float64)
        out[f"prompt_{n}"] = prompt
        out[f"fed_{n}"] = np.asarray(fed, dtype=np.int32)
        out[f"truth_{n}"] = rows_t.astype(np.float32)  # float32 keeps 2^-24 relative: far below the bf16 errors measured against it
        out[f"oracle_bits_{n}"] = (np.ascontiguousarray(rows_o, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        e = np.abs(rows_o - rows_t)
        print(f"prompt {n}: max|oracle - truth| {e.max():.4f} rms {np.sqrt((e ** 2).mean()):.5f} max|logit| {np.abs(rows_t).max():.2f} "
              f"top-2 margins {[round(float(np.partition(r, -2)[-1] - np.partition(r, -2)[-2]), 3) for r in rows_t]} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(ROOT / "tests" / "golden" / "engine_window_vectors.npz", **out)
    print("wrote tests/golden/engine_window_vectors.npz")


if __name__ == "__main__":
    main()
