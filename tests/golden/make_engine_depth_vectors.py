"""Generates tests/golden/engine_depth_vectors.npz: the float64 truth (oracle.TruthQwen3) and the bf16 oracle's distance from it
(oracle.OracleQwen3) for decode steps behind a 4,160-token prompt on a model with the Qwen3-4B layer shapes at its FULL DEPTH of 36 layers
(vocabulary cut to 8,192 so that the committed logits stay small).

Why: the 36-layer engine was held against the truth only up to 2,500 tokens on 2 layers and at the bench's 128-token prompt
(round-4 review, weak spot 3).  Beyond 4,096 tokens of context a single sequence decodes on the GQA-group plan -- one workgroup per
KV head and 256-token window, walked on the matrix cores (csrc/attn_mfma.h), 32 windows merged by attn_merge_cols_kernel -- behind a
chunked prefill through the paged FlashAttention kernel: the route of BASELINE.json configs[2] / configs[4].  The truth of 4,160 tokens x
36 layers is tens of minutes of host time, so it is committed; the checkpoint is oracle.make_fast_w4_weights (codes drawn directly:
seconds on any host, bit-reproducible), rebuilt and checksummed by tests/test_zz_engine_depth_vs_truth_gpu.py.
Reference assertion mirrored at depth: tests_refsol/test_week_3_day_3.py:386-402 (two paths of one model, log-probabilities).

Run from the repository root:  python tests/golden/make_engine_depth_vectors.py      (~30-60 minutes on 8 cores, < 12 GB)"""

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import tiny_oracle as O  # noqa: E402

CFG = dict(hidden_size=2560, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
           intermediate_size=9728, vocab_size=8192, rope_theta=1000000, rms_norm_eps=1e-6, max_position_embeddings=40960,
           tie_word_embeddings=True)
SEED, SIGMA = 36, 0.02
PROMPT = 4160
STEPS = 3


def checksum(weights) -> np.ndarray:
    mats = [weights["embed"]] + [lw[k] for lw in weights["layers"] for k in ("q", "k", "v", "o", "gate", "up", "down")]
    return np.asarray([np.asarray(m[0], dtype=np.uint32).astype(np.uint64).sum() for m in mats], dtype=np.uint64)


class TruthNoCache(O.TruthQwen3):
    """The float64 truth without its dense-weight cache (36 layers x 101 M weights x 8 bytes would be 29 GB): a matrix is dequantised
    where it is used."""

    def _weight(self, wt):
        packed, s, z = wt
        q = O.unpack_codes(packed).astype(np.float64)
        return q * np.repeat(np.asarray(s, np.float64), 128, axis=-1) + np.repeat(np.asarray(z, np.float64), 128, axis=-1)


def main() -> None:
    t0 = time.time()
    weights = O.make_fast_w4_weights(CFG, seed=SEED, sigma=SIGMA)
    print(f"weights {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(5000 + PROMPT)
    prompt = rng.integers(16, CFG["vocab_size"], size=PROMPT).astype(np.int32)
    truth = TruthNoCache(CFG, weights)
    rows_t = [truth.forward(prompt)[0, -1]]
    print(f"truth prefill {time.time() - t0:.0f} s, max|logit| {np.abs(rows_t[0]).max():.2f}", flush=True)
    fed = []
    for _ in range(STEPS):
        tok = int(np.argmax(rows_t[-1]))
        fed.append(tok)
        rows_t.append(truth.forward([tok])[0, -1])
    print(f"truth steps {time.time() - t0:.0f} s", flush=True)
    del truth
    oracle = O.OracleQwen3(CFG, weights)
    rows_o = [oracle.forward(prompt)[0, -1]]
    print(f"oracle prefill {time.time() - t0:.0f} s", flush=True)
    for tok in fed:
        rows_o.append(oracle.forward([tok])[0, -1])
    rows_t, rows_o = np.stack(rows_t), np.stack(rows_o).astype(np.float64)
    e = np.abs(rows_o - rows_t)
    print(f"max|oracle - truth| {e.max():.4f} rms {np.sqrt((e ** 2).mean()):.5f} max|logit| {np.abs(rows_t).max():.2f} "
          f"top-2 margins {[round(float(np.partition(r, -2)[-1] - np.partition(r, -2)[-2]), 3) for r in rows_t]} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(ROOT / "tests" / "golden" / "engine_depth_vectors.npz", checksum=checksum(weights), seed=np.asarray(SEED),
                        sigma=np.asarray(SIGMA), prompt=prompt, fed=np.asarray(fed, dtype=np.int32), truth=rows_t.astype(np.float32),
                        oracle_bits=(np.ascontiguousarray(rows_o, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16))
    print("wrote tests/golden/engine_depth_vectors.npz")


if __name__ == "__main__":
    main()
