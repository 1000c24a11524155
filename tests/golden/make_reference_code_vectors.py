"""Generates tests/golden/reference_code_vectors.npz: outputs of the REFERENCE'S OWN PYTHON SOURCES
(/root/reference/src/tiny_llm_ref, imported as they are) on small seeded Qwen3-shaped W4 checkpoints, for the tests that cannot
see /root/reference (the GPU box).

How the reference's code runs here: `mlx.core` / `mlx.nn` / `mlx_lm` resolve to the torch facade of this repository
(tiny-llm_amd/compat) -- so the ARITHMETIC is torch's, the CODE (wiring, dtypes, rounding points, masks, cache handling) is the
reference's.  Only paths that need no Metal extension are used: `Qwen3ModelWeek1` (dense bf16 weights, no KV cache) and
`Qwen3ModelWeek2(checkpoint="kv-cache")` (the course's readable pre-kernel path with its KV cache: reference main.py:48-60 allows
exactly this checkpoint on `--device cpu`).  Greedy decode, the ids are stored with the logits so that consumers can
teacher-force.  The checkpoints come from oracle.make_qwen3_weights (numpy, seeded): consumers rebuild them bit-identically.
Generated with ONE torch thread (the summation order of the CPU matmul, and with it the last bit of a bf16 logit, depends on the
thread count); consumers that want bit equality pin one thread too.

    python tests/golden/make_reference_code_vectors.py        (build container only: needs /root/reference)
"""

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REFERENCE = Path("/root/reference")
sys.path[:0] = [str(REFERENCE / "src"), str(REFERENCE), str(ROOT / "tiny-llm_amd" / "compat"), str(ROOT / "tiny-llm_amd"),
                str(ROOT / "tiny-llm_amd" / "extensions_hip"), str(ROOT), str(ROOT / "tests")]

import mlx.core as mx  # noqa: E402  (the facade)
import tiny_llm_ref as R  # noqa: E402  (the reference's sources)

from helpers import TINY_CFG, to_mlx_shaped  # noqa: E402
from oracle import tiny_oracle as O  # noqa: E402

assert Path(R.__file__).is_relative_to(REFERENCE), R.__file__

CASES = {  # name -> (config overrides, weight seed, prompt length, prompt seed, greedy steps after the prefill)
    "tiny_p5": (dict(), 3, 5, 5, 6),
    "tiny_p37": (dict(), 3, 37, 37, 6),
    "tiny_p150": (dict(), 3, 150, 150, 6),
    "untied_gqa3_p23": (dict(hidden_size=384, num_attention_heads=3, num_key_value_heads=1, intermediate_size=640, num_hidden_layers=3,
                             tie_word_embeddings=False), 12, 23, 23, 6),
}


BATCH_PROMPT_LENS, BATCH_STEPS = (7, 33, 12, 60), 4


def bits(t):
    assert t.dtype == mx.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def main() -> None:
    torch.set_num_threads(1)  # the fp32 summation order of torch's CPU matmul depends on the thread count: one thread = one order
    out = {}
    with mx.stream(mx.cpu):
        for name, (overrides, wseed, n, pseed, steps) in CASES.items():
            cfg = dict(TINY_CFG, **overrides)
            model = to_mlx_shaped(cfg, O.make_qwen3_weights(cfg, seed=wseed, sigma=0.05), device="cpu")
            prompt = [int(t) for t in np.random.default_rng(pseed).integers(1, cfg["vocab_size"], size=n)]
            week2 = R.Qwen3ModelWeek2(model, checkpoint="kv-cache")
            cache = week2.create_kv_cache()
            logits = week2(mx.array([prompt], dtype=mx.int32), 0, cache)  # every prompt position
            rows, ids, offset = [logits[0, -1]], [int(logits[0, -1].argmax())], n
            for _ in range(steps):
                step = week2(mx.array([[ids[-1]]], dtype=mx.int32), offset, cache, logits_to_keep=1)
                rows.append(step[0, -1])
                ids.append(int(step[0, -1].argmax()))
                offset += 1
            for layer_cache in cache:
                layer_cache.release()
            out[f"{name}/prompt"] = np.asarray(prompt, dtype=np.int32)
            out[f"{name}/ids"] = np.asarray(ids, dtype=np.int32)
            # logits are bfloat16: stored as their 16-bit patterns (exact, half the size)
            out[f"{name}/week2_kv_cache_prefill_logits_last8"] = bits(logits[0, -8:])  # the last (up to) 8 prompt positions
            out[f"{name}/week2_kv_cache_step_logits"] = bits(mx.stack(rows))  # [1 + steps, vocab]: prefill's last row, then each step
            if n <= 40:
                out[f"{name}/week1_logits_last8"] = bits(R.Qwen3ModelWeek1(model)(mx.array([prompt], dtype=mx.int32))[0, -8:])
        # continuous batching on the readable path: four requests prefilled one by one, then decoded TOGETHER on a
        # BatchingKvCache (reference kv_cache.py BatchingKvCache; the shape of tests_refsol/test_week_3_day_1.py:128-195)
        cfg = dict(TINY_CFG)
        model = to_mlx_shaped(cfg, O.make_qwen3_weights(cfg, seed=3, sigma=0.05), device="cpu")
        week2 = R.Qwen3ModelWeek2(model, checkpoint="kv-cache")
        lens = BATCH_PROMPT_LENS
        prompts = [[int(t) for t in np.random.default_rng(300 + n).integers(1, cfg["vocab_size"], size=n)] for n in lens]
        batch = [R.BatchingKvCache(max_active_requests=len(lens), max_seq_len=128) for _ in range(week2.num_hidden_layers)]
        first_rows, tokens = [], []
        for rid, prompt in enumerate(prompts):
            own = week2.create_kv_cache()
            logits = week2(mx.array([prompt], dtype=mx.int32), 0, own, logits_to_keep=1)
            first_rows.append(logits[0, -1])
            tokens.append(int(logits[0, -1].argmax()))
            for layer_batch, layer_own in zip(batch, own):
                layer_batch.add_request(layer_own, rid)
        step_rows, step_ids, offsets = [], [list(tokens)], list(lens)
        for _ in range(BATCH_STEPS):
            logits = week2(mx.array(tokens, dtype=mx.int32).reshape(-1, 1), mx.array(offsets, dtype=mx.int32), batch, logits_to_keep=1)
            step_rows.append(logits[:, -1])
            tokens = [int(logits[i, -1].argmax()) for i in range(len(lens))]
            step_ids.append(list(tokens))
            offsets = [o + 1 for o in offsets]
        for i, prompt in enumerate(prompts):
            out[f"batch4/prompt{i}"] = np.asarray(prompt, dtype=np.int32)
        out["batch4/ids"] = np.asarray(step_ids, dtype=np.int32)  # [1 + steps, 4]: the token fed at each step, then the last greedy ids
        out["batch4/prefill_last_logits"] = bits(mx.stack(first_rows))  # [4, vocab]
        out["batch4/step_logits"] = bits(mx.stack(step_rows))  # [steps, 4, vocab]
    np.savez_compressed(HERE / "reference_code_vectors.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
