"""GPU tier (collected last: written after the last GPU run of round 2): the HIP paths against logits produced by the
REFERENCE'S OWN PYTHON SOURCES (tests/golden/reference_code_vectors.npz: /root/reference/src/tiny_llm_ref's Week-2 readable
`kv-cache` checkpoint run on the torch facade of mlx, generator committed beside it).  The reference's readable path and the
kernel paths are different bf16 pipelines over the same W4 checkpoint, so the statement is the one every model-level test
makes, with the reference's own output in the oracle's seat: the HIP path is at most 1.5 x as far from the float64 truth as
the reference's code is (helpers.check_against_truth), teacher-forced on the reference's greedy ids.
"""

from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, check_against_truth, to_mlx_shaped
from oracle import tiny_oracle as O
from test_reference_code_vectors_cpu import CASES, batch_case, batch_truth_and_oracle, from_bits

# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu]
DEVICE = "cuda" if torch.cuda.is_available() else "cpu"  # "cpu" only in the build container's dry run (oracle behind the C ABI)
GOLDEN = Path(__file__).resolve().parent / "golden" / "reference_code_vectors.npz"


def truth_rows(cfg, w, prompt, ids):
    truth = O.TruthQwen3(cfg, w)
    rows = [truth.forward(prompt)[0, -1]]
    for tok in ids[:-1]:
        rows.append(truth.forward([tok])[0, -1])
    return np.stack(rows)


@pytest.mark.parametrize("name", sorted(CASES))
def test_week2_kernel_model_against_the_reference_code(name):
    from tiny_llm_hip import Qwen3ModelWeek2

    golden = np.load(GOLDEN)
    overrides, wseed = CASES[name]
    cfg = dict(TINY_CFG, **overrides)
    w = O.make_qwen3_weights(cfg, seed=wseed, sigma=0.05)
    prompt, ids = golden[f"{name}/prompt"].tolist(), golden[f"{name}/ids"].tolist()
    model = Qwen3ModelWeek2(to_mlx_shaped(cfg, w, device=DEVICE))  # the completed Week-2 model: every HIP kernel
    cache = model.create_kv_cache()
    try:
        logits = model(torch.tensor([prompt], dtype=torch.int32, device=DEVICE), 0, cache, logits_to_keep=1)
        rows, offset = [logits[0, -1].float().cpu().numpy()], len(prompt)
        for tok in ids[:-1]:
            step = model(torch.tensor([[tok]], dtype=torch.int32, device=DEVICE), offset, cache, logits_to_keep=1)
            rows.append(step[0, -1].float().cpu().numpy())
            offset += 1
    finally:
        for layer_cache in cache:
            layer_cache.release()
    check_against_truth(np.stack(rows), from_bits(golden[f"{name}/week2_kv_cache_step_logits"]), truth_rows(cfg, w, prompt, ids),
                        what=f"Week-2 kernel model vs the reference's own readable path, {name}")


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n.startswith("tiny_")])  # the checkpoint every engine test of the round ran on
def test_fused_engine_against_the_reference_code(name):
    from tiny_llm_hip.engine import DecodeEngine

    golden = np.load(GOLDEN)
    overrides, wseed = CASES[name]
    cfg = dict(TINY_CFG, **overrides)
    w = O.make_qwen3_weights(cfg, seed=wseed, sigma=0.05)
    prompt, ids = golden[f"{name}/prompt"].tolist(), golden[f"{name}/ids"].tolist()
    eng = DecodeEngine(to_mlx_shaped(cfg, w), page_size=16, num_pages=32, max_batch=1, max_prefill_rows=256)
    try:
        eng.begin(0)
        eng.prefill(0, prompt, chunk=256)
        rows = [eng.logits(1)[0].float().cpu().numpy()]
        for tok in ids[:-1]:
            eng.set_token(0, tok)  # teacher-forced on the reference's ids
            eng.decode(1, batch=1)
            rows.append(eng.logits(1)[0].float().cpu().numpy())
        eng.release(0)
    finally:
        eng.close()
    check_against_truth(np.stack(rows), from_bits(golden[f"{name}/week2_kv_cache_step_logits"]), truth_rows(cfg, w, prompt, ids),
                        what=f"fused engine vs the reference's own readable path, {name}")


def test_fused_engine_batched_decode_against_the_reference_code():
    """Four sequences decoded together (the batched GEMV / skinny-matmul and batched attention path), teacher-forced on the ids
    of the reference's own readable model decoding the same four requests on its BatchingKvCache."""
    from tiny_llm_hip.engine import DecodeEngine

    golden = np.load(GOLDEN)
    prompts, ids, first, steps = batch_case(golden)
    truth, _ = batch_truth_and_oracle(prompts, ids)
    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    eng = DecodeEngine(to_mlx_shaped(TINY_CFG, w), page_size=16, num_pages=64, max_batch=4, max_prefill_rows=128)
    try:
        for slot, prompt in enumerate(prompts):
            eng.begin(slot)
            eng.prefill(slot, prompt, chunk=128)
        rows = []
        for step in range(steps.shape[0]):
            for slot in range(4):
                eng.set_token(slot, int(ids[step, slot]))
            eng.decode(1, batch=4)
            rows.append(eng.logits(4).float().cpu().numpy())
        for slot in range(4):
            eng.release(slot)
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()
    got = np.stack(rows)  # [steps, 4, vocab]
    check_against_truth(got.reshape(-1, got.shape[-1]), steps.reshape(-1, steps.shape[-1]), truth.reshape(-1, truth.shape[-1]),
                        what="fused engine, 4 sequences per step, vs the reference's own readable path")
