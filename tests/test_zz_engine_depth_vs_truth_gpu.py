"""GPU tier: the fused engine at the FULL DEPTH of Qwen3-4B (36 layers at its layer shapes) behind a 4,160-token prompt, against a
committed float64 truth (tests/golden/engine_depth_vectors.npz + its generator make_engine_depth_vectors.py).

Beyond 4,096 tokens of context one sequence decodes on the long-context plan of BASELINE.json configs[2] / configs[4]: a workgroup per KV
head and window walking its 192-token window on the matrix cores (csrc/attn_mfma.h), 32 windows merged by attn_merge_cols_kernel, behind a
prefill in 2,048-row chunks through the W4 tile GEMM and the paged FlashAttention kernel.  The kernel-level tests hold those kernels against
the oracle to 32k tokens; the model-level tests stopped at 2,500 tokens on 2 layers (round-4 review).  This one runs all 36 layers.
Bar (helpers.check_against_truth): the engine sits as close to the float64 truth as the bf16 oracle does; the decode rows -- matvec
arithmetic on both sides -- also within 1.10 x the oracle's rms error.  Reference assertion mirrored at depth:
tests_refsol/test_week_3_day_3.py:386-402."""

import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import RMS_FACTOR_DECODE, check_against_truth, log_parity, to_mlx_shaped
from oracle import tiny_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden" / "engine_depth_vectors.npz"
sys.path.insert(0, str(GOLDEN.parent))


def test_36_layer_engine_behind_a_4160_token_prompt_against_the_truth(monkeypatch):
    import make_engine_depth_vectors as G
    from tiny_llm_hip.engine import DecodeEngine

    for name in ("TL_ATTN_RQ", "TL_ATTN_MFMA", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MIN_TOKENS"):
        monkeypatch.delenv(name, raising=False)
    vec = np.load(GOLDEN)
    weights = O.make_fast_w4_weights(G.CFG, seed=int(vec["seed"]), sigma=float(vec["sigma"]))
    assert np.array_equal(G.checksum(weights), vec["checksum"]), "the rebuilt checkpoint is not the one the fixture was generated on"
    model = to_mlx_shaped(G.CFG, weights)
    prompt, fed = [int(t) for t in vec["prompt"]], [int(t) for t in vec["fed"]]
    truth = vec["truth"].astype(np.float64)
    oracle = O.from_bf16_bits(vec["oracle_bits"]).astype(np.float64)
    eng = DecodeEngine(model, page_size=128, num_pages=(len(prompt) + len(fed) + 8 + 127) // 128 + 2, max_batch=1, max_prefill_rows=2048)
    try:
        eng.begin(0)
        eng.prefill(0, prompt, chunk=2048)
        got = [eng.logits(1)[0].float().cpu().numpy()]
        for tok in fed:  # teacher-forced on the truth's greedy ids
            eng.set_token(0, tok)
            eng.decode(1, batch=1)
            got.append(eng.logits(1)[0].float().cpu().numpy())
        st = eng.stats()
        prof = eng.profile_step(1)
        eng.release(0)
    finally:
        eng.close()
    layers = G.CFG["num_hidden_layers"]
    what = f"Qwen3-4B layer shapes x {layers} layers, prompt {len(prompt)}: {prof['n_splits']} windows"
    assert prof["n_splits"] == 32 and prof["kinds"]["attention_merge"]["launches"] == layers, f"{what}: not the long-context plan ({prof})"
    assert st["graph_replays"] >= len(fed) - 1
    got = np.stack(got)
    check_against_truth(got[:1], oracle[:1], truth[:1], what=what + " [prefill row]")
    rec = check_against_truth(got[1:], oracle[1:], truth[1:], what=what + " [decode rows]", rms_factor=RMS_FACTOR_DECODE)
    log_parity({"what": "engine_depth_vs_truth", "prompt": len(prompt), "layers": layers, **rec})
