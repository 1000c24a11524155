"""GPU tier (collected last: written after the last GPU run of round 2): the fused decode engine against a THIRD-PARTY
implementation directly -- Hugging Face transformers' Qwen3ForCausalLM in float64 on the host, same dequantised weights --
without going through the builder-written float64 truth (which agrees with transformers to 2e-6,
tests/test_truth_vs_transformers_cpu.py).  Statement as in every model-level test: the HIP path is at most 1.5 x as far from
the reference as the bf16 restatement of the reference's pipeline is (helpers.check_against_truth)."""

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, check_against_truth, to_mlx_shaped
from oracle import tiny_oracle as O

transformers = pytest.importorskip("transformers")
# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu]


def test_engine_prefill_and_decode_against_transformers_float64():
    from test_truth_vs_transformers_cpu import attention_and_norm_tensors, dense64, hf_common, load_exactly
    from tiny_llm_hip.engine import DecodeEngine

    cfg = dict(TINY_CFG)
    w = O.make_qwen3_weights(cfg, seed=3, sigma=0.05)
    hf_cfg = transformers.Qwen3Config(**hf_common(cfg))
    hf_cfg._attn_implementation = "eager"
    reference = transformers.Qwen3ForCausalLM(hf_cfg).double().eval()
    tensors = attention_and_norm_tensors(w)
    for i, lw in enumerate(w["layers"]):
        for name, key in (("mlp.gate_proj", "gate"), ("mlp.up_proj", "up"), ("mlp.down_proj", "down")):
            tensors[f"model.layers.{i}.{name}.weight"] = dense64(lw[key])
    load_exactly(reference, tensors)

    prompt = [int(t) for t in np.random.default_rng(41).integers(1, cfg["vocab_size"], size=37)]
    steps = 8
    eng = DecodeEngine(to_mlx_shaped(cfg, w), page_size=16, num_pages=32, max_batch=1, max_prefill_rows=64)
    try:
        eng.begin(0)
        eng.prefill(0, prompt, chunk=64)
        rows = [eng.logits(1)[0].float().cpu().numpy()]
        ids = [eng.read_pending(1)[0]]
        for _ in range(steps - 1):
            eng.decode(1, batch=1)
            rows.append(eng.logits(1)[0].float().cpu().numpy())
            ids.append(eng.read_pending(1)[0])
        eng.release(0)
    finally:
        eng.close()

    oracle = O.OracleQwen3(cfg, w)
    with torch.no_grad():
        out = reference(torch.tensor([prompt]), use_cache=True)
    want = [out.logits[0, -1].numpy()]
    bf16 = [oracle.forward(prompt)[0, -1]]
    past = out.past_key_values
    for tok in ids[:-1]:  # teacher-forced on the ids the engine produced
        with torch.no_grad():
            out = reference(torch.tensor([[int(tok)]]), past_key_values=past, use_cache=True)
        past = out.past_key_values
        want.append(out.logits[0, -1].numpy())
        bf16.append(oracle.forward([int(tok)])[0, -1])
    check_against_truth(np.stack(rows), np.stack(bf16), np.stack(want), what="fused engine vs transformers Qwen3 (float64)")
