"""CPU tier: oracle/torch_week2_cpu.py -- the torch-CPU restatement of the reference's Week-2 `kv-cache` path that bench.py times as
`cpu_baseline.torch_week2_kv_cache` -- computes the model: against the float64 truth it sits about as far as
the bf16 numpy oracle does (it keeps the READABLE rounding points: RMSNorm casts before the weight, fp32 attention), on prefill rows
and on KV-cached decode steps, GQA 2:1."""

import numpy as np
import torch

from helpers import TINY_CFG
from oracle import tiny_oracle as O
from oracle.torch_week2_cpu import TorchWeek2KvCacheCPU


def _dense(weights):
    def dq(t):
        packed, s, z = t
        return torch.from_numpy(np.ascontiguousarray(O.dequantize_weights(packed, s, z, dtype="bf16"))).to(torch.bfloat16)

    def nw(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)

    layers = [dict({k: dq(lw[k]) for k in ("q", "k", "v", "o", "gate", "up", "down")},
                   **{k: nw(lw[k]) for k in ("q_norm", "k_norm", "input_norm", "post_norm")}) for lw in weights["layers"]]
    return dict(embed=dq(weights["embed"]), layers=layers, norm=nw(weights["norm"]))


def test_torch_week2_restatement_tracks_the_truth_like_the_bf16_oracle():
    weights = O.make_qwen3_weights(TINY_CFG, seed=5, sigma=0.05)
    model = TorchWeek2KvCacheCPU(TINY_CFG, _dense(weights))
    truth, oracle = O.TruthQwen3(TINY_CFG, weights), O.OracleQwen3(TINY_CFG, weights)
    prompt = [7, 300, 12, 999, 45, 2, 801, 64, 5]
    rows_m, rows_t, rows_o = [model.forward(prompt).float().numpy()], [truth.forward(prompt)[0, -1]], [oracle.forward(prompt)[0, -1]]
    for _ in range(4):
        tok = int(np.argmax(rows_t[-1]))
        rows_m.append(model.forward([tok]).float().numpy())
        rows_t.append(truth.forward([tok])[0, -1])
        rows_o.append(oracle.forward([tok])[0, -1])
    m, t, o = (np.stack(r).astype(np.float64) for r in (rows_m, rows_t, rows_o))
    e_m, e_o = np.abs(m - t).max(), np.abs(o - t).max()
    assert e_m <= 2.5 * e_o + 2.0 ** -6, f"torch restatement {e_m:.4f} from the truth, the bf16 oracle {e_o:.4f}"
    assert model.offset == len(prompt) + 4 and model.k_cache[0].shape[0] == model.offset


def test_timed_decode_is_teacher_forced_when_asked():
    weights = O.make_qwen3_weights(TINY_CFG, seed=6, sigma=0.05)
    a, b = TorchWeek2KvCacheCPU(TINY_CFG, _dense(weights)), TorchWeek2KvCacheCPU(TINY_CFG, _dense(weights))
    la = a.forward([1, 2, 3])
    b.forward([1, 2, 3])
    first = int(torch.argmax(la.float()))
    _, free_ids, l0 = a.timed_decode(first, 3)
    _, fed_ids, l1 = b.timed_decode(first, 3, fed=[first] + free_ids[:2])
    assert free_ids == fed_ids and torch.equal(l0, l1)
