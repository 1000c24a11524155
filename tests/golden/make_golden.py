"""Generates the committed fixtures under tests/golden/.

  reference_literals.json  known-answer literals transcribed (values only) from the reference's own MLX-free
                           assertions, each with the reference file:line it comes from.  These pin the oracle and
                           the host logic without needing MLX.
  torch_vectors.npz        small seeded input/output vectors produced with an INDEPENDENT implementation
                           (PyTorch CPU fp32/fp64 functional ops: F.scaled_dot_product_attention, F.rms_norm, a
                           complex-number RoPE), used to cross-check the numpy oracle.  MLX itself (the reference's
                           oracle, mlx==0.32.0) cannot be imported in this environment, so these are NOT
                           reference outputs; the oracle's parity status is spelled out in oracle/tiny_oracle.py.

Run from the repo root:  python tests/golden/make_golden.py
"""

import json
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
NEG = "-inf"

LITERALS = {
    "causal_mask_3x3": {"source": "tests_refsol/test_week_1_day_3.py:72-92",
                        "value": [[0, NEG, NEG], [0, 0, NEG], [0, 0, 0]]},
    "causal_mask_3x5": {"source": "tests_refsol/test_week_1_day_3.py:95-116",
                        "value": [[0, 0, 0, NEG, NEG], [0, 0, 0, 0, NEG], [0, 0, 0, 0, 0]]},
    "batching_kv_cache": {
        "source": "tests_refsol/test_week_3_day_1.py:52-127",
        "slot0": {"key": [[[[10.0]]]], "value": [[[[110.0]]]]},
        "slot2": {"key": [[[[20.0], [21.0]]]], "value": [[[[120.0], [121.0]]]]},
        "keys": [[[[12.0], [13.0]]], [[[0.0], [0.0]]], [[[22.0], [23.0]]]],
        "values": [[[[112.0], [113.0]]], [[[0.0], [0.0]]], [[[122.0], [123.0]]]],
        "expected_keys": [[[[0.0], [10.0], [12.0], [13.0]]], [[[0.0], [0.0], [0.0], [0.0]]],
                          [[[20.0], [21.0], [22.0], [23.0]]]],
        "expected_values": [[[[0.0], [110.0], [112.0], [113.0]]], [[[0.0], [0.0], [0.0], [0.0]]],
                            [[[120.0], [121.0], [122.0], [123.0]]]],
        "expected_mask": [[[[NEG, 0.0, 0.0, NEG], [NEG, 0.0, 0.0, 0.0]]],
                          [[[NEG, NEG, NEG, NEG], [NEG, NEG, NEG, NEG]]],
                          [[[0.0, 0.0, 0.0, NEG], [0.0, 0.0, 0.0, 0.0]]]],
        "last_batch_bytes": 96, "staging_copy_bytes": 56},
    "paged_pool_growth": {"source": "tests_refsol/test_week_3_day_3.py:238-251", "page_size": 4, "tokens": 17,
                          "num_pages": 5, "capacity": 8, "storage_growths": 2, "copied_pages_on_growth": 4,
                          "copied_bytes_on_growth": 1024, "chunk_shape": [1, 2, 17, 4], "dtype": "float32"},
    "paged_metadata_single": {"source": "tests_refsol/test_week_3_day_4.py:118-149", "page_size": 4,
                              "appends": [3, 3], "block_table": [[0, 1]], "context_lens": [6]},
    "paged_metadata_batch": {"source": "tests_refsol/test_week_3_day_4.py:152-199", "page_size": 4,
                             "prefilled": {"0": 3, "2": 6}, "context_lens": [4, 0, 7], "idle_row": [-1, -1]},
    "packing_order": {"source": "src/tiny_llm_ref/quantize.py:113-115", "word": 0x76543210,
                      "elements": [0, 1, 2, 3, 4, 5, 6, 7]},
}


def sin_fixture(shape, phase):
    n = int(np.prod(shape))
    return np.sin(np.arange(n, dtype=np.float32) * 0.017 + phase).reshape(shape)


def torch_vectors():
    torch.manual_seed(0)
    out = {}
    # grouped attention on the reference's deterministic sin ramps (tests_refsol/test_week_2_day_5.py:127-143)
    for name, (B, Hq, Hkv, L, S, D, causal) in {"attn_gqa4_causal": (2, 8, 2, 8, 129, 64, True),
                                                  "attn_gqa1_plain": (1, 2, 2, 1, 31, 64, False),
                                                  "attn_decode_causal": (2, 4, 1, 1, 256, 128, True)}.items():
        q = torch.from_numpy(sin_fixture((B, Hq, L, D), 0.1))
        k = torch.from_numpy(sin_fixture((B, Hkv, S, D), 0.7))
        v = torch.from_numpy(sin_fixture((B, Hkv, S, D), 1.3))
        mask = None
        if causal:
            mask = torch.tril(torch.ones(L, S, dtype=torch.bool), diagonal=S - L)
        o = F.scaled_dot_product_attention(q.double(), k.double(), v.double(), attn_mask=mask, enable_gqa=True,
                                           scale=D ** -0.5)
        out[name + "_out"] = o.float().numpy()
        out[name + "_shape"] = np.array([B, Hq, Hkv, L, S, D, int(causal)])
    # RMSNorm (single rounding) and RoPE (complex rotation, both pairings)
    x = torch.randn(3, 5, 128)
    w = 1 + 0.1 * torch.randn(128)
    out["rms_x"], out["rms_w"] = x.numpy(), w.numpy()
    out["rms_out"] = F.rms_norm(x.double(), (128,), w.double(), eps=1e-6).float().numpy()
    xr = torch.randn(2, 7, 3, 64)
    offsets = torch.tensor([5, 1000])
    pos = (offsets[:, None] + torch.arange(7)[None, :]).double()
    freq = 1000000.0 ** (-torch.arange(32).double() / 32)
    ang = pos[:, :, None] * freq[None, None, :]
    rot = torch.polar(torch.ones_like(ang), ang)[:, :, None, :]
    half = torch.complex(xr[..., :32].double(), xr[..., 32:].double()) * rot
    out["rope_x"], out["rope_offsets"] = xr.numpy(), offsets.numpy()
    out["rope_out_default"] = torch.cat([half.real, half.imag], dim=-1).float().numpy()
    pair = torch.view_as_complex(xr.double().reshape(2, 7, 3, 32, 2).contiguous()) * rot
    out["rope_out_traditional"] = torch.view_as_real(pair).reshape(2, 7, 3, 64).float().numpy()
    g, u = torch.randn(4, 33) * 3, torch.randn(4, 33)
    out["swiglu_gate"], out["swiglu_up"] = g.numpy(), u.numpy()
    out["swiglu_out"] = (F.silu(g.double()) * u.double()).float().numpy()
    return out


if __name__ == "__main__":
    (HERE / "reference_literals.json").write_text(json.dumps(LITERALS, indent=1))
    np.savez_compressed(HERE / "torch_vectors.npz", **torch_vectors())
    print("wrote", HERE / "reference_literals.json", "and", HERE / "torch_vectors.npz")
