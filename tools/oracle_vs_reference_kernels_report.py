#!/usr/bin/env python3
"""Evidence table: the numpy oracle against the reference's own Metal kernels executed on the host (oracle/_ref, see
tests/test_oracle_vs_reference_kernels_cpu.py for the assertions): per kernel the share of bit-identical outputs and the largest
difference.   python tools/oracle_vs_reference_kernels_report.py > profiles/r02_labs/oracle_vs_reference_kernels.txt"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import ref_kernels as K  # noqa: E402
from oracle import tiny_oracle as O  # noqa: E402

rows = []


def row(kernel, source, dtype, got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    rows.append((kernel, source, dtype, f"{(got == want).mean() * 100:6.2f} %", f"{np.abs(got - want).max():.2e}", str(got.size)))


rng = np.random.default_rng(0)
for dt in ("f32", "f16", "bf16"):
    x = O.cast(rng.standard_normal((4, 2560)).astype(np.float32) * 1.7, dt)
    w = O.cast(1 + 0.1 * rng.standard_normal(2560).astype(np.float32), dt)
    row("week2_rms_norm", "week2_kernels.metal:6-48", dt, K.rms_norm(x, w, 1e-6, dt), O.rms_norm_fast(x, w, 1e-6, dt))
    h = O.cast(rng.standard_normal((2, 4, 8, 128)).astype(np.float32), dt)
    row("week2_rope", "week2_kernels.metal:50-105", dt, K.rope(h, [0, 900], 128, 1e6, False, dt), O.rope(h, np.array([0, 900]), 128, 1e6, False, dt))
    g, u = (O.cast(rng.standard_normal((4, 9728)).astype(np.float32) * s, dt) for s in (3.0, 1.0))
    row("week2_swiglu", "week2_kernels.metal:107-117", dt, K.swiglu(g, u, dt), O.swiglu(g, u, dt))
for dt in ("f16", "bf16"):
    wts = O.cast(rng.standard_normal((64, 2560)).astype(np.float32) * 0.05, dt)
    packed, scales, biases = O.quantize_affine(wts, dtype=dt)
    for M in (1, 4):
        a = O.cast(rng.standard_normal((M, 2560)).astype(np.float32), dt)
        want = O.quantized_matmul(scales, biases, a, packed, dt)
        row(f"quantized_matmul_vanilla M={M}", "quantized_matmul.metal:8-56", dt, K.quantized_matmul_vanilla(scales, biases, a, packed, dt), want)
        row(f"quantized_matvec_x4_fast M={M}", "quantized_matmul.metal:441-538", dt, K.quantized_matvec_x4_fast(scales, biases, a, packed, dt), want)
    idx = np.arange(0, 64, 5)
    row("quantized_embedding", "quantized_matmul.metal:58-89", dt, K.quantized_embedding(idx, scales, biases, packed, dt), O.quantized_embedding(idx, scales, biases, packed, dt))
    parts = O.cast(rng.standard_normal((4, 8, 64)).astype(np.float32), dt)
    row("quantized_matmul_splitk_reduce", "quantized_matmul.metal:278-293", dt, K.splitk_reduce(parts, dt), O.cast(parts.astype(np.float64).sum(0).astype(np.float32), dt))
for dt in ("f32", "bf16"):
    Hq, Hkv, D, S = 8, 2, 128, 150
    q = O.cast(rng.standard_normal((Hq, 1, D)).astype(np.float32), dt)
    k = O.cast(rng.standard_normal((Hkv, S, D)).astype(np.float32), dt)
    v = O.cast(rng.standard_normal((Hkv, S, D)).astype(np.float32), dt)
    row("week2_decode_attention", "week2_kernels.metal:119-235", dt, K.decode_attention(q, k, v, D ** -0.5, Hq, Hkv, True, None, dt),
        O.decode_attention(q, k, v, D ** -0.5, Hq, Hkv, True, None, dt))
    kp = O.cast(rng.standard_normal((5, Hkv, 64, D)).astype(np.float32), dt)
    vp = O.cast(rng.standard_normal((5, Hkv, 64, D)).astype(np.float32), dt)
    table, ctx = np.array([[3, 0, 4, -1]], dtype=np.int32), np.array([150], dtype=np.int32)
    row("paged_attention_decode", "paged_attention.metal:108-248", dt, K.paged_attention_decode(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, dt, fixed_d128=dt == "bf16"),
        O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, dt))
q32 = rng.standard_normal((4, 20, 64)).astype(np.float32)
kp32, vp32 = rng.standard_normal((6, 2, 8, 64)).astype(np.float32), rng.standard_normal((6, 2, 8, 64)).astype(np.float32)
table, ctx = np.array([[5, 1, 3, -1]], dtype=np.int32), np.array([24], dtype=np.int32)
row("paged_attention_scalar_f32", "paged_attention.metal:508-674", "f32", K.paged_attention_scalar_f32(q32, kp32, vp32, table, ctx, 0.125, True, 2, 4),
    O.paged_attention(q32, kp32, vp32, table, ctx, 0.125, True, 2, 4, "f32"))

print("# numpy oracle (oracle/tiny_oracle.py) vs the reference's own Metal kernels run on the host (oracle/_ref/libref_metal_kernels.so)")
print("# shapes at Qwen3-4B row widths where the emulation time allows; columns: bit-identical outputs, largest |difference|, outputs compared")
head = ("kernel (reference)", "source", "dtype", "identical", "max |diff|", "n")
widths = [max(len(r[i]) for r in rows + [head]) for i in range(6)]
for r in [head] + rows:
    print("  ".join(c.ljust(w) for c, w in zip(r, widths)))
