#!/usr/bin/env python3
"""Debug aid (GPU): characterise the fused GEMV at a given (rows, cols, M) against a torch fp32 dequantised product."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    sys.path.insert(0, str(p))
import torch
import tiny_llm_ext_hip as ext
from tiny_llm_hip.synthetic import quantize

torch.manual_seed(0)
for rows, cols in ((2560, 9728), (512, 9728), (2560, 8192), (2560, 4096), (2560, 9600)):
    w = (torch.randn(rows, cols, device="cuda") * 0.02).to(torch.bfloat16)
    packed, scales, biases = quantize(w)
    # dequantised reference
    q = torch.stack([(packed.to(torch.int64) >> (4 * i)) & 15 for i in range(8)], dim=-1).reshape(rows, cols).float()
    deq = q * scales.float().repeat_interleave(128, 1) + biases.float().repeat_interleave(128, 1)
    tw = ext.TiledW4(packed, scales, biases)
    for M in (1, 2, 3, 4):
        a = torch.randn(M, cols, device="cuda").to(torch.bfloat16)
        want = (a.float() @ deq.T)
        for rep in range(2):
            out, info = ext.decode_linear(tw, a, kernel=1)
            torch.cuda.synchronize()
            o = out.float()
            nan = torch.isnan(o)
            err = (o - want).abs()
            err[nan] = 0
            print(f"rows {rows} cols {cols} M {M} rep {rep} plan {info['p']} kernel {info['kernel']}: nan/row {nan.sum(1).tolist()} "
                  f"max err {err.max().item():.4f} bad>0.05/row {(err > 0.05).sum(1).tolist()}", flush=True)
            if nan.any() and rep == 0:
                cols_nan = nan[0].nonzero().flatten()
                print("   first nan cols row0:", cols_nan[:20].tolist(), " non-nan cols row0:", (~nan[0]).nonzero().flatten()[:20].tolist())
    tw.close()
