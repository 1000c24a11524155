"""Operator-level GEMV timing at the Qwen3-4B decode shapes (cf. benches/bench_week2_operators.py:336-401).

Weights are rotated through enough distinct copies to exceed the 256 MiB Infinity Cache, so every launch
streams from HBM like the real model does."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    sys.path.insert(0, str(p))
import torch
import tiny_llm_ext_hip as ext

SHAPES = {  # name: (K out, N in)
    "q": (4096, 2560), "k": (1024, 2560), "qkv": (6144, 2560), "o": (2560, 4096),
    "gate": (9728, 2560), "gateup": (19456, 2560), "down": (2560, 9728), "lm_head": (151936, 2560),
}

def main():
    dev = "cuda"
    Ms = [int(a) for a in sys.argv[1:]] or [1]
    for M in Ms:
        for name, (K, N) in SHAPES.items():
            wbytes = K * N // 2 + 2 * K * (N // 128) * 2
            copies = max(2, min(48, (600 << 20) // wbytes + 1))
            ws = [torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=dev) for _ in range(copies)]
            sc = [(torch.rand(K, N // 128, device=dev) * 0.01).to(torch.bfloat16) for _ in range(copies)]
            bi = [(torch.rand(K, N // 128, device=dev) * -0.05).to(torch.bfloat16) for _ in range(copies)]
            x = torch.randn(M, N, device=dev).to(torch.bfloat16)
            for i in range(copies):
                ext.quantized_matmul(sc[i], bi[i], 128, 4, x, ws[i], True)
            torch.cuda.synchronize()
            iters = max(copies * 3, 30)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                j = i % copies
                ext.quantized_matmul(sc[j], bi[j], 128, 4, x, ws[j], True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            print(f"M={M} {name:8s} K={K:6d} N={N:5d} {wbytes/1e6:8.2f} MB  {us:8.2f} us  {wbytes/us/1e3:7.1f} GB/s", flush=True)
            del ws, sc, bi

if __name__ == "__main__":
    main()
