"""Generates tests/golden/reference_stack_vectors.npz: logits of THE REFERENCE'S STACK -- its own Python sources
(/root/reference/src/tiny_llm_ref: Qwen3ModelWeek3 on the paged cache, Qwen3ModelWeek2 with every Week-2 kernel) calling its own
kernel code (src/extensions_ref/src/*.metal, compiled for the host: oracle/_ref, bound through oracle/ref_extension.py with the
kernel selection of the reference's C++ primitives) -- on a small seeded Qwen3-shaped W4 checkpoint.  `mlx` is the torch
facade; nothing of this repository's product or numpy oracle takes part in producing a number.

What is not the reference as executed on Apple hardware: libm behind Metal's `fast::` functions, lane-order `simd_sum`, torch
arithmetic behind the few `mx.*` calls of the model code (residual adds, reshapes).  The tile GEMM and the MMA FlashAttention
kernels are not built for the host (MLX steel headers), so prompts are fed in chunks of at most 8 tokens -- the reference's
chunked prefill takes any chunk size and then runs its decode GEMV / paged decode kernels on them.

    python tests/golden/make_reference_stack_vectors.py       (build container only; ~10 min: every GPU thread is a fiber)
"""

import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REFERENCE = Path("/root/reference")
sys.path.insert(0, str(ROOT))
from oracle import ref_extension  # noqa: E402

_pkg = types.ModuleType("extensions_ref")
_pkg.__path__ = []
_pkg.tiny_llm_ext_ref = ref_extension
sys.modules["extensions_ref"] = _pkg
sys.modules["extensions_ref.tiny_llm_ext_ref"] = ref_extension
sys.path[:0] = [str(REFERENCE / "src"), str(REFERENCE), str(ROOT / "tiny-llm_amd" / "compat"), str(ROOT / "tiny-llm_amd"),
                str(ROOT / "tiny-llm_amd" / "extensions_hip"), str(ROOT / "tests")]

import mlx.core as mx  # noqa: E402
import tiny_llm_ref as R  # noqa: E402

from helpers import TINY_CFG, to_mlx_shaped  # noqa: E402
from oracle import tiny_oracle as O  # noqa: E402

assert Path(R.__file__).is_relative_to(REFERENCE)
CHUNK, STEPS = 8, 5
CASES = {"week3_paged_p20": ("week3", 20, 61), "week2_kernels_p12": ("week2", 12, 62)}  # name -> (model, prompt tokens, prompt seed)


def bits(t):
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def main() -> None:
    torch.set_num_threads(1)
    out = {}
    cfg = dict(TINY_CFG)
    with mx.stream(mx.cpu):
        tree = to_mlx_shaped(cfg, O.make_qwen3_weights(cfg, seed=3, sigma=0.05), device="cpu")
        for name, (which, n, pseed) in CASES.items():
            model = R.Qwen3ModelWeek3(tree, page_size=16) if which == "week3" else R.Qwen3ModelWeek2(tree)
            prompt = [int(t) for t in np.random.default_rng(pseed).integers(1, cfg["vocab_size"], size=n)]
            cache = model.create_kv_cache()
            rows, offset = [], 0
            for start in range(0, n, CHUNK):  # chunked prefill, the reference's Request.try_prefill shape (batch.py:48-76)
                part = prompt[start:start + CHUNK]
                logits = model(mx.array([part], dtype=mx.int32), offset, cache, logits_to_keep=1)
                rows.append(logits[0, -1])
                offset += len(part)
                print(name, "prefilled", offset, flush=True)
            ids = [int(rows[-1].argmax())]
            for _ in range(STEPS):
                logits = model(mx.array([[ids[-1]]], dtype=mx.int32), offset, cache, logits_to_keep=1)
                rows.append(logits[0, -1])
                ids.append(int(logits[0, -1].argmax()))
                offset += 1
                print(name, "decoded", len(ids) - 1, flush=True)
            for layer_cache in cache:
                layer_cache.release()
            out[f"{name}/prompt"] = np.asarray(prompt, dtype=np.int32)
            out[f"{name}/ids"] = np.asarray(ids, dtype=np.int32)
            out[f"{name}/logits"] = bits(mx.stack(rows))  # last row of every prefill chunk, then every decode step
    np.savez_compressed(HERE / "reference_stack_vectors.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
