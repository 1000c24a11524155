"""CPU tier: the reference's OWN Python sources against the product's host mirror, side by side in one process, over identical
backends (the torch facade for `mlx`, the oracle-backed C ABI for the extension of BOTH) -- tests/reference_differential_cases.py.
Bit equality of every tensor, counter, id and printed line over: readable operators in three dtypes, quantised operators and
kernel wrappers, dense / batching caches, page pool and paged cache bookkeeping, the Week-1 model, the Week-2 model at every
checkpoint (prefill + decode), the Week-3 model paged and dense-gather with chunked prefill and staggered continuous batching
(the MoE layers to one bf16 rounding: their primitives differ), the generation loops, speculative decoding, the
continuous-batching scheduler's results and progress output, and the samplers' distributions.  Skipped without /root/reference."""

import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")


@pytest.mark.skipif(not (REFERENCE / "src" / "tiny_llm_ref").is_dir(), reason="/root/reference is not present (GPU box)")
def test_reference_sources_and_host_mirror_agree_bit_for_bit(built_libs):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=str(ROOT / "tests"))
    proc = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "reference_differential_cases.py"), "-p", "no:cacheprovider",
                           "-p", "refsol_oracle_plugin", "-q", "--tb=line"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    summary = re.search(r"(\d+) passed", proc.stdout)
    assert proc.returncode == 0 and summary and int(summary.group(1)) >= 26 and "failed" not in proc.stdout.splitlines()[-1], proc.stdout[-3000:]
