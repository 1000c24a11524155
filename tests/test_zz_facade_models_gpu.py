"""GPU tier: checkpoint directory -> the product's loader (through the facade's `mlx_lm.load`) -> the course models on the HIP
kernels, against the facade's `mlx_lm` model on the same tensors: the reference's checkpoint-dependent tests (Week 1 model,
Week 2 incremental decode, Week 3 staggered batching; it skips them without a downloaded model) plus the Week-3 model on a
Qwen3-MoE checkpoint.  tests/facade_model_cases.py holds the cases; the build container runs the very same code with the numpy
oracle behind the C ABI (tests/test_loader_cpu.py)."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
COMPAT = str(ROOT / "tiny-llm_amd" / "compat")
if COMPAT not in sys.path:
    sys.path.insert(0, COMPAT)

# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu]


def _mx():
    import mlx.core as mx

    return mx


def test_week1_model_on_a_loaded_checkpoint(tmp_path):
    import facade_model_cases as cases

    with _mx().stream(_mx().gpu):
        cases.case_week1_model(tmp_path)


@pytest.mark.parametrize("checkpoint", ["kv-cache", "quantized-matvec", "decode-attention", "split-k"])
def test_week2_incremental_decode_on_a_loaded_checkpoint(tmp_path, checkpoint):
    import facade_model_cases as cases

    with _mx().stream(_mx().gpu):
        cases.case_week2_incremental_decode(tmp_path, checkpoint)


@pytest.mark.parametrize("moe", [False, True], ids=["dense", "qwen3-moe"])
def test_week3_staggered_batching_on_a_loaded_checkpoint(tmp_path, moe):
    import facade_model_cases as cases

    with _mx().stream(_mx().gpu):
        cases.case_week3_staggered_batching(tmp_path, moe=moe)
