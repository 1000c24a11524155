"""CPU tier: the C-ABI shared library loads and exports every symbol include/*.h declares; host-only entry points
(policies, sizes, error plumbing) behave.  No kernel is launched here."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for header in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
        names.update(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound(built_libs):
    lib_path, _ = built_libs
    lib = ctypes.CDLL(str(lib_path))
    import tiny_llm_ext_hip as ext

    declared = declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported by libtinyllm_hip.so"
        assert name in ext._SIGNATURES, f"{name} has no ctypes signature in tiny_llm_ext_hip"
    for name in ext._SIGNATURES:
        assert name in declared, f"{name} is bound in Python but not declared in include/*.h"


def test_reference_operator_surface(built_libs):
    """Names, keyword names and defaults of the reference binding (bindings.cpp:14-47, pinned by
    tests_refsol/test_extension_interface_sync.py:354-378)."""
    import inspect

    import tiny_llm_ext_hip as ext

    want = {
        "quantized_matmul": ["scales", "biases", "group_size", "bits", "a", "b", "transpose_b", "use_simdgroup", "use_split_k", "stream"],
        "quantized_embedding": ["indices", "scales", "biases", "weight", "group_size", "bits", "stream"],
        "rms_norm": ["x", "weight", "eps", "stream"],
        "rope": ["x", "offsets", "dims", "base", "traditional", "stream"],
        "swiglu": ["gate", "up", "stream"],
        "decode_attention": ["query", "key", "value", "mask", "scale", "is_causal", "has_mask", "num_heads", "num_kv_heads", "stream"],
        "paged_cache_update": ["pages", "values", "page_id", "start", "stream"],
        "paged_attention": ["query", "key_pages", "value_pages", "block_table", "context_lens", "scale", "is_causal", "num_kv_heads", "num_heads", "stream"],
        "load_library": ["path"],
    }
    for name, params in want.items():
        got = list(inspect.signature(getattr(ext, name)).parameters)
        assert got[: len(params)] == params, (name, got)
    sig = inspect.signature(ext.quantized_matmul).parameters
    assert (sig["transpose_b"].default, sig["use_simdgroup"].default, sig["use_split_k"].default) == (False, True, False)
    sig = inspect.signature(ext.paged_attention).parameters
    assert sig["scale"].default == 1.0 and sig["is_causal"].default is False


def test_host_only_entry_points(built_libs):
    import torch

    import tiny_llm_ext_hip as ext

    lib = ext.lib()
    assert lib.tl_abi_version() == 1
    # split-K policy (reference quantized_matmul.cpp:138-151 re-derived for 256 CUs): never for GEMV rows, never
    # beyond N/128, and the chosen factor divides the reduction into whole groups
    assert lib.tl_quantized_matmul_split_k(1, 2560, 4096, 1, 1) == 1
    assert lib.tl_quantized_matmul_split_k(64, 2560, 4096, 1, 0) == 1
    assert lib.tl_quantized_matmul_split_k(512, 128, 2048, 1, 1) == 1
    assert lib.tl_quantized_matmul_split_k(128, 9728, 2560, 1, 1) == 19  # w_down at a 128-row chunk: 19 slices of 4 groups, not 4 of 19
    for M, N, K in [(32, 2048, 128), (16, 4096, 1024), (64, 9728, 256), (128, 2560, 2560)]:
        s = lib.tl_quantized_matmul_split_k(M, N, K, 1, 1)
        assert 1 <= s <= 20 and N % (s * 128) == 0  # (at most 20 slices since round 5: 76 groups = 4 x 19)
        assert lib.tl_quantized_matmul_workspace_bytes(M, N, K, 2, 1, 1) == (s * M * K * 2 if s > 1 else 0)
    assert lib.tl_engine_context_len(None, 0) == -1
    assert lib.tl_engine_step_bytes(None, 1) == 0
    # the extension is GPU-only, like the reference whose eval_cpu throws (quantized_matmul.cpp:103-109)
    x = torch.zeros(2, 128)
    with pytest.raises(RuntimeError, match="GPU-only"):
        ext.rms_norm(x, torch.ones(128), 1e-6)
    with pytest.raises(RuntimeError, match="GPU-only"):
        ext.swiglu(x, x)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device|GPU-only"):
            ext.load_library(".")
