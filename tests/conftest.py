"""Shared pytest configuration.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu`` on the GPU box).
* ``sys.path``: the repo root (for ``oracle``) and ``tiny-llm_amd`` + ``tiny-llm_amd/extensions_hip``
  (the source roots, like the reference's ``pythonpath = ["src", "."]``, pyproject.toml:63-65).
* Nothing here reads /root/reference: it does not exist on the GPU box.
"""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for extra in (ROOT, ROOT / "tests", ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(extra) not in sys.path:
        sys.path.insert(0, str(extra))


# the reference's own test files staged by tools/stage_reference_tests.sh run in their own pytest process, with the facade on
# the path (tests/test_zz_reference_tests_on_device_gpu.py): they are not part of this suite's collection
collect_ignore = ["_reference_staged"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (pytest -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_libs():
    """libtinyllm_hip.so (hipcc cross-compiles without a GPU) and the C oracle; built on demand."""
    lib = ROOT / "tiny-llm_amd" / "extensions_hip" / "tiny_llm_ext_hip" / "libtinyllm_hip.so"
    ora = ROOT / "oracle" / "libqwen3_oracle.so"
    if not lib.exists() or not ora.exists():
        import __graft_entry__

        __graft_entry__.build()
    return lib, ora


@pytest.fixture()
def cpu_ext(built_libs, monkeypatch):
    """Route the host-logic modules' extension calls to the oracle-backed fake (tests/fake_ext.py)."""
    from fake_ext import FakeExt

    import tiny_llm_hip.attention
    import tiny_llm_hip.paged_kv_cache

    fake = FakeExt()
    for module in (tiny_llm_hip.attention, tiny_llm_hip.paged_kv_cache):
        monkeypatch.setattr(module, "tiny_llm_ext_hip", fake)
    return fake
