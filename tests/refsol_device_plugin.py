"""pytest plugin (TEST INFRASTRUCTURE, GPU box): lets the reference's UNMODIFIED test files, staged under
tests/_reference_staged by tools/stage_reference_tests.sh, run on the MI355X through the import facade (tiny-llm_amd/compat)
with the REAL extension -- tiny_llm_ext_hip over libtinyllm_hip.so.  Nothing is replaced: this file only

* supplies a one-call stand-in for pytest-benchmark's ``benchmark`` fixture (not installed in this image; the assertions of
  benches/test_attention.py / test_quantized_matmul.py stay), and
* lets the staged ``benches`` namespace package (the reference's harness files) win over this repository's ``benches/``.

Counterpart of tests/refsol_oracle_plugin.py, which answers the C ABI with the numpy oracle in the GPU-less build container.
"""

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def pytest_configure(config):
    import tiny_llm_ext_hip as ext  # fails loudly when libtinyllm_hip.so is missing

    assert type(ext._lib).__name__ != "FakeLib", "the oracle stand-in must not be active in a device run"
    sys.modules.pop("benches", None)
    sys.path[:] = [p for p in sys.path if Path(p or ".").resolve() != ROOT]


def pytest_terminal_summary(terminalreporter):
    """Which native libraries this pytest process has mapped: the HIP extension must be among them, no oracle library may be."""
    libs = sorted({line.split()[-1] for line in open("/proc/self/maps") if line.rstrip().endswith(".so") and "/repo/" in line})
    terminalreporter.write_line("native libraries of this repository mapped by the test process: " + (", ".join(Path(p).name for p in libs) or "none"))
    assert any(p.endswith("libtinyllm_hip.so") for p in libs), "libtinyllm_hip.so is not mapped"
    assert not any("oracle" in p for p in libs), "an oracle library is mapped in a device run"


try:
    import pytest_benchmark  # noqa: F401
except ImportError:
    import pytest

    @pytest.fixture
    def benchmark():
        def run_once(function, *args, **kwargs):
            return function(*args, **kwargs)

        return run_once
