"""TEST INFRASTRUCTURE (CPU only).  Several of the reference's bench drivers time each variant in a FRESH python process
(`python -m benches.bench ...` from bench_course_progression.py / bench_chunked_prefill.py / bench_serving_progression.py,
`python -m benches.bench_long_context_attention --worker`, bench_week3_attention.py), with `<reference>/src` put FIRST on the
child's PYTHONPATH (benches/bench_course_progression.py:266-272).  When tests/run_reference_script.py runs such a driver in a
container without a GPU it puts this directory on PYTHONPATH and sets REFSOL_ORACLE_FAKELIB=1, so that every child
interpreter (a) resolves `tiny_llm_ref` / `extensions_ref` to the facade instead of the reference's MLX sources -- the facade
stands where `src/` stands in a reference checkout -- and (b) gets the numpy oracle behind libtinyllm_hip.so's C ABI
(tests/refsol_oracle_plugin.py).  Without that variable this file does nothing.  The product never loads it."""
import os
import sys

if os.environ.get("REFSOL_ORACLE_FAKELIB") == "1":
    _reference_src = os.path.join(os.environ.get("TINYLLM_REFERENCE_ROOT", "/root/reference"), "src")
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != _reference_src]
    import refsol_oracle_plugin

    refsol_oracle_plugin.pytest_configure(None)
