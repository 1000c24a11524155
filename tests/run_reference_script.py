"""TEST INFRASTRUCTURE (CPU only): runs one of the reference's OWN scripts (benches/bench.py, main.py, batch-main.py, the
operator microbenches ...) unmodified, through the import facade, in a container without a GPU: libtinyllm_hip.so's entry
points are answered by the numpy oracle exactly as in tests/refsol_oracle_plugin.py.  What it shows: the reference's harness
code runs end to end on the facade + host mirror (argument handling, model dispatch, caches, scheduler, report lines, JSON
output).  The numbers it prints are oracle timings and mean nothing.  The product never loads this file.

usage: python tests/run_reference_script.py <script path relative to /root/reference> [script arguments...]
       REFSOL_REFERENCE_SOURCES=1 ... : CONTROL run -- the reference's own tiny_llm_ref sources behind the script instead of the
       product's host mirror (same facade for mlx, same oracle-backed extension binding)
"""

import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path(os.environ.get("TINYLLM_REFERENCE_ROOT", "/root/reference"))


def main() -> None:
    script = REFERENCE / sys.argv[1]
    os.environ["REFSOL_REFERENCE_BENCHES"] = "1"
    sys.path.insert(0, str(ROOT / "tests"))
    import refsol_oracle_plugin  # noqa: E402  (puts the facade and the product on the module search path)

    refsol_oracle_plugin.pytest_configure(None)
    for extra in (REFERENCE / "src", REFERENCE):  # the reference's own `pythonpath = ["src", "."]` (pyproject.toml:63-65)
        sys.path.insert(0, str(extra))
    # ... except that `tiny_llm_ref` / `extensions_ref` / `mlx*` must resolve to the facade, not to the reference's sources --
    # unless this is the CONTROL run (REFSOL_REFERENCE_SOURCES=1: the reference's own tiny_llm_ref on the same stand-ins; the
    # plugin has already aliased the extension), where only `mlx*` comes from the facade
    if os.environ.get("REFSOL_REFERENCE_SOURCES") == "1":
        sys.path.insert(sys.path.index(str(REFERENCE / "src")) + 1, str(ROOT / "tiny-llm_amd" / "compat"))
    else:
        sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "compat"))
    # child interpreters (the drivers' fresh worker processes) get the same module search path and the same stand-in
    child_path = [ROOT / "tests" / "facade_site", ROOT / "tests", ROOT / "tiny-llm_amd" / "compat", ROOT / "tiny-llm_amd",
                  ROOT / "tiny-llm_amd" / "extensions_hip", REFERENCE]
    os.environ["PYTHONPATH"] = os.pathsep.join(str(p) for p in child_path)
    os.environ["REFSOL_ORACLE_FAKELIB"] = "1"
    os.chdir(REFERENCE)
    sys.argv = [str(script), *sys.argv[2:]]
    runpy.run_path(str(script), run_name="__main__")


if __name__ == "__main__":
    main()
