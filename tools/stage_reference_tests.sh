#!/bin/bash
# Stages the reference's OWN test files (tests_refsol/ weeks 1-3 + its benches/ harness and harness tests) next to this
# repository's tests so that they travel to the GPU box with a gpurun snapshot -- the way oracle/_ref travels: built / staged
# from /root/reference where it exists, git-ignored (no reference file enters the history), NOT gpurun-ignored.
# On the device they run UNMODIFIED through tiny-llm_amd/compat with the real libtinyllm_hip.so (no oracle plugin):
#   tests/test_zz_reference_tests_on_device_gpu.py,  tools/run_reference_tests_on_device.sh
set -e
REF=${TINYLLM_REFERENCE_ROOT:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DST=$ROOT/tests/_reference_staged
[ -d "$REF/tests_refsol" ] || { echo "no reference tree at $REF: nothing staged"; exit 0; }
rm -rf "$DST"
mkdir -p "$DST/tests_refsol" "$DST/benches"
for f in utils.py tiny_llm_base.py test_rope.py test_model_names.py; do cp "$REF/tests_refsol/$f" "$DST/tests_refsol/"; done
for w in 1 2 3; do cp "$REF"/tests_refsol/test_week_${w}_day_*.py "$DST/tests_refsol/"; done
cp "$REF"/benches/*.py "$DST/benches/"
for f in main.py batch-main.py pyproject.toml README.md; do [ -f "$REF/$f" ] && cp "$REF/$f" "$DST/"; done
# what the reference's harness tests read besides code: its book's chapter headings, its published result files, and the
# student stub package whose constants they compare with the solution's (benches/test_bench_course_progression.py:110-262,
# benches/test_profile_week2_kernels.py:126-130)
mkdir -p "$DST/book/src" "$DST/benchmark_results" "$DST/src"
cp "$REF"/book/src/*.md "$DST/book/src/"
cp "$REF"/benchmark_results/* "$DST/benchmark_results/"
cp -r "$REF/src/tiny_llm" "$DST/src/tiny_llm"
find "$DST" -name __pycache__ -type d -prune -exec rm -rf {} +
echo "staged $(ls "$DST/tests_refsol" | wc -l) + $(ls "$DST/benches" | wc -l) reference files under tests/_reference_staged (git-ignored)"
