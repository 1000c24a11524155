#!/bin/bash
# On the GPU box: the staged reference tests (tools/stage_reference_tests.sh) on the real HIP library; log under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_zz_reference_tests_on_device_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40
tail -5 gpurun_out/reference_tests_on_device.log
