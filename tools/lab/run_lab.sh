#!/bin/bash
# usage: tools/lab/run_lab.sh <tag> [grep-regex]   — rebuild the library + lab binary, run on the GPU box
set -e
cd /root/repo
make -j8 -C tiny-llm_amd/csrc 2>&1 | grep -E "error|warning" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/lab/gemv_lab.hip -o /tmp/gemv_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemv_lab.o tiny-llm_amd/csrc/build/qmv_fused.o tiny-llm_amd/csrc/build/qmv3.o tiny-llm_amd/csrc/build/capi_core.o -o tools/lab/gemv_lab
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "tools/lab/gemv_lab $3 > gpurun_out/lab_$1.log 2>&1" 2>&1 | grep -E "status|left"
grep -E "${2:-.}" gpurun_out/lab_$1.log
