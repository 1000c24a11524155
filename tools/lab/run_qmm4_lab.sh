#!/bin/bash
# usage: tools/lab/run_qmm4_lab.sh [rows ...]   -- builds the prototype GEMM lab against the library's objects and runs it on the GPU box
set -e
cd /root/repo
C=tiny-llm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/lab/qmm4_lab.hip -o /tmp/qmm4_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/qmm4_lab.o $C/build/qmm.o $C/build/qmv_fused.o $C/build/qmv3.o $C/build/capi_core.o -o tools/lab/qmm4_lab
CMD=""; for m in ${@:-2048}; do CMD="$CMD echo rows $m; timeout 120 tools/lab/qmm4_lab $m;"; done
timeout 900 /usr/local/graft/bin/gpurun --timeout 300 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
