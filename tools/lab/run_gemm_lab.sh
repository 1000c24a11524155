#!/bin/bash
# usage: tools/lab/run_gemm_lab.sh "<ablation list, e.g. 0 1 2 4>" [M]   -- builds one lab binary per ablation and runs them on the GPU box
set -e
cd /root/repo
C=tiny-llm_amd/csrc
for abl in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DQMM_ABL=$abl -c $C/qmm.hip -o /tmp/qmm_abl$abl.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/lab/gemm_lab.hip -o /tmp/gemm_lab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm_lab.o /tmp/qmm_abl$abl.o $C/build/qmv_fused.o $C/build/capi_core.o -o tools/lab/gemm_lab_abl$abl
done
CMD=""; for abl in $1; do CMD="$CMD echo ablation $abl; $3 tools/lab/gemm_lab_abl$abl ${2:-2048};"; done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
