#!/bin/bash
# usage: tools/lab/run_qmm6_lab.sh "<row counts, e.g. 64 8>" [trace: 1 = build with -DQMM6_TRACE and print the phase stamps] ["<ablations, e.g. 0 1 2 4>"] [1 = weighted rows in fragment order (and the row-streaming kernel qmm7 beside qmm6)]
set -e
cd /root/repo
C=tiny-llm_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16"
D=""; [ "${2:-0}" = "1" ] && D="-DQMM6_TRACE -DQMM7_TRACE"
/opt/rocm/bin/hipcc $F -c $C/qmm3.hip -o /tmp/qmm6_lab_q3.o
/opt/rocm/bin/hipcc $F $D -mllvm -amdgpu-mfma-vgpr-form -c $C/qmm7.hip -o /tmp/qmm6_lab_q7.o
/opt/rocm/bin/hipcc $F $D -c tools/lab/qmm6_lab.hip -o /tmp/qmm6_lab.o
CMD=""
for abl in ${3:-0}; do
  /opt/rocm/bin/hipcc $F $D -DQMM6_ABL=$abl -mllvm -amdgpu-mfma-vgpr-form -c $C/qmm6.hip -o /tmp/qmm6_lab_k$abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/qmm6_lab.o /tmp/qmm6_lab_k$abl.o /tmp/qmm6_lab_q3.o /tmp/qmm6_lab_q7.o -o tools/lab/qmm6_lab_abl$abl
  for m in ${1:-64}; do CMD="$CMD echo ablation $abl; tools/lab/qmm6_lab_abl$abl $m ${4:-0};"; done
done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
