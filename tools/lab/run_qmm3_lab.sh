#!/bin/bash
# usage: tools/lab/run_qmm3_lab.sh "<ablation list, e.g. 0 1 2 4>" "<row counts, e.g. 64 32>" [extra env for the lab binary] ["<mode or -1> <prologue 0|1>"]
set -e
cd /root/repo
C=tiny-llm_amd/csrc
for abl in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DQMM3_ABL=$abl -c $C/qmm3.hip -o /tmp/qmm3_abl$abl.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DQMM3_ABL=$abl -c tools/lab/qmm3_lab.hip -o /tmp/qmm3_lab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/qmm3_lab.o /tmp/qmm3_abl$abl.o -o tools/lab/qmm3_lab_abl$abl
done
CMD=""; for abl in $1; do for m in ${2:-64}; do CMD="$CMD echo ablation $abl rows $m; $3 tools/lab/qmm3_lab_abl$abl $m $4;"; done; done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
