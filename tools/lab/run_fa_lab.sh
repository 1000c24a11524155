#!/bin/bash
# usage: tools/lab/run_fa_lab.sh "<ablation list, e.g. 0 1 2 4 8 16>" [L] [ctx]
set -e
cd /root/repo
C=tiny-llm_amd/csrc
for abl in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFA_ABL=$abl -c $C/attention.hip -o /tmp/att_abl$abl.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/lab/fa_lab.hip -o /tmp/fa_lab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/fa_lab.o /tmp/att_abl$abl.o $C/build/capi_core.o -o tools/lab/fa_lab_abl$abl
done
CMD=""; for abl in $1; do CMD="$CMD echo ablation $abl; tools/lab/fa_lab_abl$abl ${2:-2048} ${3:-8192};"; done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
