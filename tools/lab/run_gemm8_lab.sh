#!/bin/bash
# usage: tools/lab/run_gemm8_lab.sh "<row counts>" ["<extra -D flags>"]
set -e
cd /root/repo
C=tiny-llm_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $F $2 -c $C/gemm8.hip -o /tmp/gemm8_lab_k.o
/opt/rocm/bin/hipcc $F $2 -c tools/lab/gemm8_lab.hip -o /tmp/gemm8_lab.o
/opt/rocm/bin/hipcc $F -c $C/qmm3.hip -o /tmp/gemm8_lab_q3.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm8_lab.o /tmp/gemm8_lab_k.o /tmp/gemm8_lab_q3.o -o tools/lab/gemm8_lab
CMD=""
for m in ${1:-2048}; do for e in ${3:-0}; do CMD="$CMD echo epi $e; tools/lab/gemm8_lab $m $e;"; done; done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | grep -vE "amdgpu.ids|sending"
