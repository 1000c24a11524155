#!/bin/bash
# usage: tools/lab/ab_gemm_wpipe.sh "<row counts>" [extra command run on the box afterwards]
# A/B of the prefill GEMM's weight pipeline (csrc/qmm.hip QMM_WPIPE: 0 = one 64-wide step ahead, per-step requests; 1 = by the group,
# two groups ahead) and of the 8-wave tile for gate|up (QMM_NW8), alternating on one box.
set -e
cd /root/repo
C=tiny-llm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/lab/gemm_lab.hip -o /tmp/gemm_lab.o
for v in "0 1" "1 1" "1 0"; do
  set -- $v "$@"; wp=$1; nw8=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -DQMM_WPIPE=$wp -DQMM_NW8=$nw8 -c $C/qmm.hip -o /tmp/qmm_wp${wp}_nw${nw8}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm_lab.o /tmp/qmm_wp${wp}_nw${nw8}.o $C/build/qmv_fused.o $C/build/capi_core.o -o tools/lab/gemm_lab_wp${wp}_nw${nw8}
done
CMD=""
for rep in 1 2; do for M in $1; do for v in wp0_nw1 wp1_nw1 wp1_nw0; do CMD="$CMD echo $v M=$M; tools/lab/gemm_lab_$v $M;"; done; done; done
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD $2" 2>&1 | grep -vE "amdgpu.ids|sending"
