#!/bin/bash
# builds and runs tools/lab/overlap_lab (round 5: overlapped dispatch of dependent launches); run from the repository root on a GPU box
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/overlap_lab.hip -o tools/lab/overlap_lab -lhsa-runtime64
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output tools/lab/overlap_lab.hip -o tools/lab/overlap_lab.hsaco
mkdir -p gpurun_out
timeout 120 tools/lab/overlap_lab 2>&1 | tee gpurun_out/overlap_lab.log
