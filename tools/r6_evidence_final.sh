#!/bin/bash
# round 6, final tree: the bench lines of configs 2 / 3 / 5 (each with its own rocprofv3 summary), the prefill FlashAttention A/B across wave
# counts, the GPU suite.  Run from the repository root on a GPU box; files land under gpurun_out/r06f/ and are copied into profiles/r06/.
mkdir -p gpurun_out/r06f
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06f/bench_config2.json 2> gpurun_out/r06f/bench_config2.err
cp gpurun_out/bench_rocprof/bench_config2_kernel_stats.csv gpurun_out/r06f/ 2>/dev/null; cp gpurun_out/bench_rocprof/bench_config2_kernel_stats.csv.meta.json gpurun_out/r06f/ 2>/dev/null
python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06f/bench_config3.json 2> gpurun_out/r06f/bench_config3.err
cp gpurun_out/bench_rocprof/bench_config3_kernel_stats.csv* gpurun_out/r06f/ 2>/dev/null
python bench.py --config 5 --no-cpu-baseline > gpurun_out/r06f/bench_config5.json 2> gpurun_out/r06f/bench_config5.err
cp gpurun_out/bench_rocprof/bench_config5_kernel_stats.csv* gpurun_out/r06f/ 2>/dev/null
python tools/lab/fa_waves_ab.py 2>/dev/null | grep "^L=" > gpurun_out/r06f/prefill_fa_waves_ab.log
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r06f/gpu_pytest_summary.txt
tail -3 gpurun_out/r06f/gpu_pytest_summary.txt
