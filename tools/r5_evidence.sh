#!/bin/bash
# round 5: the bench lines of configs 2 / 3 / 5 on the final build, the route A/Bs (single stream and batched), the batched probe and sweep, the serving
# traces (run from the repository root on a GPU box)
mkdir -p gpurun_out/r05
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_config2.json 2> gpurun_out/r05/bench_config2.err
cp gpurun_out/bench_rocprof/bench_config2_kernel_stats.csv gpurun_out/r05/ 2>/dev/null; cp gpurun_out/bench_rocprof/bench_config2_kernel_stats.csv.meta.json gpurun_out/r05/ 2>/dev/null
python bench.py --config 3 --no-cpu-baseline > gpurun_out/r05/bench_config3.json 2> gpurun_out/r05/bench_config3.err
cp gpurun_out/bench_rocprof/bench_config3_kernel_stats.csv* gpurun_out/r05/ 2>/dev/null
python bench.py --config 5 --no-cpu-baseline > gpurun_out/r05/bench_config5.json 2> gpurun_out/r05/bench_config5.err
cp gpurun_out/bench_rocprof/bench_config5_kernel_stats.csv* gpurun_out/r05/ 2>/dev/null
python tools/aql_ab.py --rounds 3 --steps 128 --modes graph,aql,aql_fences 2>/dev/null > gpurun_out/r05/replay_route_ab.jsonl
TL_AQL=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --rocprof off > gpurun_out/r05/bench_config2_hipgraph_route.json 2>/dev/null
for b in 2 4 8 16 32 64; do python tools/batch_decode_probe.py --batch $b --context 128 --steps 32 --profile; done 2>/dev/null > gpurun_out/r05/batched_decode_probe.jsonl
bash tools/lab/ab_batched_routes.sh > gpurun_out/r05/batched_decode_replay_route_ab.log 2>&1
python tools/batch_profile_sweep.py 5 8 12 16 17 24 32 33 48 64 2>/dev/null > gpurun_out/r05/batched_decode_profile_sweep_aql_route.jsonl
bash tools/run_serving_benches.sh > gpurun_out/r05/serving.log 2>&1; cp gpurun_out/serving/*.json gpurun_out/r05/ 2>/dev/null
