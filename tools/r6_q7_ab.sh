python -m pytest tests/test_zz_streaming_matmul_gpu.py tests/test_zz_batched_matmul_gpu.py tests/test_zz_aql_route_gpu.py tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r06_q7_pytest.txt
tail -4 gpurun_out/r06_q7_pytest.txt
for b in 5 8 16 17 32 33 48 64; do python tools/decode_ab.py --batch $b --steps 64 --profile-steps 2 - TL_ENGINE_OPTIONS=qmm7=0 - TL_ENGINE_OPTIONS=qmm7=0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(json.dumps({k:d[k] for k in d if k in ('variant','batch','ms_per_step','kinds_us','env')})[:400])
"; done > gpurun_out/r06_q7_ab.jsonl 2>&1
cat gpurun_out/r06_q7_ab.jsonl
