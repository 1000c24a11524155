for b in 8 16 32 64; do python tools/batch_decode_probe.py --batch $b --context 128 --steps 32 --profile; done 2>/dev/null | tee gpurun_out/batched_probe_base.jsonl
