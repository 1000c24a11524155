#!/usr/bin/env python3
"""Copy the artefacts of an evidence run (tools/gpu_evidence.sh -> gpurun_out/evidence) into profiles/ under their tracked names,
and recompute each bench line's `roofline.rocprof` block from the rocprofv3 summary of the SAME run (bench.py on the GPU box
reads the previously committed summary).

    python tools/collect_evidence.py [--round r02] [--src gpurun_out/evidence]
"""
import argparse
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--src", type=Path, default=ROOT / "gpurun_out" / "evidence")
    args = ap.parse_args()
    import bench

    src, prof, rnd = args.src, ROOT / "profiles", args.round
    labs, traces = prof / f"{rnd}_labs", prof / f"{rnd}_rocprofv3"
    labs.mkdir(parents=True, exist_ok=True)
    traces.mkdir(parents=True, exist_ok=True)
    copies = {
        "trace_c2/bench_kernel_stats.csv": traces / "bench_config2_kernel_stats.csv",
        "trace_c3/bench_kernel_stats.csv": traces / "bench_config3_kernel_stats.csv",
        "trace_c5/bench_kernel_stats.csv": traces / "bench_config5_kernel_stats.csv",
        "trace_b64/b64_kernel_stats.csv": traces / "batched_decode_64seq_kernel_stats.csv",
        "ab_batched.jsonl": labs / "batched_decode_final.jsonl",
        "replicas_n1.json": labs / "serve_replicas_n1_b64.json",
        "replicas_n1.log": labs / "serve_replicas_n1_b64.txt",
        "replicas_n1_reference_admission.log": labs / "serve_replicas_n1_b64_reference_admission.txt",
        "acceptance.log": labs / "acceptance_128_129.txt",
        "operators.json": labs / "operators_decode_projections.json",
        "operators.log": labs / "operators_decode_projections.txt",
        "attention.json": labs / "attention_decode_contexts.json",
        "attention.log": labs / "attention_decode_contexts.txt",
        "parity_numbers.jsonl": prof / f"{rnd}_parity_numbers.jsonl",
    }
    for name, dst in copies.items():
        if (src / name).exists():
            shutil.copyfile(src / name, dst)
        else:
            print("missing", name)
    tail = (src / "pytest.log").read_text().strip().splitlines()[-3:]
    (prof / f"{rnd}_gpu_pytest_summary.txt").write_text("\n".join(tail) + "\n")
    for name, cfg in (("bench.json", 2), ("bench_c3.json", 3), ("bench_c5.json", 5)):
        line = json.loads((src / name).read_text().strip().splitlines()[-1])
        r = line["roofline"]
        from tiny_llm_hip.synthetic import QWEN3_CONFIGS

        c = QWEN3_CONFIGS["qwen3-4b"]
        w4 = lambda rows, cols: rows * cols / 2 + rows * (cols / 128) * 4
        hs, inter, L = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"]
        qkv_rows = (c["num_attention_heads"] + 2 * c["num_key_value_heads"]) * c["head_dim"]
        kinds = {"gemv_qkv": {"bytes": L * w4(qkv_rows, hs)}, "gemv_o": {"bytes": L * w4(hs, c["num_attention_heads"] * c["head_dim"])},
                 "gemv_gate_up": {"bytes": L * w4(2 * inter, hs)}, "gemv_down": {"bytes": L * w4(hs, inter)},
                 "gemv_lm_head": {"bytes": w4(c["vocab_size"], hs)}}
        rp = bench.rocprof_gemv_rate(traces / f"bench_config{cfg}_kernel_stats.csv", kinds, L)
        if rp:
            rp["source"] = "rocprofv3 --kernel-trace --stats of the same command in the same evidence run (its own process; tools/evidence_run.sh)"
            rp["stamp_minus_rocprof_us_per_launch"] = round(rp["avg_launch_us"] - r["avg_launch_us"], 3)
            r["rocprof"] = rp
        (prof / f"{rnd}_bench_config{cfg}.json").write_text(json.dumps(line) + "\n")
        print(f"config {cfg}: {line['value']} tok/s, {line['ms_per_step']} ms/step, prefill {line.get('prefill_tokens_per_s')}, "
              f"frac {r['frac']} (rocprof {rp['frac'] if rp else None}), step_frac {r.get('step_frac')}, "
              f"kv frac {r['attention_kv']['frac']}")


if __name__ == "__main__":
    main()
