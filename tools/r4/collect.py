#!/usr/bin/env python3
"""Copy the artefacts of the round-4 evidence run (tools/r4/evidence_run.sh -> gpurun_out/evidence_r4) into profiles/ under their
tracked names.  The bench lines are committed AS PRINTED: their roofline.rocprof block was measured inside the run (bench.py's
rocprofv3 child), and the kernel-stats CSV of each is committed beside them with its capture-date sidecar.

    python tools/r4/collect.py [--src gpurun_out/evidence_r4]"""
import argparse
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", type=Path, default=ROOT / "gpurun_out" / "evidence_r4")
    args = ap.parse_args()
    src, prof = args.src, ROOT / "profiles"
    labs, traces = prof / "r04_labs", prof / "r04_rocprofv3"
    labs.mkdir(parents=True, exist_ok=True)
    traces.mkdir(parents=True, exist_ok=True)
    copies = {
        "bench.json": prof / "r04_bench_config2.json", "bench_c3.json": prof / "r04_bench_config3.json",
        "bench_c5.json": prof / "r04_bench_config5.json", "bench_driver_shape.json": prof / "r04_bench_driver_shape.json",
        "bench.err": labs / "bench_config2_progress.txt",
        "ab_batched.jsonl": labs / "batched_decode_final.jsonl",
        "ab_qmm6.jsonl": labs / "batched_decode_final_ab_without_qmm6.jsonl", "qmm6_lab.txt": labs / "batched_matmul_qmm6_lab_final.txt",
        "ab_attn_mfma.jsonl": labs / "attention_mfma_walk_final_ab.jsonl",
        "replicas_n1.json": labs / "serve_replicas_n1_b64.json", "replicas_n1.log": labs / "serve_replicas_n1_b64.txt",
        "operators.json": labs / "operators_decode_projections.json", "operators.log": labs / "operators_decode_projections.txt",
        "attention.json": labs / "attention_decode_contexts.json", "attention.log": labs / "attention_decode_contexts.txt",
        "parity_numbers.jsonl": prof / "r04_parity_numbers.jsonl",
    }
    for cfg in (2, 3, 5):
        for suffix in ("", ".meta.json"):
            copies[f"bench_rocprof/bench_config{cfg}_kernel_stats.csv{suffix}"] = traces / f"bench_config{cfg}_kernel_stats.csv{suffix}"
    for name, dst in copies.items():
        if (src / name).exists():
            shutil.copyfile(src / name, dst)
        else:
            print("missing", name)
    for tag, dst in (("trace_b64", "batched_decode_64seq_kernel_stats.csv"), ("trace_b8", "batched_decode_8seq_kernel_stats.csv")):
        found = sorted((src / tag).rglob("*kernel_stats.csv")) if (src / tag).exists() else []
        if found:
            shutil.copyfile(found[-1], traces / dst)
        else:
            print("missing", tag)
    # PMC passes (tools/lab/pmc_gemv.sh): per-kernel medians under profiles/r04_pmc/, profiles/traffic.json by tools/make_traffic_json.py
    pmc = ROOT / "gpurun_out" / "pmc_gemv"
    if pmc.exists():
        import collections
        import csv
        import subprocess
        (prof / "r04_pmc").mkdir(exist_ok=True)
        ok = True
        for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            found = sorted((pmc / kind).rglob("lab_counter_collection.csv"))
            if not found:
                print("missing pmc", kind)
                ok = False
                continue
            if found[-1] != pmc / kind / "lab_counter_collection.csv":
                shutil.copyfile(found[-1], pmc / kind / "lab_counter_collection.csv")
            rows = collections.defaultdict(list)
            for r in csv.DictReader(open(found[-1])):
                rows[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
            with open(prof / "r04_pmc" / f"gemv_lab_{kind}_size.csv", "w") as f:
                f.write("kernel,grid_size,counter,dispatches,median_KiB,min_KiB,max_KiB,sum_KiB\n")
                for (k, g), v in sorted(rows.items()):
                    v = sorted(v)
                    f.write(f'"{k}",{g},{counter},{len(v)},{v[len(v) // 2]},{v[0]},{v[-1]},{round(sum(v), 1)}\n')
        if ok:
            subprocess.run(["python3", str(ROOT / "tools" / "make_traffic_json.py")], check=False, stdout=subprocess.DEVNULL)
            print("traffic.json rewritten from this round's PMC passes")
    if (src / "pytest.log").exists():
        tail = (src / "pytest.log").read_text().strip().splitlines()[-14:]
        (prof / "r04_gpu_pytest_summary.txt").write_text("\n".join(tail) + "\n")
    for name in ("bench.json", "bench_driver_shape.json", "bench_c3.json", "bench_c5.json"):
        try:
            b = json.loads((src / name).read_text().strip().splitlines()[-1])
            r = b["roofline"]
            print(name, b["value"], "tok/s", b["ms_per_step"], "ms; frac", r["frac"], "|", r.get("frac_source"), "| stamps", r.get("frac_in_kernel_stamps"),
                  "| step", r["step_frac"], "| kv", r["attention_kv"]["frac"])
        except Exception as exc:
            print(name, "unreadable:", exc)


if __name__ == "__main__":
    main()
