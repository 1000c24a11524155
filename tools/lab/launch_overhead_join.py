#!/usr/bin/env python3
"""Lab: join the in-kernel stamps of profiled decode steps (TL_PROFILE_DUMP) with the rocprofv3 kernel trace of the same process.
Both clocks are assumed to be the device's constant-rate counter (stamps in ticks, trace in ns); the offset between them is
fitted as the median of (trace start - stamp start) and only DIFFERENCES between kernels are read from the result."""
import csv
import re
import statistics
import sys

KIND = {0: "gemv_qkv", 1: "gemv_o", 2: "gemv_gate_up", 3: "gemv_down", 4: "gemv_lm_head", 5: "attention", 6: "merge", 7: "step_end"}
stamps, steps = [], []
for line in open(sys.argv[1]):
    if line.startswith("step"):
        khz = int(line.split()[2])
        steps.append([])
    else:
        k, t0, t1 = line.split()
        steps[-1].append((int(k), int(t0) * 1e6 / khz, int(t1) * 1e6 / khz))  # ns
rows = [r for r in csv.DictReader(open(sys.argv[2])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the profiled steps are the last ones: every launch is followed by a prof_reduce_kernel dispatch
seq = [(i, r) for i, r in enumerate(rows)]
reduce_idx = [i for i, r in seq if "prof_reduce_kernel" in r["Kernel_Name"]]
last = steps[-1]
idx = reduce_idx[-len(last):]
pairs = []
for (kind, t0, t1), ri in zip(last, idx):
    k = rows[ri - 1]
    pairs.append((kind, t0, t1, int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"]))
off = statistics.median(p[3] - p[1] for p in pairs)
agg = {}
for kind, t0, t1, s, e, name in pairs:
    d = agg.setdefault(KIND[kind], [])
    d.append(((t0 + off - s) / 1e3, (e - (t1 + off)) / 1e3, (t1 - t0) / 1e3, (e - s) / 1e3))
print(f"clock offset fitted: {off:.0f} ns (median start lag forced to 0)")
print(f"{'kind':14s} {'n':>3s} {'start lag us':>13s} {'end lag us':>11s} {'in-kernel us':>13s} {'rocprof us':>11s}")
for k, v in agg.items():
    n = len(v)
    print(f"{k:14s} {n:3d} {sum(x[0] for x in v) / n:13.2f} {sum(x[1] for x in v) / n:11.2f} {sum(x[2] for x in v) / n:13.2f} {sum(x[3] for x in v) / n:11.2f}")
