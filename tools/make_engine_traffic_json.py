"""HBM bytes of the ENGINE's decode step from the rocprofv3 --pmc passes of tools/lab/pmc_engine_step.sh (tools/lab/engine_step_lab).

    python tools/make_engine_traffic_json.py <pass directory> <context> <steps> <batch>      -> <pass directory>/traffic_engine.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts a wide coalesced stream at about half its
bytes (guides/MI355X_MICROARCH.md, HBM section): the factor is derived from the driver's own calibration kernel (stream_read_kernel,
4 x 1 GiB of known reads in the same process), never from an engine kernel.  A decode step = every (kernel, grid) group whose call
count is a whole multiple of the steps run (16 warm + the counted ones); per group the MEDIAN dispatch is taken (the attention
kernels grow by one token per step)."""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

src, context, steps, batch = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
total_steps = 16 + steps


def load(which):
    files = glob.glob(str(src / which / "**" / "*counter_collection.csv"), recursive=True)
    per_dispatch = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            key = int(r["Dispatch_Id"])
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            ent = per_dispatch.setdefault(key, [name, int(r["Grid_Size"]), 0.0])
            ent[2] += float(r["Counter_Value"])  # (one row per counter instance, if the tool splits them: summed)
    groups = collections.defaultdict(list)
    for name, grid, value in per_dispatch.values():
        groups[(name, grid)].append(value * 1024.0)
    return groups


fetch, write = load("fetch"), load("write")
calib = [v for (name, _), vals in fetch.items() if "stream_read_kernel" in name for v in vals]
known = 4 * (1 << 30)
factor = known / sum(calib) if calib else None
alg = None
m = re.search(r"algorithmic bytes per step (\d+)", (src / "plain.log").read_text()) if (src / "plain.log").exists() else None
if m:
    alg = int(m.group(1))
plain = (src / "plain.log").read_text().strip().splitlines()[-2:] if (src / "plain.log").exists() else []

rows, rd_total, wr_total, launches = [], 0.0, 0.0, 0
for (name, grid), vals in sorted(fetch.items(), key=lambda kv: -sum(kv[1])):
    if "stream_read_kernel" in name or "fill_" in name:
        continue
    per_step = round(len(vals) / total_steps)
    if per_step < 1 or abs(len(vals) - per_step * total_steps) > 2:
        continue  # prefill, create-time repacking, one-off launches
    med = sorted(vals)[len(vals) // 2] * (factor or 1.0)
    wv = write.get((name, grid), [])
    wmed = sorted(wv)[len(wv) // 2] if wv else 0.0
    rows.append({"kernel": name, "grid_threads": grid, "launches_per_step": per_step, "hbm_read_bytes_per_launch": int(med),
                 "hbm_write_bytes_per_launch": int(wmed)})
    rd_total += per_step * med
    wr_total += per_step * wmed
    launches += per_step
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/lab/engine_step_lab: the engine's own decode step through the C ABI, "
                 "Qwen3-4B-shaped random W4 weights, no Python in the process",
       "context_tokens": context, "batch": batch, "steps_in_pass": total_steps,
       "fetch_size_correction": round(factor, 4) if factor else None,
       "fetch_size_calibration": "stream_read_kernel, 4 x 1 GiB of known reads in the same process",
       "launches_per_step": launches,
       "step_hbm_read_bytes": int(rd_total), "step_hbm_write_bytes": int(wr_total),
       "step_algorithmic_bytes": alg,
       "hbm_over_algorithmic": round((rd_total + wr_total) / alg, 4) if alg else None,
       "uncounted_run": plain,
       "per_kernel": rows}
(src / "traffic_engine.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
