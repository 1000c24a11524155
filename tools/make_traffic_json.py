"""profiles/traffic.json from the rocprofv3 --pmc passes over tools/lab/gemv_lab (tools/lab/pmc_gemv.sh).

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts a wide coalesced stream at
exactly half its bytes (guides/MI355X_MICROARCH.md, HBM section): the factor is re-derived here from stream_kernel,
whose byte count is known, and applied to the GEMV kernels (never from a GEMV kernel itself: that would make
its own read/algorithmic ratio 1.000 by construction)."""
import collections, csv, json, sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / "pmc_gemv"

def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}

fetch = load(src / "fetch" / "lab_counter_collection.csv")
write = load(src / "write" / "lab_counter_collection.csv")
shapes = {"qkv": (6144, 2560, "tl::qmv3_kernel<1, 2, 4, 1, 0, 10, 0>", 192 * 256), "o": (2560, 4096, "tl::qmv3_kernel<1, 4, 4, 0, 1, 8, 0>", 160 * 256),
          "gate_up": (19456, 2560, "tl::qmv3_kernel<1, 4, 4, 1, 2, 5, 0>", 1216 * 256), "down": (2560, 9728, "tl::qmv3_kernel<1, 8, 8, 0, 1, 10, 0>", 160 * 512),
          "lm_head": (151936, 2560, "tl::qmv3_kernel<1, 2, 4, 1, 0, 10, 0>", 4748 * 256)}
# calibration on stream_kernel<4, true> (tools/lab/gemv_lab.hip, pmc mode): a pure 16 B/lane non-temporal stream whose
# byte count is known exactly -- per shape 2 grids x iters dispatches of K*N/2 bytes each (the lab's own loop bounds)
known = 0.0
for K, N in ((6144, 2560), (2560, 4096), (19456, 2560), (2560, 9728), (151936, 2560)):
    wbytes = K * N // 2 + K * (N // 128) * 4
    copies = max(1, min(40, (700 << 20) // wbytes + 1))
    known += 2 * max(3 * copies, 60) * (K * N // 2)
seen = 0.0
for r in csv.DictReader(open(src / "fetch" / "lab_counter_collection.csv")):
    if "stream_kernel" in r["Kernel_Name"]:
        seen += float(r["Counter_Value"]) * 1024
factor = known / seen
import datetime
stamp = datetime.datetime.utcfromtimestamp((src / "fetch" / "lab_counter_collection.csv").stat().st_mtime).strftime("%Y-%m-%d")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/lab/gemv_lab pmc (separate passes)",
       "collected": "passes merged " + stamp + " (UTC), lab linked against that day's csrc/build/qmv3.o", "fetch_size_correction": round(factor, 4), "fetch_size_calibration": "stream_kernel<4, true>, %.3f GB of known reads" % (known / 1e9),
       "per_kind": {}}
tot_bytes, tot_alg, launches = 0.0, 0.0, 0
for name, (K, N, kern, grid) in shapes.items():
    alg = K * N / 2 + K * (N // 128) * 4
    rd = fetch[(kern, grid)] * 1024 * factor
    wr = write.get((kern, grid), 0.0) * 1024
    n = 1 if name == "lm_head" else 36
    out["per_kind"][name] = {"kernel": kern, "algorithmic_bytes": int(alg), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
                             "read_over_algorithmic": round(rd / alg, 4)}
    tot_bytes += n * (rd + wr)
    tot_alg += n * alg
    launches += n
out["qmv_hbm_bytes_per_launch"] = int(tot_bytes / launches)
out["qmv_algorithmic_bytes_per_launch"] = int(tot_alg / launches)
(root / "profiles" / "traffic.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
