"""Join the per-pass counter CSVs of tools/lab/pmc_gemm.sh: per kernel (name, grid) the mean of every counter over its dispatches."""
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "qmm_mfma" not in name and "splitk" not in name:
            continue
        key = (name.split("(")[0][-60:], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key)
    for c, v in cs.items():
        print(f"    {c:42s} {sum(v) / len(v):16.1f}   (n={len(v)})")
