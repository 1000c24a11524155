cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/prefill_probe.py --prompt 1024 --chunk 128 --repeat 3 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf128 -o pf --output-format csv -- python $R/tools/prefill_probe.py --prompt 1024 --chunk 128 --repeat 1 > $R/gpurun_out/pf128.log 2>&1
tail -1 $R/gpurun_out/pf128.log
python - <<'P'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pf128/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]: print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e3,1), round(float(r['AverageNs'])/1e3,2))
P
