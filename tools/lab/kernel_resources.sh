#!/bin/bash
# usage: tools/lab/kernel_resources.sh <file.hip> [outdir]  -- per-kernel VGPR / AGPR / scratch table + the .s under outdir (default /tmp/kres)
set -e
src=$(realpath "$1"); out=${2:-/tmp/kres}; mkdir -p "$out"; cd "$out"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 \
  -I"$(dirname "$src")" $KRES_FLAGS -Rpass-analysis=kernel-resource-usage -save-temps=obj -c "$src" -o "$out/$(basename "$src").o" 2>&1 | python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)};rows.append(cur);continue
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','TotalSGPRs'):
        m=re.search(k+r': (\d+)',l)
        if m and cur is not None: cur[k[:7]]=int(m.group(1))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['n']],capture_output=True,text=True).stdout.strip()
    print(f\"{n[:90]:90s} v {r.get('VGPRs'):>3} a {r.get('AGPRs'):>3} scratch {r.get('Scratch')}\")
"
