#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b14; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/attn_w32_check.py time > $O/run.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r5b14/prof/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'attn' in r['Name']: print(f"{float(r['AverageNs'])/1e3:9.1f} us avg  {int(r['Calls']):5d} calls  {float(r['MinNs'])/1e3:8.1f} min {float(r['MaxNs'])/1e3:8.1f} max  {r['Name'][:90]}")
PY
