#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3k; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/blaslt_probe.py > $O/probe.log 2>&1
python - <<'PY' > gpurun_out/r3k/vendor_kernels.txt
import csv, glob
f = glob.glob("gpurun_out/r3k/prof/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Calls"], f'{float(r["AverageNs"])/1e3:.1f}us', r["Name"])
PY
find $O -name "*_kernel_trace.csv" -delete
cat $O/probe.log | tail -20
