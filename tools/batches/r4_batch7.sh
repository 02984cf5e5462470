#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b7; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "" "CTMI_BLK_NOJOIN=1"; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
