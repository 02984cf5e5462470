#!/bin/bash
# the cheapest weight-gradient rule in joules (256x256 unsplit) re-measured in the step on the final build
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b34; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2; do
  for e in "X=1" "CTMI_WGRAD_NOSPLIT=1" "CTMI_WGRAD_ITEMS4=96"; do
    echo "== bench $e" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
