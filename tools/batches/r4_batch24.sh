#!/bin/bash
# weight-gradient work-item rule re-tuned on the K2 build (env only), same box
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b24; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "X=1" "CTMI_WGRAD_ITEMS4=64" "CTMI_WGRAD_ITEMS4=192" "CTMI_WGRAD_ITEMS4=256" "CTMI_WGRAD_RULE=1"; do
    echo "== bench $e" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
