#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b9; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "" "CTMI_WGRAD_NOSPLIT=2" "CTMI_GEMM_SPLIT=1" "CTMI_GEMM_SPLIT=4"; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
