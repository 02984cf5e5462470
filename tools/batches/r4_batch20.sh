#!/bin/bash
# why did the K2 build with the corrected waits measure 37.0 ms where the first K2 build measured 34.3?  same box: default, first build, LAND=1, K2 off
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b20; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  for v in k2racy land1 k2off; do
    echo "== bench $v" | tee -a $O/bench.txt; CTMI_LIB_PATH=$VD/$v/libctmi355.so $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
