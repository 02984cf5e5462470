#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b14; rm -rf $O; mkdir -p $O
V=$PWD/cleantransformer_amd/lib/variants/prio1/libctmi355.so
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3 4 5; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench prio1" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
