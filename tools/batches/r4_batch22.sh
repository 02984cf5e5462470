#!/bin/bash
# which of the counted waits makes the LAND=1 build 2.6 ms faster: the prologue's or the drain's?
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b22; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  for v in land1 landp1 landg1; do
    echo "== bench $v" | tee -a $O/bench.txt; CTMI_LIB_PATH=$VD/$v/libctmi355.so $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
