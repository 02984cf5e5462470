#!/bin/bash
# start-stagger of the workgroups of many-tiles-per-workgroup launches (LM-head forward): microbench + step A/B
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b28; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
for v in default stag4k stag12k; do
  echo "== microbench $v" >> $O/micro.txt
  if [ $v = default ]; then MB_ONLY=lm_head timeout 300 python tools/microbench.py gemm 2>&1 | grep lm_head >> $O/micro.txt; else CTMI_LIB_PATH=$VD/$v/libctmi355.so MB_ONLY=lm_head timeout 300 python tools/microbench.py gemm 2>&1 | grep lm_head >> $O/micro.txt; fi
done
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench stag4k" | tee -a $O/bench.txt; CTMI_LIB_PATH=$VD/stag4k/libctmi355.so $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|bench.py: the loss.*' | tee -a $O/bench.txt
done
cat $O/micro.txt
