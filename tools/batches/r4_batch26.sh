#!/bin/bash
# LayerNorm forward: persistent launch, weight/bias in registers, two rows per trip.  parity, microbench (256 / 512 / 1024 workgroups, old kernel), step A/B
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b26; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "layernorm or ln or block" 2>&1 | tail -3 | tee $O/tests.txt
for v in default lnf0 lnf256 lnf1024; do
  echo "== microbench ln $v" >> $O/micro.txt
  if [ $v = default ]; then timeout 120 python tools/microbench.py ln 2>&1 | grep -E "fwd|bwd" >> $O/micro.txt; else CTMI_LIB_PATH=$VD/$v/libctmi355.so timeout 120 python tools/microbench.py ln 2>&1 | grep -E "fwd|bwd" >> $O/micro.txt; fi
done
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench lnf0" | tee -a $O/bench.txt; CTMI_LIB_PATH=$VD/lnf0/libctmi355.so $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
cat $O/micro.txt
