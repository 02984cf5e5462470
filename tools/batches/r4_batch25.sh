#!/bin/bash
# LayerNorm backward: guard-free loop (counted waits) + next-row prefetch over two register sets.  parity, microbench, step A/B
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b25; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "layernorm or ln or block" 2>&1 | tail -3 | tee $O/tests.txt
for v in default lnpf0 k2off; do
  echo "== microbench ln $v" >> $O/micro.txt
  if [ $v = default ]; then timeout 120 python tools/microbench.py ln 2>&1 | tail -4 >> $O/micro.txt; else CTMI_LIB_PATH=$VD/$v/libctmi355.so timeout 120 python tools/microbench.py ln 2>&1 | tail -4 >> $O/micro.txt; fi
done
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench lnpf0" | tee -a $O/bench.txt; CTMI_LIB_PATH=$VD/lnpf0/libctmi355.so $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
cat $O/micro.txt
