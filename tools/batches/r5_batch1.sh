#!/bin/bash
# round 5, first GPU call: the grouped weight-gradient launch — parity, then A/B of the step: the round-4 tree (ab/r4: package + library as shipped
# at 109d69e) vs this tree with CTMI_WGRAD_GROUP=0 (stripped kernels, per-product weight gradients) / 1 (grouped, default) / 2 / 3
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wgrad_grouped.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests_grouped.txt
timeout 600 python -m pytest tests/test_gpu_block.py -x -q -m gpu -k "one_call or gpt2_spelling" 2>&1 | tail -5 | tee $O/tests_block.txt
timeout 120 python tools/microbench.py wgroup 2>&1 | grep -v amdgpu.ids | tee $O/wgroup.txt
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
run() {  # name, dir, env...
  n=$1; d=$2; shift 2
  echo -n "== bench [$n] " | tee -a $O/ab.txt
  (cd $d && env "$@" timeout 200 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'loss', d.get('final_loss'))" 2>&1) | tee -a $O/ab.txt
}
for i in 1 2 3; do
  run r4 $R/ab/r4 X=1
  run group0 $R CTMI_WGRAD_GROUP=0
  run group1 $R CTMI_WGRAD_GROUP=1
  if [ $i -le 2 ]; then run group2 $R CTMI_WGRAD_GROUP=2; run group3 $R CTMI_WGRAD_GROUP=3; fi
done
