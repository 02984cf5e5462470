#!/bin/bash
# new tests (scaler registration, two-rank bench path after the probe changes), does the in-run power sampler cost anything?
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_amp.py tests/test_gpu_bloom.py -x -q -m gpu -k "scaler_registration or bench_two_rank or ddp_" 2>&1 | tail -6 | tee $O/tests.txt
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do for v in "" "--no-power-sampler"; do
  echo -n "== bench [$v] " | tee -a $O/sampler_ab.txt
  timeout 200 python $B $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_mean_wall'])" | tee -a $O/sampler_ab.txt
done; done
