#!/bin/bash
# bench.py under different environment settings, interleaved: gpu_env_ab.sh rounds "VAR=val VAR2=val" "..." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$1; shift
for r in $(seq 1 $R); do
  for e in "$@"; do
    env $e timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', 'ms_per_step', d['ms_per_step'], 'lmhead', d['roofline']['achieved'], 'loss', d['final_loss'])"
  done
done
