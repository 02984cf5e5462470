#!/bin/bash
# bench.py (short) under the default library and every variant; $1 = rounds (default 2), interleaved so box drift cancels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-2}
run() { CTMI_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], 'lmhead', d['roofline']['achieved'], 'loss', d['final_loss'])"; }
for r in $(seq 1 $R); do
  run default ""
  for v in cleantransformer_amd/lib/variants/*/; do n=$(basename $v); run $n $PWD/$v/libctmi355.so; done
done
