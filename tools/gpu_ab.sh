#!/bin/bash
# generic same-box A/B: default library vs every variant under lib/variants/ — parity tests on the DEFAULT, then microbench sections
# and bench.py interleaved.  usage: gpu_ab.sh "<pytest args>" "<microbench sections>" "<grep pattern>" [bench rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ -n "$1" ]; then eval "timeout 900 python -m pytest $1 -x -q -m gpu" 2>&1 | tail -3; fi
for i in 1 2; do
  echo "== default"; timeout 300 python tools/microbench.py $2 2>&1 | grep -E "$3"
  for v in cleantransformer_amd/lib/variants/*/; do n=$(basename $v); echo "== $n"; CTMI_LIB_PATH=$PWD/$v/libctmi355.so timeout 300 python tools/microbench.py $2 2>&1 | grep -E "$3"; done
done
for i in $(seq 1 ${4:-3}); do
  echo "== bench default"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  for v in cleantransformer_amd/lib/variants/*/; do n=$(basename $v); echo "== bench $n"; CTMI_LIB_PATH=$PWD/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
done
