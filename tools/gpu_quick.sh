#!/bin/bash
# quick GPU iteration: $1 = pytest selection ("" = none), $2 = microbench sections ("" = none), $3 = "bench" to run bench.py
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/quick; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || tail -20 $O/build.log
if [ -n "${1:-}" ]; then eval "timeout 1500 python -m pytest $1 -x -q -m gpu" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log; fi
if [ -n "${2:-}" ]; then timeout 600 python tools/microbench.py $2 2>&1 | grep -v amdgpu.ids | tee $O/microbench.txt; fi
if [ "${3:-}" = "bench" ]; then timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'lmhead', d['roofline']['achieved'], 'loss', d['final_loss'])"; fi
