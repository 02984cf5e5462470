#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4full; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json; python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['timing'])"
