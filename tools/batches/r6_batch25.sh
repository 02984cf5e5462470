#!/bin/bash
# the driver's commands on the current tree: the whole -m gpu suite, smoke, default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b25
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r6b25/bench_default.json 2> gpurun_out/r6b25/bench_default.err; tail -1 gpurun_out/r6b25/bench_default.json | cut -c1-400
