#!/bin/bash
# whole-step hipGraph in bench.py: replayed vs eager, interleaved; then the full default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do
  echo "== eager"; python bench.py --no-graph --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== graph"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*\|"eager_ms_per_step": [0-9.]*\|"replays": [0-9]*' | tr '\n' ' '; echo
done
python bench.py --no-cpu-baseline > gpurun_out/r6b10_bench.json 2> gpurun_out/r6b10_bench.err; tail -1 gpurun_out/r6b10_bench.json | cut -c1-600; tail -3 gpurun_out/r6b10_bench.err
