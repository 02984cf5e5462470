#!/bin/bash
# LayerNorm forward: rows per wave (grid cap) — 2048 workgroups = one row per wave at [8192,1024] (default) vs 1024 / 512 / 256 (2 / 4 / 8 rows per wave: reads of the next row overlap the stores of the last)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for g in 2048 1024 512 256; do echo "== grid $g"; CTMI_LN_FWD_GRID=$g timeout 200 python tools/microbench.py ln 2>&1 | grep "fwd"; CTMI_LN_FWD_GRID=$g timeout 200 python tools/launch_floor_probe.py 2>&1 | grep "LayerNorm fwd"; done
for i in 1 2 3; do
  for g in 2048 1024 512; do echo "== bench grid $g"; CTMI_LN_FWD_GRID=$g python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
done
