#!/bin/bash
# Bloom-7B1 geometry: per-product weight gradients on the compute stream (CTMI_WGRAD_STREAM=0) vs on the side stream (default at this geometry)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== side stream (default)"; timeout 400 python tools/bench_bloom7b1.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== one stream"; CTMI_WGRAD_STREAM=0 timeout 400 python tools/bench_bloom7b1.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
