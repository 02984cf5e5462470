#!/bin/bash
# Bloom-7B1 geometry on one GPU: [T,H] outputs (256 tiles of 256x256 = one full round, K = 4096 ... 16384) on the 256-row ping-pong tile (CTMI_TILE3_MIN=256) vs the 128-row tile (default 350)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== default"; timeout 400 python tools/bench_bloom7b1.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== tile3_min 256"; CTMI_TILE3_MIN=256 timeout 400 python tools/bench_bloom7b1.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
