#!/bin/bash
# the new tile rule (one full round of 256x256 tiles + long K -> 256-row tile) in the library: 7B1 geometry new vs CTMI_TILE3_MIN=100000-style old rule is gone, so: parity of the 7B1-geometry tests, the bench twice, Bloom-560M step (unaffected shapes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_bloom.py -x -q -m gpu -k "7b1 or c5 or geometry or hd128 or head_dim" 2>&1 | tail -2
for i in 1 2; do echo "== 7b1"; timeout 400 python tools/bench_bloom7b1.py 2>/dev/null | tee -a gpurun_out/r6b36_7b1.txt | grep -o '"ms_per_step": [0-9.]*'; done
echo "== bloom-560m"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
echo "== gpt2"; timeout 300 python tools/bench_gpt2.py 2>/dev/null | tee gpurun_out/r6b36_gpt2.txt | grep -o '"ms_per_step": [0-9.]*'
