#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python tools/attn_w32_check.py > $O/check.log 2>&1; echo "check rc=$?" | tee -a $O/check.log
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/bench.log
tail -30 $O/check.log; tail -3 $O/bench.log | cut -c1-300
