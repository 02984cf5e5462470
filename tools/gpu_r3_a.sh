#!/bin/bash
# round 3, call A: parity + timing of the 256-row attention kernels, the attention tests, one bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a; mkdir -p $O
timeout 600 python tools/attn_w32_check.py > $O/check.log 2>&1; echo "check rc=$?" | tee -a $O/check.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/bench.log
CTMI_ATTN_W32=0 timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_old.log 2>&1
tail -40 $O/check.log; tail -5 $O/pytest.log; tail -3 $O/bench.log | cut -c1-400; tail -2 $O/bench_old.log | cut -c1-300
