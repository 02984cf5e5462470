#!/bin/bash
# head_dim 128 attention backward on the 128-row kernels (dQ, then dV and dK in two passes): parity, timing against the general kernels, 7B1 geometry step
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b15; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "128row or bloom_attention" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python tools/attn_w32_check.py time 2>&1 | grep -v amdgpu | tee $O/attention_paths.txt
timeout 300 python tools/attn_w32_fuzz.py 40 1 2>&1 | tail -3 | tee $O/fuzz.txt
timeout 300 python tools/bench_bloom7b1.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('7b1 w32 bwd', d.get('ms_per_step'))" | tee $O/7b1.txt
CTMI_ATTN_W32=1 timeout 300 python tools/bench_bloom7b1.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('7b1 general bwd', d.get('ms_per_step'))" | tee -a $O/7b1.txt
timeout 200 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('560m', d['ms_per_step'])" | tee -a $O/7b1.txt
