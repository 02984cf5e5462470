#!/bin/bash
# grouped weight gradients, K-halves through partial slabs + a second launch: parity, alone-times per mode, step A/B (two streams and one)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wgrad_grouped.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests_grouped.txt
timeout 600 python -m pytest tests/test_gpu_block.py -x -q -m gpu -k "one_call or gpt2_spelling" 2>&1 | tail -3 | tee $O/tests_block.txt
for m in 1 2 3; do CTMI_WGRAD_GROUP=$m timeout 120 python tools/microbench.py wgroup 2>&1 | grep -E "grouped|per-product" | tee -a $O/wgroup_modes.txt; done
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
run() { n=$1; shift; echo -n "== bench [$n] " | tee -a $O/ab.txt
  (env "$@" timeout 200 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['timing'].get('power_while_stepping') or {}; print(d['ms_per_step'], 'loss', d.get('final_loss'), 'W', p.get('package_power_w_median'), 'MHz', p.get('sclk_mhz_median'))" 2>&1) | tee -a $O/ab.txt; }
for i in 1 2 3; do run group0 CTMI_WGRAD_GROUP=0; run group1 CTMI_WGRAD_GROUP=1; run group3 CTMI_WGRAD_GROUP=3; run group1_1stream CTMI_WGRAD_GROUP=1 CTMI_WGRAD_STREAM=0; run group3_1stream CTMI_WGRAD_GROUP=3 CTMI_WGRAD_STREAM=0; done
run group2 CTMI_WGRAD_GROUP=2
