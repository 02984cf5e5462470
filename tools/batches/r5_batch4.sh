#!/bin/bash
# the new default (grouped weight gradients, whole backward on one stream) — step A/B against the forced alternatives, the two secondary
# configurations, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b4; rm -rf $O; mkdir -p $O
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
run() { n=$1; shift; echo -n "== bench [$n] " | tee -a $O/ab.txt
  (env "$@" timeout 200 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['timing'].get('power_while_stepping') or {}; print(d['ms_per_step'], 'loss', d.get('final_loss'), 'W', p.get('package_power_w_median'), 'MHz', p.get('sclk_mhz_median'))" 2>&1) | tee -a $O/ab.txt; }
for i in 1 2 3; do run default X=1; run per_product_2streams CTMI_WGRAD_GROUP=0; run grouped_2streams CTMI_WGRAD_STREAM=1; done
echo "== gpt2-medium default / per-product" | tee -a $O/secondary.txt
timeout 200 python tools/bench_gpt2.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gpt2 default', d.get('ms_per_step'))" | tee -a $O/secondary.txt
CTMI_WGRAD_GROUP=0 timeout 200 python tools/bench_gpt2.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gpt2 group0', d.get('ms_per_step'))" | tee -a $O/secondary.txt
timeout 300 python tools/bench_bloom7b1.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('7b1 default', d.get('ms_per_step'))" | tee -a $O/secondary.txt
CTMI_WGRAD_GROUP=0 timeout 300 python tools/bench_bloom7b1.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('7b1 group0', d.get('ms_per_step'))" | tee -a $O/secondary.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/tests_all.txt
