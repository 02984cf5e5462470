#!/bin/bash
# alone-times of the grouped launch per split mode; more interleaved pairs of the step: per-product (0) vs grouped unsplit (3)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b2; rm -rf $O; mkdir -p $O
for m in 1 2 3; do CTMI_WGRAD_GROUP=$m timeout 120 python tools/microbench.py wgroup 2>&1 | grep -E "grouped|per-product" | tee -a $O/wgroup_modes.txt; done
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
run() { n=$1; shift; echo -n "== bench [$n] " | tee -a $O/ab.txt
  (env "$@" timeout 200 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'loss', d.get('final_loss'), 'W', d['timing'].get('power_while_stepping'))" 2>&1) | tee -a $O/ab.txt; }
for i in 1 2 3 4; do run group0 CTMI_WGRAD_GROUP=0; run group3 CTMI_WGRAD_GROUP=3; done
run group0_1stream CTMI_WGRAD_GROUP=0 CTMI_WGRAD_STREAM=0
run group3_1stream CTMI_WGRAD_GROUP=3 CTMI_WGRAD_STREAM=0
run group1_1stream CTMI_WGRAD_GROUP=1 CTMI_WGRAD_STREAM=0
