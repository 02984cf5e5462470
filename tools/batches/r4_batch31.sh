#!/bin/bash
# is the step power-capped?  sample socket power / sclk every ~50 ms while 300 steps run
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b31; rm -rf $O; mkdir -p $O
( for i in $(seq 1 400); do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level" | tr '\n' ' '; echo; sleep 0.03; done ) > $O/smi_samples.txt 2>&1 &
SMI=$!
python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 300 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee $O/bench.txt
kill $SMI 2>/dev/null
wc -l $O/smi_samples.txt
