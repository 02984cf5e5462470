#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python tools/attn_w32_check.py > $O/check.log 2>&1; echo "check rc=$?" | tee -a $O/check.log
grep -v "^OK" $O/check.log | tail -16
timeout 300 python tools/attn_w32_timing.py 2>&1 | tail -12
