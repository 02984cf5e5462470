#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python tools/attn_w32_check.py > $O/check.log 2>&1; echo "check rc=$?" | tee -a $O/check.log
tail -30 $O/check.log
