#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in 0 1 2 3 4 7; do echo "== CTMI_ATTN_DBG=$d"; CTMI_ATTN_DBG=$d timeout 120 python tools/microbench.py attn 2>&1 | grep -E "fwd|bwd"; done
MB_KTRACE=1 timeout 120 python tools/microbench.py attn 2>&1 | grep -E "attn_|us avg"
