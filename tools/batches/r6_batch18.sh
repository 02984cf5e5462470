#!/bin/bash
# dGELU data gradient on every tile family (CTMI_GEMM_TILE forces it): 0 = 128x128 x3/CU, 1 = 256x128 x2/CU, 2 = 256x256 free-running, 3 = 256x256 ping-pong, 4 = 128x256 ping-pong
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for t in 4 0 1 2 3; do echo "== tile $t"; CTMI_GEMM_TILE=$t timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad\|h4h fwd"; done
