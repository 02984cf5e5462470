#!/bin/bash
# do the guarded (edge-tile) epilogue paths cost anything by being IN the ping-pong kernels (code size, instruction fetch at every launch)?  default vs -DCTMI_EXP_NO_EDGE=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_ab.sh "" "gemm" "qkv|dense|h4h|4hh" 4 2>&1 | grep -v wgrad
