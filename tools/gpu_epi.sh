#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== default"; python tools/microbench.py epi 2>&1 | grep -v amdgpu
echo "== MUL on tile 3"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/mult3/libctmi355.so python tools/microbench.py epi 2>&1 | grep "dgrad"
for t in 3 4; do echo "== CTMI_GEMM_TILE=$t"; CTMI_GEMM_TILE=$t python tools/microbench.py epi 2>&1 | grep -v amdgpu; done
