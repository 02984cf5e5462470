#!/bin/bash
# run a microbench section with the default library and every variant under cleantransformer_amd/lib/variants/
# usage: gpu_variants.sh "<microbench sections>" "<grep pattern>" [pytest selection run with the default lib first]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ -n "${3:-}" ]; then eval "timeout 900 python -m pytest $3 -x -q -m gpu" 2>&1 | tail -8; fi
echo "== default"; timeout 300 python tools/microbench.py $1 2>&1 | grep -E "$2"
for v in cleantransformer_amd/lib/variants/*/; do n=$(basename $v); echo "== $n"; CTMI_LIB_PATH=$PWD/$v/libctmi355.so timeout 300 python tools/microbench.py $1 2>&1 | grep -E "$2"; done
