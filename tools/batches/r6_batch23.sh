#!/bin/bash
# h->4h forward + GELU: what the pre-activation store and the activation arithmetic cost (timing variants)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== default"; timeout 300 python tools/microbench.py epi 2>&1 | grep "h4h fwd"
  for v in noaux nomath; do echo "== $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python tools/microbench.py epi 2>&1 | grep "h4h fwd"; done
done
