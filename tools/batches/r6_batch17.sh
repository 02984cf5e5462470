#!/bin/bash
# what does the dGELU side input cost, and why: fetches from a 64 KiB window (L2 hits) / tile-major contiguous pieces / no fetch at all (timing variants, wrong values)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== default"; timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"
  for v in auxhot auxtile auxnone; do echo "== $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"; done
done
