#!/bin/bash
# 128-row attention kernels: ring depth / waves per workgroup revisited on the final build (variants: backward ring 3 stages, forward ring 3 stages, backward 8 waves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== default"; python tools/attn_w32_check.py time 2>&1 | grep "path=3" | grep -E "S=1024|hd=128" | head -3
  for v in bnst3 fnst3 bnw8; do echo "== $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python tools/attn_w32_check.py time 2>&1 | grep "path=3" | grep -E "S=1024|hd=128" | head -3; done
done
