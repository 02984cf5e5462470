#!/bin/bash
# where the per-tile overhead of the 256x256 forward tile goes: default vs no epilogue vs epilogue without its stores (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
for v in default noepi nostore; do
  echo "== $v"
  if [ $v = default ]; then KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep -v amdgpu; else KS_LM=1 CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python tools/ksweep_probe.py 2>&1 | grep -v amdgpu; fi
done; done
