#!/bin/bash
# does the two-stages-per-phase schedule save energy or only time?  4h->h / h->4h GEMMs, shipped build vs -DCTMI_PP_K2=0, joules per launch
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b35; rm -rf $O; mkdir -p $O
echo "== shipped (K2)" | tee $O/energy_k2.txt; EP_ONLY=4hh timeout 40 python tools/energy_probe.py gemm --seconds 0.9 2>&1 | grep -E "4hh" | tee -a $O/energy_k2.txt
echo "== -DCTMI_PP_K2=0" | tee -a $O/energy_k2.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/k2off/libctmi355.so EP_ONLY=4hh timeout 40 python tools/energy_probe.py gemm --seconds 0.9 2>&1 | grep -E "4hh" | tee -a $O/energy_k2.txt
