#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b11; rm -rf $O; mkdir -p $O
timeout 120 tools/probes/mfma_energy.bin 0.5 2>&1 | tee $O/mfma_energy_sigma0.5.txt
timeout 120 tools/probes/mfma_energy.bin 0.02 2>&1 | tee $O/mfma_energy_sigma0.02.txt
