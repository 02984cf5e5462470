#!/bin/bash
# energy per launch of the hot kernels (tools/energy_probe.py): ours vs the vendor GEMM, attention, HBM-bound kernels; zero operands as the floor for the GEMMs
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b32; rm -rf $O; mkdir -p $O
timeout 200 python tools/energy_probe.py gemm vendor attn hbm --seconds 1.2 2>&1 | grep -v amdgpu.ids | tee $O/energy.txt
timeout 60 python tools/energy_probe.py gemm --zeros --seconds 1.0 2>&1 | grep -E "zero operands|idle" | tee $O/energy_zeros.txt
