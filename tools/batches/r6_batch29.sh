#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python tools/ce_padded_probe.py 2>&1 | grep -v amdgpu.ids
