#!/bin/bash
# AdamW: does the relative placement of the p / g / m / v streams matter for the HBM rate?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/adamw_probe.py 2>&1 | grep -v amdgpu.ids
