#!/bin/bash
# what a dependent launch costs on the device side, from a stream and from a hipGraph
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/launch_floor_probe.py 2>&1 | grep -v amdgpu.ids
