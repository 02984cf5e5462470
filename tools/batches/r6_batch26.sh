#!/bin/bash
# round-6 evidence on the final tree (tools/collect_profiles.sh) + the kernel table inputs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROUND=r06 bash tools/collect_profiles.sh 2>&1 | tail -5
