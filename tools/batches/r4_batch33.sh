#!/bin/bash
# energy per launch of the layer weight gradients under the tile / split rules that were ranked by step time in r04_wgrad_rule.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b33; rm -rf $O; mkdir -p $O
for e in "X=1" "CTMI_WGRAD_ITEMS4=256" "CTMI_WGRAD_RULE=1" "CTMI_WGRAD_RULE=0" "CTMI_WGRAD_NOSPLIT=1"; do
  echo "== $e" | tee -a $O/energy_wgrad.txt
  env $e timeout 100 python tools/energy_probe.py wgrad --seconds 1.0 2>&1 | grep -E "wgrad" | tee -a $O/energy_wgrad.txt
done
