#!/bin/bash
# runtime knob: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) — 464 dependent launches per step, each fetches 0.2 - 1 KiB of arguments
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4; do
  echo "== unset"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
echo "== launch floor, unset"; timeout 300 python tools/launch_floor_probe.py 2>&1 | grep -v amdgpu.ids | head -6
echo "== launch floor, =1"; HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/launch_floor_probe.py 2>&1 | grep -v amdgpu.ids | head -6
