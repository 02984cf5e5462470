#!/bin/bash
# longer randomised parity sweeps on the final build (attention 128-row kernels incl. head_dim 128 backward; three seeds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b24; rm -rf $O; mkdir -p $O
for seed in 11 12 13; do
  timeout 600 python tools/attn_w32_fuzz.py 150 $seed 2>&1 | grep -v amdgpu > $O/fuzz_$seed.txt
  echo "seed $seed: $(grep -c '^OK' $O/fuzz_$seed.txt) OK, $(grep -c '^BAD' $O/fuzz_$seed.txt) BAD, hd=128 cases: $(grep -c 'hd=128' $O/fuzz_$seed.txt); $(tail -1 $O/fuzz_$seed.txt)"
done | tee $O/summary.txt
