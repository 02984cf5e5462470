#!/bin/bash
# LM-head GEMMs: persistent launch (default) vs one workgroup per tile (CTMI_GEMM_PERSIST=0) — time, and FETCH_SIZE of the weight gradient
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b21; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  echo "== persistent"; MB_ONLY=lm_head python tools/microbench.py gemm 2>&1 | grep lm_head
  echo "== per tile"; CTMI_GEMM_PERSIST=0 MB_ONLY=lm_head python tools/microbench.py gemm 2>&1 | grep lm_head
done | tee $O/time.txt
for m in 1 0; do
  CTMI_GEMM_PERSIST=$m MB_ONLY=lm_head timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$m -- python tools/microbench.py gemm > $O/pmc_$m.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for m in (1,0):
    f=glob.glob(f'gpurun_out/r5b21/pmc_{m}/*/*counter_collection.csv')
    if not f: print('no csv', m); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name']!='FETCH_SIZE': continue
        k=r['Kernel_Name'][:70]; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items():
        if 'gemm' in k: print(f"persist={m} {k}: {n} dispatch-rows, FETCH (x2 per gfx950 note, 64 B units -> GB per launch) = {v*64*2/1e9/ max(1,n/8) :.2f} GB?  raw sum {v:.3e}")
PY
