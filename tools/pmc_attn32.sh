#!/bin/bash
# PMC passes over the 256-row attention kernels (tools/attn_w32_check.py time, B=8 S=1024 hd=64); counters only, one group per pass.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_attn32; rm -rf $OUT; mkdir -p $OUT
export W32_CASES=${W32_CASES:-0} W32_PATHS=3
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python tools/attn_w32_check.py time > $OUT/g$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_attn32/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"attn\w*_kernel<[^>]*>", r["Kernel_Name"])       # (the names start with "void (anonymous namespace)::")
        if m:
            acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    mf = d.get("SQ_INSTS_MFMA", 0)
    if mf and "SQ_BUSY_CU_CYCLES" in d and "SQ_WAVE_CYCLES" in d:
        print(f"\n{k}: per MFMA: {d['SQ_INSTS_VALU']/mf:.1f} VALU ({d.get('SQ_INSTS_VALU_TRANS_F32',0)/mf:.1f} transcendental, {d.get('SQ_INSTS_VALU_CVT',0)/mf:.1f} cvt, "
              f"{d.get('SQ_INSTS_VALU_INT32',0)/mf:.1f} int), {d['SQ_INSTS_SALU']/mf:.1f} SALU, {d['SQ_INSTS_LDS']/mf:.1f} LDS; "
              f"MFMA pipe busy {100*d['SQ_VALU_MFMA_BUSY_CYCLES']/(4*d['SQ_BUSY_CU_CYCLES']):.0f} % of CU-busy SIMD cycles; wave cycles: "
              f"{100*d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES']:.0f} % issuing, {100*d['SQ_WAIT_INST_ANY']/d['SQ_WAVE_CYCLES']:.0f} % issue stalls, "
              f"{100*d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES']:.0f} % waitcnt/barrier")
    else:
        print("\n" + k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}")
PY
