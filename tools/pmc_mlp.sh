#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export PMC_OUT=${PMC_OUT:-gpurun_out/pmc_mlp}; OUT=$PMC_OUT; mkdir -p $OUT

rocprofv3 -L > $OUT/counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-context --no-atomic-floor > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
import os
for f in glob.glob(os.environ.get("PMC_OUT", "gpurun_out/pmc_mlp") + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mlp_' in k or 'hash_' in k:
            name = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}")
PY
