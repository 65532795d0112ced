#!/bin/bash
# SQ counters of the hash backward kernel on the metric-size microbenchmark (separate passes, kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/pmc_hash_bwd; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT" "SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/bench_kernels.py hash_bwd > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_hash_bwd/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'hash_bwd' in k:
            name = k.split('(')[0].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:60]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}")
PY
