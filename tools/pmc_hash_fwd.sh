#!/bin/bash
# Counters of the hash forward kernel on the metric-size microbenchmark (separate passes, kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/pmc_hash_fwd; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/bench_kernels.py hash_fwd > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_hash_fwd/*/*/*counter_collection.csv'):
    byd = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'hash_fwd' in r['Kernel_Name']:
            byd[(r['Counter_Name'], r['Dispatch_Id'])] += float(r['Counter_Value'])
    for (c, d), v in byd.items():
        agg['hash_fwd'][c].append(v)
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:36s} {sum(v)/len(v):18.0f}")
PY
