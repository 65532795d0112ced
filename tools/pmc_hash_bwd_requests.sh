#!/bin/bash
# Memory-side request counters of the hash backward (atomics are priced per request): separate --pmc passes, kernel-trace only.
# usage: bash tools/pmc_hash_bwd_requests.sh <out dir> "<regime> [opt=val ...]" ...
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=$1; shift; mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  for set in "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_ATOMIC_sum TCC_EA0_RDREQ_sum WRITE_SIZE"; do
    tag=c${i}_$(echo $set | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/hash_bwd_one.py $cfg > $OUT/$tag.log 2>&1
    echo "$cfg | $set rc=$?"
  done
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out, cfgs = sys.argv[1], sys.argv[2:]
for i, cfg in enumerate(cfgs, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/c{i}_*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'hash_bwd' in k:
                name = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
                agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    print("==", cfg)
    for k, d in agg.items():
        print("  ", k)
        for c, v in sorted(d.items()):
            print(f"      {c:28s} {sum(v)/len(v):16.0f}")
PY
