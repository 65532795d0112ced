#!/bin/bash
# rocprofv3 kernel stats of a short bench run: tools/prof_quick.sh <tag>
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-context > $OUT/rocprof.log 2>&1; rc=$?
echo "rocprof rc=$rc"
F=$(find $OUT/prof -name "*kernel_stats*" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY
