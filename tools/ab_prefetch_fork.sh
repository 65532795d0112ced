#!/bin/bash
# Where should the side stream that marches the next step's rays fork off the captured step?  usage: bash tools/ab_prefetch_fork.sh <out>
OUT=$1; mkdir -p $(dirname $OUT)
for f in start hash_bwd; do
  export LSE_BENCH_PREFETCH_FORK=$f
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
g = lambda e: (e['graphed']['ms_per_step'], e['graphed']['marcher_prefetched']['ms_per_step'])
mm = d['m_march_graphed']
print('fork=$f  m_march %.3f -> %.3f   default_config %.3f -> %.3f   cfg2 %.3f -> %.3f' % ((mm['ms_per_step'], mm['marcher_prefetched']['ms_per_step']) + g(d['default_config']) + g(d['cfg2_composition'])))" | tee -a $OUT
done
