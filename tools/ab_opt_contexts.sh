#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# A/B one run-time option of the library on the headline step AND the real-regime compositions (bench.py context keys):
# usage: bash tools/ab_opt_contexts.sh <out file> <OPTION_NAME> v1 v2 ...
OUT=$1; NAME=$2; shift 2; mkdir -p $(dirname $OUT)
for v in "$@"; do
  export LSE_OPT_$NAME=$v
  b=$(timeout -k 10 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
g=lambda key: d[key]['graphed'].get('marcher_prefetched', d[key]['graphed'])['ms_per_step']
print('headline %.3f (hash_bwd %.3f)  m_packed %.3f  default_config %.3f (hash_bwd %.3f)  cfg2 %.3f  cfg3 %.3f  cfg4 %.3f' % (d['ms_per_step'], k['lse_hash_bwd'], d['m_packed']['ms_per_step'], g('default_config'), d['default_config']['kernel_ms_per_step']['lse_hash_bwd'], g('cfg2_composition'), g('cfg3_composition'), g('cfg4_composition')))")
  echo "$NAME=$v | $b" | tee -a $OUT
done
