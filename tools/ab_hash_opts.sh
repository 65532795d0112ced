#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# Defaults of the hash backward (run-time options hash_bwd_few_runs / hash_bwd_stage_max / hash_bwd_probes) on every regime bench.py
# measures.  usage: bash tools/ab_hash_opts.sh <out file> "FEW_RUNS=8" "FEW_RUNS=8 STAGE_MAX=24" ...
OUT=$1; shift; mkdir -p $(dirname $OUT)
for cfg in "$@"; do
  unset LSE_OPT_HASH_BWD_FEW_RUNS LSE_OPT_HASH_BWD_STAGE_MAX LSE_OPT_HASH_BWD_PROBES
  for kv in $cfg; do export LSE_OPT_HASH_BWD_$kv; done
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
hb = lambda e: e['kernel_ms_per_step']['lse_hash_bwd']
print('%-28s headline %.3f ms (hash_bwd %.3f)   m_packed %.3f   inside %.3f   default_config graphed %.3f (hash_bwd %.3f)   cfg2 graphed %.3f (hash_bwd %.3f)  cfg4 graphed %.3f' % (
    '$cfg', d['ms_per_step'], k['lse_hash_bwd'], d['m_packed']['ms_per_step'], d['m_march_inside_box']['ms_per_step'], d['default_config']['graphed']['ms_per_step'],
    hb(d['default_config']), d['cfg2_composition']['graphed']['ms_per_step'], hb(d['cfg2_composition']),
    d['cfg4_composition']['graphed']['ms_per_step']))" | tee -a $OUT
done
