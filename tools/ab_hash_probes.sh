#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# Probe rounds of the hash backward's sector cache (option hash_bwd_probes) on every regime bench.py measures.  usage: ... <out file>
OUT=$1; shift; mkdir -p $(dirname $OUT)
for v in "$@"; do
  export LSE_OPT_HASH_BWD_PROBES=$v
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
hb = lambda e: e['kernel_ms_per_step']['lse_hash_bwd']
print('probes=$v  headline %.3f ms (hash_bwd %.3f)   m_packed %.3f   default_config %.3f graphed %.3f (hash_bwd %.3f)   cfg2 %.3f graphed %.3f (hash_bwd %.3f)  cfg4 graphed %.3f' % (
    d['ms_per_step'], k['lse_hash_bwd'], d['m_packed']['ms_per_step'], d['default_config']['ms_per_step'], d['default_config']['graphed']['ms_per_step'],
    hb(d['default_config']), d['cfg2_composition']['ms_per_step'], d['cfg2_composition']['graphed']['ms_per_step'], hb(d['cfg2_composition']),
    d['cfg4_composition']['graphed']['ms_per_step']))" | tee -a $OUT
done
