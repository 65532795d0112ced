#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# A/B one run-time option of the library on the headline step: usage: bash tools/ab_opt.sh <out file> <OPTION_NAME> v1 v2 ...
OUT=$1; NAME=$2; shift 2; mkdir -p $(dirname $OUT)
for v in "$@"; do
  export LSE_OPT_$NAME=$v
  b=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-context --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('ms/step %.3f mlp_fwd %.3f mlp_bwd %.3f hash_bwd %.3f' % (d['ms_per_step'], k['lse_mlp_fwd'], k['lse_mlp_bwd'], k['lse_hash_bwd']))")
  echo "$NAME=$v | $b" | tee -a $OUT
done
