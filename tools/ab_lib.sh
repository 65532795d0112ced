#!/bin/bash
# A/B two builds of the library (same ABI) on the headline step: usage: bash tools/ab_lib.sh <out file> <lib A> <lib B> ...
OUT=$1; shift; mkdir -p $(dirname $OUT)
for rep in 1 2; do
for lib in "$@"; do
  export LSE_HIP_LIB=$lib
  b=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-context --no-atomic-floor --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('ms/step %.3f mlp_fwd %.3f mlp_bwd %.3f hash_bwd %.3f hash_fwd %.3f' % (d['ms_per_step'], k['lse_mlp_fwd'], k['lse_mlp_bwd'], k['lse_hash_bwd'], k['lse_hash_fwd']))")
  echo "$lib | $b" | tee -a $OUT
done
done
