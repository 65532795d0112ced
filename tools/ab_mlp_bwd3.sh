#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# A/B the third-generation MLP backward configurations (mlp_bwd3_cfg = CT * 100 + NW): parity tests, then per-step kernel times.
# usage: bash tools/ab_mlp_bwd3.sh <out dir> cfg...
OUT=$1; shift; mkdir -p $OUT
for cfg in "$@"; do
  export LSE_OPT_MLP_BWD3_CFG=$cfg
  r=$(timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "mlp or bf16 or config" 2>&1 | tail -1)
  b=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-context --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('ms/step %.3f mlp_fwd %.3f mlp_bwd %.3f' % (d['ms_per_step'], k['lse_mlp_fwd'], k['lse_mlp_bwd']))")
  echo "cfg=$cfg | $r | $b" | tee -a $OUT/ab_mlp_bwd3.txt
done
