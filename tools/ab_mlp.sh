#!/bin/bash
export LSE_DEV=1      # LSE_OPT_* knobs exist in the development build only (liblse_hip_dev.so, csrc/dev_knobs.h)
# A/B the MLP kernel configurations (CT*10+NW): correctness (pytest -k mlp) then per-step kernel times from bench.py
for cfg in 44 28; do
  export LSE_OPT_MLP_FWD_CFG=$cfg LSE_OPT_MLP_BWD_CFG=$cfg
  r=$(timeout -k 10 200 python -m pytest tests -m gpu -q -k "mlp or config1" 2>&1 | tail -1)
  b=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('ms/step %.2f fwd %.3f bwd %.3f' % (d['ms_per_step'], k['lse_mlp_fwd'], k['lse_mlp_bwd']))")
  echo "cfg=$cfg | $r | $b"
done
