#!/bin/bash
# Round-5 diagnostics on the development build: what bounds the two dominant kernels today (profiles/r05_*).
#   bash tools/r05_diagnostics.sh gpurun_out/r5c
OUT=${1:-gpurun_out/r5_diag}; mkdir -p $OUT
export LSE_DEV=1      # liblse_hip_dev.so: the variants below are development variants (csrc/dev_knobs.h)
timeout -k 10 500 python tools/hash_bwd_variants.py "gran=6" "gran=6 nodx=1" "gran=6 nodx=1 dbg=1" "gran=6 coarse_levels=16 nodx=1" \
    "gran=6 coarse_levels=16 nodx=1 dbg=1" "gran=5" "gran=6 few_runs=4" "gran=6 few_runs=8" "gran=6 second_probe=1" "gran=6 stage_max=32" \
    > $OUT/hash_bwd_variants.txt 2> $OUT/hash_bwd_variants.err
echo "hash variants rc=$?"; tail -45 $OUT/hash_bwd_variants.txt
for cfg in 208 204; do
  LSE_OPT_MLP_BWD3_CFG=$cfg timeout -k 10 300 python bench.py --no-cpu-baseline --no-context --no-atomic-floor --steps 30 --warmup 10 2> $OUT/mlp_cfg$cfg.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('mlp_bwd3_cfg=$cfg ms/step %.3f mlp_bwd %.4f mlp_fwd %.4f hash_bwd %.4f' % (d['ms_per_step'], k['lse_mlp_bwd'], k['lse_mlp_fwd'], k['lse_hash_bwd']))" | tee -a $OUT/mlp_bwd3_cfg.txt
done
