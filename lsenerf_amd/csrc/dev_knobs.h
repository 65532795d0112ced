/*
 * dev_knobs.h -- tuning knobs of the DEVELOPMENT build (make dev -> liblse_hip_dev.so, compiled with -DLSE_DEV_KNOBS).
 *
 * Not part of the C-ABI: the library that ships (liblse_hip.so, include/lse_hip.h) has no global or thread-local state, does not
 * export these two functions and compiles the knobs below in as constants (csrc/api.cpp); the superseded kernel variants they
 * select (other tile shapes, the one-cache-pass-per-level hash backward, the cache-free coarse-level kernel, the LDS-resident hash
 * forward, earlier flush generations) exist in the development build only.  Used by the A/B tools under tools/ and by the tests
 * that hold every variant against the oracle (tests/test_gpu_parity.py, `_lib.dev_library()`).
 *
 *   "hash_fwd_mapping"  workgroup -> (level, chunk) order of lse_hash_fwd: 4 = level-major, finest level first (default),
 *                       3 = level-major coarse first, 0 / 1 = XCD-bound levels, 2 = level-interleaved
 *   "hash_fwd_lds_levels"  k coarsest levels whose whole table fits one CU's LDS run in hash_fwd_lds_kernel (default 0;
 *                       bit-identical, 2 - 5 % slower: DESIGN.md appendix)
 *   "compact_features_groups"  level groups per launch of lse_compact_features (default 4)
 *   "mlp_fwd_cfg" / "mlp_bwd_cfg"  CT * 10 + NW tile shape of the f32-MFMA fused MLP kernels (default 28; 44 = A/B partner)
 *   "mlp_bwd_impl"      1 = contiguous-tiles fused backward (bias gradient inside the dW0 MFMAs; default), 0 = generic kernel
 *   "mlp_bwd3_cfg"      CT * 100 + NW of the bf16-piece fused backward (default 208; 112, 108, 204 = A/B partners)
 *   "mlp_act_nt"        non-temporal stores of saved activations (default 0)
 *   "hash_bwd_probes" / "hash_bwd_few_runs" / "hash_bwd_stage_max"   defaults of lse_hash_bwd_opts.second_probe (3), .few_runs (6)
 *                       and .stage_max (16), for A/B runs of whole steps (a single call sets them in lse_hash_bwd_opts)
 *   "traverse_vec"      1 = 64-steps-at-once marcher for constant step sizes (default, bit-identical), 0 = serial loop only
 * Integer outputs never depend on these; floating-point results agree within rounding.  Returns LSE_E_INVALID for an unknown name.
 */
#ifndef LSE_DEV_KNOBS_H
#define LSE_DEV_KNOBS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int lse_set_option(const char *name, int64_t value);
int lse_get_option(const char *name, int64_t *value);
#ifdef __cplusplus
}
#endif
#endif
