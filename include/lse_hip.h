/*
 * lse_hip.h -- C-ABI of the MI355X (gfx950) LSENeRF hot path.
 *
 * This is the drop-in boundary named by BASELINE.json `north_star`: the entry points below are what a
 * Python binding replaces `tinycudann._C` and `nerfacc.cuda._C` with for the path
 *     LSENeRFModel.exec_get_outputs            R:lse_nerf/lsenerf.py:278-326
 *       -> LSEOccGridEstimator.sampling        R:lse_nerf/lse_grid_estimator.py:15-143
 *       -> LSEField.get_density / get_outputs  R:lse_nerf/lse_field.py:264-360
 *       -> nerfacc volrend + renderers         R:lse_nerf/lsenerf.py:300-318, R:lse_nerf/lse_renderer.py:4-10
 * (`R:` = /root/reference/).  INTEGRATION.md shows the ctypes stubs a maintainer of the reference adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with `h_` or it is a `*_desc` struct (host);
 *   - tensors are dense row-major fp32 unless stated; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it and
 *     nothing synchronises: callers own ordering, memory and lifetime (no allocation inside the library);
 *   - return value: 0 = OK, <0 = error (LSE_E_*), message via lse_last_error();
 *   - functions marked "accumulate" add into their output (callers zero it), others overwrite;
 *   - NO GLOBAL OR THREAD-LOCAL STATE (ABI 5): every entry point is a function of its arguments alone -- what a call does never
 *     depends on an earlier call.  Kernel variants, arithmetic conventions and device-side sample counts are call arguments
 *     (lse_hash_bwd_opts, lse_mlp_desc.arith, `flags`, `n_dev`); there is nothing to set, reset or bracket, and any number of
 *     host threads may call concurrently on their own streams.  The one thread-local datum is the MESSAGE of the calling thread's
 *     last failed call (lse_last_error): written by a failing call, never read by the library.  The tuning knobs of the
 *     development build (csrc/dev_knobs.h, `make dev` -> liblse_hip_dev.so) are not part of this library.
 */
#ifndef LSE_HIP_H
#define LSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSE_ABI_VERSION 6

#define LSE_OK 0
#define LSE_E_INVALID (-1)   /* bad argument (null pointer, unsupported size) */
#define LSE_E_LAUNCH (-2)    /* HIP launch / runtime error */
#define LSE_E_UNSUPPORTED (-3)

#define LSE_MAX_GRID_LEVELS 32
#define LSE_MAX_OCC_LEVELS 8

typedef void *lse_stream_t;

/* ---- descriptors ------------------------------------------------------------------------------------ */

/* tiny-cuda-nn HashGrid level table (replaces tcnn.Encoding(...) built at R:lse_nerf/lse_field.py:72-86).
 * offsets are in ENTRIES of n_features floats; offsets[n_levels] = total entries. */
typedef struct lse_grid_desc {
    int32_t n_levels;
    int32_t n_features;                              /* 2 */
    uint32_t offsets[LSE_MAX_GRID_LEVELS + 1];
    float scales[LSE_MAX_GRID_LEVELS];               /* grid_scale(level)            */
    uint32_t resolutions[LSE_MAX_GRID_LEVELS];       /* grid_resolution(scale)       */
} lse_grid_desc;

#define LSE_IN_ROWMAJOR 0   /* in[N, n_in]                                                           */
#define LSE_IN_LEVELMAJOR 1 /* in[n_in/2][N][2]  (the layout lse_hash_fwd writes)                      */
#define LSE_ACT_NONE 0
#define LSE_ACT_SIGMOID 1

/* Bias-free fused MLP in tcnn parameter layout (replaces tcnn.Network via nerfstudio MLP,
 * R:lse_nerf/lse_field.py:199-207 and :254-262): params = [W_0 (width x n_in) | W_h (width x width) x
 * (n_hidden_layers-1) | W_out (16 x width)], each row-major [out,in].  Output always padded to 16. */
typedef struct lse_mlp_desc {
    int32_t n_in;            /* 8, 16, 32 or 64 (padded input width)                */
    int32_t width;           /* 32 or 64                                            */
    int32_t n_hidden_layers; /* 1 or 2                                              */
    int32_t out_activation;  /* LSE_ACT_*                                           */
    int32_t in_layout;       /* LSE_IN_*                                            */
    /* optional first-layer VIEW into params / d_params (all zero = the plain layout above):
     * W_0[row][c] = params[row * w0_ld + w0_col + c] for c < n_in, the other layers follow at params + width * w0_ld.
     * w0_mask_col0: input column 0 has no weight (reads 0, gets no gradient).  The head MLP uses (w0_ld 64, w0_col 15,
     * mask 1): its per-sample input h[N,16] = [density logit | 15 geometry features] meets columns 16..30 of tcnn's
     * [width x 64] input matrix in place (R:lse_nerf/lse_field.py:254-262, :347-356) -- no per-step split copy. */
    int32_t w0_ld, w0_col, w0_mask_col0;
    /* arithmetic route of the FORWARD where two are built (width 64 with 16 row-major or 32 inputs): LSE_MLP_ARITH_AUTO = the
     * bf16-piece kernels (f32 operands cut into three bf16 pieces, six piece products per multiply on the bf16 matrix cores, f32
     * accumulate: the f32 error bound, csrc/mlp_x6.h); LSE_MLP_ARITH_F32_MFMA = v_mfma_f32_16x16x4_f32 (what every other shape
     * runs on).  The two routes differ in the last bits, both within f32 rounding of the exact result.  The backward's route
     * follows act_tiled (3 = bf16 pieces). */
    int32_t arith;
} lse_mlp_desc;
#define LSE_MLP_ARITH_AUTO 0
#define LSE_MLP_ARITH_F32_MFMA 1

/* ---- misc -------------------------------------------------------------------------------------------- */
int lse_abi_version(void);
const char *lse_last_error(void);

/* ---- sampler: replaces nerfacc.grid.traverse_grids + the mask extraction at
 *      R:lse_nerf/lse_grid_estimator.py:93-106.  Two passes like the upstream kernel:
 *      mode 0 fills chunk_cnts[R]; mode 1 reads chunk_starts[R] and writes the packed samples.
 *      binaries [levels, rx, ry, rz] uint8; aabbs [levels, 6]; near/far per ray.  Integer outputs are
 *      bit-exact against oracle/c/lse_oracle.c.
 *      flags: 0 = every product and sum of the traversal set-up rounded separately (bit-exact against
 *      oracle/c/liblse_oracle.so); LSE_TRAVERSE_FMA_SETUP = the a*b+c sites of nerfacc's grid.cu (ray start / end, the two
 *      products of tmax_xyz) as fused multiply-adds -- what nvcc's default contraction emits -- bit-exact against
 *      liblse_oracle_fma.so.  The two conventions differ in about one sample interval per million. ------------ */
#define LSE_TRAVERSE_FMA_SETUP 1
int lse_traverse_grids(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries,
                       const float *aabbs, int32_t levels, int32_t rx, int32_t ry, int32_t rz,
                       const float *near_planes, const float *far_planes, float step_size, float cone_angle,
                       int32_t mode, int64_t *chunk_cnts, const int64_t *chunk_starts, int32_t *ray_indices,
                       float *t_starts, float *t_ends, int32_t flags, lse_stream_t stream);

/* Single-pass variant of the same traversal (same arithmetic, same samples): ray r writes its (t_start, t_end) pairs into
 * the fixed-capacity slots [r*cap, (r+1)*cap) and its count into chunk_cnts[r]; *overflow is OR-ed with 1 if a ray
 * produced more than cap samples (the caller then falls back to the two-pass calls above).  The host bound
 * cap >= (t_exit - t_enter) / step_size + slack holds because every sample interval is at least step_size long.
 * lse_compact_ray_slots packs the slots: ray_indices / t_starts / t_ends [N] at packed_info[r] = (start, count). */
int lse_traverse_grids_slots(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries,
                             const float *aabbs, int32_t levels, int32_t rx, int32_t ry, int32_t rz,
                             const float *near_planes, const float *far_planes, float step_size, float cone_angle,
                             int64_t cap, int64_t *chunk_cnts, float *t_start_slots, float *t_end_slots,
                             int32_t *overflow, int32_t flags, lse_stream_t stream);
int lse_compact_ray_slots(const float *t_start_slots, const float *t_end_slots, int64_t cap, const int64_t *packed_info,
                          int32_t n_rays, int32_t *ray_indices, float *t_starts, float *t_ends, lse_stream_t stream);

/* nerfacc.pack_info (R:lse_nerf/lsenerf.py:300): packed_info[R,2] = (exclusive cumsum, count); total[1]. */
int lse_pack_info_from_counts(const int64_t *chunk_cnts, int32_t n_rays, int64_t *packed_info, int64_t *total,
                              lse_stream_t stream);
/* nerfstudio VolumetricSampler.forward: "if num_samples == 0: create a single fake sample" (ray 0, starts = ends = 1,
 * packed_info[0] = (0, 1)) -- applied on the device to a device-side count (*n_dev == 0 -> 1).  Arrays need room for 1 sample.
 * feat_x01 [C,3] / feat_sel [C] / feat_y [n_levels][y_level_stride floats] (each nullable; ABI 4): the visibility pre-pass's
 * compacted survivor features that the main pass re-uses (lse_compact_features).  With no survivor they hold nothing for slot 0,
 * where the fake sample lands: the slot is zeroed (the sample has zero extent -- weight and gradients exactly 0 -- so any FINITE
 * features give the reference's outputs; uninitialised ones may hold NaN). */
int lse_fake_sample_if_empty(int64_t *packed_info, int32_t n_rays, int64_t *n_dev, int32_t *ray_indices, float *t_starts,
                             float *t_ends, float *feat_x01, uint8_t *feat_sel, float *feat_y, int64_t y_level_stride,
                             int32_t n_levels, int32_t n_features, lse_stream_t stream);

/* Per-ray near / far planes of R:lse_nerf/lse_grid_estimator.py:83-92 in one launch (bit-identical to the torch ops):
 * near = max(near_plane, t_min[r]) (+ jitter[r] * step_size when jitter is given: stratified sampling),
 * far = min(far_plane, t_max[r]).  t_min / t_max / jitter are nullable. */
int lse_ray_planes(float near_plane, float far_plane, const float *t_min, const float *t_max, const float *jitter,
                   float step_size, int32_t n_rays, float *near_planes, float *far_planes, lse_stream_t stream);

/* nerfacc.render_visibility_from_density (R:lse_nerf/lse_grid_estimator.py:120-127): mask[N] uint8 and the
 * per-ray surviving counts new_cnts[R]. */
int lse_visibility_mask(const float *t_starts, const float *t_ends, const float *sigmas,
                        const int64_t *packed_info, int32_t n_rays, float early_stop_eps, float alpha_thre,
                        uint8_t *mask, int64_t *new_cnts, lse_stream_t stream);
/* The same with R:lse_nerf/lse_grid_estimator.py:112-116's `alpha_thre = min(alpha_thre, self.occs.mean().item())` evaluated on
 * the device: alpha_cap points to occs.mean() in device memory (no host read-back; a captured launch follows grid refreshes). */
int lse_visibility_mask_cap(const float *t_starts, const float *t_ends, const float *sigmas,
                            const int64_t *packed_info, int32_t n_rays, float early_stop_eps, float alpha_thre,
                            const float *alpha_cap, uint8_t *mask, int64_t *new_cnts, lse_stream_t stream);
/* Survivors of the visibility culling keep what the sigma_fn pre-pass computed for them: x01[N,3], selector[N] and the
 * level-major hash features y[L][N][2] (F = 2) are compacted with the same mask / packed_info pair as
 * lse_compact_samples, so the main field pass does not encode them again (R:lse_nerf/lse_grid_estimator.py:109-143 evaluates
 * the field twice on every survivor). */
int lse_compact_features(const uint8_t *mask, const int64_t *packed_info, const int64_t *new_packed_info, int32_t n_rays,
                         const float *x01, const uint8_t *selector, const float *y, int32_t n_levels, int64_t n_old,
                         int64_t n_new, float *out_x01, uint8_t *out_selector, float *out_y, lse_stream_t stream);

/* nerfacc.render_visibility_from_alpha (the alpha_fn branch, R:lse_nerf/lse_grid_estimator.py:128-138): same scan on
 * per-sample opacities; T_k = prod_{i<k} (1 - alpha_i), mask = T >= early_stop_eps && alpha >= alpha_thre. */
int lse_visibility_mask_alpha(const float *alphas, const int64_t *packed_info, int32_t n_rays, float early_stop_eps,
                              float alpha_thre, uint8_t *mask, int64_t *new_cnts, lse_stream_t stream);

/* mask compaction at R:lse_nerf/lse_grid_estimator.py:139-143 (order preserving, per ray). */
int lse_compact_samples(const uint8_t *mask, const int64_t *packed_info, const int64_t *new_packed_info,
                        int32_t n_rays, const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                        int32_t *out_ray_indices, float *out_t_starts, float *out_t_ends, lse_stream_t stream);

/* ---- field ------------------------------------------------------------------------------------------- */

/* Device-side sample count `n_dev` (nullable) of the PER-SAMPLE entry points
 *     lse_positions_fwd / _bwd, lse_hash_fwd, lse_hash_bwd / _levels / _ex, lse_mlp_fwd / _bwd.
 * The sampler knows the number of packed samples only on the device (lse_pack_info_from_counts); reading it back costs a host
 * synchronisation per sampler call, twice per step with the visibility pre-pass.  With n_dev != NULL the `n` argument is a
 * CAPACITY -- buffer extents, level strides and the launch grid are sized by it -- and every kernel clamps it to the int64 that
 * n_dev addresses when it starts; workgroups past the count leave at once.  Nothing at or beyond the count is read or written.
 * The per-ray entry points (visibility, compaction, volume rendering, ray reductions) take their counts from packed_info.
 * (Until ABI 4 this pointer was ambient per-thread state, lse_set_device_count.) */

/* R:lse_nerf/lse_field.py:266-274: pos = o[ri] + d[ri]*(ts+te)/2 -> L-inf contraction -> (x+2)/4 (contraction=1)
 * or aabb normalisation (contraction=0, h_aabb[6]) -> selector -> x*selector.  ray_idx==NULL: rays_o holds N
 * positions directly (Field.density_fn). */
int lse_positions_fwd(const float *rays_o, const float *rays_d, const int32_t *ray_idx, const float *t_starts,
                      const float *t_ends, int64_t n, const int64_t *n_dev, int32_t contraction, const float *h_aabb,
                      float *x01, uint8_t *selector, lse_stream_t stream);
/* d(pos)[N,3] from d(x01)[N,3] (Jacobian of contraction/normalisation, selector-masked). */
int lse_positions_bwd(const float *rays_o, const float *rays_d, const int32_t *ray_idx, const float *t_starts,
                      const float *t_ends, int64_t n, const int64_t *n_dev, int32_t contraction, const float *h_aabb,
                      const float *d_x01, float *d_pos, lse_stream_t stream);
/* per-ray sums: d_o[r] = sum d_pos, d_d[r] = sum d_pos*(ts+te)/2 over the ray's packed samples. */
int lse_ray_grad_reduce(const float *d_pos, const float *t_starts, const float *t_ends, const int64_t *packed_info,
                        int32_t n_rays, float *d_rays_o, float *d_rays_d, lse_stream_t stream);

/* tcnn kernel_grid forward (R:lse_nerf/lse_field.py:279 via HashEncoding.forward): x01[N,3] -> y[L][N][F]. */
int lse_hash_fwd(const lse_grid_desc *desc, const float *x01, const float *table, float *y, int64_t n,
                 const int64_t *n_dev, lse_stream_t stream);
/* tcnn kernel_grid_backward (+_input): dtable accumulate (float atomics); dx[N,3] overwritten (NULL: skip). */
int lse_hash_bwd(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table, float *dtable,
                 float *dx, int64_t n, const int64_t *n_dev, lse_stream_t stream);

/* The same backward restricted to levels [level_begin, level_end); dx_accumulate != 0 adds this launch's share to dx.
 * Lets the caller finish the fine levels' table gradients (most of the bytes) first and start their all-reduce while
 * the remaining levels are still being computed (lsenerf_amd.dist.OverlappedGradExchange). */
int lse_hash_bwd_levels(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                        float *dtable, float *dx, int32_t dx_accumulate, int32_t level_begin, int32_t level_end,
                        int64_t n, const int64_t *n_dev, lse_stream_t stream);

/* Kernel selection of the hash backward as CALL ARGUMENTS (so that two variants can be compared inside one process):
 *   impl          2 = lane-per-sample kernel, per-wave LDS sector cache keyed by GLOBAL sector id, the run ends of several
 *                     levels batched into one cache pass (default); 1 = one cache pass per level; 0 = 16-lanes-per-sample kernel
 *   stage_max     impl 2: with an empty queue, a level that ends more than stage_max runs in the wave passes unstaged (default 48;
 *                     56 is better when the step size is constant: profiles/r05_hash_bwd_thresholds_final.txt)
 *   gran          cache slots: 2 = 512 slots of one 32-B sector; 3 (impl 1 only) = 256 slots of one 64-B line; 4 (impl 2) =
 *                     512 sector slots PAIRED by 64-B line, flushed in slot order -- a float-atomic request costs the same for 4 .. 64
 *                     contiguous bytes (tools/micro/atomic_gran.hip), so sibling sectors leave as one request; 5 = the same with
 *                     256 slots (three workgroups per CU: faster cache passes, more collision requests -- equal at the metric size);
 *                     6 (impl 2, default) = 4 with the second-generation flush (list stored trip-major transposed, payload read and
 *                     zeroed by one LDS exchange, keys reset in bulk: 9 instead of 20 LDS instructions per 32 flushed slots);
 *                     -- in workgroups of ONE wave since round 5 (a workgroup's LDS is released when its last wave retires);
 *                     development build only: 8 = 6 with 320 slots in workgroups of four waves (12 waves per CU), 9 / 11 = 384
 *                     slots in workgroups of two / one waves (10 / 9 waves per CU), 10 = 448 slots, one wave (9 waves per CU),
 *                     12 / 13 = 512 slots in workgroups of four (the shape until round 5) / two waves
 *   few_runs      impl 1, 2: a wave that ends <= few_runs runs at a level adds them straight to memory (default 8; 3 is better when
 *                     the step size is constant)
 *   second_probe  impl 1, 2: extra probe rounds (home slot + k * step, k = 1 .. second_probe) before a corner falls back to memory (default 3)
 *   rounds        impl 0: 16 / 32 / 64 rounds of 4 samples per wave (default 32)
 *   interleave_from_scale  impl 0: levels with scale >= this use the interleaved sample mapping (default: never)
 *   coarse_levels the levels below this one are processed first by a cache-free kernel at high occupancy (lane = sample, run
 *                     ends transposed through a 1.5 KB staging area, 16 lanes per run add straight to memory); every value is
 *                     correct, the split only moves time between the two kernels
 *   replicas, replica_levels, workspace, workspace_bytes
 *                 the coarsest levels are a few thousand 64-byte lines that EVERY wave of a launch adds to, and the memory-side
 *                 atomic units serialise requests to one line: the direct adds (few_runs path, coarse kernel) of the levels below
 *                 replica_levels (default 4: 1 MB of table) go to one of `replicas` (default 16, a power of two) zero-initialised copies
 *                 in `workspace`, chosen by the index of the wave's group of four, and a small kernel folds them into dtable at the end of the call
 *                 (leaving the workspace zero).  workspace == NULL (lse_hash_bwd, lse_hash_bwd_levels): no replicas, same values
 *   dbg           timing experiments only (bit 0: skip flush atomics, bit 1: skip run ends, bit 2: skip the scan) -> WRONG results
 * lse_hash_bwd / lse_hash_bwd_levels use lse_hash_bwd_default_opts(). */
typedef struct lse_hash_bwd_opts {
    int32_t impl, gran, few_runs, second_probe, rounds, dbg;
    float interleave_from_scale;
    int32_t stage_max;
    int32_t coarse_levels;
    int32_t replicas, replica_levels;
    void *workspace;            /* device memory, zero on entry, zero again when the call's kernels have run; NULL = no replicas */
    int64_t workspace_bytes;
    int32_t prefetch;           /* gran 6: fetch level l+1's dy / table operands before level l's cache pass (see hashgrid.hip) */
} lse_hash_bwd_opts;
void lse_hash_bwd_default_opts(lse_hash_bwd_opts *opts);
/* bytes of `workspace` that lse_hash_bwd_ex needs for opts->replicas x the levels below opts->replica_levels (NULL = defaults).
 * The replicated levels stop at the first level that would push ONE replica past 2 MB (grids whose coarse levels are already
 * large gain nothing from replicas): 0 is a valid answer and means "no workspace, no replicas". */
int64_t lse_hash_bwd_workspace_bytes(const lse_grid_desc *desc, const lse_hash_bwd_opts *opts);
int lse_hash_bwd_ex(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table, float *dtable,
                    float *dx, int32_t dx_accumulate, int32_t level_begin, int32_t level_end, int64_t n,
                    const int64_t *n_dev, const lse_hash_bwd_opts *opts /* NULL = defaults */, lse_stream_t stream);

/* fused MLP forward on the matrix cores (f32 MFMA, or bf16 MFMA on three-piece operands with the same error bound:
 * lse_mlp_desc.arith).  row_bias[R,width] (nullable) is added to layer-0 pre-activations of sample i
 * from row row_bias_idx[i].  act (nullable) receives the post-ReLU hidden activations: act_tiled = 0 -> row-major
 * [n_hidden_layers][N][width] (what lse_mlp_wgrad reads); act_tiled = 1 -> tile-major, an opaque workspace of
 * n_hidden_layers * roundup(N,16) * width floats that only lse_mlp_bwd (same act_tiled) understands; act_tiled = 2 (two
 * hidden layers, row-major input): tile-major WITHOUT the first hidden layer ((n_hidden_layers - 1) * roundup(N,16) * width
 * floats) -- lse_mlp_bwd recomputes it from `in` and `row_bias` (32 extra MFMAs per 32 samples against 2 KiB per sample of
 * activation traffic), bit-identical to what the forward computed;  act_tiled = 3 (width 64 with 16 row-major inputs and two
 * hidden layers, or 32 inputs and one): NOTHING is saved (act may be NULL) -- lse_mlp_bwd recomputes every hidden layer on the
 * bf16 matrix cores (three-piece operands, csrc/mlp_x6.h), fused weight gradients only (d_params required; d_out_pre / d_act /
 * d_act0 not available).
 * out is [N,16] (out_cols = 16) or the compact [N,4] holding outputs 0..3 (out_cols = 4, e.g. rgb).
 * sigma_out (nullable): fused density head sigma[N] = density_scale * exp(out[:,0]) * selector (trunc_exp,
 * R:lse_nerf/lse_field.py:286-287); selector nullable. */
int lse_mlp_fwd(const lse_mlp_desc *desc, const float *params, const float *in, const float *row_bias,
                const int32_t *row_bias_idx, float *out, int32_t out_cols, float *act, int32_t act_tiled,
                float *sigma_out, const uint8_t *selector, float density_scale, int64_t n, const int64_t *n_dev,
                lse_stream_t stream);
/* backward: d_out[N,out_cols] (w.r.t. the activated output) -> d_in (layout of desc->in_layout; nullable) and, when
 * d_params is given, the weight gradients of every layer accumulated into d_params (same layout as params) in the
 * same pass (`in` = the layer-0 input is then required).  d_sigma (nullable): gradient of the fused density head, folded
 * into the gradient of output 0 (trunc_exp backward: * density_scale * exp(clamp(out0,-15,15)) * selector).
 * Optional outputs: d_out_pre[N,16], d_act[n_hidden_layers][N][width] (pre-activation gradients of every layer, what
 * lse_mlp_wgrad consumes), d_act0[N][width] (layer 0 only: the row_bias gradient before the per-row sum), or directly
 * d_row_bias[rows][width] (accumulate; row_bias_idx[N] must be sorted, i.e. each row's samples contiguous). */
int lse_mlp_bwd(const lse_mlp_desc *desc, const float *params, const float *in, const float *act, int32_t act_tiled,
                const float *out, int32_t out_cols, const float *d_out, const float *d_sigma, const uint8_t *selector, float density_scale,
                float *d_out_pre, float *d_act, float *d_act0, float *d_in, float *d_params,
                const float *row_bias, const int32_t *row_bias_idx, float *d_row_bias, int64_t n, const int64_t *n_dev,
                lse_stream_t stream);
/* unfused weight gradients from materialised d_act / d_out_pre, accumulate into d_params (same layout as params). */
int lse_mlp_wgrad(const lse_mlp_desc *desc, const float *in, const float *act, const float *d_act,
                  const float *d_out_pre, float *d_params, int64_t n, lse_stream_t stream);
/* out[R,width] += per-ray sum of d_act0[N,width] (gradient of row_bias). */
int lse_segment_sum_rows(const float *rows, int32_t width, const int64_t *packed_info, int32_t n_rays, float *out,
                         lse_stream_t stream);

/* per-ray head features in tcnn column order [SH16(dir) | 15 zeros (geo slots) | emb(32) | 1]
 * (R:lse_nerf/lse_field.py:298-300, 306-310, 347-356; SH of tcnn, degree 4). emb_table NULL -> zeros. */
int lse_ray_features_fwd(const float *rays_d, const float *emb_table, const int32_t *emb_idx, int32_t n_rays,
                         int32_t emb_dim, float *feat, lse_stream_t stream);
/* d_rays_d[R,3] overwritten (nullable); d_emb_table[n_emb_rows, emb_dim] accumulate (nullable). */
int lse_ray_features_bwd(const float *rays_d, const float *d_feat, const int32_t *emb_idx, int32_t n_rays,
                         int32_t emb_dim, int32_t n_emb_rows, float *d_rays_d, float *d_emb_table, lse_stream_t stream);
/* small dense helpers on per-ray matrices: y[R,M] = x[R,K] W[M,K]^T ; dx[R,K] = dy[R,M] W[M,K] ;
 * dW[M,K] += dy^T x. */
/* The same features AND the per-ray share of the head's first layer in one launch:
 *   feat[R, in_pad] = [SH16 | 0 x 15 | emb (emb_dim) | ones padding],  in_pad = roundup(31 + emb_dim, 16)  (tcnn's layout);
 *   0 <= emb_dim <= 97 (the reference's LSEEmbeddingConfig.emb_dim, default 32: in_pad 64)
 *   row_bias[R, width] = feat * W_in^T with W_in[width][w_ld] the head's tcnn input matrix in place.
 * Backward: d_feat[R, in_pad] = d_row_bias * W_in (workspace, overwritten), d_rays_d[R,3] (nullable) through the SH
 * Jacobian, d_emb_table[n_emb_rows, emb_dim] (nullable, accumulate).  The weight gradient d W_in += d_row_bias^T feat is
 * lse_gemm_tn_acc(d_row_bias, width, feat, in_pad, ..., dw_ld = w_ld) straight into the parameter gradient. */
int lse_ray_bias_fwd(const float *rays_d, const float *emb_table, const int32_t *emb_idx, int32_t n_rays,
                     int32_t emb_dim, const float *w_in, int32_t w_ld, int32_t width, float *feat, float *row_bias,
                     lse_stream_t stream);
int lse_ray_bias_bwd(const float *rays_d, const int32_t *emb_idx, int32_t n_rays, int32_t emb_dim, int32_t n_emb_rows,
                     const float *w_in, int32_t w_ld, int32_t width, const float *d_row_bias, float *d_feat,
                     float *d_rays_d, float *d_emb_table, lse_stream_t stream);
int lse_linear_fwd(const float *w, const float *x, int32_t rows, int32_t m, int32_t k, float *y, lse_stream_t stream);
int lse_linear_bwd_input(const float *w, const float *dy, int32_t rows, int32_t m, int32_t k, float *dx,
                         lse_stream_t stream);
int lse_gemm_tn_acc(const float *g, int32_t m, const float *a, int32_t k, int32_t a_layout, int64_t n, float *dw,
                    int32_t dw_ld, lse_stream_t stream);

/* density = scale * exp(h[:,0]) * selector (trunc_exp, R:lse_nerf/lse_field.py:286-287); h is [N,16]. */
int lse_density_fwd(const float *h, const uint8_t *selector, float scale, float *sigma, int64_t n, lse_stream_t stream);
/* d_h[:,0] = d_sigma * scale * exp(clamp(h0,-15,15)) * selector (overwrites column 0 only). */
int lse_density_bwd(const float *h, const uint8_t *selector, float scale, const float *d_sigma, float *d_h, int64_t n,
                    lse_stream_t stream);

/* ---- renderer: nerfacc.render_weight_from_density + accumulate_along_rays
 *      (R:lse_nerf/lsenerf.py:301-318, R:lse_nerf/lse_renderer.py:6-10).  rgb has row stride rgb_stride floats. */
int lse_volrend_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                    int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, float *weights, float *out_rgb,
                    float *out_acc, float *out_depth_num, lse_stream_t stream);
int lse_volrend_bwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                    int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, const float *weights,
                    const float *d_out_rgb, const float *d_out_acc, const float *d_out_depth_num, float *d_sigmas,
                    float *d_rgb, lse_stream_t stream);

/* lse_volrend_fwd + the DepthRenderer("expected") epilogue of R:lse_nerf/lsenerf.py:315-317 in the same call:
 * out_depth[r] = clip(out_depth_num[r] / (out_acc[r] + 1e-10), lo, hi) with (lo, hi) the global min / max of the interval
 * mid-points (nerfstudio clips to steps.min() / steps.max()); samples are sorted inside a ray, so the range is reduced
 * from the per-ray first / last mid-point (mid_range [R,2], workspace) instead of two passes over all N samples.
 * depth_range[2] (nullable) receives (lo, hi) for the backward. */
int lse_volrend_depth_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                          int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, float *weights, float *out_rgb,
                          float *out_acc, float *out_depth_num, float *mid_range, float *out_depth, float *depth_range,
                          lse_stream_t stream);
/* nerfacc.render_weight_from_density alone (R:lse_nerf/lsenerf.py:301-306): weights, and optionally transmittance and
 * alphas per sample; the backward takes a per-sample d_weights (the generic route: renderers applied by the caller). */
int lse_render_weight_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const int64_t *packed_info,
                          int32_t n_rays, float *weights, float *trans, float *alphas, lse_stream_t stream);
int lse_render_weight_bwd(const float *t_starts, const float *t_ends, const float *sigmas, const int64_t *packed_info,
                          int32_t n_rays, const float *weights, const float *d_weights, float *d_sigmas,
                          lse_stream_t stream);

/* ---- occupancy grid (nerfacc OccGridEstimator._update) --------------------------- */
/* occs[id] = max(occs[id]*ema_decay, occ_new); duplicate ids resolve to the maximum over the duplicates (upstream:
 * an arbitrary one of them wins).  workspace: n floats. */
int lse_occ_update_cells(float *occs, const int64_t *cell_ids, const float *occ_new, int64_t n, float ema_decay,
                         float *workspace, lse_stream_t stream);
int lse_occ_binarize(const float *occs, int64_t n, const float *d_threshold, uint8_t *binaries, lse_stream_t stream);

/* ---- training epilogue: output routing + intensity mappers + both losses, O(rays), one launch each way
 *      (R:lse_nerf/lsenerf.py:329-377 routing, :392-439 losses; R:lse_nerf/intensity_mappers.py:64-94 mappers).
 * Colour bundle:  v = rgb_mapped ? m_rgb(max(rgb, 1e-5)) : rgb;  mean over deblur_group consecutive rays (the 4 virtual
 *   cameras of a pixel, R:lse_nerf/lsenerf.py:365-370);  max(., 1e-5);  rgb_loss = mean (v - col_gt)^2.
 * Event bundles (prev, next):  c = max(rgb, 1e-5);  ev_one_dim != NONE: m_evs(sum_k w_k c_k) with w = softmax(w31)
 *   (ThreeToOne, LEARNED) or the gray vector (GRAY);  NONE: gray(m_evs(c));  L = log(. + 1e-6);
 *   event_loss = evs_loss_weight * mean (L_next - L_prev - evs_gt)^2          (log_loss, R:lse_nerf/lsenerf.py:392-399), or with
 *   event_loss_kind = LSE_EVLOSS_ENERF_NORM (enerf_norm_loss, :406-419; ABI 6):  d_r = L_next - L_prev,  c_r = evs_gt[r] / e_thresh[r],
 *   event_loss = evs_loss_weight * mean (d_r / (||d||_2 + 1e-6) - c_r / (||c||_2 + 1e-6))^2, norms over the event rays, c constant.
 * m = identity | x^(1/2.4) ("gt") | x^p ("powpow", p = *pow_rgb / *pow_evs, learnable) | "mlp" | "rgb_mlp" (ABI 6): the
 *   identity-initialised nerfstudio MLP(in, num_layers = 4, layer_width = 16, out = in, ReLU, out_activation = Sigmoid) of
 *   R:lse_nerf/intensity_mappers.py:28-62 -- "mlp" maps ONE channel (in = 1: the event side behind ev_one_dim; on three channels
 *   the reference's nn.Linear(1, 16) raises, and so does the call), "rgb_mlp" the three channels together (in = 3: the colour
 *   side, or the event side with ev_one_dim == NONE).  Their parameters come as lse_mapper_mlp. */
#define LSE_MAP_IDENTITY 1
#define LSE_MAP_GT 2
#define LSE_MAP_POWPOW 3
#define LSE_MAP_MLP 4
#define LSE_MAP_RGB_MLP 5
/* The four nn.Linear layers of an MLP mapper, in = 1 ("mlp") | 3 ("rgb_mlp"): w[l] row-major [out][in] = [16,in] [16,16] [16,16]
 * [in,16], b[l] = [16] [16] [16] [in] (device pointers; the struct itself is host memory, read during the call).  dw / db: the
 * gradients, same shapes, ACCUMULATED into (+=) by lse_loss_epilogue_bwd in a fixed order (no atomics); all eight NULL = not
 * wanted; the forward ignores them. */
typedef struct lse_mapper_mlp {
    const float *w[4];
    const float *b[4];
    float *dw[4];
    float *db[4];
} lse_mapper_mlp;
#define LSE_ONE_DIM_NONE 0
#define LSE_ONE_DIM_LEARNED 1
#define LSE_ONE_DIM_GRAY 2
#define LSE_EVLOSS_LOG 0
#define LSE_EVLOSS_ENERF_NORM 1
typedef struct lse_epilogue_desc {
    int32_t rgb_mapped;      /* 0: the colour loss sees the raw render; 1: m_rgb(max(rgb, 1e-5)) */
    int32_t rgb_mapper;      /* LSE_MAP_* */
    int32_t evs_mapper;      /* LSE_MAP_* applied on the event side */
    int32_t ev_one_dim;      /* LSE_ONE_DIM_* */
    int32_t deblur_group;    /* 1, or 4 for rgb_loss_type == "deblur" */
    float evs_loss_weight;
    int32_t event_loss_kind; /* LSE_EVLOSS_* */
} lse_epilogue_desc;
/* losses[2] = (rgb_loss, event_loss); a bundle whose pointer is NULL contributes 0.  Deterministic (no atomics). */
/* mlp_rgb / mlp_evs: parameters of the colour-side / event-side mapper when its kind is LSE_MAP_MLP / LSE_MAP_RGB_MLP (else ignored,
 * NULL allowed).  e_thresh [n_ev]: the per-ray event threshold of LSE_EVLOSS_ENERF_NORM (NULL = 1; ignored by log_loss). */
int lse_loss_epilogue_fwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt, int32_t n_col,
                          const float *prev_rgb, const float *next_rgb, const float *evs_gt, const float *e_thresh, int32_t n_ev,
                          const float *pow_rgb, const float *pow_evs, const float *w31, const lse_mapper_mlp *mlp_rgb,
                          const lse_mapper_mlp *mlp_evs, float *losses, lse_stream_t stream);
/* g_rgb_loss / g_event_loss: device scalars, the upstream gradients of the two losses (NULL = 0).  d_col [n_col*deblur_group,3], d_prev / d_next [n_ev,3]
 * (each nullable) are overwritten; d_scalars[5] = (d pow_rgb, d pow_evs, d w31[3]) is overwritten; the gradients of an MLP mapper's
 * parameters are accumulated into mlp_*->dw / db. */
int lse_loss_epilogue_bwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt, int32_t n_col,
                          const float *prev_rgb, const float *next_rgb, const float *evs_gt, const float *e_thresh, int32_t n_ev,
                          const float *pow_rgb, const float *pow_evs, const float *w31, const lse_mapper_mlp *mlp_rgb,
                          const lse_mapper_mlp *mlp_evs, const float *g_rgb_loss, const float *g_event_loss, float *d_col,
                          float *d_prev, float *d_next, float *d_scalars, lse_stream_t stream);

/* ---- optimiser: torch.optim.Adam semantics on a flat buffer (R:lse_nerf/lse_config.py:29-33) ----------- */
int lse_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, float grad_scale, lse_stream_t stream);
/* The same with EVERY scalar of the step in device memory: hyper[6] = {lr, 1 - beta1^step, 1 / sqrt(1 - beta2^step), beta1, beta2,
 * eps}.  A launch captured into a HIP graph cannot carry new scalar arguments; lse_adam_schedule_dev (captured in front of it)
 * writes all six. */
int lse_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, const float *hyper,
                      float grad_scale, lse_stream_t stream);
/* Device-side optimizer clock: *step (optimizer steps taken so far, int64 in device memory) is advanced by one and hyper[6] of
 * the NEW step is written, with nerfstudio's ExponentialDecayScheduler lr = lr_init * (lr_final / lr_init)^(min(steps taken /
 * max_steps, 1)) (max_steps <= 0 or lr_final <= 0: constant lr_init; R:lse_nerf/lse_config.py:29-38).  The schedule's constants
 * are read from device memory too -- sched[6] = {lr_init, lr_final, max_steps, beta1, beta2, eps} as doubles (ABI 5; they were
 * launch arguments, frozen into a captured graph) -- so a replayed graph depends on nothing the host writes per step, however far
 * the host runs ahead, and follows a changed learning-rate schedule without being captured again. */
int lse_adam_schedule_dev(int64_t *step, float *hyper, const double *sched, lse_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSE_HIP_H */
