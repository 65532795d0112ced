/*
 * CPU oracle, plain C, strict IEEE fp32 (build with -ffp-contract=off; see oracle/c/Makefile).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.   PARITY UNPINNED.
 *
 * Restates nerfacc 0.5.2 (pinned at R:environement.yml:139; source NOT under /root/reference and not
 * installable here) as published in nerfacc/cuda/csrc/grid.cu, anchored on the reference call site
 * R:lse_nerf/lse_grid_estimator.py:93-106:
 *     intervals, samples = traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes, far_planes,
 *                                         step_size, cone_angle)
 *     t_starts = intervals.vals[intervals.is_left]; t_ends = intervals.vals[intervals.is_right]
 * Because each emitted sample contributes exactly one is_left value (t_last) and one is_right value
 * (t_next) -- merged when consecutive samples are "continuous" -- the masked extraction at :103-104
 * equals the per-sample pair (t_last, t_next) this file emits directly.
 *
 * Float-op order is the one written in the published source, evaluated without FMA contraction and
 * with correctly-rounded division; ties in the boundary sort are resolved stably (lower slot first).
 *
 * Second build, -DLSE_ORACLE_FMA (liblse_oracle_fma.so): nvcc contracts a*b+c into fused multiply-adds by default
 * (-fmad=true), so the shipped nerfacc binary very likely evaluates the four a*b+c sites of grid.cu -- ray start / end
 * `o + d*(t +- eps)`, `start_index*voxel - ray_start`, `(...)*inv_dir + tmin`, and the marching test `t_last + dt*0.5f` --
 * with one rounding each.  The LSE_MULADD macro marks exactly those sites; the variant quantifies how many sample
 * intervals that changes (tests/test_oracle_cpu.py::test_fma_contraction_changes_few_sample_intervals).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LSE_MAX_LEVELS 8

#ifdef LSE_ORACLE_FMA
#define LSE_MULADD(a, b, c) fmaf((a), (b), (c))
int lse_oracle_fma_build(void) { return 1; }
#else
#define LSE_MULADD(a, b, c) ((a) * (b) + (c))
int lse_oracle_fma_build(void) { return 0; }
#endif

typedef struct { float x, y, z; } f3;

static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* nerfacc device::ray_aabb_intersect (slab test with early outs), then clamp to [near, far]. */
static int ray_aabb(const float o[3], const float inv[3], const float *aabb, float near_p, float far_p,
                    float *tmin_out, float *tmax_out)
{
    float tmin, tmax, tmin_t, tmax_t;
    if (inv[0] >= 0) { tmin = (aabb[0] - o[0]) * inv[0]; tmax = (aabb[3] - o[0]) * inv[0]; }
    else             { tmin = (aabb[3] - o[0]) * inv[0]; tmax = (aabb[0] - o[0]) * inv[0]; }
    if (inv[1] >= 0) { tmin_t = (aabb[1] - o[1]) * inv[1]; tmax_t = (aabb[4] - o[1]) * inv[1]; }
    else             { tmin_t = (aabb[4] - o[1]) * inv[1]; tmax_t = (aabb[1] - o[1]) * inv[1]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (inv[2] >= 0) { tmin_t = (aabb[2] - o[2]) * inv[2]; tmax_t = (aabb[5] - o[2]) * inv[2]; }
    else             { tmin_t = (aabb[5] - o[2]) * inv[2]; tmax_t = (aabb[2] - o[2]) * inv[2]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return 0;
    *tmin_out = fmaxf(tmin, near_p);
    *tmax_out = fminf(tmax, far_p);
    return 1;
}

/* public: [n_rays, n_aabbs] t_mins, t_maxs, hits -- nerfacc.grid.ray_aabb_intersect */
void lse_oracle_ray_aabb_intersect(const float *rays_o, const float *rays_d, int n_rays, const float *aabbs,
                                   int n_aabbs, float near_p, float far_p, float miss_value, float *t_mins,
                                   float *t_maxs, uint8_t *hits)
{
    for (int r = 0; r < n_rays; ++r) {
        float inv[3] = {1.0f / rays_d[r * 3 + 0], 1.0f / rays_d[r * 3 + 1], 1.0f / rays_d[r * 3 + 2]};
        for (int a = 0; a < n_aabbs; ++a) {
            float t0, t1;
            int h = ray_aabb(rays_o + r * 3, inv, aabbs + a * 6, near_p, far_p, &t0, &t1);
            t_mins[r * n_aabbs + a] = h ? t0 : miss_value;
            t_maxs[r * n_aabbs + a] = h ? t1 : miss_value;
            hits[r * n_aabbs + a] = (uint8_t)h;
        }
    }
}

static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    return clampf(t * cone_angle, dt_min, dt_max);
}

/*
 * mode 0: count pass (fills chunk_cnts).  mode 1: write pass (reads chunk_starts, fills outputs).
 * binaries [n_grids, rx, ry, rz] uint8; aabbs [n_grids, 6].  near/far are per ray.
 */
void lse_oracle_traverse_grids(const float *rays_o, const float *rays_d, int n_rays, const uint8_t *binaries,
                               const float *aabbs, int n_grids, int rx, int ry, int rz, const float *near_planes,
                               const float *far_planes, float step_size, float cone_angle, int mode,
                               int64_t *chunk_cnts, const int64_t *chunk_starts, int64_t *ray_indices,
                               float *t_starts, float *t_ends)
{
    const float eps = 1e-6f;
    if (n_grids > LSE_MAX_LEVELS) return;
    for (int tid = 0; tid < n_rays; ++tid) {
        const float *o = rays_o + tid * 3, *d = rays_d + tid * 3;
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        const float ray_tmin = near_planes[tid], ray_tmax = far_planes[tid];

        /* python side: ray_aabb_intersect(rays_o, rays_d, aabbs) with near=-inf, far=+inf, miss=+inf */
        float ts[2 * LSE_MAX_LEVELS];
        int64_t tidx[2 * LSE_MAX_LEVELS];
        int hits[LSE_MAX_LEVELS];
        for (int l = 0; l < n_grids; ++l) {
            float t0, t1;
            hits[l] = ray_aabb(o, inv, aabbs + l * 6, -INFINITY, INFINITY, &t0, &t1);
            ts[l] = hits[l] ? t0 : INFINITY;
            ts[n_grids + l] = hits[l] ? t1 : INFINITY;
        }
        for (int i = 0; i < 2 * n_grids; ++i) tidx[i] = i;
        if (n_grids > 1) { /* torch.sort(cat([t_mins, t_maxs], -1), -1): stable insertion sort */
            for (int i = 1; i < 2 * n_grids; ++i) {
                float v = ts[i]; int64_t k = tidx[i]; int j = i - 1;
                while (j >= 0 && ts[j] > v) { ts[j + 1] = ts[j]; tidx[j + 1] = tidx[j]; --j; }
                ts[j + 1] = v; tidx[j + 1] = k;
            }
        }

        int64_t n_samples = 0;
        const int64_t base = (mode == 1) ? chunk_starts[tid] : 0;
        float t_last = ray_tmin;
        int continuous = 0;

        for (int i = 0; i < 2 * n_grids - 1; ++i) {
            int is_entering = tidx[i] < n_grids;
            int level = (int)(tidx[i] % n_grids);
            if (!hits[level]) continue;
            if (!is_entering) {
                int next_is_entering = tidx[i + 1] < n_grids;
                if (next_is_entering) continue;
                level = (int)(tidx[i + 1] % n_grids);
                if (!hits[level]) continue;
            }
            float this_tmin = fmaxf(ts[i], ray_tmin);
            float this_tmax = fminf(ts[i + 1], ray_tmax);
            if (this_tmin >= this_tmax) continue;

            if (!continuous) {
                if (step_size <= 0.0f) t_last = this_tmin;
                else for (;;) {
                    float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                    if (LSE_MULADD(dt, 0.5f, t_last) >= this_tmin) break;
                    t_last += dt;
                }
            }

            /* setup_traversal */
            const float *ab = aabbs + level * 6;
            const float res[3] = {(float)rx, (float)ry, (float)rz};
            const int resi[3] = {rx, ry, rz};
            float voxel[3], rs[3], re[3], tdist[3], delta[3];
            int cur[3], fin[3], stp[3], ovf[3];
            for (int a = 0; a < 3; ++a) {
                voxel[a] = (ab[3 + a] - ab[a]) / res[a];
                rs[a] = LSE_MULADD(d[a], this_tmin + eps, o[a]);
                re[a] = LSE_MULADD(d[a], this_tmax - eps, o[a]);
                cur[a] = clampi((int)(((rs[a] - ab[a]) / (ab[3 + a] - ab[a])) * res[a]), 0, resi[a] - 1);
                fin[a] = clampi((int)(((re[a] - ab[a]) / (ab[3 + a] - ab[a])) * res[a]), 0, resi[a] - 1);
                int idelta = d[a] > 0 ? 1 : 0;
                int start_index = cur[a] + idelta;
                float tmax_a = LSE_MULADD(ab[a] + LSE_MULADD((float)start_index, voxel[a], -rs[a]), inv[a], this_tmin);
                tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
                float stepf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
                stp[a] = (int)stepf;
                float delta_t = voxel[a] * inv[a] * stepf;
                delta[a] = (d[a] == 0.0f) ? this_tmax : delta_t;
                ovf[a] = fin[a] + stp[a];
            }

            for (;;) {
                float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                t_traverse = fminf(t_traverse, this_tmax);
                int64_t cell = (int64_t)cur[0] * ry * rz + (int64_t)cur[1] * rz + cur[2] +
                               (int64_t)level * rx * ry * rz;
                if (!binaries[cell]) {
                    if (step_size <= 0.0f) t_last = t_traverse;
                    else for (;;) {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (LSE_MULADD(dt, 0.5f, t_last) >= t_traverse) break;
                        t_last += dt;
                    }
                    continuous = 0;
                } else {
                    for (;;) {
                        float t_next;
                        if (step_size <= 0.0f) t_next = t_traverse;
                        else {
                            float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                            if (LSE_MULADD(dt, 0.5f, t_last) >= t_traverse) break;
                            t_next = t_last + dt;
                        }
                        if (mode == 1) {
                            ray_indices[base + n_samples] = tid;
                            t_starts[base + n_samples] = t_last;
                            t_ends[base + n_samples] = t_next;
                        }
                        n_samples++;
                        continuous = 1;
                        t_last = t_next;
                        if (t_next >= t_traverse) break;
                    }
                }
                /* single_traversal */
                int alive = 1;
                if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) {
                    cur[0] += stp[0]; tdist[0] += delta[0]; if (cur[0] == ovf[0]) alive = 0;
                } else if (tdist[1] < tdist[2]) {
                    cur[1] += stp[1]; tdist[1] += delta[1]; if (cur[1] == ovf[1]) alive = 0;
                } else {
                    cur[2] += stp[2]; tdist[2] += delta[2]; if (cur[2] == ovf[2]) alive = 0;
                }
                if (!alive) break;
                /* Guard (deviation, documented in DESIGN.md): where rounding puts the start cell past the
                 * final cell on the stepped axis the published loop walks out of the grid and reads out
                 * of bounds (undefined).  Both this oracle and the HIP kernel stop at the grid border. */
                if (cur[0] < 0 || cur[0] >= rx || cur[1] < 0 || cur[1] >= ry || cur[2] < 0 || cur[2] >= rz) break;
            }
        }
        if (mode == 0) chunk_cnts[tid] = n_samples;
    }
}

/* nerfacc exclusive_sum over packed segments (sequential per ray): out[i] = sum_{j<i in ray} in[j]. */
void lse_oracle_exclusive_sum(const float *in, const int64_t *packed_info, int n_rays, float *out)
{
    for (int r = 0; r < n_rays; ++r) {
        int64_t s = packed_info[2 * r], c = packed_info[2 * r + 1];
        float acc = 0.0f;
        for (int64_t i = s; i < s + c; ++i) { out[i] = acc; acc += in[i]; }
    }
}
