// Occupancy-grid ray marching for gfx950: nerfacc 0.5.2 `traverse_grids` semantics as the reference consumes
// them at R:lse_nerf/lse_grid_estimator.py:93-106 (t_starts = vals[is_left], t_ends = vals[is_right]).
//
// ONE RAY PER WAVE: the march is a serial float chain per ray (each sample edge is the previous one plus dt, and that
// exact sequence of roundings is part of the bit-exact contract), so the kernel is bound by instruction issue along that
// chain, not by lanes.  Giving every ray its own wave makes the ray index wave-uniform (blockIdx.x): the compiler emits
// scalar loads and scalar branches (no exec-mask bookkeeping, which tripled the instruction count of a lane-per-ray
// version), 4096 rays become 4 waves on every SIMD of the chip, and the otherwise idle lanes fetch the look-ahead
// batch's occupancy bytes in one load + ballot.  The slab tests against every level's AABB and
// the boundary sort are folded into the prologue (upstream: a separate kernel + torch.sort).  Samples are
// emitted as (t_last, t_next) pairs, which is exactly what the is_left/is_right mask extraction yields.
//
// This file is compiled with -ffp-contract=off: the integer outputs (counts, ray indices) and the f32 sample
// edges must be bit-identical to the strict-fp32 CPU oracle, so every multiply/add rounds separately and every
// division is IEEE-correct.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kMaxLevels = LSE_MAX_OCC_LEVELS;
constexpr int kBatch = 32;   // DDA cells looked ahead per batch (occupancy loads in flight); 16 until round 3

struct TraverseArgs {
    const float *rays_o, *rays_d;
    int n_rays;
    const uint8_t *binaries;
    const float *aabbs;
    int levels, rx, ry, rz;
    const float *near_planes, *far_planes;
    float step_size, cone_angle;
    int64_t *chunk_cnts;
    const int64_t *chunk_starts;
    int32_t *ray_indices;
    float *t_starts, *t_ends;
    int64_t cap;          // MODE 2: samples of ray r go to slots [r*cap, (r+1)*cap)
    int32_t *overflow;    // MODE 2: OR-ed (never cleared) when a ray produced more than cap samples (never, by the host's bound)
    int vec_march;        // constant step: march 64 steps of a cell at once (march_cell_vec); 0 = published serial loop only
    int fma_setup;        // flags & LSE_TRAVERSE_FMA_SETUP: the a*b+c sites of the traversal setup as fused multiply-adds (nvcc's default
                          // contraction of grid.cu); 0 = every product and sum rounded separately (default, == oracle build 1)
};

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    return fminf(fmaxf(t * cone_angle, dt_min), dt_max);
}

// cone_angle == 0 and 0 < step <= 1e10: clamp(t*0, step, 1e10) == step bit for bit for every finite t, so the
// three-instruction clamp leaves the serial chain (host picks the instantiation).
template <bool CONST_DT>
__device__ __forceinline__ float calc_dt_t(float t, float cone_angle, float dt_min, float dt_max)
{
    return CONST_DT ? dt_min : calc_dt(t, cone_angle, dt_min, dt_max);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// slab test with the upstream early-outs; returns hit and [tmin,tmax] clamped to [near_p, far_p]
__device__ bool slab(const float o[3], const float inv[3], const float *__restrict__ box, float near_p, float far_p,
                     float &tmin, float &tmax)
{
    float lo, hi;
    if (inv[0] >= 0) { tmin = (box[0] - o[0]) * inv[0]; tmax = (box[3] - o[0]) * inv[0]; }
    else             { tmin = (box[3] - o[0]) * inv[0]; tmax = (box[0] - o[0]) * inv[0]; }
    if (inv[1] >= 0) { lo = (box[1] - o[1]) * inv[1]; hi = (box[4] - o[1]) * inv[1]; }
    else             { lo = (box[4] - o[1]) * inv[1]; hi = (box[1] - o[1]) * inv[1]; }
    if (tmin > hi || lo > tmax) return false;
    if (lo > tmin) tmin = lo;
    if (hi < tmax) tmax = hi;
    if (inv[2] >= 0) { lo = (box[2] - o[2]) * inv[2]; hi = (box[5] - o[2]) * inv[2]; }
    else             { lo = (box[5] - o[2]) * inv[2]; hi = (box[2] - o[2]) * inv[2]; }
    if (tmin > hi || lo > tmax) return false;
    if (lo > tmin) tmin = lo;
    if (hi < tmax) tmax = hi;
    if (tmax <= 0) return false;
    tmin = fmaxf(tmin, near_p);
    tmax = fminf(tmax, far_p);
    return true;
}

// Constant step (cone_angle == 0): inside one binade the serial recurrence t <- fl(t + dt) is an exact arithmetic
// progression of the float's BIT PATTERN: every t is a multiple of the binade's ulp u, dt = q*u + r, and round-to-nearest
// turns t + dt into t + q*u or t + (q+1)*u depending on r alone -- unless r is exactly u/2 (a tie, which rounds to even and
// therefore depends on t).  So with dm = bits(fl(t0 + dt)) - bits(t0), the k-th value of the recurrence is
// bits(t0) + k*dm, bit for bit, as long as the exponent does not change and the sum is not a tie.  That lets all 64 lanes
// evaluate 64 consecutive marching steps of a cell at once (lane k: t_k, the published `reached` test on t_k and the
// `cell done` test on t_{k+1}, each with the same float operations the serial loop performs), find the first step that
// stops with a ballot, and store up to 64 samples with one instruction.  Any failed precondition (tie, binade border
// within 65 steps, no progress, tiny or non-positive t, nearly exhausted iteration budget) returns false and the caller
// runs the published serial loop for that cell.  Returns true when the cell is finished.
template <int MODE>
__device__ __forceinline__ bool march_cell_vec(bool occupied, float t_traverse, float dt, float &t_last, uint32_t &n_samples,
                                               bool &continuous, int &budget, int lane, int tid, int32_t *out_ri,
                                               float *out_ts, float *out_te, uint32_t cap32)
{
    constexpr bool WRITE = MODE != 0;
    const float half = dt * 0.5f;
    while (true) {
        const int t0b = __float_as_int(t_last);
        const float t1 = t_last + dt;
        const int dm = __float_as_int(t1) - t0b;
        const float inc = t1 - t_last;                                          // exact (same binade)
        const float r = dt - inc;                                               // exact (Sterbenz)
        const float ulp = __int_as_float(t0b & 0x7f800000) * 1.1920928955078125e-07f;   // 2^(e-23)
        const bool ok = budget > 256 && t_last >= 1e-20f && dm > 0 && dm < (1 << 22) &&
                        (((t0b + 65 * dm) ^ t0b) & 0xff800000) == 0 && fabsf(r) * 2.0f != ulp;
        if (!ok) return false;
        const float tk = __int_as_float(t0b + lane * dm), tn = __int_as_float(t0b + (lane + 1) * dm);
        const bool reached = tk + half >= t_traverse;
        const uint64_t m_reached = __builtin_amdgcn_ballot_w64(reached);
        budget -= 64;
        if (!occupied) {
            if (m_reached == 0) {
                t_last = __int_as_float(t0b + 64 * dm);
                continue;
            }
            t_last = __int_as_float(t0b + __builtin_ctzll(m_reached) * dm);
            return true;
        }
        const uint64_t m_stop = m_reached | __builtin_amdgcn_ballot_w64(tn >= t_traverse);
        int count = 64;
        if (m_stop != 0) {
            const int K = __builtin_ctzll(m_stop);
            count = ((m_reached >> K) & 1) ? K : K + 1;                           // `reached` stops before emitting sample K
        }
        if (WRITE && lane < count && n_samples + (uint32_t)lane < cap32) {
            if (MODE == 1) out_ri[n_samples + lane] = tid;
            out_ts[n_samples + lane] = tk;
            out_te[n_samples + lane] = tn;
        }
        n_samples += (uint32_t)count;
        if (count > 0) continuous = true;
        t_last = __int_as_float(t0b + count * dm);
        if (m_stop != 0) return true;
    }
}

// MODE 0: count per ray; 1: write at chunk_starts (second pass of the published two-pass scheme);
// 2: single pass -- write (t_start, t_end) into fixed-capacity per-ray slots AND count (lse_compact_ray_slots packs them).
template <int MODE, bool CONST_DT>
__global__ __launch_bounds__(64) void traverse_kernel(TraverseArgs a)
{
    __shared__ float s_tt[kBatch];
    __shared__ int s_cell[kBatch];
    const int tid = blockIdx.x;          // wave-uniform ray index
    const int lane = threadIdx.x;
    if (tid >= a.n_rays) return;
    const float eps = 1e-6f;
    const int L = a.levels;

    float o[3], d[3], inv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[tid * 3 + k];
        d[k] = a.rays_d[tid * 3 + k];
        inv[k] = 1.0f / d[k];
    }
    const float ray_tmin = a.near_planes[tid], ray_tmax = a.far_planes[tid];

    // boundary list [t_mins | t_maxs] and its stable ascending order (torch.sort upstream)
    float ts[2 * kMaxLevels];
    int order[2 * kMaxLevels];
    bool hit[kMaxLevels];
    for (int l = 0; l < L; ++l) {
        float t0, t1;
        hit[l] = slab(o, inv, a.aabbs + l * 6, -INFINITY, INFINITY, t0, t1);
        ts[l] = hit[l] ? t0 : INFINITY;
        ts[L + l] = hit[l] ? t1 : INFINITY;
    }
    for (int i = 0; i < 2 * L; ++i) order[i] = i;
    if (L > 1) {
        for (int i = 1; i < 2 * L; ++i) {
            float v = ts[i];
            int k = order[i];
            int j = i - 1;
            while (j >= 0 && ts[j] > v) {
                ts[j + 1] = ts[j];
                order[j + 1] = order[j];
                --j;
            }
            ts[j + 1] = v;
            order[j + 1] = k;
        }
    }

    uint32_t n_samples = 0;   // per ray; the host bounds it far below 2^31 (budget: 2^24 loop iterations)
    // hang guard: with absurd inputs (t ~ 1e10 and a tiny step) `t_last += dt` stops making progress and the
    // published loops never end; every loop below draws from this budget (never reached for sane inputs).
    int budget = 1 << 24;
    int64_t base = 0;
    constexpr bool WRITE = MODE != 0;
    if (MODE == 1) base = a.chunk_starts[tid];
    if (MODE == 2) base = (int64_t)tid * a.cap;
    // per-ray output cursors and a 32-bit capacity: keeps 64-bit address arithmetic out of the marching loop
    int32_t *const out_ri = WRITE ? a.ray_indices + base : nullptr;
    float *const out_ts = WRITE ? a.t_starts + base : nullptr, *const out_te = WRITE ? a.t_ends + base : nullptr;
    const uint32_t cap32 = MODE == 2 ? (uint32_t)(a.cap < 0x7fffffff ? a.cap : 0x7fffffff) : 0xffffffffu;
    float t_last = ray_tmin;
    bool continuous = false;
    const int res[3] = {a.rx, a.ry, a.rz};
    const int cells_per_level = a.rx * a.ry * a.rz;   // levels*cells < 2^31 is checked on the host

    for (int i = 0; i < 2 * L - 1; ++i) {
        const bool entering = order[i] < L;
        int level = order[i] % L;
        if (!hit[level]) continue;
        if (!entering) {
            if (order[i + 1] < L) continue;   // next boundary enters a grid: we are outside it until then
            level = order[i + 1] % L;
            if (!hit[level]) continue;
        }
        const float this_tmin = fmaxf(ts[i], ray_tmin);
        const float this_tmax = fminf(ts[i + 1], ray_tmax);
        if (this_tmin >= this_tmax) continue;

        if (!continuous) {
            if (a.step_size <= 0.0f) {
                t_last = this_tmin;
            } else {
                for (; budget > 0; --budget) {
                    const float dt = calc_dt_t<CONST_DT>(t_last, a.cone_angle, a.step_size, 1e10f);
                    if (t_last + dt * 0.5f >= this_tmin) break;
                    t_last += dt;
                }
            }
        }

        // Amanatides-Woo setup inside this level's box
        const float *box = a.aabbs + level * 6;
        float tdist[3], delta[3];
        int cur[3], stp[3], ovf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float resf = (float)res[k];
            const float ext = box[3 + k] - box[k];
            const float voxel = ext / resf;
            // (this file is compiled with -ffp-contract=off: only the explicit fmaf calls fuse)
            const float rs = a.fma_setup ? fmaf(d[k], this_tmin + eps, o[k]) : o[k] + d[k] * (this_tmin + eps);
            const float re = a.fma_setup ? fmaf(d[k], this_tmax - eps, o[k]) : o[k] + d[k] * (this_tmax - eps);
            cur[k] = clampi((int)(((rs - box[k]) / ext) * resf), 0, res[k] - 1);
            const int fin = clampi((int)(((re - box[k]) / ext) * resf), 0, res[k] - 1);
            const int start_index = cur[k] + (d[k] > 0 ? 1 : 0);
            const float tmax_k = a.fma_setup ? fmaf(box[k] + fmaf((float)start_index, voxel, -rs), inv[k], this_tmin)
                                             : ((box[k] + (((float)start_index * voxel) - rs)) * inv[k]) + this_tmin;
            tdist[k] = (d[k] == 0.0f) ? this_tmax : tmax_k;
            const float stepf = (d[k] == 0.0f) ? 0.0f : (d[k] > 0.0f ? 1.0f : -1.0f);
            stp[k] = (int)stepf;
            const float dl = voxel * inv[k] * stepf;
            delta[k] = (d[k] == 0.0f) ? this_tmax : dl;
            ovf[k] = fin + stp[k];
        }

        // The DDA cell sequence does not depend on the marching, so it is run ahead in batches of kBatch cells with all
        // occupancy loads in flight at once (each is a dependent ~1 us HBM/MALL access when done one by one); the
        // marching then consumes the batch in order.  Same arithmetic, same order of float operations as upstream.
        // Code size matters here: one wave per SIMD walks this loop nest, so it must stay resident in the instruction
        // cache -- the batch lives in LDS (per-lane slots) + a 16-bit occupancy mask, and every phase is a small
        // dynamic loop with ONE copy of the marching code.
        // the linear cell index is carried along (one add per step instead of two multiply-adds), and only the axis that was
        // stepped can leave the grid (the start cell is clamped into it): integer bookkeeping only, the float operations and
        // their order are the published ones
        int cell = (cur[0] * a.ry + cur[1]) * a.rz + cur[2] + level * cells_per_level;
        const int cstep[3] = {stp[0] * a.ry * a.rz, stp[1] * a.rz, stp[2]};
        bool seg_alive = true;
        while (seg_alive && budget > 0) {
            int nb = 0;
            for (int b = 0; b < kBatch && seg_alive; ++b) {
                float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                t_traverse = fminf(t_traverse, this_tmax);
                s_tt[b] = t_traverse;          // every lane writes the same (uniform) value
                s_cell[b] = cell;
                nb = b + 1;
                // step to the neighbour cell (ties: x only if strictly smallest, then y, else z).
                // Leaving the grid without meeting the overflow index is undefined upstream (out-of-bounds read); this
                // implementation stops at the border (DESIGN.md "deviations").
                bool alive = true;
                if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) {
                    cur[0] += stp[0]; tdist[0] += delta[0]; cell += cstep[0];
                    alive = cur[0] != ovf[0] && (unsigned)cur[0] < (unsigned)a.rx;
                } else if (tdist[1] < tdist[2]) {
                    cur[1] += stp[1]; tdist[1] += delta[1]; cell += cstep[1];
                    alive = cur[1] != ovf[1] && (unsigned)cur[1] < (unsigned)a.ry;
                } else {
                    cur[2] += stp[2]; tdist[2] += delta[2]; cell += cstep[2];
                    alive = cur[2] != ovf[2] && (unsigned)cur[2] < (unsigned)a.rz;
                }
                if (!alive) seg_alive = false;
                --budget;
            }
            // lane b fetches the occupancy byte of look-ahead cell b: one load instruction for the whole batch
            __syncthreads();
            const bool my_occ = (lane < nb) ? (a.binaries[s_cell[lane]] != 0) : false;
            const uint64_t occ_mask = __ballot(my_occ);

            if (a.step_size <= 0.0f) {
                for (int b = 0; b < nb; ++b) {   // one interval per occupied cell
                    const float t_traverse = s_tt[b];
                    if ((occ_mask >> b) & 1u) {
                        if (WRITE && n_samples < cap32) {      // (wave-uniform value and address)
                            if (MODE == 1) out_ri[n_samples] = tid;
                            out_ts[n_samples] = t_last;
                            out_te[n_samples] = t_traverse;
                        }
                        n_samples++;
                        continuous = true;
                    } else {
                        continuous = false;
                    }
                    t_last = t_traverse;
                }
            } else {
                // One ray per wave: every branch below is wave-uniform (scalar), so the published nested loops are used as
                // they are -- a tight marching loop per cell (add, compare, two stores, add, compare) instead of a flat
                // one-event-per-iteration loop full of selects; the kernel is bound by instruction issue (4 waves share a
                // SIMD, one lane of 64 does the arithmetic).  Same comparisons and additions in the same order.
                for (int b = 0; b < nb && budget > 0; ++b) {
                    const bool occupied = (occ_mask >> b) & 1u;
                    if (!occupied) {
                        // A run of unoccupied cells is left where its LAST cell is left: the cell boundaries increase along
                        // the ray and an unoccupied cell only advances t while t + dt/2 < boundary, so stepping cell by cell
                        // and stepping against the run's last boundary perform the same additions in the same order.
                        const uint64_t rest = occ_mask >> b;
                        const int run = rest ? min((int)__builtin_ctzll(rest), nb - b) : nb - b;
                        b += run - 1;
                    } else {
                        // A run of OCCUPIED cells is marched against its last boundary as well: inside the run a cell ends either
                        // because the next sample's mid-point has reached the cell's boundary -- the following cell then makes the
                        // same test against its own, later boundary and emits that very sample -- or because the sample's end has
                        // reached the boundary, where the following cell simply continues from that end.  Either way the samples
                        // and the final t are those of marching against the last boundary of the run; `continuous` stays set from
                        // the first emitted sample on.  (Round 3: the metric workload's fully occupied grid is one run per batch,
                        // so the vector march fills its 64 lanes instead of emitting the 5 - 8 samples of one cell.)
                        const uint64_t rest = ~(occ_mask >> b);
                        const int run = min((int)__builtin_ctzll(rest | (1ull << 63)), nb - b);
                        b += run - 1;
                    }
                    const float t_traverse = s_tt[b];
                    if (CONST_DT && a.vec_march &&
                        march_cell_vec<MODE>(occupied, t_traverse, a.step_size, t_last, n_samples, continuous, budget, lane, tid,
                                             out_ri, out_ts, out_te, cap32)) {
                        if (!occupied) continuous = false;
                        continue;
                    }
                    if (occupied) {
                        while (budget > 0) {
                            --budget;
                            const float dt = calc_dt_t<CONST_DT>(t_last, a.cone_angle, a.step_size, 1e10f);
                            if (t_last + dt * 0.5f >= t_traverse) break;
                            const float t_next = t_last + dt;
                            // every lane stores the same (wave-uniform) value to the same address: one request, and no
                            // exec-mask juggling for "lane 0 only" inside the serial loop
                            if (WRITE && n_samples < cap32) {
                                if (MODE == 1) out_ri[n_samples] = tid;
                                out_ts[n_samples] = t_last;
                                out_te[n_samples] = t_next;
                            }
                            n_samples++;
                            continuous = true;
                            t_last = t_next;
                            if (t_next >= t_traverse) break;
                        }
                    } else {
                        while (budget > 0) {
                            --budget;
                            const float dt = calc_dt_t<CONST_DT>(t_last, a.cone_angle, a.step_size, 1e10f);
                            if (t_last + dt * 0.5f >= t_traverse) break;
                            t_last = t_last + dt;
                        }
                        continuous = false;
                    }
                }
            }
            __syncthreads();   // the next batch overwrites the LDS slots
        }
    }
    // MODE 2: the count handed on is CLAMPED to the capacity -- pack_info / lse_compact_ray_slots then never read a neighbour's
    // slots nor write past the R*cap packed arrays: a violated bound truncates the ray and raises the sticky flag (bit 0; bit 1:
    // the direction of such a ray was shorter than 1, which is what the host's bound assumes)
    if (MODE != 1 && lane == 0) a.chunk_cnts[tid] = (int64_t)(MODE == 2 && n_samples > cap32 ? cap32 : n_samples);
    if (MODE == 2 && lane == 0 && n_samples > cap32)
        atomicOr(a.overflow, 1 | ((d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 0.998f ? 2 : 0));
}

// nerfstudio's VolumetricSampler inserts ONE fake sample (ray 0, t_start = t_end = 1) when no ray produced any, so that nothing
// downstream sees empty tensors; with the count on the device the same rule is applied there.
// `feat_*`: the visibility pre-pass's compacted features of the survivors (positions, selector, level-major hash features), which
// the main pass takes instead of encoding the survivors again.  With no survivor nothing was written there, and the fake sample
// lands in slot 0: its slot is ZEROED -- finite whatever the allocator left behind.  The sample has zero extent (weight exactly 0,
// every gradient through it exactly 0), so which finite features it carries cannot reach any output.
__global__ void fake_sample_kernel(int64_t *__restrict__ packed, int64_t *__restrict__ n_dev, int32_t *__restrict__ ri,
                                   float *__restrict__ ts, float *__restrict__ te, float *__restrict__ feat_x01,
                                   uint8_t *__restrict__ feat_sel, float *__restrict__ feat_y, int64_t y_level_stride, int n_levels,
                                   int n_feat)
{
    if (*n_dev != 0) return;        // (read by every lane before lane 0 writes below: one wave, program order)
    const int lane = threadIdx.x;
    if (feat_x01 && lane < 3) feat_x01[lane] = 0.f;
    if (feat_sel && lane == 0) feat_sel[0] = 0;
    if (feat_y)
        for (int i = lane; i < n_levels * n_feat; i += 64) feat_y[(int64_t)(i / n_feat) * y_level_stride + (i % n_feat)] = 0.f;
    if (lane == 0) {
        packed[0] = 0;
        packed[1] = 1;
        ri[0] = 0;
        ts[0] = 1.f;
        te[0] = 1.f;
        *n_dev = 1;
    }
}

// single-workgroup exclusive scan of int64 counts -> packed_info[R,2] and total.  Every thread sums a contiguous run of counts, the
// 64 partial sums of a wave are scanned with lane shuffles, the 16 wave totals by the first wave: two workgroup barriers instead of the
// twenty of a Hillis-Steele scan over 1024 LDS entries.  (The launch stays at ~8 us -- one workgroup, two dependent trips to memory --
// of which the scan was the smaller part: 9.2 -> 8.5 us for 4096 rays.)
__global__ __launch_bounds__(1024) void pack_info_kernel(const int64_t *__restrict__ cnts, int n, int64_t *packed,
                                                         int64_t *total)
{
    __shared__ int64_t wave_tot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += cnts[i];
    // inclusive scan of the 64 partial sums of this wave
    int64_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int64_t w = lane < 16 ? wave_tot[lane] : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int64_t v = __shfl_up(w, off, 64);
            if (lane >= off) w += v;
        }
        if (lane < 16) wave_tot[lane] = w;      // inclusive totals of waves 0 .. lane
    }
    __syncthreads();
    int64_t run = (wave ? wave_tot[wave - 1] : 0) + incl - s;
    for (int i = lo; i < hi; ++i) {
        const int64_t c = cnts[i];
        packed[2 * i] = run;
        packed[2 * i + 1] = c;
        run += c;
    }
    if (t == 1023 && total) *total = wave_tot[15];
}

// Packs the per-ray slots of the single-pass marcher: one wave per ray copies its count samples to the packed position.
__global__ __launch_bounds__(256) void compact_slots_kernel(const float *__restrict__ ts_slots, const float *__restrict__ te_slots,
                                                            int64_t cap, const int64_t *__restrict__ packed_info, int n_rays,
                                                            int32_t *__restrict__ ray_indices, float *__restrict__ t_starts,
                                                            float *__restrict__ t_ends)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    const int64_t start = packed_info[2 * (int64_t)ray], cnt = packed_info[2 * (int64_t)ray + 1];
    const float *src_s = ts_slots + (int64_t)ray * cap, *src_e = te_slots + (int64_t)ray * cap;
    for (int64_t k = lane; k < cnt; k += 64) {
        ray_indices[start + k] = ray;
        t_starts[start + k] = src_s[k];
        t_ends[start + k] = src_e[k];
    }
}

// near / far planes per ray exactly as R:lse_nerf/lse_grid_estimator.py:83-92 forms them with torch ops:
//   near = max(near_plane, t_min[r]);  far = min(far_plane, t_max[r]);  stratified: near += u[r] * step
// (f32 multiply, then f32 add -- this file is built with -ffp-contract=off, so the two roundings match torch's two kernels).
__global__ void ray_planes_kernel(float near_plane, float far_plane, const float *__restrict__ t_min,
                                  const float *__restrict__ t_max, const float *__restrict__ jitter, float step, int n_rays,
                                  float *__restrict__ near_out, float *__restrict__ far_out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float nr = near_plane, fr = far_plane;
    if (t_min) nr = fmaxf(nr, t_min[r]);
    if (t_max) fr = fminf(fr, t_max[r]);
    if (jitter) {
        const float j = jitter[r] * step;
        nr = nr + j;
    }
    near_out[r] = nr;
    far_out[r] = fr;
}

static int vec_march_enabled()
{
    return (int)lse::option("traverse_vec");
}

}  // namespace

extern "C" int lse_traverse_grids(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries,
                                  const float *aabbs, int32_t levels, int32_t rx, int32_t ry, int32_t rz,
                                  const float *near_planes, const float *far_planes, float step_size,
                                  float cone_angle, int32_t mode, int64_t *chunk_cnts, const int64_t *chunk_starts,
                                  int32_t *ray_indices, float *t_starts, float *t_ends, int32_t flags, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_traverse_grids: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_o && rays_d && binaries && aabbs && near_planes && far_planes, "lse_traverse_grids: null input");
    LSE_REQUIRE(levels >= 1 && levels <= LSE_MAX_OCC_LEVELS, "lse_traverse_grids: levels %d not in [1,%d]", levels,
                LSE_MAX_OCC_LEVELS);
    LSE_REQUIRE(rx > 0 && ry > 0 && rz > 0, "lse_traverse_grids: bad resolution");
    LSE_REQUIRE((int64_t)levels * rx * ry * rz < (1ll << 31), "lse_traverse_grids: grid too large (levels*cells >= 2^31)");
    LSE_REQUIRE(mode == 0 || mode == 1, "lse_traverse_grids: mode must be 0 (count) or 1 (write)");
    if (mode == 0) LSE_REQUIRE(chunk_cnts, "lse_traverse_grids: count pass needs chunk_cnts");
    if (mode == 1) LSE_REQUIRE(chunk_starts && ray_indices && t_starts && t_ends, "lse_traverse_grids: write pass needs outputs");
    TraverseArgs a{rays_o, rays_d, n_rays, binaries, aabbs, levels, rx, ry, rz, near_planes, far_planes,
                   step_size, cone_angle, chunk_cnts, chunk_starts, ray_indices, t_starts, t_ends, 0, nullptr, vec_march_enabled(), (flags & LSE_TRAVERSE_FMA_SETUP) ? 1 : 0};
    const int blocks = n_rays;   // one wave per ray
    const bool const_dt = cone_angle == 0.0f && step_size > 0.0f && step_size <= 1e10f;
    hipStream_t st = lse::as_stream(stream);
    if (mode == 0) {
        if (const_dt) hipLaunchKernelGGL((traverse_kernel<0, true>), dim3(blocks), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((traverse_kernel<0, false>), dim3(blocks), dim3(64), 0, st, a);
    } else {
        if (const_dt) hipLaunchKernelGGL((traverse_kernel<1, true>), dim3(blocks), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((traverse_kernel<1, false>), dim3(blocks), dim3(64), 0, st, a);
    }
    return lse::check_launch("lse_traverse_grids");
}

extern "C" int lse_traverse_grids_slots(const float *rays_o, const float *rays_d, int32_t n_rays, const uint8_t *binaries,
                                        const float *aabbs, int32_t levels, int32_t rx, int32_t ry, int32_t rz,
                                        const float *near_planes, const float *far_planes, float step_size,
                                        float cone_angle, int64_t cap, int64_t *chunk_cnts, float *t_start_slots,
                                        float *t_end_slots, int32_t *overflow, int32_t flags, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_traverse_grids_slots: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_o && rays_d && binaries && aabbs && near_planes && far_planes, "lse_traverse_grids_slots: null input");
    LSE_REQUIRE(levels >= 1 && levels <= LSE_MAX_OCC_LEVELS, "lse_traverse_grids_slots: levels %d not in [1,%d]", levels,
                LSE_MAX_OCC_LEVELS);
    LSE_REQUIRE(rx > 0 && ry > 0 && rz > 0, "lse_traverse_grids_slots: bad resolution");
    LSE_REQUIRE((int64_t)levels * rx * ry * rz < (1ll << 31), "lse_traverse_grids_slots: grid too large (levels*cells >= 2^31)");
    LSE_REQUIRE(cap >= 1 && chunk_cnts && t_start_slots && t_end_slots && overflow, "lse_traverse_grids_slots: bad outputs");
    TraverseArgs a{rays_o, rays_d, n_rays, binaries, aabbs, levels, rx, ry, rz, near_planes, far_planes,
                   step_size, cone_angle, chunk_cnts, nullptr, nullptr, t_start_slots, t_end_slots, cap, overflow, vec_march_enabled(), (flags & LSE_TRAVERSE_FMA_SETUP) ? 1 : 0};
    const bool const_dt = cone_angle == 0.0f && step_size > 0.0f && step_size <= 1e10f;
    hipStream_t st = lse::as_stream(stream);
    if (const_dt) hipLaunchKernelGGL((traverse_kernel<2, true>), dim3(n_rays), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((traverse_kernel<2, false>), dim3(n_rays), dim3(64), 0, st, a);
    return lse::check_launch("lse_traverse_grids_slots");
}

extern "C" int lse_compact_ray_slots(const float *t_start_slots, const float *t_end_slots, int64_t cap,
                                     const int64_t *packed_info, int32_t n_rays, int32_t *ray_indices, float *t_starts,
                                     float *t_ends, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_compact_ray_slots: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_start_slots && t_end_slots && packed_info && ray_indices && t_starts && t_ends && cap >= 1,
                "lse_compact_ray_slots: null pointer");
    hipLaunchKernelGGL(compact_slots_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_start_slots,
                       t_end_slots, cap, packed_info, n_rays, ray_indices, t_starts, t_ends);
    return lse::check_launch("lse_compact_ray_slots");
}

extern "C" int lse_pack_info_from_counts(const int64_t *chunk_cnts, int32_t n_rays, int64_t *packed_info,
                                         int64_t *total, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_pack_info_from_counts: n_rays < 0");
    LSE_REQUIRE(n_rays == 0 || (chunk_cnts && packed_info), "lse_pack_info_from_counts: null pointer");
    hipLaunchKernelGGL(pack_info_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), chunk_cnts, n_rays, packed_info,
                       total);
    return lse::check_launch("lse_pack_info_from_counts");
}

extern "C" int lse_fake_sample_if_empty(int64_t *packed_info, int32_t n_rays, int64_t *n_dev, int32_t *ray_indices,
                                        float *t_starts, float *t_ends, float *feat_x01, uint8_t *feat_sel, float *feat_y,
                                        int64_t y_level_stride, int32_t n_levels, int32_t n_features, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_fake_sample_if_empty: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(packed_info && n_dev && ray_indices && t_starts && t_ends, "lse_fake_sample_if_empty: null pointer");
    LSE_REQUIRE(!feat_y || (y_level_stride >= n_features && n_levels >= 1 && n_features >= 1),
                "lse_fake_sample_if_empty: feat_y needs its level stride (floats) and shape");
    hipLaunchKernelGGL(fake_sample_kernel, dim3(1), dim3(64), 0, lse::as_stream(stream), packed_info, n_dev, ray_indices, t_starts,
                       t_ends, feat_x01, feat_sel, feat_y, y_level_stride, n_levels, n_features);
    return lse::check_launch("lse_fake_sample_if_empty");
}

extern "C" int lse_ray_planes(float near_plane, float far_plane, const float *t_min, const float *t_max,
                              const float *jitter, float step_size, int32_t n_rays, float *near_planes,
                              float *far_planes, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_planes: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(near_planes && far_planes, "lse_ray_planes: null pointer");
    hipLaunchKernelGGL(ray_planes_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, lse::as_stream(stream), near_plane,
                       far_plane, t_min, t_max, jitter, step_size, n_rays, near_planes, far_planes);
    return lse::check_launch("lse_ray_planes");
}
