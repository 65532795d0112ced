// Multiresolution hash-grid encoding for gfx950 (tiny-cuda-nn 1.7 `HashGrid` semantics, Linear interpolation,
// F = 2 features per entry) -- the kernels behind `tcnn.Encoding` as built at R:lse_nerf/lse_field.py:72-86 and
// evaluated at R:lse_nerf/lse_field.py:279.
//
// MI355X-first layout decisions
//   * features are produced level-major, y[L][N][2]: a wave writes 64 consecutive float2 (512 contiguous
//     bytes) per level, and the fused MLP reads the same array as MFMA B-operands with full-line loads;
//   * lanes are consecutive samples.  Samples of one ray are consecutive, so at the coarse levels the 64
//     lanes of a gather instruction fall into a handful of cells (same cache lines) and the hardware
//     coalesces them; only the fine hashed levels are true 8-byte random gathers;
//   * forward: one (level, sample-chunk) per workgroup, LEVEL-MAJOR in dispatch order (finest level first): workgroups
//     are dispatched in order and dealt round-robin over the 8 XCDs, so at any time all XCDs work on the same level and
//     each private 4 MiB L2 holds just that 4 MB table.  (Binding whole levels to single XCDs -- blockIdx % 8 -- gives the
//     same locality but the levels cost 0.03 ... 0.10 ms each, and the two XCDs that own two fine levels set the time:
//     0.85 ms against 0.60 ms; tools/hash_fwd_level_cost.py.)  Placement affects speed only, never results.
//   * backward: 16 lanes per sample (one per corner x feature) with run-length pre-accumulation, see hash_bwd_kernel;
//     table gradients are f32 global atomics (memory-side on gfx950, so no level/XCD affinity is attempted there).
#include "common.h"
#include <stdlib.h>

namespace {

struct GridParams {
    int n_levels;
    int l_begin, l_end;   // backward only: levels [l_begin, l_end) of this launch
    int dx_accumulate;    // backward only: dx += instead of dx = (second launch of a split backward)
    // backward only: the direct (cache-free) adds of levels < rep_levels go to one of rep_mask + 1 replicas of those levels'
    // gradient (workspace, rep_stride floats apart, chosen by the workgroup index) instead of the table gradient itself
    int rep_levels, rep_mask;
    uint32_t rep_stride;
    uint32_t offsets[LSE_MAX_GRID_LEVELS + 1];
    float scales[LSE_MAX_GRID_LEVELS];
    uint32_t res[LSE_MAX_GRID_LEVELS];
};

constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;

struct LevelInfo {
    uint32_t offset, size, res;
    float scale;
    bool dense, pow2;
};

__device__ __forceinline__ LevelInfo level_info(const GridParams &g, int l)
{
    LevelInfo li;
    li.offset = g.offsets[l];
    li.size = g.offsets[l + 1] - g.offsets[l];
    li.res = g.res[l];
    li.scale = g.scales[l];
    // tcnn grid_index(): stride loop guarded by stride <= hashmap_size; hash iff hashmap_size < final stride
    // (computed here with scalar instructions: fetching precomputed per-level flags from the kernel arguments was slower)
    uint64_t stride = 1;
    for (int d = 0; d < 3 && stride <= li.size; ++d) stride *= li.res;
    li.dense = !(li.size < stride);
    li.pow2 = (li.size & (li.size - 1)) == 0;
    return li;
}

__device__ __forceinline__ uint32_t grid_index(const LevelInfo &li, uint32_t px, uint32_t py, uint32_t pz)
{
    uint32_t idx;
    if (li.dense) {
        idx = px + py * li.res + pz * li.res * li.res;
        // == idx % size: px,py,pz <= res so idx < 2*res^3 <= 2*size
        if (idx >= li.size) idx -= li.size;
        if (idx >= li.size) idx %= li.size;   // never taken for inputs in [0,1]; keeps the exact modulo otherwise
    } else {
        idx = px ^ (py * kPrimeY) ^ (pz * kPrimeZ);
        idx = li.pow2 ? (idx & (li.size - 1)) : (idx % li.size);
    }
    return idx;
}

// The 8 corner indices of the cell (p0, p1, p2) at once.  Same values as grid_index() per corner (all arithmetic is mod 2^32:
// (p + 1) * prime == p * prime + prime), but the two quarter-rate 32-bit multiplies of the hash are issued once per sample
// instead of once per corner, and a dense level costs one multiply-add pair plus wave-uniform offsets (the per-corner form
// compiles to 16 v_mul_lo_u32 / v_mad_u64_u32 per sample and level, each a 16-cycle instruction on CDNA).
__device__ __forceinline__ void corner_indices(const LevelInfo &li, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t (&idx)[8])
{
    if (li.dense) {
        const uint32_t r2 = li.res * li.res;                       // (scalar)
        const uint32_t b = p0 + p1 * li.res + p2 * r2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t i = b + ((c & 1) ? 1u : 0u) + ((c & 2) ? li.res : 0u) + ((c & 4) ? r2 : 0u);
            if (i >= li.size) i -= li.size;                        // == i % size for inputs in [0,1] (see grid_index)
            if (i >= li.size) i %= li.size;
            idx[c] = i;
        }
    } else {
        const uint32_t hy0 = p1 * kPrimeY, hz0 = p2 * kPrimeZ;
        const uint32_t hy1 = hy0 + kPrimeY, hz1 = hz0 + kPrimeZ;
        const uint32_t h[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t i = (p0 + (c & 1)) ^ h[c >> 1];
            idx[c] = li.pow2 ? (i & (li.size - 1)) : (i % li.size);
        }
    }
}

__device__ __forceinline__ void pos_fract(float x, float scale, float &w, uint32_t &p)
{
    const float pos = fmaf(scale, x, 0.5f);
    const float fl = floorf(pos);
    p = (uint32_t)(int)fl;
    w = pos - fl;
}

constexpr int kFwdThreads = 256;
#ifndef LSE_FWD_ITEMS
#define LSE_FWD_ITEMS 4
#endif
constexpr int kFwdItems = LSE_FWD_ITEMS;   // samples per lane -> 32 independent gathers in flight

__global__ __launch_bounds__(kFwdThreads) void hash_fwd_kernel(GridParams g, const float *__restrict__ x,
                                                               const float2 *__restrict__ table,
                                                               float2 *__restrict__ y, int64_t n_cap, int64_t chunks,
                                                               int mapping, const int64_t *__restrict__ n_dev, int l_first)
{
    // n_cap: the level stride of y and the sample capacity the grid was sized for; n: the samples that exist (device-side
    // count when one is set: workgroups past it leave at once)
    const int64_t n = lse::clamp_count(n_cap, n_dev);
    const int L = g.n_levels;
    int level;
    int64_t chunk;
    const int bid = blockIdx.x;
    if ((L & 7) == 0 && mapping == 0) {
        // XCD-affine AND level-sequential: blockIdx % 8 fixes the level residue class, and because workgroups are
        // dispatched in order, each XCD finishes all chunks of one level before starting its next one, so its 4 MiB
        // L2 holds one 4 MB level table at a time.
        const int64_t slot = bid >> 3;
        level = (bid & 7) + 8 * (int)(slot / chunks);
        chunk = slot % chunks;
    } else if (mapping == 3) {
        // level-major: all chunks of level 0, then level 1, ...  Workgroups are dispatched in order and dealt round-robin
        // to the XCDs, so at any time all 8 XCDs work on the same level (or two neighbouring ones) and every L2 holds
        // just that 4 MB table -- without binding whole levels to single XCDs, whose costs differ 3x between levels.
        level = (int)(bid / chunks);
        chunk = bid % chunks;
    } else if (mapping == 4) {   // level-major, finest level first (short tail on a cheap level); levels < l_first are not part of
        level = L - 1 - (int)(bid / chunks);       // this launch (grid sized for L - l_first levels: hash_fwd_lds_kernel serves them)
        chunk = bid % chunks;
    } else if ((L & 7) == 0 && mapping == 1) {   // XCD-affine, levels interleaved (A/B reference)
        const int per = L >> 3;
        const int slot = bid >> 3;
        level = (bid & 7) + 8 * (slot % per);
        chunk = slot / per;
    } else {
        level = bid % L;
        chunk = bid / L;
    }
    const LevelInfo li = level_info(g, level);
    const float2 *__restrict__ tab = table + li.offset;
    const int64_t base = chunk * (int64_t)(kFwdThreads * kFwdItems) + threadIdx.x;
    if (chunk * (int64_t)(kFwdThreads * kFwdItems) >= n) return;

    float w[kFwdItems][3];
    uint32_t p[kFwdItems][3];
    bool valid[kFwdItems];
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
        const int64_t i = base + (int64_t)it * kFwdThreads;
        valid[it] = i < n;
        const int64_t ii = valid[it] ? i : (n - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) pos_fract(x[ii * 3 + d], li.scale, w[it][d], p[it][d]);
    }
    float2 v[kFwdItems][8];
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
        uint32_t idx[8];
        corner_indices(li, p[it][0], p[it][1], p[it][2], idx);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[it][c] = tab[idx[c]];
    }
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
        float2 r = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wt = 1.f;
            wt *= (c & 1) ? w[it][0] : 1.f - w[it][0];
            wt *= (c & 2) ? w[it][1] : 1.f - w[it][1];
            wt *= (c & 4) ? w[it][2] : 1.f - w[it][2];
            r.x = fmaf(wt, v[it][c].x, r.x);
            r.y = fmaf(wt, v[it][c].y, r.y);
        }
        const int64_t i = base + (int64_t)it * kFwdThreads;
        if (valid[it]) y[(int64_t)level * n_cap + i] = r;
    }
}

// LDS-resident variant for the coarsest levels (BASELINE.json north_star: "coalesced HBM gathers with LDS-staged trilinear
// interpolation"; option hash_fwd_lds_levels, A/B partner of the L2-resident schedule above).  A workgroup stages the WHOLE table of
// one dense level in LDS with coalesced 8-byte loads (level 0: 16^3 = 32 KB, level 1: 23^3 = 95 KB; level 2 = 233 KB does not fit
// the 160 KB of a CU), then walks a contiguous share of the samples: positions in, 8 ds_read_b64 gathers per sample, the same
// multiply-adds in the same order as hash_fwd_kernel (bit-identical results), y out.  Consecutive samples of a ray fall into the same
// coarse cell, so most gathers of a wave-instruction are LDS broadcasts.
#ifdef LSE_DEV_KNOBS   // development build only (csrc/dev_knobs.h): superseded / A-B variant, not in liblse_hip.so
constexpr int kFwdLdsThreads = 1024;
__global__ __launch_bounds__(kFwdLdsThreads) void hash_fwd_lds_kernel(GridParams g, int level, const float *__restrict__ x,
                                                                      const float2 *__restrict__ table, float2 *__restrict__ y,
                                                                      int64_t n_cap, const int64_t *__restrict__ n_dev)
{
    extern __shared__ float2 s_tab[];
    const int64_t n = lse::clamp_count(n_cap, n_dev);
    const LevelInfo li = level_info(g, level);
    const float2 *__restrict__ tab = table + li.offset;
    for (uint32_t e = threadIdx.x; e < li.size; e += kFwdLdsThreads) s_tab[e] = tab[e];
    __syncthreads();
    // contiguous share of this workgroup, in steps of one sample per thread (adjacent lanes = adjacent samples of a ray)
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + kFwdLdsThreads - 1) / kFwdLdsThreads * kFwdLdsThreads;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(n, lo + per);
    float2 *__restrict__ yl = y + (int64_t)level * n_cap;
    for (int64_t i0 = lo; i0 < hi; i0 += 2 * kFwdLdsThreads) {
        float w[2][3];
        uint32_t p[2][3];
        bool valid[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int64_t i = i0 + it * kFwdLdsThreads + threadIdx.x;
            valid[it] = i < hi;
            const int64_t ii = valid[it] ? i : (hi - 1);
#pragma unroll
            for (int d = 0; d < 3; ++d) pos_fract(x[ii * 3 + d], li.scale, w[it][d], p[it][d]);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            uint32_t idx[8];
            corner_indices(li, p[it][0], p[it][1], p[it][2], idx);
            float2 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = s_tab[idx[c]];
            float2 r = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float wt = 1.f;
                wt *= (c & 1) ? w[it][0] : 1.f - w[it][0];
                wt *= (c & 2) ? w[it][1] : 1.f - w[it][1];
                wt *= (c & 4) ? w[it][2] : 1.f - w[it][2];
                r.x = fmaf(wt, v[c].x, r.x);
                r.y = fmaf(wt, v[c].y, r.y);
            }
            if (valid[it]) yl[i0 + it * kFwdLdsThreads + threadIdx.x] = r;
        }
    }
}
#endif  // LSE_DEV_KNOBS

// Backward.  Lane mapping: 16 lanes per sample -- lane k of a 16-lane group owns (corner k>>1, feature k&1) -- and a
// wave owns 64 consecutive samples of the packed (ray-sorted) stream, processed in 16 rounds of 4 samples.
//   * one atomic wave-instruction covers 4 samples x 8 corners x 2 features: the two features of an entry and the
//     x / x+1 corner pair are neighbouring dwords, so the instruction touches far fewer 64-byte lines than lanes
//     (memory-side float atomics are priced per line request, MI355X_MICROARCH.md "Global float atomics"; measured
//     here: identical rate for agent / workgroup / wavefront scope, tools/micro/atomic_scope.hip);
//   * every lane run-length accumulates (index, sum) in registers and only issues an atomic when its entry index
//     changes between consecutive rounds;
//   * which 4 samples share a round is chosen PER LEVEL (tools/sim_hash_bwd_requests.py models the request counts):
//       blocked      group g walks samples 16g + r  -> long runs: best where consecutive samples share a cell (coarse/mid);
//       interleaved  group g takes sample 4r + g     -> the instruction covers 4 CONSECUTIVE samples whose cells share
//                    faces, i.e. lines: best at the fine levels (~25 % fewer line requests there);
//   * positions are staged once per wave in LDS (either mapping reads them by sample id); d(x) is reduced over the 16
//     lanes of a group and accumulated per sample in LDS by the group's first lane (plain read-add-write: within a
//     round the 4 groups own 4 different samples, rounds are sequential).
template <bool WITH_DX, int kRounds>
__global__ __launch_bounds__(256) void hash_bwd_kernel(GridParams g, const float *__restrict__ x,
                                                       const float *__restrict__ dy,
                                                       const float *__restrict__ table, float *__restrict__ dtable,
                                                       float *__restrict__ dx, int64_t n, float interleave_from_scale, int dbg)
{
    constexpr int kChunk = 4 * kRounds;        // samples per wave
    // SoA + one pad word per 16-lane group: the 4 groups of a wave touch slots kRounds apart, which would otherwise fall
    // into the same LDS bank (4-way conflict on every position read and every d(x) update)
    constexpr int kPitch = kChunk + 4;
    __shared__ float s_x[4][3][kPitch];
    __shared__ float s_dx[4][3][kPitch];
#define LSE_SLOT(sl) ((sl) + (sl) / kRounds)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, k = lane & 15;
    const int corner = k >> 1, f = k & 1;
    const int64_t wave_base = ((int64_t)blockIdx.x * 4 + wave) * kChunk;
    if (wave_base >= n) return;
#pragma unroll
    for (int c = lane; c < kChunk; c += 64) {
        const int64_t i = wave_base + c;
        const int64_t ii = i < n ? i : n - 1;
        s_x[wave][0][LSE_SLOT(c)] = x[ii * 3 + 0];
        s_x[wave][1][LSE_SLOT(c)] = x[ii * 3 + 1];
        s_x[wave][2][LSE_SLOT(c)] = x[ii * 3 + 2];
        if (WITH_DX) s_dx[wave][0][LSE_SLOT(c)] = s_dx[wave][1][LSE_SLOT(c)] = s_dx[wave][2][LSE_SLOT(c)] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();   // LDS slots are private to this wave; DS ops of a wave execute in order
    const uint32_t cx = corner & 1, cy = (corner >> 1) & 1, cz = (corner >> 2) & 1;
    constexpr uint32_t kNone = 0xFFFFFFFFu;

    for (int l = g.l_begin; l < g.l_end; ++l) {
        const LevelInfo li = level_info(g, l);
        const bool interleaved = li.scale >= interleave_from_scale;
        float *__restrict__ dt = dtable + 2 * (size_t)li.offset + f;
        const float *__restrict__ tab = WITH_DX ? table + 2 * (size_t)li.offset + f : nullptr;
        const float *__restrict__ dyl = dy + 2 * (size_t)l * n + f;
        uint32_t cur = kNone;
        float acc = 0.f;
#pragma unroll 4
        for (int r = 0; r < kRounds; ++r) {
            const int sl = interleaved ? (4 * r + grp) : (kRounds * grp + r);   // sample slot within the wave's 64
            const bool valid = wave_base + sl < n;
            float w0, w1, w2;
            uint32_t p0, p1, p2;
            const int slot = LSE_SLOT(sl);
            pos_fract(s_x[wave][0][slot], li.scale, w0, p0);
            pos_fract(s_x[wave][1][slot], li.scale, w1, p1);
            pos_fract(s_x[wave][2][slot], li.scale, w2, p2);
            const uint32_t idx = valid ? grid_index(li, p0 + cx, p1 + cy, p2 + cz) : kNone;
            const float sx = cx ? w0 : 1.f - w0, sy = cy ? w1 : 1.f - w1, sz = cz ? w2 : 1.f - w2;
            const int64_t ii = valid ? wave_base + sl : n - 1;
            const float gy = dyl[2 * ii];
            const float v = sx * sy * sz * gy;
            if (idx == cur) {
                acc += v;
            } else {
                if (cur != kNone && !(dbg & 1)) atomicAdd(dt + 2 * (size_t)cur, acc);
                cur = idx;
                acc = v;
            }
            if (WITH_DX) {
                const float tv = valid ? tab[2 * (size_t)idx] : 0.f;
                const float t = li.scale * gy * tv;
                float d0 = t * (cx ? 1.f : -1.f) * (sy * sz);
                float d1 = t * (cy ? 1.f : -1.f) * (sx * sz);
                float d2 = t * (cz ? 1.f : -1.f) * (sx * sy);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    d0 += __shfl_xor(d0, o, 64);
                    d1 += __shfl_xor(d1, o, 64);
                    d2 += __shfl_xor(d2, o, 64);
                }
                if (k == 0) {
                    s_dx[wave][0][slot] += d0;
                    s_dx[wave][1][slot] += d1;
                    s_dx[wave][2][slot] += d2;
                }
            }
        }
        if (cur != kNone && !(dbg & 1)) atomicAdd(dt + 2 * (size_t)cur, acc);
    }
    if (WITH_DX) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = lane; c < kChunk; c += 64) {
            const int64_t i = wave_base + c;
            if (i < n) {
                const float a0 = g.dx_accumulate ? dx[i * 3 + 0] : 0.f, a1 = g.dx_accumulate ? dx[i * 3 + 1] : 0.f,
                            a2 = g.dx_accumulate ? dx[i * 3 + 2] : 0.f;
                dx[i * 3 + 0] = a0 + s_dx[wave][0][LSE_SLOT(c)];
                dx[i * 3 + 1] = a1 + s_dx[wave][1][LSE_SLOT(c)];
                dx[i * 3 + 2] = a2 + s_dx[wave][2][LSE_SLOT(c)];
            }
        }
    }
#undef LSE_SLOT
}

// Backward, line-cache variant (the default; LSE_HASH_BWD_IMPL=0 selects the kernel above, kept as the A/B reference).
// The 16-lanes-per-sample kernel recomputes the position maths 16x and sends ~19 line requests per sample to the
// memory-side atomic units (it is bound by them: 4.4 ms with, 1.2 ms without the atomics).  Here:
//   * lane = sample (64 consecutive samples of the packed, ray-sorted stream): the position maths, the 8 indices and
//     d(x) are computed once per sample, d(x) needs no cross-lane reduction at all;
//   * consecutive lanes that sit in the same cell form a run; a flag-based segmented scan (DPP inside the 16-lane rows,
//     early exit once no run is longer than the step, serial carry over the 3 row borders) leaves each run's 8 x 2
//     corner sums in the run's last lane;
//   * run ends add into a per-wave LDS cache of table SECTORS (key = entry index >> 2, payload 8 floats = one 32-B
//     sector of 4 entries x 2 features; 512 slots, one probe): a 32-bit compare-and-swap claims the slot, a 64-bit
//     compare-and-swap adds both features (ds_add_f32 runs at ~3 cycles per LANE on gfx950, tools/micro/lds_ops.hip);
//     a corner whose slot belongs to another sector goes straight to memory, so the cache is purely an optimisation;
//   * after each level the occupied slots (kept in a list) are flushed with one 8-lane x 32-B atomic per sector.
//   * a wave that ends at most `few_runs` (6) runs at a level -- the coarse levels -- skips the cache: the run ends park
//     their 8 indices + 16 sums in LDS and the wave adds them to memory with 16 lanes per run (lane = corner x feature),
//     because the ~500 instructions of the cache path would be spent on idle lanes (levels 0-3: 0.66 -> 0.37 ms).
//   The memory-side atomic units are priced per 32-B sector request at ~20 G/s (rocprof WRITE_SIZE / 32 B tracks the
//   kernel time).  tools/sim_hash_bwd_requests.py models the policy: 256 slots of a 64-B line flush 9.1 sectors per
//   sample and send 4.2 corner updates straight to memory (3.54 ms); 512 sector-sized slots in the same 16 KB collide
//   less: 9.8 + 2.6 (measured 12.3 sector writes per sample, 3.23 ms); writing a collision's two features through a lane
//   pair in one instruction instead of two: 3.05 ms; the few-runs path: 2.91 ms.  The kernel above writes 21 sectors per
//   sample (4.37 ms).  It now sits at the atomic rate for its ~13 requests per sample (2.7 ms); the issue slots
//   (VALU+SALU+LDS ~ 75 % of the SIMD cycles before the last two steps, at the 2 waves/SIMD that 79 KB of LDS allow) are
//   the second bound.  Tried and dropped: more probe rounds (second_probe: neutral), two half-wave phases per fine
//   level, ds_add_f32 payload adds, a slot function linear in the cell coordinates, one probe per x-row.
constexpr uint32_t kNoLine = 0xFFFFFFFFu;

// LDS pointers carry their address space so that every cache access is a ds_* instruction (generic pointers make the
// compiler fold the LDS and the global fall-back path into flat atomics).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

// relaxed LDS compare-and-swap returning the previous value (no ordering fences: DS ops of a wave execute in order)
__device__ __forceinline__ uint32_t lds_cas(lds_u32 *p, uint32_t expect, uint32_t desired)
{
    __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return expect;
}
__device__ __forceinline__ uint64_t lds_cas(lds_u64 *p, uint64_t expect, uint64_t desired)
{
    __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return expect;
}

__device__ __forceinline__ uint64_t add_pair(uint64_t bits, float a0, float a1)
{
    const float lo = __uint_as_float((uint32_t)bits) + a0, hi = __uint_as_float((uint32_t)(bits >> 32)) + a1;
    return (uint64_t)__float_as_uint(lo) | ((uint64_t)__float_as_uint(hi) << 32);
}

// Row-local DPP shift: lane i receives the value of lane i-OFF of its 16-lane row, 0 if that lane is outside the row.
template <int OFF>
__device__ __forceinline__ float row_shr_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + OFF, 0xF, 0xF, true));
}
template <int OFF>
__device__ __forceinline__ int row_shr_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + OFF, 0xF, 0xF, true);
}

// Whole-wave shifts by one lane as DPP moves (GFX9 wave_shr / wave_shl): `__shfl_up(v, 1)` compiles to ds_bpermute_b32, an LDS
// instruction, and the hash backward is bound by the LDS pipe.  Lane 0 (63) receives its own value; callers test the lane.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false); }

// lane <-> lane^1 exchange (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ int quad_swap1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }

// acc += (value of acc in the lane OFF below, same 16-lane row; 0 outside the row) * mul -- one fused DPP instruction
// (the compiler emits v_mov_b32_dpp + v_fma for the builtin form).  The s_nop covers the "VALU write -> DPP read" wait
// states for the first value of a step; the 16 values of a step are independent of each other.
#define LSE_FMAC_DPP(acc, mul, CTRL) asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL : "+v"(acc) : "v"(mul))

// One step of the flag-based segmented inclusive scan inside the 16-lane rows.
// Returns false (wave-uniform) when no lane needs this or any later step.
template <int OFF>
__device__ __forceinline__ bool seg_scan_row_step(float (&v)[16], int &flag, int row_pos)
{
    const bool need = !flag && row_pos >= OFF;
    if (__builtin_amdgcn_ballot_w64(need) == 0) return false;
    const float nf = need ? 1.f : 0.f;
    const int fo = row_shr_i32<OFF>(flag);
    asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (OFF == 1) LSE_FMAC_DPP(v[k], nf, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (OFF == 2) LSE_FMAC_DPP(v[k], nf, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (OFF == 4) LSE_FMAC_DPP(v[k], nf, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (OFF == 8) LSE_FMAC_DPP(v[k], nf, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    }
    flag |= need ? fo : 0;
    return true;
}

// Carry of the row-local scans across the three row borders: rows 1 and 3 take the last lane of rows 0 and 2
// (row_bcast:15), then rows 2 and 3 take lane 31 (row_bcast:31), each lane only while its run reaches back that far.
// open = "no run head between my row's first lane and me".
__device__ __forceinline__ void seg_scan_cross_rows(float (&v)[16], int open, int lane)
{
    const int row = lane >> 4;
    if (__builtin_amdgcn_ballot_w64(open && row != 0) == 0) return;
    const float fa = (open && (row & 1)) ? 1.f : 0.f;
    int open47 = open;       // rows 1, 3 receive open(lane 15), open(lane 47)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(open47));
#pragma unroll
    for (int k = 0; k < 16; ++k) LSE_FMAC_DPP(v[k], fa, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    const float fb = (row == 2 ? open : (row == 3 ? (open && open47) : 0)) ? 1.f : 0.f;
    asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < 16; ++k) LSE_FMAC_DPP(v[k], fb, "row_bcast:31 row_mask:0xc bank_mask:0xf");
}

#ifdef LSE_DEV_KNOBS   // development build only (csrc/dev_knobs.h): superseded / A-B variant, not in liblse_hip.so
template <bool WITH_DX, int kSlots, int kEntLog2>
__global__ __launch_bounds__(256) void hash_bwd_cached_kernel(GridParams g, const float *__restrict__ x,
                                                              const float2 *__restrict__ dy,
                                                              const float2 *__restrict__ table,
                                                              float *__restrict__ dtable, float *__restrict__ dx,
                                                              int64_t n, int dbg, int few_runs, int second_probe)
{
    constexpr int kRounds = 1;                 // one 64-sample round per wave
    constexpr int kEnt = 1 << kEntLog2;        // table entries per cache slot (8 = 64-B line, 4 = 32-B sector)
    constexpr int kPay = 2 * kEnt;             // payload floats per slot = lanes per slot in the flush
    constexpr int kSlotBits = 31 - __builtin_clz((unsigned)kSlots);
    constexpr int kChunk = 64 * kRounds;
    __shared__ uint32_t s_key[4][kSlots];
    __shared__ float s_val[4][kSlots * kPay];
    __shared__ uint16_t s_list[4][kSlots];      // occupied slots
    __shared__ uint32_t s_dummy32[4][64];
    __shared__ uint64_t s_dummy64[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wave_base = ((int64_t)blockIdx.x * 4 + wave) * kChunk;
    if (wave_base >= n) return;
    lds_u32 *key = (lds_u32 *)&s_key[wave][0];
    lds_f32 *val = (lds_f32 *)&s_val[wave][0];
    lds_u16 *list = (lds_u16 *)&s_list[wave][0];
    lds_u32 *dummy32 = (lds_u32 *)&s_dummy32[wave][lane];
    lds_u64 *dummy64 = (lds_u64 *)&s_dummy64[wave][lane];
    *dummy32 = kNoLine;
    *dummy64 = 0;
    for (int s = lane; s < kSlots; s += 64) key[s] = kNoLine;
    for (int s = lane; s < kSlots * kPay; s += 64) val[s] = 0.f;

    float px[kRounds][3], dacc[kRounds][3];
    int64_t si[kRounds];
    bool valid[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        valid[r] = i < n;
        si[r] = valid[r] ? i : n - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            px[r][d] = x[si[r] * 3 + d];
            dacc[r][d] = 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the cache is private to this wave; DS ops of a wave execute in order

    for (int l = g.l_begin; l < g.l_end; ++l) {
        const LevelInfo li = level_info(g, l);
        float *__restrict__ dt = dtable + 2 * (size_t)li.offset;
        const float2 *__restrict__ tab = table + li.offset;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            float w0, w1, w2;
            uint32_t p0, p1, p2;
            pos_fract(px[r][0], li.scale, w0, p0);
            pos_fract(px[r][1], li.scale, w1, p1);
            pos_fract(px[r][2], li.scale, w2, p2);
            float2 gy = dy[(int64_t)l * n + si[r]];
            if (!valid[r]) gy = make_float2(0.f, 0.f);
            uint32_t idx[8];
            corner_indices(li, p0, p1, p2, idx);
            float2 tv[8];
            if (WITH_DX) {
#pragma unroll
                for (int c = 0; c < 8; ++c) tv[c] = tab[idx[c]];
            }
            // runs of lanes in the same cell
            const uint32_t q0 = __shfl_up(p0, 1), q1 = __shfl_up(p1, 1), q2 = __shfl_up(p2, 1);
            const int head = (lane == 0) || (q0 != p0) || (q1 != p1) || (q2 != p2);
            const int next_head = __shfl_down(head, 1);
            const bool run_end = (lane == 63) || next_head;
            float v[16];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wt = ((c & 1) ? w0 : 1.f - w0) * ((c & 2) ? w1 : 1.f - w1) * ((c & 4) ? w2 : 1.f - w2);
                v[2 * c] = wt * gy.x;
                v[2 * c + 1] = wt * gy.y;
            }
            if (!(dbg & 4)) {
                // segmented inclusive scan: inside the 16-lane rows with DPP, then a serial carry across the 3 row borders
                const int row_pos = lane & 15;
                int flag = head | (row_pos == 0);   // "my partial sum already starts at my run's head (or my row's start)"
                if (seg_scan_row_step<1>(v, flag, row_pos) && seg_scan_row_step<2>(v, flag, row_pos) &&
                    seg_scan_row_step<4>(v, flag, row_pos))
                    seg_scan_row_step<8>(v, flag, row_pos);
                // `open` = no run head between my row's first lane and me (inclusive): my run continues from the row before
                int open = !head;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const int o = (off == 1) ? row_shr_i32<1>(open) : (off == 2) ? row_shr_i32<2>(open)
                                : (off == 4) ? row_shr_i32<4>(open) : row_shr_i32<8>(open);
                    open &= (row_pos >= off) ? o : 1;
                }
                seg_scan_cross_rows(v, open, lane);
            }
            if (WITH_DX) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float sx = (c & 1) ? w0 : 1.f - w0, sy = (c & 2) ? w1 : 1.f - w1, sz = (c & 4) ? w2 : 1.f - w2;
                    const float t = li.scale * (gy.x * tv[c].x + gy.y * tv[c].y);
                    dacc[r][0] += t * ((c & 1) ? 1.f : -1.f) * (sy * sz);
                    dacc[r][1] += t * ((c & 2) ? 1.f : -1.f) * (sx * sz);
                    dacc[r][2] += t * ((c & 4) ? 1.f : -1.f) * (sx * sy);
                }
            }
            // ---- run ends add their 8 corner sums into the line cache, then the level's lines are flushed.
            // The kernel is instruction-issue bound (rocprof: VALU+SALU+LDS issue ~ 3 ms of a 3.6 ms launch), so this part
            // is written for few instructions: one probe per corner, no retry rounds (a lost slot goes straight to memory),
            // every DS operation issued by all lanes (idle lanes aim at a private dummy word) so that a batch of 8 goes out
            // back to back with one wait.
            const uint64_t ends_mask = __builtin_amdgcn_ballot_w64(run_end && !(dbg & 2));
            const int n_ends = __builtin_popcountll(ends_mask);
            if (n_ends <= few_runs) {
                // Coarse levels: a wave ends only a handful of runs, and almost all of the ~500 instructions of the cache
                // path below would be spent on idle lanes.  Instead the run ends park their 8 indices + 16 sums in LDS
                // (the zeroed payload area doubles as staging) and the wave re-reads them with 16 lanes per run -- lane =
                // (corner, feature), 4 runs per instruction -- and adds straight to memory: the two features and the
                // x / x+1 neighbours of an entry share requests inside the instruction.
                if (n_ends > 0) {
                    lds_u32 *stage = (lds_u32 *)val;           // [run][8 idx | 16 values]
                    if (run_end && !(dbg & 2)) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends_mask, 0));
#pragma unroll
                        for (int c = 0; c < 8; ++c) stage[rank * 24 + c] = idx[c];
#pragma unroll
                        for (int k = 0; k < 16; ++k) stage[rank * 24 + 8 + k] = __float_as_uint(v[k]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int k16 = lane & 15;
                    for (int r0 = lane >> 4; r0 < n_ends; r0 += 4) {
                        const uint32_t e = stage[r0 * 24 + (k16 >> 1)];
                        const float a = __uint_as_float(stage[r0 * 24 + 8 + k16]);
                        if (a != 0.f) atomicAdd(dt + 2 * (size_t)e + (k16 & 1), a);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int s_ = lane; s_ < n_ends * 24; s_ += 64) stage[s_] = 0u;   // the payload area must read zero again
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                const bool act = run_end && !(dbg & 2);
                uint32_t used = 0;   // occupied cache slots (wave-uniform)
                if (__builtin_amdgcn_ballot_w64(act) != 0) {
                    // multiplicative hash of the line id (a slot function linear in the cell coordinates was tried: more
                    // collisions on ray-shaped line sets, tools/sim_hash_bwd_requests.py)
                    uint32_t slot[8], old[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) slot[c] = (__umul24(idx[c] >> kEntLog2, 0x9E3779u) >> (24 - kSlotBits)) & (kSlots - 1);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        old[c] = lds_cas(act ? &key[slot[c]] : dummy32, kNoLine, act ? (idx[c] >> kEntLog2) : kNoLine);
                    // float LDS atomics run at ~3 cycles per LANE on gfx950 (tools/micro/lds_ops.hip: ds_add_f32 194 cycles
                    // per instruction, ds_cmpst_b64 22): add both features with one 64-bit compare-and-swap
                    lds_u64 *va[8];
                    bool to_mem[8], okc[8];
                    uint64_t cur[8], prev[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const bool claim = act && old[c] == kNoLine;
                        const uint64_t cm = __builtin_amdgcn_ballot_w64(claim);
                        if (cm) {
                            if (claim)
                                list[used + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0))] =
                                    (uint16_t)slot[c];
                            used += __builtin_popcountll(cm);
                        }
                        okc[c] = claim || (act && old[c] == (idx[c] >> kEntLog2));
                        to_mem[c] = act && !okc[c];
                    }
                    if (second_probe) {
                        // second chance in the neighbouring slot for the corners that lost their first one (batched the same way)
                        bool any = false;
#pragma unroll
                        for (int c = 0; c < 8; ++c) any = any || to_mem[c];
                        if (__builtin_amdgcn_ballot_w64(any) != 0) {
                            uint32_t old2[8];
#pragma unroll
                            for (int c = 0; c < 8; ++c)
                                old2[c] = lds_cas(to_mem[c] ? &key[(slot[c] + 1) & (kSlots - 1)] : dummy32, kNoLine,
                                                  to_mem[c] ? (idx[c] >> kEntLog2) : kNoLine);
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const bool claim = to_mem[c] && old2[c] == kNoLine;
                                const uint64_t cm = __builtin_amdgcn_ballot_w64(claim);
                                if (cm) {
                                    if (claim)
                                        list[used + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0))] =
                                            (uint16_t)((slot[c] + 1) & (kSlots - 1));
                                    used += __builtin_popcountll(cm);
                                }
                                if (claim || (to_mem[c] && old2[c] == (idx[c] >> kEntLog2))) {
                                    slot[c] = (slot[c] + 1) & (kSlots - 1);
                                    okc[c] = true;
                                    to_mem[c] = false;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        va[c] = okc[c] ? (lds_u64 *)&val[slot[c] * kPay + (idx[c] & (kEnt - 1)) * 2] : dummy64;
#pragma unroll
                    for (int c = 0; c < 8; ++c) cur[c] = *va[c];
#pragma unroll
                    for (int c = 0; c < 8; ++c) prev[c] = lds_cas(va[c], cur[c], add_pair(cur[c], v[2 * c], v[2 * c + 1]));
                    bool retry = false;
#pragma unroll
                    for (int c = 0; c < 8; ++c) retry = retry || prev[c] != cur[c];
                    if (__builtin_amdgcn_ballot_w64(retry) != 0) {   // rare: two lanes (or two corners of a lane) on one entry
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            while (prev[c] != cur[c]) {
                                cur[c] = prev[c];
                                prev[c] = lds_cas(va[c], cur[c], add_pair(cur[c], v[2 * c], v[2 * c + 1]));
                            }
                    }
                    // Corners whose slot is owned by another sector go straight to memory.  The two features of an entry are
                    // written by a PAIR of lanes in one instruction (lane and lane^1 swap operands through DPP), so the
                    // 8 bytes cost one request to the atomic units instead of two.
                    const bool is_odd = lane & 1;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (__builtin_amdgcn_ballot_w64(to_mem[c]) == 0) continue;
                        const uint32_t idx_n = (uint32_t)quad_swap1((int)idx[c]);
                        const float v0_n = __int_as_float(quad_swap1(__float_as_int(v[2 * c])));
                        const float v1_n = __int_as_float(quad_swap1(__float_as_int(v[2 * c + 1])));
                        const bool tm_n = quad_swap1((int)to_mem[c]) != 0;
                        // first instruction serves the even lanes' corners, second one the odd lanes'
                        if (is_odd ? tm_n : to_mem[c]) atomicAdd(dt + 2 * (size_t)(is_odd ? idx_n : idx[c]) + is_odd, is_odd ? v1_n : v[2 * c]);
                        if (is_odd ? to_mem[c] : tm_n) atomicAdd(dt + 2 * (size_t)(is_odd ? idx[c] : idx_n) + is_odd, is_odd ? v[2 * c + 1] : v0_n);
                    }
                }
                // flush: 16 lanes per line, 4 lines per instruction, 16 lines per trip (loads first)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                constexpr int kPer = 64 / kPay;   // slots per flush instruction
                const int sub = lane & (kPay - 1);
                for (uint32_t e0 = lane / kPay; e0 < used; e0 += 4 * kPer) {
                    uint32_t ent[4], ln[4];
                    float vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ent[u] = (e0 + kPer * u < used) ? (uint32_t)list[e0 + kPer * u] : 0xFFFFFFFFu;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ln[u] = (ent[u] != 0xFFFFFFFFu) ? key[ent[u]] : 0u;
                        vv[u] = (ent[u] != 0xFFFFFFFFu) ? val[ent[u] * kPay + sub] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (vv[u] != 0.f && !(dbg & 1)) atomicAdd(dt + (size_t)ln[u] * kPay + sub, vv[u]);
                    __builtin_amdgcn_wave_barrier();   // every lane of a slot has read the key before lane 0 resets it
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ent[u] != 0xFFFFFFFFu) {
                            val[ent[u] * kPay + sub] = 0.f;
                            if (sub == 0) key[ent[u]] = kNoLine;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (WITH_DX) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            if (valid[r]) {
                if (g.dx_accumulate) {
                    dacc[r][0] += dx[si[r] * 3 + 0];
                    dacc[r][1] += dx[si[r] * 3 + 1];
                    dacc[r][2] += dx[si[r] * 3 + 2];
                }
                dx[si[r] * 3 + 0] = dacc[r][0];
                dx[si[r] * 3 + 1] = dacc[r][1];
                dx[si[r] * 3 + 2] = dacc[r][2];
            }
    }
}
#endif  // LSE_DEV_KNOBS

// kPair: the memory-side units charge a float-atomic REQUEST per 64-byte LINE, whatever part of it the request covers
// (tools/micro/atomic_gran.hip: 21 G requests/s for 4 ... 64 contiguous bytes, half of that for 128).  With 32-byte slots a line
// whose two sectors were both touched is flushed as two requests.  kPair hashes the LINE to a pair of neighbouring slots (even /
// odd sector) and builds the flush list by scanning the keys in slot order, so sibling sectors sit in neighbouring 8-lane groups
// of one flush instruction and go out as ONE contiguous 64-byte request; the scan also replaces the per-corner list appends of
// the insert path (fewer instructions per pass).
// kFlush2 (round 3): the kernel is bound by the CU's LDS pipe -- every DS wave-instruction costs 7 .. 8 cycles of it
// (13 / 22 for the 32- / 64-bit compare-and-swap; profiles/r02_micro_lds_ops.txt), a cache pass issued ~190 of them, 140 of
// those in the flush (per 32 slots: 4 list reads, 4 key reads, 4 payload reads, 4 payload resets, 4 key resets), and
// 256 waves x ~9 passes x 1500 cycles per CU is the 1.5 ms the ablation attributes to the cache passes.  Flush with fewer DS
// instructions: the list is stored trip-major transposed, so the 4 slots a lane group handles in a trip are one 8-byte read;
// the payload is read AND zeroed by one ds_wrxchg_rtn_b32; the keys are reset in bulk after the last trip (2 x ds_write_b128
// per lane).  Per 32 slots: 1 + 4 + 4 = 9 instead of 20 DS instructions, per pass ~75 instead of ~140 in the flush.
// kPrefetch (round 3): vmcnt retires in order, and a no-return atomic stays counted for ~3000 cycles when every CU is issuing them
// (MI355X_MICROARCH.md).  A level's dy load and table gathers, issued at the top of the level right behind the previous level's
// flush atomics, can therefore not be consumed before those atomics have retired: every level begins with that wait.  With
// kPrefetch the loads of level l+1 are issued right after the scan of level l -- BEFORE level l's cache pass -- and are waited for
// just before the pass sends its first atomic (an empty asm that consumes the registers places the s_waitcnt there): by then they
// have had the whole insert phase to arrive, and the atomics get the whole next level to retire.
// kAlign (round 3): the two sectors of a line leave as ONE request only when their list entries sit in neighbouring lane groups of the
// SAME flush instruction; in slot order a sibling pair straddles an 8-entry boundary one time in eight.  With kAlign every occupied
// pair bucket takes an even-aligned pair of list positions (an absent sibling is an idle entry), so a line is never split.
// kWaves: waves per workgroup.  Nothing below synchronises across waves (every wave owns its cache), so the workgroup size only sets
// the granularity at which LDS is handed out: 4 x 512 slots = 80 KB -> 8 waves per CU; 2 x 384 slots = 30.5 KB -> 10 waves per CU.
template <bool WITH_DX, int kSlots, int kEntLog2, bool kPair = false, bool kFlush2 = false, bool kPrefetch = false, bool kAlign = false,
          int kWaves = 4>
__global__ __launch_bounds__(64 * kWaves) void hash_bwd_batched_kernel(GridParams g, const float *__restrict__ x,
                                                              const float2 *__restrict__ dy,
                                                              const float2 *__restrict__ table,
                                                              float *__restrict__ dtable, float *__restrict__ dx,
                                                              int64_t n_cap, int dbg, int few_runs, int second_probe, int stage_max,
                                                              float *__restrict__ ws, const int64_t *__restrict__ n_dev)
{
    const int64_t n = lse::clamp_count(n_cap, n_dev);      // n_cap stays the level stride of dy
    constexpr int kRounds = 1;                 // one 64-sample round per wave
    constexpr int kEnt = 1 << kEntLog2;        // table entries per cache slot (8 = 64-B line, 4 = 32-B sector)
    constexpr int kPay = 2 * kEnt;             // payload floats per slot = lanes per slot in the flush
    constexpr int kSlotBits = 31 - __builtin_clz((unsigned)kSlots);
    constexpr bool kPow2 = (kSlots & (kSlots - 1)) == 0;     // 320 slots (three workgroups per CU): multiply-high instead of masks
    static_assert(kPow2 || (kPair && kFlush2 && kSlots % 64 == 0), "non-power-of-two caches: paired slots, second-generation flush");
    constexpr int kChunk = 64 * kRounds;
    constexpr int kStep = kPair ? 2 : 1;        // second probe: the next slot of the same parity
    auto wrap = [](uint32_t sl) -> uint32_t {   // sl < 2 * kSlots
        if constexpr (kPow2) return sl & (uint32_t)(kSlots - 1);
        else return sl >= (uint32_t)kSlots ? sl - (uint32_t)kSlots : sl;
    };
    __shared__ uint32_t s_key[kWaves][kSlots];
    __shared__ float s_val[kWaves][kSlots * kPay];
    __shared__ uint16_t s_list[kWaves][kSlots];      // occupied slots
    // LDS is handed out in 1280-byte portions: 448 slots fit nine times into a CU only when the idle lanes' 32-bit dummy word is the
    // low half of their 64-bit one (either holds don't-care values between uses: every use overwrites or ignores it)
    constexpr bool kTightLds = kSlots == 448;
    __shared__ uint32_t s_dummy32[kTightLds ? 1 : kWaves][64];
    __shared__ uint64_t s_dummy64[kWaves][64];
    __shared__ uint32_t s_perm[kWaves][64];          // rank -> lane of the run ends being staged
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Which 64-sample chunks a workgroup takes.  The hardware deals workgroups round-robin to the 8 XCDs, so with the identity
    // mapping XCD x gets the chunks c = x (mod 8) -- and in ray-ordered samples the cost of a chunk depends on where on its ray it
    // lies (dense cells near the camera, long runs in the contracted shell): with 1024 samples per ray XCD x saw the chunks x and
    // x + 8 of EVERY ray, a static load imbalance between XCDs.  Workgroup b therefore takes position (b % 8) * ceil(B / 8) + b / 8
    // of the B positions the samples actually fill (device-side count: a captured step's capacity is larger): every XCD walks one
    // contiguous eighth, i.e. whole rays whatever their length.  Inside-box rays 2.39 -> 2.19 ms, the reference's default
    // configuration 1.51 -> 1.47, headline +-0.3 % (profiles/r05_hash_bwd_xcd_mapping.txt).  (dbg & 32, development build: the
    // identity mapping, for A/B.)
    int64_t wg = blockIdx.x;
    if (!(dbg & 32)) {
        const int64_t positions = (n + (int64_t)kChunk * kWaves - 1) / ((int64_t)kChunk * kWaves);
        const int64_t per_xcd = (positions + 7) >> 3;
        if ((wg >> 3) >= per_xcd) return;      // (the grid covers the capacity, rounded up to a multiple of 8)
        wg = (wg & 7) * per_xcd + (wg >> 3);
    }
    const int64_t wave_base = (wg * kWaves + wave) * kChunk;
    if (wave_base >= n) return;
    lds_u32 *key = (lds_u32 *)&s_key[wave][0];
    lds_f32 *val = (lds_f32 *)&s_val[wave][0];
    lds_u16 *list = (lds_u16 *)&s_list[wave][0];
    lds_u32 *dummy32 = kTightLds ? (lds_u32 *)&s_dummy64[wave][lane] : (lds_u32 *)&s_dummy32[wave][lane];
    lds_u64 *dummy64 = (lds_u64 *)&s_dummy64[wave][lane];
    lds_u32 *perm = (lds_u32 *)&s_perm[wave][0];
    *dummy32 = kNoLine;
    *dummy64 = 0;
    for (int s = lane; s < kSlots; s += 64) key[s] = kNoLine;
    for (int s = lane; s < kSlots * kPay; s += 64) val[s] = 0.f;

    float px[kRounds][3], dacc[kRounds][3];
    int64_t si[kRounds];
    bool valid[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        valid[r] = i < n;
        si[r] = valid[r] ? i : n - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            px[r][d] = x[si[r] * 3 + d];
            dacc[r][d] = 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the cache is private to this wave; DS ops of a wave execute in order

    // ---- register queue of run ends waiting for a cache pass: lane t < fill holds one run end (8 GLOBAL entry indices =
    // level offset + index inside the level, and its 8 x 2 corner sums).  Cache keys are global sector ids, so run ends of
    // several levels share one pass: the insert + flush below costs ~0.1 ms per pass at the metric size however few lanes
    // take part (measured: 1.5 of the kernel's 2.9 ms), and the levels 4 .. 9 end only 5 .. 15 runs per wave each.
    uint32_t q_idx[8];
    float q_v[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) q_idx[c] = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) q_v[k] = 0.f;
    int fill = 0;

    // (kPrefetch) level l+1's operands, fetched while level l's cache pass runs
    float nw0 = 0.f, nw1 = 0.f, nw2 = 0.f;
    uint32_t np0 = 0, np1 = 0, np2 = 0, nidx[8];
    float2 ngy = make_float2(0.f, 0.f), ntv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        nidx[c] = 0u;
        ntv[c] = make_float2(0.f, 0.f);
    }
    bool pending_touch = false;     // wave-uniform: loads of the next level are in flight and not yet waited for
    auto fetch_next = [&](int l) {
        const LevelInfo nli = level_info(g, l);
        pos_fract(px[0][0], nli.scale, nw0, np0);
        pos_fract(px[0][1], nli.scale, nw1, np1);
        pos_fract(px[0][2], nli.scale, nw2, np2);
        ngy = dy[(int64_t)l * n_cap + si[0]];
        corner_indices(nli, np0, np1, np2, nidx);
        if (WITH_DX) {
            const float2 *__restrict__ ntab = table + nli.offset;
#pragma unroll
            for (int c = 0; c < 8; ++c) ntv[c] = ntab[nidx[c]];
        }
        pending_touch = true;
    };
    auto touch_next = [&]() {
        if (!kPrefetch || !pending_touch) return;
        asm volatile("" ::"v"(ngy.x), "v"(ngy.y));
        if (WITH_DX) {
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(ntv[c].x), "v"(ntv[c].y));
        }
        pending_touch = false;
    };

    auto insert_pass = [&](const bool act, const uint32_t (&gi)[8], const float (&vv)[16]) {
        uint32_t used = 0;   // occupied cache slots (wave-uniform)
            if (__builtin_amdgcn_ballot_w64(act) != 0) {
                // multiplicative hash of the line id (a slot function linear in the cell coordinates was tried: more
                // collisions on ray-shaped line sets, tools/sim_hash_bwd_requests.py)
                uint32_t slot[8], old[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if constexpr (kPair && !kPow2)      // pair bucket = floor(hash / 2^24 * (kSlots / 2)): the top bits of the product
                        slot[c] = (((__umul24(gi[c] >> (kEntLog2 + 1), 0x9E3779u) & 0xFFFFFFu) * (uint32_t)(kSlots / 2)) >> 24 << 1) |
                                  ((gi[c] >> kEntLog2) & 1u);
                    else if constexpr (kPair)
                        slot[c] = (((__umul24(gi[c] >> (kEntLog2 + 1), 0x9E3779u) >> (25 - kSlotBits)) & (kSlots / 2 - 1)) << 1) |
                                  ((gi[c] >> kEntLog2) & 1u);
                    else
                        slot[c] = (__umul24(gi[c] >> kEntLog2, 0x9E3779u) >> (24 - kSlotBits)) & (kSlots - 1);
                }
                uint32_t keyc[8];      // the sector a corner belongs to = the cache key
#pragma unroll
                for (int c = 0; c < 8; ++c) keyc[c] = gi[c] >> kEntLog2;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    old[c] = lds_cas(act ? &key[slot[c]] : dummy32, kNoLine, act ? keyc[c] : kNoLine);
                // float LDS atomics run at ~3 cycles per LANE on gfx950 (tools/micro/lds_ops.hip: ds_add_f32 194 cycles
                // per instruction, ds_cmpst_b64 22): add both features with one 64-bit compare-and-swap
                lds_u64 *va[8];
                // ONE piece of state per corner: to_mem = "active and not (yet) at home in a slot".  (Until round 5 a second array --
                // okc -- and exec-masked update blocks carried the same information: ~300 instructions per probe round, half of
                // them scalar mask bookkeeping; selects on one mask per corner need ~90.)
                bool to_mem[8];
                uint64_t cur[8], prev[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool home = (old[c] == kNoLine) || (old[c] == keyc[c]);      // claimed the slot, or found its own sector there
                    if constexpr (!kPair) {
                        const bool claim = act && old[c] == kNoLine;
                        const uint64_t cm = __builtin_amdgcn_ballot_w64(claim);
                        if (cm) {
                            if (claim)
                                list[used + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0))] =
                                    (uint16_t)slot[c];
                            used += __builtin_popcountll(cm);
                        }
                    }
                    to_mem[c] = act && !home;
                }
                for (int pr = 0; pr < second_probe; ++pr) {
                    // another chance in the next slot for the corners that lost the previous one (batched the same way); a loser keeps
                    // its home slot, so round pr probes home + (pr + 1) * kStep (until round 3 every round after the first probed the
                    // same neighbour again and could not win anything)
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < 8; ++c) any = any || to_mem[c];
                    // rounds after the first only when at least 8 lanes still hold a loser: a round costs the whole wave its
                    // instructions whatever the number of losers, and where the kernel is bound by its own instructions (samples in
                    // the contracted shell: 2.57 -> 2.71 ms with unconditional rounds, 2.66 with the threshold) a handful of direct adds
                    // is cheaper; the atomic-bound regimes keep their gain (default configuration 1.51 -> 1.48 ms either way)
                    const int min_losers = pr == 0 ? 1 : 8;
                    if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(any)) >= min_losers) {
                        uint32_t old2[8], nxt[8];
                        const uint32_t hop = (uint32_t)(pr + 1) * kStep;
#pragma unroll
                        for (int c = 0; c < 8; ++c) nxt[c] = wrap(slot[c] + hop);
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            old2[c] = lds_cas(to_mem[c] ? &key[nxt[c]] : dummy32, kNoLine, to_mem[c] ? keyc[c] : kNoLine);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const bool won = to_mem[c] && ((old2[c] == kNoLine) || (old2[c] == keyc[c]));
                            if constexpr (!kPair) {
                                const bool claim = to_mem[c] && old2[c] == kNoLine;
                                const uint64_t cm = __builtin_amdgcn_ballot_w64(claim);
                                if (cm) {
                                    if (claim)
                                        list[used + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0))] =
                                            (uint16_t)nxt[c];
                                    used += __builtin_popcountll(cm);
                                }
                            }
                            slot[c] = won ? nxt[c] : slot[c];
                            to_mem[c] = to_mem[c] && !won;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // address for every lane, then ONE select (left to itself the compiler guards each address with its own
                    // exec-masked block: 29 s_and_saveexec + 17 branches for the eight corners)
                    uint32_t home_p = (uint32_t)(uintptr_t)(lds_u64 *)&val[slot[c] * kPay + (gi[c] & (kEnt - 1)) * 2];
                    asm volatile("" : "+v"(home_p));
                    va[c] = (act && !to_mem[c]) ? (lds_u64 *)(uintptr_t)home_p : dummy64;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) cur[c] = *va[c];
#pragma unroll
                for (int c = 0; c < 8; ++c) prev[c] = lds_cas(va[c], cur[c], add_pair(cur[c], vv[2 * c], vv[2 * c + 1]));
                bool retry = false;
#pragma unroll
                for (int c = 0; c < 8; ++c) retry = retry || prev[c] != cur[c];
                if (__builtin_amdgcn_ballot_w64(retry) != 0) {   // rare: two lanes (or two corners of a lane) on one entry
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        while (prev[c] != cur[c]) {
                            cur[c] = prev[c];
                            prev[c] = lds_cas(va[c], cur[c], add_pair(cur[c], vv[2 * c], vv[2 * c + 1]));
                        }
                }
                // Corners whose slot is owned by another sector go straight to memory.  The two features of an entry are
                // written by a PAIR of lanes in one instruction (lane and lane^1 swap operands through DPP), so the
                // 8 bytes cost one request to the atomic units instead of two.
                const bool is_odd = lane & 1;
                touch_next();      // (kPrefetch) the next level's loads are consumed before this pass's first atomic
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (__builtin_amdgcn_ballot_w64(to_mem[c]) == 0) continue;
                    const uint32_t gi_n = (uint32_t)quad_swap1((int)gi[c]);
                    const float v0_n = __int_as_float(quad_swap1(__float_as_int(vv[2 * c])));
                    const float v1_n = __int_as_float(quad_swap1(__float_as_int(vv[2 * c + 1])));
                    const bool tm_n = quad_swap1((int)to_mem[c]) != 0;
                    // first instruction serves the even lanes' corners, second one the odd lanes'
                    if (is_odd ? tm_n : to_mem[c]) atomicAdd(dtable + 2 * (size_t)(is_odd ? gi_n : gi[c]) + is_odd, is_odd ? v1_n : vv[2 * c]);
                    if (is_odd ? to_mem[c] : tm_n) atomicAdd(dtable + 2 * (size_t)(is_odd ? gi[c] : gi_n) + is_odd, is_odd ? vv[2 * c + 1] : v0_n);
                }
            }
            // flush: 16 lanes per line, 4 lines per instruction, 16 lines per trip (loads first)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (kPair) {      // occupied slots in slot order: sibling sectors of a line become list neighbours
#pragma unroll
                for (int s0 = 0; s0 < kSlots; s0 += 64) {
                    const bool occ = key[s0 + lane] != kNoLine;
                    // kAlign: both slots of an occupied pair bucket (lanes 2b, 2b + 1) take a list position
                    const bool take = kAlign ? (occ || quad_swap1((int)occ) != 0) : occ;
                    const uint64_t om = __builtin_amdgcn_ballot_w64(take);
                    if (take) {
                        uint32_t pos = used + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0));
                        // trip-major transposed: list position i of trip i / 32 is handled by lane group i % 8 as its (i % 32) / 8-th slot
                        if constexpr (kFlush2) pos = (pos & ~31u) + ((pos & 7u) << 2) + ((pos >> 3) & 3u);
                        list[pos] = occ ? (uint16_t)(s0 + lane) : (uint16_t)0xFFFFu;      // 0xFFFF: the absent sibling of a pair
                    }
                    used += __builtin_popcountll(om);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            constexpr int kPer = 64 / kPay;   // slots per flush instruction
            const int sub = lane & (kPay - 1);
            if constexpr (kFlush2) {
                static_assert(!kFlush2 || (kPair && kPay == 8 && kSlots >= 256 && kSlots <= 512 && kSlots % 64 == 0),
                              "flush v2: 256 .. 512 paired 32-byte slots (bulk key reset: 8 keys per lane)");
                const uint32_t gq = (uint32_t)lane >> 3;
                touch_next();
                for (uint32_t t0 = 0; t0 < used; t0 += 32) {
                    const uint64_t four = *(lds_u64 *)&list[t0 + 4 * gq];
                    uint32_t sl[4], ln[4];
                    float vv[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t e = (uint32_t)(four >> (16 * u)) & 0xFFFFu;
                        ok[u] = t0 + 8 * u + gq < used && (!kAlign || e != 0xFFFFu);
                        sl[u] = ok[u] ? e : (uint32_t)lane;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) ln[u] = key[sl[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u)       // read + zero in one DS instruction; idle lanes swap their private dummy word
                        vv[u] = __hip_atomic_exchange(ok[u] ? &val[sl[u] * kPay + sub] : (lds_f32 *)dummy32, 0.f, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u] && vv[u] != 0.f && !(dbg & 1)) atomicAdd(dtable + (size_t)ln[u] * kPay + sub, vv[u]);
                }
                if (used) {      // every occupied key was in the list: reset all of them, 8 keys per lane
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
                    lds_u32x4 *k4 = (lds_u32x4 *)key;
                    const u32x4_t none = {kNoLine, kNoLine, kNoLine, kNoLine};
                    k4[lane] = none;
                    if (kSlots == 512 || lane < (kSlots - 256) / 4) k4[64 + lane] = none;
                    *dummy32 = kNoLine;      // (the exchange above left 0.f in the idle lanes' dummy word)
                }
            } else
            for (uint32_t e0 = lane / kPay; e0 < used; e0 += 4 * kPer) {
                uint32_t ent[4], ln[4];
                float vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ent[u] = (e0 + kPer * u < used) ? (uint32_t)list[e0 + kPer * u] : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ln[u] = (ent[u] != 0xFFFFFFFFu) ? key[ent[u]] : 0u;
                    vv[u] = (ent[u] != 0xFFFFFFFFu) ? val[ent[u] * kPay + sub] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (vv[u] != 0.f && !(dbg & 1)) atomicAdd(dtable + (size_t)ln[u] * kPay + sub, vv[u]);
                __builtin_amdgcn_wave_barrier();   // every lane of a slot has read the key before lane 0 resets it
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ent[u] != 0xFFFFFFFFu) {
                        val[ent[u] * kPay + sub] = 0.f;
                        if (sub == 0) key[ent[u]] = kNoLine;
                    }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    if (kPrefetch && g.l_begin < g.l_end) fetch_next(g.l_begin);
    for (int l = g.l_begin; l < g.l_end; ++l) {
        const LevelInfo li = level_info(g, l);
        // (few-runs path only)  The coarsest levels are a few thousand lines that every wave of the launch adds to: the
        // memory-side units serialise same-line requests (M-packed: 5 % of the kernel's requests, 0.6 of its 3.6 ms), so
        // those adds go to one of several replicas of the level, picked by the index of the wave's group of four, and
        // hash_bwd_reduce_replicas_kernel folds the replicas into the table gradient afterwards
        float *__restrict__ dt = (ws != nullptr && l < g.rep_levels)
                                     ? ws + (size_t)((blockIdx.x * kWaves / 4) & (unsigned)g.rep_mask) * g.rep_stride + 2 * (size_t)li.offset
                                     : dtable + 2 * (size_t)li.offset;
        const float2 *__restrict__ tab = table + li.offset;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            float w0, w1, w2;
            uint32_t p0, p1, p2;
            float2 gy;
            uint32_t idx[8];
            float2 tv[8];
            if constexpr (kPrefetch) {      // fetched during the previous level's cache pass (or by the prologue)
                w0 = nw0; w1 = nw1; w2 = nw2;
                p0 = np0; p1 = np1; p2 = np2;
                gy = ngy;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    idx[c] = nidx[c];
                    tv[c] = ntv[c];
                }
                pending_touch = false;
            } else {
                pos_fract(px[r][0], li.scale, w0, p0);
                pos_fract(px[r][1], li.scale, w1, p1);
                pos_fract(px[r][2], li.scale, w2, p2);
                gy = dy[(int64_t)l * n_cap + si[r]];
                corner_indices(li, p0, p1, p2, idx);
                if (WITH_DX) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) tv[c] = tab[idx[c]];
                }
            }
            if (!valid[r]) gy = make_float2(0.f, 0.f);
            // runs of lanes in the same cell
            const uint32_t q0 = wave_shr1(p0), q1 = wave_shr1(p1), q2 = wave_shr1(p2);
            const int head = (lane == 0) || (q0 != p0) || (q1 != p1) || (q2 != p2);
            const int next_head = wave_shl1(head);
            const bool run_end = (lane == 63) || next_head;
            float v[16];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wt = ((c & 1) ? w0 : 1.f - w0) * ((c & 2) ? w1 : 1.f - w1) * ((c & 4) ? w2 : 1.f - w2);
                v[2 * c] = wt * gy.x;
                v[2 * c + 1] = wt * gy.y;
            }
            if (!(dbg & 4)) {
                // segmented inclusive scan: inside the 16-lane rows with DPP, then a serial carry across the 3 row borders
                const int row_pos = lane & 15;
                int flag = head | (row_pos == 0);   // "my partial sum already starts at my run's head (or my row's start)"
                if (seg_scan_row_step<1>(v, flag, row_pos) && seg_scan_row_step<2>(v, flag, row_pos) &&
                    seg_scan_row_step<4>(v, flag, row_pos))
                    seg_scan_row_step<8>(v, flag, row_pos);
                // `open` = no run head between my row's first lane and me (inclusive): my run continues from the row before
                int open = !head;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const int o = (off == 1) ? row_shr_i32<1>(open) : (off == 2) ? row_shr_i32<2>(open)
                                : (off == 4) ? row_shr_i32<4>(open) : row_shr_i32<8>(open);
                    open &= (row_pos >= off) ? o : 1;
                }
                seg_scan_cross_rows(v, open, lane);
            }
            if (WITH_DX) {
                // d(x): t_c = <dy, table[c]>; along each axis the difference of the two faces, weighted by the other two axes
                const float a0 = 1.f - w0, a1 = 1.f - w1, a2 = 1.f - w2;
                float t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) t[c] = gy.x * tv[c].x + gy.y * tv[c].y;
                const float d0 = a2 * (a1 * (t[1] - t[0]) + w1 * (t[3] - t[2])) + w2 * (a1 * (t[5] - t[4]) + w1 * (t[7] - t[6]));
                const float d1 = a2 * (a0 * (t[2] - t[0]) + w0 * (t[3] - t[1])) + w2 * (a0 * (t[6] - t[4]) + w0 * (t[7] - t[5]));
                const float d2 = a1 * (a0 * (t[4] - t[0]) + w0 * (t[5] - t[1])) + w1 * (a0 * (t[6] - t[2]) + w0 * (t[7] - t[3]));
                dacc[r][0] = fmaf(li.scale, d0, dacc[r][0]);
                dacc[r][1] = fmaf(li.scale, d1, dacc[r][1]);
                dacc[r][2] = fmaf(li.scale, d2, dacc[r][2]);
            }
            if constexpr (kPrefetch) {
                if (l + 1 < g.l_end) fetch_next(l + 1);      // the next level's loads go out before this level's cache pass
            }
            // ---- run ends add their 8 corner sums into the line cache, then the level's lines are flushed.
            // The kernel is instruction-issue bound (rocprof: VALU+SALU+LDS issue ~ 3 ms of a 3.6 ms launch), so this part
            // is written for few instructions: one probe per corner, no retry rounds (a lost slot goes straight to memory),
            // every DS operation issued by all lanes (idle lanes aim at a private dummy word) so that a batch of 8 goes out
            // back to back with one wait.
            const uint64_t ends_mask = __builtin_amdgcn_ballot_w64(run_end && !(dbg & 2));
            const int n_ends = __builtin_popcountll(ends_mask);
            if (n_ends <= few_runs) {
                // Coarse levels: a wave ends only a handful of runs, and almost all of the ~500 instructions of the cache
                // path below would be spent on idle lanes.  Instead the run ends park their 8 indices + 16 sums in LDS
                // (the zeroed payload area doubles as staging) and the wave re-reads them with 16 lanes per run -- lane =
                // (corner, feature), 4 runs per instruction -- and adds straight to memory: the two features and the
                // x / x+1 neighbours of an entry share requests inside the instruction.
                if (n_ends > 0) {
                    lds_u32 *stage = (lds_u32 *)val;           // [run][8 idx | 16 values]
                    if (run_end && !(dbg & 2)) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends_mask, 0));
#pragma unroll
                        for (int c = 0; c < 8; ++c) stage[rank * 24 + c] = idx[c];
#pragma unroll
                        for (int k = 0; k < 16; ++k) stage[rank * 24 + 8 + k] = __float_as_uint(v[k]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int k16 = lane & 15;
                    touch_next();
                    for (int r0 = lane >> 4; r0 < n_ends; r0 += 4) {
                        const uint32_t e = stage[r0 * 24 + (k16 >> 1)];
                        const float a = __uint_as_float(stage[r0 * 24 + 8 + k16]);
                        if (a != 0.f && !(dbg & 16)) atomicAdd(dt + 2 * (size_t)e + (k16 & 1), a);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int s_ = lane; s_ < n_ends * 24; s_ += 64) stage[s_] = 0u;   // the payload area must read zero again
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                uint32_t gidx[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) gidx[c] = li.offset + idx[c];
                if (fill + n_ends > 64) {               // the queue cannot take this level: run the pending pass first
                    insert_pass(lane < fill, q_idx, q_v);
                    fill = 0;
                }
                if (fill == 0 && n_ends > stage_max) {
                    // fine level: most lanes end a run -- pass straight from the lanes that hold them (no compaction)
                    insert_pass(run_end && !(dbg & 2), gidx, v);
                } else {
                    // append: queue lane fill + r takes the r-th run end of this level (rank -> lane through LDS, then 24
                    // ds_bpermute moves); the pass runs when the queue is full, at a fine level, or after the last level
                    if (run_end && !(dbg & 2))
                        perm[__builtin_amdgcn_mbcnt_hi((uint32_t)(ends_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends_mask, 0))] = (uint32_t)lane;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int r = lane - fill;
                    const bool take = r >= 0 && r < n_ends;
                    const int src = take ? (int)perm[r] : lane;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)gidx[c]);
                        q_idx[c] = take ? t : q_idx[c];
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v[k2])));
                        q_v[k2] = take ? t : q_v[k2];
                    }
                    fill += n_ends;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (fill > 0) insert_pass(lane < fill, q_idx, q_v);
    if (WITH_DX) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            if (valid[r]) {
                if (g.dx_accumulate) {
                    dacc[r][0] += dx[si[r] * 3 + 0];
                    dacc[r][1] += dx[si[r] * 3 + 1];
                    dacc[r][2] += dx[si[r] * 3 + 2];
                }
                dx[si[r] * 3 + 0] = dacc[r][0];
                dx[si[r] * 3 + 1] = dacc[r][1];
                dx[si[r] * 3 + 2] = dacc[r][2];
            }
    }
}


// Coarse levels (round 3).  Inside the fused launch above, every level below ~9 costs 0.09 ms although a wave ends only 1 .. 10
// runs there and sends 0.1 .. 0.4 lines per sample to memory (gpurun_out/r3a): those levels pay for the fine levels'
// machinery -- the 80 KB of LDS per workgroup that hold the sector cache cap the kernel at two waves per SIMD, and at that
// occupancy nothing hides the latency of the eight table gathers, the dy load and the scan's DPP chain of a level.  This
// kernel has no cache: lane = sample, the same segmented scan leaves each run's 8 x 2 corner sums in the run's last lane, the
// run ends are transposed through a 1.5 KB per-wave staging area (16 runs per trip) and added to memory with 16 lanes per run
// (lane = corner x feature: the two features and the x / x+1 neighbours of an entry share one request).  6 KB of LDS per
// workgroup and < 128 registers: the occupancy is set by the registers, and the waves of a SIMD hide each other's latencies.
// Any level is handled correctly (a fine level just sends more requests than the cache would), so the split level is a
// tuning parameter of the launch (lse_hash_bwd_opts.coarse_levels), not a correctness condition.
#ifdef LSE_DEV_KNOBS   // development build only (csrc/dev_knobs.h): superseded / A-B variant, not in liblse_hip.so
template <bool WITH_DX>
__global__ __launch_bounds__(256) void hash_bwd_coarse_kernel(GridParams g, const float *__restrict__ x,
                                                              const float2 *__restrict__ dy,
                                                              const float2 *__restrict__ table,
                                                              float *__restrict__ dtable, float *__restrict__ dx,
                                                              int64_t n_cap, int dbg, float *__restrict__ ws,
                                                              const int64_t *__restrict__ n_dev)
{
    const int64_t n = lse::clamp_count(n_cap, n_dev);      // n_cap stays the level stride of dy
    constexpr int kBatch = 16;                      // run ends per staging trip
    __shared__ uint32_t s_stage[4][kBatch * 24];    // [run][8 idx | 16 sums]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wave_base = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (wave_base >= n) return;                     // (no workgroup barrier anywhere below)
    lds_u32 *stage = (lds_u32 *)&s_stage[wave][0];
    const int64_t i0 = wave_base + lane;
    const bool valid = i0 < n;
    const int64_t si = valid ? i0 : n - 1;
    const float px0 = x[si * 3 + 0], px1 = x[si * 3 + 1], px2 = x[si * 3 + 2];
    float dacc0 = 0.f, dacc1 = 0.f, dacc2 = 0.f;
    const int k16 = lane & 15, sub_run = lane >> 4;

    for (int l = g.l_begin; l < g.l_end; ++l) {
        const LevelInfo li = level_info(g, l);
        float *__restrict__ dt = (ws != nullptr && l < g.rep_levels)
                                     ? ws + (size_t)(blockIdx.x & (unsigned)g.rep_mask) * g.rep_stride + 2 * (size_t)li.offset
                                     : dtable + 2 * (size_t)li.offset;
        const float2 *__restrict__ tab = table + li.offset;
        float w0, w1, w2;
        uint32_t p0, p1, p2;
        pos_fract(px0, li.scale, w0, p0);
        pos_fract(px1, li.scale, w1, p1);
        pos_fract(px2, li.scale, w2, p2);
        float2 gy = dy[(int64_t)l * n_cap + si];
        if (!valid) gy = make_float2(0.f, 0.f);
        uint32_t idx[8];
        corner_indices(li, p0, p1, p2, idx);
        float2 tv[8];
        if (WITH_DX && !(dbg & 8)) {
#pragma unroll
            for (int c = 0; c < 8; ++c) tv[c] = tab[idx[c]];
        }
        const uint32_t q0 = wave_shr1(p0), q1 = wave_shr1(p1), q2 = wave_shr1(p2);
        const int head = (lane == 0) || (q0 != p0) || (q1 != p1) || (q2 != p2);
        const int next_head = wave_shl1(head);
        const bool run_end = ((lane == 63) || next_head) && !(dbg & 2);
        float v[16];
        const float a0 = 1.f - w0, a1 = 1.f - w1, a2 = 1.f - w2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wt = ((c & 1) ? w0 : a0) * ((c & 2) ? w1 : a1) * ((c & 4) ? w2 : a2);
            v[2 * c] = wt * gy.x;
            v[2 * c + 1] = wt * gy.y;
        }
        if (!(dbg & 4)) {
            const int row_pos = lane & 15;
            int flag = head | (row_pos == 0);
            if (seg_scan_row_step<1>(v, flag, row_pos) && seg_scan_row_step<2>(v, flag, row_pos) &&
                seg_scan_row_step<4>(v, flag, row_pos))
                seg_scan_row_step<8>(v, flag, row_pos);
            int open = !head;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int o = (off == 1) ? row_shr_i32<1>(open) : (off == 2) ? row_shr_i32<2>(open)
                            : (off == 4) ? row_shr_i32<4>(open) : row_shr_i32<8>(open);
                open &= (row_pos >= off) ? o : 1;
            }
            seg_scan_cross_rows(v, open, lane);
        }
        if (WITH_DX && !(dbg & 8)) {
            // d(x): t_c = <dy, table[c]>; along each axis the difference of the two faces, weighted by the other two axes
            float t[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) t[c] = gy.x * tv[c].x + gy.y * tv[c].y;
            const float d0 = a2 * (a1 * (t[1] - t[0]) + w1 * (t[3] - t[2])) + w2 * (a1 * (t[5] - t[4]) + w1 * (t[7] - t[6]));
            const float d1 = a2 * (a0 * (t[2] - t[0]) + w0 * (t[3] - t[1])) + w2 * (a0 * (t[6] - t[4]) + w0 * (t[7] - t[5]));
            const float d2 = a1 * (a0 * (t[4] - t[0]) + w0 * (t[5] - t[1])) + w1 * (a0 * (t[6] - t[2]) + w0 * (t[7] - t[3]));
            dacc0 = fmaf(li.scale, d0, dacc0);
            dacc1 = fmaf(li.scale, d1, dacc1);
            dacc2 = fmaf(li.scale, d2, dacc2);
        }
        const uint64_t ends_mask = __builtin_amdgcn_ballot_w64(run_end);
        const int n_ends = __builtin_popcountll(ends_mask);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends_mask, 0));
        for (int b0 = 0; b0 < n_ends; b0 += kBatch) {
            const int r = rank - b0;
            if (run_end && r >= 0 && r < kBatch) {
#pragma unroll
                for (int c = 0; c < 8; ++c) stage[r * 24 + c] = idx[c];
#pragma unroll
                for (int k = 0; k < 16; ++k) stage[r * 24 + 8 + k] = __float_as_uint(v[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(kBatch, n_ends - b0);
            for (int r0 = sub_run; r0 < cnt; r0 += 4) {
                const uint32_t e = stage[r0 * 24 + (k16 >> 1)];
                const float a = __uint_as_float(stage[r0 * 24 + 8 + k16]);
                if (a != 0.f && !(dbg & 1)) atomicAdd(dt + 2 * (size_t)e + (k16 & 1), a);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();      // the next trip (or level) overwrites the staging area
        }
    }
    if (WITH_DX && valid) {
        if (g.dx_accumulate) {
            dacc0 += dx[si * 3 + 0];
            dacc1 += dx[si * 3 + 1];
            dacc2 += dx[si * 3 + 2];
        }
        dx[si * 3 + 0] = dacc0;
        dx[si * 3 + 1] = dacc1;
        dx[si * 3 + 2] = dacc2;
    }
}
#endif  // LSE_DEV_KNOBS

// dtable[i] += sum over replicas; the workspace reads zero again afterwards (it is handed over zeroed and returned zeroed)
__global__ __launch_bounds__(256) void hash_bwd_reduce_replicas_kernel(float *__restrict__ ws, float *__restrict__ dtable,
                                                                       uint32_t n4, int n_rep, uint32_t stride)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < n_rep; ++r) {
        float4 *p = reinterpret_cast<float4 *>(ws + (size_t)r * stride) + i;
        const float4 v = *p;
        if (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f) {
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            *p = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (s.x != 0.f || s.y != 0.f || s.z != 0.f || s.w != 0.f) {
        float4 *d = reinterpret_cast<float4 *>(dtable) + i;
        float4 t = *d;
        t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
        *d = t;
    }
}

int fill_params(const lse_grid_desc *desc, GridParams &g, const char *who)
{
    LSE_REQUIRE(desc, "%s: null desc", who);
    LSE_REQUIRE(desc->n_levels >= 1 && desc->n_levels <= LSE_MAX_GRID_LEVELS, "%s: n_levels %d out of range", who,
                desc->n_levels);
    LSE_REQUIRE(desc->n_features == 2, "%s: only n_features == 2 is implemented (got %d)", who, desc->n_features);
    g.n_levels = desc->n_levels;
    g.l_begin = 0;
    g.l_end = desc->n_levels;
    g.dx_accumulate = 0;
    g.rep_levels = 0;
    g.rep_mask = 0;
    g.rep_stride = 0;
    for (int l = 0; l <= desc->n_levels; ++l) g.offsets[l] = desc->offsets[l];
    for (int l = 0; l < desc->n_levels; ++l) {
        LSE_REQUIRE(desc->offsets[l + 1] > desc->offsets[l], "%s: level %d is empty", who, l);
        LSE_REQUIRE(desc->resolutions[l] >= 2, "%s: level %d resolution < 2", who, l);
        g.scales[l] = desc->scales[l];
        g.res[l] = desc->resolutions[l];
    }
    return LSE_OK;
}

}  // namespace

extern "C" int lse_hash_fwd(const lse_grid_desc *desc, const float *x01, const float *table, float *y, int64_t n,
                            const int64_t *n_dev, lse_stream_t stream)
{
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_fwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_hash_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && table && y, "lse_hash_fwd: null pointer");
    const int64_t chunks = (n + kFwdThreads * kFwdItems - 1) / (kFwdThreads * kFwdItems);
    const int mapping = (int)lse::option("hash_fwd_mapping");
    int lds_levels = 0;
    hipStream_t st = lse::as_stream(stream);
#ifdef LSE_DEV_KNOBS
    // knob hash_fwd_lds_levels = k: the k coarsest levels whose whole table fits one CU's LDS run in hash_fwd_lds_kernel (needs
    // the default mapping 4, whose grid simply ends k levels earlier)
    if (mapping == 4) {
        const int want = (int)std::min<int64_t>(lse::option("hash_fwd_lds_levels"), g.n_levels - 1);
        while (lds_levels < want && (int64_t)(g.offsets[lds_levels + 1] - g.offsets[lds_levels]) * 8 <= 156 * 1024) ++lds_levels;
    }
    for (int l = 0; l < lds_levels; ++l) {
        const int lds_bytes = (int)(g.offsets[l + 1] - g.offsets[l]) * 8;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_fwd_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);     // (per device: set per call)
        if (e != hipSuccess) {
            lse::set_error("lse_hash_fwd: cannot raise dynamic LDS to %d bytes: %s", lds_bytes, hipGetErrorString(e));
            return LSE_E_LAUNCH;
        }
        // as many workgroups as fit the chip at once with this much LDS each (16 waves per workgroup: at most 2 per CU)
        const int per_cu = lds_bytes <= 78 * 1024 ? 2 : 1;
        const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(256 * per_cu, (n + 2 * kFwdLdsThreads - 1) / (2 * kFwdLdsThreads)));
        hipLaunchKernelGGL(hash_fwd_lds_kernel, dim3((unsigned)wgs), dim3(kFwdLdsThreads), lds_bytes, st, g, l, x01,
                           reinterpret_cast<const float2 *>(table), reinterpret_cast<float2 *>(y), n, n_dev);
    }
#endif
    const int64_t blocks = chunks * (g.n_levels - lds_levels);
    LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_fwd: grid too large");
    hipLaunchKernelGGL(hash_fwd_kernel, dim3((unsigned)blocks), dim3(kFwdThreads), 0, st, g, x01,
                       reinterpret_cast<const float2 *>(table), reinterpret_cast<float2 *>(y), n, chunks, mapping,
                       n_dev, lds_levels);
    return lse::check_launch("lse_hash_fwd");
}

extern "C" void lse_hash_bwd_default_opts(lse_hash_bwd_opts *o)
{
    if (!o) return;
    o->impl = 2;           // lane-per-sample + LDS sector cache, run ends of several levels batched per cache pass
    o->stage_max = (int32_t)lse::option("hash_bwd_stage_max");   // impl 2: a level ending more runs than this per wave passes unstaged
                           // when the queue is empty
    o->gran = 6;           // impl 2: 512 slots of one 32-B sector, paired by 64-B line, second-generation flush (4 = first-generation
                           // flush, 2 = unpaired); impl 1: 2 / 3
    o->few_runs = (int32_t)lse::option("hash_bwd_few_runs");     // 8 (round 5, profiles/r05_hash_bwd_thresholds.txt: with stage_max 48 the best pair for steps
                           // that grow with the distance -- the reference's default configuration 1.51 -> 1.43 ms -- and neutral
                           // (+0.2 %) at the metric size, whose constant step prefers fewer direct adds -- (3, 56) on the final kernel; callers that know the
                           // regime pass the pair (lsenerf_amd/ops.py: HASH_BWD_DENSE_STEPS); 10 and 16: slower everywhere)
    o->second_probe = (int32_t)lse::option("hash_bwd_probes");   // extra probe rounds (home + k * step) before a corner goes to memory
                           // alone; pays wherever the kernel is bound by atomic requests, costs ~2 % per round where it is issue-bound.
                           // 1 -> 3 rounds (round 3, once the later rounds probed NEW slots): requests 50.5 -> 48.2 M (metric size),
                           // 28.3 -> 26.9 M (default configuration); hash_bwd 2.66 -> 2.66 / 1.40 -> 1.34 / 1.13 -> 1.05 ms (metric size /
                           // default configuration / 3-bundle step), M-packed step 6.07 -> 5.99; 4 rounds: slower everywhere
    o->rounds = 32;
    o->dbg = 0;
    o->interleave_from_scale = 1e30f;   // measured negative on MI355X (same-address lanes of one atomic instruction serialise)
    o->coarse_levels = 0;  // levels below this one run in the cache-free high-occupancy kernel (hash_bwd_coarse_kernel) first
    o->replicas = 16;      // direct adds of the levels < replica_levels go to this many replicas (power of two); needs `workspace`
                           // (M-packed 3.67 -> 3.21 ms with 8, 3.16 with 16, 3.18 with 32; M-march and the default configuration +-1 %)
    o->replica_levels = 4; // 16^3 .. 43^3: 125 568 entries = 1 MB per replica
    o->workspace = nullptr;
    o->workspace_bytes = 0;
    o->prefetch = 0;
}

// The replicas pay where a level is a few thousand lines that every wave of the launch adds to (16^3 .. 43^3: 1 MB per replica); for
// a grid whose coarse levels are already big (base_res >= ~64 at T = 2^19: 4 hashed levels = 16 MB per replica, 268 MB for 16, all of
// it swept by hash_bwd_reduce_replicas_kernel after every backward) they would cost memory and time for nothing: the replicated
// levels stop at the first level that would push ONE replica past this budget.
constexpr int64_t kReplicaBudgetFloats = (2 << 20) / (int64_t)sizeof(float);      // 2 MB per replica

static int64_t replica_floats(const lse_grid_desc *desc, const lse_hash_bwd_opts &o, int *levels_out)
{
    int lv = std::min<int>(o.replica_levels, desc->n_levels);
    while (lv > 0 && (int64_t)2 * desc->offsets[lv] > kReplicaBudgetFloats) --lv;
    if (o.replicas < 2 || lv <= 0) { if (levels_out) *levels_out = 0; return 0; }
    if (levels_out) *levels_out = lv;
    return (int64_t)2 * desc->offsets[lv];        // floats per replica (level offsets are multiples of 8 entries)
}

extern "C" int64_t lse_hash_bwd_workspace_bytes(const lse_grid_desc *desc, const lse_hash_bwd_opts *opts)
{
    if (!desc) return 0;
    lse_hash_bwd_opts o;
    lse_hash_bwd_default_opts(&o);
    if (opts) o = *opts;
    return replica_floats(desc, o, nullptr) * (int64_t)sizeof(float) * std::max(o.replicas, 0);
}

extern "C" int lse_hash_bwd(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                            float *dtable, float *dx, int64_t n, const int64_t *n_dev, lse_stream_t stream)
{
    return lse_hash_bwd_ex(desc, x01, dy, table, dtable, dx, 0, 0, desc ? desc->n_levels : 0, n, n_dev, nullptr, stream);
}

extern "C" int lse_hash_bwd_levels(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                                   float *dtable, float *dx, int32_t dx_accumulate, int32_t level_begin,
                                   int32_t level_end, int64_t n, const int64_t *n_dev, lse_stream_t stream)
{
    return lse_hash_bwd_ex(desc, x01, dy, table, dtable, dx, dx_accumulate, level_begin, level_end, n, n_dev, nullptr, stream);
}

extern "C" int lse_hash_bwd_ex(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                               float *dtable, float *dx, int32_t dx_accumulate, int32_t level_begin, int32_t level_end,
                               int64_t n, const int64_t *n_dev, const lse_hash_bwd_opts *opts, lse_stream_t stream)
{
    lse_hash_bwd_opts o;
    lse_hash_bwd_default_opts(&o);
    if (opts) o = *opts;
#ifndef LSE_DEV_KNOBS
    // the library that ships holds the production variants only: the batched sector-cache kernel (impl 2, gran 6) and the generic
    // 16-lanes-per-sample kernel (impl 0; also taken for grids whose levels do not start on 64-byte lines)
    if (!((o.impl == 2 && o.gran == 6 && !o.prefetch) || o.impl == 0) || o.coarse_levels != 0 || o.dbg != 0 || o.rounds != 32 ||
        o.interleave_from_scale < 1e29f) {
        lse::set_error("lse_hash_bwd: impl %d / gran %d / prefetch %d / coarse_levels %d / rounds %d / dbg %d selects a development "
                       "variant (csrc/dev_knobs.h: build liblse_hip_dev.so)", o.impl, o.gran, o.prefetch, o.coarse_levels, o.rounds, o.dbg);
        return LSE_E_UNSUPPORTED;
    }
#endif
    LSE_REQUIRE(o.impl >= 0 && o.impl <= 2, "lse_hash_bwd: opts.impl must be 0, 1 or 2");
    LSE_REQUIRE(o.stage_max >= 0 && o.stage_max <= 64, "lse_hash_bwd: opts.stage_max must be in [0, 64]");
    LSE_REQUIRE(o.gran >= 2 && o.gran <= 13, "lse_hash_bwd: opts.gran must be 2 .. 13");
    LSE_REQUIRE(o.rounds == 16 || o.rounds == 32 || o.rounds == 64, "lse_hash_bwd: opts.rounds must be 16, 32 or 64");
    LSE_REQUIRE(o.few_runs >= 0 && o.few_runs <= 16, "lse_hash_bwd: opts.few_runs must be in [0, 16]");
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_bwd");
    if (rc) return rc;
    LSE_REQUIRE(0 <= level_begin && level_begin <= level_end && level_end <= g.n_levels,
                "lse_hash_bwd_levels: bad level range [%d, %d)", level_begin, level_end);
    g.l_begin = level_begin;
    g.l_end = level_end;
    g.dx_accumulate = dx_accumulate;
    if (level_begin == level_end && !(dx && !dx_accumulate)) return LSE_OK;
    LSE_REQUIRE(n >= 0, "lse_hash_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && dy && dtable, "lse_hash_bwd: null pointer");
    LSE_REQUIRE(!dx || table, "lse_hash_bwd: dx requested but table is null");
    LSE_REQUIRE(o.coarse_levels >= 0 && o.coarse_levels <= LSE_MAX_GRID_LEVELS, "lse_hash_bwd: opts.coarse_levels out of range");
    const float il_scale = o.interleave_from_scale;
    const int rounds = o.rounds, impl = o.impl, dbg = o.dbg;
    (void)rounds;
#ifndef LSE_DEV_KNOBS
    (void)dbg;      // (timing ablations exist in the development build only: rejected above)
#endif
    hipStream_t st = lse::as_stream(stream);
    const float *tb = dx ? table : nullptr;
    // device-side sample count: honoured by the default kernel (and the development build's coarse kernel)
    LSE_REQUIRE(!n_dev || (impl == 2), "lse_hash_bwd: a device-side count needs opts.impl == 2");
    // replicas of the coarsest levels for the direct adds (impl 2's few-runs path and the coarse kernel)
    float *ws = nullptr;
    int rep_lv = 0;
    const int64_t rep_floats = replica_floats(desc, o, &rep_lv);
    if (o.workspace && rep_floats > 0 && level_begin < rep_lv) {
        LSE_REQUIRE((o.replicas & (o.replicas - 1)) == 0 && o.replicas <= 64, "lse_hash_bwd: opts.replicas must be a power of two <= 64");
        LSE_REQUIRE(o.workspace_bytes >= rep_floats * (int64_t)sizeof(float) * o.replicas,
                    "lse_hash_bwd: workspace of %lld bytes, %lld needed (lse_hash_bwd_workspace_bytes)", (long long)o.workspace_bytes,
                    (long long)(rep_floats * (int64_t)sizeof(float) * o.replicas));
        LSE_REQUIRE(((uintptr_t)o.workspace & 15) == 0, "lse_hash_bwd: workspace must be 16-byte aligned");
        ws = static_cast<float *>(o.workspace);
        g.rep_levels = rep_lv;
        g.rep_mask = o.replicas - 1;
        g.rep_stride = (uint32_t)rep_floats;
    }
    struct ReduceAtExit {     // the replicas are folded into dtable after whatever kernels this call launches
        float *ws, *dtable; int64_t floats; int n_rep; hipStream_t st;
        ~ReduceAtExit() {
            if (!ws) return;
            const uint32_t n4 = (uint32_t)(floats / 4);
            hipLaunchKernelGGL(hash_bwd_reduce_replicas_kernel, dim3((n4 + 255) / 256), dim3(256), 0, st, ws, dtable, n4, n_rep,
                               (uint32_t)floats);
        }
    } reduce_at_exit{ws, dtable, rep_floats, o.replicas, st};
#ifdef LSE_DEV_KNOBS
    // the coarse share of the level range first, in the cache-free kernel; the rest of the range then accumulates into dx
    const int coarse_end = std::min<int>(o.coarse_levels, level_end);
    if (coarse_end > level_begin) {
        GridParams gc = g;
        gc.l_end = coarse_end;
        const int64_t blocks = (n + 4 * 64 - 1) / (4 * 64);
        LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");
        const float2 *dy2 = reinterpret_cast<const float2 *>(dy);
        const float2 *tb2 = reinterpret_cast<const float2 *>(tb);
        if (dx) hipLaunchKernelGGL((hash_bwd_coarse_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, gc, x01, dy2, tb2, dtable,
                                   dx, n, dbg, ws, n_dev);
        else hipLaunchKernelGGL((hash_bwd_coarse_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, gc, x01, dy2, tb2, dtable,
                                dx, n, dbg, ws, n_dev);
        rc = lse::check_launch("lse_hash_bwd (coarse levels)");
        if (rc) return rc;
        if (coarse_end == level_end) return LSE_OK;
        g.l_begin = coarse_end;
        g.dx_accumulate = 1;
    }
#endif
    // line-cache kernel: needs every level to start on a 64-B line (tcnn pads level sizes to 8 entries)
    bool lines_ok = true;
    for (int l = 0; l <= g.n_levels; ++l) lines_ok = lines_ok && (g.offsets[l] % 8 == 0);
    LSE_REQUIRE(!n_dev || lines_ok, "lse_hash_bwd: a device-side count needs 64-byte aligned levels (the default kernel)");
    if (impl == 2 && lines_ok) {     // per-wave sector cache with run ends of several levels batched into one pass
        const int64_t blocks = ((n + 4 * 64 - 1) / (4 * 64) + 7) / 8 * 8;      // (four-wave development variants; chunk mapping: a multiple of 8)
        LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");
        const float2 *dy2 = reinterpret_cast<const float2 *>(dy);
        const float2 *tb2 = reinterpret_cast<const float2 *>(tb);
        // (workgroups of WAVES waves with SLOTS cache slots per wave; nothing in the kernel synchronises across waves)
#define LSE_HASH_BWD_LAUNCH(SLOTS, WAVES)                                                                                              \
        {                                                                                                                                 \
            const int64_t bl = ((n + (WAVES) * 64 - 1) / ((WAVES) * 64) + 7) / 8 * 8;      /* (chunk mapping: a multiple of 8) */          \
            LSE_REQUIRE(bl < (1ll << 31), "lse_hash_bwd: grid too large");                                                                \
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, SLOTS, 2, true, true, false, false, WAVES>), dim3((unsigned)bl),    \
                                       dim3(64 * (WAVES)), 0, st, g, x01, dy2, tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe,       \
                                       o.stage_max, ws, n_dev);                                                                          \
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, SLOTS, 2, true, true, false, false, WAVES>), dim3((unsigned)bl),      \
                                    dim3(64 * (WAVES)), 0, st, g, x01, dy2, tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe,          \
                                    o.stage_max, ws, n_dev);                                                                             \
            return lse::check_launch("lse_hash_bwd");                                                                                     \
        }
#ifdef LSE_DEV_KNOBS
        if (o.gran == 5) {      // the same with 256 slots: half the LDS, three workgroups per CU
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 256, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                       tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 256, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                    tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            return lse::check_launch("lse_hash_bwd");
        }
        if (o.gran == 7) {      // gran 6 with pair-aligned flush lists (a line is never split between two flush instructions)
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 512, 2, true, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                       tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 512, 2, true, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                    tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            return lse::check_launch("lse_hash_bwd");
        }
        if (o.gran == 6 && o.prefetch) {      // ... and the next level's operands fetched ahead of the cache pass
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 512, 2, true, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                       tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 512, 2, true, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                    tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            return lse::check_launch("lse_hash_bwd");
        }
#endif
#ifdef LSE_DEV_KNOBS
        // gran 6 with 320 slots: 52 KB of LDS per workgroup -> three workgroups (12 waves) per CU.  Round 5: more waves hide more
        // latency, fewer slots collide more, and the two cancel where it matters -- headline 2.56 -> 2.58 ms, default configuration
        // 1.49 -> 1.54, M-packed 3.12 -> 3.23; only the inside-box workload (long runs in the contracted shell, few sectors per
        // sample) gains, 2.60 -> 2.47 (profiles/r05_hash_bwd_slots320.txt).  The production kernel sits where its CU-side bound
        // (2 waves per SIMD) and the memory-side request bound (more requests at 3 waves per SIMD) meet.
        if (o.gran == 8) {
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 320, 2, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                       tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 320, 2, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                    tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            return lse::check_launch("lse_hash_bwd");
        }
        // between the two: smaller workgroups hand the LDS out in finer portions.  gran 9: 384 slots, two waves per workgroup (30.5 KB)
        // -> 10 waves per CU; gran 10: 448 slots, one wave per workgroup (17.6 KB) -> 9 waves per CU; gran 11: 384 slots, one wave; gran 12 / 13: 512 slots in workgroups of four (the shape until round 5) / two waves.
        if (o.gran == 9) LSE_HASH_BWD_LAUNCH(384, 2)
        if (o.gran == 10) LSE_HASH_BWD_LAUNCH(448, 1)
        if (o.gran == 11) LSE_HASH_BWD_LAUNCH(384, 1)
        if (o.gran == 12) LSE_HASH_BWD_LAUNCH(512, 4)
        if (o.gran == 13) LSE_HASH_BWD_LAUNCH(512, 2)
#endif
        // THE production kernel: paired 32-byte sectors, second-generation flush, 512 slots per wave, ONE wave per workgroup.  A
        // workgroup's LDS is released when its last wave retires, and waves differ in how many cache passes their samples need:
        // in workgroups of four (until round 5) a CU held 20 KB per early finisher idle.  One wave per workgroup: samples in the
        // contracted shell 2.53 -> 2.40 ms, headline, M-packed and the default configuration within +-1 %
        // (profiles/r05_hash_bwd_workgroup_size.txt).
        if (o.gran == 6) LSE_HASH_BWD_LAUNCH(512, 1)
#undef LSE_HASH_BWD_LAUNCH
#ifdef LSE_DEV_KNOBS
        if (o.gran == 4) {      // 32-byte slots paired by 64-byte line, flush list in slot order
            if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 512, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                       tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 512, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                    tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
            return lse::check_launch("lse_hash_bwd");
        }
        if (dx) hipLaunchKernelGGL((hash_bwd_batched_kernel<true, 512, 2>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                   tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
        else hipLaunchKernelGGL((hash_bwd_batched_kernel<false, 512, 2>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2,
                                tb2, dtable, dx, n, dbg, o.few_runs, o.second_probe, o.stage_max, ws, n_dev);
        return lse::check_launch("lse_hash_bwd");
#else
        lse::set_error("lse_hash_bwd: opts.gran %d is a development variant", o.gran);     // (rejected above already)
        return LSE_E_UNSUPPORTED;
#endif
    }
#ifdef LSE_DEV_KNOBS
    if (impl >= 1 && lines_ok) {
        const int few_runs = o.few_runs, second_probe = o.second_probe, gran = o.gran;
        const int64_t blocks = (n + 4 * 64 - 1) / (4 * 64);
        LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");
        const float2 *dy2 = reinterpret_cast<const float2 *>(dy);
        const float2 *tb2 = reinterpret_cast<const float2 *>(tb);
#define LSE_LAUNCH_CACHED(DX, SLOTS, ENTLOG2)                                                                          \
    hipLaunchKernelGGL((hash_bwd_cached_kernel<DX, SLOTS, ENTLOG2>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy2, \
                       tb2, dtable, dx, n, dbg, few_runs, second_probe)
        if (gran != 3) {   // 512 slots of one 32-B sector (gran 4 = the pairing of impl 2: plain sectors here)
            if (dx) LSE_LAUNCH_CACHED(true, 512, 2);
            else LSE_LAUNCH_CACHED(false, 512, 2);
        } else {           // 256 slots of one 64-B line
            if (dx) LSE_LAUNCH_CACHED(true, 256, 3);
            else LSE_LAUNCH_CACHED(false, 256, 3);
        }
#undef LSE_LAUNCH_CACHED
        return lse::check_launch("lse_hash_bwd");
    }
#endif
#define LSE_LAUNCH_BWD(R)                                                                                             \
    do {                                                                                                              \
        const int64_t blocks = (n + 4 * 4 * R - 1) / (4 * 4 * R);                                                     \
        LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");                                            \
        if (dx) hipLaunchKernelGGL((hash_bwd_kernel<true, R>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy, tb, \
                                   dtable, dx, n, il_scale, dbg);                                                     \
        else hipLaunchKernelGGL((hash_bwd_kernel<false, R>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy, tb,   \
                                dtable, dx, n, il_scale, dbg);                                                        \
    } while (0)
#ifdef LSE_DEV_KNOBS
    if (rounds == 64) LSE_LAUNCH_BWD(64);
    else if (rounds == 16) LSE_LAUNCH_BWD(16);
    else
#endif
    LSE_LAUNCH_BWD(32);
#undef LSE_LAUNCH_BWD
    return lse::check_launch("lse_hash_bwd");
}
