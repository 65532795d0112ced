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
//   * forward: one (level, sample-chunk) per workgroup with the level chosen from blockIdx % 8.  Workgroups
//     are dealt round-robin over the 8 XCDs, so each XCD's private 4 MiB L2 only ever sees 1/8 of the levels
//     (2 of 16) instead of the whole 48.8 MB table.  Placement affects speed only, never results.
//   * backward: 16 lanes per sample (one per corner x feature) with run-length pre-accumulation, see hash_bwd_kernel;
//     table gradients are f32 global atomics (memory-side on gfx950, so no level/XCD affinity is attempted there).
#include "common.h"
#include <stdlib.h>

namespace {

struct GridParams {
    int n_levels;
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

__device__ __forceinline__ void pos_fract(float x, float scale, float &w, uint32_t &p)
{
    const float pos = fmaf(scale, x, 0.5f);
    const float fl = floorf(pos);
    p = (uint32_t)(int)fl;
    w = pos - fl;
}

constexpr int kFwdThreads = 256;
constexpr int kFwdItems = 4;   // samples per lane -> 32 independent gathers in flight

__global__ __launch_bounds__(kFwdThreads) void hash_fwd_kernel(GridParams g, const float *__restrict__ x,
                                                               const float2 *__restrict__ table,
                                                               float2 *__restrict__ y, int64_t n, int64_t chunks,
                                                               int mapping)
{
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
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = grid_index(li, p[it][0] + (c & 1), p[it][1] + ((c >> 1) & 1), p[it][2] + ((c >> 2) & 1));
            v[it][c] = tab[idx];
        }
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
        if (valid[it]) y[(int64_t)level * n + i] = r;
    }
}

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
                                                       float *__restrict__ dx, int64_t n, float interleave_from_scale)
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

    for (int l = 0; l < g.n_levels; ++l) {
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
                if (cur != kNone) atomicAdd(dt + 2 * (size_t)cur, acc);
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
        if (cur != kNone) atomicAdd(dt + 2 * (size_t)cur, acc);
    }
    if (WITH_DX) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = lane; c < kChunk; c += 64) {
            const int64_t i = wave_base + c;
            if (i < n) {
                dx[i * 3 + 0] = s_dx[wave][0][LSE_SLOT(c)];
                dx[i * 3 + 1] = s_dx[wave][1][LSE_SLOT(c)];
                dx[i * 3 + 2] = s_dx[wave][2][LSE_SLOT(c)];
            }
        }
    }
#undef LSE_SLOT
}

int fill_params(const lse_grid_desc *desc, GridParams &g, const char *who)
{
    LSE_REQUIRE(desc, "%s: null desc", who);
    LSE_REQUIRE(desc->n_levels >= 1 && desc->n_levels <= LSE_MAX_GRID_LEVELS, "%s: n_levels %d out of range", who,
                desc->n_levels);
    LSE_REQUIRE(desc->n_features == 2, "%s: only n_features == 2 is implemented (got %d)", who, desc->n_features);
    g.n_levels = desc->n_levels;
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
                            lse_stream_t stream)
{
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_fwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_hash_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && table && y, "lse_hash_fwd: null pointer");
    const int64_t chunks = (n + kFwdThreads * kFwdItems - 1) / (kFwdThreads * kFwdItems);
    const int64_t blocks = chunks * g.n_levels;
    LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_fwd: grid too large");
    static const int mapping = getenv("LSE_HASH_FWD_MAPPING") ? atoi(getenv("LSE_HASH_FWD_MAPPING")) : 0;
    hipLaunchKernelGGL(hash_fwd_kernel, dim3((unsigned)blocks), dim3(kFwdThreads), 0, lse::as_stream(stream), g, x01,
                       reinterpret_cast<const float2 *>(table), reinterpret_cast<float2 *>(y), n, chunks, mapping);
    return lse::check_launch("lse_hash_fwd");
}

extern "C" int lse_hash_bwd(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                            float *dtable, float *dx, int64_t n, lse_stream_t stream)
{
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_bwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_hash_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && dy && dtable, "lse_hash_bwd: null pointer");
    LSE_REQUIRE(!dx || table, "lse_hash_bwd: dx requested but table is null");
    // Levels with scale >= il_scale use the interleaved (4 consecutive samples per instruction) mapping.  Measured
    // negative on MI355X (same-address lanes inside one atomic instruction serialise): default = never.
    static const float il_scale = getenv("LSE_HASH_BWD_INTERLEAVE_SCALE") ? (float)atof(getenv("LSE_HASH_BWD_INTERLEAVE_SCALE")) : 1e30f;
    static const int rounds = getenv("LSE_HASH_BWD_ROUNDS") ? atoi(getenv("LSE_HASH_BWD_ROUNDS")) : 32;
    hipStream_t st = lse::as_stream(stream);
    const float *tb = dx ? table : nullptr;
#define LSE_LAUNCH_BWD(R)                                                                                             \
    do {                                                                                                              \
        const int64_t blocks = (n + 4 * 4 * R - 1) / (4 * 4 * R);                                                     \
        LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");                                            \
        if (dx) hipLaunchKernelGGL((hash_bwd_kernel<true, R>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy, tb, \
                                   dtable, dx, n, il_scale);                                                          \
        else hipLaunchKernelGGL((hash_bwd_kernel<false, R>), dim3((unsigned)blocks), dim3(256), 0, st, g, x01, dy, tb,   \
                                dtable, dx, n, il_scale);                                                             \
    } while (0)
    if (rounds == 32) LSE_LAUNCH_BWD(32);
    else if (rounds == 64) LSE_LAUNCH_BWD(64);
    else LSE_LAUNCH_BWD(16);
#undef LSE_LAUNCH_BWD
    return lse::check_launch("lse_hash_bwd");
}
