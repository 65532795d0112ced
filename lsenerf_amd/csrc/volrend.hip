// Packed per-ray transmittance scan + alpha compositing for gfx950, forward and backward:
//   nerfacc.render_weight_from_density            R:lse_nerf/lsenerf.py:301-306
//   nerfacc.render_visibility_from_density        R:lse_nerf/lse_grid_estimator.py:120-127
//   RGB / accumulation / depth renderers          R:lse_nerf/lsenerf.py:309-318, R:lse_nerf/lse_renderer.py:6-10
//
// One 64-lane wave owns one ray: its samples are contiguous (packed, ray-sorted), so a chunk of 64 samples is
// one coalesced load per stream, the exclusive sum of sigma*dt is a wave prefix scan with a scalar carry between
// chunks, and the per-ray sums (rgb, accumulation, depth numerator) are wave reductions -- no atomics, no
// index_add_, deterministic.  The backward is the same walk in reverse with suffix scans:
//     dL/d(sd_k) = dw_k * T_{k+1} - sum_{i>k} dw_i w_i ,   T_{k+1} = T_end + sum_{i>k} w_i ,  T_end = 1 - sum_i w_i
// (T_{k+1} = T_k - w_k exactly, so no transmittance array is stored and no cancellation-prone prefix is formed).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_inclusive_sum_rev(float v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float nb = __shfl_down(v, off, 64);
        if (lse::lane_id() + off < 64) v += nb;
    }
    return v;
}

__global__ __launch_bounds__(256) void volrend_fwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                          const float *__restrict__ sigma, const float *__restrict__ rgb,
                                                          int rgb_stride, const int64_t *__restrict__ packed, int n_rays,
                                                          float *__restrict__ weights, float *__restrict__ out_rgb,
                                                          float *__restrict__ out_acc, float *__restrict__ out_dep,
                                                          float *__restrict__ trans, float *__restrict__ alphas,
                                                          float *__restrict__ mid_range)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    float carry = 0.f, ar = 0.f, ag = 0.f, ab = 0.f, aw = 0.f, ad = 0.f;
    float mlo = INFINITY, mhi = -INFINITY;   // smallest / largest interval mid-point of this ray (DepthRenderer's clip range)
    for (int64_t base = 0; base < cnt; base += 64) {
        const int64_t i = s0 + base + lane;
        const bool valid = base + lane < cnt;
        float a = 0.f, b = 0.f, sg = 0.f;
        if (valid) { a = ts[i]; b = te[i]; sg = sigma[i]; }
        const float sd = sg * (b - a);
        const float incl = lse::wave_inclusive_sum(sd);
        const float excl = (incl - sd) + carry;
        const float T = expf(-excl);
        const float alpha = 1.f - expf(-sd);
        const float w = valid ? T * alpha : 0.f;
        if (valid) {
            weights[i] = w;
            if (trans) trans[i] = T;
            if (alphas) alphas[i] = alpha;
            mlo = fminf(mlo, (a + b) * 0.5f);
            mhi = fmaxf(mhi, (a + b) * 0.5f);
            if (rgb) {
                const float *c = rgb + i * rgb_stride;
                ar = fmaf(w, c[0], ar);
                ag = fmaf(w, c[1], ag);
                ab = fmaf(w, c[2], ab);
            }
            aw += w;
            ad = fmaf(w, (a + b) * 0.5f, ad);
        }
        carry += __shfl(incl, 63, 64);
    }
    ar = lse::wave_sum(ar); ag = lse::wave_sum(ag); ab = lse::wave_sum(ab);
    aw = lse::wave_sum(aw); ad = lse::wave_sum(ad);
    if (mid_range) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mlo = fminf(mlo, __shfl_xor(mlo, off, 64));
            mhi = fmaxf(mhi, __shfl_xor(mhi, off, 64));
        }
    }
    if (lane == 0) {
        if (out_rgb) { out_rgb[ray * 3 + 0] = ar; out_rgb[ray * 3 + 1] = ag; out_rgb[ray * 3 + 2] = ab; }
        if (out_acc) out_acc[ray] = aw;
        if (out_dep) out_dep[ray] = ad;
        if (mid_range) { mid_range[2 * ray] = mlo; mid_range[2 * ray + 1] = mhi; }
    }
}

// DepthRenderer("expected") epilogue (nerfstudio 0.3.2): depth = num / (acc + 1e-10), clipped to the
// GLOBAL [min, max] of the interval mid-points.  Samples are sorted inside a ray, so the global range is the min / max over
// the per-ray ranges the forward kernel leaves in mid_range[R][2] -- an O(R) reduction in one workgroup instead of two
// reductions over all N samples.  range_out[2] keeps (lo, hi) for the backward.
__global__ __launch_bounds__(1024) void depth_finish_kernel(const float *__restrict__ num, const float *__restrict__ acc,
                                                            const float *__restrict__ mid_range, int n_rays,
                                                            float *__restrict__ depth, float *__restrict__ range_out)
{
    __shared__ float s_lo[16], s_hi[16];
    float lo = INFINITY, hi = -INFINITY;
    for (int r = threadIdx.x; r < n_rays; r += 1024) {
        lo = fminf(lo, mid_range[2 * r]);
        hi = fmaxf(hi, mid_range[2 * r + 1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    lo = s_lo[0]; hi = s_hi[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); }
    if (threadIdx.x == 0 && range_out) { range_out[0] = lo; range_out[1] = hi; }
    const bool any = lo <= hi;     // no sample at all: nerfstudio's clip is skipped (steps is empty)
    for (int r = threadIdx.x; r < n_rays; r += 1024) {
        float d = num[r] / (acc[r] + 1e-10f);
        if (any) d = fminf(fmaxf(d, lo), hi);
        depth[r] = d;
    }
}

__global__ __launch_bounds__(256) void volrend_bwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                          const float *__restrict__ sigma, const float *__restrict__ rgb,
                                                          int rgb_stride, const int64_t *__restrict__ packed, int n_rays,
                                                          const float *__restrict__ weights,
                                                          const float *__restrict__ g_rgb, const float *__restrict__ g_acc,
                                                          const float *__restrict__ g_dep, float *__restrict__ d_sigma,
                                                          float *__restrict__ d_rgb, const float *__restrict__ g_w)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    if (cnt == 0) return;
    const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f,
                gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float ga = g_acc ? g_acc[ray] : 0.f, gd = g_dep ? g_dep[ray] : 0.f;
    float wtot = 0.f;
    for (int64_t base = 0; base < cnt; base += 64)
        if (base + lane < cnt) wtot += weights[s0 + base + lane];
    wtot = lse::wave_sum(wtot);
    const float t_end = 1.f - wtot;
    float carry_w = 0.f, carry_g = 0.f;
    const int64_t n_chunks = (cnt + 63) / 64;
    for (int64_t ch = n_chunks - 1; ch >= 0; --ch) {
        const int64_t base = ch * 64;
        const int64_t i = s0 + base + lane;
        const bool valid = base + lane < cnt;
        float w = 0.f, dw = 0.f, dt = 0.f;
        if (valid) {
            w = weights[i];
            const float a = ts[i], b = te[i];
            dt = b - a;
            dw = ga + gd * ((a + b) * 0.5f);
            if (g_w) dw += g_w[i];            // per-sample gradient of the weights themselves (render_weight_from_density)
            if (rgb && g_rgb) {
                const float *c = rgb + i * rgb_stride;
                dw += gr * c[0] + gg * c[1] + gb * c[2];
            }
        }
        const float gw = dw * w;
        const float incl_w = wave_inclusive_sum_rev(w);
        const float incl_g = wave_inclusive_sum_rev(gw);
        const float sw = (incl_w - w) + carry_w;     // sum_{i>k} w_i
        const float sg = (incl_g - gw) + carry_g;    // sum_{i>k} dw_i w_i
        if (valid) {
            const float dsd = dw * (t_end + sw) - sg;
            d_sigma[i] = dsd * dt;
            if (d_rgb) {
                float *dc = d_rgb + i * rgb_stride;
                if (rgb_stride == 4) *reinterpret_cast<float4 *>(dc) = make_float4(w * gr, w * gg, w * gb, 0.f);   // pad column too
                else { dc[0] = w * gr; dc[1] = w * gg; dc[2] = w * gb; }
            }
        }
        carry_w += __shfl(incl_w, 0, 64);
        carry_g += __shfl(incl_g, 0, 64);
    }
}

__global__ __launch_bounds__(256) void visibility_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                         const float *__restrict__ sigma,
                                                         const int64_t *__restrict__ packed, int n_rays, float eps,
                                                         float alpha_thre, uint8_t *__restrict__ mask,
                                                         int64_t *__restrict__ new_cnts, int from_alpha,
                                                         const float *__restrict__ alpha_cap)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    if (alpha_cap != nullptr) alpha_thre = fminf(alpha_thre, *alpha_cap);      // min(alpha_thre, occs.mean()) without the .item()
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    float carry = 0.f;
    int64_t kept = 0;
    for (int64_t base = 0; base < cnt; base += 64) {
        const int64_t i = s0 + base + lane;
        const bool valid = base + lane < cnt;
        float a = 0.f, b = 0.f, sg = 0.f;
        if (valid) {
            sg = sigma[i];
            if (!from_alpha) { a = ts[i]; b = te[i]; }
        }
        // from_alpha (nerfacc.render_visibility_from_alpha): `sigma` holds opacities; T_k = prod_{i<k} (1 - alpha_i) is
        // formed as exp(-sum -log(1 - alpha_i)) with the same scan, the alpha threshold is tested on the input value itself
        const float sd = from_alpha ? -log1pf(-fminf(sg, 1.f)) : sg * (b - a);
        const float incl = lse::wave_inclusive_sum(sd);
        // exclusive prefix = the inclusive one shifted by a lane, not `incl - sd`: a fully opaque sample (alpha == 1.0f, which
        // sigma * dt > ~17 rounds to) has sd = +inf, and inf - inf = NaN would cull the sample although its own transmittance
        // prod_{i<k} (1 - alpha_i) is finite (nerfacc's exclusive product keeps it); everything behind it gets T = 0
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0.f;
        const float T = expf(-(excl + carry));
        const float alpha = from_alpha ? sg : 1.f - expf(-sd);
        bool vis = valid && (T >= eps);
        if (alpha_thre > 0.f) vis = vis && (alpha >= alpha_thre);
        if (valid) mask[i] = vis ? 1 : 0;
        kept += __popcll(__ballot(vis));
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0 && new_cnts) new_cnts[ray] = kept;
}

__global__ __launch_bounds__(256) void compact_kernel(const uint8_t *__restrict__ mask, const int64_t *__restrict__ packed,
                                                      const int64_t *__restrict__ new_packed, int n_rays,
                                                      const int32_t *__restrict__ ri, const float *__restrict__ ts,
                                                      const float *__restrict__ te, int32_t *__restrict__ o_ri,
                                                      float *__restrict__ o_ts, float *__restrict__ o_te)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    int64_t dst = new_packed[2 * ray];
    for (int64_t base = 0; base < cnt; base += 64) {
        const int64_t i = s0 + base + lane;
        const bool keep = (base + lane < cnt) && mask[i];
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) {
            o_ri[dst + before] = ri[i];
            o_ts[dst + before] = ts[i];
            o_te[dst + before] = te[i];
        }
        dst += __popcll(bal);
    }
}

// The visibility pre-pass has already encoded every candidate sample (unit-cube position, selector, hash features); the
// survivors of the culling keep theirs instead of being encoded a second time by the main pass: same per-ray ballot /
// popcount compaction as compact_kernel, applied to x01[N,3], selector[N] and the level-major features y[L][N][2]
// (a wave's 64 lanes read 512 contiguous bytes per level).  256 B moved per survivor against 1024 B gathered by a re-encode.
__global__ __launch_bounds__(256) void compact_features_kernel(const uint8_t *__restrict__ mask, const int64_t *__restrict__ packed,
                                                               const int64_t *__restrict__ new_packed, int n_rays,
                                                               const float *__restrict__ x01, const uint8_t *__restrict__ sel,
                                                               const float2 *__restrict__ y, int n_levels, int64_t n_old,
                                                               int64_t n_new, float *__restrict__ o_x01,
                                                               uint8_t *__restrict__ o_sel, float2 *__restrict__ o_y)
{
    // blockIdx.y = level group: one wave per (ray, group of levels).  A wave per RAY alone is 3.4 waves per SIMD at 3510 rays, each
    // walking its ray's candidates 64 at a time with dependent trips to memory -- latency-bound at 3.4 TB/s; with the levels split over
    // gridDim.y groups (every group recomputes the cheap ballot / popcount positions; group 0 also moves x01 and the selector) the
    // same bytes move with gridDim.y times the waves in flight.
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int per = (n_levels + (int)gridDim.y - 1) / (int)gridDim.y;
    const int l_lo = (int)blockIdx.y * per, l_hi = min(n_levels, l_lo + per);
    const bool first = blockIdx.y == 0;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    int64_t dst = new_packed[2 * ray];
    for (int64_t base = 0; base < cnt; base += 64) {
        const int64_t i = s0 + base + lane;
        const bool keep = (base + lane < cnt) && mask[i];
        const unsigned long long bal = __ballot(keep);
        const int64_t o = dst + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) {
            if (first) {
                o_x01[o * 3 + 0] = x01[i * 3 + 0];
                o_x01[o * 3 + 1] = x01[i * 3 + 1];
                o_x01[o * 3 + 2] = x01[i * 3 + 2];
                o_sel[o] = sel[i];
            }
            int l = l_lo;
            for (; l + 4 <= l_hi; l += 4) {        // 4 independent loads in flight, then 4 stores
                float2 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = y[(int64_t)(l + u) * n_old + i];
#pragma unroll
                for (int u = 0; u < 4; ++u) o_y[(int64_t)(l + u) * n_new + o] = v[u];
            }
            for (; l < l_hi; ++l) o_y[(int64_t)l * n_new + o] = y[(int64_t)l * n_old + i];
        }
        dst += __popcll(bal);
    }
}

}  // namespace

extern "C" int lse_compact_features(const uint8_t *mask, const int64_t *packed_info, const int64_t *new_packed_info,
                                    int32_t n_rays, const float *x01, const uint8_t *selector, const float *y,
                                    int32_t n_levels, int64_t n_old, int64_t n_new, float *out_x01,
                                    uint8_t *out_selector, float *out_y, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0 && n_old >= 0 && n_new >= 0 && n_levels >= 1, "lse_compact_features: bad sizes");
    if (n_rays == 0 || n_new == 0) return LSE_OK;
    LSE_REQUIRE(mask && packed_info && new_packed_info && x01 && selector && y && out_x01 && out_selector && out_y,
                "lse_compact_features: null pointer");
    const int groups = (int)std::min<int64_t>(std::max<int64_t>(lse::option("compact_features_groups"), 1), n_levels);
    hipLaunchKernelGGL(compact_features_kernel, dim3((n_rays + 3) / 4, groups), dim3(256), 0, lse::as_stream(stream), mask, packed_info,
                       new_packed_info, n_rays, x01, selector, reinterpret_cast<const float2 *>(y), n_levels, n_old, n_new,
                       out_x01, out_selector, reinterpret_cast<float2 *>(out_y));
    return lse::check_launch("lse_compact_features");
}

extern "C" int lse_volrend_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                               int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, float *weights,
                               float *out_rgb, float *out_acc, float *out_depth_num, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_volrend_fwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights, "lse_volrend_fwd: null pointer");
    LSE_REQUIRE(!rgb || rgb_stride >= 3, "lse_volrend_fwd: rgb_stride < 3");
    hipLaunchKernelGGL(volrend_fwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, rgb, rgb_stride, packed_info, n_rays, weights, out_rgb, out_acc, out_depth_num,
                       (float *)nullptr, (float *)nullptr, (float *)nullptr);
    return lse::check_launch("lse_volrend_fwd");
}

extern "C" int lse_volrend_depth_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                                     int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, float *weights,
                                     float *out_rgb, float *out_acc, float *out_depth_num, float *mid_range /*[R,2]*/,
                                     float *out_depth, float *depth_range /*[2]*/, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_volrend_depth_fwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights && out_acc && out_depth_num && mid_range && out_depth,
                "lse_volrend_depth_fwd: null pointer");
    LSE_REQUIRE(!rgb || rgb_stride >= 3, "lse_volrend_depth_fwd: rgb_stride < 3");
    hipStream_t st = lse::as_stream(stream);
    hipLaunchKernelGGL(volrend_fwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, st, t_starts, t_ends, sigmas, rgb,
                       rgb_stride, packed_info, n_rays, weights, out_rgb, out_acc, out_depth_num, (float *)nullptr,
                       (float *)nullptr, mid_range);
    hipLaunchKernelGGL(depth_finish_kernel, dim3(1), dim3(1024), 0, st, out_depth_num, out_acc, mid_range, n_rays,
                       out_depth, depth_range);
    return lse::check_launch("lse_volrend_depth_fwd");
}

extern "C" int lse_render_weight_fwd(const float *t_starts, const float *t_ends, const float *sigmas,
                                     const int64_t *packed_info, int32_t n_rays, float *weights, float *trans,
                                     float *alphas, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_render_weight_fwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights, "lse_render_weight_fwd: null pointer");
    hipLaunchKernelGGL(volrend_fwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, (const float *)nullptr, 0, packed_info, n_rays, weights, (float *)nullptr, (float *)nullptr,
                       (float *)nullptr, trans, alphas, (float *)nullptr);
    return lse::check_launch("lse_render_weight_fwd");
}

extern "C" int lse_render_weight_bwd(const float *t_starts, const float *t_ends, const float *sigmas,
                                     const int64_t *packed_info, int32_t n_rays, const float *weights,
                                     const float *d_weights, float *d_sigmas, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_render_weight_bwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights && d_weights && d_sigmas,
                "lse_render_weight_bwd: null pointer");
    hipLaunchKernelGGL(volrend_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, (const float *)nullptr, 0, packed_info, n_rays, weights, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, d_sigmas, (float *)nullptr, d_weights);
    return lse::check_launch("lse_render_weight_bwd");
}

extern "C" int lse_volrend_bwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgb,
                               int32_t rgb_stride, const int64_t *packed_info, int32_t n_rays, const float *weights,
                               const float *d_out_rgb, const float *d_out_acc, const float *d_out_depth_num,
                               float *d_sigmas, float *d_rgb, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_volrend_bwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights && d_sigmas, "lse_volrend_bwd: null pointer");
    LSE_REQUIRE(!rgb || rgb_stride >= 3, "lse_volrend_bwd: rgb_stride < 3");
    hipLaunchKernelGGL(volrend_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, rgb, rgb_stride, packed_info, n_rays, weights, d_out_rgb, d_out_acc, d_out_depth_num,
                       d_sigmas, d_rgb, (const float *)nullptr);
    return lse::check_launch("lse_volrend_bwd");
}

extern "C" int lse_visibility_mask(const float *t_starts, const float *t_ends, const float *sigmas,
                                   const int64_t *packed_info, int32_t n_rays, float early_stop_eps, float alpha_thre,
                                   uint8_t *mask, int64_t *new_cnts, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_visibility_mask: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && mask, "lse_visibility_mask: null pointer");
    hipLaunchKernelGGL(visibility_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, packed_info, n_rays, early_stop_eps, alpha_thre, mask, new_cnts, 0, (const float *)nullptr);
    return lse::check_launch("lse_visibility_mask");
}

extern "C" int lse_visibility_mask_cap(const float *t_starts, const float *t_ends, const float *sigmas,
                                       const int64_t *packed_info, int32_t n_rays, float early_stop_eps, float alpha_thre,
                                       const float *alpha_cap, uint8_t *mask, int64_t *new_cnts, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_visibility_mask_cap: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(t_starts && t_ends && sigmas && packed_info && mask && alpha_cap, "lse_visibility_mask_cap: null pointer");
    hipLaunchKernelGGL(visibility_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), t_starts, t_ends,
                       sigmas, packed_info, n_rays, early_stop_eps, alpha_thre, mask, new_cnts, 0, alpha_cap);
    return lse::check_launch("lse_visibility_mask_cap");
}

extern "C" int lse_visibility_mask_alpha(const float *alphas, const int64_t *packed_info, int32_t n_rays,
                                         float early_stop_eps, float alpha_thre, uint8_t *mask, int64_t *new_cnts,
                                         lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_visibility_mask_alpha: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(alphas && packed_info && mask, "lse_visibility_mask_alpha: null pointer");
    hipLaunchKernelGGL(visibility_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream),
                       (const float *)nullptr, (const float *)nullptr, alphas, packed_info, n_rays, early_stop_eps,
                       alpha_thre, mask, new_cnts, 1, (const float *)nullptr);
    return lse::check_launch("lse_visibility_mask_alpha");
}

extern "C" int lse_compact_samples(const uint8_t *mask, const int64_t *packed_info, const int64_t *new_packed_info,
                                   int32_t n_rays, const int32_t *ray_indices, const float *t_starts,
                                   const float *t_ends, int32_t *out_ray_indices, float *out_t_starts,
                                   float *out_t_ends, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_compact_samples: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(mask && packed_info && new_packed_info && ray_indices && t_starts && t_ends && out_ray_indices &&
                    out_t_starts && out_t_ends, "lse_compact_samples: null pointer");
    hipLaunchKernelGGL(compact_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), mask, packed_info,
                       new_packed_info, n_rays, ray_indices, t_starts, t_ends, out_ray_indices, out_t_starts, out_t_ends);
    return lse::check_launch("lse_compact_samples");
}
