// Fused O(R) training epilogue of LSENeRFModel for gfx950: everything between the rendered per-ray radiance and the two
// loss scalars, in ONE forward launch and ONE backward launch.
//   output routing (clamp 1e-5 -> intensity mappers -> ThreeToOne / gray)     R:lse_nerf/lsenerf.py:329-377
//   intensity mappers (identity / gt x^(1/2.4) / powpow x^p, p learnable)     R:lse_nerf/intensity_mappers.py:64-94
//   deblur mean over the 4 virtual cameras of a pixel                         R:lse_nerf/lsenerf.py:365-370
//   rgb MSE and the log-intensity event MSE (log_loss, EPS 1e-6)              R:lse_nerf/lsenerf.py:392-439, R:lse_nerf/utils.py:12
//
// The work is a few thousand rays x a handful of flops, so the design target is launch count, not bandwidth: one
// 1024-thread workgroup walks all rays, reduces the losses (and, in the backward, the three scalar-parameter gradients)
// through LDS in a fixed order -- no atomics, no zero-fills, bitwise reproducible -- where the torch composition issues
// ~40 element-wise launches plus their autograd twins.
#include "common.h"

namespace {

constexpr float kClampMin = 1e-5f;   // torch.clamp(rgb, 1e-5)
constexpr float kLogEps = 1e-6f;     // EPS of R:lse_nerf/utils.py:12
constexpr float kGray[3] = {0.2989f, 0.5870f, 0.1140f};   // to_gray / ToGrayGT

struct EpiArgs {
    lse_epilogue_desc d;
    const float *col_rgb;     // [n_col * group, 3] rendered radiance of the colour bundle (nullable: no colour loss)
    const float *col_gt;      // [n_col, 3]
    const float *prev_rgb;    // [n_ev, 3] (nullable: no event loss)
    const float *next_rgb;    // [n_ev, 3]
    const float *evs_gt;      // [n_ev]
    const float *pow_rgb;     // [1] powpow coefficient of the rgb mapper (mapper kind 3)
    const float *pow_evs;     // [1] powpow coefficient of the event mapper
    const float *w31;         // [3] ThreeToOne raw weights (softmax inside)
    int n_col, n_ev;
    // forward outputs
    float *losses;            // [2] = (rgb_loss, event_loss)
    // backward
    const float *g_rgb_loss, *g_event_loss;   // upstream gradients of the two losses (device scalars; NULL = 0)
    float *d_col, *d_prev, *d_next;     // same shapes as the inputs
    float *d_scalars;         // [5] = d pow_rgb, d pow_evs, d w31[3]   (overwritten)
};

__device__ __forceinline__ float mapper_fwd(int kind, float x, float p)
{
    if (kind == LSE_MAP_GT) return powf(x, 1.0f / 2.4f);
    if (kind == LSE_MAP_POWPOW) return powf(x, p);
    return x;
}
// returns d(mapper)/dx; *dp receives d(mapper)/dp for powpow
__device__ __forceinline__ float mapper_bwd(int kind, float x, float p, float *dp)
{
    *dp = 0.f;
    if (kind == LSE_MAP_GT) return (1.0f / 2.4f) * powf(x, 1.0f / 2.4f - 1.0f);
    if (kind == LSE_MAP_POWPOW) {
        *dp = powf(x, p) * logf(x);
        return p * powf(x, p - 1.0f);
    }
    return 1.f;
}

__device__ __forceinline__ void softmax3(const float *w, float (&s)[3])
{
    const float m = fmaxf(w[0], fmaxf(w[1], w[2]));
    const float e0 = expf(w[0] - m), e1 = expf(w[1] - m), e2 = expf(w[2] - m);
    const float inv = 1.f / (e0 + e1 + e2);
    s[0] = e0 * inv; s[1] = e1 * inv; s[2] = e2 * inv;
}

// Event-side chain of one ray: radiance [3] -> log(intensity + EPS).
//   c = max(rgb, 1e-5);  one_dim: s = sum_k w_k c_k (learned softmax weights or the fixed gray vector) -> m(s);
//   otherwise m(c_k) per channel, then to_gray (R:lse_nerf/lsenerf.py:393-394).
struct EvChain {
    float c[3], mapped[3], s, g;
};
__device__ __forceinline__ float ev_chain_fwd(const lse_epilogue_desc &d, const float *rgb, const float (&w)[3], float p,
                                              EvChain &st)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) st.c[k] = fmaxf(rgb[k], kClampMin);
    if (d.ev_one_dim != LSE_ONE_DIM_NONE) {
        st.s = w[0] * st.c[0] + w[1] * st.c[1] + w[2] * st.c[2];
        st.g = mapper_fwd(d.evs_mapper, st.s, p);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) st.mapped[k] = mapper_fwd(d.evs_mapper, st.c[k], p);
        st.g = kGray[0] * st.mapped[0] + kGray[1] * st.mapped[1] + kGray[2] * st.mapped[2];
    }
    return logf(st.g + kLogEps);
}
// dL = d(loss)/d(log intensity).  Writes d rgb[3]; accumulates dp (mapper coefficient) and dw[3] (one_dim weights).
__device__ __forceinline__ void ev_chain_bwd(const lse_epilogue_desc &d, const float *rgb, const float (&w)[3], float p,
                                             const EvChain &st, float dL, float *d_rgb, float &dp_acc, float (&dw_acc)[3])
{
    const float dg = dL / (st.g + kLogEps);
    float dc[3];
    if (d.ev_one_dim != LSE_ONE_DIM_NONE) {
        float dpm;
        const float ds = dg * mapper_bwd(d.evs_mapper, st.s, p, &dpm);
        dp_acc += dg * dpm;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dc[k] = ds * w[k];
            dw_acc[k] += ds * st.c[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float dpm;
            const float dm = dg * kGray[k];
            dc[k] = dm * mapper_bwd(d.evs_mapper, st.c[k], p, &dpm);
            dp_acc += dm * dpm;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) d_rgb[k] = rgb[k] >= kClampMin ? dc[k] : 0.f;   // torch.clamp(min) backward
}

// block-wide sum of `n_vals` per-thread values, fixed order (deterministic); result valid in every thread
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float *smem /* [N][16] */)
{
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lse::wave_sum(v[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) smem[k * 16 + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += smem[k * 16 + w];
        v[k] = t;
    }
}

__device__ __forceinline__ void one_dim_weights(const EpiArgs &a, float (&w)[3])
{
    if (a.d.ev_one_dim == LSE_ONE_DIM_LEARNED) softmax3(a.w31, w);
    else { w[0] = kGray[0]; w[1] = kGray[1]; w[2] = kGray[2]; }
}

// Colour-side value of one pixel channel group: mean over `group` rays of  mapped ? m(max(rgb,1e-5)) : rgb.
__global__ __launch_bounds__(1024) void epilogue_fwd_kernel(EpiArgs a)
{
    __shared__ float smem[2 * 16];
    const int G = a.d.deblur_group;
    const float p_rgb = (a.d.rgb_mapper == LSE_MAP_POWPOW) ? a.pow_rgb[0] : 1.f;
    const float p_evs = (a.d.evs_mapper == LSE_MAP_POWPOW) ? a.pow_evs[0] : 1.f;
    float acc[2] = {0.f, 0.f};
    if (a.col_rgb) {
        for (int i = threadIdx.x; i < a.n_col * 3; i += blockDim.x) {
            const int px = i / 3, ch = i % 3;
            float m = 0.f;
            for (int g = 0; g < G; ++g) {
                const float x = a.col_rgb[((int64_t)px * G + g) * 3 + ch];
                m += a.d.rgb_mapped ? mapper_fwd(a.d.rgb_mapper, fmaxf(x, kClampMin), p_rgb) : x;
            }
            m = fmaxf(m / (float)G, kClampMin);            // training-mode clamp of the routed rgb
            const float e = m - a.col_gt[i];
            acc[0] += e * e;
        }
    }
    if (a.prev_rgb) {
        float w[3];
        one_dim_weights(a, w);
        for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
            EvChain sp, sn;
            const float lp = ev_chain_fwd(a.d, a.prev_rgb + 3 * (int64_t)r, w, p_evs, sp);
            const float ln = ev_chain_fwd(a.d, a.next_rgb + 3 * (int64_t)r, w, p_evs, sn);
            const float e = (ln - lp) - a.evs_gt[r];
            acc[1] += e * e;
        }
    }
    block_sum<2>(acc, smem);
    if (threadIdx.x == 0) {
        a.losses[0] = a.col_rgb ? acc[0] / (float)(a.n_col * 3) : 0.f;
        a.losses[1] = a.prev_rgb ? a.d.evs_loss_weight * acc[1] / (float)a.n_ev : 0.f;
    }
}

__global__ __launch_bounds__(1024) void epilogue_bwd_kernel(EpiArgs a)
{
    __shared__ float smem[5 * 16];
    const int G = a.d.deblur_group;
    const float p_rgb = (a.d.rgb_mapper == LSE_MAP_POWPOW) ? a.pow_rgb[0] : 1.f;
    const float p_evs = (a.d.evs_mapper == LSE_MAP_POWPOW) ? a.pow_evs[0] : 1.f;
    const float g_rgb = a.g_rgb_loss ? a.g_rgb_loss[0] : 0.f, g_evs = a.g_event_loss ? a.g_event_loss[0] : 0.f;
    float sc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // d pow_rgb, d pow_evs, d(one_dim weights)[3] (before the softmax Jacobian)
    if (a.col_rgb && a.d_col) {
        const float k = g_rgb * 2.f / (float)(a.n_col * 3);
        for (int i = threadIdx.x; i < a.n_col * 3; i += blockDim.x) {
            const int px = i / 3, ch = i % 3;
            float m = 0.f;
            for (int g = 0; g < G; ++g) {
                const float x = a.col_rgb[((int64_t)px * G + g) * 3 + ch];
                m += a.d.rgb_mapped ? mapper_fwd(a.d.rgb_mapper, fmaxf(x, kClampMin), p_rgb) : x;
            }
            m /= (float)G;
            const float dm = (m >= kClampMin) ? k * (fmaxf(m, kClampMin) - a.col_gt[i]) / (float)G : 0.f;
            for (int g = 0; g < G; ++g) {
                const int64_t idx = ((int64_t)px * G + g) * 3 + ch;
                const float x = a.col_rgb[idx];
                float dx = dm;
                if (a.d.rgb_mapped) {
                    float dpm;
                    const float xc = fmaxf(x, kClampMin);
                    dx = dm * mapper_bwd(a.d.rgb_mapper, xc, p_rgb, &dpm);
                    sc[0] += dm * dpm;
                    if (!(x >= kClampMin)) dx = 0.f;
                }
                a.d_col[idx] = dx;
            }
        }
    }
    if (a.prev_rgb && a.d_prev) {
        float w[3], dw[3] = {0.f, 0.f, 0.f};
        one_dim_weights(a, w);
        const float k = g_evs * a.d.evs_loss_weight * 2.f / (float)a.n_ev;
        for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
            EvChain sp, sn;
            const float *rp = a.prev_rgb + 3 * (int64_t)r, *rn = a.next_rgb + 3 * (int64_t)r;
            const float lp = ev_chain_fwd(a.d, rp, w, p_evs, sp);
            const float ln = ev_chain_fwd(a.d, rn, w, p_evs, sn);
            const float dd = k * ((ln - lp) - a.evs_gt[r]);
            ev_chain_bwd(a.d, rn, w, p_evs, sn, dd, a.d_next + 3 * (int64_t)r, sc[1], dw);
            ev_chain_bwd(a.d, rp, w, p_evs, sp, -dd, a.d_prev + 3 * (int64_t)r, sc[1], dw);
        }
        sc[2] = dw[0]; sc[3] = dw[1]; sc[4] = dw[2];
    }
    block_sum<5>(sc, smem);
    if (threadIdx.x == 0 && a.d_scalars) {
        a.d_scalars[0] = sc[0];
        a.d_scalars[1] = sc[1];
        if (a.d.ev_one_dim == LSE_ONE_DIM_LEARNED) {      // softmax Jacobian: d raw_j = s_j (dw_j - sum_k s_k dw_k)
            float s[3];
            softmax3(a.w31, s);
            const float dot = s[0] * sc[2] + s[1] * sc[3] + s[2] * sc[4];
            a.d_scalars[2] = s[0] * (sc[2] - dot);
            a.d_scalars[3] = s[1] * (sc[3] - dot);
            a.d_scalars[4] = s[2] * (sc[4] - dot);
        } else {
            a.d_scalars[2] = a.d_scalars[3] = a.d_scalars[4] = 0.f;
        }
    }
}

int check_desc(const lse_epilogue_desc *d, const char *who)
{
    LSE_REQUIRE(d, "%s: null desc", who);
    LSE_REQUIRE(d->rgb_mapper >= LSE_MAP_IDENTITY && d->rgb_mapper <= LSE_MAP_POWPOW, "%s: bad rgb_mapper %d", who, d->rgb_mapper);
    LSE_REQUIRE(d->evs_mapper >= LSE_MAP_IDENTITY && d->evs_mapper <= LSE_MAP_POWPOW, "%s: bad evs_mapper %d", who, d->evs_mapper);
    LSE_REQUIRE(d->ev_one_dim >= LSE_ONE_DIM_NONE && d->ev_one_dim <= LSE_ONE_DIM_GRAY, "%s: bad ev_one_dim %d", who, d->ev_one_dim);
    LSE_REQUIRE(d->deblur_group >= 1 && d->deblur_group <= 16, "%s: deblur_group %d out of range", who, d->deblur_group);
    return LSE_OK;
}

int fill(EpiArgs &a, const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt, int32_t n_col,
         const float *prev_rgb, const float *next_rgb, const float *evs_gt, int32_t n_ev, const float *pow_rgb,
         const float *pow_evs, const float *w31, const char *who)
{
    int rc = check_desc(desc, who);
    if (rc) return rc;
    LSE_REQUIRE(n_col >= 0 && n_ev >= 0, "%s: negative ray count", who);
    LSE_REQUIRE(!col_rgb || (col_gt && n_col > 0), "%s: colour bundle without target / rays", who);
    LSE_REQUIRE(!prev_rgb || (next_rgb && evs_gt && n_ev > 0), "%s: event bundle needs prev, next and the event target", who);
    LSE_REQUIRE(!(desc->rgb_mapped && desc->rgb_mapper == LSE_MAP_POWPOW) || !col_rgb || pow_rgb, "%s: powpow rgb mapper without coefficient", who);
    LSE_REQUIRE(desc->evs_mapper != LSE_MAP_POWPOW || !prev_rgb || pow_evs, "%s: powpow event mapper without coefficient", who);
    LSE_REQUIRE(desc->ev_one_dim != LSE_ONE_DIM_LEARNED || !prev_rgb || w31, "%s: learned ThreeToOne without weights", who);
    a.d = *desc;
    a.col_rgb = col_rgb; a.col_gt = col_gt; a.prev_rgb = prev_rgb; a.next_rgb = next_rgb; a.evs_gt = evs_gt;
    a.pow_rgb = pow_rgb; a.pow_evs = pow_evs; a.w31 = w31; a.n_col = n_col; a.n_ev = n_ev;
    return LSE_OK;
}

}  // namespace

extern "C" int lse_loss_epilogue_fwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt,
                                     int32_t n_col, const float *prev_rgb, const float *next_rgb, const float *evs_gt,
                                     int32_t n_ev, const float *pow_rgb, const float *pow_evs, const float *w31,
                                     float *losses, lse_stream_t stream)
{
    EpiArgs a{};
    int rc = fill(a, desc, col_rgb, col_gt, n_col, prev_rgb, next_rgb, evs_gt, n_ev, pow_rgb, pow_evs, w31, "lse_loss_epilogue_fwd");
    if (rc) return rc;
    LSE_REQUIRE(losses, "lse_loss_epilogue_fwd: null losses");
    a.losses = losses;
    hipLaunchKernelGGL(epilogue_fwd_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), a);
    return lse::check_launch("lse_loss_epilogue_fwd");
}

extern "C" int lse_loss_epilogue_bwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt,
                                     int32_t n_col, const float *prev_rgb, const float *next_rgb, const float *evs_gt,
                                     int32_t n_ev, const float *pow_rgb, const float *pow_evs, const float *w31,
                                     const float *g_rgb_loss, const float *g_event_loss, float *d_col, float *d_prev,
                                     float *d_next, float *d_scalars, lse_stream_t stream)
{
    EpiArgs a{};
    int rc = fill(a, desc, col_rgb, col_gt, n_col, prev_rgb, next_rgb, evs_gt, n_ev, pow_rgb, pow_evs, w31, "lse_loss_epilogue_bwd");
    if (rc) return rc;
    LSE_REQUIRE(!prev_rgb || !d_prev == !d_next, "lse_loss_epilogue_bwd: d_prev and d_next come together");
    a.g_rgb_loss = g_rgb_loss; a.g_event_loss = g_event_loss; a.d_col = d_col; a.d_prev = d_prev; a.d_next = d_next; a.d_scalars = d_scalars;
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), a);
    return lse::check_launch("lse_loss_epilogue_bwd");
}
