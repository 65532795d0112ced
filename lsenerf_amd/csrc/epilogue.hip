// Fused O(R) training epilogue of LSENeRFModel for gfx950: everything between the rendered per-ray radiance and the two
// loss scalars, in ONE forward launch and ONE backward launch.
//   output routing (clamp 1e-5 -> intensity mappers -> ThreeToOne / gray)     R:lse_nerf/lsenerf.py:329-377
//   intensity mappers (identity / gt x^(1/2.4) / powpow x^p, p learnable)     R:lse_nerf/intensity_mappers.py:64-94
//   MLP intensity mappers ("mlp" 1 -> 16 -> 16 -> 16 -> 1, "rgb_mlp" 3 -> 3)   R:lse_nerf/intensity_mappers.py:28-62
//   deblur mean over the 4 virtual cameras of a pixel                         R:lse_nerf/lsenerf.py:365-370
//   rgb MSE and the log-intensity event MSE (log_loss, EPS 1e-6)              R:lse_nerf/lsenerf.py:392-439, R:lse_nerf/utils.py:12
//   enerf_norm_loss (both sides of the event MSE divided by their 2-norm over the rays)   R:lse_nerf/lsenerf.py:406-419
//
// The work is a few thousand rays x a handful of flops, so the design target is launch count, not bandwidth: one
// 1024-thread workgroup walks all rays, reduces the losses (and, in the backward, the three scalar-parameter gradients)
// through LDS in a fixed order -- no atomics, no zero-fills, bitwise reproducible -- where the torch composition issues
// ~40 element-wise launches plus their autograd twins.
//
// A configuration with an MLP mapper takes the second kernel pair below (epilogue_mlp_*): one thread per pixel / event ray
// evaluates the 4-layer mapper from an LDS image of its weights; in the backward the weight gradients -- outer products
// dz (x) h summed over all rays -- are formed on the f32 matrix core (v_mfma_f32_16x16x4_f32: exact f32 products, 4 rays per
// instruction, the 64 rays of a wave staged through 4 KB of LDS), kept in accumulator registers across the whole launch and
// summed across the waves through LDS in wave order: still one launch each way, still no atomics.
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
    const float *e_thresh;    // [n_ev] per-ray event threshold (enerf_norm_loss only; nullable = 1)
    const float *pow_rgb;     // [1] powpow coefficient of the rgb mapper (mapper kind 3)
    const float *pow_evs;     // [1] powpow coefficient of the event mapper
    const float *w31;         // [3] ThreeToOne raw weights (softmax inside)
    lse_mapper_mlp mlp_rgb;   // parameters (+ gradient destinations) of an MLP mapper on the colour side (kind LSE_MAP_RGB_MLP)
    lse_mapper_mlp mlp_evs;   // ... on the event side (LSE_MAP_MLP behind ev_one_dim, LSE_MAP_RGB_MLP otherwise)
    int n_col, n_ev;
    bool uses_mlp;            // an MLP mapper or enerf_norm_loss is active: the epilogue_mlp_* kernel pair
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

// ---- MLP intensity mappers ---------------------------------------------------------------------------------------------
// nerfstudio MLP(in_dim = IN, num_layers = 4, layer_width = 16, out_dim = IN, ReLU, out_activation = Sigmoid, "torch"):
// four nn.Linear layers with bias (R:lse_nerf/intensity_mappers.py:30-38, :49-57), IN = 1 ("mlp") or 3 ("rgb_mlp").
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMlpImg = 672;          // floats of one weight image in LDS (659 used at IN = 3), a multiple of 4

// LDS image of one mapper: [W0 16xIN | b0 16 | W1 16x16 | b1 16 | W2 16x16 | b2 16 | W3 INx16 | b3 IN], rows as nn.Linear stores them
template <int IN>
struct MlpOff {
    static constexpr int W0 = 0, B0 = 16 * IN, W1 = B0 + 16, B1 = W1 + 256, W2 = B1 + 16, B2 = W2 + 256, W3 = B2 + 16,
                         B3 = W3 + 16 * IN, N = B3 + IN;
};

__device__ __forceinline__ void mlp_stage(const lse_mapper_mlp &m, int in, float *img)   // all threads; the caller synchronises
{
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        const int nw = (l == 0 || l == 3) ? 16 * in : 256, nb = (l == 3) ? in : 16;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) img[off + i] = m.w[l][i];
        off += nw;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) img[off + i] = m.b[l][i];
        off += nb;
    }
}

// The image is loop-invariant for the per-ray loops around an evaluation, and a compiler that notices hoists all 659 loads out of the
// loop (1300 spilled registers).  An offset of zero that the compiler cannot see through keeps the loads where they are used.
__device__ __forceinline__ const float *mlp_fresh(const float *img)
{
    int zero = 0;
    asm volatile("" : "+s"(zero));
    return img + zero;
}

template <int IN>
__device__ __forceinline__ void mlp_fwd(const float *img_in, const float (&x)[IN], float (&h1)[16], float (&h2)[16], float (&h3)[16],
                                        float (&y)[IN])
{
    using O = MlpOff<IN>;
    const float *img = mlp_fresh(img_in);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float z = img[O::B0 + i];
#pragma unroll
        for (int j = 0; j < IN; ++j) z = fmaf(img[O::W0 + i * IN + j], x[j], z);
        h1[i] = fmaxf(z, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float z = img[O::B1 + i];
#pragma unroll
        for (int j = 0; j < 16; ++j) z = fmaf(img[O::W1 + i * 16 + j], h1[j], z);
        h2[i] = fmaxf(z, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float z = img[O::B2 + i];
#pragma unroll
        for (int j = 0; j < 16; ++j) z = fmaf(img[O::W2 + i * 16 + j], h2[j], z);
        h3[i] = fmaxf(z, 0.f);
    }
#pragma unroll
    for (int o = 0; o < IN; ++o) {
        float z = img[O::B3 + o];
#pragma unroll
        for (int j = 0; j < 16; ++j) z = fmaf(img[O::W3 + o * 16 + j], h3[j], z);
        y[o] = 1.f / (1.f + expf(-z));
    }
}
template <int IN>
__device__ __forceinline__ void mlp_eval(const float *img, const float (&x)[IN], float (&y)[IN])
{
    float h1[16], h2[16], h3[16];
    mlp_fwd<IN>(img, x, h1, h2, h3, y);
}

// Weight gradients of one mapper as a wave holds them: per layer one 16x16 accumulator tile (row i = 4 * (lane >> 4) + reg = output
// neuron, column j = lane & 15 = input neuron) and one tile whose every column is the bias gradient.
struct WGradAcc {
    f32x4 w[4], b[4];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int l = 0; l < 4; ++l) w[l] = b[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

// accW[i][j] += sum over the wave's 64 rays of dz[ray][i] * h[ray][j],  accB[i][*] += sum of dz[ray][i].
// v_mfma_f32_16x16x4_f32 contracts 4 rays per instruction: lane (q = lane >> 4, c = lane & 15) supplies A[c][q] = dz[ray 4s + q][c]
// and B[q][c] = h[ray 4s + q][c].  Every lane writes its 16 values as one row of the wave's [64][16] staging tile; the operand of
// step s is then word 64 s + lane of the tile.  LDS operations of one wave execute in issue order, so no barrier is needed.
__device__ __forceinline__ void outer_acc(const float (&dz)[16], const float (&h)[16], float *stage, f32x4 &accW, f32x4 &accB)
{
    const int lane = lse::lane_id();
    f32x4 *row = reinterpret_cast<f32x4 *>(stage + lane * 16);
    float a[16], b[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = f32x4{dz[4 * q], dz[4 * q + 1], dz[4 * q + 2], dz[4 * q + 3]};
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = stage[64 * s + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = f32x4{h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) b[s] = stage[64 * s + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], accW, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], 1.f, accB, 0, 0, 0);
    }
}

// Backward of one mapper evaluation: dx = J^T dy, and the evaluation's share of every weight gradient goes to `acc`.  Called by ALL
// lanes of a wave together (a lane without a ray passes live = false: it contributes zeros).
template <int IN>
__device__ __forceinline__ void mlp_bwd(const float *img_in, const float (&x)[IN], const float (&dy)[IN], bool live, float (&dx)[IN],
                                        float *stage, WGradAcc &acc)
{
    using O = MlpOff<IN>;
    float h1[16], h2[16], h3[16], y[IN];
    mlp_fwd<IN>(img_in, x, h1, h2, h3, y);
    const float *img = mlp_fresh(img_in);
    float dz[16], dn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dz[i] = 0.f;
#pragma unroll
    for (int o = 0; o < IN; ++o) dz[o] = live ? dy[o] * y[o] * (1.f - y[o]) : 0.f;      // sigmoid
    outer_acc(dz, h3, stage, acc.w[3], acc.b[3]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < IN; ++o) s = fmaf(img[O::W3 + o * 16 + j], dz[o], s);
        dn[j] = h3[j] > 0.f ? s : 0.f;                                                   // ReLU
    }
    outer_acc(dn, h2, stage, acc.w[2], acc.b[2]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s = fmaf(img[O::W2 + i * 16 + j], dn[i], s);
        dz[j] = h2[j] > 0.f ? s : 0.f;
    }
    outer_acc(dz, h1, stage, acc.w[1], acc.b[1]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s = fmaf(img[O::W1 + i * 16 + j], dz[i], s);
        dn[j] = h1[j] > 0.f ? s : 0.f;
    }
    float xp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xp[j] = 0.f;
#pragma unroll
    for (int j = 0; j < IN; ++j) xp[j] = x[j];
    outer_acc(dn, xp, stage, acc.w[0], acc.b[0]);
#pragma unroll
    for (int j = 0; j < IN; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s = fmaf(img[O::W0 + i * IN + j], dn[i], s);
        dx[j] = s;
    }
}

// Sum the waves' accumulators in wave order and add the result to the gradient destinations; zero the accumulators.
template <int IN>
__device__ __forceinline__ void wgrad_flush(WGradAcc &acc, const lse_mapper_mlp &m, float *red /* [waves][kMlpImg] */)
{
    using O = MlpOff<IN>;
    const int wave = threadIdx.x >> 6, lane = lse::lane_id(), nw = blockDim.x >> 6;
    float *mine = red + wave * kMlpImg;
    const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = rb + r;
        if (col < IN) mine[O::W0 + i * IN + col] = acc.w[0][r];
        mine[O::W1 + i * 16 + col] = acc.w[1][r];
        mine[O::W2 + i * 16 + col] = acc.w[2][r];
        if (i < IN) mine[O::W3 + i * 16 + col] = acc.w[3][r];
        if (col == 0) {
            mine[O::B0 + i] = acc.b[0][r];
            mine[O::B1 + i] = acc.b[1][r];
            mine[O::B2 + i] = acc.b[2][r];
            if (i < IN) mine[O::B3 + i] = acc.b[3][r];
        }
    }
    acc.zero();
    __syncthreads();
    if (m.dw[0]) {
        for (int p = threadIdx.x; p < O::N; p += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[w * kMlpImg + p];
            float *dst;
            int off;
            if (p < O::B0) { dst = m.dw[0]; off = p - O::W0; }
            else if (p < O::W1) { dst = m.db[0]; off = p - O::B0; }
            else if (p < O::B1) { dst = m.dw[1]; off = p - O::W1; }
            else if (p < O::W2) { dst = m.db[1]; off = p - O::B1; }
            else if (p < O::B2) { dst = m.dw[2]; off = p - O::W2; }
            else if (p < O::W3) { dst = m.db[2]; off = p - O::B2; }
            else if (p < O::B3) { dst = m.dw[3]; off = p - O::W3; }
            else { dst = m.db[3]; off = p - O::B3; }
            dst[off] += t;
        }
    }
    __syncthreads();
}

// Colour-side value of one ray: mapped ? m(max(rgb, 1e-5)) : rgb, all three channels.
__device__ __forceinline__ void col_map_fwd(const lse_epilogue_desc &d, const float (&x)[3], float p, const float *img, float (&y)[3])
{
    if (!d.rgb_mapped) {
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = x[k];
        return;
    }
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = fmaxf(x[k], kClampMin);
    if (d.rgb_mapper == LSE_MAP_RGB_MLP) {
        mlp_eval<3>(img, c, y);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = mapper_fwd(d.rgb_mapper, c[k], p);
    }
}

// Event-side chain of one ray with any mapper kind: radiance [3] -> g (intensity before the log); c / s are kept for the backward.
__device__ __forceinline__ float ev_any_fwd(const lse_epilogue_desc &d, const float (&rgb)[3], const float (&w)[3], float p,
                                            const float *img, float (&c)[3], float &s)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = fmaxf(rgb[k], kClampMin);
    s = 0.f;
    if (d.ev_one_dim != LSE_ONE_DIM_NONE) {
        s = w[0] * c[0] + w[1] * c[1] + w[2] * c[2];
        if (d.evs_mapper == LSE_MAP_MLP) {
            float xi[1] = {s}, yo[1];
            mlp_eval<1>(img, xi, yo);
            return yo[0];
        }
        return mapper_fwd(d.evs_mapper, s, p);
    }
    float y[3];
    if (d.evs_mapper == LSE_MAP_RGB_MLP) {
        mlp_eval<3>(img, c, y);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = mapper_fwd(d.evs_mapper, c[k], p);
    }
    return kGray[0] * y[0] + kGray[1] * y[1] + kGray[2] * y[2];
}

// d_r = log(intensity_next + EPS) - log(intensity_prev + EPS) of event ray r, and its target c_r (evs_gt, over the event threshold
// for enerf_norm_loss: R:lse_nerf/lsenerf.py:416)
__device__ __forceinline__ float ev_delta(const EpiArgs &a, int r, const float (&w)[3], float p, const float *img)
{
    const float *pp = a.prev_rgb + 3 * (int64_t)r, *pn = a.next_rgb + 3 * (int64_t)r;
    const float xp[3] = {pp[0], pp[1], pp[2]}, xn[3] = {pn[0], pn[1], pn[2]};
    float c[3], s;
    const float lp = logf(ev_any_fwd(a.d, xp, w, p, img, c, s) + kLogEps);
    const float ln = logf(ev_any_fwd(a.d, xn, w, p, img, c, s) + kLogEps);
    return ln - lp;
}
__device__ __forceinline__ float ev_target(const EpiArgs &a, int r)
{
    if (a.d.event_loss_kind == LSE_EVLOSS_ENERF_NORM) return a.evs_gt[r] / (a.e_thresh ? a.e_thresh[r] : 1.f);
    return a.evs_gt[r];
}

__global__ __launch_bounds__(1024) void epilogue_mlp_fwd_kernel(EpiArgs a)
{
    __shared__ float smem[2 * 16];
    __shared__ __attribute__((aligned(16))) float img_rgb[kMlpImg];
    __shared__ __attribute__((aligned(16))) float img_evs[kMlpImg];
    const int G = a.d.deblur_group;
    const bool rgb_mlp = a.col_rgb && a.d.rgb_mapped && a.d.rgb_mapper == LSE_MAP_RGB_MLP;
    const bool evs_mlp = a.prev_rgb && a.d.evs_mapper >= LSE_MAP_MLP;
    if (rgb_mlp) mlp_stage(a.mlp_rgb, 3, img_rgb);
    if (evs_mlp) mlp_stage(a.mlp_evs, a.d.evs_mapper == LSE_MAP_MLP ? 1 : 3, img_evs);
    __syncthreads();
    const float p_rgb = (a.d.rgb_mapper == LSE_MAP_POWPOW) ? a.pow_rgb[0] : 1.f;
    const float p_evs = (a.d.evs_mapper == LSE_MAP_POWPOW) ? a.pow_evs[0] : 1.f;
    float acc[2] = {0.f, 0.f};
    if (a.col_rgb) {
        for (int px = threadIdx.x; px < a.n_col; px += blockDim.x) {
            float m[3] = {0.f, 0.f, 0.f};
            for (int g = 0; g < G; ++g) {
                const float *src = a.col_rgb + ((int64_t)px * G + g) * 3;
                const float x[3] = {src[0], src[1], src[2]};
                float y[3];
                col_map_fwd(a.d, x, p_rgb, img_rgb, y);
#pragma unroll
                for (int k = 0; k < 3; ++k) m[k] += y[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = fmaxf(m[k] / (float)G, kClampMin) - a.col_gt[(int64_t)px * 3 + k];
                acc[0] += e * e;
            }
        }
    }
    if (a.prev_rgb) {
        float w[3];
        one_dim_weights(a, w);
        float inv_nd = 1.f, inv_ne = 1.f;
        if (a.d.event_loss_kind == LSE_EVLOSS_ENERF_NORM) {
            float s2[2] = {0.f, 0.f};
            for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
                const float dl = ev_delta(a, r, w, p_evs, img_evs), ce = ev_target(a, r);
                s2[0] += dl * dl;
                s2[1] += ce * ce;
            }
            block_sum<2>(s2, smem);
            inv_nd = 1.f / (sqrtf(s2[0]) + kLogEps);
            inv_ne = 1.f / (sqrtf(s2[1]) + kLogEps);
        }
        for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
            const float e = ev_delta(a, r, w, p_evs, img_evs) * inv_nd - ev_target(a, r) * inv_ne;
            acc[1] += e * e;
        }
    }
    block_sum<2>(acc, smem);
    if (threadIdx.x == 0) {
        a.losses[0] = a.col_rgb ? acc[0] / (float)(a.n_col * 3) : 0.f;
        a.losses[1] = a.prev_rgb ? a.d.evs_loss_weight * acc[1] / (float)a.n_ev : 0.f;
    }
}

// Backward of one event-side chain (any mapper kind); every lane of the wave calls it (live = false: no ray, contributes zeros).
__device__ __forceinline__ void ev_any_bwd(const lse_epilogue_desc &d, const float (&rgb)[3], const float (&w)[3], float p,
                                           const float *img, float dL, bool live, float (&d_rgb)[3], float &dp_acc,
                                           float (&dw_acc)[3], float *stage, WGradAcc &acc)
{
    float c[3], s;
    const float g = ev_any_fwd(d, rgb, w, p, img, c, s);
    const float dg = live ? dL / (g + kLogEps) : 0.f;
    float dc[3];
    if (d.ev_one_dim != LSE_ONE_DIM_NONE) {
        float ds;
        if (d.evs_mapper == LSE_MAP_MLP) {
            const float xi[1] = {s}, dyo[1] = {dg};
            float dxi[1];
            mlp_bwd<1>(img, xi, dyo, live, dxi, stage, acc);
            ds = dxi[0];
        } else {
            float dpm;
            ds = dg * mapper_bwd(d.evs_mapper, s, p, &dpm);
            dp_acc += dg * dpm;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dc[k] = ds * w[k];
            dw_acc[k] += ds * c[k];
        }
    } else if (d.evs_mapper == LSE_MAP_RGB_MLP) {
        const float dy[3] = {dg * kGray[0], dg * kGray[1], dg * kGray[2]};
        mlp_bwd<3>(img, c, dy, live, dc, stage, acc);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float dpm;
            const float dm = dg * kGray[k];
            dc[k] = dm * mapper_bwd(d.evs_mapper, c[k], p, &dpm);
            dp_acc += dm * dpm;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) d_rgb[k] = (live && rgb[k] >= kClampMin) ? dc[k] : 0.f;
}

constexpr int kMlpBwdThreads = 512;      // 8 waves: 256 registers per lane for the recomputed activations + operand staging

__global__ __launch_bounds__(kMlpBwdThreads) void epilogue_mlp_bwd_kernel(EpiArgs a)
{
    __shared__ float smem[5 * 16];
    __shared__ __attribute__((aligned(16))) float img_rgb[kMlpImg];
    __shared__ __attribute__((aligned(16))) float img_evs[kMlpImg];
    __shared__ __attribute__((aligned(16))) float stage_all[(kMlpBwdThreads / 64) * 64 * 16];
    __shared__ float red[(kMlpBwdThreads / 64) * kMlpImg];
    const int G = a.d.deblur_group;
    const bool rgb_mlp = a.col_rgb && a.d.rgb_mapped && a.d.rgb_mapper == LSE_MAP_RGB_MLP;
    const bool evs_mlp = a.prev_rgb && a.d.evs_mapper >= LSE_MAP_MLP;
    if (rgb_mlp) mlp_stage(a.mlp_rgb, 3, img_rgb);
    if (evs_mlp) mlp_stage(a.mlp_evs, a.d.evs_mapper == LSE_MAP_MLP ? 1 : 3, img_evs);
    __syncthreads();
    float *stage = stage_all + (threadIdx.x >> 6) * 64 * 16;
    const float p_rgb = (a.d.rgb_mapper == LSE_MAP_POWPOW) ? a.pow_rgb[0] : 1.f;
    const float p_evs = (a.d.evs_mapper == LSE_MAP_POWPOW) ? a.pow_evs[0] : 1.f;
    const float g_rgb = a.g_rgb_loss ? a.g_rgb_loss[0] : 0.f, g_evs = a.g_event_loss ? a.g_event_loss[0] : 0.f;
    float sc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    WGradAcc acc;
    acc.zero();
    if (a.col_rgb && a.d_col) {
        const float k = g_rgb * 2.f / (float)(a.n_col * 3);
        for (int base = 0; base < a.n_col; base += blockDim.x) {       // uniform trip count: mlp_bwd is a wave-wide operation
            const int px = base + threadIdx.x;
            const bool live = px < a.n_col;
            float m[3] = {0.f, 0.f, 0.f}, dm[3] = {0.f, 0.f, 0.f};
            if (live) {
                for (int g = 0; g < G; ++g) {
                    const float *src = a.col_rgb + ((int64_t)px * G + g) * 3;
                    const float x[3] = {src[0], src[1], src[2]};
                    float y[3];
                    col_map_fwd(a.d, x, p_rgb, img_rgb, y);
#pragma unroll
                    for (int c = 0; c < 3; ++c) m[c] += y[c];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float mm = m[c] / (float)G;
                    dm[c] = (mm >= kClampMin) ? k * (fmaxf(mm, kClampMin) - a.col_gt[(int64_t)px * 3 + c]) / (float)G : 0.f;
                }
            }
            for (int g = 0; g < G; ++g) {
                const int64_t idx = ((int64_t)px * G + g) * 3;
                float x[3] = {0.5f, 0.5f, 0.5f}, dx[3];
                if (live) { x[0] = a.col_rgb[idx]; x[1] = a.col_rgb[idx + 1]; x[2] = a.col_rgb[idx + 2]; }
                if (!a.d.rgb_mapped) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) dx[c] = dm[c];
                } else if (rgb_mlp) {
                    float xc[3], dc[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) xc[c] = fmaxf(x[c], kClampMin);
                    mlp_bwd<3>(img_rgb, xc, dm, live, dc, stage, acc);
#pragma unroll
                    for (int c = 0; c < 3; ++c) dx[c] = x[c] >= kClampMin ? dc[c] : 0.f;
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float dpm;
                        const float xc = fmaxf(x[c], kClampMin);
                        dx[c] = dm[c] * mapper_bwd(a.d.rgb_mapper, xc, p_rgb, &dpm);
                        sc[0] += dm[c] * dpm;
                        if (!(x[c] >= kClampMin)) dx[c] = 0.f;
                    }
                }
                if (live) { a.d_col[idx] = dx[0]; a.d_col[idx + 1] = dx[1]; a.d_col[idx + 2] = dx[2]; }
            }
        }
        if (rgb_mlp) wgrad_flush<3>(acc, a.mlp_rgb, red);
    }
    if (a.prev_rgb && a.d_prev) {
        float w[3], dw[3] = {0.f, 0.f, 0.f};
        one_dim_weights(a, w);
        const float k = g_evs * a.d.evs_loss_weight * 2.f / (float)a.n_ev;
        // enerf_norm_loss: e_r = d_r / nd - c_r / ne with nd = ||d|| + EPS;  dL/dd_k = k (e_k / nd - T d_k / (nd^2 ||d||)),  T = sum_r e_r d_r
        float inv_nd = 1.f, inv_ne = 1.f, t_over = 0.f;
        const bool enerf = a.d.event_loss_kind == LSE_EVLOSS_ENERF_NORM;
        if (enerf) {
            float s2[2] = {0.f, 0.f};
            for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
                const float dl = ev_delta(a, r, w, p_evs, img_evs), ce = ev_target(a, r);
                s2[0] += dl * dl;
                s2[1] += ce * ce;
            }
            block_sum<2>(s2, smem);
            const float norm = sqrtf(s2[0]);
            inv_nd = 1.f / (norm + kLogEps);
            inv_ne = 1.f / (sqrtf(s2[1]) + kLogEps);
            float t[1] = {0.f};
            for (int r = threadIdx.x; r < a.n_ev; r += blockDim.x) {
                const float dl = ev_delta(a, r, w, p_evs, img_evs);
                t[0] += (dl * inv_nd - ev_target(a, r) * inv_ne) * dl;
            }
            block_sum<1>(t, smem);
            t_over = norm > 0.f ? t[0] * inv_nd * inv_nd / norm : 0.f;      // (torch: the norm's subgradient at 0 is 0)
        }
        for (int base = 0; base < a.n_ev; base += blockDim.x) {
            const int r = base + threadIdx.x;
            const bool live = r < a.n_ev;
            float xp[3] = {0.5f, 0.5f, 0.5f}, xn[3] = {0.5f, 0.5f, 0.5f};
            float dd = 0.f;
            if (live) {
                const float *pp = a.prev_rgb + 3 * (int64_t)r, *pn = a.next_rgb + 3 * (int64_t)r;
                xp[0] = pp[0]; xp[1] = pp[1]; xp[2] = pp[2];
                xn[0] = pn[0]; xn[1] = pn[1]; xn[2] = pn[2];
                const float dl = ev_delta(a, r, w, p_evs, img_evs);
                dd = enerf ? k * ((dl * inv_nd - ev_target(a, r) * inv_ne) * inv_nd - t_over * dl) : k * (dl - a.evs_gt[r]);
            }
            float dn[3], dp[3];
            ev_any_bwd(a.d, xn, w, p_evs, img_evs, dd, live, dn, sc[1], dw, stage, acc);
            ev_any_bwd(a.d, xp, w, p_evs, img_evs, -dd, live, dp, sc[1], dw, stage, acc);
            if (live) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a.d_next[3 * (int64_t)r + c] = dn[c];
                    a.d_prev[3 * (int64_t)r + c] = dp[c];
                }
            }
        }
        sc[2] = dw[0]; sc[3] = dw[1]; sc[4] = dw[2];
        if (evs_mlp) {
            if (a.d.evs_mapper == LSE_MAP_MLP) wgrad_flush<1>(acc, a.mlp_evs, red);
            else wgrad_flush<3>(acc, a.mlp_evs, red);
        }
    }
    block_sum<5>(sc, smem);
    if (threadIdx.x == 0 && a.d_scalars) {
        a.d_scalars[0] = sc[0];
        a.d_scalars[1] = sc[1];
        if (a.d.ev_one_dim == LSE_ONE_DIM_LEARNED) {
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
    LSE_REQUIRE(d->rgb_mapper >= LSE_MAP_IDENTITY && d->rgb_mapper <= LSE_MAP_RGB_MLP, "%s: bad rgb_mapper %d", who, d->rgb_mapper);
    LSE_REQUIRE(d->evs_mapper >= LSE_MAP_IDENTITY && d->evs_mapper <= LSE_MAP_RGB_MLP, "%s: bad evs_mapper %d", who, d->evs_mapper);
    LSE_REQUIRE(d->ev_one_dim >= LSE_ONE_DIM_NONE && d->ev_one_dim <= LSE_ONE_DIM_GRAY, "%s: bad ev_one_dim %d", who, d->ev_one_dim);
    // nn.Linear(1, 16) takes one channel, nn.Linear(3, 16) three (R:lse_nerf/intensity_mappers.py:31, :50): the reference raises otherwise
    LSE_REQUIRE(!(d->rgb_mapped && d->rgb_mapper == LSE_MAP_MLP), "%s: the one-channel \"mlp\" mapper cannot map the three colour channels", who);
    LSE_REQUIRE(d->evs_mapper != LSE_MAP_MLP || d->ev_one_dim != LSE_ONE_DIM_NONE, "%s: the one-channel \"mlp\" event mapper needs ev_one_dim", who);
    LSE_REQUIRE(d->evs_mapper != LSE_MAP_RGB_MLP || d->ev_one_dim == LSE_ONE_DIM_NONE, "%s: the three-channel \"rgb_mlp\" event mapper excludes ev_one_dim", who);
    LSE_REQUIRE(d->deblur_group >= 1 && d->deblur_group <= 16, "%s: deblur_group %d out of range", who, d->deblur_group);
    LSE_REQUIRE(d->event_loss_kind == LSE_EVLOSS_LOG || d->event_loss_kind == LSE_EVLOSS_ENERF_NORM, "%s: bad event_loss_kind %d", who, d->event_loss_kind);
    return LSE_OK;
}

int fill(EpiArgs &a, const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt, int32_t n_col,
         const float *prev_rgb, const float *next_rgb, const float *evs_gt, const float *e_thresh, int32_t n_ev,
         const float *pow_rgb, const float *pow_evs, const float *w31, const lse_mapper_mlp *mlp_rgb, const lse_mapper_mlp *mlp_evs,
         bool backward, const char *who)
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
    a.col_rgb = col_rgb; a.col_gt = col_gt; a.prev_rgb = prev_rgb; a.next_rgb = next_rgb; a.evs_gt = evs_gt; a.e_thresh = e_thresh;
    a.pow_rgb = pow_rgb; a.pow_evs = pow_evs; a.w31 = w31; a.n_col = n_col; a.n_ev = n_ev;
    const bool use_rgb = col_rgb && desc->rgb_mapped && desc->rgb_mapper == LSE_MAP_RGB_MLP;
    const bool use_evs = prev_rgb && desc->evs_mapper >= LSE_MAP_MLP;
    const struct { bool used; const lse_mapper_mlp *src; lse_mapper_mlp *dst; const char *side; } sides[2] = {
        {use_rgb, mlp_rgb, &a.mlp_rgb, "rgb"}, {use_evs, mlp_evs, &a.mlp_evs, "event"}};
    for (const auto &sd : sides) {
        if (!sd.used) continue;
        LSE_REQUIRE(sd.src, "%s: MLP %s mapper without its parameters (lse_mapper_mlp)", who, sd.side);
        int n_grad = 0;
        for (int l = 0; l < 4; ++l) {
            LSE_REQUIRE(sd.src->w[l] && sd.src->b[l], "%s: MLP %s mapper: null weight / bias of layer %d", who, sd.side, l);
            n_grad += (sd.src->dw[l] != nullptr) + (sd.src->db[l] != nullptr);
        }
        LSE_REQUIRE(!backward || n_grad == 0 || n_grad == 8, "%s: MLP %s mapper: gradient destinations come all eight or none", who, sd.side);
        *sd.dst = *sd.src;
    }
    // the second kernel pair serves every mapper kind and both event losses; the first one the closed-form mappers with log_loss
    a.uses_mlp = use_rgb || use_evs || (prev_rgb && desc->event_loss_kind == LSE_EVLOSS_ENERF_NORM);
    return LSE_OK;
}

}  // namespace

extern "C" int lse_loss_epilogue_fwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt,
                                     int32_t n_col, const float *prev_rgb, const float *next_rgb, const float *evs_gt,
                                     const float *e_thresh, int32_t n_ev, const float *pow_rgb, const float *pow_evs, const float *w31,
                                     const lse_mapper_mlp *mlp_rgb, const lse_mapper_mlp *mlp_evs, float *losses,
                                     lse_stream_t stream)
{
    EpiArgs a{};
    int rc = fill(a, desc, col_rgb, col_gt, n_col, prev_rgb, next_rgb, evs_gt, e_thresh, n_ev, pow_rgb, pow_evs, w31, mlp_rgb, mlp_evs, false,
                  "lse_loss_epilogue_fwd");
    if (rc) return rc;
    LSE_REQUIRE(losses, "lse_loss_epilogue_fwd: null losses");
    a.losses = losses;
    if (a.uses_mlp) hipLaunchKernelGGL(epilogue_mlp_fwd_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), a);
    else hipLaunchKernelGGL(epilogue_fwd_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), a);
    return lse::check_launch("lse_loss_epilogue_fwd");
}

extern "C" int lse_loss_epilogue_bwd(const lse_epilogue_desc *desc, const float *col_rgb, const float *col_gt,
                                     int32_t n_col, const float *prev_rgb, const float *next_rgb, const float *evs_gt,
                                     const float *e_thresh, int32_t n_ev, const float *pow_rgb, const float *pow_evs, const float *w31,
                                     const lse_mapper_mlp *mlp_rgb, const lse_mapper_mlp *mlp_evs, const float *g_rgb_loss,
                                     const float *g_event_loss, float *d_col, float *d_prev, float *d_next, float *d_scalars,
                                     lse_stream_t stream)
{
    EpiArgs a{};
    int rc = fill(a, desc, col_rgb, col_gt, n_col, prev_rgb, next_rgb, evs_gt, e_thresh, n_ev, pow_rgb, pow_evs, w31, mlp_rgb, mlp_evs, true,
                  "lse_loss_epilogue_bwd");
    if (rc) return rc;
    LSE_REQUIRE(!prev_rgb || !d_prev == !d_next, "lse_loss_epilogue_bwd: d_prev and d_next come together");
    a.g_rgb_loss = g_rgb_loss; a.g_event_loss = g_event_loss; a.d_col = d_col; a.d_prev = d_prev; a.d_next = d_next; a.d_scalars = d_scalars;
    if (a.uses_mlp) hipLaunchKernelGGL(epilogue_mlp_bwd_kernel, dim3(1), dim3(kMlpBwdThreads), 0, lse::as_stream(stream), a);
    else hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(1), dim3(1024), 0, lse::as_stream(stream), a);
    return lse::check_launch("lse_loss_epilogue_bwd");
}
