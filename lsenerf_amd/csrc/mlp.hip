// Fused small MLPs on the gfx950 f32 matrix cores (v_mfma_f32_16x16x4_f32) -- the kernels behind
// `tcnn.Network` as nerfstudio's MLP configures it at R:lse_nerf/lse_field.py:199-207 (mlp_base_mlp) and
// :254-262 (mlp_head): bias-free, ReLU hidden layers, optional sigmoid output, tcnn parameter layout.
//
// Orientation.  Every layer is computed transposed, H^T = W * X^T, so that the 64 SAMPLES of a wave tile sit on
// the MFMA column (lane) dimension and the NEURONS sit on the accumulator registers:
//     C/D of 16x16x4:  col = lane & 15 (sample),  row = 4*(lane>>4) + reg (neuron within a 16-row block).
// A following layer needs B[k][col] = H[sample col][neuron k] with lane (col, q) supplying one k per k-step.
// Choosing the k order  kidx(ks, q) = 16*(ks>>2) + 4*q + (ks&3)  makes that operand exactly register (ks&3) of
// row block (ks>>2) of the previous accumulator: layers chain in registers with no LDS traffic and no
// cross-lane moves.  The weight (A) operands are pre-permuted to the same k order when the workgroup stages them
// into LDS ("A images": 64 floats per (row block, k-step), read conflict-free with one ds_read_b32 per MFMA
// group and reused for the 4 column tiles of the wave).
//
// fp32 MFMA runs at the fp32 VALU rate (157 TF peak): the win is register/issue economy and exact f32
// (bitwise a k-ordered fmaf chain), which is what the reference's float32 tcnn build computes.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LSE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// k order of a 16-column block read as float4 per lane / of a chained accumulator
__device__ __host__ __forceinline__ constexpr int kidx_blk(int ks, int q) { return 16 * (ks >> 2) + 4 * q + (ks & 3); }
// k order of level-major hash features read as float2 per lane: lane q of k-step pair m reads level 4m+q
__device__ __host__ __forceinline__ constexpr int kidx_hash(int ks, int q) { return 8 * (ks >> 1) + 2 * q + (ks & 1); }

template <int INL>
__device__ __forceinline__ constexpr int kidx_in(int ks, int q)
{
    return INL == LSE_IN_LEVELMAJOR ? kidx_hash(ks, q) : kidx_blk(ks, q);
}

struct MlpArgs {
    const float *params;
    const float *in;
    const float *row_bias;
    const int32_t *row_bias_idx;
    float *out;
    float *act;
    int64_t n;                   // samples (a capacity when n_dev is set: every kernel clamps it first)
    const int64_t *n_dev;        // device-side sample count (nullable)
    int64_t n_stride;            // the host-side n: level stride of level-major inputs / gradients, layer stride of row-major activations
    int out_activation;
    // backward
    const float *d_out;
    float *d_out_pre;   // nullable
    float *d_act;       // nullable: all layers' pre-activation gradients (only the unfused wgrad path needs them)
    float *d_act0;      // nullable: layer-0 pre-activation gradients only (row-bias gradient)
    float *d_in;        // nullable
    float *d_params;    // nullable: fused weight gradients accumulate here
    float *d_row_bias;  // nullable: gradient of row_bias, accumulated per row (needs row_bias_idx, rows contiguous)
    // fused density head (base MLP): sigma = density_scale * exp(out[:,0]) * selector
    float *sigma_out;            // fwd, nullable
    const uint8_t *selector;     // nullable (all in-bounds)
    float density_scale;
    const float *d_sigma;        // bwd, nullable: adds d_sigma * d(sigma)/d(out[:,0]) to the output gradient
    int out_cols;                // 16 (padded row) or 4 (compact: only outputs 0..3 are stored / have gradients)
    int act_tiled;               // saved-activation layout: 0 = row-major [N][W], 1 = tile-major (see act_offset)
    int64_t act_layer_stride;    // floats between the activation arrays of consecutive hidden layers
    // first-layer view into `params` (and `d_params`): W0[row][c] = params[row * w0_ld + w0_col + c], c < n_in; the remaining
    // layers start at params + rest_off.  w0_mask0: input column 0 carries no weight (reads as 0, receives no gradient).
    // Lets the head consume the density logit + 15 geometry features h[N,16] against columns 15..30 of tcnn's [W x 64]
    // input matrix in place -- no split / concatenated copy of the parameters per step.
    int w0_ld, w0_col, w0_mask0;
    int arith;                   // lse_mlp_desc.arith (host-side dispatch only)
    int64_t rest_off;
};

// CT = column tiles (of 16 samples) per wave iteration, NW = waves per workgroup (template parameters below).
// (CT=4,NW=4): 64-sample tiles, fewest LDS weight reads per MFMA.  (CT=2,NW=8): half the registers -> 4 waves/SIMD,
// which hides the first-touch HBM latency of the activation loads (the kernels are latency-, not MFMA-issue-bound).

// Saved activations.  Row-major [N][W] is what the unfused lse_mlp_wgrad reads.  Tile-major stores every 16-sample x
// 16-neuron block exactly in accumulator lane order ([N/16][W/16][64 lanes][4]): one store / load wave-instruction
// moves 1 KiB contiguous instead of 16 pieces of 64 B at a 256-B stride (the forward's stores were issue-bound).
template <int WIDTH>
__device__ __forceinline__ int64_t act_offset(bool tiled, int64_t s, int rb, int q)
{
    return tiled ? ((((s >> 4) * (WIDTH / 16) + rb) * 64 + (q * 16 + (int)(s & 15))) << 2) : s * WIDTH + 16 * rb + 4 * q;
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
template <int KIN, int WIDTH, int NHL, int INL, int CT, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 2 : 1)) void mlp_fwd_kernel(MlpArgs a)
{
    constexpr int HB = WIDTH / 16, KS0 = KIN / 4, KSH = WIDTH / 4;
    constexpr int IMG0 = HB * KS0, IMGH = (NHL == 2) ? HB * KSH : 0, IMGO = KSH;
    extern __shared__ float lds[];
    float *img0 = lds, *imgH = lds + IMG0 * 64, *imgO = imgH + IMGH * 64;

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    for (int e = threadIdx.x; e < IMG0 * 64; e += 64 * NW) {
        const int img = e >> 6, ln = e & 63, rb = img / KS0, ks = img % KS0, i = ln & 15, q = ln >> 4;
        const int c = kidx_in<INL>(ks, q);
        img0[e] = (a.w0_mask0 && c == 0) ? 0.f : W0[(16 * rb + i) * a.w0_ld + c];
    }
    if (NHL == 2)
        for (int e = threadIdx.x; e < IMGH * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            imgH[e] = W1[(16 * rb + i) * WIDTH + kidx_blk(ks, q)];
        }
    for (int e = threadIdx.x; e < IMGO * 64; e += 64 * NW) {
        const int ks = e >> 6, ln = e & 63, i = ln & 15, q = ln >> 4;
        imgO[e] = Wo[i * WIDTH + kidx_blk(ks, q)];
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    a.n = lse::clamp_count(a.n, a.n_dev);
    const int64_t n = a.n, ns = a.n_stride;
    constexpr int TS = 16 * CT;   // samples per wave tile
    const int64_t n_tiles = (n + TS - 1) / TS;
    const int64_t tile_stride = (int64_t)gridDim.x * NW;

    // Inputs of a tile: layer-0 B operands breg[ct][ks] = in[sample][kidx_in(ks, q)] and the per-row layer-0 bias.
    // They are fetched ONE TILE AHEAD: every wave of the chip walks its tiles in lockstep, so without the prefetch each
    // tile's input loads queue behind a chip-wide burst of activation stores and compute + store time simply add up
    // (measured: head 0.52 + 0.39 ms, base 0.29 + 0.27 ms).  Issued before the current tile's stores, the loads return
    // early and the stores drain under the next tile's MFMAs.
    auto load_inputs = [&](int64_t tile, int64_t (&s)[CT], bool (&valid)[CT], float (&breg)[CT][KS0], f32x4 (&bias)[HB][CT]) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int64_t si = tile * TS + ct * 16 + j;
            valid[ct] = si < n;
            s[ct] = valid[ct] ? si : n - 1;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (INL == LSE_IN_LEVELMAJOR) {
#pragma unroll
                for (int m = 0; m < KIN / 8; ++m) {
                    const float2 v = reinterpret_cast<const float2 *>(a.in)[(int64_t)(4 * m + q) * ns + s[ct]];
                    breg[ct][2 * m] = v.x;
                    breg[ct][2 * m + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int b = 0; b < KIN / 16; ++b) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(a.in + s[ct] * KIN + 16 * b + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) breg[ct][4 * b + r] = v[r];
                }
            }
        }
        if (a.row_bias) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int64_t row = a.row_bias_idx ? (int64_t)a.row_bias_idx[s[ct]] : s[ct];
#pragma unroll
                for (int rb = 0; rb < HB; ++rb)
                    bias[rb][ct] = *reinterpret_cast<const f32x4 *>(a.row_bias + row * WIDTH + 16 * rb + 4 * q);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bias[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    int64_t s_nx[CT];
    bool valid_nx[CT];
    float breg_nx[CT][KS0];
    f32x4 bias_nx[HB][CT];
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    if (tile < n_tiles) load_inputs(tile, s_nx, valid_nx, breg_nx, bias_nx);
    for (; tile < n_tiles; tile += tile_stride) {
        __builtin_amdgcn_iglp_opt(0);   // scheduler hint: interleave LDS reads with the MFMA stream (forward: 0.72 -> 0.66 ms)
        int64_t s[CT];
        bool valid[CT];
        float breg[CT][KS0];
        f32x4 h[HB][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            s[ct] = s_nx[ct];
            valid[ct] = valid_nx[ct];
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) breg[ct][ks] = breg_nx[ct][ks];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) h[rb][ct] = bias_nx[rb][ct];
        }
        if (tile + tile_stride < n_tiles) load_inputs(tile + tile_stride, s_nx, valid_nx, breg_nx, bias_nx);
        // ---- layer 0
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) {
                const float aw = img0[(rb * KS0 + ks) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h[rb][ct] = LSE_MFMA(aw, breg[ct][ks], h[rb][ct]);
            }
        }
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[rb][ct][r] = fmaxf(h[rb][ct][r], 0.f);
                if (a.act && valid[ct])
                    *reinterpret_cast<f32x4 *>(a.act + act_offset<WIDTH>(a.act_tiled, s[ct], rb, q)) = h[rb][ct];
            }
        // ---- hidden layer (width x width)
        if (NHL == 2) {
            f32x4 h2[HB][CT];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h2[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb) {
                        const float aw = imgH[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h2[rb][ct] = LSE_MFMA(aw, h[bp][ct][r], h2[rb][ct]);
                    }
                }
            float *act1 = a.act ? a.act + a.act_layer_stride : nullptr;
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[rb][ct][r] = fmaxf(h2[rb][ct][r], 0.f);
                    if (act1 && valid[ct])
                        *reinterpret_cast<f32x4 *>(act1 + act_offset<WIDTH>(a.act_tiled, s[ct], rb, q)) = h[rb][ct];
                }
        }
        // ---- output layer (16 x width)
        f32x4 o[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bp = 0; bp < HB; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float aw = imgO[(4 * bp + r) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) o[ct] = LSE_MFMA(aw, h[bp][ct][r], o[ct]);
            }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (a.out_activation == LSE_ACT_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[ct][r] = 1.f / (1.f + __expf(-o[ct][r]));
            }
            if (valid[ct]) {
                if (a.out_cols == 16) *reinterpret_cast<f32x4 *>(a.out + s[ct] * 16 + 4 * q) = o[ct];
                else if (q == 0) *reinterpret_cast<f32x4 *>(a.out + s[ct] * 4) = o[ct];
                if (a.sigma_out && q == 0) {   // trunc_exp density head on output 0 (R:lse_nerf/lse_field.py:286-287)
                    const bool in_bounds = a.selector == nullptr || a.selector[s[ct]] != 0;
                    a.sigma_out[s[ct]] = in_bounds ? a.density_scale * expf(o[ct][0]) : 0.f;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// forward, second generation: the same MFMA chaining as mlp_fwd_kernel<..., CT = 2, NW = 8>, with less vector work around it:
// contiguous tile range per wave (scalar base + small per-lane offsets for every load / store), integer-max ReLU (one
// instruction; fmaxf on an MFMA result costs a canonicalising v_max as well), reciprocal-instruction sigmoid, only the
// per-sample input prefetched one tile ahead (the per-ray bias rows are L2-resident and are fetched at the start of the
// tile), which keeps the kernel at 4 waves per SIMD: the 1-KiB activation stores of one wave drain under the MFMAs of three
// others.  Measured at the metric size: head 0.69 -> see DESIGN.md; results are bit-identical to the first generation except
// for the sigmoid's reciprocal (<= 1 ulp).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float relu_bits(float x)
{
    return __int_as_float(max(__float_as_int(x), 0));      // x > 0 ? x : +0 for every non-NaN x (negative floats are negative ints)
}

__device__ __forceinline__ void store_act(float *p, f32x4 v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
}

template <int KIN, int WIDTH, int NHL, int INL>
__global__ __launch_bounds__(512) void mlp_fwd2_kernel(MlpArgs a, bool nt)
{
    constexpr int CT = 2, NW = 8, TS = 16 * CT;
    constexpr int HB = WIDTH / 16, KS0 = KIN / 4, KSH = WIDTH / 4;
    constexpr int IMG0 = HB * KS0, IMGH = (NHL == 2) ? HB * KSH : 0, IMGO = KSH;
    extern __shared__ float lds[];
    float *img0 = lds, *imgH = lds + IMG0 * 64, *imgO = imgH + IMGH * 64;

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    for (int e = threadIdx.x; e < IMG0 * 64; e += 64 * NW) {
        const int img = e >> 6, ln = e & 63, rb = img / KS0, ks = img % KS0, i = ln & 15, q = ln >> 4;
        const int c = kidx_in<INL>(ks, q);
        img0[e] = (a.w0_mask0 && c == 0) ? 0.f : W0[(16 * rb + i) * a.w0_ld + c];
    }
    if (NHL == 2)
        for (int e = threadIdx.x; e < IMGH * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            imgH[e] = W1[(16 * rb + i) * WIDTH + kidx_blk(ks, q)];
        }
    for (int e = threadIdx.x; e < IMGO * 64; e += 64 * NW) {
        const int ks = e >> 6, ln = e & 63, i = ln & 15, q = ln >> 4;
        imgO[e] = Wo[i * WIDTH + kidx_blk(ks, q)];
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    a.n = lse::clamp_count(a.n, a.n_dev);
    const int64_t n = a.n, ns = a.n_stride;
    const int64_t n_tiles = (n + TS - 1) / TS;
    const int64_t total_waves = (int64_t)gridDim.x * NW;
    const int64_t per = (n_tiles + total_waves - 1) / total_waves;
    const int64_t w_id = (int64_t)blockIdx.x * NW + wave;
    const int64_t t_begin = min(n_tiles, w_id * per), t_end = min(n_tiles, t_begin + per);
    const int oc = a.out_cols;

    // layer-0 B operands of a tile: breg[ct][ks] = in[sample][kidx_in(ks, q)]
    auto load_in = [&](int64_t tile, float (&breg)[CT][KS0]) {
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int l = ct * 16 + j;
            const int slc = l < n_rem ? l : n_rem - 1;
            if (INL == LSE_IN_LEVELMAJOR) {
                const float2 *in2 = reinterpret_cast<const float2 *>(a.in) + tile_base;
#pragma unroll
                for (int m = 0; m < KIN / 8; ++m) {
                    const float2 v = in2[(int64_t)(4 * m + q) * ns + slc];
                    breg[ct][2 * m] = v.x;
                    breg[ct][2 * m + 1] = v.y;
                }
            } else {
                const float *in_t = a.in + tile_base * KIN;
#pragma unroll
                for (int b = 0; b < KIN / 16; ++b) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(in_t + (unsigned)(slc * KIN + 16 * b + 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r) breg[ct][4 * b + r] = v[r];
                }
            }
        }
    };

    float breg_nx[CT][KS0];
    if (t_begin < t_end) load_in(t_begin, breg_nx);
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_iglp_opt(0);
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);
        int sl[CT];
        bool valid[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            valid[ct] = ct * 16 + j < n_rem;
            sl[ct] = valid[ct] ? ct * 16 + j : n_rem - 1;
        }
        float breg[CT][KS0];
        f32x4 h[HB][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) breg[ct][ks] = breg_nx[ct][ks];
        // per-row layer-0 bias = the initial accumulator (rows repeat along a ray: L2 / L1 hits)
        if (a.row_bias) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int64_t row = a.row_bias_idx ? (int64_t)a.row_bias_idx[tile_base + sl[ct]] : tile_base + sl[ct];
#pragma unroll
                for (int rb = 0; rb < HB; ++rb)
                    h[rb][ct] = *reinterpret_cast<const f32x4 *>(a.row_bias + row * WIDTH + 16 * rb + 4 * q);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (tile + 1 < t_end) load_in(tile + 1, breg_nx);
        // ---- layer 0
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) {
                const float aw = img0[(rb * KS0 + ks) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h[rb][ct] = LSE_MFMA(aw, breg[ct][ks], h[rb][ct]);
            }
        }
        // tile-major activation image of this tile: [ct][rb][64 lanes][4]; a second column tile beyond n is not stored
        float *act0_t = a.act ? a.act + tile * (CT * HB * 256) : nullptr;
        const bool st1 = 16 < n_rem;
        const bool skip0 = NHL == 2 && a.act_tiled == 2;      // the backward recomputes the first hidden layer
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[rb][ct][r] = relu_bits(h[rb][ct][r]);
                if (act0_t && !skip0 && (ct == 0 || st1)) store_act(act0_t + (unsigned)(((ct * HB + rb) * 64 + lane) * 4), h[rb][ct], nt);
            }
        // ---- hidden layer
        if (NHL == 2) {
            f32x4 h2[HB][CT];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h2[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb) {
                        const float aw = imgH[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h2[rb][ct] = LSE_MFMA(aw, h[bp][ct][r], h2[rb][ct]);
                    }
                }
            float *act1_t = act0_t ? act0_t + (skip0 ? 0 : a.act_layer_stride) : nullptr;
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[rb][ct][r] = relu_bits(h2[rb][ct][r]);
                    if (act1_t && (ct == 0 || st1)) store_act(act1_t + (unsigned)(((ct * HB + rb) * 64 + lane) * 4), h[rb][ct], nt);
                }
        }
        // ---- output layer
        f32x4 o[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bp = 0; bp < HB; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float aw = imgO[(4 * bp + r) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) o[ct] = LSE_MFMA(aw, h[bp][ct][r], o[ct]);
            }
        float *out_t = a.out + tile_base * oc;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (a.out_activation == LSE_ACT_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[ct][r] = __builtin_amdgcn_rcpf(1.f + __expf(-o[ct][r]));
            }
            if (valid[ct]) {
                if (oc == 16) *reinterpret_cast<f32x4 *>(out_t + (unsigned)(sl[ct] * 16 + 4 * q)) = o[ct];
                else if (q == 0) *reinterpret_cast<f32x4 *>(out_t + (unsigned)(sl[ct] * 4)) = o[ct];
                if (a.sigma_out && q == 0) {
                    const bool in_bounds = a.selector == nullptr || a.selector[tile_base + sl[ct]] != 0;
                    a.sigma_out[tile_base + sl[ct]] = in_bounds ? a.density_scale * expf(o[ct][0]) : 0.f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward:  dOut -> dH_last -> (dH_0) -> dIn chained in registers with transposed A images, and (WGRAD) the weight
// gradients of every layer in the same pass.
//
// Weight gradients need the SAMPLE index as the MFMA k dimension:  dW[m][k] += sum_s G[s][m] * A[s][k].
//   * G (a gradient tile) and A (the saved activation tile, already loaded for the ReLU mask) both live in accumulator
//     layout (sample on the lane, neuron on registers).  Each 16x16 (sample x neuron) block is transposed through a
//     per-wave LDS buffer (row pitch 20 floats: conflict-free ds_write_b128, 4 ds_read_b32) into operand layout:
//     lane (i,q) k-step t <- X[sample 4t+q][neuron i].  The layer-0 input is read in operand layout from memory.
//   * the dW accumulator tiles stay in REGISTERS for the whole kernel (one wave per SIMD, 512-register budget:
//     __launch_bounds__(256, 1)) and are flushed once per wave with global float atomics.  (An LDS-resident dW copy
//     updated with ds_add_f32 was measured LDS-bound: ~190 LDS cycles per atomic wave-instruction.)
// This removes the materialised d_act round trip and the separate G^T*A reduction launches.
// ------------------------------------------------------------------------------------------------------
constexpr int kTrPitch = 20;                 // floats per transposition-buffer row
constexpr int kTrBlock = 16 * kTrPitch;      // one 16x16 block
constexpr int kTrWave = 8 * kTrBlock;        // up to 4 + 4 blocks (G and A side) in flight per wave

// in: v[mb] = block (rows 16mb..16mb+15 of G^T) for the 16 samples of one column tile, accumulator layout.
// out: ga[mb][t] = A operand of k-step t (samples 4t..4t+3 of this column tile).
template <int MB>
__device__ __forceinline__ void transpose_to_a_operand(float *tr, const f32x4 (&v)[MB], float (&ga)[MB][4], int j, int q)
{
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<f32x4 *>(tr + mb * kTrBlock + j * kTrPitch + 4 * q) = v[mb];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < 4; ++t) ga[mb][t] = tr[mb * kTrBlock + (4 * t + q) * kTrPitch + j];
    __builtin_amdgcn_wave_barrier();
}

// acc[MB][KB] (persistent 16x16 accumulator tiles of dW) += G^T A for one wave tile of 16*CT samples.
//   G: MB row blocks x CT column tiles in accumulator layout (zero for invalid samples)  -> A operands via LDS transpose.
//   A (two sources):
//     wgrad_from_regs: A is also in accumulator layout (the activation tile loaded for the ReLU mask) -> B operands by
//                      the same transpose (lane (j,q) k-step t <- A[sample 4t+q][column j]);
//     wgrad_from_mem : A is the layer-0 input in memory, read in B-operand layout directly.
template <int MB, int KB, int CT>
__device__ __forceinline__ void wgrad_from_regs(f32x4 (&acc)[MB][KB], float *tr, const f32x4 (&G)[MB][CT],
                                                const f32x4 (&A)[KB][CT], int j, int q)
{
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4 gblk[MB], ablk[KB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) gblk[mb] = G[mb][ct];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) ablk[kb] = A[kb][ct];
        float ga[MB][4], gb[KB][4];
        transpose_to_a_operand<MB>(tr, gblk, ga, j, q);
        transpose_to_a_operand<KB>(tr + MB * kTrBlock, ablk, gb, j, q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc[mb][kb] = LSE_MFMA(ga[mb][t], gb[kb][t], acc[mb][kb]);
    }
}

template <int MB, int K, int AL, int CT>
__device__ __forceinline__ void wgrad_from_mem(f32x4 (&acc)[MB][(K + 15) / 16], float *tr, const f32x4 (&G)[MB][CT],
                                               const float *A, int64_t n, const int64_t tile_base, int j, int q)
{
    constexpr int KB = (K + 15) / 16;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        float gb[KB][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int64_t srow = tile_base + ct * 16 + 4 * t + q;
            srow = srow < n ? srow : n - 1;          // G is zero there
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int col = 16 * kb + j;
                float v = 0.f;
                if (col < K) {
                    if (AL == LSE_IN_LEVELMAJOR) v = A[((int64_t)(col >> 1) * n + srow) * 2 + (col & 1)];
                    else v = A[srow * K + col];
                }
                gb[kb][t] = v;
            }
        }
        f32x4 gblk[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) gblk[mb] = G[mb][ct];
        float ga[MB][4];
        transpose_to_a_operand<MB>(tr, gblk, ga, j, q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc[mb][kb] = LSE_MFMA(ga[mb][t], gb[kb][t], acc[mb][kb]);
    }
}

// flush persistent accumulator tiles: D row = 4q + r -> m, col = j -> k
template <int MB, int KB>
__device__ __forceinline__ void flush_wgrad(float *dw, int ld, int k_real, const f32x4 (&acc)[MB][KB], int j, int q,
                                            bool skip_col0 = false)
{
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            if (16 * kb + j < k_real && !(skip_col0 && 16 * kb + j == 0)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&dw[(16 * mb + 4 * q + r) * ld + 16 * kb + j], acc[mb][kb][r]);
            }
}

// d(row_bias)[row] += sum over the samples of that row of the layer-0 pre-activation gradient.  Samples are sorted by
// row (a ray's samples are contiguous), so inside one 16-sample column tile the rows form contiguous runs: a 16-lane
// segmented inclusive scan leaves every run's total in its last lane, which adds 4 x 16 B per row block to d_row_bias
// (one 64-B line request per (tile, row block) instead of a materialised [N, W] gradient + a second reduction pass).
template <int HB, int WIDTH>
__device__ __forceinline__ void row_bias_grad_tile(float *d_row_bias, int row, const f32x4 (&g)[HB], int j, int q)
{
    f32x4 v[HB];
#pragma unroll
    for (int rb = 0; rb < HB; ++rb) v[rb] = g[rb];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const int row_o = __shfl_up(row, off, 16);
        const bool take = (j >= off) && (row_o == row);
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = __shfl_up(v[rb][r], off, 16);
                v[rb][r] += take ? t : 0.f;
            }
    }
    const int row_n = __shfl_down(row, 1, 16);
    if (j == 15 || row_n != row) {
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&d_row_bias[(int64_t)row * WIDTH + 16 * rb + 4 * q + r], v[rb][r]);
    }
}

template <int KIN, int WIDTH, int NHL, int INL, bool WGRAD, int CT, int NW>
__global__ __launch_bounds__(64 * NW, ((NW == 4 && !WGRAD) ? 2 : 1)) void mlp_bwd_kernel(MlpArgs a)
{
    constexpr int HB = WIDTH / 16, KSH = WIDTH / 4;
    constexpr int RB0 = (KIN + 15) / 16;
    constexpr int IMGO = HB * 4, IMGH = (NHL == 2) ? HB * KSH : 0, IMGI = RB0 * KSH;
    constexpr int NP1 = (NHL - 1) * WIDTH * WIDTH;
    extern __shared__ float lds[];
    float *imgO = lds, *imgH = lds + IMGO * 64, *imgI = imgH + IMGH * 64;
    float *tr_all = imgI + IMGI * 64;                       // NW waves x kTrWave (WGRAD only)

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    // dH_last^T = Wo^T (WIDTH x 16) * dOut^T : A[i][k] = Wo[k = 4q+ks][16rb+i]
    for (int e = threadIdx.x; e < IMGO * 64; e += 64 * NW) {
        const int img = e >> 6, ln = e & 63, rb = img >> 2, ks = img & 3, i = ln & 15, q = ln >> 4;
        imgO[e] = Wo[(4 * q + ks) * WIDTH + 16 * rb + i];
    }
    if (NHL == 2)   // dH_0^T = W1^T * dH_1^T : A[i][k] = W1[kidx_blk(ks,q)][16rb+i]
        for (int e = threadIdx.x; e < IMGH * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            imgH[e] = W1[kidx_blk(ks, q) * WIDTH + 16 * rb + i];
        }
    if (a.d_in)     // dIn^T = W0^T (KIN x WIDTH) * dH_0^T : A[i][k] = W0[kidx_blk(ks,q)][16rb+i]
        for (int e = threadIdx.x; e < IMGI * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            const int col = 16 * rb + i;
            imgI[e] = (col < KIN && !(a.w0_mask0 && col == 0)) ? W0[kidx_blk(ks, q) * a.w0_ld + col] : 0.f;
        }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    float *tr = tr_all + wave * kTrWave;
    // persistent weight-gradient accumulators (WGRAD): dWo[16 x W], dW1[W x W], dW0[W x KIN]
    constexpr int KB0 = (KIN + 15) / 16;
    f32x4 accO[1][HB], acc1[(NHL == 2) ? HB : 1][(NHL == 2) ? HB : 1], acc0[HB][KB0];
    if (WGRAD) {
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) accO[0][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < ((NHL == 2) ? HB : 1); ++mb)
#pragma unroll
            for (int kb = 0; kb < ((NHL == 2) ? HB : 1); ++kb) acc1[mb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb) acc0[mb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    a.n = lse::clamp_count(a.n, a.n_dev);
    const int64_t n = a.n, ns = a.n_stride;
    constexpr int TS = 16 * CT;   // samples per wave tile
    const int64_t n_tiles = (n + TS - 1) / TS;
    const float *act_last = a.act + (int64_t)(NHL - 1) * a.act_layer_stride;
    float *dact_last = a.d_act ? a.d_act + (int64_t)(NHL - 1) * ns * WIDTH : nullptr;
    // One tile ahead: the raw operands of the NEXT tile (output gradient, saved last-layer activations, the output
    // values the activation / density backward needs) are requested before the current tile's MFMA work starts -- the
    // fused-wgrad kernel runs one wave per SIMD, so nothing else would hide that latency.
    const bool need_out = a.out_activation == LSE_ACT_SIGMOID;
    auto fetch = [&](int64_t tile, int64_t (&s)[CT], bool (&valid)[CT], f32x4 (&g)[1][CT], f32x4 (&ov)[CT], float (&dsg)[CT],
                     f32x4 (&hv)[HB][CT]) {
        const int64_t tile_base = tile * TS;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int64_t si = tile_base + ct * 16 + j;
            valid[ct] = si < n;
            s[ct] = valid[ct] ? si : n - 1;
            if (a.out_cols == 16) g[0][ct] = *reinterpret_cast<const f32x4 *>(a.d_out + s[ct] * 16 + 4 * q);
            else g[0][ct] = (q == 0) ? *reinterpret_cast<const f32x4 *>(a.d_out + s[ct] * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            ov[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (need_out) {
                if (a.out_cols == 16) ov[ct] = *reinterpret_cast<const f32x4 *>(a.out + s[ct] * 16 + 4 * q);
                else if (q == 0) ov[ct] = *reinterpret_cast<const f32x4 *>(a.out + s[ct] * 4);
            }
            dsg[ct] = 0.f;
            if (a.d_sigma && q == 0) {   // trunc_exp backward: d out0 += d_sigma * scale * exp(clamp(out0, -15, 15)) * selector
                const bool in_bounds = a.selector == nullptr || a.selector[s[ct]] != 0;
                if (!need_out) ov[ct][0] = a.out[s[ct] * a.out_cols];
                dsg[ct] = in_bounds ? a.d_sigma[s[ct]] : 0.f;
            }
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
                hv[rb][ct] = *reinterpret_cast<const f32x4 *>(act_last + act_offset<WIDTH>(a.act_tiled, s[ct], rb, q));
        }
    };
    const int64_t tile_stride = (int64_t)gridDim.x * NW;
    int64_t s[CT], s_n[CT];
    bool valid[CT], valid_n[CT];
    f32x4 g[1][CT], g_n[1][CT], ov[CT], ov_n[CT], hv[HB][CT], hv_n[HB][CT];
    float dsg[CT], dsg_n[CT];
    // (the two-hidden-layer kernel already uses all 256 registers a wave gets at 8 waves per CU: it fetches in place)
    constexpr bool PF = (NHL == 1) || (NW == 4 && WGRAD);
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    if (PF && tile < n_tiles) fetch(tile, s_n, valid_n, g_n, ov_n, dsg_n, hv_n);
    for (; tile < n_tiles; tile += tile_stride) {
        __builtin_amdgcn_iglp_opt(0);   // scheduler hint: interleave LDS reads with the MFMA stream (forward: 0.72 -> 0.66 ms)
        const int64_t tile_base = tile * TS;
        if (!PF) fetch(tile, s_n, valid_n, g_n, ov_n, dsg_n, hv_n);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            s[ct] = s_n[ct];
            valid[ct] = valid_n[ct];
            g[0][ct] = g_n[0][ct];
            ov[ct] = ov_n[ct];
            dsg[ct] = dsg_n[ct];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) hv[rb][ct] = hv_n[rb][ct];
        }
        if (PF && tile + tile_stride < n_tiles) fetch(tile + tile_stride, s_n, valid_n, g_n, ov_n, dsg_n, hv_n);
        f32x4 hv0[(PF && NHL == 2) ? HB : 1][CT];   // first-layer activations of THIS tile, requested before the MFMA work
        if constexpr (PF && NHL == 2) {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    hv0[rb][ct] = *reinterpret_cast<const f32x4 *>(a.act + act_offset<WIDTH>(a.act_tiled, s[ct], rb, q));
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float x0 = fminf(fmaxf(ov[ct][0], -15.f), 15.f);   // lanes q == 0 hold out[s][0]
            if (need_out) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g[0][ct][r] = g[0][ct][r] * ov[ct][r] * (1.f - ov[ct][r]);
            }
            if (a.d_sigma && q == 0) g[0][ct][0] += dsg[ct] * a.density_scale * expf(x0);
            if (!valid[ct]) g[0][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};   // tail columns contribute nothing downstream
            if (a.d_out_pre && valid[ct]) *reinterpret_cast<f32x4 *>(a.d_out_pre + s[ct] * 16 + 4 * q) = g[0][ct];
        }
        // ---- output-layer weights: dWo[16 x WIDTH] += g^T * act_last
        if (WGRAD) wgrad_from_regs<1, HB, CT>(accO, tr, g, hv, j, q);
        // ---- dH_last
        f32x4 dh[HB][CT];
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dh[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) {
                const float aw = imgO[(rb * 4 + r) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) dh[rb][ct] = LSE_MFMA(aw, g[0][ct][r], dh[rb][ct]);
            }
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[rb][ct][r] = hv[rb][ct][r] > 0.f ? dh[rb][ct][r] : 0.f;
                if (valid[ct]) {
                    if (dact_last) *reinterpret_cast<f32x4 *>(dact_last + s[ct] * WIDTH + 16 * rb + 4 * q) = dh[rb][ct];
                    if (NHL == 1 && a.d_act0) *reinterpret_cast<f32x4 *>(a.d_act0 + s[ct] * WIDTH + 16 * rb + 4 * q) = dh[rb][ct];
                }
            }
        // ---- dH_0 (two hidden layers)
        if (NHL == 2) {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if constexpr (PF && NHL == 2) hv[rb][ct] = hv0[rb][ct];
                    else hv[rb][ct] = *reinterpret_cast<const f32x4 *>(a.act + act_offset<WIDTH>(a.act_tiled, s[ct], rb, q));
                }
            if constexpr (WGRAD && NHL == 2) wgrad_from_regs<HB, HB, CT>(acc1, tr, dh, hv, j, q);
            f32x4 d0[HB][CT];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) d0[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb) {
                        const float aw = imgH[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) d0[rb][ct] = LSE_MFMA(aw, dh[bp][ct][r], d0[rb][ct]);
                    }
                }
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[rb][ct][r] = hv[rb][ct][r] > 0.f ? d0[rb][ct][r] : 0.f;
                    if (valid[ct]) {
                        if (a.d_act) *reinterpret_cast<f32x4 *>(a.d_act + s[ct] * WIDTH + 16 * rb + 4 * q) = dh[rb][ct];
                        if (a.d_act0) *reinterpret_cast<f32x4 *>(a.d_act0 + s[ct] * WIDTH + 16 * rb + 4 * q) = dh[rb][ct];
                    }
                }
        }
        // ---- gradient of the per-row layer-0 bias
        if (a.d_row_bias) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 gcol[HB];
#pragma unroll
                for (int rb = 0; rb < HB; ++rb) gcol[rb] = dh[rb][ct];      // zero for tail columns
                row_bias_grad_tile<HB, WIDTH>(a.d_row_bias, a.row_bias_idx[s[ct]], gcol, j, q);
            }
        }
        // ---- layer-0 weights: dW0[WIDTH x KIN] += dH_0^T * in
        if (WGRAD) wgrad_from_mem<HB, KIN, INL, CT>(acc0, tr, dh, a.in, n, tile_base, j, q);
        // ---- dIn
        if (a.d_in) {
            f32x4 di[RB0][CT];
#pragma unroll
            for (int rb = 0; rb < RB0; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) di[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < RB0; ++rb) {
                        const float aw = imgI[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) di[rb][ct] = LSE_MFMA(aw, dh[bp][ct][r], di[rb][ct]);
                    }
                }
#pragma unroll
            for (int rb = 0; rb < RB0; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (!valid[ct]) continue;
                    if (INL == LSE_IN_LEVELMAJOR) {
                        // rows 16rb+4q+r are features; feature f -> level f>>1, component f&1
                        if (16 * rb + 4 * q < KIN) {
                            float2 *d2 = reinterpret_cast<float2 *>(a.d_in);
                            const int lv = 8 * rb + 2 * q;
                            d2[(int64_t)lv * ns + s[ct]] = make_float2(di[rb][ct][0], di[rb][ct][1]);
                            d2[(int64_t)(lv + 1) * ns + s[ct]] = make_float2(di[rb][ct][2], di[rb][ct][3]);
                        }
                    } else {
                        if (16 * rb + 4 * q < KIN)
                            *reinterpret_cast<f32x4 *>(a.d_in + s[ct] * KIN + 16 * rb + 4 * q) = di[rb][ct];
                    }
                }
        }
    }
    if (WGRAD) {
        flush_wgrad<HB, KB0>(a.d_params + a.w0_col, a.w0_ld, KIN, acc0, j, q, a.w0_mask0 != 0);
        if constexpr (NHL == 2) flush_wgrad<HB, HB>(a.d_params + a.rest_off, WIDTH, WIDTH, acc1, j, q);
        flush_wgrad<1, HB>(a.d_params + a.rest_off + NP1, WIDTH, WIDTH, accO, j, q);
    }
}


// ------------------------------------------------------------------------------------------------------
// backward, second generation: the same mathematics and operand flow as mlp_bwd_kernel<..., WGRAD = true, CT = 2, NW = 8>
// (fused data + weight gradients, tile-major saved activations), with the vector-instruction overhead around the MFMAs cut:
//   * every wave walks a CONTIGUOUS range of 32-sample tiles, so all per-tile addresses are "scalar base + small per-lane
//     offset" (one SALU update per array and tile instead of 64-bit VALU arithmetic per load / store);
//   * per-row bias gradient (the head's SH + embedding share, lsenerf_amd/field.py) WITHOUT the 16-lane segmented scan
//     (4 x 16 x {ds_bpermute, v_cndmask, v_add} per column tile = a third of the old kernel's vector instructions):
//     with the head's first-layer view, input column 0 (the density logit) carries no weight, so its column of the dW0
//     accumulator tile is free.  Feeding the constant 1 as that input column makes the MFMAs that run anyway accumulate
//     sum_s dH0[s][m] there -- the bias gradient of the current ray.  Rays are contiguous: the column is flushed (16 atomic
//     instructions) only when the wave moves on to another ray; a column tile that straddles two rays (one in 64 at 1024
//     samples per ray) falls back to the scan for that tile alone.  Template parameter BIAS_ONES.
// ------------------------------------------------------------------------------------------------------
//   * RECOMP (two hidden layers, row-major input): the forward did not save the first hidden layer (act_tiled = 2); it is
//     recomputed here from the layer-0 input and the per-row bias -- the same MFMA chain in the same k order, so the values
//     are bit-identical to the forward's -- at 32 extra MFMAs per tile against 1 KiB x 8 of activations written and read.
template <int KIN, int WIDTH, int NHL, int INL, bool BIAS_ONES, bool RECOMP = false>
__global__ __launch_bounds__(512, 1) void mlp_bwd2_kernel(MlpArgs a)
{
    constexpr int CT = 2, NW = 8, TS = 16 * CT;
    constexpr int HB = WIDTH / 16, KSH = WIDTH / 4, KS0 = KIN / 4;
    constexpr int RB0 = (KIN + 15) / 16, KB0 = RB0;
    constexpr int IMGO = HB * 4, IMGH = (NHL == 2) ? HB * KSH : 0, IMGI = RB0 * KSH, IMG0 = RECOMP ? HB * KS0 : 0;
    constexpr int NP1 = (NHL - 1) * WIDTH * WIDTH;
    static_assert(!RECOMP || (NHL == 2 && INL == LSE_IN_ROWMAJOR && KIN % 16 == 0), "recompute: two hidden layers, row-major input");
    extern __shared__ float lds[];
    float *imgO = lds, *imgH = lds + IMGO * 64, *imgI = imgH + IMGH * 64;
    float *img0 = imgI + IMGI * 64;                           // forward image of W0 (RECOMP only)
    float *tr_all = img0 + IMG0 * 64;

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    for (int e = threadIdx.x; e < IMGO * 64; e += 64 * NW) {
        const int img = e >> 6, ln = e & 63, rb = img >> 2, ks = img & 3, i = ln & 15, q = ln >> 4;
        imgO[e] = Wo[(4 * q + ks) * WIDTH + 16 * rb + i];
    }
    if (NHL == 2)
        for (int e = threadIdx.x; e < IMGH * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            imgH[e] = W1[kidx_blk(ks, q) * WIDTH + 16 * rb + i];
        }
    if (a.d_in)
        for (int e = threadIdx.x; e < IMGI * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KSH, ks = img % KSH, i = ln & 15, q = ln >> 4;
            const int col = 16 * rb + i;
            imgI[e] = (col < KIN && !(a.w0_mask0 && col == 0)) ? W0[kidx_blk(ks, q) * a.w0_ld + col] : 0.f;
        }
    if constexpr (RECOMP)
        for (int e = threadIdx.x; e < IMG0 * 64; e += 64 * NW) {
            const int img = e >> 6, ln = e & 63, rb = img / KS0, ks = img % KS0, i = ln & 15, q = ln >> 4;
            const int c = kidx_blk(ks, q);
            img0[e] = (a.w0_mask0 && c == 0) ? 0.f : W0[(16 * rb + i) * a.w0_ld + c];
        }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    float *tr = tr_all + wave * kTrWave;
    f32x4 accO[1][HB], acc1[(NHL == 2) ? HB : 1][(NHL == 2) ? HB : 1], acc0[HB][KB0];
#pragma unroll
    for (int kb = 0; kb < HB; ++kb) accO[0][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < ((NHL == 2) ? HB : 1); ++mb)
#pragma unroll
        for (int kb = 0; kb < ((NHL == 2) ? HB : 1); ++kb) acc1[mb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < HB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) acc0[mb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    a.n = lse::clamp_count(a.n, a.n_dev);
    const int64_t n = a.n, ns = a.n_stride;
    const int64_t n_tiles = (n + TS - 1) / TS;
    const int64_t total_waves = (int64_t)gridDim.x * NW;
    const int64_t per = (n_tiles + total_waves - 1) / total_waves;
    const int64_t w_id = (int64_t)blockIdx.x * NW + wave;
    const int64_t t_begin = min(n_tiles, w_id * per), t_end = min(n_tiles, t_begin + per);
    const float *act_last = a.act + (RECOMP ? 0 : (int64_t)(NHL - 1) * a.act_layer_stride);
    const bool need_out = a.out_activation == LSE_ACT_SIGMOID;
    const int oc = a.out_cols;
    int cur_row = -1;            // BIAS_ONES: the row whose running sum sits in column 0 of acc0[.][0]

    // column 0 of the dW0 accumulator tile = sum_s dH0[s][m] of row `cur_row`: add it to d_row_bias and clear it
    auto flush_bias_col = [&]() {
        if (cur_row >= 0 && j == 0) {
            float *dst = a.d_row_bias + (int64_t)cur_row * WIDTH + 4 * q;
#pragma unroll
            for (int mb = 0; mb < HB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dst + 16 * mb + r, acc0[mb][0][r]);
        }
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[mb][0][r] = (j == 0) ? 0.f : acc0[mb][0][r];
    };

    // raw operands of one tile: output gradient, output values (activation / density backward), last-layer activations
    auto fetch = [&](int64_t tile, f32x4 (&g)[1][CT], f32x4 (&ov)[CT], float (&dsg)[CT], f32x4 (&hv)[HB][CT]) {
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);
        const float *dout_t = a.d_out + tile_base * oc;
        const float *out_t = a.out ? a.out + tile_base * oc : nullptr;
        const float *actl_t = act_last + tile * (CT * HB * 256);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int l = ct * 16 + j;
            const int slc = l < n_rem ? l : n_rem - 1;
            if (oc == 16) g[0][ct] = *reinterpret_cast<const f32x4 *>(dout_t + (unsigned)(slc * 16 + 4 * q));
            else g[0][ct] = (q == 0) ? *reinterpret_cast<const f32x4 *>(dout_t + (unsigned)(slc * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            ov[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (need_out) {
                if (oc == 16) ov[ct] = *reinterpret_cast<const f32x4 *>(out_t + (unsigned)(slc * 16 + 4 * q));
                else if (q == 0) ov[ct] = *reinterpret_cast<const f32x4 *>(out_t + (unsigned)(slc * 4));
            }
            dsg[ct] = 0.f;
            if (a.d_sigma && q == 0) {
                const bool in_bounds = a.selector == nullptr || a.selector[tile_base + slc] != 0;
                if (!need_out) ov[ct][0] = out_t[(unsigned)(slc * oc)];
                dsg[ct] = in_bounds ? a.d_sigma[tile_base + slc] : 0.f;
            }
            const int cta = (ct == 0 || 16 >= n_rem) ? 0 : 1;       // a fully invalid second column tile re-reads the first
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
                hv[rb][ct] = *reinterpret_cast<const f32x4 *>(actl_t + (unsigned)(((cta * HB + rb) * 64 + lane) * 4));
        }
    };
    // one tile ahead where the registers allow it (single hidden layer): two waves per SIMD alone do not hide the HBM
    // latency of the activation / gradient loads (measured: 48 % of the wave cycles in s_waitcnt without it)
    constexpr bool PF = (NHL == 1);
    f32x4 g_n[1][CT], ov_n[CT], hv_n[HB][CT];
    float dsg_n[CT];
    if (PF && t_begin < t_end) fetch(t_begin, g_n, ov_n, dsg_n, hv_n);

    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_iglp_opt(0);
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);       // valid samples of this tile (wave-uniform, >= 1)
        int sl[CT];                                                    // this lane's sample slot per column tile, clamped
        bool valid[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            valid[ct] = ct * 16 + j < n_rem;
            sl[ct] = valid[ct] ? ct * 16 + j : n_rem - 1;
        }
        const int act_ct1 = (16 < n_rem) ? 1 : 0;
        const float *act0_t = a.act + tile * (CT * HB * 256);
        f32x4 g[1][CT], ov[CT], hv[HB][CT];
        float dsg[CT];
        if constexpr (PF) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                g[0][ct] = g_n[0][ct];
                ov[ct] = ov_n[ct];
                dsg[ct] = dsg_n[ct];
#pragma unroll
                for (int rb = 0; rb < HB; ++rb) hv[rb][ct] = hv_n[rb][ct];
            }
            if (tile + 1 < t_end) fetch(tile + 1, g_n, ov_n, dsg_n, hv_n);
        } else {
            fetch(tile, g, ov, dsg, hv);
        }
        f32x4 hv0[(NHL == 2) ? HB : 1][CT];
        float breg0[RECOMP ? CT : 1][RECOMP ? KS0 : 1];        // RECOMP: layer-0 B operands, requested now, used after dH_last
        if constexpr (NHL == 2 && !RECOMP) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int cta = ct == 0 ? 0 : act_ct1;
#pragma unroll
                for (int rb = 0; rb < HB; ++rb)
                    hv0[rb][ct] = *reinterpret_cast<const f32x4 *>(act0_t + (unsigned)(((cta * HB + rb) * 64 + lane) * 4));
            }
        }
        if constexpr (RECOMP) {
            const float *in_t = a.in + tile_base * KIN;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int b = 0; b < KIN / 16; ++b) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(in_t + (unsigned)(sl[ct] * KIN + 16 * b + 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r) breg0[ct][4 * b + r] = v[r];
                }
        }
        // layer-0 input in B-operand layout for dW0 (lane (j, q), k-step t: in[sample 4t+q][column 16kb+j]), requested now so
        // that its latency hides under the MFMA chain below (single hidden layer; the two-layer kernel has no registers left)
        constexpr bool EARLY_IN = (NHL == 1);
        float gbe[EARLY_IN ? CT : 1][KB0][4];
        auto load_gb = [&](int ct, float (&gb)[KB0][4]) {
            const float *in_t = (INL == LSE_IN_LEVELMAJOR) ? a.in + tile_base * 2 : a.in + tile_base * KIN;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int srow = ct * 16 + 4 * t + q;
                srow = srow < n_rem ? srow : n_rem - 1;                       // dH is zero there
#pragma unroll
                for (int kb = 0; kb < KB0; ++kb) {
                    const int col = 16 * kb + j;
                    float v = 0.f;
                    if (col < KIN) {
                        if (INL == LSE_IN_LEVELMAJOR) v = in_t[((int64_t)(col >> 1) * ns + srow) * 2 + (col & 1)];
                        else v = in_t[(unsigned)(srow * KIN + col)];
                    }
                    gb[kb][t] = v;
                }
            }
        };
        if constexpr (EARLY_IN) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) load_gb(ct, gbe[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float x0 = fminf(fmaxf(ov[ct][0], -15.f), 15.f);
            if (need_out) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g[0][ct][r] = g[0][ct][r] * ov[ct][r] * (1.f - ov[ct][r]);
            }
            if (a.d_sigma && q == 0) g[0][ct][0] += dsg[ct] * a.density_scale * expf(x0);
            if (!valid[ct]) g[0][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- output-layer weights
        wgrad_from_regs<1, HB, CT>(accO, tr, g, hv, j, q);
        // ---- dH_last
        f32x4 dh[HB][CT];
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dh[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) {
                const float aw = imgO[(rb * 4 + r) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) dh[rb][ct] = LSE_MFMA(aw, g[0][ct][r], dh[rb][ct]);
            }
#pragma unroll
        for (int rb = 0; rb < HB; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[rb][ct][r] = hv[rb][ct][r] > 0.f ? dh[rb][ct][r] : 0.f;
        // ---- dH_0 (two hidden layers)
        if constexpr (NHL == 2) {
            if constexpr (RECOMP) {     // first hidden layer again: relu(W0 * in + row bias), the forward's MFMA chain
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (a.row_bias) {
                        const int64_t row = a.row_bias_idx ? (int64_t)a.row_bias_idx[tile_base + sl[ct]] : tile_base + sl[ct];
#pragma unroll
                        for (int rb = 0; rb < HB; ++rb)
                            hv0[rb][ct] = *reinterpret_cast<const f32x4 *>(a.row_bias + row * WIDTH + 16 * rb + 4 * q);
                    } else {
#pragma unroll
                        for (int rb = 0; rb < HB; ++rb) hv0[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb) {
                        const float aw = img0[(rb * KS0 + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) hv0[rb][ct] = LSE_MFMA(aw, breg0[ct][ks], hv0[rb][ct]);
                    }
#pragma unroll
                for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hv0[rb][ct][r] = relu_bits(hv0[rb][ct][r]);
            }
            wgrad_from_regs<HB, HB, CT>(acc1, tr, dh, hv0, j, q);
            f32x4 d0[HB][CT];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) d0[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb) {
                        const float aw = imgH[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) d0[rb][ct] = LSE_MFMA(aw, dh[bp][ct][r], d0[rb][ct]);
                    }
                }
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[rb][ct][r] = hv0[rb][ct][r] > 0.f ? d0[rb][ct][r] : 0.f;
        }
        // ---- layer-0 weights dW0 += dH_0^T * in, and the per-row bias gradient
        {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float ones = 0.f;
                if (a.d_row_bias) {
                    const int row = a.row_bias_idx[tile_base + sl[ct]];
                    if constexpr (BIAS_ONES) {
                        const int first = __builtin_amdgcn_readfirstlane(row);
                        if (__builtin_amdgcn_ballot_w64(row != first) == 0) {     // one ray owns this column tile
                            if (first != cur_row) {
                                flush_bias_col();
                                cur_row = first;
                            }
                            ones = 1.f;
                        }
                    }
                    if (!BIAS_ONES || ones == 0.f) {                              // straddles two rows: 16-lane segmented scan
                        f32x4 gcol[HB];
#pragma unroll
                        for (int rb = 0; rb < HB; ++rb) gcol[rb] = dh[rb][ct];
                        row_bias_grad_tile<HB, WIDTH>(a.d_row_bias, row, gcol, j, q);
                    }
                }
                float gb[KB0][4];
                if constexpr (EARLY_IN) {
#pragma unroll
                    for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) gb[kb][t] = gbe[ct][kb][t];
                } else {
                    load_gb(ct, gb);
                }
                if constexpr (BIAS_ONES) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) gb[0][t] = (j == 0) ? ones : gb[0][t];   // the free input column carries the constant
                }
                f32x4 gblk[HB];
#pragma unroll
                for (int mb = 0; mb < HB; ++mb) gblk[mb] = dh[mb][ct];
                float ga[HB][4];
                transpose_to_a_operand<HB>(tr, gblk, ga, j, q);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mb = 0; mb < HB; ++mb)
#pragma unroll
                        for (int kb = 0; kb < KB0; ++kb) acc0[mb][kb] = LSE_MFMA(ga[mb][t], gb[kb][t], acc0[mb][kb]);
            }
        }
        // ---- dIn
        if (a.d_in) {
            f32x4 di[RB0][CT];
#pragma unroll
            for (int rb = 0; rb < RB0; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) di[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int bp = 0; bp < HB; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ks = 4 * bp + r;
#pragma unroll
                    for (int rb = 0; rb < RB0; ++rb) {
                        const float aw = imgI[(rb * KSH + ks) * 64 + lane];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) di[rb][ct] = LSE_MFMA(aw, dh[bp][ct][r], di[rb][ct]);
                    }
                }
#pragma unroll
            for (int rb = 0; rb < RB0; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (!valid[ct] || 16 * rb + 4 * q >= KIN) continue;
                    if (INL == LSE_IN_LEVELMAJOR) {
                        float2 *d2 = reinterpret_cast<float2 *>(a.d_in) + tile_base;
                        const int lv = 8 * rb + 2 * q;
                        d2[(int64_t)lv * ns + sl[ct]] = make_float2(di[rb][ct][0], di[rb][ct][1]);
                        d2[(int64_t)(lv + 1) * ns + sl[ct]] = make_float2(di[rb][ct][2], di[rb][ct][3]);
                    } else {
                        *reinterpret_cast<f32x4 *>(a.d_in + tile_base * KIN + (unsigned)(sl[ct] * KIN + 16 * rb + 4 * q)) = di[rb][ct];
                    }
                }
        }
    }
    if (BIAS_ONES) flush_bias_col();
    flush_wgrad<HB, KB0>(a.d_params + a.w0_col, a.w0_ld, KIN, acc0, j, q, a.w0_mask0 != 0);
    if constexpr (NHL == 2) flush_wgrad<HB, HB>(a.d_params + a.rest_off, WIDTH, WIDTH, acc1, j, q);
    flush_wgrad<1, HB>(a.d_params + a.rest_off + NP1, WIDTH, WIDTH, accO, j, q);
}

#include "mlp_x6.h"

// ------------------------------------------------------------------------------------------------------
// weight gradients:  dW[M x K] += G[N x M]^T * A[N x K]   (samples are the MFMA k dimension, so both
// operands are read straight from their row-major rows: lane (i, q) of k-step s reads row 4s+q, column i)
// ------------------------------------------------------------------------------------------------------
template <int M, int K, int AL>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float *__restrict__ G, const float *__restrict__ A,
                                                         int64_t n, float *__restrict__ dW, int dw_ld)
{
    constexpr int MB = M / 16, KB = (K + 15) / 16;
    // one private copy of the tile per wave, plain stores: ds_add_f32 costs ~3 cycles per lane on gfx950
    __shared__ float red[4][M * KB * 16];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    f32x4 acc[MB][KB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc[mb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t n_steps = (n + 3) / 4;
    const int64_t total_waves = (int64_t)gridDim.x * 4;
    const int64_t per = (n_steps + total_waves - 1) / total_waves;
    const int64_t w_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t s_lo = w_id * per, s_hi = min(n_steps, s_lo + per);
    constexpr int U = 2;
    for (int64_t st = s_lo; st < s_hi; st += U) {
        float av[U][MB], bv[U][KB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (st + u) * 4 + q;
            const bool ok = (st + u) < s_hi && row < n;
            const int64_t rr = ok ? row : 0;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float v = G[rr * M + 16 * mb + i];
                av[u][mb] = ok ? v : 0.f;
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int col = 16 * kb + i;
                float v = 0.f;
                if (col < K) {
                    if (AL == LSE_IN_LEVELMAJOR) v = A[((int64_t)(col >> 1) * n + rr) * 2 + (col & 1)];
                    else v = A[rr * K + col];
                }
                bv[u][kb] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc[mb][kb] = LSE_MFMA(av[u][mb], bv[u][kb], acc[mb][kb]);
    }
    // D: col = lane&15 -> A column, row = 4q + r -> G column
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(16 * mb + 4 * q + r) * (KB * 16) + 16 * kb + i] = acc[mb][kb][r];
    __syncthreads();
    for (int e = threadIdx.x; e < M * KB * 16; e += 256) {
        const int row = e / (KB * 16), col = e % (KB * 16);
        if (col < K) atomicAdd(&dW[row * dw_ld + col], (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]));
    }
}

// per-ray sums of rows[N, width] over packed segments: one wave per ray, lane = column
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float *__restrict__ rows, int width,
                                                               const int64_t *__restrict__ packed, int n_rays,
                                                               float *__restrict__ out)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    if (lane >= width) return;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int64_t k = 0;
    for (; k + 4 <= cnt; k += 4) {
        acc0 += rows[(s0 + k) * width + lane];
        acc1 += rows[(s0 + k + 1) * width + lane];
        acc2 += rows[(s0 + k + 2) * width + lane];
        acc3 += rows[(s0 + k + 3) * width + lane];
    }
    for (; k < cnt; ++k) acc0 += rows[(s0 + k) * width + lane];
    out[(int64_t)ray * width + lane] += (acc0 + acc1) + (acc2 + acc3);
}

// small dense helpers on per-ray matrices (R x <=64): one thread per output element
__global__ void linear_fwd_kernel(const float *__restrict__ w, const float *__restrict__ x, int rows, int m, int k,
                                  float *__restrict__ y)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)rows * m) return;
    const int r = e / m, o = e % m;
    float acc = 0.f;
    for (int c = 0; c < k; ++c) acc = fmaf(x[(int64_t)r * k + c], w[o * k + c], acc);
    y[e] = acc;
}

__global__ void linear_bwd_input_kernel(const float *__restrict__ w, const float *__restrict__ dy, int rows, int m,
                                        int k, float *__restrict__ dx)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)rows * k) return;
    const int r = e / k, c = e % k;
    float acc = 0.f;
    for (int o = 0; o < m; ++o) acc = fmaf(dy[(int64_t)r * m + o], w[o * k + c], acc);
    dx[e] = acc;
}

int check_desc(const lse_mlp_desc *d, const char *who)
{
    LSE_REQUIRE(d, "%s: null desc", who);
    LSE_REQUIRE(d->n_in == 8 || d->n_in == 16 || d->n_in == 32 || d->n_in == 64, "%s: n_in %d not in {8,16,32,64}",
                who, d->n_in);
    LSE_REQUIRE(d->width == 32 || d->width == 64, "%s: width %d not in {32,64}", who, d->width);
    LSE_REQUIRE(d->n_hidden_layers == 1 || d->n_hidden_layers == 2, "%s: n_hidden_layers %d not in {1,2}", who,
                d->n_hidden_layers);
    LSE_REQUIRE(d->in_layout == LSE_IN_ROWMAJOR || d->in_layout == LSE_IN_LEVELMAJOR, "%s: bad in_layout", who);
    LSE_REQUIRE(d->in_layout == LSE_IN_LEVELMAJOR || d->n_in >= 16, "%s: row-major input needs n_in >= 16", who);
    LSE_REQUIRE(d->out_activation == LSE_ACT_NONE || d->out_activation == LSE_ACT_SIGMOID, "%s: bad out_activation",
                who);
    LSE_REQUIRE(d->w0_ld == 0 || d->w0_ld >= d->w0_col + d->n_in, "%s: first-layer view [%d, %d) exceeds its leading dimension %d",
                who, d->w0_col, d->w0_col + d->n_in, d->w0_ld);
    LSE_REQUIRE(d->w0_col >= 0 && (d->w0_ld != 0 || d->w0_col == 0), "%s: w0_col needs w0_ld", who);
    LSE_REQUIRE(d->arith == LSE_MLP_ARITH_AUTO || d->arith == LSE_MLP_ARITH_F32_MFMA, "%s: bad arith", who);
    return LSE_OK;
}

void fill_view(MlpArgs &a, const lse_mlp_desc *d)
{
    a.w0_ld = d->w0_ld ? d->w0_ld : d->n_in;
    a.w0_col = d->w0_col;
    a.w0_mask0 = d->w0_mask_col0;
    a.arith = d->arith;
    a.rest_off = (int64_t)d->width * a.w0_ld;
}

template <int KIN, int WIDTH, int NHL, int INL, int CT, int NW>
int launch_fwd_cfg(const MlpArgs &a, hipStream_t st)
{
    constexpr int HB = WIDTH / 16;
    constexpr int imgs = HB * (KIN / 4) + (NHL == 2 ? HB * (WIDTH / 4) : 0) + WIDTH / 4;
    const int64_t tiles = (a.n + 16 * CT - 1) / (16 * CT);
    const int blocks = (int)std::min<int64_t>((tiles + NW - 1) / NW, 2048);
    hipLaunchKernelGGL((mlp_fwd_kernel<KIN, WIDTH, NHL, INL, CT, NW>), dim3(blocks), dim3(64 * NW), imgs * 256, st, a);
    return lse::check_launch("lse_mlp_fwd");
}

template <int KIN, int WIDTH, int NHL, int INL>
int launch_fwd(const MlpArgs &a, hipStream_t st)
{
    const int cfg = (int)lse::option("mlp_fwd_cfg");   // CT*10 + NW
    if constexpr (WIDTH == 64 && ((KIN == 16 && INL == LSE_IN_ROWMAJOR) || KIN == 32)) {
        // third generation: f32-equivalent arithmetic on the bf16 matrix cores (mlp_x6.h)
        if (cfg == 28 && (a.act == nullptr || a.act_tiled) && a.arith == LSE_MLP_ARITH_AUTO) {
            const int64_t tiles3 = (a.n + 31) / 32;
            const int blocks3 = (int)std::min<int64_t>((tiles3 + 7) / 8, 512);
            constexpr int lds3 = X6Fwd<KIN, NHL, INL>::lds_bytes;
            hipLaunchKernelGGL((mlp_fwd3_kernel<KIN, NHL, INL>), dim3(blocks3), dim3(512), lds3, st, a,
                               lse::option("mlp_act_nt") != 0);
            return lse::check_launch("lse_mlp_fwd");
        }
    }
    if (cfg == 28 && (a.act == nullptr || a.act_tiled)) {
        constexpr int HB2 = WIDTH / 16;
        constexpr int imgs2 = HB2 * (KIN / 4) + (NHL == 2 ? HB2 * (WIDTH / 4) : 0) + WIDTH / 4;
        const int64_t tiles2 = (a.n + 31) / 32;
        const int blocks2 = (int)std::min<int64_t>((tiles2 + 7) / 8, 512);      // two resident workgroups per CU
        hipLaunchKernelGGL((mlp_fwd2_kernel<KIN, WIDTH, NHL, INL>), dim3(blocks2), dim3(512), imgs2 * 256, st, a,
                           lse::option("mlp_act_nt") != 0);
        return lse::check_launch("lse_mlp_fwd");
    }
    if (a.act && a.act_tiled == 2) {
        lse::set_error("lse_mlp_fwd: act_tiled = 2 is only understood by the second-generation fused forward");
        return LSE_E_UNSUPPORTED;
    }
#ifdef LSE_DEV_KNOBS
    if (cfg == 44) return launch_fwd_cfg<KIN, WIDTH, NHL, INL, 4, 4>(a, st);     // (216 / 116 / 24 were measured and dropped in round 1)
#endif
    return launch_fwd_cfg<KIN, WIDTH, NHL, INL, 2, 8>(a, st);      // row-major saved activations (what lse_mlp_wgrad reads)
}

template <int KIN, int WIDTH, int NHL, int INL, int CT, int NW>
int launch_bwd_cfg(const MlpArgs &a, hipStream_t st)
{
    constexpr int HB = WIDTH / 16;
    constexpr int imgs = HB * 4 + (NHL == 2 ? HB * (WIDTH / 4) : 0) + ((KIN + 15) / 16) * (WIDTH / 4);
    constexpr int n_params = WIDTH * KIN + (NHL - 1) * WIDTH * WIDTH + 16 * WIDTH;
    const int64_t tiles = (a.n + 16 * CT - 1) / (16 * CT);
    if (a.d_params) {
        // one resident workgroup per CU (512-register waves); every wave flushes its dW accumulators once at the end
        const int blocks = (int)std::min<int64_t>((tiles + NW - 1) / NW, 256);
        const size_t lds_bytes = imgs * 256 + (size_t)(NW * kTrWave) * sizeof(float);
        (void)n_params;
        static std::atomic<uint64_t> lds_allowed{0};   // per instantiation: devices that allow more than the default 64 KiB of dynamic LDS
        if (lds_bytes > 64 * 1024) {
            const int rc_lds = lse::allow_dynamic_lds(reinterpret_cast<const void *>(&mlp_bwd_kernel<KIN, WIDTH, NHL, INL, true, CT, NW>), (int)lds_bytes, lds_allowed,
                                                      "lse_mlp_bwd");
            if (rc_lds) return rc_lds;
        }
        hipLaunchKernelGGL((mlp_bwd_kernel<KIN, WIDTH, NHL, INL, true, CT, NW>), dim3(blocks), dim3(64 * NW), lds_bytes, st, a);
    } else {
        const int blocks = (int)std::min<int64_t>((tiles + NW - 1) / NW, 2048);
        hipLaunchKernelGGL((mlp_bwd_kernel<KIN, WIDTH, NHL, INL, false, CT, NW>), dim3(blocks), dim3(64 * NW), imgs * 256, st, a);
    }
    return lse::check_launch("lse_mlp_bwd");
}

template <int KIN, int WIDTH, int NHL, int INL, bool BIAS_ONES, bool RECOMP = false>
int launch_bwd2(const MlpArgs &a, hipStream_t st)
{
    constexpr int HB = WIDTH / 16, NW = 8;
    constexpr int imgs = HB * 4 + (NHL == 2 ? HB * (WIDTH / 4) : 0) + ((KIN + 15) / 16) * (WIDTH / 4) + (RECOMP ? HB * (KIN / 4) : 0);
    const int64_t tiles = (a.n + 31) / 32;
    const int blocks = (int)std::min<int64_t>((tiles + NW - 1) / NW, 256);     // one resident workgroup per CU
    const size_t lds_bytes = imgs * 256 + (size_t)(NW * kTrWave) * sizeof(float);
    static std::atomic<uint64_t> lds_allowed{0};
    if (lds_bytes > 64 * 1024) {
        const int rc_lds = lse::allow_dynamic_lds(reinterpret_cast<const void *>(&mlp_bwd2_kernel<KIN, WIDTH, NHL, INL, BIAS_ONES, RECOMP>), (int)lds_bytes, lds_allowed, "lse_mlp_bwd");
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL((mlp_bwd2_kernel<KIN, WIDTH, NHL, INL, BIAS_ONES, RECOMP>), dim3(blocks), dim3(64 * NW), lds_bytes, st, a);
    return lse::check_launch("lse_mlp_bwd");
}

template <int KIN, int NHL, int INL, bool BIAS, bool BIAS_ONES, int CT, int NW>
int launch_bwd3_cfg(const MlpArgs &a, hipStream_t st)
{
    constexpr int lds_bytes = X6Bwd<KIN, NHL, CT, NW>::lds_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget of one CU");
    const int64_t tiles = (a.n + 16 * CT - 1) / (16 * CT);
    const int blocks = (int)std::min<int64_t>((tiles + NW - 1) / NW, 256);     // one resident workgroup per CU
    static std::atomic<uint64_t> lds_allowed{0};      // (per instantiation) devices of this process that allow lds_bytes of dynamic LDS
    const int rc_lds = lse::allow_dynamic_lds(reinterpret_cast<const void *>(&mlp_bwd3_kernel<KIN, NHL, INL, BIAS, BIAS_ONES, CT, NW>), lds_bytes, lds_allowed, "lse_mlp_bwd");
    if (rc_lds) return rc_lds;
    hipLaunchKernelGGL((mlp_bwd3_kernel<KIN, NHL, INL, BIAS, BIAS_ONES, CT, NW>), dim3(blocks), dim3(64 * NW), lds_bytes, st, a);
    return lse::check_launch("lse_mlp_bwd");
}

template <int KIN, int NHL, int INL, bool BIAS, bool BIAS_ONES>
int launch_bwd3(const MlpArgs &a, hipStream_t st)
{
#ifdef LSE_DEV_KNOBS
    switch ((int)lse::option("mlp_bwd3_cfg")) {      // CT * 100 + NW
    case 112: return launch_bwd3_cfg<KIN, NHL, INL, BIAS, BIAS_ONES, 1, 12>(a, st);
    case 108: return launch_bwd3_cfg<KIN, NHL, INL, BIAS, BIAS_ONES, 1, 8>(a, st);
    case 204: return launch_bwd3_cfg<KIN, NHL, INL, BIAS, BIAS_ONES, 2, 4>(a, st);      // one wave per SIMD, 512 registers, free scheduling
    default: break;
    }
#endif
    return launch_bwd3_cfg<KIN, NHL, INL, BIAS, BIAS_ONES, 2, 8>(a, st);
}

template <int KIN, int WIDTH, int NHL, int INL>
int launch_bwd(const MlpArgs &a, hipStream_t st)
{
    const int cfg = (int)lse::option("mlp_bwd_cfg");   // CT*10 + NW
    if (a.act_tiled == 3) {     // nothing saved: third generation, hidden layers recomputed on the bf16 matrix cores (mlp_x6.h)
        if constexpr (WIDTH == 64 && ((KIN == 16 && NHL == 2 && INL == LSE_IN_ROWMAJOR) || (KIN == 32 && NHL == 1))) {
            if (a.d_params && !a.d_out_pre && !a.d_act && !a.d_act0) {
                if (a.row_bias && a.d_row_bias && a.w0_mask0) return launch_bwd3<KIN, NHL, INL, true, true>(a, st);
                if (a.row_bias) return launch_bwd3<KIN, NHL, INL, true, false>(a, st);
                return launch_bwd3<KIN, NHL, INL, false, false>(a, st);
            }
        }
        lse::set_error("lse_mlp_bwd: act_tiled = 3 (all hidden layers recomputed) is built for the 16->64->64 row-major and "
                       "32->64 shapes with fused weight gradients and no materialised activation gradients");
        return LSE_E_UNSUPPORTED;
    }
    // second-generation kernel: fused weight gradients on tile-major activations, the default tile shape
    if (a.d_params && a.act_tiled && cfg == 28 && !a.d_out_pre && !a.d_act && !a.d_act0 && lse::option("mlp_bwd_impl") == 1) {
        if constexpr (NHL == 2 && INL == LSE_IN_ROWMAJOR && KIN % 16 == 0) {
            if (a.act_tiled == 2) {
                if (a.d_row_bias && a.w0_mask0) return launch_bwd2<KIN, WIDTH, NHL, INL, true, true>(a, st);
                return launch_bwd2<KIN, WIDTH, NHL, INL, false, true>(a, st);
            }
        }
        if (a.act_tiled == 2) {
            lse::set_error("lse_mlp_bwd: act_tiled = 2 (first hidden layer recomputed) needs two hidden layers and a row-major input");
            return LSE_E_UNSUPPORTED;
        }
        if (a.d_row_bias && a.w0_mask0) return launch_bwd2<KIN, WIDTH, NHL, INL, true>(a, st);
        return launch_bwd2<KIN, WIDTH, NHL, INL, false>(a, st);
    }
    if (a.act_tiled == 2) {
        lse::set_error("lse_mlp_bwd: act_tiled = 2 is only understood by the second-generation fused backward");
        return LSE_E_UNSUPPORTED;
    }
#ifdef LSE_DEV_KNOBS
    if (cfg == 44) return launch_bwd_cfg<KIN, WIDTH, NHL, INL, 4, 4>(a, st);
#endif
    // the generic kernel: data gradients alone (frozen parameters), materialised activation gradients, row-major activations
    return launch_bwd_cfg<KIN, WIDTH, NHL, INL, 2, 8>(a, st);
}

#define LSE_MLP_DISPATCH(FN, d, a, st)                                                                   \
    do {                                                                                                 \
        const int key = (d)->n_in * 1000 + (d)->width * 10 + (d)->n_hidden_layers;                        \
        if ((d)->in_layout == LSE_IN_LEVELMAJOR) {                                                       \
            switch (key) {                                                                               \
            case 32641: return FN<32, 64, 1, LSE_IN_LEVELMAJOR>(a, st);                                  \
            case 32642: return FN<32, 64, 2, LSE_IN_LEVELMAJOR>(a, st);                                  \
            case 8321: return FN<8, 32, 1, LSE_IN_LEVELMAJOR>(a, st);                                    \
            case 8641: return FN<8, 64, 1, LSE_IN_LEVELMAJOR>(a, st);                                    \
            case 32321: return FN<32, 32, 1, LSE_IN_LEVELMAJOR>(a, st);                                  \
            default: break;                                                                              \
            }                                                                                            \
        } else {                                                                                         \
            switch (key) {                                                                               \
            case 16642: return FN<16, 64, 2, LSE_IN_ROWMAJOR>(a, st);                                    \
            case 16641: return FN<16, 64, 1, LSE_IN_ROWMAJOR>(a, st);                                    \
            case 16322: return FN<16, 32, 2, LSE_IN_ROWMAJOR>(a, st);                                    \
            case 16321: return FN<16, 32, 1, LSE_IN_ROWMAJOR>(a, st);                                    \
            case 32641: return FN<32, 64, 1, LSE_IN_ROWMAJOR>(a, st);                                    \
            case 64642: return FN<64, 64, 2, LSE_IN_ROWMAJOR>(a, st);                                    \
            default: break;                                                                              \
            }                                                                                            \
        }                                                                                                \
        lse::set_error("mlp: no kernel instance for n_in=%d width=%d n_hidden_layers=%d in_layout=%d",   \
                       (d)->n_in, (d)->width, (d)->n_hidden_layers, (d)->in_layout);                      \
        return LSE_E_UNSUPPORTED;                                                                        \
    } while (0)

template <int M, int K, int AL>
int launch_gemm_tn(const float *g, const float *a, int64_t n, float *dw, int dw_ld, hipStream_t st)
{
    const int64_t steps = (n + 3) / 4;
    // >= 16 k-steps (64 rows) per wave; small row counts (per-ray matrices: 4096 rows) still spread over 16 workgroups
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((steps + 63) / 64, 1024));
    hipLaunchKernelGGL((gemm_tn_kernel<M, K, AL>), dim3(blocks), dim3(256), 0, st, g, a, n, dw, dw_ld);
    return lse::check_launch("lse_gemm_tn_acc");
}

int gemm_tn_dispatch(const float *g, int m, const float *a, int k, int al, int64_t n, float *dw, int dw_ld,
                     hipStream_t st)
{
    if (n == 0) return LSE_OK;
#define CASE(MM, KK, LL) \
    if (m == MM && k == KK && al == LL) return launch_gemm_tn<MM, KK, LL>(g, a, n, dw, dw_ld, st)
    CASE(64, 32, LSE_IN_LEVELMAJOR);
    CASE(64, 8, LSE_IN_LEVELMAJOR);
    CASE(32, 8, LSE_IN_LEVELMAJOR);
    CASE(32, 32, LSE_IN_LEVELMAJOR);
    CASE(64, 16, LSE_IN_ROWMAJOR);
    CASE(64, 32, LSE_IN_ROWMAJOR);
    CASE(64, 64, LSE_IN_ROWMAJOR);
    CASE(32, 16, LSE_IN_ROWMAJOR);
    CASE(32, 64, LSE_IN_ROWMAJOR);
    CASE(32, 32, LSE_IN_ROWMAJOR);
    CASE(16, 64, LSE_IN_ROWMAJOR);
    CASE(16, 32, LSE_IN_ROWMAJOR);
#undef CASE
    lse::set_error("lse_gemm_tn_acc: no kernel instance for m=%d k=%d layout=%d", m, k, al);
    return LSE_E_UNSUPPORTED;
}

}  // namespace

extern "C" int lse_mlp_fwd(const lse_mlp_desc *desc, const float *params, const float *in, const float *row_bias,
                           const int32_t *row_bias_idx, float *out, int32_t out_cols, float *act, int32_t act_tiled,
                           float *sigma_out, const uint8_t *selector, float density_scale, int64_t n, const int64_t *n_dev,
                           lse_stream_t stream)
{
    int rc = check_desc(desc, "lse_mlp_fwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_mlp_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(params && in && out, "lse_mlp_fwd: null pointer");
    MlpArgs a{};
    LSE_REQUIRE(out_cols == 16 || out_cols == 4, "lse_mlp_fwd: out_cols must be 16 or 4");
    a.params = params; a.in = in; a.row_bias = row_bias; a.row_bias_idx = row_bias_idx; a.out = out; a.act = act;
    a.n = n; a.n_stride = n; a.n_dev = n_dev;
    a.out_activation = desc->out_activation; a.out_cols = out_cols; a.sigma_out = sigma_out;
    a.selector = selector; a.density_scale = density_scale;
    a.act_tiled = act_tiled;
    a.act_layer_stride = act_tiled ? ((n + 15) / 16 * 16) * (int64_t)desc->width : n * (int64_t)desc->width;
    fill_view(a, desc);
    hipStream_t st = lse::as_stream(stream);
    LSE_MLP_DISPATCH(launch_fwd, desc, a, st);
}

extern "C" int lse_mlp_bwd(const lse_mlp_desc *desc, const float *params, const float *in, const float *act,
                           int32_t act_tiled, const float *out, int32_t out_cols, const float *d_out, const float *d_sigma,
                           const uint8_t *selector, float density_scale, float *d_out_pre, float *d_act, float *d_act0,
                           float *d_in, float *d_params, const float *row_bias, const int32_t *row_bias_idx,
                           float *d_row_bias, int64_t n, const int64_t *n_dev, lse_stream_t stream)
{
    int rc = check_desc(desc, "lse_mlp_bwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_mlp_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(params && d_out && (act || act_tiled == 3), "lse_mlp_bwd: null pointer");
    LSE_REQUIRE(desc->out_activation == LSE_ACT_NONE || out, "lse_mlp_bwd: sigmoid backward needs `out`");
    LSE_REQUIRE(!d_params || in, "lse_mlp_bwd: fused weight gradients need the layer-0 input `in`");
    LSE_REQUIRE(out_cols == 16 || out_cols == 4, "lse_mlp_bwd: out_cols must be 16 or 4");
    LSE_REQUIRE(!d_sigma || out, "lse_mlp_bwd: the density gradient needs `out`");
    LSE_REQUIRE(!d_row_bias || row_bias_idx, "lse_mlp_bwd: d_row_bias needs row_bias_idx (sorted rows)");
    LSE_REQUIRE(act_tiled >= 0 && act_tiled <= 3, "lse_mlp_bwd: act_tiled must be 0, 1, 2 or 3");
    LSE_REQUIRE(out_cols == 16 || !d_out_pre, "lse_mlp_bwd: d_out_pre needs the padded 16-column layout");
    MlpArgs a{};
    a.params = params; a.in = in; a.act = const_cast<float *>(act); a.out = const_cast<float *>(out); a.d_out = d_out;
    a.d_out_pre = d_out_pre; a.d_act = d_act; a.d_act0 = d_act0; a.d_in = d_in; a.d_params = d_params; a.n = n;
    a.n_stride = n; a.n_dev = n_dev;
    a.out_activation = desc->out_activation; a.out_cols = out_cols; a.d_sigma = d_sigma; a.selector = selector;
    a.density_scale = density_scale;
    a.act_tiled = act_tiled;
    a.act_layer_stride = act_tiled ? ((n + 15) / 16 * 16) * (int64_t)desc->width : n * (int64_t)desc->width;
    a.row_bias = row_bias; a.row_bias_idx = row_bias_idx; a.d_row_bias = d_row_bias;
    LSE_REQUIRE(act_tiled < 2 || in, "lse_mlp_bwd: act_tiled = 2 / 3 recompute hidden layers and need `in`");
    fill_view(a, desc);
    hipStream_t st = lse::as_stream(stream);
    LSE_MLP_DISPATCH(launch_bwd, desc, a, st);
}

extern "C" int lse_gemm_tn_acc(const float *g, int32_t m, const float *a, int32_t k, int32_t a_layout, int64_t n,
                               float *dw, int32_t dw_ld, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_gemm_tn_acc: n < 0");
    LSE_REQUIRE(n == 0 || (g && a && dw), "lse_gemm_tn_acc: null pointer");
    LSE_REQUIRE(dw_ld >= k, "lse_gemm_tn_acc: dw_ld < k");
    return gemm_tn_dispatch(g, m, a, k, a_layout, n, dw, dw_ld, lse::as_stream(stream));
}

extern "C" int lse_mlp_wgrad(const lse_mlp_desc *desc, const float *in, const float *act, const float *d_act,
                             const float *d_out_pre, float *d_params, int64_t n, lse_stream_t stream)
{
    int rc = check_desc(desc, "lse_mlp_wgrad");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_mlp_wgrad: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(in && act && d_act && d_out_pre && d_params, "lse_mlp_wgrad: null pointer");
    LSE_REQUIRE((desc->w0_ld == 0 || desc->w0_ld == desc->n_in) && !desc->w0_mask_col0,
                "lse_mlp_wgrad: first-layer views are only supported by the fused lse_mlp_bwd");
    hipStream_t st = lse::as_stream(stream);
    const int W = desc->width, K0 = desc->n_in, NHL = desc->n_hidden_layers;
    float *dW0 = d_params, *dW1 = dW0 + W * K0, *dWo = dW1 + (NHL - 1) * W * W;
    rc = gemm_tn_dispatch(d_act, W, in, K0, desc->in_layout, n, dW0, K0, st);
    if (rc) return rc;
    if (NHL == 2) {
        rc = gemm_tn_dispatch(d_act + n * W, W, act, W, LSE_IN_ROWMAJOR, n, dW1, W, st);
        if (rc) return rc;
    }
    return gemm_tn_dispatch(d_out_pre, 16, act + (int64_t)(NHL - 1) * n * W, W, LSE_IN_ROWMAJOR, n, dWo, W, st);
}

extern "C" int lse_segment_sum_rows(const float *rows, int32_t width, const int64_t *packed_info, int32_t n_rays,
                                    float *out, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0 && width >= 1 && width <= 64, "lse_segment_sum_rows: width must be in [1,64]");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rows && packed_info && out, "lse_segment_sum_rows: null pointer");
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), rows,
                       width, packed_info, n_rays, out);
    return lse::check_launch("lse_segment_sum_rows");
}

extern "C" int lse_linear_fwd(const float *w, const float *x, int32_t rows, int32_t m, int32_t k, float *y,
                              lse_stream_t stream)
{
    LSE_REQUIRE(rows >= 0 && m > 0 && k > 0, "lse_linear_fwd: bad shape");
    if (rows == 0) return LSE_OK;
    LSE_REQUIRE(w && x && y, "lse_linear_fwd: null pointer");
    const int64_t tot = (int64_t)rows * m;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, lse::as_stream(stream), w, x,
                       rows, m, k, y);
    return lse::check_launch("lse_linear_fwd");
}

extern "C" int lse_linear_bwd_input(const float *w, const float *dy, int32_t rows, int32_t m, int32_t k, float *dx,
                                    lse_stream_t stream)
{
    LSE_REQUIRE(rows >= 0 && m > 0 && k > 0, "lse_linear_bwd_input: bad shape");
    if (rows == 0) return LSE_OK;
    LSE_REQUIRE(w && dy && dx, "lse_linear_bwd_input: null pointer");
    const int64_t tot = (int64_t)rows * k;
    hipLaunchKernelGGL(linear_bwd_input_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       lse::as_stream(stream), w, dy, rows, m, k, dx);
    return lse::check_launch("lse_linear_bwd_input");
}
