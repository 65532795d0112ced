// Streaming pieces around the hash grid and the MLPs (all O(N) or O(R), HBM-bound, one pass each):
//   positions / L-inf contraction / selector        R:lse_nerf/lse_field.py:264-274
//   trunc_exp density activation                    R:lse_nerf/lse_field.py:286-287
//   per-ray head features: tcnn SH degree 4 of the (d+1)/2-shifted direction + appearance embedding
//                                                   R:lse_nerf/lse_field.py:298-310, 347-356
// Directions and appearance ids are per RAY, so everything that depends only on them (SH, embedding, the
// constant-1 padding column) is evaluated once per ray and enters the head MLP as a per-ray layer-0 bias.
#include "common.h"

namespace {

struct Box { float lo[3], hi[3]; };

__device__ __forceinline__ void sample_pos(const float *__restrict__ o, const float *__restrict__ d,
                                           const int32_t *__restrict__ ri, const float *__restrict__ ts,
                                           const float *__restrict__ te, int64_t i, float p[3], float &tmid2)
{
    if (ri) {
        const int64_t r = ri[i];
        const float s = ts[i] + te[i];
        tmid2 = s;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[r * 3 + k] + d[r * 3 + k] * s / 2.f;   // origins + directions*(starts+ends)/2
    } else {
        tmid2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[i * 3 + k];
    }
}

__global__ __launch_bounds__(256) void positions_fwd_kernel(const float *__restrict__ o, const float *__restrict__ d,
                                                            const int32_t *__restrict__ ri, const float *__restrict__ ts,
                                                            const float *__restrict__ te, int64_t n, int contraction,
                                                            Box box, float *__restrict__ x01, uint8_t *__restrict__ sel,
                                                            const int64_t *__restrict__ n_dev)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lse::clamp_count(n, n_dev)) return;
    float p[3], t2;
    sample_pos(o, d, ri, ts, te, i, p, t2);
    float x[3];
    if (contraction) {
        const float mag = fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2])));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = mag < 1.f ? p[k] : (2.f - (1.f / mag)) * (p[k] / mag);
            x[k] = (c + 2.f) / 4.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = (p[k] - box.lo[k]) / (box.hi[k] - box.lo[k]);
    }
    const bool s = x[0] > 0.f && x[0] < 1.f && x[1] > 0.f && x[1] < 1.f && x[2] > 0.f && x[2] < 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[i * 3 + k] = s ? x[k] : 0.f;
    sel[i] = s ? 1 : 0;
}

__global__ __launch_bounds__(256) void positions_bwd_kernel(const float *__restrict__ o, const float *__restrict__ d,
                                                            const int32_t *__restrict__ ri, const float *__restrict__ ts,
                                                            const float *__restrict__ te, int64_t n, int contraction,
                                                            Box box, const float *__restrict__ dx01,
                                                            float *__restrict__ dpos, const int64_t *__restrict__ n_dev)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lse::clamp_count(n, n_dev)) return;
    float p[3], t2;
    sample_pos(o, d, ri, ts, te, i, p, t2);
    float g[3] = {dx01[i * 3 + 0], dx01[i * 3 + 1], dx01[i * 3 + 2]};
    float x[3], out[3];
    if (contraction) {
        const float a0 = fabsf(p[0]), a1 = fabsf(p[1]), a2 = fabsf(p[2]);
        const float mag = fmaxf(a0, fmaxf(a1, a2));
        if (mag < 1.f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { x[k] = (p[k] + 2.f) / 4.f; out[k] = g[k] * 0.25f; }
        } else {
            // y = s(m) p, s = 2/m - 1/m^2, m = |p_a| (a = argmax):  dy_i/dp_j = s delta_ij + p_i s'(m) sign(p_a) delta_ja
            const int am = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
            const float inv = 1.f / mag;
            const float s = (2.f - inv) * inv;
            const float ds = (-2.f * inv * inv) + (2.f * inv * inv * inv);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { x[k] = (s * p[k] + 2.f) / 4.f; dot += g[k] * p[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = 0.25f * s * g[k];
            out[am] += 0.25f * dot * ds * (p[am] >= 0.f ? 1.f : -1.f);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[k] = (p[k] - box.lo[k]) / (box.hi[k] - box.lo[k]);
            out[k] = g[k] / (box.hi[k] - box.lo[k]);
        }
    }
    const bool s = x[0] > 0.f && x[0] < 1.f && x[1] > 0.f && x[1] < 1.f && x[2] > 0.f && x[2] < 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) dpos[i * 3 + k] = s ? out[k] : 0.f;
}

// one wave per ray: d_o = sum dpos, d_d = sum dpos * (ts+te)/2
__global__ __launch_bounds__(256) void ray_grad_kernel(const float *__restrict__ dpos, const float *__restrict__ ts,
                                                       const float *__restrict__ te, const int64_t *__restrict__ packed,
                                                       int n_rays, float *__restrict__ d_o, float *__restrict__ d_d)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = packed[2 * ray], cnt = packed[2 * ray + 1];
    float ao[3] = {0.f, 0.f, 0.f}, ad[3] = {0.f, 0.f, 0.f};
    for (int64_t k = lane; k < cnt; k += 64) {
        const int64_t i = s0 + k;
        const float tm = (ts[i] + te[i]) * 0.5f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = dpos[i * 3 + c];
            ao[c] += g;
            ad[c] = fmaf(g, tm, ad[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { ao[c] = lse::wave_sum(ao[c]); ad[c] = lse::wave_sum(ad[c]); }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (d_o) d_o[ray * 3 + c] = ao[c];
            if (d_d) d_d[ray * 3 + c] = ad[c];
        }
    }
}

__global__ __launch_bounds__(256) void density_fwd_kernel(const float *__restrict__ h, const uint8_t *__restrict__ sel,
                                                          float scale, float *__restrict__ sigma, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = scale * expf(h[i * 16]);
    sigma[i] = (sel == nullptr || sel[i]) ? v : 0.f;
}

__global__ __launch_bounds__(256) void density_bwd_kernel(const float *__restrict__ h, const uint8_t *__restrict__ sel,
                                                          float scale, const float *__restrict__ dsigma,
                                                          float *__restrict__ dh, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = fminf(fmaxf(h[i * 16], -15.f), 15.f);
    const float g = dsigma[i] * scale * expf(x);
    dh[i * 16] = (sel == nullptr || sel[i]) ? g : 0.f;
}

// ---- per-ray head features [SH16 | 15 zeros | emb | 1] -------------------------------------------------
__device__ __forceinline__ void sh4(float x, float y, float z, float *o)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// gradient of sum_k g[k] * sh4(x, y, z)[k] w.r.t. (x, y, z)
__device__ __forceinline__ void sh4_grad(float x, float y, float z, const float *g, float &gx, float &gy, float &gz)
{
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        const float c1 = 0.48860251190291987f, c4 = 1.0925484305920792f, c6 = 0.94617469575755997f,
                    c8 = 0.54627421529603959f, c9 = 0.59004358992664352f, c10 = 2.8906114426405538f,
                    c11 = 0.45704579946446572f, c12 = 0.3731763325901154f, c14 = 1.4453057213202769f;
    gx = 0.f; gy = 0.f; gz = 0.f;
        gy += g[1] * -c1;
        gz += g[2] * c1;
        gx += g[3] * -c1;
        gx += g[4] * c4 * y;            gy += g[4] * c4 * x;
        gy += g[5] * -c4 * z;           gz += g[5] * -c4 * y;
        gz += g[6] * 2.f * c6 * z;
        gx += g[7] * -c4 * z;           gz += g[7] * -c4 * x;
        gx += g[8] * 2.f * c8 * x;      gy += g[8] * -2.f * c8 * y;
        gx += g[9] * c9 * y * -6.f * x; gy += g[9] * c9 * (-3.f * x2 + 3.f * y2);
        gx += g[10] * c10 * y * z;      gy += g[10] * c10 * x * z;     gz += g[10] * c10 * x * y;
        gy += g[11] * c11 * (1.f - 5.f * z2);                          gz += g[11] * c11 * y * -10.f * z;
        gz += g[12] * c12 * (15.f * z2 - 3.f);
        gx += g[13] * c11 * (1.f - 5.f * z2);                          gz += g[13] * c11 * x * -10.f * z;
        gx += g[14] * c14 * z * 2.f * x; gy += g[14] * c14 * z * -2.f * y; gz += g[14] * c14 * (x2 - y2);
        gx += g[15] * c9 * (-3.f * x2 + 3.f * y2);                     gy += g[15] * c9 * x * 6.f * y;
}

__global__ __launch_bounds__(256) void ray_features_fwd_kernel(const float *__restrict__ dirs,
                                                               const float *__restrict__ emb,
                                                               const int32_t *__restrict__ eidx, int n_rays, int emb_dim,
                                                               float *__restrict__ feat)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float *f = feat + (int64_t)r * 64;
    // shift_directions_for_tcnn then tcnn's own 2x-1 re-centring, both in f32
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = ((dirs[r * 3 + k] + 1.f) / 2.f) * 2.f - 1.f;
    float sh[16];
    sh4(v[0], v[1], v[2], sh);
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = sh[k];
    for (int k = 16; k < 31; ++k) f[k] = 0.f;
    const int64_t e = (emb && eidx) ? eidx[r] : 0;
    for (int k = 0; k < 32; ++k) f[31 + k] = (emb && k < emb_dim) ? emb[e * emb_dim + k] : 0.f;
    // input padding value of tcnn's Identity encoding is 1 -> the 64th weight column acts as a bias.
    // With emb_dim < 32 the padded width changes (63 -> 47 -> pad 48); the host packs columns accordingly.
    f[63] = 1.f;
}

__global__ __launch_bounds__(256) void ray_features_bwd_kernel(const float *__restrict__ dirs,
                                                               const float *__restrict__ dfeat,
                                                               const int32_t *__restrict__ eidx, int n_rays, int emb_dim,
                                                               float *__restrict__ d_dirs, float *__restrict__ d_emb)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float *g = dfeat + (int64_t)r * 64;
    if (d_dirs) {
        const float x = ((dirs[r * 3 + 0] + 1.f) / 2.f) * 2.f - 1.f;
        const float y = ((dirs[r * 3 + 1] + 1.f) / 2.f) * 2.f - 1.f;
        const float z = ((dirs[r * 3 + 2] + 1.f) / 2.f) * 2.f - 1.f;
        float gx, gy, gz;
        sh4_grad(x, y, z, g, gx, gy, gz);
        // d v / d dir = (1/2)*2 = 1
        d_dirs[r * 3 + 0] = gx; d_dirs[r * 3 + 1] = gy; d_dirs[r * 3 + 2] = gz;
    }
}

// d(embedding row e) = sum over the rays that use row e.  All samples of a ray share one row and a batch uses only a
// handful of rows, so atomics would pile onto a few addresses (14x slower per MI355X_MICROARCH.md "contention");
// instead one workgroup per embedding row scans the ray list (R is a few thousand) and reduces in LDS.
__global__ __launch_bounds__(256) void emb_grad_kernel(const float *__restrict__ dfeat, const int32_t *__restrict__ eidx,
                                                       int n_rays, int emb_dim, float *__restrict__ d_emb,
                                                       int feat_ld = 64)
{
    // thread (part, col): 8 ray-partitions x 32 columns; the ray-id test is wave-uniform per iteration (all 32 column
    // lanes of a partition look at the same ray), partial sums stay in a register, one LDS pass combines the partitions.
    // (embeddings wider than 32: blockIdx.z walks the 32-column blocks)
    __shared__ float red[8][32];
    const int e = blockIdx.x, lcol = threadIdx.x & 31, col = blockIdx.z * 32 + lcol, part = threadIdx.x >> 5;
    float acc = 0.f;
    // blockIdx.y splits the ray list further (more loads in flight chip-wide); partial sums meet in d_emb via atomics
    const int nsplit = 8 * gridDim.y;
    const int per = (n_rays + nsplit - 1) / nsplit;
    const int r0 = min(n_rays, (blockIdx.y * 8 + part) * per), r1 = min(n_rays, r0 + per);
    int r = r0;
    for (; r + 4 <= r1; r += 4) {   // 4 independent id loads in flight
        const int i0 = eidx[r], i1 = eidx[r + 1], i2 = eidx[r + 2], i3 = eidx[r + 3];
        if (col < emb_dim) {
            if (i0 == e) acc += dfeat[(int64_t)r * feat_ld + 31 + col];
            if (i1 == e) acc += dfeat[(int64_t)(r + 1) * feat_ld + 31 + col];
            if (i2 == e) acc += dfeat[(int64_t)(r + 2) * feat_ld + 31 + col];
            if (i3 == e) acc += dfeat[(int64_t)(r + 3) * feat_ld + 31 + col];
        }
    }
    for (; r < r1; ++r)
        if (eidx[r] == e && col < emb_dim) acc += dfeat[(int64_t)r * feat_ld + 31 + col];
    red[part][lcol] = acc;
    __syncthreads();
    if (part == 0 && col < emb_dim) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) s += red[p][lcol];
        atomicAdd(&d_emb[(int64_t)e * emb_dim + col], s);
    }
}

// ---- per-ray part of the head's first layer, fused ------------------------------------------------------------------
// tcnn feeds the head [SH16 | geo15 | emb | ones-padding] (R:lse_nerf/lse_field.py:347-356; padded to a multiple of 16 with
// ones).  Everything but the 15 geometry features depends only on the RAY, so it enters layer 0 as a per-ray bias
//     row_bias[r][o] = sum_c feat[r][c] * W_in[o][c],   feat[r] = [SH16(dir_r) | 0 x 15 | emb[idx_r] | 1 ...]  (in_pad wide)
// computed here in one launch (feature construction + the [R x in_pad] x [in_pad x W] product); feat is kept for the backward.
constexpr int kRbRays = 16;     // rays per workgroup

__device__ __forceinline__ void ray_feat(const float *dirs, const float *emb, const int32_t *eidx, int r, int emb_dim,
                                         int in_dim, int in_pad, float *f /* [in_pad], LDS or registers */)
{
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = ((dirs[r * 3 + k] + 1.f) / 2.f) * 2.f - 1.f;   // shift_directions_for_tcnn, tcnn's 2x-1
    float sh[16];
    sh4(v[0], v[1], v[2], sh);
    for (int k = 0; k < 16; ++k) f[k] = sh[k];
    for (int k = 16; k < 31; ++k) f[k] = 0.f;
    const int64_t e = (emb && eidx) ? eidx[r] : 0;
    for (int k = 0; k < emb_dim; ++k) f[31 + k] = emb ? emb[e * emb_dim + k] : 0.f;
    for (int k = in_dim; k < in_pad; ++k) f[k] = 1.f;                               // tcnn pads network inputs with ones
}

// PITCH = LDS row pitch: 65 for padded input widths up to 64 (the reference's 32-wide appearance embedding: 31 + 32 -> 64), 129 up to 128
template <int WIDTH, int PITCH>
__global__ __launch_bounds__(256) void ray_bias_fwd_kernel(const float *__restrict__ dirs, const float *__restrict__ emb,
                                                           const int32_t *__restrict__ eidx, int n_rays, int emb_dim,
                                                           int in_pad, const float *__restrict__ W /* [WIDTH][w_ld] */,
                                                           int w_ld, float *__restrict__ feat, float *__restrict__ row_bias)
{
    __shared__ float s_f[kRbRays][PITCH];
    __shared__ float s_w[WIDTH][PITCH];
    const int in_dim = 31 + emb_dim;
    const int r0 = blockIdx.x * kRbRays;
    for (int e = threadIdx.x; e < WIDTH * in_pad; e += 256) s_w[e / in_pad][e % in_pad] = W[(e / in_pad) * w_ld + e % in_pad];
    if (threadIdx.x < kRbRays && r0 + (int)threadIdx.x < n_rays)
        ray_feat(dirs, emb, eidx, r0 + threadIdx.x, emb_dim, in_dim, in_pad, s_f[threadIdx.x]);
    __syncthreads();
    for (int e = threadIdx.x; e < kRbRays * in_pad; e += 256) {
        const int rr = e / in_pad, c = e % in_pad;
        if (r0 + rr < n_rays) feat[(int64_t)(r0 + rr) * in_pad + c] = s_f[rr][c];
    }
    for (int e = threadIdx.x; e < kRbRays * WIDTH; e += 256) {
        const int rr = e / WIDTH, o = e % WIDTH;
        if (r0 + rr >= n_rays) continue;
        float acc = 0.f;
        for (int c = 0; c < in_pad; ++c) acc = fmaf(s_f[rr][c], s_w[o][c], acc);
        row_bias[(int64_t)(r0 + rr) * WIDTH + o] = acc;
    }
}

// d_feat = d_row_bias * W_in  ->  d(directions) through the SH Jacobian, and the embedding slice [R, 32] for emb_grad_kernel
template <int WIDTH, int PITCH>
__global__ __launch_bounds__(256) void ray_bias_bwd_kernel(const float *__restrict__ dirs, const float *__restrict__ d_rb,
                                                           int n_rays, int emb_dim, int in_pad,
                                                           const float *__restrict__ W, int w_ld,
                                                           float *__restrict__ d_feat /* [R][in_pad] */,
                                                           float *__restrict__ d_dirs /* nullable */)
{
    __shared__ float s_g[kRbRays][WIDTH + 1];
    __shared__ float s_w[WIDTH][PITCH];
    __shared__ float s_df[kRbRays][PITCH];
    const int r0 = blockIdx.x * kRbRays;
    for (int e = threadIdx.x; e < WIDTH * in_pad; e += 256) s_w[e / in_pad][e % in_pad] = W[(e / in_pad) * w_ld + e % in_pad];
    for (int e = threadIdx.x; e < kRbRays * WIDTH; e += 256) {
        const int rr = e / WIDTH, o = e % WIDTH;
        s_g[rr][o] = (r0 + rr < n_rays) ? d_rb[(int64_t)(r0 + rr) * WIDTH + o] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kRbRays * in_pad; e += 256) {
        const int rr = e / in_pad, c = e % in_pad;
        float acc = 0.f;
        for (int o = 0; o < WIDTH; ++o) acc = fmaf(s_g[rr][o], s_w[o][c], acc);
        s_df[rr][c] = acc;
        if (r0 + rr < n_rays) d_feat[(int64_t)(r0 + rr) * in_pad + c] = acc;
    }
    __syncthreads();
    if (d_dirs && threadIdx.x < kRbRays && r0 + (int)threadIdx.x < n_rays) {
        const int r = r0 + threadIdx.x;
        const float *g = s_df[threadIdx.x];
        const float x = ((dirs[r * 3 + 0] + 1.f) / 2.f) * 2.f - 1.f;
        const float y = ((dirs[r * 3 + 1] + 1.f) / 2.f) * 2.f - 1.f;
        const float z = ((dirs[r * 3 + 2] + 1.f) / 2.f) * 2.f - 1.f;
        float gx, gy, gz;
        sh4_grad(x, y, z, g, gx, gy, gz);
        d_dirs[r * 3 + 0] = gx; d_dirs[r * 3 + 1] = gy; d_dirs[r * 3 + 2] = gz;
    }
}

}  // namespace

static Box make_box(const float *h_aabb)
{
    Box b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = h_aabb ? h_aabb[k] : -1.f; b.hi[k] = h_aabb ? h_aabb[3 + k] : 1.f; }
    return b;
}

extern "C" int lse_positions_fwd(const float *rays_o, const float *rays_d, const int32_t *ray_idx,
                                 const float *t_starts, const float *t_ends, int64_t n, const int64_t *n_dev,
                                 int32_t contraction, const float *h_aabb, float *x01, uint8_t *selector, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_positions_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(rays_o && x01 && selector, "lse_positions_fwd: null pointer");
    LSE_REQUIRE(!ray_idx || (rays_d && t_starts && t_ends), "lse_positions_fwd: ray mode needs rays_d, t_starts, t_ends");
    LSE_REQUIRE(contraction || h_aabb, "lse_positions_fwd: aabb normalisation needs h_aabb");
    hipLaunchKernelGGL(positions_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lse::as_stream(stream),
                       rays_o, rays_d, ray_idx, t_starts, t_ends, n, contraction, make_box(h_aabb), x01, selector,
                       n_dev);
    return lse::check_launch("lse_positions_fwd");
}

extern "C" int lse_positions_bwd(const float *rays_o, const float *rays_d, const int32_t *ray_idx,
                                 const float *t_starts, const float *t_ends, int64_t n, const int64_t *n_dev,
                                 int32_t contraction, const float *h_aabb, const float *d_x01, float *d_pos, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_positions_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(rays_o && d_x01 && d_pos, "lse_positions_bwd: null pointer");
    LSE_REQUIRE(!ray_idx || (rays_d && t_starts && t_ends), "lse_positions_bwd: ray mode needs rays_d, t_starts, t_ends");
    LSE_REQUIRE(contraction || h_aabb, "lse_positions_bwd: aabb normalisation needs h_aabb");
    hipLaunchKernelGGL(positions_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lse::as_stream(stream),
                       rays_o, rays_d, ray_idx, t_starts, t_ends, n, contraction, make_box(h_aabb), d_x01, d_pos,
                       n_dev);
    return lse::check_launch("lse_positions_bwd");
}

extern "C" int lse_ray_grad_reduce(const float *d_pos, const float *t_starts, const float *t_ends,
                                   const int64_t *packed_info, int32_t n_rays, float *d_rays_o, float *d_rays_d,
                                   lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_grad_reduce: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(d_pos && t_starts && t_ends && packed_info, "lse_ray_grad_reduce: null pointer");
    hipLaunchKernelGGL(ray_grad_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, lse::as_stream(stream), d_pos, t_starts,
                       t_ends, packed_info, n_rays, d_rays_o, d_rays_d);
    return lse::check_launch("lse_ray_grad_reduce");
}

extern "C" int lse_density_fwd(const float *h, const uint8_t *selector, float scale, float *sigma, int64_t n,
                               lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_density_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(h && sigma, "lse_density_fwd: null pointer");
    hipLaunchKernelGGL(density_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lse::as_stream(stream), h,
                       selector, scale, sigma, n);
    return lse::check_launch("lse_density_fwd");
}

extern "C" int lse_density_bwd(const float *h, const uint8_t *selector, float scale, const float *d_sigma, float *d_h,
                               int64_t n, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_density_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(h && d_sigma && d_h, "lse_density_bwd: null pointer");
    hipLaunchKernelGGL(density_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lse::as_stream(stream), h,
                       selector, scale, d_sigma, d_h, n);
    return lse::check_launch("lse_density_bwd");
}

extern "C" int lse_ray_features_fwd(const float *rays_d, const float *emb_table, const int32_t *emb_idx, int32_t n_rays,
                                    int32_t emb_dim, float *feat, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_features_fwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_d && feat, "lse_ray_features_fwd: null pointer");
    LSE_REQUIRE(emb_dim == 0 || emb_dim == 32, "lse_ray_features_fwd: emb_dim must be 0 or 32 (got %d)", emb_dim);
    hipLaunchKernelGGL(ray_features_fwd_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, lse::as_stream(stream), rays_d,
                       emb_table, emb_idx, n_rays, emb_dim, feat);
    return lse::check_launch("lse_ray_features_fwd");
}

extern "C" int lse_ray_features_bwd(const float *rays_d, const float *d_feat, const int32_t *emb_idx, int32_t n_rays,
                                    int32_t emb_dim, int32_t n_emb_rows, float *d_rays_d, float *d_emb_table,
                                    lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_features_bwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_d && d_feat, "lse_ray_features_bwd: null pointer");
    LSE_REQUIRE(emb_dim == 0 || emb_dim == 32, "lse_ray_features_bwd: emb_dim must be 0 or 32 (got %d)", emb_dim);
    LSE_REQUIRE(!d_emb_table || n_emb_rows > 0, "lse_ray_features_bwd: d_emb_table needs n_emb_rows");
    if (d_rays_d)
        hipLaunchKernelGGL(ray_features_bwd_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, lse::as_stream(stream),
                           rays_d, d_feat, emb_idx, n_rays, emb_dim, d_rays_d, d_emb_table);
    if (d_emb_table && emb_idx && emb_dim > 0)
        hipLaunchKernelGGL(emb_grad_kernel, dim3(n_emb_rows, 8), dim3(256), 0, lse::as_stream(stream), d_feat, emb_idx, n_rays,
                           emb_dim, d_emb_table);
    return lse::check_launch("lse_ray_features_bwd");
}

extern "C" int lse_ray_bias_fwd(const float *rays_d, const float *emb_table, const int32_t *emb_idx, int32_t n_rays,
                                int32_t emb_dim, const float *w_in, int32_t w_ld, int32_t width, float *feat,
                                float *row_bias, lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_bias_fwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_d && w_in && feat && row_bias, "lse_ray_bias_fwd: null pointer");
    LSE_REQUIRE(emb_dim >= 0 && emb_dim <= 97, "lse_ray_bias_fwd: emb_dim %d not in [0, 97] (padded input width <= 128)", emb_dim);
    LSE_REQUIRE(width == 32 || width == 64, "lse_ray_bias_fwd: width %d not in {32,64}", width);
    const int in_pad = (31 + emb_dim + 15) / 16 * 16;
    LSE_REQUIRE(w_ld >= in_pad, "lse_ray_bias_fwd: w_ld %d < padded input width %d", w_ld, in_pad);
    const dim3 grid((n_rays + kRbRays - 1) / kRbRays);
#define LSE_RB_FWD(W, P)                                                                                                       \
    hipLaunchKernelGGL((ray_bias_fwd_kernel<W, P>), grid, dim3(256), 0, lse::as_stream(stream), rays_d, emb_table, emb_idx, n_rays, \
                       emb_dim, in_pad, w_in, w_ld, feat, row_bias)
    if (width == 64 && in_pad <= 64) LSE_RB_FWD(64, 65);
    else if (width == 64) LSE_RB_FWD(64, 129);
    else if (in_pad <= 64) LSE_RB_FWD(32, 65);
    else LSE_RB_FWD(32, 129);
#undef LSE_RB_FWD
    return lse::check_launch("lse_ray_bias_fwd");
}

extern "C" int lse_ray_bias_bwd(const float *rays_d, const int32_t *emb_idx, int32_t n_rays, int32_t emb_dim,
                                int32_t n_emb_rows, const float *w_in, int32_t w_ld, int32_t width,
                                const float *d_row_bias, float *d_feat, float *d_rays_d, float *d_emb_table,
                                lse_stream_t stream)
{
    LSE_REQUIRE(n_rays >= 0, "lse_ray_bias_bwd: n_rays < 0");
    if (n_rays == 0) return LSE_OK;
    LSE_REQUIRE(rays_d && w_in && d_row_bias && d_feat, "lse_ray_bias_bwd: null pointer");
    LSE_REQUIRE(emb_dim >= 0 && emb_dim <= 97, "lse_ray_bias_bwd: emb_dim %d not in [0, 97] (padded input width <= 128)", emb_dim);
    LSE_REQUIRE(width == 32 || width == 64, "lse_ray_bias_bwd: width %d not in {32,64}", width);
    LSE_REQUIRE(!d_emb_table || (n_emb_rows > 0 && emb_idx), "lse_ray_bias_bwd: d_emb_table needs n_emb_rows and emb_idx");
    const int in_pad = (31 + emb_dim + 15) / 16 * 16;
    LSE_REQUIRE(w_ld >= in_pad, "lse_ray_bias_bwd: w_ld %d < padded input width %d", w_ld, in_pad);
    hipStream_t st = lse::as_stream(stream);
    const dim3 grid((n_rays + kRbRays - 1) / kRbRays);
#define LSE_RB_BWD(W, P)                                                                                                       \
    hipLaunchKernelGGL((ray_bias_bwd_kernel<W, P>), grid, dim3(256), 0, st, rays_d, d_row_bias, n_rays, emb_dim, in_pad, w_in, w_ld, \
                       d_feat, d_rays_d)
    if (width == 64 && in_pad <= 64) LSE_RB_BWD(64, 65);
    else if (width == 64) LSE_RB_BWD(64, 129);
    else if (in_pad <= 64) LSE_RB_BWD(32, 65);
    else LSE_RB_BWD(32, 129);
#undef LSE_RB_BWD
    if (d_emb_table && emb_dim > 0)
        hipLaunchKernelGGL(emb_grad_kernel, dim3(n_emb_rows, 8, (emb_dim + 31) / 32), dim3(256), 0, st, d_feat, emb_idx, n_rays, emb_dim,
                           d_emb_table, in_pad);
    return lse::check_launch("lse_ray_bias_bwd");
}
