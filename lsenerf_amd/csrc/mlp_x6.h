// Fused small MLPs on the gfx950 bf16 matrix cores with f32-equivalent arithmetic ("bf16x6").  Included by mlp.hip (inside
// its anonymous namespace, after MlpArgs).
//
// v_mfma_f32_16x16x4_f32 runs at the f32 vector rate (157 TFLOP/s); v_mfma_f32_16x16x32_bf16 moves 16x the multiply-adds
// per cycle.  Every f32 operand x is cut into three bf16 pieces
//     x = hi + mid + lo,   hi = top 8 significant bits, mid = the next 8, lo = the last 8      (exact: 24 = 3 x 8, and bf16
//                                                                                                has f32's exponent range)
// and a product a*b is evaluated as the six piece products of weight 2^0 .. 2^-16
//     a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid
// each of which is EXACT in f32 (8 x 8 significant bits) and is accumulated in f32 inside the MFMA.  The dropped pieces
// (mid*lo, lo*mid, lo*lo) are <= 2^-23 |a b| together: the same size as the rounding of one f32 multiply-add, so the result
// carries f32's error bound -- this is not a reduced-precision path (tests/test_gpu_parity.py holds both MLP generations
// against the fp64 oracle with the same tolerance).  Six 16-cycle MFMAs of k = 32 replace eight 32-cycle MFMAs of k = 4:
// 96 instead of 256 matrix-core cycles per 16 x 16 x 32 block of multiply-adds.
//
// Orientation is the one of mlp.hip: H^T = W X^T, samples on the MFMA column (lane & 15), neurons on the accumulator rows.
//   C/D of 16x16x32:  col = lane & 15 (sample), row = 4 * (lane >> 4) + reg.
//   B of 16x16x32:    lane (col j, q = lane >> 4) holds k-slots 8q .. 8q+7 of its sample, two per 32-bit register.
// With the k order  neuron(s, q, t) = 16 * (2s + (t >> 2)) + 4q + (t & 3)  (s = 32-wide k slab, t = slot inside the lane) a
// lane's 8 slots of slab s are registers 0..3 of accumulator row blocks 2s and 2s+1 of the previous layer: layers chain in
// registers, the only work between two layers is ReLU + cutting 16 floats into pieces (on the matrix core, see split_pair).
// Weights (A operands) are cut once per workgroup into LDS images in the same k order, one ds_read_b128 per lane and piece.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LSE_MFMA_BF(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LSE_MFMA_BF16K(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (a)), __builtin_bit_cast(s16x4, (b)), (c), 0, 0, 0)

// one float -> piece p (0 hi, 1 mid, 2 lo) as a bf16 bit pattern, by truncation: the remainders are exact in f32 and the third
// piece is exactly representable (weight images, cut once per workgroup)
__device__ __forceinline__ uint32_t piece_of(float x, int p)
{
    uint32_t b = __float_as_uint(x);
    float r = x;
    for (int i = 0; i < p; ++i) {
        r = r - __uint_as_float(b & 0xffff0000u);
        b = __float_as_uint(r);
    }
    return b >> 16;
}

// ---- cutting on the matrix core.  Cutting on the vector ALU (and / subtract / permute) costs 5.5 instructions per value and
// made the kernels bound by exactly those instructions.  Instead:  hi = bf16(x) (v_cvt_pk_bf16_f32, two values per instruction, round to nearest), then
// the REMAINDER x - hi is produced by one MFMA with the negated identity as the A operand and the packed hi pieces as B
// (D = C - I * hi: the product is exact, and x - bf16(x) is exactly representable, so the MFMA returns it exactly); the same
// again for mid; lo = bf16(second remainder).  8 values per lane (slots 0..3 = c0, 4..7 = c1 -- e.g. accumulator row blocks
// 2s and 2s+1 of one column tile, which is one B operand of the next layer) cost 12 conversions + 4 MFMAs of 16 cycles.
// With round-to-nearest pieces (8 bits + sign each) hi + mid + lo == x exactly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2v));
}

// A operands "minus identity": variant b maps k-slot 4b + r of lane group q to output row 4q + r
__device__ __forceinline__ void neg_identity(int lane, u32x4 (&nI)[2])
{
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int t = 4 * b + (i & 3);
        const uint32_t v = (q == (i >> 2)) ? (0xBF80u << (16 * (t & 1))) : 0u;
        nI[b] = (u32x4){(t >> 1) == 0 ? v : 0u, (t >> 1) == 1 ? v : 0u, (t >> 1) == 2 ? v : 0u, (t >> 1) == 3 ? v : 0u};
    }
}

// c0, c1 -> pieces o[hi, mid, lo]; c0 / c1 are consumed (they end as the third remainders)
template <typename NI>
__device__ __forceinline__ void split_pair(f32x4 c0, f32x4 c1, const NI &nI, u32x4 (&o)[3])
{
    o[0] = (u32x4){cvt_pk(c0[0], c0[1]), cvt_pk(c0[2], c0[3]), cvt_pk(c1[0], c1[1]), cvt_pk(c1[2], c1[3])};
    c0 = LSE_MFMA_BF(nI[0], o[0], c0);
    c1 = LSE_MFMA_BF(nI[1], o[0], c1);
    o[1] = (u32x4){cvt_pk(c0[0], c0[1]), cvt_pk(c0[2], c0[3]), cvt_pk(c1[0], c1[1]), cvt_pk(c1[2], c1[3])};
    c0 = LSE_MFMA_BF(nI[0], o[1], c0);
    c1 = LSE_MFMA_BF(nI[1], o[1], c1);
    o[2] = (u32x4){cvt_pk(c0[0], c0[1]), cvt_pk(c0[2], c0[3]), cvt_pk(c1[0], c1[1]), cvt_pk(c1[2], c1[3])};
}

// four values per lane (k = 16 operands): pieces as 2-register halves
template <typename NI>
__device__ __forceinline__ void split_half(f32x4 c0, const NI &nI, uint32_t (&hi)[2], uint32_t (&mid)[2], uint32_t (&lo)[2])
{
    // the remainder products have k = 16: the two-register v_mfma_f32_16x16x16_bf16 takes the same 16 cycles as the k = 32 form
    // (tools/micro/mfma_rate.hip) and needs no zero upper half (two v_mov per product in the k = 32 form).  Its A operand is the
    // first half of "minus identity" variant 0: lane (row i, group q) holds k = 4q .. 4q+3 -> -1 at k = i.
    const u32x4 n4 = nI[0];
    const u32x2 n2 = (u32x2){n4[0], n4[1]};
    hi[0] = cvt_pk(c0[0], c0[1]);
    hi[1] = cvt_pk(c0[2], c0[3]);
    c0 = LSE_MFMA_BF16K(n2, ((u32x2){hi[0], hi[1]}), c0);
    mid[0] = cvt_pk(c0[0], c0[1]);
    mid[1] = cvt_pk(c0[2], c0[3]);
    c0 = LSE_MFMA_BF16K(n2, ((u32x2){mid[0], mid[1]}), c0);
    lo[0] = cvt_pk(c0[0], c0[1]);
    lo[1] = cvt_pk(c0[2], c0[3]);
}

// k order of a chained accumulator (see above)
__device__ __host__ __forceinline__ constexpr int kslot_chain(int s, int q, int t) { return 16 * (2 * s + (t >> 2)) + 4 * q + (t & 3); }

// The three pieces of the B operands of one column tile whose 16 x HB values sit in accumulator layout: P[s][piece] (4 regs).
template <int HB>
struct PiecesB {
    u32x4 p[HB / 2][3];
};

// acc[rb][ct] += W[rows 16rb.., :] * X for CT column tiles at once (the A pieces are read once); img = LDS image
// [piece][rb][slab][64 lanes] of u32x4
template <int RB, int S, int CT>
__device__ __forceinline__ void layer_x6_ct(f32x4 (&acc)[RB][CT], const u32x4 *img, const PiecesB<2 * S> (&x)[CT], int lane)
{
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const u32x4 ah = img[((0 * RB + rb) * S + s) * 64 + lane];
            const u32x4 am = img[((1 * RB + rb) * S + s) * 64 + lane];
            const u32x4 al = img[((2 * RB + rb) * S + s) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(al, x[ct].p[s][0], acc[rb][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(ah, x[ct].p[s][2], acc[rb][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(am, x[ct].p[s][1], acc[rb][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(am, x[ct].p[s][0], acc[rb][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(ah, x[ct].p[s][1], acc[rb][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(ah, x[ct].p[s][0], acc[rb][ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- k = 16 layers (the head's first layer): both halves of the 32 k-slots of one MFMA carry the SAME 16 inputs, cut into
// different pieces, so the six piece products take three MFMAs:
//     combo 0:  A = (hi | hi)   B = (hi | mid)      W_hi x_hi + W_hi x_mid
//     combo 1:  A = (mid | lo)  B = (hi | hi)       W_mid x_hi + W_lo x_hi
//     combo 2:  A = (hi | mid)  B = (lo | mid)      W_hi x_lo + W_mid x_mid
// slot t < 4 of lane q is input column 4q + t in the first piece, slot t >= 4 the same column in the second piece.
struct PiecesB16 {
    u32x4 c[3];
};

template <typename NI>
__device__ __forceinline__ void split_in16(const f32x4 v, const NI &nI, PiecesB16 &o)
{
    uint32_t h[2], m[2], l[2];
    split_half(v, nI, h, m, l);
    o.c[0] = (u32x4){h[0], h[1], m[0], m[1]};
    o.c[1] = (u32x4){h[0], h[1], h[0], h[1]};
    o.c[2] = (u32x4){l[0], l[1], m[0], m[1]};
}

__device__ __forceinline__ constexpr int combo_piece(int combo, int half)
{
    return combo == 0 ? 0 : combo == 1 ? (half ? 2 : 1) : (half ? 1 : 0);
}

template <int RB, int CT>
__device__ __forceinline__ void layer16_x6_ct(f32x4 (&acc)[RB][CT], const u32x4 *img, const PiecesB16 (&x)[CT], int lane)
{
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const u32x4 a0 = img[(0 * RB + rb) * 64 + lane];
        const u32x4 a1 = img[(1 * RB + rb) * 64 + lane];
        const u32x4 a2 = img[(2 * RB + rb) * 64 + lane];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(a2, x[ct].c[2], acc[rb][ct]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(a1, x[ct].c[1], acc[rb][ct]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = LSE_MFMA_BF(a0, x[ct].c[0], acc[rb][ct]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- weight images -------------------------------------------------------------------------------------------------
// chained layer: W [rows x 64] row-major with leading dimension ld; image [piece][rb][slab][lane] (u32x4 = 8 bf16 slots)
template <int RB, int S>
__device__ __forceinline__ void stage_chain_image(u32x4 *img, const float *W, int ld, int rows_real, int nthreads)
{
    uint32_t *w32 = reinterpret_cast<uint32_t *>(img);
    for (int e = threadIdx.x; e < 3 * RB * S * 64 * 4; e += nthreads) {
        const int word = e & 3, ln = (e >> 2) & 63, rest = e >> 8;
        const int s = rest % S, rb = (rest / S) % RB, p = rest / (S * RB);
        const int i = ln & 15, q = ln >> 4, row = 16 * rb + i;
        const float v0 = row < rows_real ? W[row * ld + kslot_chain(s, q, 2 * word)] : 0.f;
        const float v1 = row < rows_real ? W[row * ld + kslot_chain(s, q, 2 * word + 1)] : 0.f;
        w32[e] = piece_of(v0, p) | (piece_of(v1, p) << 16);
    }
}

// level-major / row-major 32-input first layer: slot t of lane q is input column 8q + t; image [piece][rb][lane]
template <int RB>
__device__ __forceinline__ void stage_in32_image(u32x4 *img, const float *W, int ld, int mask0, int nthreads)
{
    uint32_t *w32 = reinterpret_cast<uint32_t *>(img);
    for (int e = threadIdx.x; e < 3 * RB * 64 * 4; e += nthreads) {
        const int word = e & 3, ln = (e >> 2) & 63, rest = e >> 8;
        const int rb = rest % RB, p = rest / RB;
        const int i = ln & 15, q = ln >> 4, row = 16 * rb + i;
        const int c0 = 8 * q + 2 * word, c1 = c0 + 1;
        const float v0 = (mask0 && c0 == 0) ? 0.f : W[row * ld + c0];
        const float v1 = W[row * ld + c1];
        w32[e] = piece_of(v0, p) | (piece_of(v1, p) << 16);
    }
}

// 16-input first layer, image [combo][rb][lane]
template <int RB>
__device__ __forceinline__ void stage_in16_image(u32x4 *img, const float *W, int ld, int mask0, int nthreads)
{
    uint32_t *w32 = reinterpret_cast<uint32_t *>(img);
    for (int e = threadIdx.x; e < 3 * RB * 64 * 4; e += nthreads) {
        const int word = e & 3, ln = (e >> 2) & 63, rest = e >> 8;
        const int rb = rest % RB, combo = rest / RB;
        const int i = ln & 15, q = ln >> 4, row = 16 * rb + i;
        const int half = word >> 1, c0 = 4 * q + 2 * (word & 1), c1 = c0 + 1;
        const int p = combo_piece(combo, half);
        const float v0 = (mask0 && c0 == 0) ? 0.f : W[row * ld + c0];
        const float v1 = W[row * ld + c1];
        w32[e] = piece_of(v0, p) | (piece_of(v1, p) << 16);
    }
}

// ------------------------------------------------------------------------------------------------------
// forward (third generation).  Same tile walk, prefetch, activation layout and stores as mlp_fwd2_kernel; WIDTH = 64.
//   KIN = 16 row-major (head) or 32 level-major / row-major (base).
// ------------------------------------------------------------------------------------------------------
template <int KIN, int NHL, int INL>
struct X6Fwd {
    static constexpr int WIDTH = 64, HB = 4, S = 2;
    static constexpr int IMG0 = 3 * HB * 64;                       // u32x4 entries
    static constexpr int IMGH = (NHL == 2) ? 3 * HB * S * 64 : 0;
    static constexpr int IMGO = 3 * 1 * S * 64;
    static constexpr int lds_bytes = (IMG0 + IMGH + IMGO + 128) * 16;      // + the two "minus identity" operands
};

template <int KIN, int INL, int CT>
__device__ __forceinline__ void load_in_x6(const MlpArgs &a, int64_t tile, int j, int q, f32x4 (&raw)[CT][KIN / 16])
{
    constexpr int TS = 16 * CT;
    const int64_t n = a.n, tile_base = tile * TS;
    const int n_rem = (int)min((int64_t)TS, n - tile_base);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int l = ct * 16 + j;
        const int slc = l < n_rem ? l : n_rem - 1;
        if constexpr (KIN == 16) {
            raw[ct][0] = *reinterpret_cast<const f32x4 *>(a.in + tile_base * 16 + (unsigned)(slc * 16 + 4 * q));
        } else if constexpr (INL == LSE_IN_LEVELMAJOR) {
            const float2 *in2 = reinterpret_cast<const float2 *>(a.in) + tile_base;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float2 u = in2[(int64_t)(4 * q + 2 * m) * a.n_stride + slc], v = in2[(int64_t)(4 * q + 2 * m + 1) * a.n_stride + slc];
                raw[ct][m] = (f32x4){u.x, u.y, v.x, v.y};
            }
        } else {
            const float *p = a.in + tile_base * 32 + (unsigned)(slc * 32 + 8 * q);
            raw[ct][0] = *reinterpret_cast<const f32x4 *>(p);
            raw[ct][1] = *reinterpret_cast<const f32x4 *>(p + 4);
        }
    }
}

// first hidden layer of one tile: h = relu(W0 * in + bias), shared by the forward and the recomputing backward so that both
// produce the same bits
template <int KIN, int CT, typename NI>
__device__ __forceinline__ void first_layer_x6(f32x4 (&h)[4][CT], const u32x4 *img0, const f32x4 (&raw)[CT][KIN / 16],
                                               const NI &nI, int lane)
{
    if constexpr (KIN == 16) {
        PiecesB16 x[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) split_in16(raw[ct][0], nI, x[ct]);
        layer16_x6_ct<4, CT>(h, img0, x, lane);
    } else {
        PiecesB<2> x[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) split_pair(raw[ct][0], raw[ct][1], nI, x[ct].p[0]);
        layer_x6_ct<4, 1, CT>(h, img0, x, lane);
    }
}

template <int KIN, int NHL, int INL>
__global__ __launch_bounds__(512) void mlp_fwd3_kernel(MlpArgs a, bool nt)
{
    using C = X6Fwd<KIN, NHL, INL>;
    constexpr int CT = 2, NW = 8, TS = 16 * CT, HB = 4, WIDTH = 64;
    static_assert(KIN == 16 || KIN == 32, "first layer: 16 or 32 inputs");
    extern __shared__ float lds[];
    u32x4 *img0 = reinterpret_cast<u32x4 *>(lds), *imgH = img0 + C::IMG0, *imgO = imgH + C::IMGH;

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    if constexpr (KIN == 16) stage_in16_image<HB>(img0, W0, a.w0_ld, a.w0_mask0, 64 * NW);
    else stage_in32_image<HB>(img0, W0, a.w0_ld, a.w0_mask0, 64 * NW);
    if constexpr (NHL == 2) stage_chain_image<HB, 2>(imgH, W1, WIDTH, WIDTH, 64 * NW);
    stage_chain_image<1, 2>(imgO, Wo, WIDTH, 16, 64 * NW);
    if (threadIdx.x < 64) {
        u32x4 t[2];
        neg_identity(threadIdx.x, t);
        imgO[C::IMGO + threadIdx.x] = t[0];
        imgO[C::IMGO + 64 + threadIdx.x] = t[1];
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    a.n = lse::clamp_count(a.n, a.n_dev);
    const int64_t n = a.n;
    const int64_t n_tiles = (n + TS - 1) / TS;
    const int64_t total_waves = (int64_t)gridDim.x * NW;
    const int64_t per = (n_tiles + total_waves - 1) / total_waves;
    const int64_t w_id = (int64_t)blockIdx.x * NW + wave;
    const int64_t t_begin = min(n_tiles, w_id * per), t_end = min(n_tiles, t_begin + per);
    const int oc = a.out_cols;
    struct NegI {          // "minus identity" operands from LDS (8 registers less: one more wave per SIMD for the base shape)
        const u32x4 *p;
        __device__ __forceinline__ u32x4 operator[](int b) const { return p[64 * b]; }
    };
    const NegI nI{imgO + C::IMGO + lane};

    f32x4 raw_nx[CT][KIN / 16];
    if (t_begin < t_end) load_in_x6<KIN, INL, CT>(a, t_begin, j, q, raw_nx);
    int idx_nx[CT] = {};
    if (a.row_bias && a.row_bias_idx && t_begin < t_end) {
        const int64_t b0 = t_begin * TS;
        const int nr = (int)min((int64_t)TS, n - b0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) idx_nx[ct] = a.row_bias_idx[b0 + min(ct * 16 + j, nr - 1)];
    }
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);
        int sl[CT];
        bool valid[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            valid[ct] = ct * 16 + j < n_rem;
            sl[ct] = valid[ct] ? ct * 16 + j : n_rem - 1;
        }
        f32x4 raw[CT][KIN / 16];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int b = 0; b < KIN / 16; ++b) raw[ct][b] = raw_nx[ct][b];
        // (density head) the selector is requested here, not behind the last product of the tile
        uint32_t selv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) selv[ct] = (a.sigma_out && a.selector && q == 0) ? (uint32_t)a.selector[tile_base + sl[ct]] : 1u;
        f32x4 h[HB][CT];
        if (a.row_bias) {
            int64_t rowv[CT];
            if (a.row_bias_idx) {      // this tile's rows were requested during the previous tile; the next tile's are requested now
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) rowv[ct] = idx_nx[ct];
                if (tile + 1 < t_end) {
                    const int64_t nb_ = tile_base + TS;
                    const int nr = (int)min((int64_t)TS, n - nb_);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) idx_nx[ct] = a.row_bias_idx[nb_ + min(ct * 16 + j, nr - 1)];
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) rowv[ct] = tile_base + sl[ct];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int64_t row = rowv[ct];
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
        if (tile + 1 < t_end) load_in_x6<KIN, INL, CT>(a, tile + 1, j, q, raw_nx);
        // ---- layer 0
        first_layer_x6<KIN, CT>(h, img0, raw, nI, lane);
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
        PiecesB<HB> x[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) split_pair(h[2 * s2][ct], h[2 * s2 + 1][ct], nI, x[ct].p[s2]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hidden layer
        if constexpr (NHL == 2) {
            f32x4 h2[HB][CT];
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) h2[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            layer_x6_ct<HB, 2, CT>(h2, imgH, x, lane);
            float *act1_t = act0_t ? act0_t + (skip0 ? 0 : a.act_layer_stride) : nullptr;
#pragma unroll
            for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[rb][ct][r] = relu_bits(h2[rb][ct][r]);
                    if (act1_t && (ct == 0 || st1)) store_act(act1_t + (unsigned)(((ct * HB + rb) * 64 + lane) * 4), h2[rb][ct], nt);
                }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) split_pair(h2[2 * s2][ct], h2[2 * s2 + 1][ct], nI, x[ct].p[s2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- output layer
        f32x4 o[1][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) o[0][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        layer_x6_ct<1, 2, CT>(o, imgO, x, lane);
        float *out_t = a.out + tile_base * oc;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (a.out_activation == LSE_ACT_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[0][ct][r] = __builtin_amdgcn_rcpf(1.f + __expf(-o[0][ct][r]));
            }
            if (valid[ct]) {
                if (oc == 16) *reinterpret_cast<f32x4 *>(out_t + (unsigned)(sl[ct] * 16 + 4 * q)) = o[0][ct];
                else if (q == 0) *reinterpret_cast<f32x4 *>(out_t + (unsigned)(sl[ct] * 4)) = o[0][ct];
                if (a.sigma_out && q == 0) a.sigma_out[tile_base + sl[ct]] = selv[ct] != 0u ? a.density_scale * expf(o[0][ct][0]) : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward (third generation): data + weight gradients in one pass with NOTHING saved by the forward (act_tiled = 3):
// the hidden layers are recomputed (the matrix-core time of a recomputed layer is a quarter of the time its activation
// took to store and load).  WIDTH = 64; head <16, 2, row-major> and base <32, 1, level-major / row-major>.
//
// Weight gradients sum over SAMPLES, so both operands of dW += G^T H need the sample index on the k slots.  Everything is
// cut into pieces ONCE, in accumulator layout (lane = sample), and written as packed bf16 to a per-wave LDS tile
// [32 samples][64 neurons] per piece; ds_read_b64_tr_b16 reads that tile back transposed (lane = neuron, 4 samples per
// read) -- the hardware transpose replaces the f32 LDS transposition + second split of the f32 kernels.  The same
// instruction serves the weights: every matrix is staged once as plain [out][in] piece images; row reads feed the forward
// products, transposed reads the data gradients (A = W^T).
//
// LDS images use 32-byte segments XOR-swizzled with the row so that 8-byte row accesses and transposed block reads are both
// conflict-free (bank rule: (addr/4) mod 64 for ds_read_b64 / _tr_b16, mod 32 for writes).
// ------------------------------------------------------------------------------------------------------
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

// byte offset of 8-byte chunk `chunk` of 32-byte segment `seg` of row `row` in an image with S segments (16 bf16) per row
template <int S>
__device__ __forceinline__ int img_off(int row, int seg, int chunk)
{
    // segment swizzle: the 8 rows of a transposed half-wave read land on 8 x 32 B = all 64 banks;  chunk swizzle: 16 lanes on 16
    // consecutive rows (8-byte row accesses, which the compiler may merge into ds_read2 / ds_write_b64 with 32 banks and
    // 16-lane groups) land on 16 different bank pairs
    const int sw = S == 4 ? ((row >> 1) & 3) : S == 2 ? ((row >> 2) & 1) : 0;
    const int cw = S == 4 ? ((row & 1) | (((row >> 3) & 1) << 1)) : S == 1 ? ((row >> 2) & 3) : 0;
    return row * (32 * S) + ((seg ^ sw) << 5) + ((chunk ^ cw) << 3);
}

__device__ __forceinline__ u32x2 lds_tr_read(const uint8_t *p)
{
    const v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    return __builtin_bit_cast(u32x2, v);
}
__device__ __forceinline__ u32x2 lds_read8(const uint8_t *p) { return *reinterpret_cast<const u32x2 *>(p); }
__device__ __forceinline__ void lds_write8(uint8_t *p, uint32_t w0, uint32_t w1) { *reinterpret_cast<u32x2 *>(p) = (u32x2){w0, w1}; }

// plain piece images of W [ROWS x COLS] (leading dimension ld): piece p at img + p * ROWS * COLS * 2
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_plain(uint8_t *img, const float *W, int ld, int mask0, int nthreads)
{
    constexpr int S = COLS / 16;
    for (int e = threadIdx.x; e < 3 * ROWS * COLS / 2; e += nthreads) {
        const int cp = e % (COLS / 2), row = (e / (COLS / 2)) % ROWS, p = e / (ROWS * COLS / 2);
        const int c0 = 2 * cp;
        const float v0 = (mask0 && c0 == 0) ? 0.f : W[row * ld + c0];
        const float v1 = W[row * ld + c0 + 1];
        *reinterpret_cast<uint32_t *>(img + p * (ROWS * COLS * 2) + img_off<S>(row, c0 >> 4, (c0 & 15) >> 2) + 4 * (cp & 1)) =
            piece_of(v0, p) | (piece_of(v1, p) << 16);
    }
}

// gradient through a ReLU whose "was positive" bits are packed in m: bit `pos` set -> g, clear -> +0.  Two instructions
// (v_bfe_i32 spreads the bit over the word, v_and) instead of the three of a test + select.
__device__ __forceinline__ float relu_gate(float g, uint32_t m, int pos)
{
    return __uint_as_float(__float_as_uint(g) & (uint32_t)__builtin_amdgcn_sbfe((int)m, (unsigned)pos, 1u));
}

// six piece products into one accumulator: a[], b[] = {hi, mid, lo}
__device__ __forceinline__ f32x4 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c)
{
    c = LSE_MFMA_BF(a[2], b[0], c);
    c = LSE_MFMA_BF(a[0], b[2], c);
    c = LSE_MFMA_BF(a[1], b[1], c);
    c = LSE_MFMA_BF(a[1], b[0], c);
    c = LSE_MFMA_BF(a[0], b[1], c);
    c = LSE_MFMA_BF(a[0], b[0], c);
    return c;
}

// k = 16 products: w[] = {hi, mid, lo} halves (2 registers each), x = combos of the other operand (split_in16)
struct Halves {
    u32x2 p[3];
};
// the three A-side operands (hi|hi), (mid|lo), (hi|mid) from 8-byte LDS reads: rd(p) reads the halves of piece p
template <typename F>
__device__ __forceinline__ void combos16_lds(F rd, u32x4 (&c)[3])
{
    const u32x2 h0 = rd(0), h1 = rd(0), m0 = rd(1), m1 = rd(1), l0 = rd(2);
    c[0] = (u32x4){h0[0], h0[1], h1[0], h1[1]};
    c[1] = (u32x4){m0[0], m0[1], l0[0], l0[1]};
    c[2] = (u32x4){h1[0], h1[1], m1[0], m1[1]};
}

template <int KIN, int NHL, int CT, int NW>
struct X6Bwd {
    static constexpr int S0 = KIN / 16;
    static constexpr int W0_PIECE = 64 * 32 * S0, W1_PIECE = (NHL == 2) ? 64 * 128 : 0, WO_PIECE = 16 * 128;
    static constexpr int W_BYTES = 3 * (W0_PIECE + W1_PIECE + WO_PIECE);
    static constexpr int BUF_PIECE = 16 * CT * 128, SM_PIECE = 16 * CT * 32;
    static constexpr int WAVE_BYTES = 3 * (BUF_PIECE + SM_PIECE);
    static constexpr int NI_BYTES = 2 * 1024;                 // the two "minus identity" A operands, one 16-byte entry per lane
    static constexpr int lds_bytes = W_BYTES + NI_BYTES + NW * WAVE_BYTES;
};

// CT = column tiles (16 samples) per wave tile, NW = waves per workgroup (one workgroup per CU).  CT = 2 / NW = 8: the weight
// operands are read once per 32 samples, two waves per SIMD.  CT = 1: half the per-tile state (registers, LDS tile) so that more
// waves fit a SIMD; the weight gradients then sum over 16 samples per product and use the k = 16 form (three MFMAs on two-piece
// windows) -- the same matrix-core cycles per sample.
template <int KIN, int NHL, int INL, bool BIAS, bool BIAS_ONES, int CT, int NW>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void mlp_bwd3_kernel(MlpArgs a)
{
    using C = X6Bwd<KIN, NHL, CT, NW>;
#define LSE_X6_SB()                                        \
    do {                                                   \
        if constexpr (NW > 4) __builtin_amdgcn_sched_barrier(0); \
    } while (0)
// (tools/asm_mix.py: -DLSE_PHASE_MARKS leaves "; LSE_PHASE name" comments in the listing so that the instruction mix can be read per phase)
#ifdef LSE_PHASE_MARKS
#define LSE_PHASE(name) asm volatile("; LSE_PHASE " name)
#else
#define LSE_PHASE(name) do {} while (0)
#endif
    constexpr int TS = 16 * CT, HB = 4, WIDTH = 64, S0 = C::S0, KB0 = KIN / 16;
    constexpr bool MS = (KIN == 32);      // remainder MFMAs on pairs of row blocks (base) / on single blocks (head: fewer live registers)
    static_assert((KIN == 16 && NHL == 2 && INL == LSE_IN_ROWMAJOR) || (KIN == 32 && NHL == 1), "head or base shape");
    extern __shared__ float lds[];
    uint8_t *imgW0 = reinterpret_cast<uint8_t *>(lds);
    uint8_t *imgW1 = imgW0 + 3 * C::W0_PIECE;
    uint8_t *imgWo = imgW1 + 3 * C::W1_PIECE;

    const float *W0 = a.params + a.w0_col;
    const float *W1 = a.params + a.rest_off;
    const float *Wo = W1 + (NHL - 1) * WIDTH * WIDTH;
    stage_plain<64, KIN>(imgW0, W0, a.w0_ld, a.w0_mask0, 64 * NW);
    if constexpr (NHL == 2) stage_plain<64, 64>(imgW1, W1, WIDTH, 0, 64 * NW);
    stage_plain<16, 64>(imgWo, Wo, WIDTH, 0, 64 * NW);
    if (threadIdx.x < 64) {
        u32x4 t[2];
        neg_identity(threadIdx.x, t);
        u32x4 *dst = reinterpret_cast<u32x4 *>(imgWo + 3 * C::WO_PIECE);
        dst[threadIdx.x] = t[0];
        dst[64 + threadIdx.x] = t[1];
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int tq = j >> 2, tp = j & 3;               // transposed reads: this lane addresses row tq, chunk tp of its group's block
    // "minus identity" operands live in LDS (registers are the scarce resource of this kernel): read where a split needs them
    const u32x4 *nI_lds = reinterpret_cast<const u32x4 *>(imgWo + 3 * C::WO_PIECE) + lane;
    struct NegI {
        const u32x4 *p;
        __device__ __forceinline__ u32x4 operator[](int b) const { return p[64 * b]; }
    };
    const NegI nI{nI_lds};
    uint8_t *buf = imgWo + 3 * C::WO_PIECE + C::NI_BYTES + wave * C::WAVE_BYTES;      // [piece][32 samples][64 neurons]
    uint8_t *sm = buf + 3 * C::BUF_PIECE;                               // [piece][32 samples][16 columns]
    // Per-lane byte offsets, computed once: every LDS address below is one of these + a compile-time constant (the DS
    // instructions' immediate offset), so the address arithmetic costs no registers inside the tile loop.
    //   "row" accesses   : image row 16X + j, 8-byte chunk q of segment seg          (128-byte rows: so_j[seg] + 2048 X)
    //   "transposed" ones: image row 8X + 4q + tq, 8-byte chunk tp of segment seg    (128-byte rows: so_t[seg] + 1024 X)
    int so_j[4], so_t[4];
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
        so_j[seg] = img_off<4>(j, seg, q);
        so_t[seg] = img_off<4>(4 * q + tq, seg, tp);
    }
    const int sm_j = img_off<1>(j, 0, q), sm_t = img_off<1>(4 * q + tq, 0, tp);   // 32-byte rows (small tile, 16-column W0)
    const int w0b_j = img_off<2>(j, q >> 1, 2 * (q & 1));                        // 64-byte rows (32-column W0): 16-byte row read
    int w0b_t[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) w0b_t[cb] = img_off<2>(4 * q + tq, cb, tp);

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
    const bool need_out = a.out_activation == LSE_ACT_SIGMOID;
    const int oc = a.out_cols;
    int cur_row = -1;            // BIAS_ONES: the row whose running sum sits in column 0 of acc0[.][0]

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
    auto lds_sync = [&]() {      // the tiles are private to the wave; DS operations of a wave execute in order
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // Operands of the weight-gradient products (k = samples), read transposed from the tile (lane = neuron).
    //   CT = 2: the three pieces, slots = samples 4q+t | 16+4q+t  -> six piece products (mfma6)
    //   CT = 1: k = 16 windows of two pieces (slots 0..3 | 4..7 carry the same samples 4q+t):
    //           A side (hi|hi), (mid|lo), (hi|mid);  B side (hi|mid), (hi|hi), (lo|mid)  -> three products
    auto windows = [&](auto rd, bool a_side, u32x4 (&o)[3]) {     // rd(piece) -> the 4 slots of that piece (2 registers)
        if (a_side) {
            const u32x2 h = rd(0), m = rd(1), l = rd(2);
            o[0] = (u32x4){h[0], h[1], h[0], h[1]};
            o[1] = (u32x4){m[0], m[1], l[0], l[1]};
            o[2] = (u32x4){h[0], h[1], m[0], m[1]};
        } else {
            const u32x2 h = rd(0), m = rd(1), l = rd(2);
            o[0] = (u32x4){h[0], h[1], m[0], m[1]};
            o[1] = (u32x4){h[0], h[1], h[0], h[1]};
            o[2] = (u32x4){l[0], l[1], m[0], m[1]};
        }
    };
    auto tr_block = [&](int nb, bool a_side, u32x4 (&o)[3]) {
        if constexpr (CT == 2) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const u32x2 lo = lds_tr_read(buf + (p * C::BUF_PIECE + so_t[nb]));
                const u32x2 hi = lds_tr_read(buf + (p * C::BUF_PIECE + 2048 + so_t[nb]));
                o[p] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        } else {
            windows([&](int p) { return lds_tr_read(buf + (p * C::BUF_PIECE + so_t[nb])); }, a_side, o);
        }
    };
    auto tr_small = [&](bool a_side, u32x4 (&o)[3]) {
        if constexpr (CT == 2) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const u32x2 lo = lds_tr_read(sm + (p * C::SM_PIECE + sm_t));
                const u32x2 hi = lds_tr_read(sm + (p * C::SM_PIECE + 512 + sm_t));
                o[p] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        } else {
            windows([&](int p) { return lds_tr_read(sm + (p * C::SM_PIECE + sm_t)); }, a_side, o);
        }
    };
    auto wg_mfma = [&](const u32x4 (&a3)[3], const u32x4 (&b3)[3], f32x4 c) {
        if constexpr (CT == 2) {
            return mfma6(a3, b3, c);
        } else {
            c = LSE_MFMA_BF(a3[2], b3[2], c);
            c = LSE_MFMA_BF(a3[1], b3[1], c);
            return LSE_MFMA_BF(a3[0], b3[0], c);
        }
    };
    // f32x4 (4 consecutive columns of one sample) -> plain pieces
    auto split4 = [&](const f32x4 v, Halves &h) {
        uint32_t hi[2], mid[2], lo[2];
        split_half(v, nI, hi, mid, lo);
        h.p[0] = (u32x2){hi[0], hi[1]};
        h.p[1] = (u32x2){mid[0], mid[1]};
        h.p[2] = (u32x2){lo[0], lo[1]};
    };
    auto write_sm = [&](int ct, const Halves &h) {
#pragma unroll
        for (int p = 0; p < 3; ++p) lds_write8(sm + (p * C::SM_PIECE + 512 * ct + sm_j), h.p[p][0], h.p[p][1]);
    };
    // chained-k row read of W1 (row 16rb + j): slots 0..3 = columns 32s + 4q .., slots 4..7 = columns 32s + 16 + 4q ..
    auto row_chain_w1 = [&](int rb, int s, u32x4 (&o)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const u32x2 lo = lds_read8(imgW1 + (p * C::W1_PIECE + 2048 * rb + so_j[2 * s]));
            const u32x2 hi = lds_read8(imgW1 + (p * C::W1_PIECE + 2048 * rb + so_j[2 * s + 1]));
            o[p] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    // chained-k TRANSPOSED reads (A = W^T): output rows = image columns 16cb + (lane & 15), slots = image rows 32s + ..
    auto tr_chain_w1 = [&](int cb, int s, u32x4 (&o)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const u32x2 lo = lds_tr_read(imgW1 + (p * C::W1_PIECE + 4096 * s + so_t[cb]));
            const u32x2 hi = lds_tr_read(imgW1 + (p * C::W1_PIECE + 4096 * s + 2048 + so_t[cb]));
            o[p] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    auto tr_chain_w0 = [&](int cb, int s, u32x4 (&o)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int base = (KIN == 16) ? sm_t : w0b_t[cb & 1];
            const int row32 = 32 * (32 * S0);        // bytes of 32 image rows
            const u32x2 lo = lds_tr_read(imgW0 + (p * C::W0_PIECE + row32 * s + base));
            const u32x2 hi = lds_tr_read(imgW0 + (p * C::W0_PIECE + row32 * s + row32 / 2 + base));
            o[p] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };

    // pieces of two accumulator row blocks 2s2, 2s2+1 (= one k slab of the next product), both column tiles
    auto split_blocks = [&](int s2, const f32x4 (&c0)[CT], const f32x4 (&c1)[CT], PiecesB<HB> (&x)[CT]) {
        if constexpr (MS) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) split_pair(c0[ct], c1[ct], nI, x[ct].p[s2]);
        }
    };
    // head shape (registers are scarce): one block at a time, as soon as it is complete -- the same two remainder MFMAs per
    // block as the paired form (the unused half of the B operand is zero)
    auto split_block_single = [&](int s2, int b, const f32x4 (&c)[CT], PiecesB<HB> (&x)[CT]) {
        if constexpr (!MS) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint32_t hi[2], mid[2], lo[2];
                split_half(c[ct], nI, hi, mid, lo);
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    x[ct].p[s2][0][2 * b + w2] = hi[w2];
                    x[ct].p[s2][1][2 * b + w2] = mid[w2];
                    x[ct].p[s2][2][2 * b + w2] = lo[w2];
                }
            }
        }
    };
    auto write_buf_rb = [&](int rb, const PiecesB<HB> (&x)[CT]) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                lds_write8(buf + (p * C::BUF_PIECE + 2048 * ct + so_j[rb]), x[ct].p[rb >> 1][p][2 * (rb & 1)], x[ct].p[rb >> 1][p][2 * (rb & 1) + 1]);
    };
    auto read_buf_own = [&](PiecesB<HB> (&x)[CT]) {          // this lane's own pieces back from the tile
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int rb = 0; rb < HB; ++rb) {
                    const u32x2 v = lds_read8(buf + (p * C::BUF_PIECE + 2048 * ct + so_j[rb]));
                    x[ct].p[rb >> 1][p][2 * (rb & 1)] = v[0];
                    x[ct].p[rb >> 1][p][2 * (rb & 1) + 1] = v[1];
                }
    };

    int brow_nx[CT] = {};
    if (BIAS && a.row_bias_idx && t_begin < t_end) {
        const int64_t b0 = t_begin * TS;
        const int nr = (int)min((int64_t)TS, n - b0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) brow_nx[ct] = a.row_bias_idx[b0 + min(ct * 16 + j, nr - 1)];
    }
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        const int64_t tile_base = tile * TS;
        const int n_rem = (int)min((int64_t)TS, n - tile_base);       // valid samples of this tile (wave-uniform, >= 1)
        int sl[CT];
        bool valid[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            valid[ct] = ct * 16 + j < n_rem;
            sl[ct] = valid[ct] ? ct * 16 + j : n_rem - 1;
        }
        // ---- loads
        LSE_PHASE("loads");
        f32x4 raw[CT][KIN / 16];
        load_in_x6<KIN, INL, CT>(a, tile, j, q, raw);
        // per-row bias: row of this lane's sample per column tile (accumulator layout), as a 32-bit offset from a wave-uniform base
        const float *rbase = a.row_bias;
        int brow[CT] = {};
        if constexpr (BIAS) {
            if (a.row_bias_idx) {       // this tile's rows were requested during the previous tile; request the next tile's now
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) brow[ct] = brow_nx[ct];
                if (tile + 1 < t_end) {
                    const int64_t nb_ = tile_base + TS;
                    const int nr = (int)min((int64_t)TS, n - nb_);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) brow_nx[ct] = a.row_bias_idx[nb_ + min(ct * 16 + j, nr - 1)];
                }
            } else {
                rbase = a.row_bias + tile_base * WIDTH;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) brow[ct] = sl[ct];
            }
        }
        f32x4 g[CT];
        {
            // every load of the tile is issued before the first use (no load behind a branch on another load's value)
            const float *dout_t = a.d_out + tile_base * oc;
            const float *out_t = a.out ? a.out + tile_base * oc : nullptr;
            f32x4 ovv[CT];
            float dsgv[CT], o0v[CT];
            uint32_t selv[CT];
            const bool lane_sig = a.d_sigma != nullptr && q == 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int slc = sl[ct];
                ovv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (oc == 16) g[ct] = *reinterpret_cast<const f32x4 *>(dout_t + (unsigned)(slc * 16 + 4 * q));
                else g[ct] = (q == 0) ? *reinterpret_cast<const f32x4 *>(dout_t + (unsigned)(slc * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (need_out) {
                    if (oc == 16) ovv[ct] = *reinterpret_cast<const f32x4 *>(out_t + (unsigned)(slc * 16 + 4 * q));
                    else if (q == 0) ovv[ct] = *reinterpret_cast<const f32x4 *>(out_t + (unsigned)(slc * 4));
                }
                dsgv[ct] = lane_sig ? a.d_sigma[tile_base + slc] : 0.f;
                o0v[ct] = (lane_sig && !need_out) ? out_t[(unsigned)(slc * oc)] : 0.f;
                selv[ct] = (lane_sig && a.selector) ? (uint32_t)a.selector[tile_base + slc] : 1u;
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 ov = ovv[ct];
                if (lane_sig && !need_out) ov[0] = o0v[ct];
                const float dsg = selv[ct] != 0u ? dsgv[ct] : 0.f;
                const float x0 = fminf(fmaxf(ov[0], -15.f), 15.f);
                if (need_out) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[ct][r] = g[ct][r] * ov[r] * (1.f - ov[r]);
                }
                if (lane_sig) g[ct][0] += dsg * a.density_scale * expf(x0);
                if (!valid[ct]) g[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- first hidden layer again, one row block at a time: relu(W0 in + bias) -> mask bits + pieces
        LSE_PHASE("h0");
        PiecesB<HB> x[CT];
        uint32_t m0[CT] = {}, m1[CT] = {};      // ReLU masks, bit 4rb + r
        {
            PiecesB16 xin16[CT];
            u32x4 xin[CT][3];
            if constexpr (KIN == 16) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) split_in16(raw[ct][0], nI, xin16[ct]);
            } else {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) split_pair(raw[ct][0], raw[ct][1], nI, xin[ct]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f32x4 h[2][CT];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rb = 2 * s2 + b;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        if constexpr (BIAS) h[b][ct] = *reinterpret_cast<const f32x4 *>(rbase + (unsigned)(brow[ct] * WIDTH + 16 * rb + 4 * q));
                        else h[b][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    if constexpr (KIN == 16) {
                        u32x4 wc[3];
                        combos16_lds([&](int p) { return lds_read8(imgW0 + (p * C::W0_PIECE + 512 * rb + sm_j)); }, wc);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h[b][ct] = LSE_MFMA_BF(wc[2], xin16[ct].c[2], h[b][ct]);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h[b][ct] = LSE_MFMA_BF(wc[1], xin16[ct].c[1], h[b][ct]);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h[b][ct] = LSE_MFMA_BF(wc[0], xin16[ct].c[0], h[b][ct]);
                    } else {
                        u32x4 wa[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p)      // slots = columns 8q .. 8q+7: 16 contiguous bytes inside one segment
                            wa[p] = *reinterpret_cast<const u32x4 *>(imgW0 + (p * C::W0_PIECE + 1024 * rb + w0b_j));
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h[b][ct] = mfma6(wa, xin[ct], h[b][ct]);
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            h[b][ct][r] = relu_bits(h[b][ct][r]);
                            m0[ct] |= (h[b][ct][r] > 0.f ? 1u : 0u) << (4 * rb + r);
                        }
                    split_block_single(s2, b, h[b], x);
                    if constexpr (!MS) LSE_X6_SB();
                }
                split_blocks(s2, h[0], h[1], x);
                LSE_X6_SB();
            }
        }
        if constexpr (NHL == 2) {
            // ---- second hidden layer again; its pieces go straight to the tile (they are only needed transposed, for dWo)
            LSE_PHASE("h1");
            PiecesB<HB> x1[CT];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f32x4 h[2][CT];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rb = 2 * s2 + b;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) h[b][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        u32x4 wa[3];
                        row_chain_w1(rb, s, wa);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) h[b][ct] = mfma6(wa, x[ct].p[s], h[b][ct]);
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            h[b][ct][r] = relu_bits(h[b][ct][r]);
                            m1[ct] |= (h[b][ct][r] > 0.f ? 1u : 0u) << (4 * rb + r);
                        }
                    split_block_single(s2, b, h[b], x1);
                    if constexpr (!MS) {
                        write_buf_rb(rb, x1);
                        LSE_X6_SB();
                    }
                }
                split_blocks(s2, h[0], h[1], x1);
                if constexpr (MS) {
                    write_buf_rb(2 * s2, x1);
                    write_buf_rb(2 * s2 + 1, x1);
                }
                LSE_X6_SB();
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < HB; ++rb) write_buf_rb(rb, x);
        }
        // ---- output gradient: pieces to the small tile (-> A operand of dWo), combos for dH_last;  dWo += G_out^T * H_last
        LSE_PHASE("dWo");
        PiecesB16 gx[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            Halves h;
            split4(g[ct], h);
            write_sm(ct, h);
            gx[ct].c[0] = (u32x4){h.p[0][0], h.p[0][1], h.p[1][0], h.p[1][1]};
            gx[ct].c[1] = (u32x4){h.p[0][0], h.p[0][1], h.p[0][0], h.p[0][1]};
            gx[ct].c[2] = (u32x4){h.p[2][0], h.p[2][1], h.p[1][0], h.p[1][1]};
        }
        lds_sync();
        {
            u32x4 ga[3];
            tr_small(true, ga);
            u32x4 hb[2][3];
            tr_block(0, false, hb[0]);
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
                if (kb + 1 < HB) tr_block(kb + 1, false, hb[(kb + 1) & 1]);
                accO[0][kb] = wg_mfma(ga, hb[kb & 1], accO[0][kb]);
                LSE_X6_SB();
            }
        }
        lds_sync();
        // (32-input shape) the layer-0 input in operand layout for dW0 -- lane = input column, slots = samples -- is read from
        // memory; requested here so that the loads land under the products below instead of in front of dW0
        float vin[(KIN == 32) ? KB0 : 1][4 * CT];
        if constexpr (KIN == 32) {
#pragma unroll
            for (int nb = 0; nb < KB0; ++nb) {
                const int col = 16 * nb + j;
#pragma unroll
                for (int t = 0; t < 4 * CT; ++t) {
                    int s_t = 16 * (t >> 2) + 4 * q + (t & 3);
                    s_t = s_t < n_rem ? s_t : n_rem - 1;
                    if (INL == LSE_IN_LEVELMAJOR) vin[nb][t] = a.in[((int64_t)(col >> 1) * ns + tile_base + s_t) * 2 + (col & 1)];
                    else vin[nb][t] = a.in[(tile_base + s_t) * KIN + col];
                }
            }
        }
        // ---- dH_last = Wo^T G_out (16 rows of Wo: k = 16), masked; pieces to the tile and into registers
        LSE_PHASE("dHlast");
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f32x4 dh[2][CT];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int rb = 2 * s2 + b;
                u32x4 wc[3];
                combos16_lds([&](int p) { return lds_tr_read(imgWo + (p * C::WO_PIECE + so_t[rb])); }, wc);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
                    c = LSE_MFMA_BF(wc[2], gx[ct].c[2], c);
                    c = LSE_MFMA_BF(wc[1], gx[ct].c[1], c);
                    dh[b][ct] = LSE_MFMA_BF(wc[0], gx[ct].c[0], c);
                    const uint32_t m = (NHL == 2) ? m1[ct] : m0[ct];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[b][ct][r] = relu_gate(dh[b][ct][r], m, 4 * rb + r);
                }
                split_block_single(s2, b, dh[b], x);
                if constexpr (!MS) {
                    write_buf_rb(rb, x);
                    LSE_X6_SB();
                }
            }
            split_blocks(s2, dh[0], dh[1], x);
            if constexpr (MS) {
                write_buf_rb(2 * s2, x);
                write_buf_rb(2 * s2 + 1, x);
            }
            LSE_X6_SB();
        }
        if constexpr (NHL == 2) {
            lds_sync();
            LSE_PHASE("dW1");
            // ---- dW1 += G1^T * H0, with H0^T computed directly in operand layout (rows = samples): in * W0^T + bias.
            // (the pieces of G1 in registers are dropped here and come back from the tile for dH0)
            {
                PiecesB16 xin16[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) split_in16(raw[ct][0], nI, xin16[ct]);
                // bias rows of samples 16ct + 4q + r: lane 4q + r of this lane's 16-lane group holds that sample's row
                int trow[CT][4] = {};
                if constexpr (BIAS) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) trow[ct][r] = __shfl(brow[ct], 4 * q + r, 16);
                }
                // bias^T of all four blocks requested at once (one exposed latency instead of four)
                f32x4 t0a[HB][CT];
#pragma unroll
                for (int nb = 0; nb < HB; ++nb)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t0a[nb][ct][r] = BIAS ? rbase[(unsigned)(trow[ct][r] * WIDTH + 16 * nb + j)] : 0.f;
#pragma unroll
                for (int nb = 0; nb < HB; ++nb) {
                    f32x4 t0[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) t0[ct] = t0a[nb][ct];
                    u32x4 wc[3];
                    combos16_lds([&](int p) { return lds_read8(imgW0 + (p * C::W0_PIECE + 512 * nb + sm_j)); }, wc);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        t0[ct] = LSE_MFMA_BF(xin16[ct].c[2], wc[2], t0[ct]);
                        t0[ct] = LSE_MFMA_BF(xin16[ct].c[1], wc[1], t0[ct]);
                        t0[ct] = LSE_MFMA_BF(xin16[ct].c[0], wc[0], t0[ct]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) t0[ct][r] = relu_bits(t0[ct][r]);
                    }
                    u32x4 hb[3];
                    if constexpr (CT == 2) {
                        split_pair(t0[0], t0[CT - 1], nI, hb);
                    } else {
                        uint32_t hi[2], mid[2], lo[2];
                        split_half(t0[0], nI, hi, mid, lo);
                        hb[0] = (u32x4){hi[0], hi[1], mid[0], mid[1]};
                        hb[1] = (u32x4){hi[0], hi[1], hi[0], hi[1]};
                        hb[2] = (u32x4){lo[0], lo[1], mid[0], mid[1]};
                    }
                    u32x4 ga[2][3];          // operand of block mb + 1 requested before the products of block mb are issued
                    tr_block(0, true, ga[0]);
#pragma unroll
                    for (int mb = 0; mb < HB; ++mb) {
                        if (mb + 1 < HB) tr_block(mb + 1, true, ga[(mb + 1) & 1]);
                        acc1[mb][nb] = wg_mfma(ga[mb & 1], hb, acc1[mb][nb]);
                        LSE_X6_SB();
                    }
                }
            }
            // ---- dH0 = W1^T G1, masked, one output block at a time; its pieces replace G1's in the tile
            LSE_PHASE("dH0");
            read_buf_own(x);
            lds_sync();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f32x4 d0[2][CT];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int cb = 2 * s2 + b;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) d0[b][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        u32x4 wa[3];
                        tr_chain_w1(cb, s, wa);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) d0[b][ct] = mfma6(wa, x[ct].p[s], d0[b][ct]);
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d0[b][ct][r] = relu_gate(d0[b][ct][r], m0[ct], 4 * cb + r);
                    if constexpr (!MS) {
                        PiecesB<HB> y1[CT];
                        split_block_single(s2, b, d0[b], y1);
                        write_buf_rb(cb, y1);
                        LSE_X6_SB();
                    }
                }
                if constexpr (MS) {
                    PiecesB<HB> y[CT];
                    split_blocks(s2, d0[0], d0[1], y);
                    write_buf_rb(2 * s2, y);
                    write_buf_rb(2 * s2 + 1, y);
                }
                LSE_X6_SB();
            }
            read_buf_own(x);         // dH0 pieces (B operand of dIn)
        }
        // ---- per-row bias gradient of a column tile that straddles two rows (rare): segmented scan on dH0 = hi + mid + lo
        LSE_PHASE("bias");
        float ones[CT] = {};
        if (BIAS && a.d_row_bias) {
            bool col_taken = false;          // both column tiles feed ONE product: only one row per tile may use the column
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int row = brow[ct];            // (d_row_bias needs row_bias_idx: lse_mlp_bwd checks)
                if constexpr (BIAS_ONES) {
                    const int first = __builtin_amdgcn_readfirstlane(row);
                    if (__builtin_amdgcn_ballot_w64(row != first) == 0) {     // one row owns this column tile
                        if (first == cur_row) {
                            ones[ct] = 1.f;
                            col_taken = true;
                        } else if (!col_taken) {
                            flush_bias_col();
                            cur_row = first;
                            ones[ct] = 1.f;
                            col_taken = true;
                        }
                    }
                }
                if (!BIAS_ONES || ones[ct] == 0.f) {
                    f32x4 gcol[HB];
#pragma unroll
                    for (int rb = 0; rb < HB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int w2 = 2 * (rb & 1) + (r >> 1), sh = (r & 1) ? 0 : 16;
                            const uint32_t hi = x[ct].p[rb >> 1][0][w2], mid = x[ct].p[rb >> 1][1][w2], lo = x[ct].p[rb >> 1][2][w2];
                            gcol[rb][r] = __uint_as_float((hi << sh) & 0xffff0000u) + __uint_as_float((mid << sh) & 0xffff0000u) +
                                          __uint_as_float((lo << sh) & 0xffff0000u);
                        }
                    row_bias_grad_tile<HB, WIDTH>(a.d_row_bias, row, gcol, j, q);
                }
            }
        }
        // layer-0 input as the B operand of dW0 (lane = input column, slots = samples)
        LSE_PHASE("dW0");
        u32x4 ib[KB0][3];
        if constexpr (KIN == 16) {
            // through the small tile (the output gradient there has been consumed)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                Halves h;
                split4(raw[ct][0], h);
                write_sm(ct, h);
            }
            lds_sync();
            tr_small(false, ib[0]);
            if constexpr (BIAS_ONES) {
                // the unweighted input column 0 carries the constant 1 of the samples whose column tile belongs to one row
                const uint32_t o0 = ones[0] != 0.f ? 0x3F803F80u : 0u, o1 = ones[CT - 1] != 0.f ? 0x3F803F80u : 0u;
                if (j == 0) {
                    if constexpr (CT == 2) {
                        ib[0][0] = (u32x4){o0, o0, o1, o1};
                        ib[0][1] = (u32x4){0u, 0u, 0u, 0u};
                        ib[0][2] = (u32x4){0u, 0u, 0u, 0u};
                    } else {            // windows (hi|mid), (hi|hi), (lo|mid) of the constant: hi = 1, mid = lo = 0
                        ib[0][0] = (u32x4){o0, o0, 0u, 0u};
                        ib[0][1] = (u32x4){o0, o0, o0, o0};
                        ib[0][2] = (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
        } else {
            lds_sync();
#pragma unroll
            for (int nb = 0; nb < KB0; ++nb) {
                const float (&v)[4 * CT] = vin[nb];
                if constexpr (CT == 2) {
                    split_pair((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4 * CT - 4], v[4 * CT - 3], v[4 * CT - 2], v[4 * CT - 1]}, nI, ib[nb]);
                } else {
                    uint32_t hi[2], mid[2], lo[2];
                    split_half((f32x4){v[0], v[1], v[2], v[3]}, nI, hi, mid, lo);
                    ib[nb][0] = (u32x4){hi[0], hi[1], mid[0], mid[1]};
                    ib[nb][1] = (u32x4){hi[0], hi[1], hi[0], hi[1]};
                    ib[nb][2] = (u32x4){lo[0], lo[1], mid[0], mid[1]};
                }
            }
        }
        {
            u32x4 ga[2][3];
            tr_block(0, true, ga[0]);
#pragma unroll
            for (int mb = 0; mb < HB; ++mb) {
                if (mb + 1 < HB) tr_block(mb + 1, true, ga[(mb + 1) & 1]);
#pragma unroll
                for (int nb = 0; nb < KB0; ++nb) acc0[mb][nb] = wg_mfma(ga[mb & 1], ib[nb], acc0[mb][nb]);
                LSE_X6_SB();
            }
        }
        // ---- dIn = W0^T dH0
        LSE_PHASE("dIn");
        if (a.d_in) {
#pragma unroll
            for (int cb = 0; cb < KB0; ++cb) {
                f32x4 di[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) di[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 wa[3];
                    tr_chain_w0(cb, s, wa);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) di[ct] = mfma6(wa, x[ct].p[s], di[ct]);
                }
                LSE_X6_SB();
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (!valid[ct]) continue;
                    if (INL == LSE_IN_LEVELMAJOR) {
                        float2 *d2 = reinterpret_cast<float2 *>(a.d_in) + tile_base;
                        const int lv = 8 * cb + 2 * q;
                        d2[(int64_t)lv * ns + sl[ct]] = make_float2(di[ct][0], di[ct][1]);
                        d2[(int64_t)(lv + 1) * ns + sl[ct]] = make_float2(di[ct][2], di[ct][3]);
                    } else {
                        *reinterpret_cast<f32x4 *>(a.d_in + tile_base * KIN + (unsigned)(sl[ct] * KIN + 16 * cb + 4 * q)) = di[ct];
                    }
                }
            }
        }
        lds_sync();
        LSE_PHASE("end");
    }
    if (BIAS_ONES) flush_bias_col();
    // Weight gradients: the NW waves of the workgroup first sum their accumulators through LDS (the weight images and tiles are dead
    // by now) -- a tree of plain 16-byte writes and reads, three rounds for eight waves -- and ONE wave adds the sums to memory.
    // Until round 5 every wave sent its own: 2048 waves x 384 (head) / 192 (base) 64-byte float-atomic requests onto the same
    // few hundred lines, 1.2 M requests per step at the memory side's 21 G/s = the better part of the ~60 us a launch of this
    // kernel cost whatever its size (tools/mlp_bwd_fixed_cost.py).
    constexpr int SLOTS = HB * KB0 + ((NHL == 2) ? HB * HB : 0) + HB;
    // (development shapes with 12 waves per workgroup keep the per-wave flush)
    if constexpr (NW > 1 && (NW & (NW - 1)) == 0 && (NW / 2) * SLOTS * 64 * 16 <= C::lds_bytes) {
        f32x4 *red = reinterpret_cast<f32x4 *>(lds);
        auto each_acc = [&](auto fn) {      // fn(accumulator tile, slot)
            int slot = 0;
#pragma unroll
            for (int mb = 0; mb < HB; ++mb)
#pragma unroll
                for (int kb = 0; kb < KB0; ++kb) fn(acc0[mb][kb], slot++);
            if constexpr (NHL == 2) {
#pragma unroll
                for (int mb = 0; mb < HB; ++mb)
#pragma unroll
                    for (int kb = 0; kb < HB; ++kb) fn(acc1[mb][kb], slot++);
            }
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) fn(accO[0][kb], slot++);
        };
        __syncthreads();      // every wave is done with the weight images and its tiles
#pragma unroll
        for (int half = NW / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
                f32x4 *dst = red + (size_t)(wave - half) * SLOTS * 64 + lane;
                each_acc([&](f32x4 &t, int slot) { dst[slot * 64] = t; });
            }
            __syncthreads();
            if (wave < half) {
                const f32x4 *src = red + (size_t)wave * SLOTS * 64 + lane;
                each_acc([&](f32x4 &t, int slot) { t += src[slot * 64]; });
            }
            if (half > 1) __syncthreads();      // the next round's writers wait for this round's readers
        }
        if (wave != 0) return;
    }
    flush_wgrad<HB, KB0>(a.d_params + a.w0_col, a.w0_ld, KIN, acc0, j, q, a.w0_mask0 != 0);
    if constexpr (NHL == 2) flush_wgrad<HB, HB>(a.d_params + a.rest_off, WIDTH, WIDTH, acc1, j, q);
    flush_wgrad<1, HB>(a.d_params + a.rest_off + (NHL - 1) * WIDTH * WIDTH, WIDTH, WIDTH, accO, j, q);
}
