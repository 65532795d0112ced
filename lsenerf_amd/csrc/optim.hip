// Flat-buffer Adam (torch.optim.Adam semantics, R:lse_nerf/lse_config.py:29-33: lr 1e-2, eps 1e-15) and the
// occupancy-grid EMA update (nerfacc OccGridEstimator._update).  Pure HBM streams:
// 16-byte vector accesses, grid capped at 2048 workgroups with a grid-stride loop.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float inv_sqrt_bc2,
                                                   float gscale, const float *__restrict__ hyper)
{
    if (hyper != nullptr) {      // every scalar of the step from device memory (a captured launch cannot carry new ones)
        lr = hyper[0];
        bc1 = hyper[1];
        inv_sqrt_bc2 = hyper[2];
        b1 = hyper[3];
        b2 = hyper[4];
        eps = hyper[5];
    }
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4 *>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4 *>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4 *>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gv[k] * gscale;
            mv[k] = b1 * mv[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
            pv[k] -= step * (mv[k] / denom);
        }
        reinterpret_cast<f32x4 *>(p)[i] = pv;
        reinterpret_cast<f32x4 *>(m)[i] = mv;
        reinterpret_cast<f32x4 *>(v)[i] = vv;
    }
    // tail
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gk = g[t] * gscale;
        const float mk = b1 * m[t] + (1.f - b1) * gk;
        const float vk = b2 * v[t] + (1.f - b2) * gk * gk;
        m[t] = mk;
        v[t] = vk;
        p[t] -= step * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
    }
}

// One thread: advance the device-side step counter and derive the step's scalars from it, in double like the host does for
// lse_adam_step.  Captured into a HIP graph in front of adam_kernel, a replayed step needs nothing from the host: there is no
// staging buffer a host that runs ahead of the device could overwrite before the queued copy has executed.
__global__ void adam_schedule_kernel(int64_t *__restrict__ step, float *__restrict__ hyper, const double *__restrict__ sched)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // the schedule's constants live in device memory too: a host that changes them (a loaded param group, a manual lr drop)
    // rewrites six doubles and every later replay follows
    const double lr_init = sched[0], lr_final = sched[1], b1 = sched[3], b2 = sched[4];
    const int64_t max_steps = (int64_t)sched[2];
    const int64_t done = *step;                 // optimizer steps taken so far: the learning rate is the one of step `done`
    const int64_t t = done + 1;
    *step = t;
    double lr = lr_init;
    if (max_steps > 0 && lr_final > 0.0) {      // nerfstudio ExponentialDecayScheduler without warm-up
        const double f = fmin(fmax((double)done / (double)max_steps, 0.0), 1.0);
        lr = exp(log(lr_init) * (1.0 - f) + log(lr_final) * f);
    }
    hyper[0] = (float)lr;
    hyper[1] = (float)(1.0 - pow(b1, (double)t));
    hyper[2] = (float)(1.0 / sqrt(1.0 - pow(b2, (double)t)));
    hyper[3] = (float)b1;
    hyper[4] = (float)b2;
    hyper[5] = (float)sched[5];
}

// occs[id] = max(occs[id]*ema, occ_new) with duplicate ids resolved as the maximum over the duplicates:
//   pass 1: ws[i] = max(occs[id_i]*ema, occ_i)      (reads only)
//   pass 2: occs[id_i] = 0                           (benign same-value race)
//   pass 3: atomic max (values are >= 0, so the int ordering equals the float ordering)
__global__ void occ_pass1(const float *__restrict__ occs, const int64_t *__restrict__ ids, const float *__restrict__ nw,
                          int64_t n, float ema, float *__restrict__ ws)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = fmaxf(occs[ids[i]] * ema, nw[i]);
}
__global__ void occ_pass2(float *__restrict__ occs, const int64_t *__restrict__ ids, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) occs[ids[i]] = 0.f;
}
__global__ void occ_pass3(float *__restrict__ occs, const int64_t *__restrict__ ids, const float *__restrict__ ws,
                          int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(reinterpret_cast<int *>(occs) + ids[i], __float_as_int(fmaxf(ws[i], 0.f)));
}

__global__ void occ_binarize_kernel(const float *__restrict__ occs, int64_t n, const float *__restrict__ thre,
                                    uint8_t *__restrict__ bin)
{
    const float t = *thre;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bin[i] = occs[i] > t ? 1 : 0;
}

}  // namespace

extern "C" int lse_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, float grad_scale, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0 && step >= 1, "lse_adam_step: need n >= 0 and step >= 1");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(params && grads && exp_avg && exp_avg_sq, "lse_adam_step: null pointer");
    LSE_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "lse_adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t n4 = (n + 3) / 4;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, 2048));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, lse::as_stream(stream), params, grads, exp_avg, exp_avg_sq,
                       n, lr, beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, (const float *)nullptr);
    return lse::check_launch("lse_adam_step");
}

extern "C" int lse_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 const float *hyper, float grad_scale, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_adam_step_dev: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(params && grads && exp_avg && exp_avg_sq && hyper, "lse_adam_step_dev: null pointer");
    LSE_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "lse_adam_step_dev: buffers must be 16-byte aligned");
    const int64_t n4 = (n + 3) / 4;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, 2048));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, lse::as_stream(stream), params, grads, exp_avg, exp_avg_sq,
                       n, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, grad_scale, hyper);
    return lse::check_launch("lse_adam_step_dev");
}

extern "C" int lse_adam_schedule_dev(int64_t *step, float *hyper, const double *sched, lse_stream_t stream)
{
    LSE_REQUIRE(step && hyper && sched, "lse_adam_schedule_dev: null pointer");
    hipLaunchKernelGGL(adam_schedule_kernel, dim3(1), dim3(64), 0, lse::as_stream(stream), step, hyper, sched);
    return lse::check_launch("lse_adam_schedule_dev");
}

extern "C" int lse_occ_update_cells(float *occs, const int64_t *cell_ids, const float *occ_new, int64_t n,
                                    float ema_decay, float *workspace, lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_occ_update_cells: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(occs && cell_ids && occ_new && workspace, "lse_occ_update_cells: null pointer");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipStream_t st = lse::as_stream(stream);
    hipLaunchKernelGGL(occ_pass1, dim3(blocks), dim3(256), 0, st, occs, cell_ids, occ_new, n, ema_decay, workspace);
    hipLaunchKernelGGL(occ_pass2, dim3(blocks), dim3(256), 0, st, occs, cell_ids, n);
    hipLaunchKernelGGL(occ_pass3, dim3(blocks), dim3(256), 0, st, occs, cell_ids, workspace, n);
    return lse::check_launch("lse_occ_update_cells");
}

extern "C" int lse_occ_binarize(const float *occs, int64_t n, const float *d_threshold, uint8_t *binaries,
                                lse_stream_t stream)
{
    LSE_REQUIRE(n >= 0, "lse_occ_binarize: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(occs && d_threshold && binaries, "lse_occ_binarize: null pointer");
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(occ_binarize_kernel, dim3(blocks), dim3(256), 0, lse::as_stream(stream), occs, n, d_threshold,
                       binaries);
    return lse::check_launch("lse_occ_binarize");
}
