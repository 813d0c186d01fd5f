// Fused Adam over one flat fp32 parameter buffer (gfx950, HBM-bound: 16 B read + 12 B written per element).
//
// Replaces torch.optim.Adam(lr, weight_decay) as used by /root/reference/scripts/train.py:186,318:
// betas (0.9, 0.999), eps 1e-8, L2 weight decay ADDED TO THE GRADIENT (not AdamW), bias correction
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
// The flat buffer is also the RCCL all-reduce bucket; grad_scale folds the 1/world_size of the mean.
#include "common.h"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                   float gscale) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * gscale + wd * pv[e];
                mv[e] = b1 * mv[e] + (1.f - b1) * gg;
                vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
                pv[e] -= (lr / bc1) * mv[e] / (sqrtf(vv[e]) * rsqrt_bc2 + eps);
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            for (size_t j = i; j < n; ++j) {
                const float gg = g[j] * gscale + wd * p[j];
                m[j] = b1 * m[j] + (1.f - b1) * gg;
                v[j] = b2 * v[j] + (1.f - b2) * gg * gg;
                p[j] -= (lr / bc1) * m[j] / (sqrtf(v[j]) * rsqrt_bc2 + eps);
            }
        }
    }
}

extern "C" int mrisr_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v) MRISR_FAIL(MRISR_E_ARG, "adam_step: null pointer");
    if (step < 1) MRISR_FAIL(MRISR_E_ARG, "adam_step: step %d < 1", step);
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) MRISR_FAIL(MRISR_E_ARG, "adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const size_t nv = (n + 3) / 4;
    const int blocks = (int)((nv + 255) / 256 < 2048 ? (nv + 255) / 256 : 2048);
    adam_kernel<<<blocks ? blocks : 1, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                                                                     (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    MRISR_CHECK_LAUNCH("adam_step");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// fp16 autocast (scripts/train.py:303-311: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()):
// the step with torch.amp.GradScaler's device-side contract - gradients are UNSCALED by 1 / *loss_scale inside the
// kernel, the whole update is skipped when *found_inf != 0, and the bias-correction step count lives on the device
// (*step, advanced by a one-thread kernel only when the update ran), so nothing is read back by the host.
__global__ __launch_bounds__(256) void adam_amp_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                       float b1, float b2, float eps, float wd, float gmul,
                                                       const int* __restrict__ step, const float* __restrict__ loss_scale,
                                                       const float* __restrict__ found_inf) {
    if (found_inf && found_inf[0] != 0.f) return;
    const float t = (float)(step[0] + 1);
    const float bc1 = 1.f - powf(b1, t), rsqrt_bc2 = rsqrtf(1.f - powf(b2, t));
    const float gscale = gmul / (loss_scale ? loss_scale[0] : 1.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        const size_t e1 = i + 4 < n ? i + 4 : n;
        for (size_t j = i; j < e1; ++j) {
            const float gg = g[j] * gscale + wd * p[j];
            const float mj = b1 * m[j] + (1.f - b1) * gg, vj = b2 * v[j] + (1.f - b2) * gg * gg;
            m[j] = mj;
            v[j] = vj;
            p[j] -= (lr / bc1) * mj / (sqrtf(vj) * rsqrt_bc2 + eps);
        }
    }
}
__global__ void adam_amp_advance_kernel(int* step, const float* found_inf) {
    if (!(found_inf && found_inf[0] != 0.f)) step[0] += 1;
}

extern "C" int mrisr_adam_step_amp(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int* step_device, float grad_mul,
                                   const float* loss_scale_device, const float* found_inf_device, void* stream) {
    if (!p || !g || !m || !v || !step_device) MRISR_FAIL(MRISR_E_ARG, "adam_step_amp: null pointer");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) MRISR_FAIL(MRISR_E_ARG, "adam_step_amp: buffers must be 16-byte aligned");
    const size_t nv = (n + 3) / 4;
    const int blocks = (int)((nv + 255) / 256 < 2048 ? (nv + 255) / 256 : 2048);
    adam_amp_kernel<<<blocks ? blocks : 1, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                                                                         grad_mul, step_device, loss_scale_device, found_inf_device);
    MRISR_CHECK_LAUNCH("adam_step_amp");
    adam_amp_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(step_device, found_inf_device);
    MRISR_CHECK_LAUNCH("adam_step_amp(advance)");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = from_f32<D>(to_f32(s[i]));
}

extern "C" int mrisr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, size_t n, void* stream) {
    if (!src || !dst) MRISR_FAIL(MRISR_E_ARG, "cast: null pointer");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == MRISR_F32 && dst_dtype == MRISR_BF16) cast_kernel<float, bf16_t><<<blocks, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == MRISR_BF16 && dst_dtype == MRISR_F32) cast_kernel<bf16_t, float><<<blocks, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == MRISR_F32 && dst_dtype == MRISR_F32) cast_kernel<float, float><<<blocks, 256, 0, s>>>((const float*)src, (float*)dst, n);
    else if (src_dtype == MRISR_BF16 && dst_dtype == MRISR_BF16) cast_kernel<bf16_t, bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n);
    else if (src_dtype == MRISR_F32 && dst_dtype == MRISR_F16) cast_kernel<float, f16_t><<<blocks, 256, 0, s>>>((const float*)src, (f16_t*)dst, n);
    else if (src_dtype == MRISR_F16 && dst_dtype == MRISR_F32) cast_kernel<f16_t, float><<<blocks, 256, 0, s>>>((const f16_t*)src, (float*)dst, n);
    else if (src_dtype == MRISR_F16 && dst_dtype == MRISR_F16) cast_kernel<f16_t, f16_t><<<blocks, 256, 0, s>>>((const f16_t*)src, (f16_t*)dst, n);
    else MRISR_FAIL(MRISR_E_DTYPE, "cast: dtypes %d -> %d", src_dtype, dst_dtype);
    MRISR_CHECK_LAUNCH("cast");
    return MRISR_OK;
}
