// Glue kernels of the frozen VGG19 perceptual branch (/root/reference/utils/losses.py:83-151): the
// convolutions themselves are conv_igemm_kernel launches (bias + ReLU epilogue forward, relu_mask epilogue
// for the input gradient); this file holds what sits between them.  All HBM-bound element-wise passes on
// NHWC tensors, 16-byte vectors per lane.
//   vgg_input_fwd / bwd : gray -> 3 channel repeat + ImageNet normalise (losses.py:105-114) and its adjoint
//   maxpool2_fwd / bwd  : nn.MaxPool2d(2) of the VGG stack (features[4,9,18,27]) and its adjoint fused with
//                         the ReLU backward of the layer in front of it
//   feature_loss        : nn.L1Loss / nn.MSELoss between the two feature maps (losses.py:127-131,150)
#include "common.h"

constexpr int kVggInC = 8;   // stored channels of the normalised input (3 real + zero padding to a 16-B bf16 vector)

struct VggNorm { float mean[3], inv_std[3]; };

template <typename T>
__global__ __launch_bounds__(256) void vgg_input_fwd_kernel(const float* __restrict__ x, T* __restrict__ out,
                                                            size_t npix, VggNorm nm) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        T* o = out + i * kVggInC;
#pragma unroll
        for (int c = 0; c < kVggInC; ++c) o[c] = from_f32<T>(c < 3 ? (v - nm.mean[c]) * nm.inv_std[c] : 0.f);
    }
}

// dimg[i] += gscale * scale * sum_c dx3[i][c] / std_c
template <typename T>
__global__ __launch_bounds__(256) void vgg_input_bwd_kernel(const T* __restrict__ dx3, const float* __restrict__ gscale,
                                                            float scale, float* __restrict__ dimg, size_t npix,
                                                            VggNorm nm) {
    const float sc = scale * (gscale ? gscale[0] : 1.f);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const T* d = dx3 + i * kVggInC;
        const float g = to_f32(d[0]) * nm.inv_std[0] + to_f32(d[1]) * nm.inv_std[1] + to_f32(d[2]) * nm.inv_std[2];
        dimg[i] += sc * g;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int H,
                                                           int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int nvec = C / VEC, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * nvec;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cv = idx % nvec;
        size_t r = idx / nvec;
        const int xo = r % Wo; r /= Wo;
        const int yo = r % Ho;
        const int n = r / Ho;
        const T* b = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + cv * VEC;
        const Vec16<T> v00 = load_vec16(b), v01 = load_vec16(b + C), v10 = load_vec16(b + (size_t)W * C),
                       v11 = load_vec16(b + (size_t)W * C + C);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, fmaxf(fmaxf(v00.get(e), v01.get(e)), fmaxf(v10.get(e), v11.get(e))));
        store_vec16(out + idx * VEC, o);
    }
}

// dx[window] = dy at the FIRST maximum of the window (aten max_pool2d_with_indices scans row-major and keeps
// the first), zero elsewhere; with relu_gate the result is also zeroed where x <= 0 (x is a ReLU output).
// Rows / columns beyond 2*(H/2), 2*(W/2) (odd sizes) receive zero.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           T* __restrict__ dx, int N, int H, int W, int C,
                                                           int relu_gate) {
    constexpr int VEC = Vec16<T>::N;
    const int nvec = C / VEC, Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
    const size_t total = (size_t)N * Hc * Wc * nvec;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cv = idx % nvec;
        size_t r = idx / nvec;
        const int xo = r % Wc; r /= Wc;
        const int yo = r % Hc;
        const int n = r / Hc;
        const size_t base = (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + cv * VEC;
        const bool full = yo < Ho && xo < Wo;
        if (!full) {   // ragged border of an odd-sized input: outside every pooling window
            Vec16<T> z;
            z.zero();
            store_vec16(dx + base, z);
            if (2 * xo + 1 < W) store_vec16(dx + base + C, z);
            if (2 * yo + 1 < H) {
                store_vec16(dx + base + (size_t)W * C, z);
                if (2 * xo + 1 < W) store_vec16(dx + base + (size_t)W * C + C, z);
            }
            continue;
        }
        const Vec16<T> v00 = load_vec16(x + base), v01 = load_vec16(x + base + C),
                       v10 = load_vec16(x + base + (size_t)W * C), v11 = load_vec16(x + base + (size_t)W * C + C);
        const Vec16<T> g = load_vec16(dy + (((size_t)n * Ho + yo) * Wo + xo) * C + cv * VEC);
        Vec16<T> o00, o01, o10, o11;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float a = v00.get(e), b = v01.get(e), c = v10.get(e), d = v11.get(e);
            int arg = 0;
            float m = a;
            if (b > m) { m = b; arg = 1; }
            if (c > m) { m = c; arg = 2; }
            if (d > m) { m = d; arg = 3; }
            const float gv = (relu_gate && !(m > 0.f)) ? 0.f : g.get(e);
            o00.set(e, arg == 0 ? gv : 0.f);
            o01.set(e, arg == 1 ? gv : 0.f);
            o10.set(e, arg == 2 ? gv : 0.f);
            o11.set(e, arg == 3 ? gv : 0.f);
        }
        store_vec16(dx + base, o00);
        store_vec16(dx + base + C, o01);
        store_vec16(dx + base + (size_t)W * C, o10);
        store_vec16(dx + base + (size_t)W * C + C, o11);
    }
}

// kind 0: sum |a-b|, grad sign(a-b);  kind 1: sum (a-b)^2, grad 2(a-b).  The gradient is written UNSCALED
// (the 1/numel of the mean and the upstream gradient are applied in fp32 at the end of the input-gradient
// chain, which is linear), optionally gated by a > 0 (a is then the ReLU output the loss is taken on).
template <typename T>
__global__ __launch_bounds__(256) void feature_loss_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                           size_t nvecs, int kind, double* __restrict__ sum,
                                                           T* __restrict__ da, int relu_gate) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvecs; i += (size_t)gridDim.x * blockDim.x) {
        const Vec16<T> va = load_vec16(a + i * VEC), vb = load_vec16(b + i * VEC);
        Vec16<T> g;
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float x = va.get(e), d = x - vb.get(e);
            float gv;
            if (kind == 0) { part += fabsf(d); gv = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
            else { part += d * d; gv = 2.f * d; }
            if (relu_gate && !(x > 0.f)) gv = 0.f;
            g.set(e, gv);
        }
        acc += (double)part;
        if (da) store_vec16(da + i * VEC, g);
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f64(sum + (blockIdx.x & 15), red[0] + red[1] + red[2] + red[3]);
}

// out[0] = (sum over the 16 slots) / numel
__global__ void feature_loss_finalize_kernel(const double* __restrict__ sum, double numel, float* __restrict__ out) {
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += sum[i];
    out[0] = (float)(s / numel);
}

static VggNorm vgg_norm() {   // losses.py:7-8
    VggNorm nm;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, std[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean[c]; nm.inv_std[c] = 1.f / std[c]; }
    return nm;
}

static int grid_for(size_t total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

extern "C" int mrisr_vgg_input_channels(void) { return kVggInC; }

extern "C" int mrisr_vgg_input_forward(int dtype, const float* x, void* out, size_t npix, void* stream) {
    if (!x || !out) MRISR_FAIL(MRISR_E_ARG, "vgg_input_forward: null pointer");
    if (!npix) return MRISR_OK;
    if (dtype == MRISR_BF16) vgg_input_fwd_kernel<bf16_t><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)out, npix, vgg_norm());
    else if (dtype == MRISR_F16) vgg_input_fwd_kernel<f16_t><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>(x, (f16_t*)out, npix, vgg_norm());
    else if (dtype == MRISR_F32) vgg_input_fwd_kernel<float><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>(x, (float*)out, npix, vgg_norm());
    else MRISR_FAIL(MRISR_E_DTYPE, "vgg_input_forward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("vgg_input_forward");
    return MRISR_OK;
}

extern "C" int mrisr_vgg_input_backward(int dtype, const void* dx3, const float* gscale, float scale, float* dimg,
                                        size_t npix, void* stream) {
    if (!dx3 || !dimg) MRISR_FAIL(MRISR_E_ARG, "vgg_input_backward: null pointer");
    if (!npix) return MRISR_OK;
    if (dtype == MRISR_BF16) vgg_input_bwd_kernel<bf16_t><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dx3, gscale, scale, dimg, npix, vgg_norm());
    else if (dtype == MRISR_F16) vgg_input_bwd_kernel<f16_t><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>((const f16_t*)dx3, gscale, scale, dimg, npix, vgg_norm());
    else if (dtype == MRISR_F32) vgg_input_bwd_kernel<float><<<grid_for(npix), 256, 0, (hipStream_t)stream>>>((const float*)dx3, gscale, scale, dimg, npix, vgg_norm());
    else MRISR_FAIL(MRISR_E_DTYPE, "vgg_input_backward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("vgg_input_backward");
    return MRISR_OK;
}

extern "C" int mrisr_maxpool2_forward(int dtype, const void* x, void* out, int N, int H, int W, int C, void* stream) {
    if (!x || !out) MRISR_FAIL(MRISR_E_ARG, "maxpool2_forward: null pointer");
    const int vec = mrisr_vec(dtype);
    if (N <= 0 || C <= 0 || C % vec || H < 2 || W < 2) MRISR_FAIL(MRISR_E_SHAPE, "maxpool2_forward: N %d C %d H %d W %d", N, C, H, W);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / vec);
    if (dtype == MRISR_BF16) maxpool2_fwd_kernel<bf16_t><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)out, N, H, W, C);
    else if (dtype == MRISR_F16) maxpool2_fwd_kernel<f16_t><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const f16_t*)x, (f16_t*)out, N, H, W, C);
    else if (dtype == MRISR_F32) maxpool2_fwd_kernel<float><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const float*)x, (float*)out, N, H, W, C);
    else MRISR_FAIL(MRISR_E_DTYPE, "maxpool2_forward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("maxpool2_forward");
    return MRISR_OK;
}

extern "C" int mrisr_maxpool2_backward(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                                       int relu_gate, void* stream) {
    if (!x || !dy || !dx) MRISR_FAIL(MRISR_E_ARG, "maxpool2_backward: null pointer");
    const int vec = mrisr_vec(dtype);
    if (N <= 0 || C <= 0 || C % vec || H < 2 || W < 2) MRISR_FAIL(MRISR_E_SHAPE, "maxpool2_backward: N %d C %d H %d W %d", N, C, H, W);
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
    if (dtype == MRISR_BF16) maxpool2_bwd_kernel<bf16_t><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, relu_gate);
    else if (dtype == MRISR_F16) maxpool2_bwd_kernel<f16_t><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const f16_t*)x, (const f16_t*)dy, (f16_t*)dx, N, H, W, C, relu_gate);
    else if (dtype == MRISR_F32) maxpool2_bwd_kernel<float><<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const float*)x, (const float*)dy, (float*)dx, N, H, W, C, relu_gate);
    else MRISR_FAIL(MRISR_E_DTYPE, "maxpool2_backward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("maxpool2_backward");
    return MRISR_OK;
}

extern "C" int mrisr_feature_loss(int dtype, const void* a, const void* b, size_t n, int kind, double* sum16,
                                  float* out, void* da, int relu_gate, void* stream) {
    if (!a || !b || !sum16 || !out) MRISR_FAIL(MRISR_E_ARG, "feature_loss: null pointer");
    if (kind != 0 && kind != 1) MRISR_FAIL(MRISR_E_ARG, "feature_loss: kind %d", kind);
    const int vec = mrisr_vec(dtype);
    if (!n || n % vec) MRISR_FAIL(MRISR_E_SHAPE, "feature_loss: %zu elements not a multiple of %d", n, vec);
    const size_t nvecs = n / vec;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sum16, 0, 16 * sizeof(double), s) != hipSuccess) MRISR_FAIL(MRISR_E_HIP, "feature_loss: memset");
    const int blocks = (int)((nvecs + 255) / 256 < 2048 ? (nvecs + 255) / 256 : 2048);
    if (dtype == MRISR_BF16) feature_loss_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)b, nvecs, kind, sum16, (bf16_t*)da, relu_gate);
    else if (dtype == MRISR_F16) feature_loss_kernel<f16_t><<<blocks, 256, 0, s>>>((const f16_t*)a, (const f16_t*)b, nvecs, kind, sum16, (f16_t*)da, relu_gate);
    else if (dtype == MRISR_F32) feature_loss_kernel<float><<<blocks, 256, 0, s>>>((const float*)a, (const float*)b, nvecs, kind, sum16, (float*)da, relu_gate);
    else MRISR_FAIL(MRISR_E_DTYPE, "feature_loss: dtype %d", dtype);
    feature_loss_finalize_kernel<<<1, 1, 0, s>>>(sum16, (double)n, out);
    MRISR_CHECK_LAUNCH("feature_loss");
    return MRISR_OK;
}
