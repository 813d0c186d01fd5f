// Device-side pre- and post-processing of the inference driver (gfx950), SURVEY.md 8(f) rank 3.
//
// Replaces the numpy code either side of the forward in /root/reference/scripts/infer.py:
//   :107-117  np.percentile(img, 0.5 / 99.5) -> np.clip -> (img - lo) / (hi - lo)       (8-bit grayscale PNG input)
//   :276,331  clamp(0, 1) -> (x * 255).astype(np.uint8)
// An 8-bit image has at most 256 distinct values, so the percentiles come exactly out of a 256-bin histogram and the
// whole normalisation is a 256-entry look-up table: one pass builds the histograms (LDS atomics, one global atomic per
// bin and block), the second one derives each image's table and maps the pixels.  The arithmetic follows numpy's
// float32 path operation by operation (np.percentile(method='linear') on a float32 array: virtual index (n-1) q / 100 in
// float32, _lerp's two branches; then float32 subtract and divide) - checked against numpy in
// tests/test_gpu_image.py.
#include "common.h"

__global__ __launch_bounds__(256) void u8_histogram_kernel(const uint8_t* __restrict__ img, size_t n, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    const int t = threadIdx.x, b = blockIdx.y;
    h[t] = 0;
    __syncthreads();
    const uint8_t* p = img + (size_t)b * n;
    // 16 bytes per thread and iteration where aligned; the byte tail by the last threads
    const size_t nv = ((uintptr_t)p & 15) == 0 ? n / 16 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + t; i < nv; i += (size_t)gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + i * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int s = 0; s < 32; s += 8) atomicAdd(&h[(v[k] >> s) & 255u], 1u);
    }
    for (size_t i = nv * 16 + (size_t)blockIdx.x * 256 + t; i < n; i += (size_t)gridDim.x * 256) atomicAdd(&h[p[i]], 1u);
    __syncthreads();
    if (h[t]) atomicAdd(&hist[(size_t)b * 256 + t], h[t]);
}

// value of rank k (0-based) in the sorted image, from the inclusive cumulative histogram in LDS
__device__ __forceinline__ float rank_value(const unsigned* cum, unsigned long long k) {
    int lo = 0, hi = 255;
    while (lo < hi) {          // first bin whose cumulative count exceeds k
        const int mid = (lo + hi) >> 1;
        if ((unsigned long long)cum[mid] > k) hi = mid; else lo = mid + 1;
    }
    return (float)lo;
}
// np.percentile(float32 array, q, method='linear'): numpy's _quantile + _lerp, float32 arithmetic, no contraction
// (for a float32 array numpy carries the quantile and the virtual index in float32 too: q32 = float32(q) / 100,
// virtual = float32(n - 1) * q32 - checked against np.percentile on 800 random 8-bit images in the build container)
__device__ __forceinline__ float np_percentile_u8(const unsigned* cum, size_t n, double q) {
    const float virt = __fmul_rn((float)(n - 1), __fdiv_rn((float)q, 100.f));
    const float prev = floorf(virt);
    unsigned long long k0 = (unsigned long long)prev, k1 = k0 + 1 < n ? k0 + 1 : n - 1;
    const float a = rank_value(cum, k0), b = rank_value(cum, k1);
    const float tg = __fsub_rn(virt, prev);
    const float diff = __fsub_rn(b, a);
    if (tg >= 0.5f) return __fsub_rn(b, __fmul_rn(diff, __fsub_rn(1.f, tg)));
    return __fadd_rn(a, __fmul_rn(diff, tg));
}

__global__ __launch_bounds__(256) void u8_percentile_normalise_kernel(const uint8_t* __restrict__ img, const unsigned* __restrict__ hist,
                                                                      size_t n, double q_lo, double q_hi, float* __restrict__ out,
                                                                      float* __restrict__ lohi) {
    __shared__ unsigned cum[256];
    __shared__ float lut[256];
    __shared__ float s_lo, s_hi;
    const int t = threadIdx.x, b = blockIdx.y;
    cum[t] = hist[(size_t)b * 256 + t];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {       // inclusive scan (256 entries: Hillis-Steele is fine)
        const unsigned v = t >= o ? cum[t - o] : 0u;
        __syncthreads();
        cum[t] += v;
        __syncthreads();
    }
    if (t == 0) {
        s_lo = np_percentile_u8(cum, n, q_lo);
        s_hi = np_percentile_u8(cum, n, q_hi);
        if (lohi && blockIdx.x == 0) { lohi[2 * b] = s_lo; lohi[2 * b + 1] = s_hi; }
    }
    __syncthreads();
    const float lo = s_lo, hi = s_hi;
    const float c = fminf(fmaxf((float)t, lo), hi);            // np.clip
    // infer.py:115-117: normalised only when max_val > min_val, else the clipped values stay
    lut[t] = hi > lo ? __fdiv_rn(__fsub_rn(c, lo), __fsub_rn(hi, lo)) : c;
    __syncthreads();
    const uint8_t* p = img + (size_t)b * n;
    float* o = out + (size_t)b * n;
    for (size_t i = (size_t)blockIdx.x * 256 + t; i < n; i += (size_t)gridDim.x * 256) o[i] = lut[p[i]];
}

__global__ __launch_bounds__(256) void f32_to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = fminf(fmaxf(x[i], 0.f), 1.f);           // infer.py:276 clamp; NaN -> 0 like fmaxf
        out[i] = (uint8_t)(int)__fmul_rn(v, 255.f);             // infer.py:331: (x * 255).astype(np.uint8) truncates
    }
}

static int blocks_for(size_t n, size_t per_block, int cap) {
    size_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? (size_t)cap : b));
}

extern "C" int mrisr_u8_histogram(const uint8_t* img, size_t pixels_per_image, int batch, unsigned* hist, void* stream) {
    if (!img || !hist) MRISR_FAIL(MRISR_E_ARG, "u8_histogram: null pointer");
    if (batch < 1 || batch > 65535 || pixels_per_image == 0) MRISR_FAIL(MRISR_E_SHAPE, "u8_histogram: batch %d, %zu pixels", batch, pixels_per_image);
    dim3 grid(blocks_for(pixels_per_image, 256 * 64, 256), batch);
    u8_histogram_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img, pixels_per_image, hist);
    MRISR_CHECK_LAUNCH("u8_histogram");
    return MRISR_OK;
}

extern "C" int mrisr_u8_percentile_normalise(const uint8_t* img, const unsigned* hist, size_t pixels_per_image, int batch,
                                             double q_lo, double q_hi, float* out, float* lohi, void* stream) {
    if (!img || !hist || !out) MRISR_FAIL(MRISR_E_ARG, "u8_percentile_normalise: null pointer");
    if (batch < 1 || batch > 65535 || pixels_per_image == 0) MRISR_FAIL(MRISR_E_SHAPE, "u8_percentile_normalise: batch %d, %zu pixels", batch, pixels_per_image);
    if (!(q_lo >= 0.0 && q_lo <= q_hi && q_hi <= 100.0)) MRISR_FAIL(MRISR_E_ARG, "u8_percentile_normalise: percentiles %g, %g", q_lo, q_hi);
    dim3 grid(blocks_for(pixels_per_image, 256 * 16, 512), batch);
    u8_percentile_normalise_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img, hist, pixels_per_image, q_lo, q_hi, out, lohi);
    MRISR_CHECK_LAUNCH("u8_percentile_normalise");
    return MRISR_OK;
}

extern "C" int mrisr_f32_to_u8(const float* x, uint8_t* out, size_t n, void* stream) {
    if (!x || !out) MRISR_FAIL(MRISR_E_ARG, "f32_to_u8: null pointer");
    if (n == 0) return MRISR_OK;
    f32_to_u8_kernel<<<blocks_for(n, 256 * 8, 4096), 256, 0, (hipStream_t)stream>>>(x, out, n);
    MRISR_CHECK_LAUNCH("f32_to_u8");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// Paired augmentation on the device (/root/reference/utils/dataset.py:138-175).  The reference
// augments PIL images on DataLoader workers: horizontal flip, rotation by +-5 degrees (TF.rotate: NEAREST, no expand,
// image-mean fill), brightness and contrast (PIL ImageEnhance = Image.blend with black / with the mean-gray image),
// Gaussian noise on the LOW-resolution image only, then ToTensor (uint8 / 255).  Here the uint8 pairs stay resident in
// HBM and two kernels per image do the same arithmetic in the same uint8 stages (truncation after every stage, like
// PIL's (UINT8) casts and numpy's astype): geometry + brightness -> uint8, then contrast + noise + ToTensor -> fp32.
// The random draws come from the host (a few floats per sample); the noise is a counter-based generator per pixel.
__device__ __forceinline__ float pil_blend_u8(float a, float b, float alpha) {   // Image.blend(a, b, alpha) for mode "L" (Blend.c)
    float v = __fadd_rn(a, __fmul_rn(alpha, b - a));     // separate float multiply and add, as the C source compiles
    if (alpha < 0.f || alpha > 1.f) v = fminf(fmaxf(v, 0.f), 255.f);
    return (float)(int)v;
}

__global__ __launch_bounds__(256) void augment_geo_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W,
                                                             const mrisr_aug_geo* __restrict__ params, const double* __restrict__ mean) {
    mrisr_aug_geo q = params[blockIdx.y];
    if (mean) q.fill = (int)mean[blockIdx.y];       // int(mean) of the un-augmented image, kept on the device
    const size_t n = (size_t)H * W;
    const uint8_t* src = in + (size_t)blockIdx.y * n;
    uint8_t* dst = out + (size_t)blockIdx.y * n;
    const float cx = 0.5f * W, cy = 0.5f * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int yo = (int)(i / W), xo = (int)(i - (size_t)yo * W);
        float v;
        if (q.rotate) {
            // PIL Image.rotate -> transform(AFFINE, NEAREST): input coordinate of the output pixel CENTRE, COORD() = floor for
            // non-negative values, outside -> fill
            const float dx = (float)xo + 0.5f - cx, dy = (float)yo + 0.5f - cy;
            const float xin = q.cos_a * dx + q.sin_a * dy + cx, yin = -q.sin_a * dx + q.cos_a * dy + cy;
            const int xi = xin < 0.f ? -1 : (int)xin, yi = yin < 0.f ? -1 : (int)yin;
            if (xi < 0 || yi < 0 || xi >= W || yi >= H) v = (float)q.fill;
            else v = (float)src[(size_t)yi * W + (q.flip ? W - 1 - xi : xi)];       // flip first, then rotate (dataset.py:143-155)
        } else {
            v = (float)src[(size_t)yo * W + (q.flip ? W - 1 - xo : xo)];
        }
        if (q.brightness != 1.f) v = pil_blend_u8(0.f, v, q.brightness);
        dst[i] = (uint8_t)v;
    }
}

__device__ __forceinline__ unsigned mix32(unsigned x) {      // lowbias32 finaliser
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void augment_finish_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t n,
                                                                const mrisr_aug_photo* __restrict__ params, const double* __restrict__ mean) {
    mrisr_aug_photo q = params[blockIdx.y];
    if (mean) q.mean = (int)(mean[blockIdx.y] + 0.5);
    const uint8_t* src = in + (size_t)blockIdx.y * n;
    float* dst = out + (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = (float)src[i];
        if (q.contrast != 1.f) v = pil_blend_u8((float)q.mean, v, q.contrast);
        if (q.noise_sigma > 0.f) {
            // Box-Muller on two hashed 32-bit draws of (seed, pixel): np.clip(a + N(0, sigma), 0, 255).astype(uint8)
            const unsigned h1 = mix32(q.seed ^ mix32((unsigned)i * 2u + 1u)), h2 = mix32(q.seed + 0x9e3779b9u + mix32((unsigned)i * 2u));
            const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777216.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
            const float g = sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
            v = (float)(int)fminf(fmaxf(v + q.noise_sigma * g, 0.f), 255.f);
        }
        dst[i] = __fdiv_rn(v, 255.f);       // ToTensor
    }
}

extern "C" int mrisr_augment_geo_u8(const uint8_t* in, uint8_t* out, int batch, int H, int W, const mrisr_aug_geo* params_device,
                                    const double* mean_device, void* stream) {
    if (!in || !out || !params_device || in == out) MRISR_FAIL(MRISR_E_ARG, "augment_geo_u8: null / aliased pointer");
    if (batch < 1 || batch > 65535 || H < 1 || W < 1) MRISR_FAIL(MRISR_E_SHAPE, "augment_geo_u8: batch %d H %d W %d", batch, H, W);
    dim3 grid(blocks_for((size_t)H * W, 256 * 8, 256), batch);
    augment_geo_u8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, H, W, params_device, mean_device);
    MRISR_CHECK_LAUNCH("augment_geo_u8");
    return MRISR_OK;
}

extern "C" int mrisr_augment_finish_u8(const uint8_t* in, float* out, int batch, size_t pixels_per_image,
                                       const mrisr_aug_photo* params_device, const double* mean_device, void* stream) {
    if (!in || !out || !params_device) MRISR_FAIL(MRISR_E_ARG, "augment_finish_u8: null pointer");
    if (batch < 1 || batch > 65535 || pixels_per_image == 0) MRISR_FAIL(MRISR_E_SHAPE, "augment_finish_u8: batch %d, %zu pixels", batch, pixels_per_image);
    dim3 grid(blocks_for(pixels_per_image, 256 * 8, 256), batch);
    augment_finish_u8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, pixels_per_image, params_device, mean_device);
    MRISR_CHECK_LAUNCH("augment_finish_u8");
    return MRISR_OK;
}
