// Fused L1 + Gaussian-window SSIM for single-channel fp32 images (gfx950, HBM-bound).
//
// Replaces the 5 depthwise F.conv2d + ~15 element-wise aten ops of /root/reference/utils/losses.py:55-73
// and nn.L1Loss (:177,206) with ONE pass over the two images: an LDS tile with a 5-pixel halo, the
// separable 11+11 Gaussian (zero padding, truncated un-renormalised border windows as the reference),
// wave-shuffle + atomic reduction.  The backward pass is analytic: with S = A1*A2/(B1*B2),
//   dS/dmu1 = 2 mu2 (A2-A1)/(B1 B2) - 2 mu1 S/B1 + 2 mu1 S/B2,  dS/dE[x^2] = -S/B2,  dS/dE[xy] = 2 A1/(B1 B2)
// and d(sum S)/dx = G(dS/dmu1) + 2x G(dS/dE[x^2]) + y G(dS/dE[xy])   (G = the same zero-padded blur).
#include "common.h"

constexpr int kMaxWin = 15;                        // odd window sizes 3 .. 15 (the reference's default and only caller value: 11)
constexpr int kTW = 32, kTH = 8;                   // output tile of the backward kernel

struct GaussWin { float g[kMaxWin]; };

static GaussWin make_window(float sigma, int win) {   // losses.py:10-18 in fp32
    GaussWin w;
    float sum = 0.f;
    for (int i = 0; i < kMaxWin; ++i) w.g[i] = 0.f;
    for (int i = 0; i < win; ++i) {
        const float c = (float)(i - win / 2);
        w.g[i] = expf(-(c * c) / (2.0f * sigma * sigma));
        sum += w.g[i];
    }
    for (int i = 0; i < win; ++i) w.g[i] /= sum;
    return w;
}

// 32 x 32 output tile per block, 4 outputs per thread in both passes (register blocking): the 11-tap windows of
// 4 neighbouring outputs share 14 inputs, so the LDS reads per output drop from 104 (one output per thread, 32 x 8
// tile) to 27 and the halo overhead from 2.95x to 1.72x.
constexpr int kFT = 32;                              // forward tile edge
template <int kWin>
__global__ __launch_bounds__(256) void ssim_l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          double* __restrict__ sums, float* __restrict__ coef, int N,
                                                          int H, int W, float c1, float c2, const GaussWin win) {
    constexpr int kHalo = kWin / 2, kFL = kFT + 2 * kHalo;     // 42 for the 11-tap window
    __shared__ float ta[kFL][kFL + 2], tb[kFL][kFL + 2];
    __shared__ float hz[5][kFL][kFT + 1];            // horizontally blurred x, y, xx, yy, xy
    __shared__ float part[4][2];
    const int t = threadIdx.x, n = blockIdx.z;
    const int x0 = blockIdx.x * kFT, y0 = blockIdx.y * kFT;
    const float* an = a + (size_t)n * H * W;
    const float* bn = b + (size_t)n * H * W;
    for (int i = t; i < kFL * kFL; i += 256) {
        const int ly = i / kFL, lx = i - ly * kFL;
        const int gy = y0 + ly - kHalo, gx = x0 + lx - kHalo;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        ta[ly][lx] = in ? an[(size_t)gy * W + gx] : 0.f;
        tb[ly][lx] = in ? bn[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    // horizontal pass: segment = 4 consecutive outputs of one staged row (42 rows x 8 segments)
    for (int sgm = t; sgm < kFL * (kFT / 4); sgm += 256) {
        const int ly = sgm / (kFT / 4), lx = (sgm - ly * (kFT / 4)) * 4;
        float u[kWin + 3], v[kWin + 3];
#pragma unroll
        for (int k = 0; k < kWin + 3; ++k) { u[k] = ta[ly][lx + k]; v[k] = tb[ly][lx + k]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const float uu = u[o + k], vv = v[o + k], g = win.g[k];
                s0 += g * uu; s1 += g * vv; s2 += g * uu * uu; s3 += g * vv * vv; s4 += g * uu * vv;
            }
            hz[0][ly][lx + o] = s0; hz[1][ly][lx + o] = s1; hz[2][ly][lx + o] = s2; hz[3][ly][lx + o] = s3; hz[4][ly][lx + o] = s4;
        }
    }
    __syncthreads();
    // vertical pass: thread = column lx, 4 consecutive output rows ly0..ly0+3
    const int lx = t & (kFT - 1), ly0 = (t >> 5) * 4;
    float m[4][5];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int q = 0; q < 5; ++q) m[o][q] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        float col[kWin + 3];
#pragma unroll
        for (int k = 0; k < kWin + 3; ++k) col[k] = hz[q][ly0 + k][lx];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int k = 0; k < kWin; ++k) m[o][q] += win.g[k] * col[o + k];
    }
    float l1 = 0.f, sv = 0.f;
    const int gx = x0 + lx;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + ly0 + o;
        if (gy < H && gx < W) {
            const float mu1 = m[o][0], mu2 = m[o][1];
            const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s11 = m[o][2] - mu1sq, s22 = m[o][3] - mu2sq, s12 = m[o][4] - mu12;
            const float A1 = 2.f * mu12 + c1, A2 = 2.f * s12 + c2, B1 = mu1sq + mu2sq + c1, B2 = s11 + s22 + c2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            sv += S;
            l1 += fabsf(ta[ly0 + o + kHalo][lx + kHalo] - tb[ly0 + o + kHalo][lx + kHalo]);
            if (coef) {
                const size_t plane = (size_t)N * H * W, oo = ((size_t)n * H + gy) * W + gx;
                coef[oo] = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * S / B1 + 2.f * mu1 * S / B2;
                coef[plane + oo] = -S / B2;
                coef[2 * plane + oo] = 2.f * A1 * inv;
            }
        }
    }
    l1 = wave_sum(l1);
    sv = wave_sum(sv);
    if ((t & 63) == 0) { part[t >> 6][0] = l1; part[t >> 6][1] = sv; }
    __syncthreads();
    if (t < 2) atomic_add_f64(&sums[(size_t)n * 2 + t], (double)(part[0][t] + part[1][t] + part[2][t] + part[3][t]));
}

template <int WIN>
static void launch_ssim_fwd(dim3 grid, hipStream_t s, const float* a, const float* b, double* sums, float* coef, int N, int H, int W,
                            float c1, float c2, float sigma) {
    ssim_l1_fwd_kernel<WIN><<<grid, 256, 0, s>>>(a, b, sums, coef, N, H, W, c1, c2, make_window(sigma, WIN));
}
#define MRISR_SSIM_WIN_SWITCH(win, CALL) \
    switch (win) { case 3: CALL(3); break; case 5: CALL(5); break; case 7: CALL(7); break; case 9: CALL(9); break; \
                   case 11: CALL(11); break; case 13: CALL(13); break; default: CALL(15); break; }

// window_size: odd, 3 .. 15 (losses.py:27's parameter; even sizes change the reference's output size and are refused)
extern "C" int mrisr_ssim_l1_forward_win(const float* a, const float* b, double* sums, float* coef, int N, int H, int W,
                                         float val_range, float sigma, int window_size, void* stream) {
    if (!a || !b || !sums) MRISR_FAIL(MRISR_E_ARG, "ssim_l1_forward: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || N > 65535) MRISR_FAIL(MRISR_E_SHAPE, "ssim_l1_forward: N%d H%d W%d", N, H, W);
    if (window_size < 3 || window_size > kMaxWin || !(window_size & 1)) MRISR_FAIL(MRISR_E_UNSUPPORTED, "ssim_l1_forward: window_size %d (odd, 3..15)", window_size);
    const float c1 = (0.01f * val_range) * (0.01f * val_range), c2 = (0.03f * val_range) * (0.03f * val_range);
    dim3 grid(ceil_div(W, kFT), ceil_div(H, kFT), N);
    hipStream_t s = (hipStream_t)stream;
#define MRISR_CALL(WIN) launch_ssim_fwd<WIN>(grid, s, a, b, sums, coef, N, H, W, c1, c2, sigma)
    MRISR_SSIM_WIN_SWITCH(window_size, MRISR_CALL)
#undef MRISR_CALL
    MRISR_CHECK_LAUNCH("ssim_l1_forward");
    return MRISR_OK;
}
extern "C" int mrisr_ssim_l1_forward(const float* a, const float* b, double* sums, float* coef, int N, int H, int W,
                                     float val_range, float sigma, void* stream) {
    return mrisr_ssim_l1_forward_win(a, b, sums, coef, N, H, W, val_range, sigma, 11, stream);
}

template <int kWin>
__global__ __launch_bounds__(256) void ssim_l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ coef, const double* __restrict__ sums,
                                                          const float* __restrict__ gscale, float l1_w, float ssim_w,
                                                          float* __restrict__ da, int N, int H, int W, const GaussWin win) {
    constexpr int kHalo = kWin / 2, kLW = kTW + 2 * kHalo, kLH = kTH + 2 * kHalo;     // 42 x 18 for the 11-tap window
    __shared__ float tc[3][kLH][kLW + 1];
    __shared__ float hz[3][kLH][kTW + 1];
    const int t = threadIdx.x, n = blockIdx.z;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const size_t plane = (size_t)N * H * W;
    const float gs = gscale ? gscale[0] : 1.f;
    const double numel = (double)N * H * W;
    float sw = 0.f;
    if (ssim_w != 0.f) {
        if (sums) {                           // clamp(ssim, 0, 1) passes gradient only inside [0, 1] (losses.py:221)
            double tot = 0.0;
            for (int i = 0; i < N; ++i) tot += sums[(size_t)i * 2 + 1];
            const double mean = tot / numel;
            if (mean >= 0.0 && mean <= 1.0) sw = ssim_w;
        } else {
            sw = ssim_w;                      // plain ssim(): no clamp
        }
    }
    if (sw != 0.f) {
        for (int i = t; i < kLH * kLW; i += 256) {
            const int ly = i / kLW, lx = i - ly * kLW;
            const int gy = y0 + ly - kHalo, gx = x0 + lx - kHalo;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = ((size_t)n * H + gy) * W + gx;
#pragma unroll
            for (int q = 0; q < 3; ++q) tc[q][ly][lx] = in ? coef[q * plane + o] : 0.f;
        }
        __syncthreads();
        for (int i = t; i < kLH * kTW; i += 256) {
            const int ly = i / kTW, lx = i - ly * kTW;
            float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kWin; ++k)
#pragma unroll
                for (int q = 0; q < 3; ++q) s[q] += win.g[k] * tc[q][ly][lx + k];
#pragma unroll
            for (int q = 0; q < 3; ++q) hz[q][ly][lx] = s[q];
        }
        __syncthreads();
    }
    const int ly = t / kTW, lx = t - ly * kTW;
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy < H && gx < W) {
        const size_t o = ((size_t)n * H + gy) * W + gx;
        const float av = a[o], bv = b[o];
        float g = 0.f;
        if (l1_w != 0.f) {
            const float d = av - bv;
            g += l1_w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (sw != 0.f) {
            float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kWin; ++k)
#pragma unroll
                for (int q = 0; q < 3; ++q) m[q] += win.g[k] * hz[q][ly + k][lx];
            g -= sw * (m[0] + 2.f * av * m[1] + bv * m[2]);
        }
        da[o] = gs * g * (float)(1.0 / numel);
    }
}

template <int WIN>
static void launch_ssim_bwd(dim3 grid, hipStream_t s, const float* a, const float* b, const float* coef, const double* sums,
                            const float* gscale, float l1_w, float ssim_w, float* da, int N, int H, int W, float sigma) {
    ssim_l1_bwd_kernel<WIN><<<grid, 256, 0, s>>>(a, b, coef, sums, gscale, l1_w, ssim_w, da, N, H, W, make_window(sigma, WIN));
}
// Gradient w.r.t. the FIRST image; SSIM and L1 are symmetric, so the gradient w.r.t. the second one is this call with the
// images swapped and `coef` from a forward call with the images swapped.
extern "C" int mrisr_ssim_l1_backward_win(const float* a, const float* b, const float* coef, const double* sums,
                                          const float* gscale, float l1_w, float ssim_w, float* da, int N, int H, int W,
                                          float sigma, int window_size, void* stream) {
    if (!a || !b || !da) MRISR_FAIL(MRISR_E_ARG, "ssim_l1_backward: null pointer");
    if (ssim_w != 0.f && !coef) MRISR_FAIL(MRISR_E_ARG, "ssim_l1_backward: ssim term needs coef");
    if (N <= 0 || H <= 0 || W <= 0 || N > 65535) MRISR_FAIL(MRISR_E_SHAPE, "ssim_l1_backward: N%d H%d W%d", N, H, W);
    if (window_size < 3 || window_size > kMaxWin || !(window_size & 1)) MRISR_FAIL(MRISR_E_UNSUPPORTED, "ssim_l1_backward: window_size %d (odd, 3..15)", window_size);
    dim3 grid(ceil_div(W, kTW), ceil_div(H, kTH), N);
    hipStream_t s = (hipStream_t)stream;
#define MRISR_CALL(WIN) launch_ssim_bwd<WIN>(grid, s, a, b, coef, sums, gscale, l1_w, ssim_w, da, N, H, W, sigma)
    MRISR_SSIM_WIN_SWITCH(window_size, MRISR_CALL)
#undef MRISR_CALL
    MRISR_CHECK_LAUNCH("ssim_l1_backward");
    return MRISR_OK;
}
extern "C" int mrisr_ssim_l1_backward(const float* a, const float* b, const float* coef, const double* sums,
                                      const float* gscale, float l1_w, float ssim_w, float* da, int N, int H, int W,
                                      float sigma, void* stream) {
    return mrisr_ssim_l1_backward_win(a, b, coef, sums, gscale, l1_w, ssim_w, da, N, H, W, sigma, 11, stream);
}

// ------------------------------------------------------------------------------------------------
// CombinedLoss scalar (losses.py:200-240 without the perceptual term): out[0] = total,
// out[1] = L1 mean, out[2] = SSIM mean (unclamped), out[3..3+N) = per-sample SSIM.
__global__ void loss_finalize_kernel(const double* __restrict__ sums, int N, double inv_numel, double inv_per_sample,
                                     float l1_w, float ssim_w, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double l1 = 0.0, ss = 0.0;
    for (int i = 0; i < N; ++i) {
        l1 += sums[2 * i];
        ss += sums[2 * i + 1];
        out[3 + i] = (float)(sums[2 * i + 1] * inv_per_sample);
    }
    const float l1m = (float)(l1 * inv_numel), sm = (float)(ss * inv_numel);
    float total = 0.f;
    if (l1_w > 0.f) total += l1_w * l1m;
    if (ssim_w > 0.f) total += ssim_w * (1.f - fminf(fmaxf(sm, 0.f), 1.f));
    out[0] = total;
    out[1] = l1m;
    out[2] = sm;
}

extern "C" int mrisr_loss_finalize(const double* sums, int N, int H, int W, float l1_w, float ssim_w, float* out,
                                   void* stream) {
    if (!sums || !out) MRISR_FAIL(MRISR_E_ARG, "loss_finalize: null pointer");
    loss_finalize_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sums, N, 1.0 / ((double)N * H * W), 1.0 / ((double)H * W), l1_w, ssim_w, out);
    MRISR_CHECK_LAUNCH("loss_finalize");
    return MRISR_OK;
}
