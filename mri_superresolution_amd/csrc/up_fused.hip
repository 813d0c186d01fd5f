// Decoder up-path in ONE launch (gfx950, 16-bit storage): z = bilinear x2 (align_corners=True) of conv1x1(LeakyReLU(GroupNorm(x)))
// plus the GroupNorm statistics of z.
//
// Replaces nn.Upsample -> nn.Conv2d(1x1) of /root/reference/models/unet_model.py:71-72 (evaluated as conv -> upsample: the two
// linear maps commute, 4x fewer FLOPs), i.e. what mrisr_conv_forward (k = 1, low resolution) + mrisr_upsample2_stats did in two
// launches with the low-resolution tensor going through HBM.  The 1x1 convolution is a small GEMM (4.3 GFLOP per layer at the
// headline shapes) for which the persistent 3x3 kernel pays a whole tick of staging per 8 MFMAs: 37 us per launch at 115 TFLOP/s.
// Here a workgroup owns a 16 x 16 OUTPUT tile x 64 output channels: it computes the 10 x 10 low-resolution patch the tile's
// bilinear footprint touches (GEMM 128 px x 64 co x Cin on MFMA, operands staged through registers with the GroupNorm +
// LeakyReLU transform, double-buffered LDS), keeps the patch in LDS in the storage type (the same rounding point as the
// two-launch form), interpolates from there and writes z with 16-byte stores, accumulating the statistics on the way.
#include <mutex>

#include "conv_common.h"

struct UpParams {
    const void* x;           // [N][h][w][Cin] raw conv output of the producer
    const float* scale;      // [N][Cin] GroupNorm affine of x
    const float* shift;
    const void* wpk;         // classic packed 1x1 image: [cout block of 64][cin chunk of 32][64 rows][64 B], chunks pre-swizzled
    void* z;                 // [N][2h][2w][Cout]
    double* stats;           // [MRISR_STAT_SLOTS][N][groups][2] or NULL
    int N, h, w, Cin, Cout, groups, nchunks, tiles_x, tiles_y;
};

constexpr int kUpThreads = 512;
constexpr int kUpPatch = 10;                       // low-resolution rows / columns under a 16-pixel output span (see below)
constexpr int kUpXBytes = 128 * 64, kUpWBytes = 64 * 64;

template <typename T>
__global__ __launch_bounds__(kUpThreads) void up1x1_fused_kernel(const UpParams p) {
    // two (pixel chunk, weight chunk) buffers; the pixel buffers are re-used for the low-resolution patch [128 px][64 co]
    __shared__ __attribute__((aligned(16))) char smem[2 * (kUpXBytes + kUpWBytes)];
    __shared__ double red[16];
    typedef typename Frag16<T>::type frag_t;
    constexpr int VEC = 8;
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mf = wave & 3, nf = wave >> 2;       // pixel fragment (32 patch pixels), cout fragment (32 channels)
    const int H = 2 * p.h, W = 2 * p.w;
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y, n = tile / p.tiles_y;
    const int cb = blockIdx.y, co0 = cb * 64;
    const int Y0 = ty * 16, X0 = tx * 16;
    // Low-resolution patch origin.  src(Y) = Y (h-1)/(2h-1) < Y/2: over 16 consecutive Y the floor spans at most 9 values and
    // the second tap one more: 10 rows from floor(src(Y0)) cover the tile (checked on the host for every tile).
    int ly0, lx0, tmp;
    float wtmp;
    up2_coord(Y0, p.h, ly0, tmp, wtmp);
    up2_coord(X0, p.w, lx0, tmp, wtmp);

    if (t < 16) red[t] = 0.0;

    // ---- staging roles: pixel vector = 16 B (8 channels) of patch pixel t >> 2, chunk t & 3 of the 64-B row; weight
    // vector = 16 B of the 4-KiB image (threads 0 .. 255)
    const int pp = t >> 2, ch = t & 3;
    const int ppy = pp / kUpPatch, ppx = pp - ppy * kUpPatch;
    const bool pvalid = pp < kUpPatch * kUpPatch;
    const int sy = min(ly0 + ppy, p.h - 1), sx = min(lx0 + ppx, p.w - 1);      // clamped: taps beyond the plane are never used
    const T* xsrc = (const T*)p.x + ((size_t)(n * p.h + sy) * p.w + sx) * p.Cin + ch * VEC;
    const float* scn = p.scale + (size_t)n * p.Cin + ch * VEC;
    const float* shn = p.shift + (size_t)n * p.Cin + ch * VEC;
    const char* wsrc = (const char*)p.wpk + (size_t)cb * p.nchunks * kUpWBytes + (t & 255) * 16;
    // Two register sets, two chunks ahead: the GEMM phase of a chunk is 2 MFMAs per wave, so a workgroup's time is a chain of
    // global-load latencies - with the loads of chunks kc+1 AND kc+2 in flight the chain is half as long.  Indices beyond the
    // last chunk re-load it (no branch around a load, nothing commits it).
    struct Stage {
        Vec16<T> xv;
        u32x4 wv;
        float sc[VEC], sh[VEC];
    };
    Stage S0, S1;
    const int last = p.nchunks - 1;
    auto issue = [&](Stage& S, int kc_) {
        const int kc = min(kc_, last);
        S.xv = gload_vec16(xsrc + kc * 32);
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 a = gload<f32x4>(scn + kc * 32 + e), b = gload<f32x4>(shn + kc * 32 + e);
            S.sc[e] = a[0]; S.sc[e + 1] = a[1]; S.sc[e + 2] = a[2]; S.sc[e + 3] = a[3];
            S.sh[e] = b[0]; S.sh[e + 1] = b[1]; S.sh[e + 2] = b[2]; S.sh[e + 3] = b[3];
        }
        S.wv = gload<u32x4>(wsrc + (size_t)kc * kUpWBytes);      // (threads 256 .. 511 re-read the image's vectors: never stored)
    };
    auto commit = [&](Stage& S, int buf) {
        char* bx = smem + buf * (kUpXBytes + kUpWBytes);
        Vec16<T> v = S.xv, z;
        z.zero();
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float y = fmaf(v.get(e), S.sc[e], S.sh[e]);
            v.set(e, fmaxf(y, LRELU_SLOPE * y));
        }
        if (!pvalid) v = z;
        *reinterpret_cast<decltype(v.v)*>(bx + lds_off(pp, ch)) = v.v;
        if (t < 256) *reinterpret_cast<u32x4*>(bx + kUpXBytes + t * 16) = S.wv;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int xfo = lds_off(32 * mf + lr, lh);                 // B fragment (pixels): k-step 1 = XOR 32
    const int wfo = kUpXBytes + lds_off(32 * nf + lr, lh);     // A fragment (weights)
    auto mfma_chunk = [&](int buf) {
        const char* b = smem + buf * (kUpXBytes + kUpWBytes);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const frag_t wfr = *reinterpret_cast<const frag_t*>(b + (wfo ^ (32 * ks)));
            const frag_t xfr = *reinterpret_cast<const frag_t*>(b + (xfo ^ (32 * ks)));
            acc = Frag16<T>::mma(wfr, xfr, acc);
        }
    };

    issue(S0, 0);
    issue(S1, 1);
    commit(S0, 0);
    issue(S0, 2);
    __syncthreads();
    // chunk kc is multiplied from buffer kc & 1 while chunk kc+1 (set S1 on even kc, S0 on odd kc) goes to the other buffer
    for (int kc = 0; kc < p.nchunks; kc += 2) {
        mfma_chunk(0);
        commit(S1, 1);
        issue(S1, kc + 3);
        __syncthreads();
        if (kc + 1 >= p.nchunks) break;
        mfma_chunk(1);
        commit(S0, 0);
        issue(S0, kc + 4);
        __syncthreads();
    }

    // ---- low-resolution patch -> LDS [128 px][64 co] in the storage type (rows of 128 B): lane = pixel, register quads = 4
    // consecutive channels
    char* zl = smem;                                           // 128 * 128 B = 16 KiB: both pixel buffers... and the first weight buffer
    {
        char* row = zl + (32 * mf + lr) * 128 + (32 * nf + 4 * lh) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef __attribute__((ext_vector_type(4))) T t4_t;
            const t4_t v = {(T)acc[4 * q], (T)acc[4 * q + 1], (T)acc[4 * q + 2], (T)acc[4 * q + 3]};
            *reinterpret_cast<t4_t*>(row + 8 * q * 2) = v;
        }
    }
    __syncthreads();

    // ---- bilinear x2 from the patch, 16-byte stores, statistics.  Vector v = t + 512 i: output pixel v >> 3, channels 8 (v & 7)
    const int c8 = t & 7;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int op = (t >> 3) + 64 * i;
        const int Y = Y0 + (op >> 4), X = X0 + (op & 15);
        if (Y < H && X < W) {
            int y0, y1, x0, x1;
            float wy, wx;
            up2_coord(Y, p.h, y0, y1, wy);
            up2_coord(X, p.w, x0, x1, wx);
            const char* r0 = zl + ((y0 - ly0) * kUpPatch - lx0) * 128 + c8 * 16;
            const char* r1 = zl + ((y1 - ly0) * kUpPatch - lx0) * 128 + c8 * 16;
            Vec16<T> v00, v01, v10, v11, o;
            v00.v = *reinterpret_cast<const decltype(v00.v)*>(r0 + x0 * 128);
            v01.v = *reinterpret_cast<const decltype(v00.v)*>(r0 + x1 * 128);
            v10.v = *reinterpret_cast<const decltype(v00.v)*>(r1 + x0 * 128);
            v11.v = *reinterpret_cast<const decltype(v00.v)*>(r1 + x1 * 128);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {       // aten upsample_bilinear2d's order (as upsample2_stats_kernel)
                const float a = (1.f - wy) * ((1.f - wx) * v00.get(e) + wx * v01.get(e)) + wy * ((1.f - wx) * v10.get(e) + wx * v11.get(e));
                o.set(e, a);
                s += a;
                ss += a * a;
            }
            store_vec16((T*)p.z + ((size_t)(n * H + Y) * W + X) * p.Cout + co0 + c8 * VEC, o);
        }
    }
    if (p.stats) {
        const int gs = p.Cout / p.groups;                      // >= 8: a thread's 8 channels lie in one group
        const int g = (co0 + c8 * VEC) / gs, g0 = co0 / gs;    // groups of this cout block: g0 .. g0 + 64 / gs - 1
        atomicAdd(&red[2 * (g - g0)], (double)s);
        atomicAdd(&red[2 * (g - g0) + 1], (double)ss);
        __syncthreads();
        const int ng = 64 / gs > 0 ? 64 / gs : 1;
        if (t < 2 * ng)
            atomic_add_f64(&p.stats[stat_slot_off_id(blockIdx.x, p.N, p.groups) + ((size_t)n * p.groups + g0) * 2 + t], red[t]);
    }
}

#ifndef MRISR_KERNEL_ONLY
extern "C" int mrisr_up_conv1x1_fused(int dtype, const void* x, const float* scale, const float* shift, const void* wpacked,
                                      void* z, double* stats, int N, int h, int w, int Cin, int Cout, int groups, void* stream) {
    if (!x || !scale || !shift || !wpacked || !z) MRISR_FAIL(MRISR_E_ARG, "up_conv1x1_fused: null pointer");
    if (dtype != MRISR_BF16 && dtype != MRISR_F16) MRISR_FAIL(MRISR_E_DTYPE, "up_conv1x1_fused: 16-bit storage only (dtype %d)", dtype);
    if (N <= 0 || h <= 0 || w <= 0 || Cin % 32 || Cout % 64) MRISR_FAIL(MRISR_E_SHAPE, "up_conv1x1_fused: N%d h%d w%d Cin%d Cout%d", N, h, w, Cin, Cout);
    // a group is 8 .. 64 channels dividing a 64-channel block, or a whole number of blocks
    if (stats && (groups <= 0 || Cout % groups || (Cout / groups) % 8 || (64 % (Cout / groups) && (Cout / groups) % 64)))
        MRISR_FAIL(MRISR_E_SHAPE, "up_conv1x1_fused: groups %d for Cout %d", groups, Cout);
    if ((size_t)N * 4 * h * w * Cout >= (1ull << 31)) MRISR_FAIL(MRISR_E_SHAPE, "up_conv1x1_fused: output exceeds 2^31 elements");
    UpParams p{x, scale, shift, wpacked, z, stats, N, h, w, Cin, Cout, stats ? groups : 0, Cin / 32, ceil_div(2 * w, 16), ceil_div(2 * h, 16)};
    // the 10 x 10 patch must cover every tile's footprint (true for every size; checked rather than assumed)
    for (int axis = 0; axis < 2; ++axis) {
        const int in = axis ? w : h, out = 2 * in;
        const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        for (int o0 = 0; o0 < out; o0 += 16) {
            const int lo = (int)(sc * (float)o0), last = o0 + 15 < out ? o0 + 15 : out - 1;
            int hi = (int)(sc * (float)last);
            hi = hi < in - 1 ? hi + 1 : in - 1;
            if (hi - lo >= kUpPatch) MRISR_FAIL(MRISR_E_UNSUPPORTED, "up_conv1x1_fused: footprint of tile %d exceeds the patch", o0);
        }
    }
    dim3 grid(N * p.tiles_x * p.tiles_y, Cout / 64);
    if (dtype == MRISR_BF16) up1x1_fused_kernel<bf16_t><<<grid, kUpThreads, 0, (hipStream_t)stream>>>(p);
    else up1x1_fused_kernel<f16_t><<<grid, kUpThreads, 0, (hipStream_t)stream>>>(p);
    MRISR_CHECK_LAUNCH("up_conv1x1_fused");
    return MRISR_OK;
}
#endif
