// Weight gradient of the 3x3 / 1x1 convolution on MFMA (gfx950).
//
// Replaces aten convolution_backward's weight branch for the nn.Conv2d layers of
// /root/reference/models/unet_model.py (58 % of the reference's CPU step).
//   dW[co][tap][ci] += sum_pixels dy[pixel][co] * in[pixel + tap][ci]
// GEMM view per tap: M = co, N = ci, K = pixels.  Both operands are pixel-major (NHWC), i.e.
// K-strided, so fragments come from LDS through the transposing read ds_read_b64_tr_b16 (bf16) or
// single-dword reads (exact-fp32 32x32x2 MFMA).  The conv input is re-created by the same fused
// loader as the forward pass (GroupNorm+LeakyReLU apply / pool / bilinear / concat / blend).
//
// Workgroup = (co block, ci block) of 128-byte channel rows, looping over a slice of the pixel
// tiles (split-K over the grid); partial sums are added to the fp32 master gradient with
// 128-byte-segment float atomics (ci contiguous).
#include "conv_common.h"

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> { static constexpr int BC = 64; };   // channels per 128-B row
template <> struct WgTraits<float> { static constexpr int BC = 32; };

__device__ __forceinline__ bf16x8 tr_read_frag(const char* base0, const char* base1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base1);
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

template <typename T, int SPATIAL, int KS>
__global__ __launch_bounds__(kConvThreads, 2) void conv_wgrad_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTAPS = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int BC = WgTraits<T>::BC;
    constexpr int VEC = Vec16<T>::N;
    constexpr bool kBf16 = sizeof(T) == 2;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int TW = 1 << p.tw_log2, TH = p.th;
    const int hw = TW + 2 * PAD, hh = TH + 2 * PAD;
    const int npix_halo = hw * hh;
    char* lds_dy = smem;                       // [256 px][128 B]
    char* lds_in = smem + 256 * 128;           // [halo px][128 B]

    const int ncib = (p.Cin + BC - 1) / BC;
    const int cib = blockIdx.x % ncib, cob = blockIdx.x / ncib;
    const int co0 = cob * BC, ci0 = cib * BC;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x;

    float blend_a = 0.f;
    if (p.combine == MRISR_COMBINE_BLEND) blend_a = 1.f / (1.f + __expf(-p.blend_alpha[0]));

    f32x16 acc[NTAPS];
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    // bf16: wave -> (co fragment, ci fragment); fp32: one fragment pair, waves split the pixels
    const int fo = kBf16 ? (wave >> 1) : 0, fi = kBf16 ? (wave & 1) : 0;

    for (int tile = blockIdx.y; tile < total_tiles; tile += gridDim.y) {
        int b = tile;
        const int tx = b % p.tiles_x; b /= p.tiles_x;
        const int ty = b % p.tiles_y;
        const int n = b / p.tiles_y;
        const int ty0 = ty * TH, tx0 = tx * TW;

        // ---- stage dy tile: thread -> 16-B chunk (t&7) of pixels (t>>3)+32i
        {
            const int ch = t & 7;
            const int c = co0 + ch * VEC;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pl = (t >> 3) + 32 * i;
                const int oy = ty0 + (pl >> p.tw_log2), ox = tx0 + (pl & (TW - 1));
                Vec16<T> v;
                v.zero();
                if (oy < p.H && ox < p.W && c < p.Cout)
                    v = load_vec16((const T*)p.dy + ((size_t)(n * p.H + oy) * p.W + ox) * p.Cout + c);
                *reinterpret_cast<decltype(v.v)*>(lds_dy + lds_off128(pl, ch >> 2, ch & 3)) = v.v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the staging phases apart: 144 accumulator VGPRs are live
        // ---- stage the transformed input halo: two 64-B cin sub-chunks
        {
            HaloGeom<SPATIAL> geom;
#pragma unroll
            for (int i = 0; i < kMaxHaloIter; ++i)
                halo_geom_init<SPATIAL>(geom, i, (t >> 2) + 64 * i, npix_halo, hw, PAD, n, ty0, tx0, p);
            stage_halo<T, SPATIAL, 128, 2>(lds_in, geom, 2 * cib, n, npix_halo, blend_a, p, 0);
            __builtin_amdgcn_sched_barrier(0);
            stage_halo<T, SPATIAL, 128, 2>(lds_in, geom, 2 * cib + 1, n, npix_halo, blend_a, p, 1);
        }
        __syncthreads();

        if constexpr (kBf16) {
            const int li = lane & 15, gq = li >> 2, gp = li & 3, gr = (lane >> 4) & 1;
            // 16 pixels per k-step; this lane addresses pixel k0 + 8*lh + 4*t + gq, channels 16*gr+4*gp
            const int chb = 16 * gr + 4 * gp;            // channel inside the 32-wide fragment
#pragma unroll 1
            for (int ks = 0; ks < 16; ++ks) {
                const int pk = ks * 16 + 8 * lh + gq;    // pixel (t = 0); t = 1 adds 4
                const int c_dy = fo * 32 + chb;          // channel within the 64-wide row
                const bf16x8 af = tr_read_frag(
                    lds_dy + lds_off128(pk, c_dy >> 5, (c_dy >> 3) & 3) + ((c_dy & 4) << 1),
                    lds_dy + lds_off128(pk + 4, c_dy >> 5, (c_dy >> 3) & 3) + ((c_dy & 4) << 1));
                const int hp = (pk >> p.tw_log2) * hw + (pk & (TW - 1));   // pk and pk+4 share the row
                const int c_in = fi * 32 + chb;
#pragma unroll
                for (int tap = 0; tap < NTAPS; ++tap) {
                    const int h0 = hp + (tap / KS) * hw + (tap % KS);
                    const bf16x8 bfr = tr_read_frag(
                        lds_in + lds_off128(h0, c_in >> 5, (c_in >> 3) & 3) + ((c_in & 4) << 1),
                        lds_in + lds_off128(h0 + 4, c_in >> 5, (c_in >> 3) & 3) + ((c_in & 4) << 1));
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tap], 0, 0, 0);
                }
            }
        } else {
            // exact fp32: one 32x32x2 MFMA per pixel pair; wave w takes pairs w, w+4, ...
            for (int kp = wave; kp < 128; kp += 4) {
                const int pk = 2 * kp + lh;
                const float a = *reinterpret_cast<const float*>(lds_dy + lds_off128(pk, lr >> 4, (lr >> 2) & 3) + ((lr & 3) << 2));
                const int hp = (pk >> p.tw_log2) * hw + (pk & (TW - 1));
#pragma unroll
                for (int tap = 0; tap < NTAPS; ++tap) {
                    const int h0 = hp + (tap / KS) * hw + (tap % KS);
                    const float bb = *reinterpret_cast<const float*>(lds_in + lds_off128(h0, lr >> 4, (lr >> 2) & 3) + ((lr & 3) << 2));
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[tap], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- accumulate into dW[co][tap][ci]: lane = ci (128-B contiguous per half wave), regs = co
    const int ci = ci0 + fi * 32 + lr;
    if (ci < p.Cin) {
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + fo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) atomic_add_f32(p.dw + ((size_t)co * NTAPS + tap) * p.Cin + ci, acc[tap][r]);
            }
    }
}

int conv_fill_params(const mrisr_conv_desc* d, ConvParams& p, const char* who);

template <typename T, int SPATIAL, int KS>
static int launch_wgrad(ConvParams& p, hipStream_t s) {
    constexpr int BC = WgTraits<T>::BC;
    const int TW = 1 << p.tw_log2, pad = KS / 2;
    const size_t lds = 256 * 128 + (size_t)(TW + 2 * pad) * (p.th + 2 * pad) * 128;
    const int nblk = ceil_div(p.Cout, BC) * ceil_div(p.Cin, BC);
    const int total_tiles = p.N * p.tiles_y * p.tiles_x;
    int ksplit = ceil_div(512, nblk);
    if (ksplit > total_tiles) ksplit = total_tiles;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 65535) ksplit = 65535;
    auto kern = conv_wgrad_kernel<T, SPATIAL, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, ksplit), dim3(kConvThreads), lds, s, p);
    MRISR_CHECK_LAUNCH("conv_wgrad");
    return MRISR_OK;
}

template <typename T>
static int dispatch_wgrad(ConvParams& p, int spatial, int ks, hipStream_t s) {
    if (ks == 3) {
        if (spatial == MRISR_SP_NONE) return launch_wgrad<T, MRISR_SP_NONE, 3>(p, s);
        if (spatial == MRISR_SP_POOL2) return launch_wgrad<T, MRISR_SP_POOL2, 3>(p, s);
        return launch_wgrad<T, MRISR_SP_UP2, 3>(p, s);
    }
    if (spatial == MRISR_SP_NONE) return launch_wgrad<T, MRISR_SP_NONE, 1>(p, s);
    if (spatial == MRISR_SP_UP2) return launch_wgrad<T, MRISR_SP_UP2, 1>(p, s);
    MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_wgrad: 1x1 conv with pooled source");
}

extern "C" int mrisr_conv_wgrad(const mrisr_conv_desc* d, const void* dy, float* dw, void* stream) {
    ConvParams p;
    int rc = conv_fill_params(d, p, "conv_wgrad");
    if (rc) return rc;
    if (!dy || !dw) MRISR_FAIL(MRISR_E_ARG, "conv_wgrad: null dy/dw");
    const int vec = d->dtype == MRISR_BF16 ? 8 : 4;
    if (d->Cout % vec) MRISR_FAIL(MRISR_E_SHAPE, "conv_wgrad: Cout %d not a multiple of %d", d->Cout, vec);
    p.dy = dy;
    p.dw = dw;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == MRISR_BF16) return dispatch_wgrad<bf16_t>(p, d->src[0].spatial, d->ksize, s);
    return dispatch_wgrad<float>(p, d->src[0].spatial, d->ksize, s);
}
