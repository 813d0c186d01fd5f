// Implicit-GEMM 3x3 / 1x1 convolution on MFMA for gfx950 (forward and input-gradient).
//
// Replaces the aten conv2d calls behind nn.Conv2d in /root/reference/models/unet_model.py
// (:29,34,72,101,152,168), with the surrounding GroupNorm-apply + LeakyReLU (:30-31), MaxPool2d
// (:52), bilinear Upsample (:71,151), torch.cat (:93), PixelShuffle (:102) and the alpha blend
// (:206-207) folded into the operand loader / epilogue so none of those tensors is materialised.
//
// Decomposition: one workgroup (4 waves) = 256 output pixels (TH x TW tile of one image) x BN output
// channels.  K loop = cin chunks of 64 bytes; per chunk the transformed (TH+2)x(TW+2) halo tile and the
// 9 x BN x 64 B weight image are staged in LDS once and re-used by all 9 taps (LDS-tiled direct conv on
// MFMA).  D = W(BN x K) * X(K x pixels): the accumulator has a pixel per lane and 4 consecutive output
// channels per register quad, so NHWC stores are 8/16-byte pieces.
#include "conv_common.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 frag;
    // lane half h holds k = 4h..4h+3 of an 8-deep step: four exact-fp32 32x32x2 MFMAs, pairing
    // element j of both halves (any K permutation is valid as long as A and B agree).
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};

template <typename T, int BN, int SPATIAL, int KS>
__global__ __launch_bounds__(kConvThreads, 2) void conv_igemm_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTAPS = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int NF = BN / 32;               // cout fragments per wave
    constexpr int VEC = Vec16<T>::N;
    typedef typename Mma<T>::frag frag_t;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int TW = 1 << p.tw_log2, TH = p.th;
    const int hw = TW + 2 * PAD, hh = TH + 2 * PAD;
    const int npix_halo = hw * hh;
    char* lds_halo = smem;
    char* lds_w = smem + npix_halo * kRowBytes;

    // block -> (image, tile, cout block); cout block fastest so neighbours share the input tile
    int bid = blockIdx.x;
    const int cb = bid % p.ncb; bid /= p.ncb;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int ty0 = ty * TH, tx0 = tx * TW, bn0 = cb * BN;

    HaloGeom<SPATIAL> geom;
#pragma unroll
    for (int i = 0; i < kMaxHaloIter; ++i)
        halo_geom_init<SPATIAL>(geom, i, (t >> 2) + 64 * i, npix_halo, hw, PAD, n, ty0, tx0, p);

    float blend_a = 0.f;
    if (p.combine == MRISR_COMBINE_BLEND) blend_a = 1.f / (1.f + __expf(-p.blend_alpha[0]));

    // per-lane halo row of its two pixels (tap (0,0))
    int hp0[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pl = wave * 64 + mi * 32 + lr;
        hp0[mi] = (pl >> p.tw_log2) * hw + (pl & (TW - 1));
    }

    f32x16 acc[NF][2];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    constexpr int WIMG_VECS = NTAPS * BN * 4;          // 16-B vectors in one weight image
    const char* wbase = (const char*)p.wpacked + (size_t)cb * p.nchunks * (WIMG_VECS * 16);

    for (int kc = 0; kc < p.nchunks; ++kc) {
        // ---- stage: weights image (already in LDS order, swizzled by the packer) + transformed halo
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wbase + (size_t)kc * (WIMG_VECS * 16));
#pragma unroll
        for (int j = 0; j < (WIMG_VECS + kConvThreads - 1) / kConvThreads; ++j) {
            const int v = t + j * kConvThreads;
            if (v < WIMG_VECS) reinterpret_cast<u32x4*>(lds_w)[v] = wsrc[v];
        }
        stage_halo<T, SPATIAL>(lds_halo, geom, kc, n, npix_halo, blend_a, p);
        __syncthreads();

        // ---- compute: 9 taps x 2 k-steps, operands straight from LDS
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int tapoff = (tap / KS) * hw + (tap % KS);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t xf[2], wf[NF];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    xf[mi] = *reinterpret_cast<const frag_t*>(lds_halo + lds_off(hp0[mi] + tapoff, 2 * ks + lh));
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    wf[ni] = *reinterpret_cast<const frag_t*>(lds_w + lds_off(tap * BN + ni * 32 + lr, 2 * ks + lh));
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mma<T>::run(wf[ni], xf[mi], acc[ni][mi]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, (relu), store NHWC / pixel-shuffled, GroupNorm partial statistics
    const int gs = p.groups > 0 ? p.Cout / p.groups : 1;        // channels per group
    const int g_first = bn0 / gs;
    float* lds_stats = reinterpret_cast<float*>(smem);          // [ngl][2]
    const int ngl = p.groups > 0 ? ((min(bn0 + BN, p.Cout) - 1) / gs - g_first + 1) : 0;
    if (p.stats) {
        for (int i = t; i < 2 * ngl; i += kConvThreads) lds_stats[i] = 0.f;
        __syncthreads();
    }
    T* outp = (T*)p.out;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pl = wave * 64 + mi * 32 + lr;
        const int oy = ty0 + (pl >> p.tw_log2), ox = tx0 + (pl & (TW - 1));
        const bool pv = oy < p.H && ox < p.W;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = bn0 + ni * 32 + 8 * q + 4 * lh;      // first of 4 consecutive couts
                float v[4];
                float s = 0.f, ss = 0.f;
                float sj[4], ssj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = acc[ni][mi][4 * q + j];
                    if (p.bias && co + j < p.Cout) a += p.bias[co + j];
                    if (p.relu_out) a = fmaxf(a, 0.f);
                    a = to_f32(from_f32<T>(a));                   // statistics of what is stored
                    v[j] = a;
                    const bool ok = pv && (co + j < p.Cout);
                    sj[j] = ok ? a : 0.f;
                    ssj[j] = ok ? a * a : 0.f;
                    s += sj[j];
                    ss += ssj[j];
                }
                if (pv && co < p.Cout) {
                    if (p.out_mode == MRISR_OUT_PLAIN) {
                        T* dst = outp + ((size_t)(n * p.H + oy) * p.W + ox) * p.Cout + co;
                        if (co + 3 < p.Cout) {
                            if constexpr (sizeof(T) == 2) {
                                bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                                *reinterpret_cast<bf16x4*>(dst) = pk;
                            } else {
                                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                            }
                        } else {
                            for (int j = 0; j < 4 && co + j < p.Cout; ++j) dst[j] = from_f32<T>(v[j]);
                        }
                    } else {   // PixelShuffle(2): channel 4c'+2i+j -> (2y+i, 2x+j, c')
                        const int C4 = p.Cout >> 2, c4 = co >> 2;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            outp[((size_t)(n * 2 * p.H + 2 * oy + (j >> 1)) * (2 * p.W) + 2 * ox + (j & 1)) * C4 + c4] =
                                from_f32<T>(v[j]);
                    }
                }
                if (p.stats) {
                    if ((gs & 3) == 0) {   // the 4 channels share a group
                        s = half_wave_sum(s);
                        ss = half_wave_sum(ss);
                        if (lr == 0 && co < p.Cout) {
                            const int gl = co / gs - g_first;
                            atomicAdd(&lds_stats[2 * gl], s);
                            atomicAdd(&lds_stats[2 * gl + 1], ss);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = half_wave_sum(sj[j]), b = half_wave_sum(ssj[j]);
                            if (lr == 0 && co + j < p.Cout) {
                                const int gl = (co + j) / gs - g_first;
                                atomicAdd(&lds_stats[2 * gl], a);
                                atomicAdd(&lds_stats[2 * gl + 1], b);
                            }
                        }
                    }
                }
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        for (int i = t; i < 2 * ngl; i += kConvThreads)
            atomic_add_f64(&p.stats[((size_t)n * p.groups + g_first) * 2 + i], (double)lds_stats[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// Weight packer: fp32 [Cout][k][k][Cin] -> sequence of LDS images [cout block][cin chunk][tap][BN][64 B]
// (swizzled exactly as the kernel reads them).  transpose_flip: the dgrad operand, i.e. the image of
// W'[ci][2-r][2-s][co] with the roles of Cin and Cout exchanged.
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                    int KS, int flip, int BN, int ncb, int nchunks) {
    constexpr int BK = kRowBytes / (int)sizeof(T);
    const int ntaps = KS * KS;
    const size_t total = (size_t)ncb * nchunks * ntaps * BN * BK;
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;   // logical (output, input) of the image
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int e = r % BK; r /= BK;          // position inside the 64-B row (after swizzle)
        const int row = r % BN; r /= BN;
        const int tap = r % ntaps; r /= ntaps;
        const int kc = r % nchunks;
        const int cb = r / nchunks;
        constexpr int EPC = 16 / (int)sizeof(T);         // elements per 16-B chunk
        const int q = tap * BN + row;
        const int chunk_pos = e / EPC, chunk = chunk_pos ^ ((q >> 2) & 3);
        const int k = kc * BK + chunk * EPC + (e % EPC);  // logical input channel
        const int co = cb * BN + row;
        float v = 0.f;
        if (co < Co && k < Ci) {
            if (!flip) v = w[((size_t)co * ntaps + tap) * Cin + k];
            else v = w[((size_t)k * ntaps + (ntaps - 1 - tap)) * Cin + co];   // W[k][mirrored tap][co]
        }
        out[idx] = from_f32<T>(v);
    }
}

extern "C" size_t mrisr_packed_weight_bytes(int dtype, int Cout, int Cin, int ksize) {
    const int BN = conv_choose_bn(Cout), BK = conv_bk(dtype);
    const size_t ncb = ceil_div(Cout, BN), nch = ceil_div(Cin, BK);
    return ncb * nch * (size_t)(ksize * ksize) * BN * kRowBytes;
}

extern "C" int mrisr_pack_weights(int dtype, const float* w, int Cout, int Cin, int ksize, int transpose_flip,
                                  void* packed, void* stream) {
    if (!w || !packed) MRISR_FAIL(MRISR_E_ARG, "pack_weights: null pointer");
    if (ksize != 1 && ksize != 3) MRISR_FAIL(MRISR_E_UNSUPPORTED, "pack_weights: ksize %d", ksize);
    const int Co = transpose_flip ? Cin : Cout, Ci = transpose_flip ? Cout : Cin;
    const int BN = conv_choose_bn(Co), BK = conv_bk(dtype);
    const int ncb = ceil_div(Co, BN), nch = ceil_div(Ci, BK);
    const size_t total = (size_t)ncb * nch * ksize * ksize * BN * BK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == MRISR_BF16)
        pack_weights_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(w, (bf16_t*)packed, Cout, Cin, ksize,
                                                                            transpose_flip, BN, ncb, nch);
    else if (dtype == MRISR_F32)
        pack_weights_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(w, (float*)packed, Cout, Cin, ksize,
                                                                           transpose_flip, BN, ncb, nch);
    else
        MRISR_FAIL(MRISR_E_DTYPE, "pack_weights: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("pack_weights");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
int conv_fill_params(const mrisr_conv_desc* d, ConvParams& p, const char* who) {
    if (!d) MRISR_FAIL(MRISR_E_ARG, "%s: null descriptor", who);
    if (d->dtype != MRISR_F32 && d->dtype != MRISR_BF16) MRISR_FAIL(MRISR_E_DTYPE, "%s: dtype %d", who, d->dtype);
    if (d->ksize != 1 && d->ksize != 3) MRISR_FAIL(MRISR_E_UNSUPPORTED, "%s: ksize %d", who, d->ksize);
    if (d->nsrc < 1 || d->nsrc > 2) MRISR_FAIL(MRISR_E_ARG, "%s: nsrc %d", who, d->nsrc);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0)
        MRISR_FAIL(MRISR_E_SHAPE, "%s: bad dims N%d H%d W%d Cin%d Cout%d", who, d->N, d->H, d->W, d->Cin, d->Cout);
    const int vec = d->dtype == MRISR_BF16 ? 8 : 4;
    memset(&p, 0, sizeof(p));
    int csum = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        const mrisr_src& a = d->src[s];
        if (!a.ptr) MRISR_FAIL(MRISR_E_ARG, "%s: src%d null", who, s);
        if (a.C % vec) MRISR_FAIL(MRISR_E_SHAPE, "%s: src%d channels %d not a multiple of %d", who, s, a.C, vec);
        if (a.mode == MRISR_SRC_NORM && (!a.scale || !a.shift)) MRISR_FAIL(MRISR_E_ARG, "%s: src%d NORM without scale/shift", who, s);
        if (s > 0 && a.spatial != MRISR_SP_NONE) MRISR_FAIL(MRISR_E_UNSUPPORTED, "%s: spatial transform on src1", who);
        if (d->nsrc > 1 && a.spatial != MRISR_SP_NONE) MRISR_FAIL(MRISR_E_UNSUPPORTED, "%s: spatial transform with 2 sources", who);
        int vh = a.H, vw = a.W;                                  // virtual extent inside the conv input
        if (a.spatial == MRISR_SP_POOL2) { vh = a.H / 2; vw = a.W / 2; }
        if (a.spatial == MRISR_SP_UP2) { vh = 2 * a.H; vw = 2 * a.W; }
        if (a.off_y < 0 || a.off_x < 0 || a.off_y + vh > d->H || a.off_x + vw > d->W)
            MRISR_FAIL(MRISR_E_SHAPE, "%s: src%d extent %dx%d (+%d,%d) exceeds conv input %dx%d", who, s, vh, vw, a.off_y, a.off_x, d->H, d->W);
        if (a.spatial == MRISR_SP_POOL2 && (vh != d->H || vw != d->W || a.off_y || a.off_x))
            MRISR_FAIL(MRISR_E_SHAPE, "%s: pooled src%d %dx%d != conv input %dx%d", who, s, vh, vw, d->H, d->W);
        if ((size_t)d->N * a.H * a.W * a.C >= (1ull << 31)) MRISR_FAIL(MRISR_E_SHAPE, "%s: src%d exceeds 2^31 elements", who, s);
        p.src[s] = SrcDev{a.ptr, a.scale, a.shift, a.C, a.H, a.W, a.mode, a.off_y, a.off_x};
        csum += a.C;
    }
    if (d->combine == MRISR_COMBINE_BLEND) {
        if (d->nsrc != 2 || d->src[0].C != d->src[1].C || !d->blend_alpha) MRISR_FAIL(MRISR_E_ARG, "%s: blend needs 2 equal sources + alpha", who);
        csum = d->src[0].C;
    }
    if (csum != d->Cin) MRISR_FAIL(MRISR_E_SHAPE, "%s: sources carry %d channels, Cin=%d", who, csum, d->Cin);
    const int BK = conv_bk(d->dtype), BN = conv_choose_bn(d->Cout);
    p.blend_alpha = d->blend_alpha; p.wpacked = d->wpacked; p.bias = d->bias; p.out = d->out; p.stats = d->stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.nchunks = ceil_div(d->Cin, BK); p.CinP = p.nchunks * BK;
    p.ncb = ceil_div(d->Cout, BN); p.CoutP = p.ncb * BN;
    p.nsrc = d->nsrc; p.combine = d->combine; p.out_mode = d->out_mode; p.groups = d->stats ? d->groups : 0;
    p.relu_out = d->relu_out;
    conv_choose_tile(d->W, p.th, p.tw_log2);
    p.tiles_x = ceil_div(d->W, 1 << p.tw_log2); p.tiles_y = ceil_div(d->H, p.th);
    return MRISR_OK;
}

template <typename T, int BN, int SPATIAL, int KS>
static int launch_conv(const ConvParams& p, hipStream_t s) {
    const int TW = 1 << p.tw_log2, pad = KS / 2;
    const size_t lds = (size_t)(TW + 2 * pad) * (p.th + 2 * pad) * kRowBytes + (size_t)KS * KS * BN * kRowBytes;
    const int grid = p.N * p.tiles_y * p.tiles_x * p.ncb;
    auto kern = conv_igemm_kernel<T, BN, SPATIAL, KS>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kConvThreads), lds, s, p);
    MRISR_CHECK_LAUNCH("conv_forward");
    return MRISR_OK;
}

template <typename T, int BN>
static int dispatch_conv_sp(const ConvParams& p, int spatial, int ks, hipStream_t s) {
    if (ks == 3) {
        if (spatial == MRISR_SP_NONE) return launch_conv<T, BN, MRISR_SP_NONE, 3>(p, s);
        if (spatial == MRISR_SP_POOL2) return launch_conv<T, BN, MRISR_SP_POOL2, 3>(p, s);
        return launch_conv<T, BN, MRISR_SP_UP2, 3>(p, s);
    }
    if (spatial == MRISR_SP_NONE) return launch_conv<T, BN, MRISR_SP_NONE, 1>(p, s);
    if (spatial == MRISR_SP_UP2) return launch_conv<T, BN, MRISR_SP_UP2, 1>(p, s);
    MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: 1x1 conv with pooled source");
}

extern "C" int mrisr_conv_forward(const mrisr_conv_desc* d, void* stream) {
    ConvParams p;
    int rc = conv_fill_params(d, p, "conv_forward");
    if (rc) return rc;
    if (!d->wpacked || !d->out) MRISR_FAIL(MRISR_E_ARG, "conv_forward: null weights/out");
    if (d->out_mode == MRISR_OUT_PIXEL_SHUFFLE2 && (d->Cout % 4)) MRISR_FAIL(MRISR_E_SHAPE, "conv_forward: pixel shuffle needs Cout%%4==0");
    if (d->stats && (d->groups <= 0 || d->Cout % d->groups)) MRISR_FAIL(MRISR_E_SHAPE, "conv_forward: Cout %d not divisible by groups %d", d->Cout, d->groups);
    const int sp = d->src[0].spatial;
    hipStream_t s = (hipStream_t)stream;
    const int BN = conv_choose_bn(d->Cout);
    if (d->dtype == MRISR_BF16) return BN == 64 ? dispatch_conv_sp<bf16_t, 64>(p, sp, d->ksize, s) : dispatch_conv_sp<bf16_t, 32>(p, sp, d->ksize, s);
    return BN == 64 ? dispatch_conv_sp<float, 64>(p, sp, d->ksize, s) : dispatch_conv_sp<float, 32>(p, sp, d->ksize, s);
}
