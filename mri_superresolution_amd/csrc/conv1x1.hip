// 1x1 convolution as a plain MFMA GEMM (gfx950): out[pixel][co] = sum_ci act(x[pixel][ci]) * W[co][ci], 16-bit storage.
//
// Replaces the nn.Conv2d(kernel_size=1) of Up (/root/reference/models/unet_model.py:72, evaluated at LOW resolution in front of
// the bilinear x2: the two linear maps commute) and autograd's input gradient of it.  conv_igemm_kernel<..., KS = 1> treats a
// 1x1 layer as a 3x3 layer with one tap: 256-pixel items of ONE 64-byte channel chunk, a barrier pair and a weight image per
// chunk - 31-45 us for 4.3 GFLOP (95-140 TFLOP/s).  Here a workgroup owns 128 pixels x 64 / 128 / 256 output channels for the
// whole reduction: per 32-channel chunk it stages 8 KB of pixels (GroupNorm + LeakyReLU applied on the way) and 4-16 KB of
// the packed weight image, register-prefetched one chunk ahead, one barrier per chunk.
#include <type_traits>

#include "conv_common.h"

constexpr int kC1Threads = 256;
constexpr int kC1BM = 128;                        // pixels per workgroup
constexpr int kC1ABytes = kC1BM * kRowBytes;      // 8 KB: [pixel][64 B], chunk position XOR-swizzled (lds_off)
constexpr int kC1WPiece = 64 * kRowBytes;         // 4 KB: one 64-row block of the packed weight image (conv_fwd.hip layout)

// NB = 64-channel weight blocks per workgroup (1, 2, 4); the four waves split 2 (pixel halves) x 2 (channel halves)
template <typename T, bool NORM, int NB>
__global__ __launch_bounds__(kC1Threads) void conv1x1_gemm_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Frag16<T>::type frag_t;
    constexpr int VEC = 8, NFW = NB;                              // 32-channel fragments per wave
    constexpr int STAGE = kC1ABytes + NB * kC1WPiece;
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int ncbw = p.Cout / (64 * NB);
    const int tile = blockIdx.x / ncbw, cbw = blockIdx.x - tile * ncbw;
    const int HW = p.H * p.W;
    const size_t pix0 = (size_t)tile * kC1BM;
    const int n = (int)(pix0 / HW);                               // (host-checked: a tile lies in one image)
    float* aff = reinterpret_cast<float*>(smem + 2 * STAGE);      // [2][Cin]: scale, shift of image n
    if (NORM) {
        for (int c = t; c < p.Cin; c += kC1Threads) {
            aff[c] = p.src[0].scale[(size_t)n * p.Cin + c];
            aff[p.Cin + c] = p.src[0].shift[(size_t)n * p.Cin + c];
        }
    }
    // staging duty: pixel (t >> 2) + 64 j, 16-byte chunk t & 3 of the 64-byte row; weight vector t of each 4-KB piece
    const int sp = t >> 2, sc4 = t & 3;
    const T* xg = (const T*)p.src[0].ptr + (pix0 + sp) * p.Cin + sc4 * VEC;
    const char* wg = (const char*)p.wpacked + (size_t)(cbw * NB) * p.nchunks * kC1WPiece + t * 16;
    Vec16<T> ra[2];
    u32x4 rw[NB];
    auto issue = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 2; ++j) ra[j] = gload_vec16(xg + (size_t)(64 * j) * p.Cin + kc * 32);
#pragma unroll
        for (int b = 0; b < NB; ++b) rw[b] = gload<u32x4>(wg + ((size_t)b * p.nchunks + kc) * kC1WPiece);
    };
    auto commit = [&](int kc, char* st) {
        if (NORM) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(aff + kc * 32 + sc4 * VEC), s1 = *reinterpret_cast<const f32x4*>(aff + kc * 32 + sc4 * VEC + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(aff + p.Cin + kc * 32 + sc4 * VEC), h1 = *reinterpret_cast<const f32x4*>(aff + p.Cin + kc * 32 + sc4 * VEC + 4);
            const float sc[VEC] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            const float sh[VEC] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float y = fmaf(ra[j].get(e), sc[e], sh[e]);
                    ra[j].set(e, fmaxf(y, LRELU_SLOPE * y));
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<decltype(ra[j].v)*>(st + lds_off(sp + 64 * j, sc4)) = ra[j].v;
#pragma unroll
        for (int b = 0; b < NB; ++b) *reinterpret_cast<u32x4*>(st + kC1ABytes + b * kC1WPiece + t * 16) = rw[b];
    };

    f32x16 acc[NFW][2];
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    const int fo = lds_off(lr, lh);                                 // k-step 1 = this XOR 32
    const int a_base = (wm * 2) * 32 * kRowBytes + fo;
    const int w_base = kC1ABytes + (wn * NFW) * 32 * kRowBytes + fo;

    issue(0);
    __syncthreads();                                                // the affine table is visible
#pragma unroll 1
    for (int kc = 0; kc < p.nchunks; ++kc) {
        char* st = smem + (kc & 1) * STAGE;
        commit(kc, st);
        if (kc + 1 < p.nchunks) issue(kc + 1);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t af[2], wf[NFW];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const frag_t*>(st + ((a_base + mi * 32 * kRowBytes) ^ (ks * 32)));
#pragma unroll
            for (int ni = 0; ni < NFW; ++ni) wf[ni] = *reinterpret_cast<const frag_t*>(st + ((w_base + ni * 32 * kRowBytes) ^ (ks * 32)));
#pragma unroll
            for (int ni = 0; ni < NFW; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Frag16<T>::mma(wf[ni], af[mi], acc[ni][mi]);
        }
    }
    // ---- epilogue (as conv_ring.hip): packed 16-bit values exchanged between the lane halves -> 16-byte NHWC stores
    char* obase = (char*)p.out + ((pix0 + (size_t)(wm * 2) * 32 + lr) * p.Cout + cbw * 64 * NB + wn * NFW * 32) * sizeof(T) + 16 * lh;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NFW; ++ni) {
            u32x2 packed[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef __attribute__((ext_vector_type(4))) T t4_t;
                union { t4_t b; u32x2 u; } cv;
                cv.b = t4_t{(T)acc[ni][mi][4 * q], (T)acc[ni][mi][4 * q + 1], (T)acc[ni][mi][4 * q + 2], (T)acc[ni][mi][4 * q + 3]};
                packed[q] = cv.u;
            }
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const u32x2 a = packed[q], b = packed[q + 1];
                auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                gstore(obase + (size_t)mi * 32 * p.Cout * sizeof(T) + (ni * 32 + 8 * q) * 2, o);
            }
        }
}

// ------------------------------------------------------------------------------------------------ host side
#ifndef MRISR_KERNEL_ONLY
bool conv1x1_gemm_eligible(const mrisr_conv_desc* d, const ConvParams& p) {
#ifdef MRISR_NO_C1X1
    return false;
#endif
    if (d->dtype == MRISR_F32 || d->ksize != 1 || !d->wpacked) return false;
    if (d->Cin % 32 || d->Cout % 64 || d->Cin > 2048) return false;
    if (d->out_mode != MRISR_OUT_PLAIN || d->relu_mask || d->relu_out || d->bias || d->stats) return false;
    if (d->nsrc != 1 || d->combine != MRISR_COMBINE_CONCAT || d->src[0].spatial != MRISR_SP_NONE) return false;
    if (d->src[0].mode != MRISR_SRC_NORM && d->src[0].mode != MRISR_SRC_RAW) return false;
    if (d->src[0].H != d->H || d->src[0].W != d->W || d->src[0].off_y || d->src[0].off_x || d->src[0].C != d->Cin) return false;
    const long HW = (long)d->H * d->W;
    if (HW % kC1BM) return false;                        // whole 128-pixel tiles, each inside one image
    return p.nchunks == d->Cin / 32;
}

template <typename T, bool NORM, int NB>
static void launch_c1_t(const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_gemm_kernel<T, NORM, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv1x1_gemm_kernel<T, NORM, NB>), dim3(grid), dim3(kC1Threads), lds, s, p);
}

template <typename T>
static int launch_c1(const mrisr_conv_desc* d, const ConvParams& p, hipStream_t s) {
    const int tiles = (int)((long)d->N * d->H * d->W / kC1BM);
    // the widest channel block that still fills the chip twice over (each workgroup re-reads its pixels once per block)
    int nb = 1;
    for (int c : {4, 2})
        if (d->Cout % (64 * c) == 0 && (long)tiles * (d->Cout / (64 * c)) >= 2l * p.cus) { nb = c; break; }
    const bool norm = d->src[0].mode == MRISR_SRC_NORM;
    const int grid = tiles * (d->Cout / (64 * nb));
    const size_t lds = 2 * (size_t)(kC1ABytes + nb * kC1WPiece) + (norm ? (size_t)d->Cin * 8 : 0);
    if (norm) {
        if (nb == 4) launch_c1_t<T, true, 4>(p, grid, lds, s);
        else if (nb == 2) launch_c1_t<T, true, 2>(p, grid, lds, s);
        else launch_c1_t<T, true, 1>(p, grid, lds, s);
    } else {
        if (nb == 4) launch_c1_t<T, false, 4>(p, grid, lds, s);
        else if (nb == 2) launch_c1_t<T, false, 2>(p, grid, lds, s);
        else launch_c1_t<T, false, 1>(p, grid, lds, s);
    }
    MRISR_CHECK_LAUNCH("conv_forward(1x1 gemm)");
    return MRISR_OK;
}

int launch_conv1x1_gemm(const mrisr_conv_desc* d, const ConvParams& p, hipStream_t s) {
    if (d->dtype == MRISR_BF16) return launch_c1<bf16_t>(d, p, s);
    return launch_c1<f16_t>(d, p, s);
}
#endif
