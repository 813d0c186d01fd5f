// The two narrow ends of the U-Net, which have no MFMA-shaped contraction and are HBM-bound:
//   stem  : nn.Conv2d(in_channels, f, 3, padding=1, bias=False)                    (unet_model.py:29 for `inc`)
//   head  : GroupNorm+LeakyReLU -> nn.Conv2d(f/2, out_channels, 1) + bias -> sigmoid (unet_model.py:169-172, 211)
// in_channels = out_channels = 1 is the reference's configuration (scripts/train.py:167-173) and the tuned case; a few
// image channels either side (unet_model.py:129: RGB in, RGB out) run the same kernels with a channel loop.
#include "common.h"

// ------------------------------------------------------------------------------------------------ stem
// Input rows of the block's pixel range (+1 row above and below) staged in LDS: the 9 taps of a pixel are LDS reads
// shared by the Cout/VEC threads of that pixel; as 9 predicated 4-byte global loads per thread the pass was bound by
// load-instruction issue (1.7 TB/s).  rows_lds = rows staged, first image row r0 (may be -1: zero row).
__device__ __forceinline__ void stem_stage_rows(const float* __restrict__ xb, float* rows, int H, int W, int r0, int nrows, int t) {
    for (int i = t; i < nrows * W; i += 256) {
        const int r = r0 + i / W;
        rows[i] = (r >= 0 && r < H) ? xb[(size_t)r * W + (i - (i / W) * W)] : 0.f;
    }
}

// MULTI: Cin > 1 image channels (x is NCHW, w is [Cout][9][Cin] as the flat parameter storage holds it): rows of every channel
// plane and the weights staged in LDS
template <typename T, bool MULTI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       T* __restrict__ out, double* __restrict__ stats, int H, int W,
                                                       int Cin, int Cout, int groups, int pix_per_block, int max_rows) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ double smd[];          // [groups*2] statistics (fp64: the order of the LDS atomics must not
    double* sst = smd;                       //  perturb mean / rstd - a 1e-7 wobble flips LeakyReLU signs run to run)
    float* rows = reinterpret_cast<float*>(smd + 2 * (groups > 0 ? groups : 1));     // then [max_rows][W] input rows
    const int t = threadIdx.x, n = blockIdx.y;
    const int HW = H * W;
    const int p0 = blockIdx.x * pix_per_block, pend = min(HW, p0 + pix_per_block);
    const int r0 = p0 / W - 1, nrows = (pend - 1) / W + 1 - r0 + 1;
    const float* xb = x + (size_t)n * Cin * HW;
    const int plane = max_rows * W;
    for (int ci = 0; ci < (MULTI ? Cin : 1); ++ci) stem_stage_rows(xb + (size_t)ci * HW, rows + ci * plane, H, W, r0, nrows, t);
    for (int i = t; i < groups * 2; i += 256) sst[i] = 0.0;
    const int nvec = Cout / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    const int gs = groups > 0 ? Cout / groups : Cout;
    float wr[VEC][9];                        // this thread's 8 output channels x 9 taps, in registers
    float* wl = rows + Cin * plane;          // MULTI: [Cin][9][Cout]
    if (MULTI) {
        for (int i = t; i < Cout * 9 * Cin; i += 256) {
            const int ci = i % Cin, ck = i / Cin;                     // ck = co * 9 + k
            wl[(ci * 9 + ck % 9) * Cout + ck / 9] = w[i];
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[e][k] = w[(c + e) * 9 + k];
    }
    __syncthreads();
    float s[VEC], ss[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    if (pl < ppb)
        for (int pix = p0 + pl; pix < pend; pix += ppb) {
            const int y = pix / W, xx = pix - y * W;
            const float* rl = rows + (y - 1 - r0) * W + xx;     // row y-1 of the staged rows, column xx
            float a[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] = 0.f;
            for (int ci = 0; ci < (MULTI ? Cin : 1); ++ci) {
                float in[9];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    in[r * 3 + 0] = xx > 0 ? rl[r * W - 1] : 0.f;
                    in[r * 3 + 1] = rl[r * W];
                    in[r * 3 + 2] = xx + 1 < W ? rl[r * W + 1] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) a[e] += in[k] * (MULTI ? wl[(ci * 9 + k) * Cout + c + e] : wr[e][k]);
                rl += plane;
            }
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                o.set(e, a[e]);
                const float q = o.get(e);
                s[e] += q;
                ss[e] += q * q;
            }
            store_vec16(out + ((size_t)n * HW + pix) * Cout + c, o);
        }
    if (stats) {
        if (pl < ppb) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                atomicAdd(&sst[2 * ((c + e) / gs)], (double)s[e]);
                atomicAdd(&sst[2 * ((c + e) / gs) + 1], (double)ss[e]);
            }
        }
        __syncthreads();
        for (int i = t; i < groups * 2; i += 256)
            atomic_add_f64(&stats[stat_slot_off(gridDim.y, groups) + (size_t)n * groups * 2 + i], sst[i]);
    }
}

constexpr int kStemMaxCin = 4, kHeadMaxCout = 4;

static int stem_block_pixels(int W, int ppb, int& max_rows, int mult) {
    // pixels per block: mult pixels per thread (forward 32: more blocks in flight; weight gradient 64: every block ends in
    // Cout*9 same-address atomics - 59.6 us at 64, 68.6 at 32, 72 at 128 for 64 channels at 256^2 x 16), at least one row
    int ppblk = ppb * mult;
    if (ppblk < W) ppblk = ceil_div(W, ppb) * ppb;
    max_rows = ppblk / W + 4;
    return ppblk;
}

template <typename T>
static void launch_stem_fwd(dim3 grid, size_t lds, hipStream_t s, const float* x, const float* w, void* out, double* stats, int H,
                            int W, int Cin, int Cout, int groups, int ppblk, int max_rows) {
    if (Cin == 1) stem_fwd_kernel<T, false><<<grid, 256, lds, s>>>(x, w, (T*)out, stats, H, W, 1, Cout, groups, ppblk, max_rows);
    else stem_fwd_kernel<T, true><<<grid, 256, lds, s>>>(x, w, (T*)out, stats, H, W, Cin, Cout, groups, ppblk, max_rows);
}

extern "C" int mrisr_stem_forward_multi(int dtype, const float* x, const float* w, void* out, double* stats, int N, int H,
                                        int W, int Cin, int Cout, int groups, void* stream) {
    if (!x || !w || !out) MRISR_FAIL(MRISR_E_ARG, "stem_forward: null pointer");
    const int vec = mrisr_vec(dtype);
    if (Cout % vec || Cout / vec > 256 || (stats && (groups <= 0 || Cout % groups))) MRISR_FAIL(MRISR_E_SHAPE, "stem_forward: Cout %d", Cout);
    if (Cin < 1 || Cin > kStemMaxCin) MRISR_FAIL(MRISR_E_SHAPE, "stem_forward: %d input channels (1..%d)", Cin, kStemMaxCin);
    const int ppb = 256 / (Cout / vec);
    int max_rows;
    const int ppblk = stem_block_pixels(W, ppb, max_rows, 32);
    dim3 grid(ceil_div(H * W, ppblk), N);
    const size_t lds = (size_t)Cin * max_rows * W * sizeof(float) + (size_t)(groups > 0 ? groups : 1) * 2 * sizeof(double) +
                       (Cin > 1 ? (size_t)Cin * 9 * Cout * sizeof(float) : 0);
    if (lds > 64 * 1024) MRISR_FAIL(MRISR_E_UNSUPPORTED, "stem_forward: image width %d (x %d channels) too large for the row cache", W, Cin);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) launch_stem_fwd<bf16_t>(grid, lds, s, x, w, out, stats, H, W, Cin, Cout, groups, ppblk, max_rows);
    else if (dtype == MRISR_F16) launch_stem_fwd<f16_t>(grid, lds, s, x, w, out, stats, H, W, Cin, Cout, groups, ppblk, max_rows);
    else if (dtype == MRISR_F32) launch_stem_fwd<float>(grid, lds, s, x, w, out, stats, H, W, Cin, Cout, groups, ppblk, max_rows);
    else MRISR_FAIL(MRISR_E_DTYPE, "stem_forward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("stem_forward");
    return MRISR_OK;
}

extern "C" int mrisr_stem_forward(int dtype, const float* x, const float* w, void* out, double* stats, int N, int H,
                                  int W, int Cout, int groups, void* stream) {
    return mrisr_stem_forward_multi(dtype, x, w, out, stats, N, H, W, 1, Cout, groups, stream);
}

// dw[co][tap][ci] += sum_{n,y,x} dy[n,y,x,co] * x[n,ci,y+r-1,x+s-1]     (blockIdx.z = ci)
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ dw, int H, int W, int Cin, int Cout, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float lds[4 * 128 * 9];       // [wave][Cout <= 128][9] (fast epilogue) or [256][9] (one channel element at a time)
    extern __shared__ float rows[];          // [max_rows][W] input rows of this block's pixel range (see stem_fwd_kernel)
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = Cout / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    const int HW = H * W;
    const int ci = blockIdx.z;
    const float* xb = x + ((size_t)n * Cin + ci) * HW;
    dw += ci;
    const int p0 = blockIdx.x * pix_per_block;
    const int r0 = p0 / W - 1, nrows = (min(HW, p0 + pix_per_block) - 1) / W + 1 - r0 + 1;
    stem_stage_rows(xb, rows, H, W, r0, nrows, t);
    __syncthreads();
    float acc[VEC][9];
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[e][k] = 0.f;
    const int pend = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    // four pixels per trip: with one 16-byte load in flight per thread and two workgroups per CU (the row cache keeps
    // the blocks large) the pass ran at 1.3 TB/s
    constexpr int UN = 4;
    if (pl < ppb)
        for (int pix0 = blockIdx.x * pix_per_block + pl; pix0 < pend; pix0 += UN * ppb) {
            Vec16<T> d[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int pix = min(pix0 + j * ppb, pend - 1);          // clamped: a repeated pixel gets weight 0 below
                d[j] = load_vec16(dy + ((size_t)n * HW + pix) * Cout + c);
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int pix = pix0 + j * ppb;
                if (pix >= pend) break;
                const int y = pix / W, xx = pix - y * W;
                const float* rl = rows + (y - 1 - r0) * W + xx;
                float in[9];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    in[r * 3 + 0] = xx > 0 ? rl[r * W - 1] : 0.f;
                    in[r * 3 + 1] = rl[r * W];
                    in[r * 3 + 2] = xx + 1 < W ? rl[r * W + 1] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e)
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc[e][k] += d[j].get(e) * in[k];
            }
        }
    // Epilogue.  Every block ends in Cout*9 float atomics on the same addresses: eight rounds of (barrier, LDS, 72 scattered
    // atomics) cost 0.1 us per block - more than half of the pass at 512 blocks.  When the channel vectors of a pixel are a
    // power of two per wave, the lanes that share a channel vector are summed with wave shuffles first, the four waves meet
    // in LDS once, and the atomics go out as contiguous 256-byte wave instructions.
    if ((nvec & (nvec - 1)) == 0 && nvec <= 64 && Cout <= 128) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = (pl < ppb) ? acc[e][k] : 0.f;
                for (int off = nvec; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
                acc[e][k] = v;
            }
        const int lane = t & 63, wave = t >> 6;
        if (lane < nvec) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int k = 0; k < 9; ++k) lds[(wave * Cout + lane * VEC + e) * 9 + k] = acc[e][k];
        }
        __syncthreads();
        for (int i = t; i < Cout * 9; i += 256)
            atomic_add_f32(&dw[i * Cin], lds[i] + lds[Cout * 9 + i] + lds[2 * Cout * 9 + i] + lds[3 * Cout * 9 + i]);
        return;
    }
    // general shapes: reduce over pixel lanes, one output channel element at a time (keeps LDS small)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) lds[t * 9 + k] = (pl < ppb) ? acc[e][k] : 0.f;
        __syncthreads();
        for (int i = t; i < nvec * 9; i += 256) {
            const int cvj = i / 9, k = i - cvj * 9;
            float a = 0.f;
            for (int q = 0; q < ppb; ++q) a += lds[(q * nvec + cvj) * 9 + k];
            atomic_add_f32(&dw[((cvj * VEC + e) * 9 + k) * Cin], a);
        }
    }
}

extern "C" int mrisr_stem_wgrad_multi(int dtype, const float* x, const void* dy, float* dw, int N, int H, int W, int Cin,
                                      int Cout, void* stream) {
    if (!x || !dy || !dw) MRISR_FAIL(MRISR_E_ARG, "stem_wgrad: null pointer");
    const int vec = mrisr_vec(dtype);
    if (Cout % vec || Cout / vec > 256) MRISR_FAIL(MRISR_E_SHAPE, "stem_wgrad: Cout %d", Cout);
    if (Cin < 1 || Cin > kStemMaxCin) MRISR_FAIL(MRISR_E_SHAPE, "stem_wgrad: %d input channels (1..%d)", Cin, kStemMaxCin);
    const int ppb = 256 / (Cout / vec);
    int max_rows;
    const int ppblk = stem_block_pixels(W, ppb, max_rows, 64);
    dim3 grid(ceil_div(H * W, ppblk), N, Cin);
    const size_t lds = (size_t)max_rows * W * sizeof(float);
    if (lds > 48 * 1024) MRISR_FAIL(MRISR_E_UNSUPPORTED, "stem_wgrad: image width %d too large for the row cache", W);
    if (dtype == MRISR_BF16) stem_wgrad_kernel<bf16_t><<<grid, 256, lds, (hipStream_t)stream>>>(x, (const bf16_t*)dy, dw, H, W, Cin, Cout, ppblk);
    else if (dtype == MRISR_F16) stem_wgrad_kernel<f16_t><<<grid, 256, lds, (hipStream_t)stream>>>(x, (const f16_t*)dy, dw, H, W, Cin, Cout, ppblk);
    else if (dtype == MRISR_F32) stem_wgrad_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>(x, (const float*)dy, dw, H, W, Cin, Cout, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "stem_wgrad: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("stem_wgrad");
    return MRISR_OK;
}

extern "C" int mrisr_stem_wgrad(int dtype, const float* x, const void* dy, float* dw, int N, int H, int W, int Cout,
                                void* stream) {
    return mrisr_stem_wgrad_multi(dtype, x, dy, dw, N, H, W, 1, Cout, stream);
}

// ------------------------------------------------------------------------------------------------ head
// thread = pixel; channels looped (C <= 128); K output channels, out / dout are NCHW fp32 ([N][K][H*W])
template <typename T, int K>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, int HW, int C) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ float sm[];   // scale[C], shift[C], w[K][C]
    const int n = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < C; i += 256) {
        sm[i] = scale[(size_t)n * C + i];
        sm[C + i] = shift[(size_t)n * C + i];
#pragma unroll
        for (int k = 0; k < K; ++k) sm[(2 + k) * C + i] = w[k * C + i];
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + t;
    if (pix >= HW) return;
    const T* px = x + ((size_t)n * HW + pix) * C;
    float z[K];
#pragma unroll
    for (int k = 0; k < K; ++k) z[k] = b[k];
    for (int c = 0; c < C; c += VEC) {
        const Vec16<T> v = load_vec16(px + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float a = lrelu(v.get(e) * sm[c + e] + sm[C + c + e]);
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] += a * sm[(2 + k) * C + c + e];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) out[((size_t)n * K + k) * HW + pix] = 1.f / (1.f + __expf(-z[k]));
}

template <typename T>
static void launch_head_fwd(int K, dim3 grid, size_t lds, hipStream_t s, const void* x, const float* scale, const float* shift,
                            const float* w, const float* b, float* out, int HW, int C) {
    if (K == 1) head_fwd_kernel<T, 1><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, b, out, HW, C);
    else if (K == 2) head_fwd_kernel<T, 2><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, b, out, HW, C);
    else if (K == 3) head_fwd_kernel<T, 3><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, b, out, HW, C);
    else head_fwd_kernel<T, 4><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, b, out, HW, C);
}

extern "C" int mrisr_head_forward_multi(int dtype, const void* x, const float* scale, const float* shift, const float* w,
                                        const float* b, float* out, int N, int H, int W, int C, int K, void* stream) {
    if (!x || !scale || !shift || !w || !b || !out) MRISR_FAIL(MRISR_E_ARG, "head_forward: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec) MRISR_FAIL(MRISR_E_SHAPE, "head_forward: C %d", C);
    if (K < 1 || K > kHeadMaxCout) MRISR_FAIL(MRISR_E_SHAPE, "head_forward: %d output channels (1..%d)", K, kHeadMaxCout);
    dim3 grid(ceil_div(H * W, 256), N);
    const size_t lds = (size_t)(2 + K) * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) launch_head_fwd<bf16_t>(K, grid, lds, s, x, scale, shift, w, b, out, H * W, C);
    else if (dtype == MRISR_F16) launch_head_fwd<f16_t>(K, grid, lds, s, x, scale, shift, w, b, out, H * W, C);
    else if (dtype == MRISR_F32) launch_head_fwd<float>(K, grid, lds, s, x, scale, shift, w, b, out, H * W, C);
    else MRISR_FAIL(MRISR_E_DTYPE, "head_forward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("head_forward");
    return MRISR_OK;
}

extern "C" int mrisr_head_forward(int dtype, const void* x, const float* scale, const float* shift, const float* w,
                                  const float* b, float* out, int N, int H, int W, int C, void* stream) {
    return mrisr_head_forward_multi(dtype, x, scale, shift, w, b, out, N, H, W, C, 1, stream);
}

// dz[k] = dout[k] * out[k] * (1-out[k]);  da[c] = sum_k dz[k] * w[k][c];  dw[k][c] += sum dz[k] * act[c];  db[k] += sum dz[k]
// Thread = (pixel lane, 16-byte channel vector): a wave reads / writes whole pixel rows (C*sizeof(T) contiguous
// bytes per pixel), so x is fetched once and da is written in full lines (a thread-per-pixel loop over channel
// vectors re-fetched every line C/VEC times: 1.2 GB read for a 268 MB tensor).
// (The network's own one-channel head does not come through here in training: norm.hip's GroupNorm-backward passes form
// dz * w on the fly.  This is the stand-alone form, and the one the engine uses for out_channels > 1.)
template <typename T, int K>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       T* __restrict__ da, float* __restrict__ dw, float* __restrict__ db,
                                                       int HW, int C, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ float sm[];   // dwacc[K][C], dbacc[K]
    const int n = blockIdx.y, t = threadIdx.x;
    const int nvec = C / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    for (int i = t; i < K * (C + 1); i += 256) sm[i] = 0.f;
    __syncthreads();
    float sc[VEC], sh[VEC], wv[K][VEC], dws[K][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = scale[(size_t)n * C + c + e];
        sh[e] = shift[(size_t)n * C + c + e];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wv[k][e] = w[k * C + c + e];
            dws[k][e] = 0.f;
        }
    }
    float dbs[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dbs[k] = 0.f;
    const int pend = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    if (pl < ppb) {
        for (int pix = blockIdx.x * pix_per_block + pl; pix < pend; pix += ppb) {
            const size_t gp = (size_t)n * HW + pix;
            float dz[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t go = ((size_t)n * K + k) * HW + pix;
                const float o = out[go];
                dz[k] = dout[go] * o * (1.f - o);
                if (cv == 0) dbs[k] += dz[k];
            }
            const Vec16<T> v = load_vec16(x + gp * C + c);
            Vec16<T> g;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = lrelu(v.get(e) * sc[e] + sh[e]);
                float ga = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    dws[k][e] += dz[k] * a;
                    ga += dz[k] * wv[k][e];
                }
                g.set(e, ga);
            }
            store_vec16(da + gp * C + c, g);
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) atomicAdd(&sm[k * C + c + e], dws[k][e]);     // LDS: ppb adders per address, once per block
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float d = wave_sum(dbs[k]);
        if ((t & 63) == 0) atomicAdd(&sm[K * C + k], d);
    }
    __syncthreads();
    for (int i = t; i < K * C; i += 256) atomic_add_f32(&dw[i], sm[i]);
    if (t < K) atomic_add_f32(db + t, sm[K * C + t]);
}

template <typename T>
static void launch_head_bwd(int K, dim3 grid, size_t lds, hipStream_t s, const void* x, const float* scale, const float* shift,
                            const float* w, const float* out, const float* dout, void* da, float* dw, float* db, int HW, int C,
                            int ppblk) {
    if (K == 1) head_bwd_kernel<T, 1><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, out, dout, (T*)da, dw, db, HW, C, ppblk);
    else if (K == 2) head_bwd_kernel<T, 2><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, out, dout, (T*)da, dw, db, HW, C, ppblk);
    else if (K == 3) head_bwd_kernel<T, 3><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, out, dout, (T*)da, dw, db, HW, C, ppblk);
    else head_bwd_kernel<T, 4><<<grid, 256, lds, s>>>((const T*)x, scale, shift, w, out, dout, (T*)da, dw, db, HW, C, ppblk);
}

extern "C" int mrisr_head_backward_multi(int dtype, const void* x, const float* scale, const float* shift, const float* w,
                                         const float* out, const float* dout, void* da, float* dw, float* db, int N, int H,
                                         int W, int C, int K, void* stream) {
    if (!x || !scale || !shift || !w || !out || !dout || !da || !dw || !db) MRISR_FAIL(MRISR_E_ARG, "head_backward: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec || C / vec > 256) MRISR_FAIL(MRISR_E_SHAPE, "head_backward: C %d", C);
    if (K < 1 || K > kHeadMaxCout) MRISR_FAIL(MRISR_E_SHAPE, "head_backward: %d output channels (1..%d)", K, kHeadMaxCout);
    const int ppblk = 256 * 8;
    dim3 grid(ceil_div(H * W, ppblk), N);
    const size_t lds = (size_t)K * (C + 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) launch_head_bwd<bf16_t>(K, grid, lds, s, x, scale, shift, w, out, dout, da, dw, db, H * W, C, ppblk);
    else if (dtype == MRISR_F16) launch_head_bwd<f16_t>(K, grid, lds, s, x, scale, shift, w, out, dout, da, dw, db, H * W, C, ppblk);
    else if (dtype == MRISR_F32) launch_head_bwd<float>(K, grid, lds, s, x, scale, shift, w, out, dout, da, dw, db, H * W, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "head_backward: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("head_backward");
    return MRISR_OK;
}

extern "C" int mrisr_head_backward(int dtype, const void* x, const float* scale, const float* shift, const float* w,
                                   const float* out, const float* dout, void* da, float* dw, float* db, int N, int H,
                                   int W, int C, void* stream) {
    return mrisr_head_backward_multi(dtype, x, scale, shift, w, out, dout, da, dw, db, N, H, W, C, 1, stream);
}
