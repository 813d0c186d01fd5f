// GroupNorm(8,C)+LeakyReLU(0.2) support kernels (HBM-bound element-wise / reduction passes).
//
// Forward: the convolution epilogue accumulates per-(n,group) sums; gn_finalize turns them into the
// per-(n,c) affine (scale, shift) that consumer convolutions apply while loading
// (/root/reference/models/unet_model.py:30-31 and siblings).
// Backward: aten native_group_norm_backward + leaky_relu_backward + the adjoints of max-pool,
// bilinear upsample, concat and blend, restated as
//   pass 1 (act_bwd_reduce)   g = LeakyReLU'(.) * sum_consumers dL/dact ;  red[n][c] = (sum g, sum g*xhat)
//   finalize                  dgamma, dbeta, and per-(n,c) coefficients A, B, C
//   pass 2 (act_bwd_apply)    dx = g*A + x*B + C
#include "common.h"

// ------------------------------------------------------------------------------------------------
__global__ void gn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ meanrstd, int N, int C,
                                   int groups, double count, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx - n * C;
    const int gs = C / groups, g = c / gs;
    // the 16 slot pairs as 16-byte loads, all in flight at once (this launch sits between every two convolutions of the
    // forward pass: its latency is paid 20 times per step), summed in slot order
    typedef __attribute__((ext_vector_type(2))) double f64x2;
    f64x2 v[MRISR_STAT_SLOTS];
#pragma unroll
    for (int k = 0; k < MRISR_STAT_SLOTS; ++k) v[k] = *reinterpret_cast<const f64x2*>(stats + ((size_t)(k * N + n) * groups + g) * 2);
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int k = 0; k < MRISR_STAT_SLOTS; ++k) { s += v[k][0]; ss += v[k][1]; }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float m = (float)mean;
    const float sc = gamma[c] * rstd;
    scale[idx] = sc;
    shift[idx] = beta[c] - m * sc;
    if (c == g * gs) {
        meanrstd[((size_t)n * groups + g) * 2] = m;
        meanrstd[((size_t)n * groups + g) * 2 + 1] = rstd;
    }
}

extern "C" int mrisr_gn_finalize(const double* stats, const float* gamma, const float* beta, float* scale,
                                 float* shift, float* meanrstd, int N, int C, int groups, double count,
                                 float eps, void* stream) {
    if (!stats || !gamma || !beta || !scale || !shift || !meanrstd) MRISR_FAIL(MRISR_E_ARG, "gn_finalize: null pointer");
    if (groups <= 0 || C % groups) MRISR_FAIL(MRISR_E_SHAPE, "gn_finalize: C %d groups %d", C, groups);
    gn_finalize_kernel<<<ceil_div(N * C, 256), 256, 0, (hipStream_t)stream>>>(stats, gamma, beta, scale, shift,
                                                                             meanrstd, N, C, groups, count, eps);
    MRISR_CHECK_LAUNCH("gn_finalize");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
struct ConsumerDev {
    const void* da;
    int C_total, c_off, H, W, spatial, off_y, off_x, weight_mode;
    const float* head_out;      // MRISR_SP_HEAD: sigmoid output, 1x1 weight, per-image partial sums [N][C+1], and the
    const float* head_w;        // head's own gradient accumulators
    float* head_part;
    float* head_dw;
    float* head_db;
};
struct ActBwdParams {
    const void* x;
    const float* scale;
    const float* shift;
    const float* meanrstd;
    const float* blend_alpha;
    void* g;
    float* red;      // [N][C][2]
    ConsumerDev cons[2];
    int ncons, N, H, W, C, groups, pix_per_block;
    int nvec_shift;  // log2(C / VEC) when that is a power of two, else -1
    float* alpha_slots;   // [256] partial sums of (first consumer's gradient) * activation, or NULL (blend alpha gradient)
};

// Pass-2 coefficients computed inside the apply kernels (mrisr_gn_bwd_fin): what act_bwd_finalize_kernel does for the
// g-tensor path, without its launch in the middle of the gradient chain.
struct FinDev {
    const float* red;        // [N][C][2], complete (written by the pass-1 launch)
    const float* gamma;
    const float* meanrstd;
    float* dgamma;
    float* dbeta;
    const float* alpha_slots;
    const float* alpha;
    float* dalpha;
    float inv_count, alpha_sign;
    int groups;
    const float* head_part;   // MRISR_SP_HEAD consumer: per-image partial sums [N][C+1] -> head_dw [C], head_db [1]
    float* head_dw;
    float* head_db;
};
constexpr int kMaxGroups = 32;

// Every thread of the block calls this (two barriers inside).  s12 = 2 * kMaxGroups floats of LDS.  Fills the
// coefficients of channels c .. c+VEC-1 of image n (see act_bwd_finalize_kernel for the formulas); the block with
// `owner` set adds image n's share to dgamma / dbeta for the channels its threads with `lead` set hold, and the block
// with `first` set turns the blend-alpha slots into dalpha.
template <int VEC>
__device__ __forceinline__ void gn_bwd_coefs(const FinDev& f, float* s12, int n, int C, int c, bool valid, bool owner,
                                             bool lead, bool first, float* ca, float* cb, float* cc) {
    const int t = threadIdx.x, gs = C / f.groups;
    if (t < 2 * kMaxGroups) s12[t] = 0.f;
    __syncthreads();
    const float* rn = f.red + (size_t)n * C * 2;
    for (int j = t; j < C; j += blockDim.x) {
        const float gm = f.gamma[j];
        atomicAdd(&s12[j / gs], gm * rn[2 * j]);
        atomicAdd(&s12[kMaxGroups + j / gs], gm * rn[2 * j + 1]);
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int g = (c + e) / gs;
            const float mean = f.meanrstd[((size_t)n * f.groups + g) * 2], rstd = f.meanrstd[((size_t)n * f.groups + g) * 2 + 1];
            const float S1 = s12[g] * f.inv_count, S2 = s12[kMaxGroups + g] * f.inv_count;
            ca[e] = rstd * f.gamma[c + e];
            cb[e] = -rstd * rstd * S2;
            cc[e] = mean * rstd * rstd * S2 - rstd * S1;
            if (owner && lead) {
                atomic_add_f32(&f.dbeta[c + e], rn[2 * (c + e)]);
                atomic_add_f32(&f.dgamma[c + e], rn[2 * (c + e) + 1]);
                if (f.head_part) atomic_add_f32(&f.head_dw[c + e], f.head_part[(size_t)n * (C + 1) + c + e]);
            }
        }
        if (owner && lead && f.head_part && c == 0) atomic_add_f32(f.head_db, f.head_part[(size_t)n * (C + 1) + C]);
    }
    if (first && f.alpha_slots && t < 64) {      // dalpha += sign * sigmoid'(alpha) * sum d*act  (unet_model.py:206-207)
        float v = 0.f;
        for (int i = t; i < 256; i += 64) v += f.alpha_slots[i];
        v = wave_sum(v);
        if (t == 0) {
            const float sg = 1.f / (1.f + __expf(-f.alpha[0]));
            atomic_add_f32(f.dalpha, f.alpha_sign * sg * (1.f - sg) * v);
        }
    }
}

template <typename T>
__device__ __forceinline__ void act_of(const T* p, const float* sc, const float* sh, float* o) {
    const Vec16<T> v = load_vec16(p);
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) o[e] = lrelu(v.get(e) * sc[e] + sh[e]);
}

// candidate high-res rows of a bilinear x2 (align_corners) adjoint and their weights for low-res index y:
// src(Y) = Y (n-1)/(2n-1) touches row y only for Y in [2y-1, 2y+2] (checked exhaustively for n <= 512)
constexpr int kUpAdj = 4;
__device__ __forceinline__ void up2_adjoint_weights(int y, int in_size, int* idx, float* w) {
#pragma unroll
    for (int k = 0; k < kUpAdj; ++k) {
        const int Y = 2 * y - 1 + k;
        float wt = 0.f;
        if (Y >= 0 && Y < 2 * in_size) {
            int i0, i1;
            float w1;
            up2_coord(Y, in_size, i0, i1, w1);
            if (i0 == y) wt += 1.f - w1;
            if (i1 == y) wt += w1;
        }
        idx[k] = Y;
        w[k] = wt;
    }
}

// PLAIN: every consumer is MRISR_SP_NONE (17 of the 20 nodes of the U-Net) - compiled without the pool / bilinear
// adjoint code, whose 32-float windows cost the generic kernel 155 VGPRs = 3 waves per SIMD, too few loads in
// flight for an HBM-bound pass (measured 3.0 TB/s).
// KIND 3: the single consumer is the output head (MRISR_SP_HEAD): dL/dact = dz * w[c] is formed on the fly from the
// one-channel dz = dL/dout * out * (1 - out), and the head's dW / db fall out of the same pass.
template <typename T, int KIND>   // 0: plain consumers only, 1: plain + 2x2 max-pool, 2: + bilinear adjoint gather, 3: head
__global__ __launch_bounds__(256) void act_bwd_reduce_kernel(const ActBwdParams p) {
    constexpr bool PLAIN = KIND == 0 || KIND == 3;
    constexpr bool HEAD = KIND == 3;
    constexpr int VEC = Vec16<T>::N;
    __shared__ float lds[256 * VEC * 2];
    const int t = threadIdx.x, n = blockIdx.y;
    // blockIdx.z = channel slice of 256 vectors (wide fp32 nodes: C / VEC > 256); one slice everywhere else
    const int cz = blockIdx.z * 256 * VEC;
    const int nvec = min(256, p.C / VEC - (int)blockIdx.z * 256), ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec;
    const bool active = pl < ppb;
    const int c = cz + cv * VEC;
    const int HW = p.H * p.W;
    const int gs = p.C / p.groups;

    float sc[VEC], sh[VEC], sA[VEC], sB[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = p.scale[(size_t)n * p.C + c + e];
        sh[e] = p.shift[(size_t)n * p.C + c + e];
        sA[e] = 0.f;
        sB[e] = 0.f;
    }
    float bw[3] = {1.f, 1.f, 1.f};
    if (p.blend_alpha) {
        const float a = 1.f / (1.f + __expf(-p.blend_alpha[0]));
        bw[1] = a;
        bw[2] = 1.f - a;
    }
    const T* xb = (const T*)p.x + (size_t)n * HW * p.C;
    T* gb = (T*)p.g + (size_t)n * HW * p.C;
    const int pix_end = min(HW, (int)(blockIdx.x + 1) * p.pix_per_block);
    float adot = 0.f;
    float hw[HEAD ? VEC : 1], hdw[HEAD ? VEC : 1], hdb = 0.f;
    if constexpr (HEAD) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { hw[e] = p.cons[0].head_w[c + e]; hdw[e] = 0.f; }
    }
    if (active) {
        for (int pix = blockIdx.x * p.pix_per_block + pl; pix < pix_end; pix += ppb) {
            const int y = pix / p.W, x = pix - y * p.W;
            const Vec16<T> xv = load_vec16(xb + (size_t)pix * p.C + c);
            float gact[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) gact[e] = 0.f;
            if constexpr (HEAD) {
                const size_t gp = (size_t)n * HW + pix;
                const float o = p.cons[0].head_out[gp];
                const float dz = ((const float*)p.cons[0].da)[gp] * o * (1.f - o);
                if (cv == 0 && blockIdx.z == 0) hdb += dz;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    gact[e] = dz * hw[e];
                    hdw[e] += dz * lrelu(xv.get(e) * sc[e] + sh[e]);
                }
            }
            for (int k = 0; k < (HEAD ? 0 : p.ncons); ++k) {
                const ConsumerDev& cs = p.cons[k];
                const T* dab = (const T*)cs.da + (size_t)n * cs.H * cs.W * cs.C_total + cs.c_off + c;
                const float wgt = cs.weight_mode == 0 ? 1.f : (cs.weight_mode == 1 ? bw[1] : bw[2]);
                if (PLAIN || cs.spatial == MRISR_SP_NONE) {
                    const int yy = y + cs.off_y, xx = x + cs.off_x;
                    if (yy < cs.H && xx < cs.W) {
                        const Vec16<T> d = load_vec16(dab + ((size_t)yy * cs.W + xx) * cs.C_total);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) gact[e] += wgt * d.get(e);
                        if (p.alpha_slots && k == 0) {       // blend branch: sum d * act feeds dL/dalpha
#pragma unroll
                            for (int e = 0; e < VEC; ++e) adot += d.get(e) * lrelu(xv.get(e) * sc[e] + sh[e]);
                        }
                    }
                } else if constexpr (PLAIN) {
                } else if (cs.spatial == MRISR_SP_POOL2) {
                    const int py = y >> 1, px = x >> 1;
                    if (py < cs.H && px < cs.W) {
                        // recompute the 2x2 window's activations one at a time (two 8-float sets live instead of
                        // five): this pixel wins iff it beats every EARLIER window element strictly and every
                        // later one weakly - aten's scan keeps the first maximum
                        const int me = ((y & 1) << 1) | (x & 1);
                        float am[VEC];
                        bool win[VEC];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) { am[e] = lrelu(xv.get(e) * sc[e] + sh[e]); win[e] = true; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q == me) continue;
                            float aq[VEC];
                            act_of<T>(xb + ((size_t)(2 * py + (q >> 1)) * p.W + 2 * px + (q & 1)) * p.C + c, sc, sh, aq);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) win[e] = win[e] && (q < me ? aq[e] < am[e] : aq[e] <= am[e]);
                        }
                        const Vec16<T> d = load_vec16(dab + ((size_t)py * cs.W + px) * cs.C_total);
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (win[e]) gact[e] += wgt * d.get(e);
                    }
                } else if constexpr (KIND == 2) {   // adjoint of bilinear x2 (align_corners=True)
                    int iy[kUpAdj], ix[kUpAdj];
                    float wy[kUpAdj], wx[kUpAdj];
                    up2_adjoint_weights(y, p.H, iy, wy);
                    up2_adjoint_weights(x, p.W, ix, wx);
#pragma unroll
                    for (int a = 0; a < kUpAdj; ++a) {
                        if (wy[a] == 0.f) continue;
#pragma unroll
                        for (int b = 0; b < kUpAdj; ++b) {
                            if (wx[b] == 0.f) continue;
                            const int yy = iy[a] + cs.off_y, xx = ix[b] + cs.off_x;
                            const Vec16<T> d = load_vec16(dab + ((size_t)yy * cs.W + xx) * cs.C_total);
                            const float wv = wgt * wy[a] * wx[b];
#pragma unroll
                            for (int e = 0; e < VEC; ++e) gact[e] += wv * d.get(e);
                        }
                    }
                }
            }
            Vec16<T> gv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xr = xv.get(e);
                const float pre = xr * sc[e] + sh[e];
                const float gy = gact[e] * (pre > 0.f ? 1.f : LRELU_SLOPE);
                gv.set(e, gy);
                const float gq = p.g ? gv.get(e) : gy;      // the value pass 2 will use (rounded when it goes through g)
                sA[e] += gq;
                sB[e] += gq * xr;                           // sum g*x; turned into sum g*xhat after the loop
            }
            if (p.g) store_vec16(gb + (size_t)pix * p.C + c, gv);
        }
    }
    // sum g*xhat = rstd * (sum g*x - mean * sum g), per thread (<= 16 pixels each: no cancellation to speak of); the
    // group statistics are only needed here, so they stay out of the streaming loop's register budget
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = (c + e) / gs;
        const float mean = p.meanrstd[((size_t)n * p.groups + g) * 2], rstd = p.meanrstd[((size_t)n * p.groups + g) * 2 + 1];
        sB[e] = rstd * (sB[e] - mean * sA[e]);
    }
    if (p.alpha_slots) {      // one atomic per wave, spread over 256 slots (summed by act_bwd_finalize)
        const float w = wave_sum(active ? adot : 0.f);
        if ((t & 63) == 0) atomic_add_f32(&p.alpha_slots[(blockIdx.x * 4 + (t >> 6) + blockIdx.y * 37) & 255], w);
    }
    // block reduction over the pixel lanes that share a channel vector
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        lds[(t * VEC + e) * 2] = active ? sA[e] : 0.f;
        lds[(t * VEC + e) * 2 + 1] = active ? sB[e] : 0.f;
    }
    __syncthreads();
    for (int i = t; i < nvec * VEC * 2; i += 256) {
        const int j = i >> 1, which = i & 1;
        const int cvj = j / VEC, e = j - cvj * VEC;
        float s = 0.f;
        for (int q = 0; q < ppb; ++q) s += lds[((q * nvec + cvj) * VEC + e) * 2 + which];
        atomic_add_f32(&p.red[((size_t)n * p.C + cz + cvj * VEC + e) * 2 + which], s);
    }
    if constexpr (HEAD) {     // the head's weight / bias gradient (what head_bwd_kernel accumulates), per image: all the
        __syncthreads();      // blocks of the launch adding into ONE row of C floats serialise at the memory side
#pragma unroll
        for (int e = 0; e < VEC; ++e) lds[t * VEC + e] = active ? hdw[e] : 0.f;
        const float b = wave_sum(active ? hdb : 0.f);
        if ((t & 63) == 0) lds[256 * VEC + (t >> 6)] = b;
        __syncthreads();
        float* part = p.cons[0].head_part + (size_t)n * (p.C + 1);
        for (int j = t; j < nvec * VEC; j += 256) {
            const int cvj = j / VEC, e = j - cvj * VEC;
            float s = 0.f;
            for (int q = 0; q < ppb; ++q) s += lds[(q * nvec + cvj) * VEC + e];
            atomic_add_f32(&part[cz + cvj * VEC + e], s);
        }
        if (t == 0 && blockIdx.z == 0) atomic_add_f32(&part[p.C], lds[256 * VEC] + lds[256 * VEC + 1] + lds[256 * VEC + 2] + lds[256 * VEC + 3]);
    }
}

// Nodes whose activation is also max-pooled (the encoder skips x1..x3): thread = (pooled pixel, channel vector) owning
// the whole 2x2 window - the four activations and the arg-max are computed ONCE per window (the per-pixel kernel
// recomputed them in each of the window's four threads and needed 120-140 VGPRs), for pass 1 (APPLY = false:
// per-(n,c) sums) and pass 2 (APPLY = true: dx, no intermediate g tensor).  Needs even H, W and - if there is a
// second consumer - a plain one with the node's own geometry (host-checked: act_bwd_window_ok).
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void act_bwd_pool_window_kernel(const ActBwdParams p, const float* __restrict__ coef,
                                                                  T* __restrict__ dx, const FinDev fin) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float lds[APPLY ? 2 * kMaxGroups : 256 * VEC * 2];
    const int t = threadIdx.x, n = blockIdx.y;
    const int cz = blockIdx.z * 256 * VEC;        // channel slice, see act_bwd_reduce_kernel
    const int nvec = min(256, p.C / VEC - (int)blockIdx.z * 256), ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cz + cv * VEC;
    const bool active = pl < ppb;
    const int Hp = p.H / 2, Wp = p.W / 2, HWp = Hp * Wp, W = p.W;
    const int kp = p.cons[0].spatial == MRISR_SP_POOL2 ? 0 : 1, ko = 1 - kp;
    const bool has_o = p.ncons > 1;
    float bw1 = 1.f, bw2 = 1.f;
    if (p.blend_alpha) { bw1 = 1.f / (1.f + __expf(-p.blend_alpha[0])); bw2 = 1.f - bw1; }
    const int mp = p.cons[kp].weight_mode, mo = has_o ? p.cons[ko].weight_mode : 0;
    const float wp = mp == 0 ? 1.f : (mp == 1 ? bw1 : bw2), wo = mo == 0 ? 1.f : (mo == 1 ? bw1 : bw2);
    const size_t k0 = (size_t)n * p.C + c, NC = (size_t)p.N * p.C;
    float sc[VEC], sh[VEC], u0[VEC], u1[VEC], u2[VEC];     // APPLY: u = (cA, cB, cC); reduce: u0 = sum g, u1 = sum g*x
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = p.scale[k0 + e]; sh[e] = p.shift[k0 + e];
        if constexpr (APPLY) {
            if (coef) { u0[e] = coef[k0 + e]; u1[e] = coef[NC + k0 + e]; u2[e] = coef[2 * NC + k0 + e]; }
        } else { u0[e] = 0.f; u1[e] = 0.f; u2[e] = 0.f; }
    }
    if constexpr (APPLY) {
        if (!coef) gn_bwd_coefs<VEC>(fin, lds, n, p.C, c, active, blockIdx.x == 0, pl == 0, blockIdx.x + blockIdx.y + blockIdx.z == 0, u0, u1, u2);
    }
    const T* xb = (const T*)p.x + (size_t)n * p.H * W * p.C + c;
    const T* dpb = (const T*)p.cons[kp].da + (size_t)n * HWp * p.cons[kp].C_total + p.cons[kp].c_off + c;
    const int Cto = has_o ? p.cons[ko].C_total : 0;
    const T* dob = has_o ? (const T*)p.cons[ko].da + (size_t)n * p.H * W * Cto + p.cons[ko].c_off + c : xb;
    T* ob = APPLY ? dx + (size_t)n * p.H * W * p.C + c : nullptr;
    const int pend = min(HWp, (int)(blockIdx.x + 1) * p.pix_per_block);
    if (active) {
        for (int pp = blockIdx.x * p.pix_per_block + pl; pp < pend; pp += ppb) {
            const int py = pp / Wp, px = pp - py * Wp;
            const size_t b0 = (size_t)(2 * py) * W + 2 * px;
            const size_t off[4] = {b0, b0 + 1, b0 + W, b0 + W + 1};
            Vec16<T> xv[4], dv[4], ov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = load_vec16(xb + off[q] * p.C);
            const Vec16<T> dp = load_vec16(dpb + (size_t)pp * p.cons[kp].C_total);
            if (has_o) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dv[q] = load_vec16(dob + off[q] * Cto);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float xr[4], pre[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { xr[q] = xv[q].get(e); pre[q] = xr[q] * sc[e] + sh[e]; }
                // first maximum of the activations in scan order (aten max_pool2d); LeakyReLU is monotone, so the
                // comparison runs on act = lrelu(pre)
                int win = 0;
                float m = lrelu(pre[0]);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float a = lrelu(pre[q]);
                    if (a > m) { m = a; win = q; }
                }
                const float gp = wp * dp.get(e);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float gact = q == win ? gp : 0.f;
                    if (has_o) gact += wo * dv[q].get(e);
                    const float gy = gact * (pre[q] > 0.f ? 1.f : LRELU_SLOPE);
                    if constexpr (APPLY) ov[q].set(e, gy * u0[e] + xr[q] * u1[e] + u2[e]);
                    else { u0[e] += gy; u1[e] += gy * xr[q]; }
                }
            }
            if constexpr (APPLY) {
#pragma unroll
                for (int q = 0; q < 4; ++q) store_vec16(ob + off[q] * p.C, ov[q]);
            }
        }
    }
    if constexpr (!APPLY) {
        const int gs = p.C / p.groups;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int g = (c + e) / gs;
            const float mean = p.meanrstd[((size_t)n * p.groups + g) * 2], rstd = p.meanrstd[((size_t)n * p.groups + g) * 2 + 1];
            u1[e] = rstd * (u1[e] - mean * u0[e]);
            lds[(t * VEC + e) * 2] = active ? u0[e] : 0.f;
            lds[(t * VEC + e) * 2 + 1] = active ? u1[e] : 0.f;
        }
        __syncthreads();
        for (int i = t; i < nvec * VEC * 2; i += 256) {
            const int j = i >> 1, which = i & 1;
            const int cvj = j / VEC, e = j - cvj * VEC;
            float s = 0.f;
            for (int q = 0; q < ppb; ++q) s += lds[((q * nvec + cvj) * VEC + e) * 2 + which];
            atomic_add_f32(&p.red[((size_t)n * p.C + cz + cvj * VEC + e) * 2 + which], s);
        }
    }
}

// window kernels apply when: one 2x2-pool consumer, even H and W, and the optional second consumer is plain with the
// node's own geometry
static bool act_bwd_window_ok(int nconsumers, const mrisr_consumer* cs, int H, int W) {
    if ((H | W) & 1) return false;
    int npool = 0;
    for (int k = 0; k < nconsumers; ++k) {
        if (cs[k].spatial == MRISR_SP_POOL2) ++npool;
        else if (cs[k].spatial != MRISR_SP_NONE || cs[k].H != H || cs[k].W != W || cs[k].off_y || cs[k].off_x) return false;
    }
    return npool == 1;
}

template <typename T, bool SAME, bool HEAD = false>
__global__ void act_bwd_apply_fused_kernel(const ActBwdParams p, const float* __restrict__ coef, T* __restrict__ dx, const FinDev fin);

// MRISR_SP_HEAD: the only consumer, same geometry as the node, no blend weight
static int check_head_consumer(int nconsumers, const mrisr_consumer* cs, int H, int W, int C, bool reduce, const char* who) {
    bool head = false;
    for (int k = 0; k < nconsumers; ++k) head = head || cs[k].spatial == MRISR_SP_HEAD;
    if (!head) return MRISR_OK;
    const mrisr_consumer& c = cs[0];
    if (nconsumers != 1 || c.H != H || c.W != W || c.off_y || c.off_x || c.weight_mode || c.c_off || c.C_total != C)
        MRISR_FAIL(MRISR_E_ARG, "%s: a head consumer is the only consumer and has the node's geometry", who);
    if (!c.head_out || !c.head_w || !c.head_part || (!reduce && (!c.head_dw || !c.head_db))) MRISR_FAIL(MRISR_E_ARG, "%s: head consumer null pointer", who);
    return MRISR_OK;
}

static int fill_act_bwd_params(ActBwdParams& p, int dtype, int nconsumers, const mrisr_consumer* consumers,
                               const float* blend_alpha, int H, int W, int C, const char* who) {
    const int vec = mrisr_vec(dtype);
    for (int k = 0; k < nconsumers; ++k) {
        const mrisr_consumer& c = consumers[k];
        if (!c.da) MRISR_FAIL(MRISR_E_ARG, "%s: consumer %d null", who, k);
        if (c.weight_mode < 0 || c.weight_mode > 2 || (c.weight_mode && !blend_alpha)) MRISR_FAIL(MRISR_E_ARG, "%s: weight_mode", who);
        if (c.c_off % vec || c.C_total % vec || c.c_off + C > c.C_total) MRISR_FAIL(MRISR_E_SHAPE, "%s: consumer %d channels", who, k);
        if (c.spatial == MRISR_SP_UP2 && (c.off_y + 2 * H > c.H || c.off_x + 2 * W > c.W)) MRISR_FAIL(MRISR_E_SHAPE, "%s: consumer %d UP2 extent", who, k);
        if (c.spatial == MRISR_SP_POOL2 && (c.H != H / 2 || c.W != W / 2)) MRISR_FAIL(MRISR_E_SHAPE, "%s: consumer %d POOL2 extent", who, k);
        p.cons[k] = ConsumerDev{c.da, c.C_total, c.c_off, c.H, c.W, c.spatial, c.off_y, c.off_x, c.weight_mode, c.head_out, c.head_w, c.head_part, c.head_dw, c.head_db};
    }
    return MRISR_OK;
}

extern "C" int mrisr_act_bwd_apply_fused(int dtype, const void* x, const float* scale, const float* shift,
                                         int nconsumers, const mrisr_consumer* consumers, const float* blend_alpha,
                                         const float* coef, const mrisr_gn_bwd_fin* fin, void* dx, int N, int H, int W,
                                         int C, void* stream) {
    if (!x || !scale || !shift || !dx || !consumers) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: null pointer");
    if ((coef != nullptr) == (fin != nullptr)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: exactly one of coef / fin");
    FinDev fd;
    memset(&fd, 0, sizeof(fd));
    if (fin) {
        if (!fin->red || !fin->gamma || !fin->meanrstd || !fin->dgamma || !fin->dbeta) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: fin null pointer");
        if (fin->groups <= 0 || fin->groups > kMaxGroups || C % fin->groups || !(fin->count > 0)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply_fused: fin groups %d", fin->groups);
        if (fin->alpha_slots && (!fin->alpha || !fin->dalpha)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: alpha_slots without alpha/dalpha");
        fd = FinDev{fin->red, fin->gamma, fin->meanrstd, fin->dgamma, fin->dbeta, fin->alpha_slots, fin->alpha, fin->dalpha,
                    (float)(1.0 / fin->count), fin->alpha_sign, fin->groups, nullptr, nullptr, nullptr};
    }
    if (nconsumers < 1 || nconsumers > 2) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: %d consumers", nconsumers);
    const int vec = mrisr_vec(dtype);
    if (C % vec) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply_fused: C %d", C);
    int hrc = check_head_consumer(nconsumers, consumers, H, W, C, false, "act_bwd_apply_fused");
    if (hrc) return hrc;
    const bool head = consumers[0].spatial == MRISR_SP_HEAD;
    if (head) {
        if (!fin) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused: a head consumer needs fin (its dW / db are added up in this launch)");
        fd.head_part = consumers[0].head_part; fd.head_dw = consumers[0].head_dw; fd.head_db = consumers[0].head_db;
    }
    bool plain = true;
    for (int k = 0; k < nconsumers; ++k) plain = plain && (consumers[k].spatial == MRISR_SP_NONE || head);
    const bool window = !plain && act_bwd_window_ok(nconsumers, consumers, H, W);
    if (!plain && !window) MRISR_FAIL(MRISR_E_UNSUPPORTED, "act_bwd_apply_fused: plain consumers (or one 2x2-pool consumer on an even-sized node) only");
    ActBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.scale = scale; p.shift = shift; p.blend_alpha = blend_alpha;
    p.ncons = nconsumers; p.N = N; p.H = H; p.W = W; p.C = C;
    int rc = fill_act_bwd_params(p, dtype, nconsumers, consumers, blend_alpha, H, W, C, "act_bwd_apply_fused");
    if (rc) return rc;
    // channel slices of 256 vectors (blockIdx.z); the pixel blocking follows the widest slice
    const int nslice = ceil_div(C / vec, 256), nvs = nslice > 1 ? 256 : C / vec;
    hipStream_t s = (hipStream_t)stream;
    if (window) {
        const int nvw = nvs, ppbw = 256 / nvw, HWp = (H / 2) * (W / 2);
        int ppw = ppbw * 16;
        if (ppw > HWp) ppw = ceil_div(HWp, ppbw) * ppbw;
        p.pix_per_block = ppw;
        dim3 gridw(ceil_div(HWp, ppw), N, nslice);
        if (dtype == MRISR_BF16) act_bwd_pool_window_kernel<bf16_t, true><<<gridw, 256, 0, s>>>(p, coef, (bf16_t*)dx, fd);
        else if (dtype == MRISR_F16) act_bwd_pool_window_kernel<f16_t, true><<<gridw, 256, 0, s>>>(p, coef, (f16_t*)dx, fd);
        else if (dtype == MRISR_F32) act_bwd_pool_window_kernel<float, true><<<gridw, 256, 0, s>>>(p, coef, (float*)dx, fd);
        else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_apply_fused: dtype %d", dtype);
        MRISR_CHECK_LAUNCH("act_bwd_apply_fused");
        return MRISR_OK;
    }
    bool same = true;
    for (int k = 0; k < nconsumers; ++k)
        same = same && consumers[k].H == H && consumers[k].W == W && consumers[k].off_y == 0 && consumers[k].off_x == 0;
    const int nvec = nvs, ppb = 256 / nvec, HW = H * W;
    int ppblk = ppb * 32;
    if (ppblk > HW) ppblk = ceil_div(HW, ppb) * ppb;
    p.pix_per_block = ppblk;
    dim3 grid(ceil_div(HW, ppblk), N, nslice);
    if (dtype == MRISR_BF16) {
        if (head) act_bwd_apply_fused_kernel<bf16_t, true, true><<<grid, 256, 0, s>>>(p, coef, (bf16_t*)dx, fd);
        else if (same) act_bwd_apply_fused_kernel<bf16_t, true><<<grid, 256, 0, s>>>(p, coef, (bf16_t*)dx, fd);
        else act_bwd_apply_fused_kernel<bf16_t, false><<<grid, 256, 0, s>>>(p, coef, (bf16_t*)dx, fd);
    } else if (dtype == MRISR_F16) {
        if (head) act_bwd_apply_fused_kernel<f16_t, true, true><<<grid, 256, 0, s>>>(p, coef, (f16_t*)dx, fd);
        else if (same) act_bwd_apply_fused_kernel<f16_t, true><<<grid, 256, 0, s>>>(p, coef, (f16_t*)dx, fd);
        else act_bwd_apply_fused_kernel<f16_t, false><<<grid, 256, 0, s>>>(p, coef, (f16_t*)dx, fd);
    } else if (dtype == MRISR_F32) {
        if (head) act_bwd_apply_fused_kernel<float, true, true><<<grid, 256, 0, s>>>(p, coef, (float*)dx, fd);
        else if (same) act_bwd_apply_fused_kernel<float, true><<<grid, 256, 0, s>>>(p, coef, (float*)dx, fd);
        else act_bwd_apply_fused_kernel<float, false><<<grid, 256, 0, s>>>(p, coef, (float*)dx, fd);
    } else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_apply_fused: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("act_bwd_apply_fused");
    return MRISR_OK;
}

extern "C" int mrisr_act_bwd_reduce(int dtype, const void* x, const float* scale, const float* shift,
                                    const float* meanrstd, int nconsumers, const mrisr_consumer* consumers,
                                    const float* blend_alpha, void* g, float* red, float* alpha_slots, int N, int H,
                                    int W, int C, int groups, void* stream) {
    if (!x || !scale || !shift || !meanrstd || !red || !consumers) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: null pointer");
    if (nconsumers < 1 || nconsumers > 2) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: %d consumers", nconsumers);
    const int vec = mrisr_vec(dtype);
    if (C % vec || groups <= 0 || C % groups) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_reduce: C %d", C);
    ActBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.scale = scale; p.shift = shift; p.meanrstd = meanrstd; p.blend_alpha = blend_alpha; p.g = g; p.red = red;
    p.alpha_slots = alpha_slots;
    if (alpha_slots && (consumers[0].spatial != MRISR_SP_NONE || !blend_alpha)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: alpha_slots needs a plain first consumer and blend_alpha");
    int hrc = check_head_consumer(nconsumers, consumers, H, W, C, true, "act_bwd_reduce");
    if (hrc) return hrc;
    const bool head = consumers[0].spatial == MRISR_SP_HEAD;
    if (head && g) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: a head consumer goes with g = NULL (mrisr_act_bwd_apply_fused)");
    p.ncons = nconsumers; p.N = N; p.H = H; p.W = W; p.C = C; p.groups = groups;
    for (int k = 0; k < nconsumers; ++k) {
        const mrisr_consumer& c = consumers[k];
        if (!c.da) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: consumer %d null", k);
        if (c.weight_mode < 0 || c.weight_mode > 2 || (c.weight_mode && !blend_alpha)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: weight_mode");
        if (c.c_off % vec || c.C_total % vec || c.c_off + C > c.C_total) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_reduce: consumer %d channels", k);
        if (c.spatial == MRISR_SP_UP2 && (c.off_y + 2 * H > c.H || c.off_x + 2 * W > c.W)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_reduce: consumer %d UP2 extent", k);
        if (c.spatial == MRISR_SP_POOL2 && (c.H != H / 2 || c.W != W / 2)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_reduce: consumer %d POOL2 extent", k);
        p.cons[k] = ConsumerDev{c.da, c.C_total, c.c_off, c.H, c.W, c.spatial, c.off_y, c.off_x, c.weight_mode, c.head_out, c.head_w, c.head_part, c.head_dw, c.head_db};
    }
    const int nslice = ceil_div(C / vec, 256);     // channel slices of 256 vectors (blockIdx.z)
    const int nvec = nslice > 1 ? 256 : C / vec, ppb = 256 / nvec;
    int ppblk = ppb * 16;      // (32 measured slower: fewer blocks, longer tail)
    const int HW = H * W;
    if (ppblk > HW) ppblk = ceil_div(HW, ppb) * ppb;
    p.pix_per_block = ppblk;
    dim3 grid(ceil_div(HW, ppblk), N, nslice);
    if (!g && !alpha_slots && act_bwd_window_ok(nconsumers, consumers, H, W)) {     // pass 1 of the window pair
        const int HWp = (H / 2) * (W / 2);
        int ppw = ppb * 16;
        if (ppw > HWp) ppw = ceil_div(HWp, ppb) * ppb;
        p.pix_per_block = ppw;
        dim3 gridw(ceil_div(HWp, ppw), N, nslice);
        hipStream_t sw = (hipStream_t)stream;
        if (dtype == MRISR_BF16) act_bwd_pool_window_kernel<bf16_t, false><<<gridw, 256, 0, sw>>>(p, nullptr, (bf16_t*)nullptr, FinDev{});
        else if (dtype == MRISR_F16) act_bwd_pool_window_kernel<f16_t, false><<<gridw, 256, 0, sw>>>(p, nullptr, (f16_t*)nullptr, FinDev{});
        else if (dtype == MRISR_F32) act_bwd_pool_window_kernel<float, false><<<gridw, 256, 0, sw>>>(p, nullptr, (float*)nullptr, FinDev{});
        else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_reduce: dtype %d", dtype);
        MRISR_CHECK_LAUNCH("act_bwd_reduce");
        return MRISR_OK;
    }
    if (!g && !head) {
        for (int k = 0; k < nconsumers; ++k)
            if (consumers[k].spatial != MRISR_SP_NONE) MRISR_FAIL(MRISR_E_ARG, "act_bwd_reduce: g = NULL needs plain consumers or a window-eligible pooled node");
    }
    int kind = head ? 3 : 0;
    for (int k = 0; k < nconsumers; ++k) {
        if (consumers[k].spatial == MRISR_SP_POOL2 && kind < 1) kind = 1;
        if (consumers[k].spatial == MRISR_SP_UP2) kind = 2;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) {
        if (kind == 0) act_bwd_reduce_kernel<bf16_t, 0><<<grid, 256, 0, s>>>(p);
        else if (kind == 1) act_bwd_reduce_kernel<bf16_t, 1><<<grid, 256, 0, s>>>(p);
        else if (kind == 3) act_bwd_reduce_kernel<bf16_t, 3><<<grid, 256, 0, s>>>(p);
        else act_bwd_reduce_kernel<bf16_t, 2><<<grid, 256, 0, s>>>(p);
    } else if (dtype == MRISR_F16) {
        if (kind == 0) act_bwd_reduce_kernel<f16_t, 0><<<grid, 256, 0, s>>>(p);
        else if (kind == 1) act_bwd_reduce_kernel<f16_t, 1><<<grid, 256, 0, s>>>(p);
        else if (kind == 3) act_bwd_reduce_kernel<f16_t, 3><<<grid, 256, 0, s>>>(p);
        else act_bwd_reduce_kernel<f16_t, 2><<<grid, 256, 0, s>>>(p);
    } else if (dtype == MRISR_F32) {
        if (kind == 0) act_bwd_reduce_kernel<float, 0><<<grid, 256, 0, s>>>(p);
        else if (kind == 1) act_bwd_reduce_kernel<float, 1><<<grid, 256, 0, s>>>(p);
        else if (kind == 3) act_bwd_reduce_kernel<float, 3><<<grid, 256, 0, s>>>(p);
        else act_bwd_reduce_kernel<float, 2><<<grid, 256, 0, s>>>(p);
    } else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_reduce: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("act_bwd_reduce");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// red[N][C][2] -> dgamma[C] += sum_n B, dbeta[C] += sum_n A, and the pass-2 coefficients
//   dx = g*cA + x*cB + cC  with  cA = rstd*gamma, cB = -rstd^2*S2/M, cC = mean*rstd^2*S2/M - rstd*S1/M
//   S1[n,g] = sum_{c in g} gamma_c A[n,c],  S2[n,g] = sum_{c in g} gamma_c B[n,c],  M = count.
__global__ void act_bwd_finalize_kernel(const float* __restrict__ red, const float* __restrict__ gamma,
                                        const float* __restrict__ meanrstd, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta, float* __restrict__ coef, int N, int C,
                                        int groups, float inv_count, const float* __restrict__ alpha_slots,
                                        const float* __restrict__ alpha, float* __restrict__ dalpha, float alpha_sign) {
    if (alpha_slots && blockIdx.x == 0) {   // dalpha += sign * sigmoid'(alpha) * sum d*act  (unet_model.py:206-207)
        float v = 0.f;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) v += alpha_slots[i];
        v = wave_sum(v);
        if (threadIdx.x == 0) {
            const float sg = 1.f / (1.f + __expf(-alpha[0]));
            atomic_add_f32(dalpha, alpha_sign * sg * (1.f - sg) * v);
        }
    }
    // one block per (n, group); coef is [3][N][C]
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int gs = C / groups;
    __shared__ float s1s, s2s;
    if (threadIdx.x == 0) { s1s = 0.f; s2s = 0.f; }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < gs; j += blockDim.x) {
        const int c = g * gs + j;
        const float A = red[((size_t)n * C + c) * 2], B = red[((size_t)n * C + c) * 2 + 1];
        s1 += gamma[c] * A;
        s2 += gamma[c] * B;
        atomic_add_f32(&dbeta[c], A);
        atomic_add_f32(&dgamma[c], B);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s1s, s1); atomicAdd(&s2s, s2); }
    __syncthreads();
    const float mean = meanrstd[((size_t)n * groups + g) * 2], rstd = meanrstd[((size_t)n * groups + g) * 2 + 1];
    const float S1 = s1s * inv_count, S2 = s2s * inv_count;
    const size_t NC = (size_t)N * C;
    for (int j = threadIdx.x; j < gs; j += blockDim.x) {
        const int c = g * gs + j;
        coef[(size_t)n * C + c] = rstd * gamma[c];
        coef[NC + (size_t)n * C + c] = -rstd * rstd * S2;
        coef[2 * NC + (size_t)n * C + c] = mean * rstd * rstd * S2 - rstd * S1;
    }
}

extern "C" int mrisr_act_bwd_finalize(const float* red, const float* gamma, const float* meanrstd, float* dgamma,
                                      float* dbeta, float* coef, int N, int C, int groups, double count,
                                      const float* alpha_slots, const float* alpha, float* dalpha, float alpha_sign,
                                      void* stream) {
    if (!red || !gamma || !meanrstd || !dgamma || !dbeta || !coef) MRISR_FAIL(MRISR_E_ARG, "act_bwd_finalize: null pointer");
    if (groups <= 0 || C % groups) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_finalize: C %d groups %d", C, groups);
    if (alpha_slots && (!alpha || !dalpha)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_finalize: alpha_slots without alpha/dalpha");
    act_bwd_finalize_kernel<<<N * groups, 64, 0, (hipStream_t)stream>>>(red, gamma, meanrstd, dgamma, dbeta, coef, N, C,
                                                                       groups, (float)(1.0 / count), alpha_slots, alpha,
                                                                       dalpha, alpha_sign);
    MRISR_CHECK_LAUNCH("act_bwd_finalize");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                            const float* __restrict__ coef, T* __restrict__ dx, int N,
                                                            int H, int W, int C, int out_mode) {
    constexpr int VEC = Vec16<T>::N;
    const int nvec = C / VEC;
    const size_t NC = (size_t)N * C;
    const size_t total = (size_t)N * H * W * nvec;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cv = idx % nvec;
        const size_t pix = idx / nvec;             // n*H*W + y*W + x
        const int n = pix / ((size_t)H * W);
        const int c = cv * VEC;
        const Vec16<T> xv = load_vec16(x + pix * C + c), gv = load_vec16(g + pix * C + c);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const size_t k = (size_t)n * C + c + e;
            o.set(e, gv.get(e) * coef[k] + xv.get(e) * coef[NC + k] + coef[2 * NC + k]);
        }
        if (out_mode == MRISR_OUT_PLAIN) {
            store_vec16(dx + pix * C + c, o);
        } else {   // inverse PixelShuffle(2): (n, Y, X, c') -> (n, Y/2, X/2, 4c' + 2(Y&1) + (X&1))
            const int rem = pix - (size_t)n * H * W;
            const int Y = rem / W, X = rem - Y * W;
            T* dst = dx + (((size_t)n * (H / 2) + (Y >> 1)) * (W / 2) + (X >> 1)) * (4 * C) + 2 * (Y & 1) + (X & 1);
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[4 * (c + e)] = from_f32<T>(o.get(e));
        }
    }
}

// Pass 2 without the intermediate g tensor (plain consumers only): dL/dact is gathered from the consumers again,
// LeakyReLU' applied, then dx = g*cA + x*cB + cC.  Saves the 2-byte write of pass 1 and reads da instead of g.
// SAME: every consumer gradient has the node's own H x W and no pad offset (all but the odd-size decoder nodes): the
// consumer element is addressed by the linear pixel index - no div/mod per element.
// Block = (pixel range, image), thread = (pixel lane, 16-byte channel vector): the five per-(n,c) coefficient vectors
// are loaded ONCE per thread.  Re-loading them per element put 160 B of L1 traffic next to every 48 B of HBM traffic
// and capped the pass at ~3.7 TB/s.
template <typename T, bool SAME, bool HEAD>     // HEAD: the single consumer is the output head (MRISR_SP_HEAD)
__global__ __launch_bounds__(256) void act_bwd_apply_fused_kernel(const ActBwdParams p, const float* __restrict__ coef,
                                                                  T* __restrict__ dx, const FinDev fin) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float s12[2 * kMaxGroups];
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = min(256, p.C / VEC - (int)blockIdx.z * 256), ppb = 256 / nvec;     // blockIdx.z: channel slice
    const int cv = t % nvec, pl = t / nvec, c = blockIdx.z * 256 * VEC + cv * VEC;
    float sc[VEC], sh[VEC], ca[VEC], cb[VEC], cc[VEC];
    if (!coef) gn_bwd_coefs<VEC>(fin, s12, n, p.C, c, pl < ppb, blockIdx.x == 0, pl == 0, blockIdx.x + blockIdx.y + blockIdx.z == 0, ca, cb, cc);
    if (pl >= ppb) return;
    const size_t NC = (size_t)p.N * p.C;
    const int HW = p.H * p.W;
    float bw1 = 1.f, bw2 = 1.f;
    if (p.blend_alpha) {
        bw1 = 1.f / (1.f + __expf(-p.blend_alpha[0]));
        bw2 = 1.f - bw1;
    }
    const int m0 = p.cons[0].weight_mode, m1 = p.cons[1].weight_mode;
    const float w0 = m0 == 0 ? 1.f : (m0 == 1 ? bw1 : bw2), w1 = m1 == 0 ? 1.f : (m1 == 1 ? bw1 : bw2);
    const size_t k0 = (size_t)n * p.C + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = p.scale[k0 + e]; sh[e] = p.shift[k0 + e];
        if (coef) { ca[e] = coef[k0 + e]; cb[e] = coef[NC + k0 + e]; cc[e] = coef[2 * NC + k0 + e]; }
    }
    const T* xb = (const T*)p.x + (size_t)n * HW * p.C + c;
    T* ob = dx + (size_t)n * HW * p.C + c;
    const T* d0b = (const T*)p.cons[0].da + (size_t)n * p.cons[0].H * p.cons[0].W * p.cons[0].C_total + p.cons[0].c_off + c;
    const T* d1b = p.ncons > 1 ? (const T*)p.cons[1].da + (size_t)n * p.cons[1].H * p.cons[1].W * p.cons[1].C_total + p.cons[1].c_off + c : d0b;
    float hw[HEAD ? VEC : 1];
    if constexpr (HEAD) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) hw[e] = p.cons[0].head_w[c + e];
    }
    const int pend = min(HW, (int)(blockIdx.x + 1) * p.pix_per_block);
    for (int pix = blockIdx.x * p.pix_per_block + pl; pix < pend; pix += ppb) {
        const Vec16<T> xv = load_vec16(xb + (size_t)pix * p.C);
        float gact[VEC];
        if constexpr (HEAD) {
            const size_t gp = (size_t)n * HW + pix;
            const float o = p.cons[0].head_out[gp];
            const float dz = ((const float*)p.cons[0].da)[gp] * o * (1.f - o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) gact[e] = dz * hw[e];
        } else if constexpr (SAME) {
            const Vec16<T> d0 = load_vec16(d0b + (size_t)pix * p.cons[0].C_total);
            if (p.ncons > 1) {
                const Vec16<T> d1 = load_vec16(d1b + (size_t)pix * p.cons[1].C_total);
#pragma unroll
                for (int e = 0; e < VEC; ++e) gact[e] = w0 * d0.get(e) + w1 * d1.get(e);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) gact[e] = w0 * d0.get(e);
            }
        } else {
            const int y = pix / p.W, x = pix - y * p.W;
#pragma unroll
            for (int e = 0; e < VEC; ++e) gact[e] = 0.f;
            for (int k = 0; k < p.ncons; ++k) {
                const ConsumerDev& cs = p.cons[k];
                const int yy = y + cs.off_y, xx = x + cs.off_x;
                if (yy < cs.H && xx < cs.W) {
                    const Vec16<T> d = load_vec16((k ? d1b : d0b) + ((size_t)yy * cs.W + xx) * cs.C_total);
                    const float wgt = k ? w1 : w0;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) gact[e] += wgt * d.get(e);
                }
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xr = xv.get(e);
            const float pre = xr * sc[e] + sh[e];
            const float gy = gact[e] * (pre > 0.f ? 1.f : LRELU_SLOPE);
            o.set(e, gy * ca[e] + xr * cb[e] + cc[e]);
        }
        store_vec16(ob + (size_t)pix * p.C, o);
    }
}

// ------------------------------------------------------------------------------------------------
// ONE-PASS GroupNorm + LeakyReLU backward (plain consumers with the node's own geometry, 16-bit storage): what
// act_bwd_reduce_kernel<T,0> + act_bwd_apply_fused_kernel<T,true> do in two launches that each read x and every consumer
// gradient from HBM (5 tensor passes for one consumer, 7 for two), in one launch that reads them ONCE (3 / 4 passes):
// a block keeps its pixels' x and consumer-gradient vectors in registers across the reduction.
//   phase 1: per-(n,c) sums of g and g*xhat of the block's pixels -> atomics into red[n][c][2] (as the reduce kernel)
//   image barrier: the blocks of image n count themselves in arrive[n] and wait until all of them have
//   phase 2: coefficients from red[n] (as gn_bwd_coefs), dx from the registers
// The barrier is safe because the workgroups of a 2-D grid start in linear order (x fastest): the blocks of image n are
// all dispatched before any block of image n+1, so the ones a waiting block needs are resident or next in line - as long
// as ONE image's blocks fit on the chip together, which the host guarantees with a wide margin (<= 256 blocks of 256
// threads per image against >= 480 resident ones with the weight-gradient stream holding its share of the CUs).  A block
// that waits unreasonably long (~0.3 s) gives up and poisons its output with NaN: a broken assumption must neither hang the
// GPU nor pass silently.
constexpr int kOnePassPPT = 8;       // pixels per thread
constexpr int kOnePassMaxBlocks = 256;
// the per-(n,c) sums are spread over this many copies of red[] (block index mod kOnePassSlots): the blocks of an image all
// add into the same C x 2 words right before they wait for each other, and 256 same-address atomics in a row (performed at
// the memory side) took ~25 us per image
constexpr int kOnePassSlots = 16;
constexpr int kOnePassGroups = 16;      // block groups of the image barrier
constexpr int kOnePassWords = 544;      // barrier words per image: 16 group counters, the image counter, 16 group flags - 64 bytes apart

__device__ __forceinline__ float ld_coherent(const float* p) {      // red[] is written by other blocks of THIS launch
    return __hip_atomic_load((const GLOBAL_AS float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The part every one-pass kernel shares.  In: this thread's sums of g (sA) and g*x (sB) over its pixels, channels c .. c+VEC-1 of
// image n.  Block reduction -> atomics into this block's copy of red[] -> image barrier -> coefficients of dx = g cA + x cB + cC
// for the thread's channels (act_bwd_finalize_kernel's formulas), dgamma / dbeta added once per image.  Every thread of the block
// calls it.  lds: 256 * VEC * 2 + 2 * kMaxGroups floats.
template <int VEC>
__device__ __forceinline__ void onepass_meet(const ActBwdParams& p, const FinDev& fin, unsigned* __restrict__ arrive, float* lds,
                                             int* timed_out, int n, int c, int nvec, int ppb, int pl, float* sA, float* sB,
                                             float* ca, float* cb, float* cc) {
    const int t = threadIdx.x, gs = p.C / p.groups;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {      // sum g*xhat = rstd * (sum g*x - mean * sum g), per thread (8 pixels)
        const int g = (c + e) / gs;
        const float mean = p.meanrstd[((size_t)n * p.groups + g) * 2], rstd = p.meanrstd[((size_t)n * p.groups + g) * 2 + 1];
        sB[e] = rstd * (sB[e] - mean * sA[e]);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        lds[(t * VEC + e) * 2] = sA[e];
        lds[(t * VEC + e) * 2 + 1] = sB[e];
    }
    if (t == 0) *timed_out = 0;
    __syncthreads();
    for (int i = t; i < nvec * VEC * 2; i += 256) {
        const int j = i >> 1, which = i & 1;
        const int cvj = j / VEC, e = j - cvj * VEC;
        float s = 0.f;
        for (int q = 0; q < ppb; ++q) s += lds[((q * nvec + cvj) * VEC + e) * 2 + which];
        atomic_add_f32(&p.red[(((size_t)(blockIdx.x % kOnePassSlots) * p.N + n) * p.C + cvj * VEC + e) * 2 + which], s);
    }
    // ---- image barrier.  Everything that crosses it goes through agent-scope RELAXED atomics (the sums above, the counter,
    // the reads of red[] below), which are performed at the memory side: no release / acquire fences - on this multi-XCD part
    // each of them writes back or invalidates a whole L2, and with 4096 blocks doing so the launch ran 8x slower than the
    // two launches it replaces.  `s_waitcnt vmcnt(0)`: this thread's atomics have been acknowledged before it is counted.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        // Two-level count, then a release flag per group of blocks: 256 blocks bumping and then polling ONE word is a chain of
        // 256 same-address atomics at the memory side (~40 us per image, measured).  Block b belongs to group b % 16: it bumps
        // its group's counter; the block that completes a group bumps the image's counter; the block that completes the image
        // sets the 16 group flags; everybody polls its own group's flag (16 pollers per word, one cache line per word).
        unsigned* img = arrive + (size_t)n * kOnePassWords;
        const unsigned nblk = gridDim.x, g = blockIdx.x % kOnePassGroups;
        const unsigned ngroups = min(nblk, (unsigned)kOnePassGroups), members = (nblk - g + kOnePassGroups - 1) / kOnePassGroups;
        if (__hip_atomic_fetch_add(img + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == members) {
            if (__hip_atomic_fetch_add(img + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == ngroups) {
                for (unsigned k = 0; k < ngroups; ++k)
                    __hip_atomic_store(img + 272 + k * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int spin = 0;
        while (__hip_atomic_load(img + 272 + g * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(16);      // ~0.5 us between polls
            if (++spin > 600000) { *timed_out = 1; break; }
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    const bool bad = *timed_out != 0;

    // ---- coefficients of image n (act_bwd_finalize_kernel's formulas; red read around the L1)
    float* s12 = lds;
    if (t < 2 * kMaxGroups) s12[t] = 0.f;
    __syncthreads();
    const float* rn = fin.red + (size_t)n * p.C * 2;
    const size_t slot_stride = (size_t)p.N * p.C * 2;
    float* sums = lds + 2 * kMaxGroups;            // [C][2]: the image's per-channel sums, all slots added up
    for (int j = t; j < 2 * p.C; j += 256) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < kOnePassSlots; ++sl) v += ld_coherent(rn + sl * slot_stride + j);
        sums[j] = v;
        atomicAdd(&s12[(j & 1) * kMaxGroups + (j >> 1) / gs], fin.gamma[j >> 1] * v);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = (c + e) / gs;
        const float mean = fin.meanrstd[((size_t)n * fin.groups + g) * 2], rstd = fin.meanrstd[((size_t)n * fin.groups + g) * 2 + 1];
        const float S1 = s12[g] * fin.inv_count, S2 = s12[kMaxGroups + g] * fin.inv_count;
        ca[e] = rstd * fin.gamma[c + e];
        cb[e] = -rstd * rstd * S2;
        cc[e] = bad ? __builtin_nanf("") : mean * rstd * rstd * S2 - rstd * S1;
        if (blockIdx.x == 0 && pl == 0) {          // one block per image adds the image's share of dgamma / dbeta
            atomic_add_f32(&fin.dbeta[c + e], sums[2 * (c + e)]);
            atomic_add_f32(&fin.dgamma[c + e], sums[2 * (c + e) + 1]);
        }
    }
}

template <typename T, int NCONS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void act_bwd_onepass_kernel(const ActBwdParams p, T* __restrict__ dx, const FinDev fin,
                                                              unsigned* __restrict__ arrive) {
    constexpr int VEC = Vec16<T>::N, PPT = kOnePassPPT;
    __shared__ float lds[256 * VEC * 2 + 2 * kMaxGroups];
    __shared__ int timed_out;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = p.C / VEC, ppb = 256 / nvec;             // host-checked: nvec is a power of two <= 256
    const int cv = t & (nvec - 1), pl = t / nvec, c = cv * VEC;
    const int HW = p.H * p.W, gs = p.C / p.groups;
    const int pix0 = blockIdx.x * (ppb * PPT) + pl;

    float sc[VEC], sh[VEC];
    const size_t k0 = (size_t)n * p.C + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = p.scale[k0 + e]; sh[e] = p.shift[k0 + e]; }
    const T* xb = (const T*)p.x + (size_t)n * HW * p.C + c;
    const T* d0b = (const T*)p.cons[0].da + (size_t)n * HW * p.cons[0].C_total + p.cons[0].c_off + c;
    const T* d1b = NCONS > 1 ? (const T*)p.cons[1].da + (size_t)n * HW * p.cons[1].C_total + p.cons[1].c_off + c : d0b;
    const int ct0 = p.cons[0].C_total, ct1 = NCONS > 1 ? p.cons[1].C_total : ct0;

    // ---- phase 1: everything this thread will need, loaded once
    Vec16<T> xv[PPT], d0[PPT], d1[NCONS > 1 ? PPT : 1];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pix = min(pix0 + i * ppb, HW - 1);          // (the tail block re-reads the last pixel; masked below)
        xv[i] = load_vec16(xb + (size_t)pix * p.C);
        d0[i] = load_vec16(d0b + (size_t)pix * ct0);
        if (NCONS > 1) d1[i] = load_vec16(d1b + (size_t)pix * ct1);
    }
    float sA[VEC], sB[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sA[e] = 0.f; sB[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        if (pix0 + i * ppb < HW) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xr = xv[i].get(e);
                const float pre = xr * sc[e] + sh[e];
                const float ga = NCONS > 1 ? d0[i].get(e) + d1[i].get(e) : d0[i].get(e);
                const float gy = ga * (pre > 0.f ? 1.f : LRELU_SLOPE);
                sA[e] += gy;
                sB[e] += gy * xr;
            }
        }
    }
    float ca[VEC], cb[VEC], cc[VEC];
    onepass_meet<VEC>(p, fin, arrive, lds, &timed_out, n, c, nvec, ppb, pl, sA, sB, ca, cb, cc);
    T* ob = dx + (size_t)n * HW * p.C + c;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pix = pix0 + i * ppb;
        if (pix < HW) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xr = xv[i].get(e);
                const float pre = xr * sc[e] + sh[e];
                const float ga = NCONS > 1 ? d0[i].get(e) + d1[i].get(e) : d0[i].get(e);
                const float gy = ga * (pre > 0.f ? 1.f : LRELU_SLOPE);
                o.set(e, gy * ca[e] + xr * cb[e] + cc[e]);
            }
            store_vec16(ob + (size_t)pix * p.C, o);
        }
    }
}

// The same for a node whose activation is ALSO max-pooled (the encoder skips: one 2x2-pool consumer + at most one plain
// consumer of the node's geometry; even H, W).  Thread = (pooled pixel, channel vector) owning whole 2x2 windows as in
// act_bwd_pool_window_kernel: two windows per thread = 8 pixels' x, their plain-consumer gradients and two pooled gradients
// in registers across the barrier; the arg-max is recomputed in phase 2 (a few vector instructions per element).
constexpr int kOnePassWPT = 2;       // windows per thread
template <typename T, bool HAS_O>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void act_bwd_onepass_window_kernel(const ActBwdParams p, T* __restrict__ dx,
                                                                                                                  const FinDev fin, unsigned* __restrict__ arrive) {
    constexpr int VEC = Vec16<T>::N, WPT = kOnePassWPT;
    __shared__ float lds[256 * VEC * 2 + 2 * kMaxGroups];
    __shared__ int timed_out;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = p.C / VEC, ppb = 256 / nvec;
    const int cv = t & (nvec - 1), pl = t / nvec, c = cv * VEC;
    const int W = p.W, Hp = p.H / 2, Wp = W / 2, HWp = Hp * Wp;
    const int kp = p.cons[0].spatial == MRISR_SP_POOL2 ? 0 : 1, ko = 1 - kp;
    const int w0 = blockIdx.x * (ppb * WPT) + pl;
    float sc[VEC], sh[VEC];
    const size_t k0 = (size_t)n * p.C + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = p.scale[k0 + e]; sh[e] = p.shift[k0 + e]; }
    const T* xb = (const T*)p.x + (size_t)n * p.H * W * p.C + c;
    const T* dpb = (const T*)p.cons[kp].da + (size_t)n * HWp * p.cons[kp].C_total + p.cons[kp].c_off + c;
    const int Cto = HAS_O ? p.cons[ko].C_total : 0;
    const T* dob = HAS_O ? (const T*)p.cons[ko].da + (size_t)n * p.H * W * Cto + p.cons[ko].c_off + c : xb;

    Vec16<T> xv[WPT][4], dv[HAS_O ? WPT : 1][4], dp[WPT];
    unsigned off[WPT];                  // pixel index of the window's top-left corner
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int pp = min(w0 + i * ppb, HWp - 1);              // (the tail block re-reads the last window; masked below)
        const int py = pp / Wp, px = pp - py * Wp;
        off[i] = (unsigned)(2 * py) * W + 2 * px;
        const unsigned o4[4] = {off[i], off[i] + 1, off[i] + (unsigned)W, off[i] + (unsigned)W + 1};
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[i][q] = load_vec16(xb + (size_t)o4[q] * p.C);
        dp[i] = load_vec16(dpb + (size_t)pp * p.cons[kp].C_total);
        if (HAS_O) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dv[i][q] = load_vec16(dob + (size_t)o4[q] * Cto);
        }
    }
    // gradient w.r.t. the GroupNorm output at the four pixels of window i, channel e (act_bwd_pool_window_kernel's arithmetic:
    // first maximum of the activations in scan order takes the pooled gradient)
    auto window_g = [&](int i, int e, float* xr, float* gy) {
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { xr[q] = xv[i][q].get(e); pre[q] = xr[q] * sc[e] + sh[e]; }
        int win = 0;
        float m = lrelu(pre[0]);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float a = lrelu(pre[q]);
            if (a > m) { m = a; win = q; }
        }
        const float gp = dp[i].get(e);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gact = q == win ? gp : 0.f;
            if (HAS_O) gact += dv[i][q].get(e);
            gy[q] = gact * (pre[q] > 0.f ? 1.f : LRELU_SLOPE);
        }
    };
    float sA[VEC], sB[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sA[e] = 0.f; sB[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        if (w0 + i * ppb < HWp) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float xr[4], gy[4];
                window_g(i, e, xr, gy);
#pragma unroll
                for (int q = 0; q < 4; ++q) { sA[e] += gy[q]; sB[e] += gy[q] * xr[q]; }
            }
        }
    }
    float ca[VEC], cb[VEC], cc[VEC];
    onepass_meet<VEC>(p, fin, arrive, lds, &timed_out, n, c, nvec, ppb, pl, sA, sB, ca, cb, cc);
    T* ob = dx + (size_t)n * p.H * W * p.C + c;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        if (w0 + i * ppb < HWp) {
            Vec16<T> ov[4];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float xr[4], gy[4];
                window_g(i, e, xr, gy);
#pragma unroll
                for (int q = 0; q < 4; ++q) ov[q].set(e, gy[q] * ca[e] + xr[q] * cb[e] + cc[e]);
            }
            const unsigned o4[4] = {off[i], off[i] + 1, off[i] + (unsigned)W, off[i] + (unsigned)W + 1};
#pragma unroll
            for (int q = 0; q < 4; ++q) store_vec16(ob + (size_t)o4[q] * p.C, ov[q]);
        }
    }
}

// Does a node qualify for the one-pass kernel?  (16-bit storage, plain consumers of the node's own geometry without blend
// weights, C / 8 a power of two <= 256, at most kOnePassMaxBlocks blocks per image)
// 0 = no, 1 = plain consumers (act_bwd_onepass_kernel), 2 = one 2x2-pool consumer (+ at most one plain one): window kernel
extern "C" int mrisr_act_bwd_onepass_ok(int dtype, int nconsumers, const mrisr_consumer* consumers, int N, int H, int W, int C) {
    if (dtype != MRISR_BF16 && dtype != MRISR_F16) return 0;
    if (!consumers || nconsumers < 1 || nconsumers > 2 || N < 1) return 0;
    const int vec = mrisr_vec(dtype);
    if (C % vec) return 0;
    const int nvec = C / vec;
    if (nvec > 256 || (nvec & (nvec - 1))) return 0;
    const long HW = (long)H * W;
    if (HW * C >= (1l << 30)) return 0;
    bool plain = true;
    for (int k = 0; k < nconsumers; ++k) {
        const mrisr_consumer& c = consumers[k];
        if (c.weight_mode) return 0;
        if (c.spatial != MRISR_SP_NONE || c.H != H || c.W != W || c.off_y || c.off_x) plain = false;
    }
    if (plain) {
        const int ppblk = (256 / nvec) * kOnePassPPT;
        return (HW + ppblk - 1) / ppblk <= kOnePassMaxBlocks ? 1 : 0;
    }
    if (!act_bwd_window_ok(nconsumers, consumers, H, W)) return 0;
    for (int k = 0; k < nconsumers; ++k)
        if (consumers[k].spatial == MRISR_SP_POOL2 && (consumers[k].H != H / 2 || consumers[k].W != W / 2)) return 0;
    const int wpblk = (256 / nvec) * kOnePassWPT;
    return (HW / 4 + wpblk - 1) / wpblk <= kOnePassMaxBlocks ? 2 : 0;
}

// red: [kOnePassSlots = 16][N][C][2] floats and arrive: [N][kOnePassWords = 544] barrier words, both ZERO on entry (the engine's per-backward arena).
extern "C" int mrisr_act_bwd_onepass_slots(void) { return kOnePassSlots; }
extern "C" int mrisr_act_bwd_onepass_barrier_words(void) { return kOnePassWords; }
extern "C" int mrisr_act_bwd_onepass(int dtype, const void* x, const float* scale, const float* shift, const float* meanrstd,
                                     int nconsumers, const mrisr_consumer* consumers, float* red, unsigned* arrive,
                                     const mrisr_gn_bwd_fin* fin, void* dx, int N, int H, int W, int C, void* stream) {
    if (!x || !scale || !shift || !meanrstd || !consumers || !red || !arrive || !fin || !dx) MRISR_FAIL(MRISR_E_ARG, "act_bwd_onepass: null pointer");
    if (!mrisr_act_bwd_onepass_ok(dtype, nconsumers, consumers, N, H, W, C))
        MRISR_FAIL(MRISR_E_UNSUPPORTED, "act_bwd_onepass: node does not qualify (mrisr_act_bwd_onepass_ok)");
    if (fin->red != red || !fin->gamma || !fin->meanrstd || !fin->dgamma || !fin->dbeta) MRISR_FAIL(MRISR_E_ARG, "act_bwd_onepass: fin");
    if (fin->groups <= 0 || fin->groups > kMaxGroups || C % fin->groups || !(fin->count > 0)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_onepass: fin groups %d", fin->groups);
    if (fin->alpha_slots) MRISR_FAIL(MRISR_E_UNSUPPORTED, "act_bwd_onepass: blend branches take the two-pass kernels");
    FinDev fd;
    memset(&fd, 0, sizeof(fd));
    fd = FinDev{fin->red, fin->gamma, fin->meanrstd, fin->dgamma, fin->dbeta, nullptr, nullptr, nullptr,
                (float)(1.0 / fin->count), fin->alpha_sign, fin->groups, nullptr, nullptr, nullptr};
    ActBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.scale = scale; p.shift = shift; p.meanrstd = meanrstd; p.red = red;
    p.ncons = nconsumers; p.N = N; p.H = H; p.W = W; p.C = C; p.groups = fin->groups;
    int rc = fill_act_bwd_params(p, dtype, nconsumers, consumers, nullptr, H, W, C, "act_bwd_onepass");
    if (rc) return rc;
    const int kind = mrisr_act_bwd_onepass_ok(dtype, nconsumers, consumers, N, H, W, C);
    const int nvec = C / mrisr_vec(dtype);
    hipStream_t s = (hipStream_t)stream;
    if (kind == 2) {
        const bool has_o = nconsumers > 1;
        dim3 grid(ceil_div(H * W / 4, (256 / nvec) * kOnePassWPT), N);
        if (dtype == MRISR_BF16) {
            if (has_o) act_bwd_onepass_window_kernel<bf16_t, true><<<grid, 256, 0, s>>>(p, (bf16_t*)dx, fd, arrive);
            else act_bwd_onepass_window_kernel<bf16_t, false><<<grid, 256, 0, s>>>(p, (bf16_t*)dx, fd, arrive);
        } else {
            if (has_o) act_bwd_onepass_window_kernel<f16_t, true><<<grid, 256, 0, s>>>(p, (f16_t*)dx, fd, arrive);
            else act_bwd_onepass_window_kernel<f16_t, false><<<grid, 256, 0, s>>>(p, (f16_t*)dx, fd, arrive);
        }
        MRISR_CHECK_LAUNCH("act_bwd_onepass(window)");
        return MRISR_OK;
    }
    dim3 grid(ceil_div(H * W, (256 / nvec) * kOnePassPPT), N);
    if (dtype == MRISR_BF16) {
        if (nconsumers == 1) act_bwd_onepass_kernel<bf16_t, 1><<<grid, 256, 0, s>>>(p, (bf16_t*)dx, fd, arrive);
        else act_bwd_onepass_kernel<bf16_t, 2><<<grid, 256, 0, s>>>(p, (bf16_t*)dx, fd, arrive);
    } else {
        if (nconsumers == 1) act_bwd_onepass_kernel<f16_t, 1><<<grid, 256, 0, s>>>(p, (f16_t*)dx, fd, arrive);
        else act_bwd_onepass_kernel<f16_t, 2><<<grid, 256, 0, s>>>(p, (f16_t*)dx, fd, arrive);
    }
    MRISR_CHECK_LAUNCH("act_bwd_onepass");
    return MRISR_OK;
}

// out_mode PIXEL_SHUFFLE2: dx is stored un-shuffled, [N][H/2][W/2][4C] with channel 4c + 2(Y&1) + (X&1).  Thread =
// one 16-byte vector of the DESTINATION (VEC/4 source channels x the 2x2 source pixels), so the stores are full
// vectors; the (n, Y, X, c) -> destination scatter of 2-byte elements ran at a third of the plain kernel's rate.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_apply_unshuffle_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                                      const float* __restrict__ coef, T* __restrict__ dx,
                                                                      int N, int H, int W, int C, float* __restrict__ dbias) {
    constexpr int VEC = Vec16<T>::N, NSC = VEC / 4;      // source channels per thread
    __shared__ float bsum[256 * VEC];
    float bs[VEC];                                       // per-thread channel sums of dx (= the conv's bias gradient)
#pragma unroll
    for (int e = 0; e < VEC; ++e) bs[e] = 0.f;
    const int H2 = H / 2, W2 = W / 2, ndv = 4 * C / VEC;
    const size_t NC = (size_t)N * C;
    const size_t total = (size_t)N * H2 * W2 * ndv;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int dv = idx % ndv;
        size_t r = idx / ndv;
        const int X2 = r % W2; r /= W2;
        const int Y2 = r % H2;
        const int n = r / H2;
        const int c0 = dv * NSC;
        Vec16<T> o;
#pragma unroll
        for (int sc = 0; sc < NSC; ++sc) {
            const size_t k = (size_t)n * C + c0 + sc;
            const float cA = coef[k], cB = coef[NC + k], cC = coef[2 * NC + k];
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) {
                const size_t sp = (((size_t)n * H + 2 * Y2 + (ij >> 1)) * W + 2 * X2 + (ij & 1)) * C + c0 + sc;
                o.set(4 * sc + ij, to_f32(g[sp]) * cA + to_f32(x[sp]) * cB + cC);
            }
        }
        store_vec16(dx + idx * VEC, o);
        if (dbias) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) bs[e] += o.get(e);     // (the rounded value the weight-gradient kernel will read)
        }
    }
    if (dbias) {
        // the grid stride is a multiple of ndv (host-checked), so a thread keeps its destination vector dv = t % ndv
        const int t = threadIdx.x;
#pragma unroll
        for (int e = 0; e < VEC; ++e) bsum[t * VEC + e] = bs[e];
        __syncthreads();
        for (int i = t; i < ndv * VEC; i += 256) {
            const int dv = i / VEC, e = i - dv * VEC;
            float v = 0.f;
            for (int q = dv; q < 256; q += ndv) v += bsum[q * VEC + e];
            atomic_add_f32(&dbias[dv * VEC + e], v);
        }
    }
}

// Pass 2 of a pixel-shuffled node without the intermediate g tensor: dx is stored un-shuffled, [N][H/2][W/2][4C] with
// channel 4c + 2(Y&1) + (X&1) (the producing conv's own output layout).  Thread = (2x2 window of the shuffled image,
// 16-byte channel vector): eight full-vector loads (x and the consumer's gradient at the four pixels) and 4*VEC
// consecutive destination channels = one contiguous 4*16-byte store.  dbias (optional, [4C]) += channel sums of dx.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_unshuffle_window_kernel(const ActBwdParams p, T* __restrict__ dx,
                                                                       float* __restrict__ dbias, const FinDev fin) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float lds[256 * 4 * VEC];
    const int t = threadIdx.x, n = blockIdx.y;
    const int cz = blockIdx.z * 256 * VEC;
    const int nvec = min(256, p.C / VEC - (int)blockIdx.z * 256), ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cz + cv * VEC;
    const bool active = pl < ppb;
    const int Hp = p.H / 2, Wp = p.W / 2, HWp = Hp * Wp, W = p.W;
    float sc[VEC], sh[VEC], ca[VEC], cb[VEC], cc[VEC], bs[4 * VEC];
    gn_bwd_coefs<VEC>(fin, lds, n, p.C, c, active, blockIdx.x == 0, pl == 0, blockIdx.x + blockIdx.y + blockIdx.z == 0, ca, cb, cc);
    float w0 = 1.f;
    if (p.blend_alpha) {
        const float a = 1.f / (1.f + __expf(-p.blend_alpha[0]));
        w0 = p.cons[0].weight_mode == 0 ? 1.f : (p.cons[0].weight_mode == 1 ? a : 1.f - a);
    }
    const size_t k0 = (size_t)n * p.C + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = active ? p.scale[k0 + e] : 0.f;
        sh[e] = active ? p.shift[k0 + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4 * VEC; ++i) bs[i] = 0.f;
    const T* xb = (const T*)p.x + (size_t)n * p.H * W * p.C + c;
    const T* db = (const T*)p.cons[0].da + (size_t)n * p.H * W * p.cons[0].C_total + p.cons[0].c_off + c;
    const int Ct = p.cons[0].C_total;
    T* ob = dx + (size_t)n * HWp * 4 * p.C + 4 * c;
    const int pend = min(HWp, (int)(blockIdx.x + 1) * p.pix_per_block);
    if (active) {
        for (int pp = blockIdx.x * p.pix_per_block + pl; pp < pend; pp += ppb) {
            const int py = pp / Wp, px = pp - py * Wp;
            const size_t b0 = (size_t)(2 * py) * W + 2 * px;
            const size_t off[4] = {b0, b0 + 1, b0 + W, b0 + W + 1};
            Vec16<T> xv[4], dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xv[q] = load_vec16(xb + off[q] * p.C); dv[q] = load_vec16(db + off[q] * Ct); }
            Vec16<T> ov[4];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xr = xv[q].get(e);
                    const float pre = xr * sc[e] + sh[e];
                    const float gy = w0 * dv[q].get(e) * (pre > 0.f ? 1.f : LRELU_SLOPE);
                    const int i = 4 * e + q;          // destination channel 4(c + e) + q
                    ov[i / VEC].set(i % VEC, gy * ca[e] + xr * cb[e] + cc[e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) store_vec16(ob + (size_t)pp * 4 * p.C + j * VEC, ov[j]);
            if (dbias) {
#pragma unroll
                for (int i = 0; i < 4 * VEC; ++i) bs[i] += ov[i / VEC].get(i % VEC);     // (the rounded value the weight-gradient kernel will read)
            }
        }
    }
    if (dbias) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4 * VEC; ++i) lds[t * 4 * VEC + i] = active ? bs[i] : 0.f;
        __syncthreads();
        for (int j = t; j < nvec * 4 * VEC; j += 256) {
            const int cvj = j / (4 * VEC), i = j - cvj * 4 * VEC;
            float v = 0.f;
            for (int q = 0; q < ppb; ++q) v += lds[(q * nvec + cvj) * 4 * VEC + i];
            atomic_add_f32(&dbias[4 * (cz + cvj * VEC) + i], v);
        }
    }
}

extern "C" int mrisr_act_bwd_apply_fused_unshuffle(int dtype, const void* x, const float* scale, const float* shift,
                                                   const mrisr_consumer* consumer, const float* blend_alpha,
                                                   const mrisr_gn_bwd_fin* fin, void* dx, float* dbias, int N, int H, int W,
                                                   int C, void* stream) {
    if (!x || !scale || !shift || !dx || !consumer || !fin) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused_unshuffle: null pointer");
    if (!fin->red || !fin->gamma || !fin->meanrstd || !fin->dgamma || !fin->dbeta) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused_unshuffle: fin null pointer");
    if (fin->groups <= 0 || fin->groups > kMaxGroups || C % fin->groups || !(fin->count > 0)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply_fused_unshuffle: fin groups %d", fin->groups);
    if (fin->alpha_slots && (!fin->alpha || !fin->dalpha)) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply_fused_unshuffle: alpha_slots without alpha/dalpha");
    const int vec = mrisr_vec(dtype);
    if (C % vec || ((H | W) & 1)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply_fused_unshuffle: C %d, H %d, W %d (even dims)", C, H, W);
    if (consumer->spatial != MRISR_SP_NONE || consumer->H != H || consumer->W != W || consumer->off_y || consumer->off_x)
        MRISR_FAIL(MRISR_E_UNSUPPORTED, "act_bwd_apply_fused_unshuffle: one plain consumer with the node's geometry");
    ActBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.scale = scale; p.shift = shift; p.blend_alpha = blend_alpha;
    p.ncons = 1; p.N = N; p.H = H; p.W = W; p.C = C;
    int rc = fill_act_bwd_params(p, dtype, 1, consumer, blend_alpha, H, W, C, "act_bwd_apply_fused_unshuffle");
    if (rc) return rc;
    const FinDev fd{fin->red, fin->gamma, fin->meanrstd, fin->dgamma, fin->dbeta, fin->alpha_slots, fin->alpha, fin->dalpha,
                    (float)(1.0 / fin->count), fin->alpha_sign, fin->groups, nullptr, nullptr, nullptr};
    const int nslice = ceil_div(C / vec, 256), nvs = nslice > 1 ? 256 : C / vec;
    const int ppbw = 256 / nvs, HWp = (H / 2) * (W / 2);
    // every block ends in 4C same-row float atomics when the bias gradient is asked for: few, long-running blocks
    int ppw = ppbw * 16;
    if (ppw > HWp) ppw = ceil_div(HWp, ppbw) * ppbw;
    p.pix_per_block = ppw;
    dim3 grid(ceil_div(HWp, ppw), N, nslice);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) act_bwd_unshuffle_window_kernel<bf16_t><<<grid, 256, 0, s>>>(p, (bf16_t*)dx, dbias, fd);
    else if (dtype == MRISR_F16) act_bwd_unshuffle_window_kernel<f16_t><<<grid, 256, 0, s>>>(p, (f16_t*)dx, dbias, fd);
    else if (dtype == MRISR_F32) act_bwd_unshuffle_window_kernel<float><<<grid, 256, 0, s>>>(p, (float*)dx, dbias, fd);
    else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_apply_fused_unshuffle: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("act_bwd_apply_fused_unshuffle");
    return MRISR_OK;
}

extern "C" int mrisr_act_bwd_apply(int dtype, const void* x, const void* g, const float* coef, void* dx, int N, int H,
                                   int W, int C, int out_mode, float* dbias, void* stream) {
    if (!x || !g || !coef || !dx) MRISR_FAIL(MRISR_E_ARG, "act_bwd_apply: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply: C %d", C);
    if (out_mode == MRISR_OUT_PIXEL_SHUFFLE2 && ((H | W) & 1)) MRISR_FAIL(MRISR_E_SHAPE, "act_bwd_apply: odd dims with pixel shuffle");
    const size_t total = (size_t)N * H * W * (C / vec);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dbias && !(out_mode == MRISR_OUT_PIXEL_SHUFFLE2 && (4 * C) % vec == 0 && 256 % (4 * C / vec) == 0))
        MRISR_FAIL(MRISR_E_UNSUPPORTED, "act_bwd_apply: dbias needs the pixel-shuffle output with 4C/vec dividing 256");
    if (out_mode == MRISR_OUT_PIXEL_SHUFFLE2 && (4 * C) % vec == 0) {
        // with the bias gradient every block ends in 4C same-address atomics: 8192 blocks made that tail as long as
        // the pass itself (measured 298 vs 144 us) -> fewer, longer-running blocks
        const int blocks = dbias ? (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024) : (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        if (dtype == MRISR_BF16)
            act_bwd_apply_unshuffle_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)g, coef, (bf16_t*)dx, N, H, W, C, dbias);
        else if (dtype == MRISR_F16)
            act_bwd_apply_unshuffle_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, (const f16_t*)g, coef, (f16_t*)dx, N, H, W, C, dbias);
        else if (dtype == MRISR_F32)
            act_bwd_apply_unshuffle_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x, (const float*)g, coef, (float*)dx, N, H, W, C, dbias);
        else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_apply: dtype %d", dtype);
        MRISR_CHECK_LAUNCH("act_bwd_apply");
        return MRISR_OK;
    }
    if (dtype == MRISR_BF16)
        act_bwd_apply_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)g, coef, (bf16_t*)dx, N, H, W, C, out_mode);
    else if (dtype == MRISR_F16)
        act_bwd_apply_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, (const f16_t*)g, coef, (f16_t*)dx, N, H, W, C, out_mode);
    else if (dtype == MRISR_F32)
        act_bwd_apply_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x, (const float*)g, coef, (float*)dx, N, H, W, C, out_mode);
    else MRISR_FAIL(MRISR_E_DTYPE, "act_bwd_apply: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("act_bwd_apply");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// out = sigmoid(alpha) * act0 + (1 - sigmoid(alpha)) * act1, act_i = LeakyReLU(x_i*scale_i+shift_i)  (unet_model.py:206-207):
// the blended input of final_conv.0, materialised.  With 32 channels the conv and its weight gradient are
// staging-bound, and the two-source blending loader doubles that staging; one plain tensor halves it.
// Block = (pixel range, image), thread = (pixel lane, 16-byte channel vector): the four per-(n,c) coefficient vectors are
// loaded once per thread and all index arithmetic is 32-bit.
template <typename T>
__global__ __launch_bounds__(256) void norm_blend_kernel(const T* __restrict__ x0, const float* __restrict__ sc0,
                                                         const float* __restrict__ sh0, const T* __restrict__ x1,
                                                         const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                         const float* __restrict__ alpha, T* __restrict__ out, int HW,
                                                         int C, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = C / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    if (pl >= ppb) return;
    const float a = 1.f / (1.f + __expf(-alpha[0])), b = 1.f - a;
    float s0[VEC], t0[VEC], s1[VEC], t1[VEC];
    const size_t k0 = (size_t)n * C + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s0[e] = sc0[k0 + e]; t0[e] = sh0[k0 + e]; s1[e] = sc1[k0 + e]; t1[e] = sh1[k0 + e]; }
    const size_t base = (size_t)n * HW * C + c;
    const int pend = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    for (int pix = blockIdx.x * pix_per_block + pl; pix < pend; pix += ppb) {
        const size_t o_ = base + (size_t)pix * C;
        const Vec16<T> v0 = load_vec16(x0 + o_), v1 = load_vec16(x1 + o_);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o.set(e, a * lrelu(v0.get(e) * s0[e] + t0[e]) + b * lrelu(v1.get(e) * s1[e] + t1[e]));
        store_vec16(out + o_, o);
    }
}

extern "C" int mrisr_norm_blend(int dtype, const void* x0, const float* scale0, const float* shift0, const void* x1,
                                const float* scale1, const float* shift1, const float* alpha, void* out, int N, int H,
                                int W, int C, void* stream) {
    if (!x0 || !scale0 || !shift0 || !x1 || !scale1 || !shift1 || !alpha || !out) MRISR_FAIL(MRISR_E_ARG, "norm_blend: null pointer");
    const int vec = mrisr_vec(dtype);
    if (N <= 0 || N > 65535 || H <= 0 || W <= 0 || C % vec || C / vec > 256) MRISR_FAIL(MRISR_E_SHAPE, "norm_blend: N %d H %d W %d C %d", N, H, W, C);
    if ((size_t)H * W >= (1u << 30)) MRISR_FAIL(MRISR_E_SHAPE, "norm_blend: image %dx%d", H, W);
    const int nvec = C / vec, ppb = 256 / nvec, HW = H * W;
    int ppblk = ppb * 32;
    if (ppblk > HW) ppblk = ceil_div(HW, ppb) * ppb;
    dim3 grid(ceil_div(HW, ppblk), N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) norm_blend_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x0, scale0, shift0, (const bf16_t*)x1, scale1, shift1, alpha, (bf16_t*)out, HW, C, ppblk);
    else if (dtype == MRISR_F16) norm_blend_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)x0, scale0, shift0, (const f16_t*)x1, scale1, shift1, alpha, (f16_t*)out, HW, C, ppblk);
    else if (dtype == MRISR_F32) norm_blend_kernel<float><<<grid, 256, 0, s>>>((const float*)x0, scale0, shift0, (const float*)x1, scale1, shift1, alpha, (float*)out, HW, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "norm_blend: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("norm_blend");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// dalpha += sigmoid'(alpha) * sum da * (act0 - act1)        (unet_model.py:206-207)
template <typename T>
__global__ __launch_bounds__(256) void blend_alpha_grad_kernel(const T* __restrict__ da, const T* __restrict__ x0,
                                                               const float* __restrict__ sc0, const float* __restrict__ sh0,
                                                               const T* __restrict__ x1, const float* __restrict__ sc1,
                                                               const float* __restrict__ sh1, const float* __restrict__ alpha,
                                                               float* __restrict__ dalpha, int N, int HW, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int nvec = C / VEC;
    const size_t total = (size_t)N * HW * nvec;
    float s = 0.f;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cv = idx % nvec;
        const size_t pix = idx / nvec;
        const int n = pix / HW, c = cv * VEC;
        const Vec16<T> d = load_vec16(da + pix * C + c), a = load_vec16(x0 + pix * C + c), b = load_vec16(x1 + pix * C + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const size_t k = (size_t)n * C + c + e;
            s += d.get(e) * (lrelu(a.get(e) * sc0[k] + sh0[k]) - lrelu(b.get(e) * sc1[k] + sh1[k]));
        }
    }
    __shared__ float part[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sg = 1.f / (1.f + __expf(-alpha[0]));
        atomic_add_f32(dalpha, (part[0] + part[1] + part[2] + part[3]) * sg * (1.f - sg));
    }
}

extern "C" int mrisr_blend_alpha_grad(int dtype, const void* da, const void* x0, const float* scale0,
                                      const float* shift0, const void* x1, const float* scale1, const float* shift1,
                                      const float* alpha, float* dalpha, int N, int H, int W, int C, void* stream) {
    if (!da || !x0 || !x1 || !scale0 || !shift0 || !scale1 || !shift1 || !alpha || !dalpha) MRISR_FAIL(MRISR_E_ARG, "blend_alpha_grad: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec) MRISR_FAIL(MRISR_E_SHAPE, "blend_alpha_grad: C %d", C);
    const size_t total = (size_t)N * H * W * (C / vec);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dtype == MRISR_BF16)
        blend_alpha_grad_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)da, (const bf16_t*)x0, scale0, shift0, (const bf16_t*)x1, scale1, shift1, alpha, dalpha, N, H * W, C);
    else if (dtype == MRISR_F16)
        blend_alpha_grad_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)da, (const f16_t*)x0, scale0, shift0, (const f16_t*)x1, scale1, shift1, alpha, dalpha, N, H * W, C);
    else if (dtype == MRISR_F32)
        blend_alpha_grad_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)da, (const float*)x0, scale0, shift0, (const float*)x1, scale1, shift1, alpha, dalpha, N, H * W, C);
    else MRISR_FAIL(MRISR_E_DTYPE, "blend_alpha_grad: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("blend_alpha_grad");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// out[c] += sum over pixels of x[pixel][c]   (bias gradient of nn.Conv2d(bias=True), unet_model.py:101)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, float* __restrict__ out, size_t npix,
                                                          int C, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float lds[256 * VEC];
    const int t = threadIdx.x;
    const int nvec = C / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec;
    const bool active = pl < ppb;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    const size_t end = min(npix, (size_t)(blockIdx.x + 1) * pix_per_block);
    if (active)
        for (size_t pix = (size_t)blockIdx.x * pix_per_block + pl; pix < end; pix += ppb) {
            const Vec16<T> v = load_vec16(x + pix * C + cv * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += v.get(e);
        }
#pragma unroll
    for (int e = 0; e < VEC; ++e) lds[t * VEC + e] = active ? s[e] : 0.f;
    __syncthreads();
    for (int i = t; i < nvec * VEC; i += 256) {
        const int cvj = i / VEC, e = i - cvj * VEC;
        float a = 0.f;
        for (int q = 0; q < ppb; ++q) a += lds[(q * nvec + cvj) * VEC + e];
        atomic_add_f32(&out[i], a);
    }
}

extern "C" int mrisr_channel_sum(int dtype, const void* x, float* out, size_t npix, int C, void* stream) {
    if (!x || !out) MRISR_FAIL(MRISR_E_ARG, "channel_sum: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec || C / vec > 256) MRISR_FAIL(MRISR_E_SHAPE, "channel_sum: C %d", C);
    const int ppb = 256 / (C / vec);
    const int ppblk = ppb * 32;
    const int blocks = (int)((npix + ppblk - 1) / ppblk);
    if (dtype == MRISR_BF16) channel_sum_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, out, npix, C, ppblk);
    else if (dtype == MRISR_F16) channel_sum_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, out, npix, C, ppblk);
    else if (dtype == MRISR_F32) channel_sum_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x, out, npix, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "channel_sum: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("channel_sum");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone GroupNorm statistics of an NHWC tensor: stats[n][g] += (sum, sum of squares).  Used by
// conv_forward only when a group has fewer than 4 channels (tiny test networks); otherwise the
// statistics come out of the convolution epilogue.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, int HW,
                                                       int C, int groups, int pix_per_block) {
    __shared__ double lds[64 * 2];           // fp64: order-independent to ~1e-16
    const int t = threadIdx.x, n = blockIdx.y;
    const int gs = C / groups;
    for (int i = t; i < groups * 2; i += 256) lds[i] = 0.0;
    __syncthreads();
    const int pend = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    const T* xb = x + (size_t)n * HW * C;
    const int ppb = 256 / C, c = t % C, pl = t / C;       // C <= 256: one channel per thread column
    if (pl < ppb) {
        float s = 0.f, ss = 0.f;
        for (int pix = blockIdx.x * pix_per_block + pl; pix < pend; pix += ppb) {
            const float v = to_f32(xb[(size_t)pix * C + c]);
            s += v;
            ss += v * v;
        }
        atomicAdd(&lds[2 * (c / gs)], (double)s);
        atomicAdd(&lds[2 * (c / gs) + 1], (double)ss);
    }
    __syncthreads();
    for (int i = t; i < groups * 2; i += 256)
        atomic_add_f64(&stats[stat_slot_off(gridDim.y, groups) + (size_t)n * groups * 2 + i], lds[i]);
}

int launch_gn_stats(int dtype, const void* x, double* stats, int N, int HW, int C, int groups, hipStream_t s) {
    if (groups > 64 || C > 256) MRISR_FAIL(MRISR_E_UNSUPPORTED, "gn_stats: C %d groups %d", C, groups);
    const int ppblk = 1024;
    dim3 grid(ceil_div(HW, ppblk), N);
    if (dtype == MRISR_BF16) gn_stats_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, stats, HW, C, groups, ppblk);
    else if (dtype == MRISR_F16) gn_stats_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)x, stats, HW, C, groups, ppblk);
    else gn_stats_kernel<float><<<grid, 256, 0, s>>>((const float*)x, stats, HW, C, groups, ppblk);
    MRISR_CHECK_LAUNCH("gn_stats");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// out[n][y][x][c] = max over the 2x2 window of LeakyReLU(x*scale+shift)   (MaxPool2d(2) of the activated tensor,
// unet_model.py:52; floor semantics).  Materialises the (4x smaller) pooled activation so that the encoder
// convolutions and their weight gradients run on the plain prefetching loader.
template <typename T>
__global__ __launch_bounds__(256) void norm_pool2_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ out, int H, int W,
                                                         int C, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = C / VEC, ppb = 256 / nvec, Ho = H / 2, Wo = W / 2, HWo = Ho * Wo;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    if (pl >= ppb) return;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = scale[(size_t)n * C + c + e]; sh[e] = shift[(size_t)n * C + c + e]; }
    const T* xb = x + (size_t)n * H * W * C + c;
    T* ob = out + (size_t)n * HWo * C + c;
    const int pend = min(HWo, (int)(blockIdx.x + 1) * pix_per_block);
    for (int pp = blockIdx.x * pix_per_block + pl; pp < pend; pp += ppb) {
        const int yo = pp / Wo, xo = pp - yo * Wo;
        const T* b = xb + ((size_t)(2 * yo) * W + 2 * xo) * C;
        const Vec16<T> v00 = load_vec16(b), v01 = load_vec16(b + C), v10 = load_vec16(b + (size_t)W * C), v11 = load_vec16(b + (size_t)W * C + C);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float a = lrelu(v00.get(e) * sc[e] + sh[e]), bq = lrelu(v01.get(e) * sc[e] + sh[e]);
            const float cq = lrelu(v10.get(e) * sc[e] + sh[e]), d = lrelu(v11.get(e) * sc[e] + sh[e]);
            o.set(e, fmaxf(fmaxf(a, bq), fmaxf(cq, d)));
        }
        store_vec16(ob + (size_t)pp * C, o);
    }
}

extern "C" int mrisr_norm_pool2(int dtype, const void* x, const float* scale, const float* shift, void* out, int N,
                                int H, int W, int C, void* stream) {
    if (!x || !scale || !shift || !out) MRISR_FAIL(MRISR_E_ARG, "norm_pool2: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec || C / vec > 256 || H < 2 || W < 2 || N <= 0 || N > 65535) MRISR_FAIL(MRISR_E_SHAPE, "norm_pool2: N %d C %d H %d W %d", N, C, H, W);
    const int nvec = C / vec, ppb = 256 / nvec, HWo = (H / 2) * (W / 2);
    int ppblk = ppb * 16;
    if (ppblk > HWo) ppblk = ceil_div(HWo, ppb) * ppb;
    dim3 grid(ceil_div(HWo, ppblk), N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) norm_pool2_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, scale, shift, (bf16_t*)out, H, W, C, ppblk);
    else if (dtype == MRISR_F16) norm_pool2_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)x, scale, shift, (f16_t*)out, H, W, C, ppblk);
    else if (dtype == MRISR_F32) norm_pool2_kernel<float><<<grid, 256, 0, s>>>((const float*)x, scale, shift, (float*)out, H, W, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "norm_pool2: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("norm_pool2");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// Bilinear x2 (align_corners=True) of a raw tensor z_low [N][h][w][C] -> z [N][2h][2w][C] plus the GroupNorm
// statistics of z.  The decoder's "Upsample -> conv1x1" (unet_model.py:71-72) is evaluated as
// "conv1x1 -> Upsample" (both are linear; identical up to fp rounding, 4x fewer conv FLOPs).
template <typename T>
__global__ __launch_bounds__(256) void upsample2_stats_kernel(const T* __restrict__ zl, T* __restrict__ z,
                                                              double* __restrict__ stats, int h, int w, int C,
                                                              int groups, int pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ double lds[64 * 2];
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = C / VEC, ppb = 256 / nvec;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    const int H = 2 * h, W = 2 * w, HW = H * W;
    const int gs = C / groups;
    for (int i = t; i < groups * 2; i += 256) lds[i] = 0.0;
    __syncthreads();
    float s[VEC], ss[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    const T* zb = zl + (size_t)n * h * w * C + c;
    const int pend = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    if (pl < ppb)
        for (int pix = blockIdx.x * pix_per_block + pl; pix < pend; pix += ppb) {
            const int y = pix / W, x = pix - y * W;
            int y0, y1, x0, x1;
            float wy, wx;
            up2_coord(y, h, y0, y1, wy);
            up2_coord(x, w, x0, x1, wx);
            const Vec16<T> v00 = load_vec16(zb + ((size_t)y0 * w + x0) * C), v01 = load_vec16(zb + ((size_t)y0 * w + x1) * C);
            const Vec16<T> v10 = load_vec16(zb + ((size_t)y1 * w + x0) * C), v11 = load_vec16(zb + ((size_t)y1 * w + x1) * C);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = (1.f - wy) * ((1.f - wx) * v00.get(e) + wx * v01.get(e)) +
                                wy * ((1.f - wx) * v10.get(e) + wx * v11.get(e));
                o.set(e, a);
                s[e] += a;
                ss[e] += a * a;
            }
            store_vec16(z + ((size_t)n * HW + pix) * C + c, o);
        }
    if (stats) {
        if (pl < ppb) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                atomicAdd(&lds[2 * ((c + e) / gs)], (double)s[e]);
                atomicAdd(&lds[2 * ((c + e) / gs) + 1], (double)ss[e]);
            }
        }
        __syncthreads();
        for (int i = t; i < groups * 2; i += 256)
            atomic_add_f64(&stats[stat_slot_off(gridDim.y, groups) + (size_t)n * groups * 2 + i], lds[i]);
    }
}

extern "C" int mrisr_upsample2_stats(int dtype, const void* z_low, void* z, double* stats, int N, int h, int w, int C,
                                     int groups, void* stream) {
    if (!z_low || !z) MRISR_FAIL(MRISR_E_ARG, "upsample2_stats: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec || C / vec > 256 || groups <= 0 || groups > 64 || C % groups) MRISR_FAIL(MRISR_E_SHAPE, "upsample2_stats: C %d groups %d", C, groups);
    const int ppb = 256 / (C / vec);
    const int ppblk = ppb * 16;
    dim3 grid(ceil_div(4 * h * w, ppblk), N);
    if (dtype == MRISR_BF16) upsample2_stats_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)z_low, (bf16_t*)z, stats, h, w, C, groups, ppblk);
    else if (dtype == MRISR_F16) upsample2_stats_kernel<f16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const f16_t*)z_low, (f16_t*)z, stats, h, w, C, groups, ppblk);
    else if (dtype == MRISR_F32) upsample2_stats_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)z_low, (float*)z, stats, h, w, C, groups, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "upsample2_stats: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("upsample2_stats");
    return MRISR_OK;
}

// adjoint: dz [N][2h][2w][C] -> dz_low [N][h][w][C].  One low-resolution pixel gathers a 4 x 4 neighbourhood of dz; as
// one pixel per thread that is 16 vector loads per output and the pass is bound by the L1 / texture path (2.1 GB of
// loads for a 537 MB tensor: 3 TB/s of HBM traffic).  Here a thread owns a 2 x 2 block of outputs: their 6 x 6
// neighbourhood is read once, row by row (9 loads per output), reduced along x into the two output columns first and
// then along y into the two output rows.  Block = (block range, image), thread = (block lane, 16-byte channel vector).
template <typename T>
__global__ __launch_bounds__(256) void upsample2_adjoint_kernel(const T* __restrict__ dz, T* __restrict__ dzl, int h, int w,
                                                                int C, int blk_per_block) {
    constexpr int VEC = Vec16<T>::N;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = C / VEC, ppb = 256 / nvec, bw = (w + 1) >> 1, nblk = ((h + 1) >> 1) * bw;
    const int cv = t % nvec, pl = t / nvec;
    if (pl >= ppb) return;
    const T* b = dz + (size_t)n * 4 * h * w * C + cv * VEC;
    T* ob = dzl + (size_t)n * h * w * C + cv * VEC;
    const int bend = min(nblk, (int)(blockIdx.x + 1) * blk_per_block);
    for (int bi = blockIdx.x * blk_per_block + pl; bi < bend; bi += ppb) {
        const int by = bi / bw, bx = bi - by * bw;
        const int y0 = 2 * by, x0 = 2 * bx;
        const bool vy1 = y0 + 1 < h, vx1 = x0 + 1 < w;
        // candidate rows / columns of output i: 2(y0+i)-1 .. 2(y0+i)+2 = local index 2i .. 2i+3 of the 6 staged ones
        int iy[2][kUpAdj], ix[2][kUpAdj];
        float wy[2][kUpAdj], wx[2][kUpAdj];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            up2_adjoint_weights(min(y0 + i, h - 1), h, iy[i], wy[i]);
            up2_adjoint_weights(min(x0 + i, w - 1), w, ix[i], wx[i]);
        }
        float acc[2][2][VEC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int Y = min(max(4 * by - 1 + r, 0), 2 * h - 1);       // clamped: out-of-image rows carry zero weights
            Vec16<T> d[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int X = min(max(4 * bx - 1 + q, 0), 2 * w - 1);
                d[q] = load_vec16(b + (size_t)(Y * (2 * w) + X) * C);
            }
            float hx[2][VEC];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < kUpAdj; ++k) v += wx[j][k] * d[2 * j + k].get(e);
                    hx[j][e] = v;
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (r < 2 * i || r >= 2 * i + kUpAdj) continue;        // row r is candidate r - 2i of output row i
                const float wv = wy[i][r - 2 * i];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[i][j][e] += wv * hx[j][e];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if ((i && !vy1) || (j && !vx1)) continue;
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.set(e, acc[i][j][e]);
                store_vec16(ob + (size_t)((y0 + i) * w + x0 + j) * C, o);
            }
    }
}

extern "C" int mrisr_upsample2_adjoint(int dtype, const void* dz, void* dz_low, int N, int h, int w, int C, void* stream) {
    if (!dz || !dz_low) MRISR_FAIL(MRISR_E_ARG, "upsample2_adjoint: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec || C / vec > 256 || N <= 0 || N > 65535 || h <= 0 || w <= 0 || (size_t)h * w >= (1u << 28))
        MRISR_FAIL(MRISR_E_SHAPE, "upsample2_adjoint: N %d h %d w %d C %d", N, h, w, C);
    const int nvec = C / vec, ppb = 256 / nvec, nblk = ((h + 1) / 2) * ((w + 1) / 2);
    int ppblk = ppb * 4;             // 2 x 2 output blocks per workgroup
    if (ppblk > nblk) ppblk = ceil_div(nblk, ppb) * ppb;
    dim3 grid(ceil_div(nblk, ppblk), N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MRISR_BF16) upsample2_adjoint_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dz, (bf16_t*)dz_low, h, w, C, ppblk);
    else if (dtype == MRISR_F16) upsample2_adjoint_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)dz, (f16_t*)dz_low, h, w, C, ppblk);
    else if (dtype == MRISR_F32) upsample2_adjoint_kernel<float><<<grid, 256, 0, s>>>((const float*)dz, (float*)dz_low, h, w, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "upsample2_adjoint: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("upsample2_adjoint");
    return MRISR_OK;
}

// ------------------------------------------------------------------------------------------------
// out [N][2h][2w][C] = bilinear x2 (align_corners=True) of LeakyReLU(x*scale+shift): materialises the input of
// final_up_bilinear's 3x3 conv (unet_model.py:151-152).  Gathering 4 taps per element inside the conv loader of a
// 32-channel-wide layer is load-bound, so here the interpolated activation is written once and the conv (and
// its weight gradient) run on the plain prefetching loader.
// Thread = (group, 16-byte channel vector); group (gy, gx), gy in [0, h], gx in [0, w], owns the output pixels
// Y in {2gy-1, 2gy}, X in {2gx-1, 2gx}: both rows interpolate between the SAME two source rows (gy-1, gy) and both
// columns between the same two source columns, so the group loads and activates 4 source vectors for its 4 outputs -
// one GroupNorm+LeakyReLU evaluation per output element instead of four (as one output pixel per thread the pass was
// VALU-bound: 24 instructions per element, 3.2 TB/s).  Coordinates and weights are aten's (up2_coord) per output
// row / column; a group whose two rows (columns) do not share their source pair - fp32 rounding of the source
// coordinate at the last row - falls back to one output at a time.
template <typename T>
__global__ __launch_bounds__(256) void norm_upsample2_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ out, int N,
                                                             int h, int w, int C, int grp_per_block) {
    constexpr int VEC = Vec16<T>::N;
    const int t = threadIdx.x, n = blockIdx.y;
    const int nvec = C / VEC, ppb = 256 / nvec, H = 2 * h, W = 2 * w, gw = w + 1, ngrp = (h + 1) * gw;
    const int cv = t % nvec, pl = t / nvec, c = cv * VEC;
    if (pl >= ppb) return;
    float sc[VEC], sh[VEC];            // hoisted: one (n, channel vector) per thread
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = scale[(size_t)n * C + c + e]; sh[e] = shift[(size_t)n * C + c + e]; }
    const T* b = x + (size_t)n * h * w * C + c;
    T* ob = out + (size_t)n * H * W * C + c;
    auto act = [&](const T* src, float* a) {
        const Vec16<T> v = load_vec16(src);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = lrelu(v.get(e) * sc[e] + sh[e]);
    };
    const int gend = min(ngrp, (int)(blockIdx.x + 1) * grp_per_block);
    for (int g = blockIdx.x * grp_per_block + pl; g < gend; g += ppb) {
        const int gy = g / gw, gx = g - gy * gw;
        int Y[2] = {2 * gy - 1, 2 * gy}, X[2] = {2 * gx - 1, 2 * gx};
        const bool vy[2] = {Y[0] >= 0, Y[1] < H}, vx[2] = {X[0] >= 0, X[1] < W};
        int y0[2], y1[2], x0[2], x1[2];
        float wy[2], wx[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            up2_coord(min(max(Y[k], 0), H - 1), h, y0[k], y1[k], wy[k]);
            up2_coord(min(max(X[k], 0), W - 1), w, x0[k], x1[k], wx[k]);
        }
        const int ky = vy[0] ? 0 : 1, kx = vx[0] ? 0 : 1;        // the pair the group shares
        const bool shared = (!vy[0] || !vy[1] || (y0[0] == y0[1] && y1[0] == y1[1])) &&
                            (!vx[0] || !vx[1] || (x0[0] == x0[1] && x1[0] == x1[1]));
        if (shared) {
            float a00[VEC], a01[VEC], a10[VEC], a11[VEC];
            act(b + (size_t)(y0[ky] * w + x0[kx]) * C, a00);
            act(b + (size_t)(y0[ky] * w + x1[kx]) * C, a01);
            act(b + (size_t)(y1[ky] * w + x0[kx]) * C, a10);
            act(b + (size_t)(y1[ky] * w + x1[kx]) * C, a11);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!vy[i] || !vx[j]) continue;
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        o.set(e, (1.f - wy[i]) * ((1.f - wx[j]) * a00[e] + wx[j] * a01[e]) + wy[i] * ((1.f - wx[j]) * a10[e] + wx[j] * a11[e]));
                    store_vec16(ob + (size_t)(Y[i] * W + X[j]) * C, o);
                }
        } else {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    if (!vy[i] || !vx[j]) continue;
                    float a00[VEC], a01[VEC], a10[VEC], a11[VEC];
                    act(b + (size_t)(y0[i] * w + x0[j]) * C, a00);
                    act(b + (size_t)(y0[i] * w + x1[j]) * C, a01);
                    act(b + (size_t)(y1[i] * w + x0[j]) * C, a10);
                    act(b + (size_t)(y1[i] * w + x1[j]) * C, a11);
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        o.set(e, (1.f - wy[i]) * ((1.f - wx[j]) * a00[e] + wx[j] * a01[e]) + wy[i] * ((1.f - wx[j]) * a10[e] + wx[j] * a11[e]));
                    store_vec16(ob + (size_t)(Y[i] * W + X[j]) * C, o);
                }
        }
    }
}

extern "C" int mrisr_norm_upsample2(int dtype, const void* x, const float* scale, const float* shift, void* out, int N,
                                    int h, int w, int C, void* stream) {
    if (!x || !scale || !shift || !out) MRISR_FAIL(MRISR_E_ARG, "norm_upsample2: null pointer");
    const int vec = mrisr_vec(dtype);
    if (C % vec) MRISR_FAIL(MRISR_E_SHAPE, "norm_upsample2: C %d", C);
    if (C / vec > 256 || N <= 0 || N > 65535) MRISR_FAIL(MRISR_E_SHAPE, "norm_upsample2: C %d N %d", C, N);
    if (h <= 0 || w <= 0 || (size_t)h * w >= (1u << 26)) MRISR_FAIL(MRISR_E_SHAPE, "norm_upsample2: h %d w %d", h, w);
    const int nvec = C / vec, ppb = 256 / nvec, ngrp = (h + 1) * (w + 1);
    int ppblk = ppb * 8;             // groups per block (4 output pixels each)
    if (ppblk > ngrp) ppblk = ceil_div(ngrp, ppb) * ppb;
    dim3 grid(ceil_div(ngrp, ppblk), N);
    if (dtype == MRISR_BF16) norm_upsample2_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, scale, shift, (bf16_t*)out, N, h, w, C, ppblk);
    else if (dtype == MRISR_F16) norm_upsample2_kernel<f16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, scale, shift, (f16_t*)out, N, h, w, C, ppblk);
    else if (dtype == MRISR_F32) norm_upsample2_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, scale, shift, (float*)out, N, h, w, C, ppblk);
    else MRISR_FAIL(MRISR_E_DTYPE, "norm_upsample2: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("norm_upsample2");
    return MRISR_OK;
}
