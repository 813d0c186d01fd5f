// Shared pieces of the implicit-GEMM convolution kernels (forward/dgrad and wgrad):
// the fused input loader (GroupNorm+LeakyReLU apply, 2x2 max-pool, bilinear x2, concat, blend)
// and the LDS image conventions.
#pragma once
#include "common.h"

constexpr int kConvThreads = 256;   // 4 waves (wgrad; one half of the forward kernel's 8-wave workgroup)
constexpr int kRowBytes = 64;       // bytes of K (input channels) per LDS row and cin-chunk
constexpr int kMaxHaloIter = 6;     // ceil(340 / 64): tiles 8x32, 16x16, 32x8 (+halo)

// LDS rows are 64 B = four 16-B chunks; chunk c of row r lives at chunk position c ^ swz(r), which
// makes the MFMA fragment reads (ds_read_b128, 32 consecutive rows, same chunk) conflict-free.
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * kRowBytes + ((chunk ^ swz(row)) << 4);
}

// Forward halo image: 64 B of channels per pixel in 80-byte rows (16 B pad).  Linear in the row index, so
// the 9 tap offsets are wave-uniform scalars added to one per-lane base, and 16 consecutive rows cover
// all 16 sixteen-byte bank slots (5*i mod 16 is a bijection): conflict-free ds_read_b128 fragments.
constexpr int kHaloRowBytes = 80;
__device__ __forceinline__ int halo_off(int row, int chunk) { return row * kHaloRowBytes + (chunk << 4); }

// forward kernel LDS budget: two halo tiles of kMaxHaloIter*64 rows (every commit slot has a row, so the LDS stores
// need no bounds predicate) + the weight images + BN floats of bias
static inline size_t conv_halo_bytes() { return (size_t)kMaxHaloIter * 64 * kHaloRowBytes; }
// bf16 plain epilogue staging: 4 waves x 32 pixel rows x (64 couts x 2 B + 16 B pad)
static inline size_t conv_stage_bytes() {
#ifdef MRISR_STAGED_STORES
    return 4 * 32 * (64 * 2 + 16);
#else
    return 0;
#endif
}
// LDS-DMA halo variant (sources stored as-is: input gradients, materialised activations, VGG): four halo buffers (two
// per half, double-buffered) of 340 rows x 64 B - rows are 64 B with the 16-B chunk position XOR-swizzled by
// (row >> 2) & 3 like the weight image (an LDS-DMA piece is 1 KiB of consecutive LDS bytes, so rows cannot be padded)
constexpr int kDmaHaloRows = 340;     // max over the tile shapes: 10 x 34 (8x32 and 32x8 tiles), 18 x 18 = 324
constexpr int kDmaHaloBytes = kDmaHaloRows * 64;
static inline size_t conv_halo_total(bool dma) { return dma ? 4 * (size_t)kDmaHaloBytes : 2 * conv_halo_bytes(); }
static inline bool conv_weights_stationary(int nchunks, size_t wimg, bool dma = false) {
    return nchunks * wimg + conv_halo_total(dma) + (64 + 128) * sizeof(float) + conv_stage_bytes() <= 159 * 1024;   // + bias, affine tables, staging
}
// wgrad kernel variant: 0 = generic, 1 / 2 / 4 = FAST with that k-step interleave (bf16 3x3 plain loader, 8x32 tiles,
// every channel block of the launch holding the same number of 32x32 fragment pairs)
static inline int conv_wgrad_fast(int dtype, int loader, int ks, int tw_log2, int Cout, int Cin) {
    if (dtype == MRISR_F32 || loader != MRISR_SP_NONE || ks != 3 || tw_log2 != 5) return 0;
    const int nfo = Cout % 64 == 0 ? 2 : (Cout <= 32 ? 1 : 0), nfi = Cin % 64 == 0 ? 2 : (Cin <= 32 ? 1 : 0);
    if (!nfo || !nfi) return 0;
    return 4 / (nfo * nfi);
}
// output-channel block of the ring weight layout (csrc/conv_ring.hip, csrc/conv_pc.hip) for an operand, 0 = no such image
// (128-channel blocks, else 64-channel blocks; one 32-channel block for the 32-channel layer of the 2x head)
__host__ __device__ static inline int conv_ring_bn(int dtype, int Cout, int Cin, int ksize) {
    if (dtype != MRISR_BF16 && dtype != MRISR_F16) return 0;
    if (ksize != 3 || Cin % 16) return 0;
    if (Cout % 128 == 0) return 128;
    if (Cout % 64 == 0) return 64;                 // conv_pc.hip's tall-tile variant
    return Cout == 32 ? 32 : 0;
}
static inline int conv_choose_bn(int Cout) { return Cout >= 64 ? 64 : 32; }
static inline int conv_bk(int dtype) { return dtype == MRISR_F32 ? 16 : 32; }

// 8-element fragment of a 16-bit storage type and its 32x32x16 MFMA (fp32 accumulate)
template <typename T> struct Frag16;
template <> struct Frag16<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x16 mma(type a, type b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Frag16<f16_t> {
    typedef f16x8 type;
    static __device__ __forceinline__ f32x16 mma(type a, type b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag16<float> {      // (placeholder so that dead 16-bit branches of fp32 instantiations still parse)
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x16 mma(type, type, f32x16 c) { return c; }
};

struct SrcDev {
    const void* ptr;
    const float* scale;
    const float* shift;
    int C, H, W, mode, off_y, off_x;
    unsigned img_bytes;   // H * W * C * sizeof(T): byte stride between images (host-checked < 2^32)
};

struct ConvParams {
    SrcDev src[2];
    const float* blend_alpha;
    const void* wpacked;
    const float* bias;
    void* out;
    double* stats;
    const void* mask; // forward: relu_mask (epilogue kEpiMask), same layout as out
    const void* dy;   // wgrad only
    float* dw;        // wgrad only
    float* wsp;       // wgrad only: partial-sum workspace (two-stage reduction) or NULL (atomics into dw)
    int N, H, W, Cin, CinP, Cout, CoutP;
    int nsrc, combine, out_mode, groups, relu_out;
    int tw_log2, th, tiles_x, tiles_y, ncb, nchunks;
    int ksplit;       // wgrad only
    int ws;           // forward: all cin chunks of the weight image stay resident in LDS
    int ntiles, tiles_per_block;   // forward: persistent blocks own contiguous tile ranges
    int cus;          // CUs the persistent grid is sized for (mrisr_conv_desc.cu_limit, else all)
    int dbg;          // tuning builds only (-DMRISR_TUNING, env MRISR_DEBUG): 1 no stores, 2 no LDS commit, 4 no global loads, 8 no MFMA
};
// Ablation bits exist in tuning builds only (tools/build_prof.sh): the product library never reads the environment
// and compiles every `DBG(p) & bit` test away.
#ifdef MRISR_TUNING
#define DBG(p) ((p).dbg)
#else
#define DBG(p) 0
#endif

// hipcc re-loads kernel arguments with s_load + s_waitcnt at every use inside long loops (they are "free" to
// rematerialise); pinning the hot ones as opaque scalars keeps them in SGPRs instead.
#define PIN_F(q, src, f) { auto v__ = (src).f; asm volatile("" : "+s"(v__)); (q).f = v__; }
__device__ __forceinline__ ConvParams pin_params(const ConvParams& in) {
    ConvParams q = in;
    PIN_F(q, in, H) PIN_F(q, in, W) PIN_F(q, in, Cout) PIN_F(q, in, N) PIN_F(q, in, nchunks) PIN_F(q, in, tw_log2)
    PIN_F(q, in, th) PIN_F(q, in, tiles_x) PIN_F(q, in, tiles_y) PIN_F(q, in, nsrc) PIN_F(q, in, dbg)
    PIN_F(q, in, groups) PIN_F(q, in, relu_out) PIN_F(q, in, out) PIN_F(q, in, stats) PIN_F(q, in, bias)
    PIN_F(q, in, dy) PIN_F(q, in, dw) PIN_F(q, in, wsp) PIN_F(q, in, mask) PIN_F(q, in, Cin) PIN_F(q, in, combine)
    PIN_F(q, in, src[0].ptr) PIN_F(q, in, src[0].scale) PIN_F(q, in, src[0].shift) PIN_F(q, in, src[0].C)
    PIN_F(q, in, src[0].H) PIN_F(q, in, src[0].W) PIN_F(q, in, src[0].mode) PIN_F(q, in, src[0].off_y)
    PIN_F(q, in, src[0].off_x) PIN_F(q, in, src[0].img_bytes)
    PIN_F(q, in, src[1].ptr) PIN_F(q, in, src[1].scale) PIN_F(q, in, src[1].shift) PIN_F(q, in, src[1].C)
    PIN_F(q, in, src[1].H) PIN_F(q, in, src[1].W) PIN_F(q, in, src[1].mode) PIN_F(q, in, src[1].off_y)
    PIN_F(q, in, src[1].off_x) PIN_F(q, in, src[1].img_bytes)
    return q;
}

// Scalar (SALU) base of image n of a source, and the 24-bit multiply-add the per-lane byte offsets are made of
// (inline asm: hipcc otherwise widens these to 64-bit VALU multiplies, v_mad_u64_u32 / v_mul_lo_u32, quarter rate).
__device__ __forceinline__ const char* image_base(const void* ptr, int n, unsigned img_bytes) {
    unsigned lo, hi;
    asm("s_mul_i32 %0, %2, %3\n\ts_mul_hi_u32 %1, %2, %3" : "=&s"(lo), "=s"(hi) : "s"(n), "s"(img_bytes));
    return (const char*)ptr + (((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {   // (a & 0xffffff) * (b & 0xffffff) + c
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Per-thread precomputed geometry of the halo pixels this thread stages (same for every cin chunk).
template <int SPATIAL> struct HaloGeom;
template <> struct HaloGeom<MRISR_SP_NONE> {
    int off0[kMaxHaloIter];   // element offset into src0 (without channel), -1 = zero
    int off1[kMaxHaloIter];   // same for src1 (concat / blend)
};
template <> struct HaloGeom<MRISR_SP_POOL2> {
    int off0[kMaxHaloIter];
};
template <> struct HaloGeom<MRISR_SP_UP2> {
    int off0[kMaxHaloIter];
    int dyo[kMaxHaloIter], dxo[kMaxHaloIter];
    float wy[kMaxHaloIter], wx[kMaxHaloIter];
};

template <typename T>
__device__ __forceinline__ void transform_vec(Vec16<T>& v, int mode, const float* sc, const float* sh) {
    if (mode == MRISR_SRC_NORM) {
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) v.set(e, lrelu(v.get(e) * sc[e] + sh[e]));
    } else if (mode == MRISR_SRC_RELU) {
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) v.set(e, fmaxf(v.get(e), 0.f));
    }
}
template <typename T>
__device__ __forceinline__ void transform_f(const Vec16<T>& v, float* o, int mode, const float* sc,
                                            const float* sh) {
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) {
        float x = v.get(e);
        if (mode == MRISR_SRC_NORM) x = lrelu(x * sc[e] + sh[e]);
        else if (mode == MRISR_SRC_RELU) x = fmaxf(x, 0.f);
        o[e] = x;
    }
}

// Computes the geometry of halo pixel `hp` = (hy, hx) of the tile at (n, ty0, tx0); conv padding pad.
template <int SPATIAL>
__device__ __forceinline__ void halo_geom_yx(HaloGeom<SPATIAL>& g, int i, int hy, int hx, bool slot_ok,
                                             int pad, int n, int ty0, int tx0, const ConvParams& p);

template <int SPATIAL>
__device__ __forceinline__ void halo_geom_init(HaloGeom<SPATIAL>& g, int i, int hp, int npix_halo, int hw,
                                               int pad, int n, int ty0, int tx0, const ConvParams& p) {
    const int hy = hp / hw, hx = hp - hy * hw;
    halo_geom_yx<SPATIAL>(g, i, hy, hx, hp < npix_halo, pad, n, ty0, tx0, p);
}

template <int SPATIAL>
__device__ __forceinline__ void halo_geom_yx(HaloGeom<SPATIAL>& g, int i, int hy, int hx, bool slot_ok,
                                             int pad, int n, int ty0, int tx0, const ConvParams& p) {
    int o0 = -1;
    const int y = ty0 + hy - pad, x = tx0 + hx - pad;
    const bool inside = slot_ok & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
    const SrcDev& s0 = p.src[0];
    if constexpr (SPATIAL == MRISR_SP_NONE) {
        // branch-free (selects): this runs for 6 slots at every tile change of the persistent forward kernel.
        // (unsigned)(v) < (unsigned)n  <=>  0 <= v < n
        const int ys = y - s0.off_y, xs = x - s0.off_x;
        const bool ok0 = inside & ((unsigned)ys < (unsigned)s0.H) & ((unsigned)xs < (unsigned)s0.W);
        const int e0 = ((n * s0.H + ys) * s0.W + xs) * s0.C;
        o0 = ok0 ? e0 : -1;
        const SrcDev& s1 = p.src[1];
        const int y1 = y - s1.off_y, x1 = x - s1.off_x;
        const bool ok1 = inside & (p.nsrc > 1) & ((unsigned)y1 < (unsigned)s1.H) & ((unsigned)x1 < (unsigned)s1.W);
        const int e1 = ((n * s1.H + y1) * s1.W + x1) * s1.C;
        g.off0[i] = o0;
        g.off1[i] = ok1 ? e1 : -1;
    } else if constexpr (SPATIAL == MRISR_SP_POOL2) {
        if (inside) o0 = ((n * s0.H + 2 * y) * s0.W + 2 * x) * s0.C;
        g.off0[i] = o0;
    } else {
        int dyo = 0, dxo = 0;
        float wy = 0.f, wx = 0.f;
        if (inside) {
            const int ys = y - s0.off_y, xs = x - s0.off_x;
            if (ys >= 0 && ys < 2 * s0.H && xs >= 0 && xs < 2 * s0.W) {
                int y0, y1, x0, x1;
                up2_coord(ys, s0.H, y0, y1, wy);
                up2_coord(xs, s0.W, x0, x1, wx);
                o0 = ((n * s0.H + y0) * s0.W + x0) * s0.C;
                dyo = (y1 - y0) * s0.W * s0.C;
                dxo = (x1 - x0) * s0.C;
            }
        }
        g.off0[i] = o0;
        g.dyo[i] = dyo;
        g.dxo[i] = dxo;
        g.wy[i] = wy;
        g.wx[i] = wx;
    }
}

// Loads + transforms the 16-byte channel vector [c0, c0+VEC) of halo pixel slot i.
template <typename T, int SPATIAL>
__device__ __forceinline__ Vec16<T> halo_load(const HaloGeom<SPATIAL>& g, int i, int c0, const ConvParams& p,
                                              const float* sc0, const float* sh0, const float* sc1,
                                              const float* sh1, float blend_a, int which_src, int cs) {
    constexpr int VEC = Vec16<T>::N;
    Vec16<T> out;
    out.zero();
    if constexpr (SPATIAL == MRISR_SP_NONE) {
        if (p.combine == MRISR_COMBINE_BLEND) {
            const int o0 = g.off0[i], o1 = g.off1[i];
            if (o0 >= 0 && o1 >= 0) {   // both sources share the conv geometry
                Vec16<T> a = load_vec16((const T*)p.src[0].ptr + o0 + c0);
                Vec16<T> b = load_vec16((const T*)p.src[1].ptr + o1 + c0);
                float fa[VEC], fb[VEC];
                transform_f(a, fa, p.src[0].mode, sc0, sh0);
                transform_f(b, fb, p.src[1].mode, sc1, sh1);
#pragma unroll
                for (int e = 0; e < VEC; ++e) out.set(e, blend_a * fa[e] + (1.f - blend_a) * fb[e]);
            }
        } else {
            const int o = which_src ? g.off1[i] : g.off0[i];
            if (o >= 0 && cs >= 0) {
                out = load_vec16((const T*)p.src[which_src].ptr + o + cs);
                transform_vec(out, p.src[which_src].mode, which_src ? sc1 : sc0, which_src ? sh1 : sh0);
            }
        }
    } else if constexpr (SPATIAL == MRISR_SP_POOL2) {
        const int o = g.off0[i];
        if (o >= 0 && cs >= 0) {
            const T* b = (const T*)p.src[0].ptr + o + cs;
            const int rs = p.src[0].W * p.src[0].C;
            Vec16<T> v00 = load_vec16(b), v01 = load_vec16(b + p.src[0].C);
            Vec16<T> v10 = load_vec16(b + rs), v11 = load_vec16(b + rs + p.src[0].C);
            float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
            transform_f(v00, f00, p.src[0].mode, sc0, sh0);
            transform_f(v01, f01, p.src[0].mode, sc0, sh0);
            transform_f(v10, f10, p.src[0].mode, sc0, sh0);
            transform_f(v11, f11, p.src[0].mode, sc0, sh0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) out.set(e, fmaxf(fmaxf(f00[e], f01[e]), fmaxf(f10[e], f11[e])));
        }
    } else {
        const int o = g.off0[i];
        if (o >= 0 && cs >= 0) {
            const T* b = (const T*)p.src[0].ptr + o + cs;
            Vec16<T> v00 = load_vec16(b), v01 = load_vec16(b + g.dxo[i]);
            Vec16<T> v10 = load_vec16(b + g.dyo[i]), v11 = load_vec16(b + g.dyo[i] + g.dxo[i]);
            float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
            transform_f(v00, f00, p.src[0].mode, sc0, sh0);
            transform_f(v01, f01, p.src[0].mode, sc0, sh0);
            transform_f(v10, f10, p.src[0].mode, sc0, sh0);
            transform_f(v11, f11, p.src[0].mode, sc0, sh0);
            const float wy1 = g.wy[i], wx1 = g.wx[i], wy0 = 1.f - wy1, wx0 = 1.f - wx1;
#pragma unroll
            for (int e = 0; e < VEC; ++e)   // aten upsample_bilinear2d: l0*(w0*a+w1*b) + l1*(w0*c+w1*d)
                out.set(e, wy0 * (wx0 * f00[e] + wx1 * f01[e]) + wy1 * (wx0 * f10[e] + wx1 * f11[e]));
        }
    }
    return out;
}

// Loads scale/shift vectors for channel vector c (source s) of image n; fills 1/0 when not NORM.
template <int VEC>
__device__ __forceinline__ void load_affine(const SrcDev& s, int n, int c, float* sc, float* sh) {
    if (s.mode == MRISR_SRC_NORM && c >= 0) {
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 a = gload<f32x4>(s.scale + (size_t)n * s.C + c + e);
            const f32x4 b = gload<f32x4>(s.shift + (size_t)n * s.C + c + e);
            sc[e] = a[0]; sc[e + 1] = a[1]; sc[e + 2] = a[2]; sc[e + 3] = a[3];
            sh[e] = b[0]; sh[e + 1] = b[1]; sh[e + 2] = b[2]; sh[e + 3] = b[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    }
}

// wgrad LDS rows are 128 B (two 64-B cin sub-chunks); the 64-B halves of a row are exchanged on
// every other row pair so that the transposed fragment reads (4 consecutive rows x 64 B per
// 32-lane half) cover all 64 banks.
__device__ __forceinline__ int lds_off128(int row, int sub, int chunk) {
    return row * 128 + ((sub ^ ((row >> 1) & 1)) << 6) + (chunk << 4);
}

// Stages the transformed halo tile of cin chunk kc into LDS (rows = halo pixels; ROWB = 64: forward
// image, ROWB = 128: wgrad image, `sub` selects the 64-B half).
template <typename T, int SPATIAL, int ROWB = 64, int BATCH = kMaxHaloIter>
__device__ __forceinline__ void stage_halo(char* lds_halo, const HaloGeom<SPATIAL>& g, int kc, int n,
                                           int npix_halo, float blend_a, const ConvParams& p, int sub = 0,
                                           int t = threadIdx.x) {
    constexpr int VEC = Vec16<T>::N;
    const int chunk = t & 3;
    const int c0 = kc * (kRowBytes / (int)sizeof(T)) + chunk * VEC;   // channel in the conv input
    int which = 0, cs = c0;
    if (p.combine == MRISR_COMBINE_CONCAT) {
        if (p.nsrc > 1 && c0 >= p.src[0].C) { which = 1; cs = c0 - p.src[0].C; }
        if (cs >= p.src[which].C) cs = -1;     // zero padding of Cin up to CinP
    } else if (c0 >= p.src[0].C) {
        cs = -1;
    }
    float sc0[VEC], sh0[VEC], sc1[VEC], sh1[VEC];
    if constexpr (SPATIAL != MRISR_SP_NONE) {      // single-source kernels
        load_affine<VEC>(p.src[0], n, cs, sc0, sh0);
#pragma unroll
        for (int i = 0; i < kMaxHaloIter; ++i) {
            const int hp = (t >> 2) + 64 * i;
            const Vec16<T> v = halo_load<T, SPATIAL>(g, i, c0, p, sc0, sh0, sc0, sh0, blend_a, 0, cs);
            if (hp < npix_halo)
                *reinterpret_cast<decltype(v.v)*>(
                    lds_halo + (ROWB == 64 ? halo_off(hp, chunk) : lds_off128(hp, sub, chunk))) = v.v;
            __builtin_amdgcn_sched_barrier(0);   // one gather (4 loads) in flight: keeps VGPRs for the accumulators
        }
        return;
    } else {
        if (p.combine == MRISR_COMBINE_BLEND) {
            load_affine<VEC>(p.src[0], n, cs, sc0, sh0);
            load_affine<VEC>(p.src[1], n, cs, sc1, sh1);
        } else if (which == 0) {
            load_affine<VEC>(p.src[0], n, cs, sc0, sh0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sc1[e] = 1.f; sh1[e] = 0.f; }
        } else {
            load_affine<VEC>(p.src[1], n, cs, sc1, sh1);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sc0[e] = 1.f; sh0[e] = 0.f; }
        }
        const bool blend_dead = (p.combine == MRISR_COMBINE_BLEND && cs < 0);
#pragma unroll
        for (int i0 = 0; i0 < kMaxHaloIter; i0 += BATCH) {   // BATCH gathers in flight at a time
            Vec16<T> vals[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (blend_dead || i0 + j >= kMaxHaloIter) vals[j].zero();
                else vals[j] = halo_load<T, SPATIAL>(g, i0 + j, c0, p, sc0, sh0, sc1, sh1, blend_a, which, cs);
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int hp = (t >> 2) + 64 * (i0 + j);
                if (i0 + j < kMaxHaloIter && hp < npix_halo)
                    *reinterpret_cast<decltype(vals[j].v)*>(
                        lds_halo + (ROWB == 64 ? halo_off(hp, chunk) : lds_off128(hp, sub, chunk))) = vals[j].v;
            }
            if (BATCH < kMaxHaloIter) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// tile shape for a W-wide image: TH*TW = 256
static inline void conv_choose_tile(int W, int& th, int& tw_log2) {
    if (W > 16) { tw_log2 = 5; th = 8; }
    else if (W > 8) { tw_log2 = 4; th = 16; }
    else { tw_log2 = 3; th = 32; }
}
