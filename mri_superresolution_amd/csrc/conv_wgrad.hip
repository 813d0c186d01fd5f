// Weight gradient of the 3x3 / 1x1 convolution on MFMA (gfx950).
//
// Replaces aten convolution_backward's weight branch for the nn.Conv2d layers of
// /root/reference/models/unet_model.py (58 % of the reference's CPU step).
//   dW[co][tap][ci] += sum_pixels dy[pixel][co] * in[pixel + tap][ci]
// GEMM view per tap: M = co, N = ci, K = pixels.  Both operands are pixel-major (NHWC), i.e.
// K-strided, so fragments come from LDS through the transposing read ds_read_b64_tr_b16 (bf16) or
// single-dword reads (exact-fp32 32x32x2 MFMA).  The conv input is re-created by the same fused
// loader as the forward pass (GroupNorm+LeakyReLU apply / concat / blend; gathers staged synchronously).
//
// Workgroup = 8 waves on one (co block, ci block) of 128-byte channel rows, looping over a strided
// slice of the 256-pixel tiles (split-K over the grid).  bf16: wave = (32x32 fragment pair, tap group
// {0-4} / {5-8}), so the two waves that share a SIMD split the 9 taps and keep its matrix pipe busy
// while the per-wave accumulator stays at 80 VGPRs - which leaves room to prefetch the next tile's
// operands through registers while the MFMAs run.  LDS is double-buffered (one barrier per tile).
// Partial sums are added to the fp32 master gradient with 128-byte-segment float atomics.
#include <stdlib.h>

#include <type_traits>

#include <mutex>

#include "conv_common.h"

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> { static constexpr int BC = 64; };   // channels per 128-B row
template <> struct WgTraits<f16_t> { static constexpr int BC = 64; };
template <> struct WgTraits<float> { static constexpr int BC = 32; };

constexpr int kWgThreads = 512;
constexpr int kWgSlots = 6;          // halo pixels per thread: slot i = halo pixel (t>>3) + 64 i
static_assert(kWgSlots == kMaxHaloIter, "HaloGeom holds kMaxHaloIter slots");

template <typename T>
__device__ __forceinline__ typename Frag16<T>::type tr_read_frag(const char* base0, const char* base1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base1);
    union { struct { s16x4 a, b; } s; typename Frag16<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

constexpr int kLoaderBlendW = 3;     // template-only loader kind: sigmoid(alpha)-blend of two sources

// FAST = 1, 2 or 4 (0 = generic): bf16 3x3, 8x32 tiles, every channel block of the launch holding the same number of
// 32x32 fragment pairs (4 / FAST: Cout and Cin each a multiple of 64 or <= 32); FAST = k-step interleave of the waves
// that share a pair.  Only the fully unrolled immediate-offset MFMA block is compiled in (the generic block's hoisted address pieces would
// otherwise push this variant over the register budget).
template <typename T, int SPATIAL, int KS, int FAST = 0>
__global__ __launch_bounds__(kWgThreads, 2) void conv_wgrad_kernel(const ConvParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams p = pin_params(p_in);
    constexpr int NTAPS = KS * KS;
    constexpr int NT0 = (NTAPS + 1) / 2;      // taps of tap group 0 (group 1 takes the rest)
    constexpr int PAD = KS / 2;
    constexpr int BC = WgTraits<T>::BC;
    constexpr int VEC = Vec16<T>::N;
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int GSP = (SPATIAL == kLoaderBlendW) ? MRISR_SP_NONE : SPATIAL;
    constexpr int NH = SPATIAL == MRISR_SP_NONE ? 1 : SPATIAL == kLoaderBlendW ? 2 : 0;   // prefetched vectors per slot

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int TW = 1 << p.tw_log2, TH = p.th;
    const int hw = TW + 2 * PAD, hh = TH + 2 * PAD;
    const int npix_halo = hw * hh;
    const int buf_bytes = 256 * 128 + npix_halo * 128;       // [dy 256 px][halo px], 128-B rows

    const int ncib = (p.Cin + BC - 1) / BC;
    // XCD-aware order (as in the forward kernel's weights-stationary variants): workgroups are dispatched x-fastest and
    // land on XCD (linear id) % 8; the logical index (id % 8) * (G / 8) + id / 8 puts the channel-block pairs of one
    // k-slice - which all read the same dy / activation tiles - and the neighbouring slices behind one L2
    int bx = blockIdx.x, by = blockIdx.y;
#ifndef MRISR_NO_XCD_REMAP
    {
        const int G = gridDim.x * gridDim.y;
        if ((G & 7) == 0) {
            const int id = by * gridDim.x + bx, lid = (id & 7) * (G >> 3) + (id >> 3);
            by = lid / gridDim.x;
            bx = lid - by * gridDim.x;
        }
    }
#endif
    const int cib = bx % ncib, cob = bx / ncib;
    const int co0 = cob * BC, ci0 = cib * BC;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x;

    float blend_a = 0.f;
    if (p.combine == MRISR_COMBINE_BLEND) blend_a = 1.f / (1.f + __expf(-gload<float>(p.blend_alpha)));

    // wave roles: tap group, 32x32 fragment pair, and - when the channel block holds fewer than 4 pairs
    // (Cout or Cin <= 32) - a share of the k-steps, so no wave multiplies padding
    const int tg = wave >> 2;                                 // tap group (1x1: K half)
    const int tgu = __builtin_amdgcn_readfirstlane(tg);       // the same, as a scalar (uniform branches)
    const int nfo = kBf16 ? (min(p.Cout - co0, BC) > 32 ? 2 : 1) : 1;
    const int nfi = kBf16 ? (min(p.Cin - ci0, BC) > 32 ? 2 : 1) : 1;
    const int npairs = nfo * nfi, kparts = 4 / npairs;        // 1, 2 or 4 pairs
    const int pair = (wave & 3) % npairs, kpart = (wave & 3) / npairs;
    const int kpartu = __builtin_amdgcn_readfirstlane(kpart);
    const int fo = pair / nfi, fi = pair % nfi;
    f32x16 acc[NT0];
#pragma unroll
    for (int i = 0; i < NT0; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // staging roles: dy -> 16-B chunk (t&7) of pixels (t>>3) + 64 i, i < 4; halo -> chunk (t&7) of slots (t>>3) + 64 i
    const int ch8 = t & 7;
    int hyx[kWgSlots];
#pragma unroll
    for (int i = 0; i < kWgSlots; ++i) {
        const int hp = (t >> 3) + 64 * i;
        const int hy = hp / hw;
        hyx[i] = hp < npix_halo ? ((hy << 16) | (hp - hy * hw)) : -1;
    }
    HaloGeom<GSP> geom;
    Vec16<T> pdy[4];
    Vec16<T> ph[kWgSlots][NH > 0 ? NH : 1];
    float sc[VEC], sh[VEC], sc1[NH == 2 ? VEC : 1], sh1[NH == 2 ? VEC : 1];
    int pmask = 0, pmode = 0, dymask = 0;
    float pslope = 1.f;                 // NH == 1: activation as max(y, slope*y): 0.2 NORM, 1 RAW, 0 RELU

    auto decode = [&](int tile, int& n, int& ty0, int& tx0) {
        const int tx = tile % p.tiles_x;
        const int r = tile / p.tiles_x;
        n = r / p.tiles_y;
        ty0 = (r - n * p.tiles_y) * TH;
        tx0 = tx * TW;
    };
    auto set_geom = [&](int n, int ty0, int tx0) {
        if constexpr (FAST) {      // halo 10 x 34 known at compile time: slot coordinates recomputed (3 VALU), not kept live
            int tq = t >> 3;
            asm volatile("" : "+v"(tq));
#pragma unroll
            for (int i = 0; i < kWgSlots; ++i) {
                const int hp = tq + 64 * i, hy = hp / 34;
                halo_geom_yx<GSP>(geom, i, hy, hp - hy * 34, hp < 340, PAD, n, ty0, tx0, p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kWgSlots; ++i)
                halo_geom_yx<GSP>(geom, i, hyx[i] >> 16, hyx[i] & 0xffff, hyx[i] >= 0, PAD, n, ty0, tx0, p);
        }
    };
    // channel vector of this thread inside the conv input for the halo: sub-chunk (ch8>>2), chunk (ch8&3)
    const int c_in = ci0 + ch8 * VEC;
    auto issue = [&](int n, int ty0, int tx0) {
        if (DBG(p) & 4) return;
        const int c = co0 + ch8 * VEC;
        int dm = 0;
        // opaque copy of the thread index: the per-slot pixel coordinates are 2 VALU ops each to recompute, hoisted
        // out of the persistent loop they are 8 long-lived VGPRs (and spill)
        int tq = t >> 3;
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = tq + 64 * i;
            const int oy = ty0 + (pl >> p.tw_log2), ox = tx0 + (pl & (TW - 1));
            // unconditional load from a clamped address + validity bit: four back-to-back loads, no exec juggling
            const bool ok = oy < p.H && ox < p.W && c < p.Cout;
            pdy[i] = gload_vec16((const T*)p.dy + (ok ? ((size_t)(n * p.H + oy) * p.W + ox) * p.Cout + c : 0));
            dm |= (ok ? 1 : 0) << i;
        }
        dymask = dm;
        if constexpr (NH > 0) {
            int which = 0, cs = c_in;
            if constexpr (NH == 1) {
                if (p.nsrc > 1 && c_in >= p.src[0].C) { which = 1; cs = c_in - p.src[0].C; }
            }
            if (cs >= p.src[which].C) cs = -1;
            load_affine<VEC>(p.src[which], n, cs, sc, sh);
            if constexpr (NH == 2) load_affine<VEC>(p.src[1], n, cs, sc1, sh1);
            pmode = p.src[which].mode;
            const T* base = (const T*)p.src[which].ptr;
            int mask = 0;
            const int csafe = cs >= 0 ? cs : 0;
            if constexpr (NH == 1) pslope = pmode == MRISR_SRC_NORM ? LRELU_SLOPE : (pmode == MRISR_SRC_RELU ? 0.f : 1.f);
#pragma unroll
            for (int i = 0; i < kWgSlots; ++i) {
                if constexpr (NH == 1) {
                    const int o = which ? geom.off1[i] : geom.off0[i];
                    const bool ok = o >= 0 && cs >= 0;
                    ph[i][0] = gload_vec16(base + (ok ? o : 0) + csafe);
                    mask |= (ok ? 1 : 0) << i;
                } else {
                    const int o0 = geom.off0[i], o1 = geom.off1[i];
                    const bool ok = o0 >= 0 && o1 >= 0 && cs >= 0;
                    ph[i][0] = gload_vec16(base + (ok ? o0 : 0) + csafe);
                    ph[i][1] = gload_vec16((const T*)p.src[1].ptr + (ok ? o1 : 0) + csafe);
                    mask |= (ok ? 1 : 0) << i;
                }
            }
            pmask = mask;
        }
    };
    // every source stored as-is (materialised pooled / up-sampled / blended inputs): no transform in the commit
    const bool all_raw = p.src[0].mode == MRISR_SRC_RAW && (p.nsrc < 2 || p.src[1].mode == MRISR_SRC_RAW);
    auto commit = [&](char* buf, int n) {
        if (DBG(p) & 2) return;
        char* lds_dy = buf;
        char* lds_in = buf + 256 * 128;
        int tq = t >> 3;                       // opaque: the 10 LDS store addresses are recomputed, not kept live
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = tq + 64 * i;
            Vec16<T> v = pdy[i];
            if (!((dymask >> i) & 1)) v.zero();
            *reinterpret_cast<decltype(v.v)*>(lds_dy + lds_off128(pl, ch8 >> 2, ch8 & 3)) = v.v;
        }
        if constexpr (NH > 0) {
#pragma unroll
            for (int i = 0; i < kWgSlots; ++i) {
                Vec16<T> v = ph[i][0];
                if constexpr (NH == 1) {     // straight-line: y = x*sc+sh, act = max(y, slope*y)
                    if (!all_raw) {          // (scalar branch: sources stored as-is go to LDS as loaded)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float y = fmaf(v.get(e), sc[e], sh[e]);
                            v.set(e, fmaxf(y, pslope * y));
                        }
                    }
                } else {
                    float fa[VEC], fb[VEC];
                    transform_f(ph[i][0], fa, p.src[0].mode, sc, sh);
                    transform_f(ph[i][1], fb, p.src[1].mode, sc1, sh1);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v.set(e, blend_a * fa[e] + (1.f - blend_a) * fb[e]);
                }
                if (!((pmask >> i) & 1)) v.zero();
                // 3x3: slots 0..4 exist for every tile shape (halo >= 320 pixels), 1x1: slots 0..3 (256 pixels)
                if (i < (KS == 3 ? kWgSlots - 1 : 4) || (FAST ? tq + 64 * i < 340 : hyx[i] >= 0))
                    *reinterpret_cast<decltype(v.v)*>(lds_in + lds_off128(tq + 64 * i, ch8 >> 2, ch8 & 3)) = v.v;
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // gathers (pool / bilinear): load + transform + store here, one slot at a time
            int cs = c_in;
            if (cs >= p.src[0].C) cs = -1;
            load_affine<VEC>(p.src[0], n, cs, sc, sh);
#pragma unroll
            for (int i = 0; i < kWgSlots; ++i) {
                const Vec16<T> v = halo_load<T, GSP>(geom, i, c_in, p, sc, sh, sc, sh, blend_a, 0, cs);
                if (hyx[i] >= 0)
                    *reinterpret_cast<decltype(v.v)*>(lds_in + lds_off128((t >> 3) + 64 * i, ch8 >> 2, ch8 & 3)) = v.v;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // Schedule.  The two waves that share a SIMD (tap groups 0 and 1) run the iteration in opposite orders, so that
    // one is in its MFMA block while the other does the VALU/LDS-store work of staging the next tile:
    //   tap group 0:  MFMA(tile k) -> commit(tile k+1) -> issue loads(tile k+2)
    //   tap group 1:  commit(tile k+1) -> issue loads(tile k+2) -> MFMA(tile k)
    // Either way the operands of tile k+1 were requested a full iteration earlier (the commit does not stall on
    // vmcnt) and sit in registers during MFMA(k).  Each thread stages its own slots of the double-buffered LDS image;
    // the barrier at the end of iteration k separates the writes of buffer (k+1)&1 from its reads in iteration k+1
    // and from its last reads in k-1.
    int tile = by, n = 0, ty0 = 0, tx0 = 0, cur = 0;
    int pn = 0, pty0 = 0, ptx0 = 0;                           // group 1: tile whose operands are in registers
    if (tile < total_tiles) {
        decode(tile, n, ty0, tx0);
        set_geom(n, ty0, tx0);
        issue(n, ty0, tx0);
        commit(smem, n);
        if (tile + (int)gridDim.y < total_tiles) {
            decode(tile + gridDim.y, pn, pty0, ptx0);
            set_geom(pn, pty0, ptx0);
            issue(pn, pty0, ptx0);
        }
    }
    __syncthreads();
    // Fast path of the MFMA block (bf16, 3x3, 8x32 tiles, every wave owns a full fragment pair): fully unrolled over
    // the 16 k-steps with the halo width known at compile time, so that every fragment address is one of a few
    // per-lane base registers plus an immediate - the generic loop below spends ~45 VALU instructions of address
    // arithmetic per k-step (5 MFMAs) and is VALU-bound.
    //   B row of (k-step ks, tap): h0 = K + lp, K = (ks>>1)*34 + (ks&1)*16 + tapoff (compile time), lp = 8*lh + gq;
    //   lds_off128's row swizzle ((h0>>1)&1) depends on K only through K mod 4 -> four per-lane bases.
    static_assert(!FAST || (kBf16 && KS == 3), "FAST wgrad variant: bf16 3x3 only");
    int a_lane = 0, b_lane[4] = {0, 0, 0, 0};
    if constexpr (kBf16) {
        const int li = lane & 15, gq = li >> 2, gp = li & 3, gr = (lane >> 4) & 1;
        const int chb = 16 * gr + 4 * gp;
        const int lp = 8 * lh + gq;
        const int cbits = (((chb >> 3) & 3) << 4) + ((chb & 4) << 1);
        a_lane = lp * 128 + ((fo ^ ((gq >> 1) & 1)) << 6) + cbits;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_lane[kk] = 256 * 128 + lp * 128 + ((fi ^ (((kk + gq) >> 1) & 1)) << 6) + cbits;
    }
    auto mfma_fast = [&](auto tg_tag) {
        constexpr int TG = decltype(tg_tag)::value;
        constexpr int HWC = 32 + 2 * PAD;
        constexpr int KP = FAST > 0 ? FAST : 1;           // this wave takes k-steps kpart, kpart + KP, ...
        constexpr int NTG = TG ? NTAPS - NT0 : NT0;       // taps of this group
        constexpr int NS = (16 / KP) * NTG;               // (k-step, tap) steps, one MFMA each
        const char* bufp = smem + cur * buf_bytes;
        // wave part of the halo row K (k-step offset kpart): folded into the bases; it also rotates the swizzle class
        const int kw = KP == 2 ? kpartu * 16 : (KP == 4 ? (kpartu >> 1) * HWC + (kpartu & 1) * 16 : 0);
        const char* ab = bufp + a_lane + kpartu * 2048;
        const char* bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cls = (j + kw) & 3;      // scalar
            bb[j] = bufp + kw * 128 + (cls == 0 ? b_lane[0] : cls == 1 ? b_lane[1] : cls == 2 ? b_lane[2] : b_lane[3]);
        }
        auto load_a = [&](int i) { return tr_read_frag<T>(ab + i * KP * 2048, ab + i * KP * 2048 + 512); };
        auto load_b = [&](int st) {
            const int ks = (st / NTG) * KP, tap = (TG ? NT0 : 0) + st % NTG;
            const int K = (ks >> 1) * HWC + (ks & 1) * 16 + (tap / KS) * HWC + (tap % KS);
            const char* b = bb[K & 3] + K * 128;
            return tr_read_frag<T>(b, b + 512);
        };
        // software pipeline: the B fragment of step s+2 and the A fragment of the next k-step are requested before
        // the MFMA of step s is issued, so two LDS reads are always in flight behind the matrix pipe
        typename Frag16<T>::type aq[2], bq[3];
        aq[0] = load_a(0);
        bq[0] = load_b(0);
        bq[1] = load_b(1);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int i = st / NTG, j = st % NTG;
            if (st + 2 < NS) bq[(st + 2) % 3] = load_b(st + 2);
            if (j == 0 && (i + 1) * KP < 16) aq[(i + 1) & 1] = load_a(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[j] = Frag16<T>::mma(aq[i & 1], bq[st % 3], acc[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mfma_block = [&](auto tg_tag) {      // tg_tag: the (wave-uniform) tap group of the calling site
        const char* lds_dy = smem + cur * buf_bytes;
        const char* lds_in = lds_dy + 256 * 128;
        if (DBG(p) & 8) {
        } else if constexpr (FAST) {
            if constexpr (decltype(tg_tag)::value == 0) mfma_fast(std::integral_constant<int, 0>{});
            else mfma_fast(std::integral_constant<int, 1>{});
        } else if constexpr (kBf16) {
            // generic path (small images / channel blocks narrower than 64): keep its per-lane address pieces from
            // being hoisted out of the persistent loop, where they would cost the fast path ~25 live VGPRs
            int lane_o = lane, lh_o = lh;
            asm volatile("" : "+v"(lane_o), "+v"(lh_o));
            const int li = lane_o & 15, gq = li >> 2, gp = li & 3, gr = (lane_o >> 4) & 1;
            const int chb = 16 * gr + 4 * gp;            // channel inside the 32-wide fragment
            const int c_dy = fo * 32 + chb, c_b = fi * 32 + chb;
            // 1x1: both groups take the single tap and split the k-steps
            const int ks_lo = (NTAPS == 1 ? 8 * tg : 0) + kpart, ks_hi = NTAPS == 1 ? 8 * tg + 8 : 16;
#pragma unroll 2
            for (int ks = ks_lo; ks < ks_hi; ks += kparts) {
                const int pk = ks * 16 + 8 * lh_o + gq;  // pixel (second read: +4)
                const typename Frag16<T>::type af = tr_read_frag<T>(
                    lds_dy + lds_off128(pk, c_dy >> 5, (c_dy >> 3) & 3) + ((c_dy & 4) << 1),
                    lds_dy + lds_off128(pk + 4, c_dy >> 5, (c_dy >> 3) & 3) + ((c_dy & 4) << 1));
                const int hp = (pk >> p.tw_log2) * hw + (pk & (TW - 1));   // pk and pk+4 share the row
#pragma unroll
                for (int j = 0; j < NT0; ++j) {
                    const int tap = NTAPS == 1 ? 0 : (tg ? NT0 + j : j);
                    if (tap < NTAPS) {
                        const int h0 = hp + (tap / KS) * hw + (tap % KS);
                        const typename Frag16<T>::type bfr = tr_read_frag<T>(
                            lds_in + lds_off128(h0, c_b >> 5, (c_b >> 3) & 3) + ((c_b & 4) << 1),
                            lds_in + lds_off128(h0 + 4, c_b >> 5, (c_b >> 3) & 3) + ((c_b & 4) << 1));
                        acc[j] = Frag16<T>::mma(af, bfr, acc[j]);
                    }
                }
            }
        } else {
            // exact fp32: one 32x32x2 MFMA per pixel pair and tap; wave (tg, kq) takes pairs kq, kq+4, ...
            const int kq = wave & 3;
            const int kp_lo = NTAPS == 1 ? 64 * tg + kq : kq, kp_hi = NTAPS == 1 ? 64 * tg + 64 : 128;
            for (int kp = kp_lo; kp < kp_hi; kp += 4) {
                const int pk = 2 * kp + lh;
                const float a = *reinterpret_cast<const float*>(lds_dy + lds_off128(pk, lr >> 4, (lr >> 2) & 3) + ((lr & 3) << 2));
                const int hp = (pk >> p.tw_log2) * hw + (pk & (TW - 1));
#pragma unroll
                for (int j = 0; j < NT0; ++j) {
                    const int tap = NTAPS == 1 ? 0 : (tg ? NT0 + j : j);
                    if (tap < NTAPS) {
                        const int h0 = hp + (tap / KS) * hw + (tap % KS);
                        const float bb = *reinterpret_cast<const float*>(lds_in + lds_off128(h0, lr >> 4, (lr >> 2) & 3) + ((lr & 3) << 2));
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[j], 0, 0, 0);
                    }
                }
            }
        }
    };
    while (tile < total_tiles) {
        const int nxt = tile + gridDim.y;
        const bool has_next = nxt < total_tiles;
        if (tgu == 0) mfma_block(std::integral_constant<int, 0>{});
        if (has_next) commit(smem + (cur ^ 1) * buf_bytes, pn);      // operands loaded during the previous iteration
        const int it = nxt + (int)gridDim.y;                          // start loading the tile after next
        if (it < total_tiles) {
            decode(it, pn, pty0, ptx0);
            set_geom(pn, pty0, ptx0);
            issue(pn, pty0, ptx0);                  // a full iteration (~2 us) for the loads to land
        }
        if (tgu == 1) mfma_block(std::integral_constant<int, 1>{});
        __syncthreads();
        cur ^= 1;
        tile = nxt;
    }

    // ---- two-stage reduction: dump the accumulators as they stand (lane-contiguous 256-B stores) into this
    // workgroup's slab; wgrad_reduce_kernel sums the split-K slabs.  The float atomics below run at the memory side
    // at ~1.3 TB/s chip-wide: 38 MB of them per launch cost 25-36 us, the slab stores + the reduction about half.
    if (p.wsp) {
        if (!(DBG(p) & 1)) {
            float* wsb = p.wsp + ((size_t)(by * gridDim.x + bx) * 8 + wave) * (NT0 * 16 * 64) + lane;
#pragma unroll
            for (int j = 0; j < NT0; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) gstore(wsb + (j * 16 + r) * 64, acc[j][r]);
        }
        return;
    }
    // ---- accumulate into dW[co][tap][ci]: lane = ci (128-B contiguous per half wave), regs = co
    const int ci = ci0 + fi * 32 + lr;
    if (ci < p.Cin && !(DBG(p) & 1)) {
#pragma unroll
        for (int j = 0; j < NT0; ++j) {
            const int tap = NTAPS == 1 ? 0 : (tg ? NT0 + j : j);
            if (tap < NTAPS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + fo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < p.Cout) atomic_add_f32(p.dw + ((size_t)co * NTAPS + tap) * p.Cin + ci, acc[j][r]);
                }
            }
        }
    }
}

// Second stage: dw[co][tap][ci] += sum over the split-K slabs.  Thread = one accumulator element (block, wave,
// register, lane) of the first stage's layout, mapped back with the same role arithmetic.
constexpr int kRedChunk = 8;     // split-K slabs summed per thread of the second stage
template <int BC, bool kBf16>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk,
                                                           int ksplit, int Cout, int Cin, int KS) {
    const int NTAPS = KS * KS, NT0 = (NTAPS + 1) / 2, per_wave = NT0 * 16 * 64, per_blk = 8 * per_wave;
    const size_t total = (size_t)nblk * per_blk;
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // blockIdx.y = chunk of kRedChunk slabs: enough threads (and independent loads) in flight also when the channel
    // block count is 1 and the split-K factor is 256
    float s = 0.f;
    const int k0 = blockIdx.y * kRedChunk;
#pragma unroll
    for (int k = 0; k < kRedChunk; ++k)
        if (k0 + k < ksplit) s += ws[(size_t)(k0 + k) * total + idx];
    const int blk = idx / per_blk, rem = idx - (size_t)blk * per_blk;
    const int wave = rem / per_wave, jr = (rem - wave * per_wave) >> 6, lane = rem & 63;
    const int j = jr >> 4, r = jr & 15, lr = lane & 31, lh = lane >> 5;
    const int ncib = (Cin + BC - 1) / BC, cib = blk % ncib, cob = blk / ncib, co0 = cob * BC, ci0 = cib * BC;
    const int nfo = kBf16 ? (min(Cout - co0, BC) > 32 ? 2 : 1) : 1, nfi = kBf16 ? (min(Cin - ci0, BC) > 32 ? 2 : 1) : 1;
    const int npairs = nfo * nfi, pair = (wave & 3) % npairs, tg = wave >> 2;
    const int fo = pair / nfi, fi = pair % nfi;
    const int tap = NTAPS == 1 ? 0 : (tg ? NT0 + j : j);
    const int co = co0 + fo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, ci = ci0 + fi * 32 + lr;
    if (tap < NTAPS && co < Cout && ci < Cin && s != 0.f) atomic_add_f32(dw + ((size_t)co * NTAPS + tap) * Cin + ci, s);
}

#ifndef MRISR_KERNEL_ONLY
int conv_fill_params(const mrisr_conv_desc* d, ConvParams& p, const char* who);
int num_cus();
// conv_wgrad_rows.hip: the row-streaming producer / consumer kernel (16-bit, 3x3, 64 x 64 channel blocks)
bool conv_wgrad_rows_ok(const mrisr_conv_desc* d);
size_t conv_wgrad_rows_workspace_floats(const ConvParams& p);
int launch_wgrad_rows(int dtype, ConvParams& p, size_t ws_floats, hipStream_t s);

static void wgrad_grid(const ConvParams& p, int BC, int& nblk, int& ksplit) {
    nblk = ceil_div(p.Cout, BC) * ceil_div(p.Cin, BC);
    const int total_tiles = p.N * p.tiles_y * p.tiles_x;
    ksplit = ceil_div(p.cus > 0 ? p.cus : num_cus(), nblk);          // one 8-wave workgroup per CU (LDS-limited)
    if (ksplit > total_tiles) ksplit = total_tiles;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 65535) ksplit = 65535;
}

static thread_local size_t p_ws_floats = 0;      // capacity of the caller's workspace for the current call

template <typename T, int SPATIAL, int KS, int FAST = 0>
static int launch_wgrad(ConvParams& p, hipStream_t s) {
    constexpr int BC = WgTraits<T>::BC;
    const int TW = 1 << p.tw_log2, pad = KS / 2;
    const size_t lds = 2 * (256 * 128 + (size_t)(TW + 2 * pad) * (p.th + 2 * pad) * 128);
    int nblk, ksplit;
    wgrad_grid(p, BC, nblk, ksplit);
    // two-stage reduction only where it pays: several slabs per output and a workspace that holds them
    const size_t need = (size_t)nblk * ksplit * 8 * ((KS * KS + 1) / 2) * 16 * 64;
    if (!(p.wsp && ksplit >= 4 && need <= p_ws_floats)) p.wsp = nullptr;
    auto kern = conv_wgrad_kernel<T, SPATIAL, KS, FAST>;
    static std::once_flag attr_once;   // per instantiation
    std::call_once(attr_once, [kern] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(kern, dim3(nblk, ksplit), dim3(kWgThreads), lds, s, p);
    MRISR_CHECK_LAUNCH("conv_wgrad");
    if (p.wsp) {
        const size_t total = need / ksplit;
        wgrad_reduce_kernel<BC, sizeof(T) == 2><<<dim3((unsigned)((total + 255) / 256), ceil_div(ksplit, kRedChunk)), 256, 0, s>>>(p.wsp, p.dw, nblk, ksplit, p.Cout, p.Cin, KS);
        MRISR_CHECK_LAUNCH("conv_wgrad(reduce)");
    }
    return MRISR_OK;
}

template <typename T>
static int dispatch_wgrad(ConvParams& p, int spatial, int ks, hipStream_t s) {
    if (p.combine == MRISR_COMBINE_BLEND) {
        if (ks != 3) MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_wgrad: blend needs a 3x3 conv");
        return launch_wgrad<T, kLoaderBlendW, 3>(p, s);
    }
    if (ks == 3) {
        if constexpr (sizeof(T) == 2) {
            const int fast = conv_wgrad_fast(TypeTraits<T>::kDtype, spatial, 3, p.tw_log2, p.Cout, p.Cin);
            if (fast == 1) return launch_wgrad<T, MRISR_SP_NONE, 3, 1>(p, s);
            if (fast == 2) return launch_wgrad<T, MRISR_SP_NONE, 3, 2>(p, s);
            if (fast == 4) return launch_wgrad<T, MRISR_SP_NONE, 3, 4>(p, s);
        }
        if (spatial == MRISR_SP_NONE) return launch_wgrad<T, MRISR_SP_NONE, 3>(p, s);
        if (spatial == MRISR_SP_POOL2) return launch_wgrad<T, MRISR_SP_POOL2, 3>(p, s);
        return launch_wgrad<T, MRISR_SP_UP2, 3>(p, s);
    }
    if (spatial == MRISR_SP_NONE) return launch_wgrad<T, MRISR_SP_NONE, 1>(p, s);
    if (spatial == MRISR_SP_UP2) return launch_wgrad<T, MRISR_SP_UP2, 1>(p, s);
    MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_wgrad: 1x1 conv with pooled source");
}

extern "C" size_t mrisr_conv_wgrad_workspace_floats(const mrisr_conv_desc* d) {
    ConvParams p;
    if (conv_fill_params(d, p, "conv_wgrad_workspace_floats")) return 0;
    const int BC = d->dtype == MRISR_F32 ? 32 : 64;
    int nblk, ksplit;
    wgrad_grid(p, BC, nblk, ksplit);
    const size_t classic = (size_t)nblk * ksplit * 8 * ((d->ksize * d->ksize + 1) / 2) * 16 * 64;
    if (!conv_wgrad_rows_ok(d)) return classic;
    const size_t rows = conv_wgrad_rows_workspace_floats(p);
    return rows > classic ? rows : classic;
}

extern "C" int mrisr_conv_wgrad(const mrisr_conv_desc* d, const void* dy, float* dw, float* workspace,
                                size_t workspace_floats, void* stream) {
    ConvParams p;
    int rc = conv_fill_params(d, p, "conv_wgrad");
    if (rc) return rc;
    if (!dy || !dw) MRISR_FAIL(MRISR_E_ARG, "conv_wgrad: null dy/dw");
    const int vec = mrisr_vec(d->dtype);
    if (d->Cout % vec) MRISR_FAIL(MRISR_E_SHAPE, "conv_wgrad: Cout %d not a multiple of %d", d->Cout, vec);
    if ((size_t)d->N * d->H * d->W * d->Cout >= (1ull << 31)) MRISR_FAIL(MRISR_E_SHAPE, "conv_wgrad: dy exceeds 2^31 elements");
    p.dy = dy;
    p.dw = dw;
    p.wsp = workspace;
    p_ws_floats = workspace ? workspace_floats : 0;
    hipStream_t s = (hipStream_t)stream;
    if (conv_wgrad_rows_ok(d)) return launch_wgrad_rows(d->dtype, p, p_ws_floats, s);
    if (d->dtype == MRISR_BF16) return dispatch_wgrad<bf16_t>(p, d->src[0].spatial, d->ksize, s);
    if (d->dtype == MRISR_F16) return dispatch_wgrad<f16_t>(p, d->src[0].spatial, d->ksize, s);
    return dispatch_wgrad<float>(p, d->src[0].spatial, d->ksize, s);
}
#endif  // MRISR_KERNEL_ONLY
