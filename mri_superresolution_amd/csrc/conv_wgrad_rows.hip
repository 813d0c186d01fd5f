// Weight gradient of the 3x3 convolution on MFMA, ROW-STREAMING form (gfx950): 16-bit storage, channel blocks of 64 x 64.
//
// Replaces aten convolution_backward's weight branch for the nn.Conv2d layers of /root/reference/models/unet_model.py
// (:29,34,101,152,168), like conv_wgrad.hip, whose generic kernel keeps the 1x1 / 32-channel / fp32 / gather cases.
//   dW[co][ky][kx][ci] += sum_pixels dy[y][x][co] * in[y + ky - 1][x + kx - 1][ci]
//
// What conv_wgrad_kernel<T,0,3,1> spends its time on (profiles/NOTES.md R2-12, R3-2): every wave owns ONE 32 x 32 output
// fragment per tap, so a B fragment is read from LDS for every MFMA (1.2 KB of transposing reads per MFMA = ~60 % of the
// LDS array at the full MFMA rate, plus 76 KB of staging writes per tile), and the same eight waves that run the MFMAs also
// load, GroupNorm+LeakyReLU-transform and store the next tile.  Two changes:
//   * vertical tap re-use.  A tile is 16 x 16 pixels and a k-step is ONE tile row (16 pixels).  The B fragment of halo row h
//     and column shift kx serves the taps (ky, kx) of pixel rows h - ky, ky = 0..2, and the A fragment (dy) of pixel row y
//     serves halo rows y .. y + 2: a wave that owns all nine taps of one (co, ci) fragment pair loads ONE new A and THREE
//     new B fragments per halo row for NINE MFMAs - 0.44 instead of 1.2 KB of LDS reads per MFMA;
//   * producer / consumer waves.  Waves 0-3 (one per SIMD, the older ones: they win the matrix-pipe arbitration) only read
//     fragments and issue MFMAs - 144 accumulator registers each; waves 4-7 only stage: global loads of the tile after
//     next into registers, GroupNorm + LeakyReLU on the tile that has arrived, LDS stores.  One barrier per tile.
#include <mutex>
#include <type_traits>

#include "conv_common.h"

// Ablation switches of tuning builds (tools/build_src_variant.sh; results invalid by construction, timing only):
// 1 no MFMA, 2 no GroupNorm / LeakyReLU transform, 4 no global loads, 8 no LDS stores.  0 in the product library.
#ifndef MRISR_WR_DBG
#define MRISR_WR_DBG 0
#endif
// Phase profile (tuning builds only, -DMRISR_WR_PT; tools/conv_bench.py prints it): s_memtime stamps per wave of the middle
// workgroup: slot 6 = work (MFMA block / commit + issue), slot 5 = barrier wait, slot 10 = 100 MHz ticks of the loop.
#ifdef MRISR_WR_PT
__device__ unsigned long long g_wr_cycles[8][12];
#define WPT_DECL unsigned long long pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime(); const unsigned long long pt_r0 = __builtin_amdgcn_s_memrealtime();
#define WPT_MARK(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_now = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_now - pt_t; pt_t = pt_now; __builtin_amdgcn_sched_barrier(0); }
#define WPT_DUMP() if (blockIdx.x + blockIdx.y * gridDim.x == (gridDim.x * gridDim.y) / 2 && lane == 0) { pt_acc[10] = __builtin_amdgcn_s_memrealtime() - pt_r0; for (int k = 0; k < 12; ++k) atomicAdd(&g_wr_cycles[wave][k], pt_acc[k]); if (threadIdx.x == 0) atomicAdd(&g_wr_cycles[0][11], 1ull); }
#else
#define WPT_DECL
#define WPT_MARK(k)
#define WPT_DUMP()
#endif
constexpr int kWrThreads = 512;
constexpr int kWrDyBytes = 256 * 128;          // [16 x 16 px][64 co] 128-B rows
constexpr int kWrInRows = 18 * 18;             // halo tile
constexpr int kWrInBytes = 352 * 128;          // [18 x 18 px][64 ci] (+ 28 rows so that the last staging slot stores unpredicated)
constexpr int kWrBuf = kWrDyBytes + kWrInBytes;
// Channel block = 32 NCO output x 32 NCI input channels (NCO, NCI = 1 or 2; LDS rows keep their 128-byte pitch).  The four
// MFMA waves own the NCO * NCI (co, ci) fragment pairs; with fewer than four pairs they split the tile's 16 pixel rows
// between them (a wave then streams 16 / parts rows + 2 halo rows and the split-K reduction adds the parts up).
constexpr int kWrAcc = 9 * 16 * 64;            // accumulator floats per consumer wave (slab layout of the two-stage reduction)

template <typename T>
__device__ __forceinline__ typename Frag16<T>::type wr_tr_read(const char* base0, const char* base1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base1);
    union { struct { s16x4 a, b; } s; typename Frag16<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

template <typename T, int NCO, int NCI, bool RAW>
__global__ __launch_bounds__(kWrThreads, 2) void conv_wgrad_rows_kernel(const ConvParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams p = pin_params(p_in);
    constexpr int VEC = 8;
    constexpr int BCO = 32 * NCO, BCI = 32 * NCI;
    constexpr int NPAIRS = NCO * NCI, NPARTS = 4 / NPAIRS, RP = 16 / NPARTS;     // fragment pairs, row parts, rows per part
    // staging: 16-B vectors per thread - dy 256 px x 4 NCO chunks, halo 324 px x 4 NCI chunks over 256 threads
    constexpr int DYCH = 4 * NCO, XCH = 4 * NCI;
    constexpr int DYSTEP = 256 / DYCH, XSTEP = 256 / XCH;
    constexpr int kWrDySlots = DYCH, kWrInSlots = (kWrInRows + XSTEP - 1) / XSTEP;
    typedef typename Frag16<T>::type frag_t;
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool consumer = wave < 4;

    const int ncib = p.Cin / BCI;
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-aware order (conv_wgrad.hip): the channel-block pairs of one k-slice share an L2
        const int G = gridDim.x * gridDim.y;
        if ((G & 7) == 0) {
            const int id = by * gridDim.x + bx, lid = (id & 7) * (G >> 3) + (id >> 3);
            by = lid / gridDim.x;
            bx = lid - by * gridDim.x;
        }
    }
    const int cib = bx % ncib, cob = bx / ncib;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x;

    auto decode = [&](int tile, int& n, int& ty0, int& tx0) {
        const int tx = tile % p.tiles_x;
        const int r = tile / p.tiles_x;
        n = r / p.tiles_y;
        ty0 = (r - n * p.tiles_y) * 16;
        tx0 = tx * 16;
    };

    // ------------------------------------------------------------------ schedule: one barrier per tile
    //   consumers:  MFMA(tile k)                                              from buffer k & 1
    //   producers:  commit(tile k+1) into buffer (k+1) & 1  ->  issue loads(tile k+2)
    // the operands of tile k+1 were requested a full iteration earlier and sit in the producers' registers during MFMA(k);
    // buffer (k+1) & 1 was last read in iteration k-1, before the barrier that ended it.  The two roles are two separate
    // loops with the same barrier sequence, so that neither role's registers are live in the other's code (one loop with a
    // role branch keeps 144 accumulators AND 92 staging registers alive for every wave: 343 spills).
    if (!consumer) {
#ifdef MRISR_WR_PRIO
        // the staging waves are the younger ones of their SIMD and lose the issue arbitration to the MFMA wave
        __builtin_amdgcn_s_setprio(MRISR_WR_PRIO);
#endif
        // ------------------------------------------------------------------ producer (waves 4-7): 256 staging threads
        // Software pipeline per SLOT: while tile k is being multiplied, slot i of tile k+1 (requested one iteration ago)
        // is transformed and stored to LDS and the same registers immediately take the load of slot i of tile k+2 - every
        // load has a whole iteration to arrive, whichever slot it is (with all loads issued behind the commit, the next
        // iteration's first use waited for loads that were a barrier old: measured 26 of 95 us).
        const int pt = t & 255;
        const int dch = pt % DYCH, drow = pt / DYCH;                  // dy: 16-B chunk of the row, first pixel of this thread
        const int xch = pt % XCH, xrow = pt / XCH;                    // halo: the same for the conv input
        // TWO register sets, two tiles ahead: a tile's loads are issued two iterations (~8 us) before its commit.  With one
        // set (one tile = 76 KB per CU in flight) the commit waited for its data every iteration - 19.5 MB in flight
        // chip-wide is ~4 us at the rate the memory system delivers this pattern, longer than a tile's MFMA time.
        struct StageSet {
            Vec16<T> dy[kWrDySlots], x[kWrInSlots];
            float sc[VEC], sh[VEC];        // GroupNorm affine of the tile's image (this thread's 8 channels)
            int ty0, tx0;                  // where the tile lies (edge handling at commit time)
        };
        StageSet S0, S1;
        const int c_out = co0 + dch * VEC, c_in = ci0 + xch * VEC;
        // conv input: the 64-channel block lies in ONE concat source (host-checked: src0.C is a multiple of 64) - workgroup-uniform
        const bool w1 = __builtin_amdgcn_readfirstlane((p.nsrc > 1 && ci0 >= p.src[0].C) ? 1 : 0) != 0;
        const int cs = w1 ? c_in - p.src[0].C : c_in;
        const int Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W, Cs = w1 ? p.src[1].C : p.src[0].C;
        const int offy = w1 ? p.src[1].off_y : p.src[0].off_y, offx = w1 ? p.src[1].off_x : p.src[0].off_x;
        const int pmode = w1 ? p.src[1].mode : p.src[0].mode;
        const float slope = pmode == MRISR_SRC_NORM ? LRELU_SLOPE : (pmode == MRISR_SRC_RELU ? 0.f : 1.f);
        // Loads are BUFFER loads over the whole tensor: offset = per-tile scalar part + per-slot part computed ONCE, one vector
        // add per load and no validity arithmetic in the common path (the staging waves' vector instructions run ~2.4x slower
        // beside the other wave's MFMAs, and addresses + masks + zero selects of 19 slots were as much work as the GroupNorm
        // transform itself).  An offset outside the tensor returns zeros (hardware range check); halo pixels beside the plane
        // alias into neighbouring rows - tiles that touch the plane's edge overwrite their invalid slots with zeros in a
        // second pass (a scalar branch with LDS stores only: no load inside a branch, see commit_issue).
        const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.dy), 0, (int)((unsigned)p.N * p.H * p.W * p.Cout * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(w1 ? p.src[1].ptr : p.src[0].ptr), 0, (int)((unsigned)p.N * Hs * Ws * Cs * 2u), 0x00020000);
        unsigned rel_dy[kWrDySlots], rel_x[kWrInSlots];
    #pragma unroll
        for (int i = 0; i < kWrDySlots; ++i) {
            const int px = drow + DYSTEP * i;
            rel_dy[i] = (unsigned)(((px >> 4) * p.W + (px & 15)) * p.Cout + c_out) * 2u;
        }
    #pragma unroll
        for (int i = 0; i < kWrInSlots; ++i) {
            const int hp = xrow + XSTEP * i;
            const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;          // hp / 18 for hp < 400
            rel_x[i] = (unsigned)((hy * Ws + hx) * Cs + cs) * 2u;
        }
        auto dy_tile_off = [&](int n, int ty0, int tx0) { return (unsigned)(((n * p.H + ty0) * p.W + tx0) * p.Cout) * 2u; };
        auto x_tile_off = [&](int n, int ty0, int tx0) {      // (may be "negative": the sum with a valid slot's part is not)
            return (unsigned)(((n * Hs + ty0 - 1 - offy) * Ws + tx0 - 1 - offx) * Cs) * 2u;
        };
        auto load_slot_dy = [&](StageSet& S, int i, unsigned tile_off) {
            if (MRISR_WR_DBG & 4) return;
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc, (int)(tile_off + rel_dy[i]), 0, 0);
            S.dy[i].v = __builtin_bit_cast(decltype(S.dy[i].v), r);
        };
        auto load_slot_x = [&](StageSet& S, int i, unsigned tile_off) {
            if (MRISR_WR_DBG & 4) return;
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(tile_off + rel_x[i]), 0, 0);
            S.x[i].v = __builtin_bit_cast(decltype(S.x[i].v), r);
        };
        auto load_aff = [&](StageSet& S, int n) {
            if (w1) load_affine<VEC>(p.src[1], n, cs, S.sc, S.sh);
            else load_affine<VEC>(p.src[0], n, cs, S.sc, S.sh);
        };
        auto load_all = [&](StageSet& S, int n, int ty0, int tx0) {
            const unsigned dyo = dy_tile_off(n, ty0, tx0), xo = x_tile_off(n, ty0, tx0);
    #pragma unroll
            for (int i = 0; i < kWrDySlots; ++i) load_slot_dy(S, i, dyo);
    #pragma unroll
            for (int i = 0; i < kWrInSlots; ++i) load_slot_x(S, i, xo);
            if (!RAW) load_aff(S, n);
            S.ty0 = ty0; S.tx0 = tx0;
        };
        // Stores the tile held by set S into `buf` and refills every slot of S with tile (n, ty0, tx0).  STRAIGHT-LINE on
        // purpose: no branch around a load (the tail re-loads the last tile and stores a tile nobody reads) - with branches
        // around the loads hipcc gives up counting and waits `vmcnt(0)` / `vmcnt(1)` before EVERY slot, i.e. for the load it
        // has just issued.
        // (sources stored as-is run the same arithmetic with scale 1, shift 0, slope 1: exact for 16-bit values)
        auto commit_issue = [&](StageSet& S, char* buf, int n, int ty0, int tx0) {
            char* lds_dy = buf;
            char* lds_in = buf + kWrDyBytes;
            int dq = drow, xq = xrow;
            asm volatile("" : "+v"(dq), "+v"(xq));      // slot rows are recomputed, not kept live
            const unsigned dyo = dy_tile_off(n, ty0, tx0), xo = x_tile_off(n, ty0, tx0);
            const int pty0 = S.ty0, ptx0 = S.tx0;
            float sc[VEC], sh[VEC];
    #pragma unroll
            for (int e = 0; e < VEC; ++e) { sc[e] = S.sc[e]; sh[e] = S.sh[e]; }
    #pragma unroll
            for (int i = 0; i < kWrDySlots; ++i) {
                if (!(MRISR_WR_DBG & 8)) *reinterpret_cast<decltype(S.dy[i].v)*>(lds_dy + lds_off128(dq + DYSTEP * i, dch >> 2, dch & 3)) = S.dy[i].v;
                else asm volatile("" ::"v"(S.dy[i].v));
                load_slot_dy(S, i, dyo);
            }
    #pragma unroll
            for (int i = 0; i < kWrInSlots; ++i) {
                Vec16<T> v = S.x[i];
                if (!RAW && !(MRISR_WR_DBG & 2)) {      // (sources stored as-is - pooled / up-sampled / blended inputs - go to LDS as loaded)
    #pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float y = fmaf(v.get(e), sc[e], sh[e]);
                        v.set(e, fmaxf(y, slope * y));
                    }
                }
                // (rows 324 .. of the last slot are never read; the image holds 352 rows)
                if (i < kWrInSlots - 1 || xq + XSTEP * i < 352) {
                    if (!(MRISR_WR_DBG & 8)) *reinterpret_cast<decltype(v.v)*>(lds_in + lds_off128(xq + XSTEP * i, xch >> 2, xch & 3)) = v.v;
                    else asm volatile("" ::"v"(v.v));
                }
                load_slot_x(S, i, xo);
            }
            if (!RAW) load_aff(S, n);
            S.ty0 = ty0; S.tx0 = tx0;
            // edge tiles: zeros over the slots that lie outside the plane / the source (conv padding, partial tiles)
            const int ys0 = pty0 - 1 - offy, xs0 = ptx0 - 1 - offx;
            const bool interior = pty0 + 16 <= p.H && ptx0 + 16 <= p.W && ys0 >= 0 && xs0 >= 0 && ys0 + 18 <= Hs && xs0 + 18 <= Ws;
            if (!interior) {
                Vec16<T> z;
                z.zero();
    #pragma unroll
                for (int i = 0; i < kWrDySlots; ++i) {
                    const int px = dq + DYSTEP * i;
                    if (!(pty0 + (px >> 4) < p.H && ptx0 + (px & 15) < p.W))
                        *reinterpret_cast<decltype(z.v)*>(lds_dy + lds_off128(px, dch >> 2, dch & 3)) = z.v;
                }
    #pragma unroll
                for (int i = 0; i < kWrInSlots; ++i) {
                    const int hp = xq + XSTEP * i;
                    const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;
                    const unsigned y = ys0 + hy, x = xs0 + hx;
                    if (hp < kWrInRows && !(y < (unsigned)Hs && x < (unsigned)Ws))
                        *reinterpret_cast<decltype(z.v)*>(lds_in + lds_off128(hp, xch >> 2, xch & 3)) = z.v;
                }
            }
        };

        const int last_tile = total_tiles > by ? by + ((total_tiles - 1 - by) / (int)gridDim.y) * (int)gridDim.y : by;   // this workgroup's last tile
        // schedule (G = gridDim.y, tile sequence t0 = by, t1 = t0 + G, ...; indices beyond the last tile re-load the last tile):
        //   prologue: S0 <- t0, S1 <- t1; commit S0 -> buffer 0 and S0 <- t2; barrier
        //   iteration k (consumers multiply t_k from buffer k & 1): commit S[(k+1) & 1] = t_{k+1} -> buffer (k+1) & 1, refill <- t_{k+3}
        const int G = (int)gridDim.y;
        int tile = by, cur = 0;
        int n = 0, ty0 = 0, tx0 = 0;
        auto tile_at = [&](int idx) { return min(idx, last_tile); };
        if (tile < total_tiles) {
            decode(tile, n, ty0, tx0);
            load_all(S0, n, ty0, tx0);
            decode(tile_at(tile + G), n, ty0, tx0);
            load_all(S1, n, ty0, tx0);
            decode(tile_at(tile + 2 * G), n, ty0, tx0);
            commit_issue(S0, smem, n, ty0, tx0);
        }
        __syncthreads();
        WPT_DECL
        while (tile < total_tiles) {
            // even step: S1 holds the next tile
            decode(tile_at(tile + 3 * G), n, ty0, tx0);
            commit_issue(S1, smem + (cur ^ 1) * kWrBuf, n, ty0, tx0);
            WPT_MARK(6)
            __syncthreads();
            WPT_MARK(5)
            cur ^= 1;
            tile += G;
            if (tile >= total_tiles) break;
            // odd step: S0
            decode(tile_at(tile + 3 * G), n, ty0, tx0);
            commit_issue(S0, smem + (cur ^ 1) * kWrBuf, n, ty0, tx0);
            WPT_MARK(6)
            __syncthreads();
            WPT_MARK(5)
            cur ^= 1;
            tile += G;
        }
        WPT_DUMP()
        return;
    }

    // ------------------------------------------------------------------ consumer state (waves 0-3): fragment pair (fo, fi)
    const int pair = wave % NPAIRS, rpart = wave / NPAIRS;      // fragment pair and row part of this wave
    const int fo = pair / NCI, fi = pair % NCI;
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // per-lane fragment addresses of the transposing reads (layout lds_off128, as conv_wgrad.hip's fast path):
    //   A (dy, pixel row y of the tile): a_lane + y * 2048 (second read + 512 = 4 pixels on)
    //   B (halo pixel K .. K + 15, K = h * 18 + kx): b_lane[K & 3] + K * 128
    int a_lane, b_lane[4];
    {
        const int li = lane & 15, gq = li >> 2, gp = li & 3, gr = (lane >> 4) & 1;
        const int chb = 16 * gr + 4 * gp;
        const int lp = 8 * lh + gq;
        const int cbits = (((chb >> 3) & 3) << 4) + ((chb & 4) << 1);
        a_lane = lp * 128 + ((fo ^ ((gq >> 1) & 1)) << 6) + cbits;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_lane[kk] = kWrDyBytes + lp * 128 + ((fi ^ (((kk + gq) >> 1) & 1)) << 6) + cbits;
    }
    // ------------------------------------------------------------------ the MFMA block of one tile (consumer waves)
    auto mfma_tile = [&](const char* bufp) {
        const char* ab = bufp + a_lane;
        const char* bb[4] = {bufp + b_lane[0], bufp + b_lane[1], bufp + b_lane[2], bufp + b_lane[3]};
        auto load_a = [&](int y) { return wr_tr_read<T>(ab + y * 2048, ab + y * 2048 + 512); };
        auto load_b = [&](int h, int kx) {
            const int K = h * 18 + kx;
            const char* b = bb[K & 3] + K * 128;
            return wr_tr_read<T>(b, b + 512);
        };
        // A: ring of four pixel rows (row y is used by halo rows y .. y + 2 while row y + 1 ... y + 3 arrive), B: two sets of
        // three column shifts; everything for halo row h + 1 is requested before the MFMAs of halo row h are issued
        // this wave's rows: pixel rows r0 .. r0 + RP - 1, halo rows r0 .. r0 + RP + 1 (r0 * 18 is a multiple of 4: the
        // swizzle class of K = h * 18 + kx depends on the row INSIDE the part only)
        const int r0 = rpart * RP;
        ab += r0 * 2048;
#pragma unroll
        for (int j = 0; j < 4; ++j) bb[j] += r0 * 18 * 128;
        frag_t A[4], B[2][3];
        A[0] = load_a(0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) B[0][kx] = load_b(0, kx);
#pragma unroll
        for (int h = 0; h < RP + 2; ++h) {
            if (h + 1 < RP + 2) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) B[(h + 1) & 1][kx] = load_b(h + 1, kx);
                if (h + 1 < RP) A[(h + 1) & 3] = load_a(h + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int y = h - ky;
                if (y >= 0 && y < RP) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (MRISR_WR_DBG & 1) asm volatile("" ::"v"(A[y & 3]), "v"(B[h & 1][kx]));
                        else acc[ky * 3 + kx] = Frag16<T>::mma(A[y & 3], B[h & 1][kx], acc[ky * 3 + kx]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        int tile = by, cur = 0;
        __syncthreads();
        WPT_DECL
        while (tile < total_tiles) {
            mfma_tile(smem + cur * kWrBuf);
            WPT_MARK(6)
            __syncthreads();
            WPT_MARK(5)
            cur ^= 1;
            tile += gridDim.y;
        }
        WPT_DUMP()
    }

    // ---- two-stage reduction: the accumulators as they stand into this workgroup's slab (lane-contiguous 256-B stores)
    if (p.wsp) {
        float* wsb = p.wsp + ((size_t)(by * gridDim.x + bx) * 4 + wave) * kWrAcc + lane;
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) gstore(wsb + (a * 16 + r) * 64, acc[a][r]);
        return;
    }
    // ---- or float atomics straight into dW[co][tap][ci]: lane = ci, registers = co
    const int ci = ci0 + fi * 32 + lr;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + fo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomic_add_f32(p.dw + ((size_t)co * 9 + a) * p.Cin + ci, acc[a][r]);
        }
}

// Second stage: dw[co][tap][ci] += sum over the split-K slabs and the row parts of a fragment pair.  Thread = one element of dw
// (fragment pair, tap, register, lane); blockIdx.y = chunk of kWrRedChunk slabs.  With one chunk the thread is the only writer
// of its element: a plain read-modify-write instead of a float atomic (the version with 8 slabs per chunk and one thread per
// SLAB element spent 18 us per launch on 0.2 GB: 0.9 TB/s, three scattered atomics per element).
constexpr int kWrRedChunk = 24;
__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk,
                                                                int ksplit, int Cout, int Cin, int nco, int nci) {
    const int npairs = nco * nci, nparts = 4 / npairs;
    const int per_blk = 4 * kWrAcc, per_out = npairs * kWrAcc;
    const size_t total = (size_t)nblk * per_blk, total_out = (size_t)nblk * per_out;
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= total_out) return;
    const int blk = idx / per_out, rem = idx - (size_t)blk * per_out;
    const int pair = rem / kWrAcc, ar = (rem - pair * kWrAcc) >> 6, lane = rem & 63;
    const float* src = ws + (size_t)blk * per_blk + (size_t)pair * kWrAcc + (ar << 6) + lane;
    float s = 0.f;
    const int k0 = blockIdx.y * kWrRedChunk, k1 = min(k0 + kWrRedChunk, ksplit);
    for (int part = 0; part < nparts; ++part) {
        const float* sp = src + (size_t)part * npairs * kWrAcc;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) s += sp[(size_t)k * total];
    }
    const int a = ar >> 4, r = ar & 15, lr = lane & 31, lh = lane >> 5;
    const int ncib = Cin / (32 * nci), cib = blk % ncib, cob = blk / ncib;
    const int fo = pair / nci, fi = pair % nci;
    const int co = cob * 32 * nco + fo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, ci = cib * 32 * nci + fi * 32 + lr;
    float* d = dw + ((size_t)co * 9 + a) * Cin + ci;
    if (gridDim.y == 1) *d += s;
    else if (s != 0.f) atomic_add_f32(d, s);
}

#ifndef MRISR_KERNEL_ONLY
int num_cus();
#ifdef MRISR_WR_PT
extern "C" int mrisr_debug_phase_reset() {
    static unsigned long long zeros[96];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wr_cycles), zeros, sizeof(zeros));
}
extern "C" int mrisr_debug_phase_cycles(unsigned long long* out96) {
    return (int)hipMemcpyFromSymbol(out96, HIP_SYMBOL(g_wr_cycles), sizeof(unsigned long long) * 96);
}
#endif

// does mrisr_conv_wgrad take the row-streaming kernel for this descriptor?
bool conv_wgrad_rows_ok(const mrisr_conv_desc* d) {
#ifdef MRISR_NO_WGRAD_ROWS
    return false;
#endif
    if (d->dtype != MRISR_BF16 && d->dtype != MRISR_F16) return false;
    if (d->ksize != 3 || d->combine != MRISR_COMBINE_CONCAT) return false;
    if (d->Cout % 32 || d->Cin % 32 || d->H < 16 || d->W < 16) return false;
    for (int s = 0; s < d->nsrc; ++s)
        if (d->src[s].spatial != MRISR_SP_NONE) return false;
    const int bci = d->Cin % 64 ? 32 : 64;
    if (d->nsrc > 1 && d->src[0].C % bci) return false;      // an input-channel block lies in one concat source
    return true;
}

static void wgrad_rows_grid(const ConvParams& p, int& nblk, int& ksplit, int& total_tiles) {
    const int bco = p.Cout % 64 ? 32 : 64, bci = p.Cin % 64 ? 32 : 64;
    nblk = (p.Cout / bco) * (p.Cin / bci);
    total_tiles = p.N * ceil_div(p.H, 16) * ceil_div(p.W, 16);
    ksplit = ceil_div(p.cus > 0 ? p.cus : num_cus(), nblk);
    if (ksplit > total_tiles) ksplit = total_tiles;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 65535) ksplit = 65535;
}

size_t conv_wgrad_rows_workspace_floats(const ConvParams& p) {
    int nblk, ksplit, tt;
    wgrad_rows_grid(p, nblk, ksplit, tt);
    return (size_t)nblk * ksplit * 4 * kWrAcc;
}

template <typename T, int NCO, int NCI>
static int launch_wgrad_rows_t(ConvParams& p, size_t ws_floats, hipStream_t s) {
    p.th = 16; p.tw_log2 = 4;
    p.tiles_x = ceil_div(p.W, 16); p.tiles_y = ceil_div(p.H, 16);
    int nblk, ksplit, tt;
    wgrad_rows_grid(p, nblk, ksplit, tt);
    const size_t need = (size_t)nblk * ksplit * 4 * kWrAcc;
    if (!(p.wsp && ksplit >= 4 && need <= ws_floats)) p.wsp = nullptr;
    const size_t lds = 2 * (size_t)kWrBuf;
    bool raw = true;      // every source stored as-is (pooled / up-sampled / blended inputs): no GroupNorm arithmetic in the staging
    for (int i = 0; i < p.nsrc; ++i) raw = raw && p.src[i].mode == MRISR_SRC_RAW;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_kernel<T, NCO, NCI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_kernel<T, NCO, NCI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (raw) hipLaunchKernelGGL((conv_wgrad_rows_kernel<T, NCO, NCI, true>), dim3(nblk, ksplit), dim3(kWrThreads), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_rows_kernel<T, NCO, NCI, false>), dim3(nblk, ksplit), dim3(kWrThreads), lds, s, p);
    MRISR_CHECK_LAUNCH("conv_wgrad(rows)");
    if (p.wsp) {
        const size_t total = need / ksplit;
        wgrad_rows_reduce_kernel<<<dim3((unsigned)((total / (4 / (NCO * NCI)) + 255) / 256), ceil_div(ksplit, kWrRedChunk)), 256, 0, s>>>(p.wsp, p.dw, nblk, ksplit, p.Cout, p.Cin, NCO, NCI);
        MRISR_CHECK_LAUNCH("conv_wgrad(rows reduce)");
    }
    return MRISR_OK;
}

template <typename T>
static int launch_wgrad_rows_d(ConvParams& p, size_t ws_floats, hipStream_t s) {
    const bool co2 = p.Cout % 64 == 0, ci2 = p.Cin % 64 == 0;
    if (co2 && ci2) return launch_wgrad_rows_t<T, 2, 2>(p, ws_floats, s);
    if (co2) return launch_wgrad_rows_t<T, 2, 1>(p, ws_floats, s);
    if (ci2) return launch_wgrad_rows_t<T, 1, 2>(p, ws_floats, s);
    return launch_wgrad_rows_t<T, 1, 1>(p, ws_floats, s);
}

int launch_wgrad_rows(int dtype, ConvParams& p, size_t ws_floats, hipStream_t s) {
    if (dtype == MRISR_BF16) return launch_wgrad_rows_d<bf16_t>(p, ws_floats, s);
    return launch_wgrad_rows_d<f16_t>(p, ws_floats, s);
}
#endif
