// Raw-source 3x3 convolution on MFMA with a DEEP LDS-DMA ring (gfx950): forward of the layers whose input is a stored
// tensor (pooled / up-sampled / blended activations) and every input gradient of the wide layers.
//
// Replaces the same aten conv2d / convolution_backward(input) calls as conv_fwd.hip
// (/root/reference/models/unet_model.py:29,34,152,168 and autograd's dgrad of them).
//
// Why a second kernel (profiles/NOTES.md R2-13/14, R3-1): conv_igemm_kernel<..., DMA = 1> stages 78 KB per 18.9 MFLOP
// item pair through ~84 one-KiB LDS-DMA instructions, waits for ALL of them (`vmcnt(0)`) at every item boundary and
// spends 20-30 % of every wave's cycles issuing them.  This kernel changes the three things that set that cost:
//   * work item = 16 x 32 pixels x 128 output channels x 16 input channels (32-byte rows): a wave owns 64 pixels x 128
//     channels (128 accumulator registers, 0.75 instead of 1.0 KB of fragment reads per MFMA); per 18.9 MFLOP the
//     workgroup stages 20 + 36 = 56 one-KiB pieces instead of 84 (-33 %);
//   * the halo tiles run in a FOUR-deep ring and the weight images in a two-deep one, with counted `s_waitcnt vmcnt(N)`
//     and one raw `s_barrier` per item: a halo piece has three item times (~5 us) to arrive from HBM, not one;
//   * roles: waves 0-3 issue only the weight pieces (L2 hits, needed one item later), waves 4-7 only the halo pieces - the
//     counter of a wave then holds one kind of traffic in issue order and the counted wait is exact; every SIMD hosts
//     one wave of each role.
// Out-of-image halo pixels are fetched from a 1-KiB block of zeros in global memory (no exec-masked instruction, so the
// number of DMAs per item is a compile-time constant per wave).
#include <mutex>
#include <type_traits>

#include "conv_common.h"

template <typename T> struct RMma;
template <> struct RMma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct RMma<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

struct RingSrc {
    const void* ptr;
    int C, H, W, off_y, off_x;
    unsigned img_bytes;
};
struct RingParams {
    RingSrc src;           // ONE stored NHWC source (raw concat sources do not occur in the network)
    const void* wpk;       // ring-packed weights: [cout block][cin chunk of 16][tap][BN rows][32 B], slots pre-swizzled
    const float* bias;
    void* out;
    double* stats;
    const void* zeros;     // >= 1 KiB of zeros in global memory (source of out-of-image halo pixels)
    int N, H, W, Cin, Cout;
    int nchunks, ncb, tiles_x, tiles_y, ntiles, tiles_per_block;
    int groups, relu_out;
};

// Ablation switches of tuning builds (tools/build_ring_variant.sh; results invalid by construction, timing only):
// 1 no DMA issue, 2 no MFMA, 4 no epilogue.  The product library is compiled with 0: every test folds away.
#ifndef MRISR_RING_DBG
#define MRISR_RING_DBG 0
#endif
// Phase profile (tuning builds only, -DMRISR_RING_PT; tools/conv_bench.py prints it): s_memtime stamps around the parts
// of a step, accumulated per wave by the middle workgroup.  Not compiled into libmrisr.so.
#ifdef MRISR_RING_PT
__device__ unsigned long long g_ring_cycles[8][12];
#define RPT_DECL unsigned long long pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime(); const unsigned long long pt_r0 = __builtin_amdgcn_s_memrealtime();
#define RPT_MARK(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_now = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_now - pt_t; pt_t = pt_now; __builtin_amdgcn_sched_barrier(0); }
#else
#define RPT_DECL
#define RPT_MARK(k)
#endif
#ifndef MRISR_RING_AD
#define MRISR_RING_AD 3      // weight-fragment prefetch distance (fragments), see the fragment pipeline
#endif
constexpr int kRingThreads = 512;
constexpr int kRingHaloStages = 4;
constexpr int kRingWeightStages = 2;

__device__ __forceinline__ void ring_dma16(const void* g, unsigned lds_dst) {
    if (MRISR_RING_DBG & 1) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

// s_waitcnt vmcnt(N) with a compile-time N (the instruction takes an immediate)
template <int N> __device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N >= 0 && N <= 63, "6-bit vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// ... for N = a * A + b * B with wave-uniform run-time a in 0..2 and b in 0..3 (twelve immediates, one short branch chain)
template <int A, int B> __device__ __forceinline__ void wait_vmcnt_ab(int a, int b) {
    switch (a * 4 + b) {
        case 0: wait_vmcnt_c<0>(); break;
        case 1: wait_vmcnt_c<B>(); break;
        case 2: wait_vmcnt_c<2 * B>(); break;
        case 3: wait_vmcnt_c<3 * B>(); break;
        case 4: wait_vmcnt_c<A>(); break;
        case 5: wait_vmcnt_c<A + B>(); break;
        case 6: wait_vmcnt_c<A + 2 * B>(); break;
        case 7: wait_vmcnt_c<A + 3 * B>(); break;
        case 8: wait_vmcnt_c<2 * A>(); break;
        case 9: wait_vmcnt_c<2 * A + B>(); break;
        case 10: wait_vmcnt_c<2 * A + 2 * B>(); break;
        default: wait_vmcnt_c<2 * A + 3 * B>(); break;
    }
}

// MI = 32-pixel row fragments per wave (tile = 8*MI rows x 32 columns), NI = 32-channel fragments per wave (BN = 32*NI)
template <typename T, int MI, int NI, bool STATS>
__global__ __launch_bounds__(kRingThreads, 2) void conv_ring_kernel(const RingParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename RMma<T>::frag frag_t;
    constexpr int BN = 32 * NI;
    constexpr int TROWS = 8 * MI;
    constexpr int HWID = 34, HHGT = TROWS + 2, HROWS = HWID * HHGT;     // halo tile: rows of 32 B (16 channels)
    constexpr int NHP = (HROWS + 31) / 32;                               // 1-KiB pieces per halo image (32 rows each)
    constexpr int NWP = 9 * BN / 32;                                     // 1-KiB pieces per weight image
    constexpr int HB = NHP * 1024, WB = NWP * 1024;
    constexpr int NHJ = (NHP + 3) / 4, NWJ = (NWP + 3) / 4;              // pieces per issuing wave (4 waves per role)
    constexpr int NST = MI * NI * 2;                                     // 16-byte output stores per lane and tile
    static_assert(NHJ <= 9 && NWJ <= 9, "one DMA per tap step");
    static_assert(NHP % 4 == 0 && NWP % 4 == 0, "every wave of a role issues the same number of pieces per item");
    static_assert(2 * NHJ + 3 * NST <= 63, "counted waits fit the 6-bit vmcnt");

    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool halo_role = wave >= 4;              // waves 4-7 stage halo tiles, waves 0-3 weight images
    const int rw = wave & 3;                       // index inside the role
    char* lds_h = smem;
    char* lds_w = smem + kRingHaloStages * HB;
    float* lds_bias = reinterpret_cast<float*>(smem + kRingHaloStages * HB + kRingWeightStages * WB);
    const unsigned lds_h_s = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_h;
    const unsigned lds_w_s = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_w;

    // XCD-aware workgroup order (conv_fwd.hip): the cout blocks of a tile range, then the neighbouring ranges, share an L2
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cb = bid % p.ncb;
    const int bt0 = (bid / p.ncb) * p.tiles_per_block;
    const int bt1 = min(bt0 + p.tiles_per_block, p.ntiles);
    const int total = max(bt1 - bt0, 0) * p.nchunks;     // items of this workgroup
    const int bn0 = cb * BN;
    const char* wbase = (const char*)p.wpk + (size_t)cb * p.nchunks * WB;

    if (threadIdx.x < BN) lds_bias[threadIdx.x] = (p.bias && bn0 + (int)threadIdx.x < p.Cout) ? gload<float>(p.bias + bn0 + threadIdx.x) : 0.f;

    // ---- per-lane fragment offsets
    // weight image rows q = tap*BN + ni*32 + lr, 32 B each, 16-B slot s at position s ^ ((q >> 3) & 1) = s ^ ((lr >> 3) & 1)
    const int a_off = lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    // halo image rows r = (MI*wave + mi + ky)*34 + kx + lr, slot s at position s ^ ((r >> 3) & 1): one offset per (mi+ky, kx)
    int xb[(MI + 2) * 3];
#pragma unroll
    for (int yy = 0; yy < MI + 2; ++yy)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = (MI * wave + yy) * HWID + kx + lr;
            xb[yy * 3 + kx] = r * 32 + ((lh ^ ((r >> 3) & 1)) << 4);
        }

    // ---- staging duty of this lane: piece rw + 4 j of its role's image, lane l = byte 16 l of the piece
    // halo: row = 32 piece + (l >> 1), physical slot l & 1 -> it fetches the LOGICAL slot (l & 1) ^ ((row >> 3) & 1)
    int hyx[NHJ];
#pragma unroll
    for (int j = 0; j < NHJ; ++j) {
        const int row = 32 * (rw + 4 * j) + (lane >> 1);
        const int hy = row / HWID;
        hyx[j] = row < HROWS ? ((hy << 16) | (row - hy * HWID)) : -1;
    }
    const int lslot = (lane & 1) ^ ((lane >> 4) & 1);       // (row >> 3) & 1 = (lane >> 4) & 1: pieces start at multiples of 32 rows

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    constexpr bool has_stats = STATS;      // input-gradient launches carry no GroupNorm statistics: no registers for them
    const bool has_br = p.bias != nullptr || p.relu_out != 0;
    const int gs = p.groups > 0 ? p.Cout / p.groups : 4;
    // GroupNorm partial sums per lane: a group spans >= 16 channels here (host-checked), i.e. the quad pairs {0,1} and {2,3}
    // of a fragment each lie in one group
    constexpr int NSQ = STATS ? 2 : 1;
    float st_s[NI][NSQ], st_ss[NI][NSQ];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < NSQ; ++q) { st_s[ni][q] = 0.f; st_ss[ni][q] = 0.f; }

    struct Cursor { int tile, kc, n, ty0, tx0; };
    auto decode = [&](Cursor& c) {
        const int tx = c.tile % p.tiles_x;
        const int r = c.tile / p.tiles_x;
        c.n = r / p.tiles_y;
        c.ty0 = (r - c.n * p.tiles_y) * TROWS;
        c.tx0 = tx * 32;
    };
    auto advance = [&](Cursor& c) {
        if (++c.kc == p.nchunks) {
            c.kc = 0;
            ++c.tile;
            if (c.tile < bt1) decode(c);
        }
    };

    // one halo piece of item `it` into ring slot `slot`
    auto dma_halo = [&](const Cursor& it, int slot, int j) {
        const int kcs = __builtin_amdgcn_readfirstlane(it.kc), ns = __builtin_amdgcn_readfirstlane(it.n);
        const int ty0s = __builtin_amdgcn_readfirstlane(it.ty0), tx0s = __builtin_amdgcn_readfirstlane(it.tx0);
        const char* base = image_base(p.src.ptr, ns, p.src.img_bytes);
        const unsigned Hs = p.src.H, Ws = p.src.W, C2 = (unsigned)p.src.C * 2u;
        const int ys0 = ty0s - 1 - p.src.off_y, xs0 = tx0s - 1 - p.src.off_x;
        const unsigned y = ys0 + (hyx[j] >> 16), x = xs0 + (hyx[j] & 0xffff);      // hyx = -1: x out of range
        const bool ok = (y < Hs) & (x < Ws);
        const unsigned off = mad_u24(mad_u24(y, Ws, x), C2, (unsigned)(kcs * 32 + lslot * 16));
        const char* src = ok ? base + off : (const char*)p.zeros + lane * 16;
        ring_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_h_s + slot * HB + (rw + 4 * j) * 1024)));
    };
    auto dma_weight = [&](int kc, int slot, int j) {
        const int piece = rw + 4 * j;
        const char* src = wbase + (size_t)kc * WB + piece * 1024 + lane * 16;
        ring_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_w_s + slot * WB + piece * 1024)));
    };

    // ---- epilogue of a finished tile (as conv_fwd.hip's plain NHWC epilogue): bias / ReLU, GroupNorm partial sums,
    // packed 16-bit values exchanged between the lane halves (permlane32) -> 16-byte stores
    auto epilogue = [&](const Cursor& it) {
        const size_t e0 = ((size_t)(it.n * p.H + it.ty0) * p.W + it.tx0) * p.Cout + bn0;
        char* obase = (char*)p.out + e0 * sizeof(T);
        int lh_e = lh;
        asm volatile("" : "+v"(lh_e));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // (the host takes this kernel only for planes made of whole tiles and Cout a multiple of BN: every pixel and
            // channel is valid, every store below executes - the counted waits rely on NST stores per tile and wave)
            const int py = MI * wave + mi, px = lr;
            const unsigned loff = mad_u24(mad_u24(py, p.W, px), p.Cout * 2u, 16u * lh_e);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if (has_br) {
                    const float floor_v = p.relu_out ? 0.f : -INFINITY;
                    const float* bl = lds_bias + 4 * lh_e;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ni * 32 + 8 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ni][mi][4 * q + j] = fmaxf(acc[ni][mi][4 * q + j] + b[j], floor_v);
                    }
                }
                u32x2 packed[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[ni][mi][4 * q + j];
                        acc[ni][mi][4 * q + j] = 0.f;
                    }
                    if (has_stats) {
                        const float qs = (v[0] + v[1]) + (v[2] + v[3]);
                        const float qq = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                        st_s[ni][q >> 1] += qs;
                        st_ss[ni][q >> 1] += qq;
                    }
                    typedef __attribute__((ext_vector_type(4))) T t4_t;
                    union { t4_t b; u32x2 u; } cv;
                    cv.b = t4_t{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                    packed[q] = cv.u;
                }
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    const u32x2 a = packed[q], b = packed[q + 1];
                    auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    gstore(obase + loff + (ni * 32 + 8 * q) * 2, o);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < NSQ; ++q) {
                int co = bn0 + ni * 32 + 16 * q + 4 * lh;
                asm volatile("" : "+v"(co));
                const float s = half_wave_sum(st_s[ni][q]), ss = half_wave_sum(st_ss[ni][q]);
                if (lr == 0) {
                    const int g = co / gs;
                    double* sp = p.stats + stat_slot_off_id(bid, p.N, p.groups) + ((size_t)n * p.groups + g) * 2;
                    atomic_add_f64(sp, (double)s);
                    atomic_add_f64(sp + 1, (double)ss);
                }
                st_s[ni][q] = 0.f;
                st_ss[ni][q] = 0.f;
            }
    };

    if (total <= 0) return;
    __syncthreads();                   // the bias table is visible (no DMA is in flight yet: this drains nothing)
    // ---- prologue: halo items 0 .. 2 and weight image 0 in flight
    Cursor cur{bt0, 0, 0, 0, 0};
    decode(cur);
    Cursor hcur = cur;                 // halo role: the item whose tile is issued next (runs 3 ahead)
    int h_issued = 0;                  // halo items issued so far
    int w_kc = 0;                      // weight role: chunk of the next image to issue
    if (halo_role) {
#pragma unroll 1
        for (int k = 0; k < kRingHaloStages - 1 && k < total; ++k) {
#pragma unroll
            for (int j = 0; j < NHJ; ++j)
                dma_halo(hcur, k, j);
            advance(hcur);
            ++h_issued;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NWJ; ++j)
            dma_weight(0, 0, j);
        w_kc = p.nchunks > 1 ? 1 : 0;
    }
#ifdef MRISR_RING_PRIO
    // static priority for the younger half (waves 4-7 lose the arbitration to the older wave of their SIMD)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    int ep1 = 0, ep2 = 0, ep3 = 0;     // 1 if this wave ran an epilogue (NST output stores) in the previous step / the two before it

    RPT_DECL
#pragma unroll 1
    for (int c = 0; c < total; ++c) {
        RPT_MARK(8)
        // ---- counted wait: everything this wave issued for item c has landed; YOUNGER operations stay in flight
        // (vmcnt retires in issue order and counts stores too).  Halo role: tile c went out in step c-3, so the younger
        // operations are the epilogue stores of steps c-3 .. c-1 and the tiles of items c+1 .. issued since.  Weight role:
        // image c went out in step c-1, before that step's epilogue stores.  Never over-count: a too large N is a race.
        if (halo_role) wait_vmcnt_ab<NHJ, NST>(h_issued - (c + 1), ep1 + ep2 + ep3);
        else wait_vmcnt_ab<NHJ, NST>(0, ep1);
        RPT_MARK(0)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RPT_MARK(5)
        // after the barrier every wave has finished item c-1: halo slot (c+3) % 4 = (c-1) % 4 and weight slot (c+1) & 1
        // are free
        const bool issue_h = halo_role && h_issued < total;
        const bool issue_w = !halo_role && c + 1 < total;
        const int hslot = h_issued & (kRingHaloStages - 1);
        const int wslot_n = (c + 1) & 1;
        const char* hbuf = lds_h + (c & (kRingHaloStages - 1)) * HB;
        const char* wl = lds_w + (c & 1) * WB + a_off;

        // Fragment pipeline: the pixel fragments of tap t+1 and the weight fragments up to AD steps ahead are requested
        // before the MFMAs that use the current ones are issued.  A prefetch distance of one fragment (two MFMAs = 64
        // pipe cycles) is shorter than an LDS round trip with eight waves reading (~150-200 cycles): measured without
        // any DMA, 1.19 PFLOP/s at distance 1.  B: two sets, A: a ring of AD + 1 fragments - 16 + 4 (AD + 1) operand
        // registers beside the 128 accumulators.  One DMA per tap step is dealt out behind the reads.
        constexpr int AD = MRISR_RING_AD;
        frag_t af[AD + 1], bf[2][MI];
        auto load_b = [&](int tap, int set) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) bf[set][mi] = *reinterpret_cast<const frag_t*>(hbuf + xb[(mi + ky) * 3 + kx]);
        };
        auto load_a = [&](int idx) {      // idx = tap * NI + ni
            af[idx % (AD + 1)] = *reinterpret_cast<const frag_t*>(wl + ((idx / NI) * BN + (idx % NI) * 32) * 32);
        };
        load_b(0, 0);
#pragma unroll
        for (int i = 0; i < AD; ++i) load_a(i);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int idx = tap * NI + ni;
                if (idx + AD < 9 * NI) load_a(idx + AD);
                if (ni == 0 && tap + 1 < 9) load_b(tap + 1, (tap + 1) & 1);
                if (ni == 1) {
                    if (issue_h) { if (tap < NHJ) dma_halo(hcur, hslot, tap); }
                    else if (issue_w) { if (tap < NWJ) dma_weight(w_kc, wslot_n, tap); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (MRISR_RING_DBG & 2) asm volatile("" ::"v"(af[idx % (AD + 1)]), "v"(bf[tap & 1][mi]));
                    else acc[ni][mi] = RMma<T>::run(af[idx % (AD + 1)], bf[tap & 1][mi], acc[ni][mi]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        RPT_MARK(6)
        if (issue_h) { advance(hcur); ++h_issued; }
        if (issue_w) { if (++w_kc == p.nchunks) w_kc = 0; }
        ep3 = ep2;
        ep2 = ep1;
        ep1 = 0;
        if (cur.kc == p.nchunks - 1) {
            if (!(MRISR_RING_DBG & 4)) {
                epilogue(cur);
                ep1 = 1;
            }
            Cursor nx = cur;
            advance(nx);
            if (has_stats && (c + 1 >= total || nx.n != cur.n)) flush_stats(cur.n);
            cur = nx;
            RPT_MARK(4)
        } else {
            ++cur.kc;
        }
    }
#ifdef MRISR_RING_PT
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
        pt_acc[10] = __builtin_amdgcn_s_memrealtime() - pt_r0;
#pragma unroll
        for (int k = 0; k < 12; ++k) atomicAdd(&g_ring_cycles[wave][k], pt_acc[k]);
        if (threadIdx.x == 0) atomicAdd(&g_ring_cycles[0][11], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host side
#ifndef MRISR_KERNEL_ONLY
#ifdef MRISR_RING_PT
extern "C" int mrisr_debug_phase_reset() {
    static unsigned long long zeros[96];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_cycles), zeros, sizeof(zeros));
}
extern "C" int mrisr_debug_phase_cycles(unsigned long long* out96) {
    return (int)hipMemcpyFromSymbol(out96, HIP_SYMBOL(g_ring_cycles), sizeof(unsigned long long) * 96);
}
#endif
int num_cus();

// ring layout of a packed weight: rows of 16 input channels (32 B), BN output channels per block
extern "C" int mrisr_conv_ring_bn(int dtype, int Cout, int Cin, int ksize) { return conv_ring_bn(dtype, Cout, Cin, ksize); }
extern "C" size_t mrisr_packed_weight_bytes_ring(int dtype, int Cout, int Cin, int ksize) {
    const int bn = mrisr_conv_ring_bn(dtype, Cout, Cin, ksize);
    if (!bn) return 0;
    return (size_t)(Cout / bn) * (Cin / 16) * 9 * bn * 32;
}

static const void* ring_zeros() {
    static const void* z = [] {
        void* d = nullptr;
        if (hipMalloc(&d, 1024) != hipSuccess) return (const void*)nullptr;
        (void)hipMemset(d, 0, 1024);
        (void)hipDeviceSynchronize();
        return (const void*)d;
    }();
    return z;
}

// Does this launch take the ring kernel?  (p: filled by conv_fill_params.)  Static conditions + enough work items to fill
// the persistent grid: a 16 x 32 tile per workgroup and one cout block of 128 means small planes leave CUs idle.
bool conv_ring_eligible(const mrisr_conv_desc* d, const ConvParams& p) {
#ifdef MRISR_NO_RING
    return false;
#endif
    if (!d->wpacked_ring) return false;
    if (mrisr_conv_ring_bn(d->dtype, d->Cout, d->Cin, d->ksize) != 128) return false;
    if (d->Cin < 256) return false;      // the ring pays from 256 input channels on (profiles/r03_ring_kernel.txt)
    if (d->out_mode != MRISR_OUT_PLAIN || d->relu_mask || d->combine != MRISR_COMBINE_CONCAT) return false;
    if (d->nsrc != 1 || d->src[0].mode != MRISR_SRC_RAW || d->src[0].spatial != MRISR_SP_NONE) return false;
    if (d->stats && ((d->Cout / d->groups) & 15)) return false;     // a GroupNorm group spans whole 16-channel quad pairs
    if (d->H % 16 || d->W % 32) return false;               // whole 16 x 32 tiles only (unpredicated stores, counted waits)
    // (a tile's epilogue - 16 scattered 1-KiB stores per wave, ~4-8 k cycles, both waves of a SIMD at once - is amortised over
    // Cin / 16 items: measured per layer, the ring wins from 256 input channels on)
    const int tiles = d->N * ceil_div(d->H, 16) * ceil_div(d->W, 32);
    const int ncb = d->Cout / 128;
    return (long)tiles * ncb * 4 >= (long)p.cus * 3;       // >= 0.75 work items per CU
}

template <typename T>
static int launch_ring_t(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s) {
    constexpr int MI = 2, NI = 4, BN = 128;
    RingParams p;
    memset(&p, 0, sizeof(p));
    p.src = RingSrc{d->src[0].ptr, d->src[0].C, d->src[0].H, d->src[0].W, d->src[0].off_y, d->src[0].off_x, cp.src[0].img_bytes};
    p.wpk = d->wpacked_ring; p.bias = d->bias; p.out = d->out; p.stats = d->stats; p.zeros = ring_zeros();
    if (!p.zeros) MRISR_FAIL(MRISR_E_HIP, "conv_forward(ring): could not allocate the zero block");
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.nchunks = d->Cin / 16; p.ncb = d->Cout / BN;
    p.tiles_x = ceil_div(d->W, 32); p.tiles_y = ceil_div(d->H, 8 * MI);
    p.ntiles = d->N * p.tiles_x * p.tiles_y;
    p.groups = d->stats ? d->groups : 0; p.relu_out = d->relu_out;
    int per_cb = cp.cus / p.ncb;
    if (per_cb < 1) per_cb = 1;
    if (per_cb > p.ntiles) per_cb = p.ntiles;
    p.tiles_per_block = ceil_div(p.ntiles, per_cb);
    per_cb = ceil_div(p.ntiles, p.tiles_per_block);
    const int grid = per_cb * p.ncb;
    constexpr int HB = ((34 * (8 * MI + 2) + 31) / 32) * 1024, WB = 9 * BN * 32;
    const size_t lds = kRingHaloStages * HB + kRingWeightStages * WB + BN * sizeof(float);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_kernel<T, MI, NI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_kernel<T, MI, NI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (p.stats) hipLaunchKernelGGL((conv_ring_kernel<T, MI, NI, true>), dim3(grid), dim3(kRingThreads), lds, s, p);
    else hipLaunchKernelGGL((conv_ring_kernel<T, MI, NI, false>), dim3(grid), dim3(kRingThreads), lds, s, p);
    MRISR_CHECK_LAUNCH("conv_forward(ring)");
    return MRISR_OK;
}

int launch_conv_ring(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s) {
    if (d->dtype == MRISR_BF16) return launch_ring_t<bf16_t>(d, cp, s);
    return launch_ring_t<f16_t>(d, cp, s);
}
#endif
