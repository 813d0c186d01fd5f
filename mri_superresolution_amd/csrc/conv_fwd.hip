// Implicit-GEMM 3x3 / 1x1 convolution on MFMA for gfx950 (forward and input-gradient).
//
// Replaces the aten conv2d calls behind nn.Conv2d in /root/reference/models/unet_model.py
// (:29,34,72,101,152,168), with the surrounding GroupNorm-apply + LeakyReLU (:30-31), MaxPool2d
// (:52), bilinear Upsample (:71,151), torch.cat (:93), PixelShuffle (:102) and the alpha blend
// (:206-207) folded into the operand loader / epilogue so none of those tensors is materialised.
//
// Decomposition: a work item = 256 output pixels (TH x TW tile of one image) x BN output channels x one cin
// chunk of 64 bytes; per item the transformed (TH+2)x(TW+2) halo tile (and, unless all chunks' weights are
// LDS-resident, the 9 x BN x 64 B weight image) is staged in LDS once and re-used by all 9 taps (LDS-tiled
// direct conv on MFMA).  D = W(BN x K) * X(K x pixels): the accumulator has a pixel per lane and 4 consecutive
// output channels per register quad, so NHWC stores are 8/16-byte pieces.  A persistent 8-wave workgroup runs
// two such item streams in antiphase (see conv_igemm_kernel).  The materialised exceptions to "nothing is
// materialised" (pooled / upsampled / blended inputs of the narrow layers) are listed in DESIGN.md section 3.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "conv_common.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
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

// Register-resident prefetch of the next work item's operands (global loads stay in flight while the
// current item's MFMAs run).
// NH = raw vectors kept per halo slot: 1 plain / 2 blend (both sources) / 4 gathers (2x2 pool window or the
// 4 bilinear taps) / 0 = no halo prefetch (staged synchronously in the vector phase).
template <typename T, int NW, int NH>
struct Prefetch {
    Vec16<T> w[NW];                               // slice of the weight image (streamed mode)
    Vec16<T> h[kMaxHaloIter][NH > 0 ? NH : 1];    // raw halo vectors
    float sc[Vec16<T>::N], sh[Vec16<T>::N];       // GroupNorm affine of the chunk's channels (source 0 / the only one)
    float sc1[NH == 2 ? Vec16<T>::N : 1], sh1[NH == 2 ? Vec16<T>::N : 1];   // blend: source 1
    int mask, mode;
    bool ok[kMaxHaloIter];                        // NH == 1: slot holds an in-image pixel of an existing channel (else: zero)
    float aff;                                    // NH == 1, first wave of the half: one entry of the chunk's affine table
    float slope;                                  // NH == 1: activation as max(y, slope*y): 0.2 NORM, 1 RAW, 0 RELU
};

struct NoPace {};                   // run_mma without DMA pacing (antiphase schedule)
constexpr int kFwdThreads = 512;    // two 4-wave halves working in antiphase
constexpr int kLoaderBlend = 3;     // template-only loader kind: sigmoid(alpha)-blend of two sources
constexpr int kEpiMask = 2;         // template-only epilogue kind: plain store gated by relu_mask > 0 (VGG dgrad)

// Work decomposition: a persistent 8-wave workgroup owns one cout block (BN channels) and a contiguous range
// of 256-pixel tiles.  Its two halves (waves 0-3 / 4-7) each walk their own tiles; a work item is (tile, cin
// chunk of 64 B).  In every tick one half runs the MFMAs of its current item while the other half does the vector
// work: GroupNorm+LeakyReLU transform and LDS commit of its next item, the global loads of the item after that,
// then the epilogue of a finished tile.  The workgroup barrier at the end of a tick swaps the roles, so each SIMD
// always has one matrix wave and one vector wave.
// Phase profile (tuning builds only, -DMRISR_PHASE_TIMING, tools/build_prof.sh + tools/conv_bench.py): s_memtime stamps
// around the parts of a tick, accumulated in SGPRs by every wave of the middle workgroup.  Not compiled into
// libmrisr.so.
#ifdef MRISR_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8][12];
#define PT_DECL unsigned long long pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime(); const unsigned long long pt_r0 = __builtin_amdgcn_s_memrealtime();
#define PT_MARK(k) { const unsigned long long pt_now = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_now - pt_t; pt_t = pt_now; }
#define PT_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0f70)   /* vmcnt(0) only (gfx9 encoding: lgkmcnt 15, expcnt 7) */
#else
#define PT_DECL
#define PT_MARK(k)
#define PT_WAIT_LOADS()
#endif

// One 16-byte-per-lane LDS-DMA (global_load_lds_dwordx4): the active lanes' 16 bytes go to LDS bytes
// [lds_dst + 16 * lane, + 16) with no register staging; counted on vmcnt like a load.  Inline asm: hipcc's own waitcnt
// bookkeeping does not see it (in-order retirement makes that safe: an unknown operation can only make the compiler's
// counted waits longer), completion waits are explicit.  M0 carries the LDS base and is restored.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

// DMA = true (plain loader, every source MRISR_SRC_RAW): the halo tile goes global -> LDS by LDS-DMA, double-buffered
// per half; the vector phase of a tick then holds no loads, no transform and no LDS commit - only the DMA issue of the
// next item, the zero fill of conv-padding slots and the epilogue.
template <typename T, int BN, int SPATIAL, int KS, bool WS, int EPI, bool DMA = false>
__global__ __launch_bounds__(kFwdThreads, 2) void conv_igemm_kernel(const ConvParams p_in) {
    static_assert(!DMA || SPATIAL == MRISR_SP_NONE, "the LDS-DMA halo path is the plain loader's");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams p = pin_params(p_in);
    constexpr int NTAPS = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int NF = BN / 32;               // cout fragments per wave
    constexpr int VEC = Vec16<T>::N;
    constexpr int WIMG_VECS = NTAPS * BN * 4; // 16-B vectors in one (cout block, cin chunk) weight image
    constexpr int NW = WS ? 1 : (WIMG_VECS + kConvThreads - 1) / kConvThreads;
    typedef typename Mma<T>::frag frag_t;

    const int half = threadIdx.x >> 8;        // wave-uniform
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int TW = 1 << p.tw_log2, TH = p.th;
    const int hw = TW + 2 * PAD, hh = TH + 2 * PAD;
    const int npix_halo = hw * hh;
    // register-staged: one tile of full slots per half (commits are unpredicated); DMA: two 340-row buffers per half
    constexpr int halo_bytes = DMA ? 2 * kDmaHaloBytes : kMaxHaloIter * 64 * kHaloRowBytes;
    char* lds_halo = smem + half * halo_bytes;
    // streamed weights: TWO images shared by both halves - item c (the halves run the same (tile, chunk) sequence one
    // tick apart) uses image c & 1; each half loads and writes the image of every other item (see the schedule below)
    char* lds_w = smem + 2 * halo_bytes;
    float* lds_bias = reinterpret_cast<float*>(smem + 2 * halo_bytes + (WS ? p.nchunks : 2) * (WIMG_VECS * 16));   // [BN]
    // plain loader: GroupNorm scale (entries 0..31) / shift (32..63) of the cin chunk this half commits next
    float* lds_aff = lds_bias + BN + half * 64;
    // (experiment, -DMRISR_STAGED_STORES, measured -0.7 % on the step and left off: coalescing the output stores does
    // not make them cheaper, the extra LDS round trip costs more than it saves)
    // bf16 plain epilogue: per-wave staging rows ([32 pixels][BN couts], 16 B of padding per row) through which a
    // fragment's outputs are re-read pixel-contiguously, so that one store instruction writes whole 128-byte lines
    // (8 lanes per pixel) instead of 64 scattered 16-byte pieces; shared by the halves (their vector phases alternate)
    constexpr int kStageRow = BN * 2 + 16;
    char* lds_stage = reinterpret_cast<char*>(lds_bias + BN + 128) + (threadIdx.x >> 6 & 3) * (32 * kStageRow);

    // this workgroup: one cout block, tiles [bt0, bt1); this half: [tile0, tile1)
    // XCD-aware order: workgroup b runs on XCD b % 8, so the logical index (b % 8) * (grid / 8) + b / 8 puts CONSECUTIVE
    // logical workgroups - the ncb cout blocks of the same tiles, then the neighbouring tiles - on one XCD, i.e. behind
    // one L2: an input tile is fetched into that L2 once instead of once per cout block, and neighbouring tiles share
    // their halo rows there.  The GroupNorm statistics slot must then come from the LOGICAL index too: the workgroups of
    // one XCD work on the same image, and with the slot taken from blockIdx (= XCD + 8 k) they shared 2 of the 16 slots
    // - same-address fp64 atomics that cost the forward convs 10-30 % until the slot followed the logical order.
    // Measured per kernel inside the training step (A/B on one box): 64-channel layers 138 -> 121 us, the pixel-shuffle
    // conv (2 cout blocks) 314 -> 218 us, streamed-weights layers 103 -> 101.5 us; step +2.3 %.
    int bid = blockIdx.x;
#ifndef MRISR_NO_XCD_REMAP
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
#endif
    const int cb = bid % p.ncb;
    const int bt0 = (bid / p.ncb) * p.tiles_per_block;
    const int bt1 = min(bt0 + p.tiles_per_block, p.ntiles);
    const int nbt = bt1 - bt0, nh0 = (nbt + 1) >> 1;
    const int tile0 = half ? bt0 + nh0 : bt0;
    const int tile1 = half ? bt1 : bt0 + nh0;
    const int nitems = (tile1 - tile0) * p.nchunks;              // this half
    const int nticks = 2 * nh0 * p.nchunks + 2;                  // workgroup-uniform
    const int bn0 = cb * BN;
    const char* wbase = (const char*)p.wpacked + (size_t)cb * p.nchunks * (WIMG_VECS * 16);
    // the halo is prefetched through registers only for the plain loader; gathers (pool / bilinear / blend)
    // are staged synchronously in the commit phase
    // (SPATIAL == kLoaderBlend: the two-source alpha blend, geometry of SP_NONE)
    constexpr int GSP = (SPATIAL == kLoaderBlend) ? MRISR_SP_NONE : SPATIAL;     // geometry / gather kind
    // halo prefetch through registers: plain 1 vector per slot, blend 2, gathers 4 (only where the accumulators
    // leave room: BN = 32); otherwise the halo is staged synchronously in the vector phase
    constexpr int NH = SPATIAL == MRISR_SP_NONE ? 1 : SPATIAL == kLoaderBlend ? 2 : (BN == 32 ? 4 : 0);
    constexpr bool pf_halo = NH > 0;
    // halo slots the plain loader touches: a 1x1 conv has no halo ring, its 256 pixels fill 4 slots exactly
    constexpr int NSLOT = KS == 1 ? 4 : kMaxHaloIter;

    float blend_a = 0.f;
    if (p.combine == MRISR_COMBINE_BLEND) blend_a = 1.f / (1.f + __expf(-gload<float>(p.blend_alpha)));

    if constexpr (WS) {   // weights-stationary: every cin chunk's image is loaded once, by all 512 threads
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wbase);
        for (int v = threadIdx.x; v < p.nchunks * WIMG_VECS; v += kFwdThreads) reinterpret_cast<u32x4*>(lds_w)[v] = gload<u32x4>(wsrc + v);
    }
    const bool has_br = p.bias != nullptr || p.relu_out != 0;      // block-uniform: epilogue with bias / ReLU
    const bool has_stats = p.stats != nullptr;
    const bool all_raw = p.src[0].mode == MRISR_SRC_RAW && (p.nsrc < 2 || p.src[1].mode == MRISR_SRC_RAW);   // block-uniform
    if (threadIdx.x < BN) lds_bias[threadIdx.x] = (p.bias && bn0 + (int)threadIdx.x < p.Cout) ? gload<float>(p.bias + bn0 + threadIdx.x) : 0.f;

    // per-lane LDS byte offsets of its two pixels' halo rows (tap (0,0), k-step 0) and of its weight row
    int xb[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pl = wave * 64 + mi * 32 + lr;
        xb[mi] = halo_off((pl >> p.tw_log2) * hw + (pl & (TW - 1)), lh);
    }
    int wb = lds_off(lr, lh);     // k-step 1 = this XOR 32; taps / fragments are constant offsets
    // DMA halo image: row r = halo pixel, 64 B, chunk c at position c ^ ((r >> 2) & 3) - the tap shift changes the
    // swizzle per lane, so the 2 x NTAPS fragment offsets are tile-independent per-lane constants (k-step 1 = XOR 32)
    int xa[2][DMA ? NTAPS : 1];
    if constexpr (DMA) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int pl = wave * 64 + mi * 32 + lr;
            const int r0 = (pl >> p.tw_log2) * hw + (pl & (TW - 1));
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                const int r = r0 + (tap / KS) * hw + (tap % KS);
                xa[mi][tap] = r * 64 + ((lh ^ ((r >> 2) & 3)) << 4);
            }
        }
    }

    // tile-independent halo slot coordinates of this thread: slot i = halo pixel (t>>2) + 64 i
    int hyx[kMaxHaloIter];
#pragma unroll
    for (int i = 0; i < kMaxHaloIter; ++i) {
        const int hp = (t >> 2) + 64 * i;
        const int hy = hp / hw;
        hyx[i] = hp < npix_halo ? ((hy << 16) | (hp - hy * hw)) : -1;
    }

    f32x16 acc[NF][2];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    // GroupNorm partial statistics, kept per lane across the tiles of one image (host guarantees that a
    // group spans a multiple of 4 channels whenever p.stats is set)
    const int gs = p.groups > 0 ? p.Cout / p.groups : 4;
    float st_s[NF][4], st_ss[NF][4];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) { st_s[ni][q] = 0.f; st_ss[ni][q] = 0.f; }

    HaloGeom<GSP> geom;
    Prefetch<T, NW, NH> pf;
    pf.mask = 0;
    pf.mode = 0;

    auto decode = [&](int tile, int& n, int& ty0, int& tx0) {
        const int tx = tile % p.tiles_x;
        const int r = tile / p.tiles_x;
        n = r / p.tiles_y;
        ty0 = (r - n * p.tiles_y) * TH;
        tx0 = tx * TW;
    };
    int upflags = 0;      // UP2 gather prefetch: bit 2i = second row differs, bit 2i+1 = second column differs
    auto set_geom = [&](int n, int ty0, int tx0) {
        if (DBG(p) & 32) return;
        if constexpr (NH == 1) return;     // plain loader: addresses are derived per item from scalars (see issue)
#pragma unroll
        for (int i = 0; i < kMaxHaloIter; ++i)
            halo_geom_yx<GSP>(geom, i, hyx[i] >> 16, hyx[i] & 0xffff, hyx[i] >= 0, PAD, n, ty0, tx0, p);
        if constexpr (NH == 4 && GSP == MRISR_SP_UP2) {
            int f = 0;
#pragma unroll
            for (int i = 0; i < kMaxHaloIter; ++i) f |= ((geom.dyo[i] != 0) << (2 * i)) | ((geom.dxo[i] != 0) << (2 * i + 1));
            upflags = f;
        }
    };
    // issue the global loads of a work item (geometry in `geom`, image n, cin chunk kc)
    // streamed weights (not weights-stationary): the next chunk's image, issued at the start of the matrix phase
    // (L2-resident, and keeping these 36 VGPRs dead during the epilogue avoids spills)
    auto issue_weights = [&](int kc) {
        if (DBG(p) & (4 | 256)) return;     // (ablation: 256 = no weight loads)
        if constexpr (!WS) {
            const u32x4* wsrc = reinterpret_cast<const u32x4*>(wbase + (size_t)kc * (WIMG_VECS * 16));
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int v = t + j * kConvThreads;
                if (v < WIMG_VECS) pf.w[j].v = gload<decltype(pf.w[j].v)>(wsrc + v);
            }
        }
    };
    auto store_weights = [&](int item) {     // prefetched image -> LDS image (item & 1)
        if (DBG(p) & 128) return;             // (ablation: 128 = no weight LDS stores)
        if constexpr (!WS) {
            char* dst = lds_w + (size_t)(item & 1) * (WIMG_VECS * 16);
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int v = t + j * kConvThreads;
                if (v < WIMG_VECS) *reinterpret_cast<decltype(pf.w[j].v)*>(dst + (size_t)v * 16) = pf.w[j].v;
            }
        }
    };
#ifndef MRISR_NO_DMA_WEIGHTS
#define MRISR_DMA_WEIGHTS 1
#endif
#ifdef MRISR_DMA_WEIGHTS
    // The streamed weight image by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, global ->
    // LDS with no register staging and no ds_write).  The packed image is already the LDS image byte for byte, so piece
    // q is a linear 1 KiB copy.  Inline asm: hipcc's own waitcnt bookkeeping does not see it (in-order retirement makes
    // that safe: an unknown older operation can only make its counted waits longer), the completion wait is explicit.
    // Half 0 issues the image of item c + 1 at the start of its matrix phase of item c - image (c + 1) & 1 was last read
    // two ticks earlier - and waits for it at the start of its next vector phase; both halves read it after the barrier.
    // Measured against the register-staged variant (each half loading / storing every other image): same results, 204
    // instead of 244 VGPRs, streamed-weights kernel 108.5 -> 107.7 us inside the step (-DMRISR_NO_DMA_WEIGHTS builds that
    // variant).
    // (NWV = 4: issued by the four waves of half 0 - the antiphase schedule; NWV = 8: by all eight waves - symmetric one)
    auto dma_weights = [&](int kc, int item, auto nwv_tag) {
        constexpr int NWV = decltype(nwv_tag)::value;
        if (DBG(p) & (4 | 256)) return;
        if constexpr (!WS) {
            constexpr int NPIECE = WIMG_VECS * 16 / 1024;
            const int kcs = __builtin_amdgcn_readfirstlane(kc), its = __builtin_amdgcn_readfirstlane(item);
            const int w0 = __builtin_amdgcn_readfirstlane(NWV == 8 ? (int)(threadIdx.x >> 6) : wave);
            const char* src = wbase + (size_t)kcs * (WIMG_VECS * 16) + lane * 16;
            const unsigned dst0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_w + (unsigned)(its & 1) * (WIMG_VECS * 16);
#pragma unroll
            for (int j = 0; j < (NPIECE + NWV - 1) / NWV; ++j) {
                const int piece = w0 + NWV * j;
                if (piece < NPIECE) {
                    unsigned keep;
                    const char* g = src + piece * 1024;
                    const unsigned d = dst0 + piece * 1024;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
                }
            }
        }
    };
#endif
    // LDS-DMA of a work item's halo tile into buffer `buf` of this half.  Slot i of thread t = halo pixel (t >> 2) + 64 i,
    // chunk position t & 3 - i.e. wave w's instruction i fills the 16 rows 64 i + 16 w .. + 15 of the image, lane l at
    // byte 16 l of that 1 KiB piece - so the lane fetches the LOGICAL chunk (t & 3) ^ ((row >> 2) & 3) of its pixel
    // (row >> 2 = (t >> 4) mod 4 for every i).  Addresses as in the plain loader below (scalar image base + two 24-bit
    // mads).  Slots that must read as zero (conv padding, channels beyond Cin) are not fetched but zero-filled.
    auto issue_dma = [&](int n, int kc, int ty0, int tx0, int buf) {
        if (DBG(p) & 4) return;
        const int ns = __builtin_amdgcn_readfirstlane(n), kcs = __builtin_amdgcn_readfirstlane(kc);
        const int ty0s = __builtin_amdgcn_readfirstlane(ty0), tx0s = __builtin_amdgcn_readfirstlane(tx0);
        const int bufs = __builtin_amdgcn_readfirstlane(buf), w0 = __builtin_amdgcn_readfirstlane(wave);
        const int c0 = kcs * (kRowBytes / (int)sizeof(T)) + ((t & 3) ^ ((t >> 4) & 3)) * VEC;
        const bool w1 = p.nsrc > 1 && c0 >= p.src[0].C;        // per lane: a chunk may straddle the two concat sources
        const int cs = w1 ? c0 - p.src[0].C : c0;
        const int Cs = w1 ? p.src[1].C : p.src[0].C;
        const bool cok = cs < Cs;
        const char* b0 = image_base(p.src[0].ptr, ns, p.src[0].img_bytes);
        const char* b1 = image_base(p.src[1].ptr, ns, p.src[1].img_bytes);
        const char* base = w1 ? b1 : b0;
        const unsigned Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W;
        const int ys0 = ty0s - PAD - (w1 ? p.src[1].off_y : p.src[0].off_y);
        const int xs0 = tx0s - PAD - (w1 ? p.src[1].off_x : p.src[0].off_x);
        const unsigned C2 = Cs * (unsigned)sizeof(T), cbytes = (cok ? cs : 0) * (unsigned)sizeof(T);
        char* hb = lds_halo + bufs * kDmaHaloBytes;
        const unsigned hb_s = (unsigned)(size_t)(__attribute__((address_space(3))) char*)hb;
        u32x4 zv = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            // (slots beyond the halo have hyx = -1: x = xs0 + 0xffff is out of range for every W < 32768)
            const unsigned y = ys0 + (hyx[i] >> 16), x = xs0 + (hyx[i] & 0xffff);
            const bool ok = cok & (y < Hs) & (x < Ws);
            const unsigned off = mad_u24(mad_u24(y, Ws, x), C2, cbytes);
            const int row0 = 64 * i + 16 * w0;                 // wave-uniform: first row of this instruction's piece
            if (ok) lds_dma16(base + off, (unsigned)__builtin_amdgcn_readfirstlane((int)(hb_s + row0 * 64)));
            else if (i < 4 || hyx[i] >= 0) *reinterpret_cast<u32x4*>(hb + row0 * 64 + lane * 16) = zv;
        }
    };
    auto issue = [&](int n, int kc, int ty0, int tx0) {
        if (DBG(p) & 4) return;
        if constexpr (DMA) return;
        if constexpr (NH == 1) {
            // Plain loader.  Everything is derived per item from wave-uniform scalars (image, tile origin, chunk: SALU
            // after the readfirstlanes) and the packed halo-slot coordinates: no per-tile geometry registers, no 64-bit
            // per-lane multiplies.  Address = scalar image base of the lane's source + a 32-bit byte offset
            // (y * W + x) * C * sizeof(T) + channel bytes from two 24-bit mads (host-checked ranges).  Every slot loads
            // (out-of-image / padding slots from the image base, a valid address); their validity travels as a lane mask.
            const int ns = __builtin_amdgcn_readfirstlane(n), kcs = __builtin_amdgcn_readfirstlane(kc);
            const int ty0s = __builtin_amdgcn_readfirstlane(ty0), tx0s = __builtin_amdgcn_readfirstlane(tx0);
            const int c0 = kcs * (kRowBytes / (int)sizeof(T)) + (t & 3) * VEC;
            const bool w1 = p.nsrc > 1 && c0 >= p.src[0].C;        // per lane: a chunk may straddle the two concat sources
            const int cs = w1 ? c0 - p.src[0].C : c0;
            const int Cs = w1 ? p.src[1].C : p.src[0].C;
            const bool cok = cs < Cs;
            if (wave == 0) {
                // the chunk's per-channel affine: ONE dword per lane of the half's first wave (lanes 0-31 scale, 32-63
                // shift of channel kc*BK + lane%32) instead of four 16-byte loads in every thread; it goes to an LDS
                // table at the end of the next matrix phase and is read back by the commit after that
                constexpr int BKE = kRowBytes / (int)sizeof(T);
                const int j = lane & 31, ch = kcs * BKE + j;
                const bool wj = p.nsrc > 1 && ch >= p.src[0].C;
                const int cj = wj ? ch - p.src[0].C : ch;
                // (opaque local copies: a select between two fields of `p` is otherwise folded into a dynamically indexed
                // load of the struct, which then lives in scratch memory)
                const float *s0p = p.src[0].scale, *s1p = p.src[1].scale, *h0p = p.src[0].shift, *h1p = p.src[1].shift;
                int C0j = p.src[0].C, C1j = p.src[1].C, m0j = p.src[0].mode, m1j = p.src[1].mode;
                asm volatile("" : "+s"(s0p), "+s"(s1p), "+s"(h0p), "+s"(h1p), "+s"(C0j), "+s"(C1j), "+s"(m0j), "+s"(m1j));
                const int Cj = wj ? C1j : C0j, mj = wj ? m1j : m0j;
                const float* tj = lane < 32 ? (wj ? s1p : s0p) : (wj ? h1p : h0p);
                float v = lane < 32 ? 1.f : 0.f;
                if (j < BKE && mj == MRISR_SRC_NORM && cj < Cj) v = gload<float>(tj + (size_t)ns * Cj + cj);
                pf.aff = v;
            }
            pf.mode = w1 ? p.src[1].mode : p.src[0].mode;
            pf.slope = pf.mode == MRISR_SRC_NORM ? LRELU_SLOPE : (pf.mode == MRISR_SRC_RELU ? 0.f : 1.f);
            const char* b0 = image_base(p.src[0].ptr, ns, p.src[0].img_bytes);
            const char* b1 = image_base(p.src[1].ptr, ns, p.src[1].img_bytes);
            const char* base = w1 ? b1 : b0;
            const unsigned Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W;
            const int ys0 = ty0s - PAD - (w1 ? p.src[1].off_y : p.src[0].off_y);
            const int xs0 = tx0s - PAD - (w1 ? p.src[1].off_x : p.src[0].off_x);
            const unsigned C2 = Cs * (unsigned)sizeof(T), cbytes = (cok ? cs : 0) * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                // (slots beyond the halo have hyx = -1: x = xs0 + 0xffff is out of range for every W < 32768)
                const unsigned y = ys0 + (hyx[i] >> 16), x = xs0 + (hyx[i] & 0xffff);
                const bool ok = cok & (y < Hs) & (x < Ws);
                const unsigned off = mad_u24(mad_u24(y, Ws, x), C2, cbytes);
                pf.h[i][0] = gload_vec16(reinterpret_cast<const T*>(base + (ok ? off : 0u)));
                pf.ok[i] = ok;
            }
            return;
        }
        if constexpr (pf_halo) {
            const int c0 = kc * (kRowBytes / (int)sizeof(T)) + (t & 3) * VEC;
            int which = 0, cs = c0;
            if (cs >= p.src[which].C) cs = -1;
            load_affine<VEC>(p.src[which], n, cs, pf.sc, pf.sh);
            if constexpr (NH == 2) load_affine<VEC>(p.src[1], n, cs, pf.sc1, pf.sh1);
            pf.mode = p.src[which].mode;
            const T* base = (const T*)p.src[which].ptr;
            int mask = 0;
#pragma unroll
            for (int i = 0; i < kMaxHaloIter; ++i) {
                if constexpr (NH == 2) {
                    const int o0 = geom.off0[i], o1 = geom.off1[i];
                    if (o0 >= 0 && o1 >= 0 && cs >= 0) {
                        pf.h[i][0] = gload_vec16(base + o0 + cs);
                        pf.h[i][1] = gload_vec16((const T*)p.src[1].ptr + o1 + cs);
                        mask |= 1 << i;
                    } else { pf.h[i][0].zero(); pf.h[i][1].zero(); }
                } else {
                    const int o = geom.off0[i];
                    if (o >= 0 && cs >= 0) {
                        int d1, d2;
                        if constexpr (GSP == MRISR_SP_POOL2) { d1 = p.src[0].C; d2 = p.src[0].W * p.src[0].C; }
                        else {
                            d1 = ((upflags >> (2 * i + 1)) & 1) ? p.src[0].C : 0;
                            d2 = ((upflags >> (2 * i)) & 1) ? p.src[0].W * p.src[0].C : 0;
                        }
                        const T* b = base + o + cs;
                        pf.h[i][0] = gload_vec16(b); pf.h[i][1] = gload_vec16(b + d1);
                        pf.h[i][2] = gload_vec16(b + d2); pf.h[i][3] = gload_vec16(b + d2 + d1);
                        mask |= 1 << i;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) pf.h[i][q].zero();
                    }
                }
            }
            pf.mask = mask;
        }
    };
    // transform + store the prefetched item into LDS (gather modes: stage synchronously)
    auto commit = [&](int n, int kc, int ty0, int tx0) {
        if (DBG(p) & 2) return;
        if constexpr (DMA) return;
        if constexpr (NH == 1) {
            // straight-line: y = x*sc+sh, act = max(y, slope*y), unpredicated 16-B LDS
            // store of every slot; the slots that must read as zero (conv padding, channels beyond Cin) are then
            // overwritten by an exec-masked zero store - no per-element selects, and nothing at all inside the image
            if (!all_raw) {   // (all sources stored as-is - input gradients, materialised activations, VGG: no arithmetic)
                float sc[VEC], sh[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(lds_aff + (t & 3) * VEC + e);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_aff + 32 + (t & 3) * VEC + e);
                    sc[e] = a4[0]; sc[e + 1] = a4[1]; sc[e + 2] = a4[2]; sc[e + 3] = a4[3];
                    sh[e] = b4[0]; sh[e + 1] = b4[1]; sh[e + 2] = b4[2]; sh[e + 3] = b4[3];
                }
                if constexpr (std::is_same<T, f16_t>::value) {
                    // fp16 storage: the transform runs on packed halves - v_pk_fma_f16, v_pk_mul_f16, v_pk_max_f16 = 1.5
                    // VALU instructions per element where the bf16 path needs ~4.5 (unpack, fma, mul, max, pack).  One
                    // fp16 rounding per operation (the reference's autocast rounds once, after fp32 GroupNorm + LeakyReLU);
                    // covered by the fp16 parity tolerances.
                    f16x2 sc2[VEC / 2], sh2[VEC / 2];
#pragma unroll
                    for (int e = 0; e < VEC; e += 2) {
                        sc2[e / 2] = f16x2{(f16_t)sc[e], (f16_t)sc[e + 1]};
                        sh2[e / 2] = f16x2{(f16_t)sh[e], (f16_t)sh[e + 1]};
                    }
                    const f16_t sl = (f16_t)pf.slope;
                    const f16x2 sl2 = {sl, sl};
#pragma unroll
                    for (int i = 0; i < NSLOT; ++i) {
#pragma unroll
                        for (int e = 0; e < VEC; e += 2) {
                            const f16x2 x2 = {pf.h[i][0].v[e], pf.h[i][0].v[e + 1]};
                            const f16x2 y2 = x2 * sc2[e / 2] + sh2[e / 2];
                            const f16x2 a2 = __builtin_elementwise_max(y2, y2 * sl2);
                            pf.h[i][0].v[e] = a2[0];
                            pf.h[i][0].v[e + 1] = a2[1];
                        }
                    }
                } else {
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    // (scalar fp32 ops on purpose: measured with the per-wave phase profile, v_pk_fma_f32 / v_pk_mul_f32 in
                    // this loop run at half speed whenever the SIMD's other wave is in its MFMA block)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float y = fmaf(pf.h[i][0].get(e), sc[e], sh[e]);
                        pf.h[i][0].set(e, fmaxf(y, pf.slope * y));
                    }
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two slots' temporaries live at a time
                }
                }
            }
#pragma unroll
            for (int i = 0; i < NSLOT; ++i)
                *reinterpret_cast<decltype(pf.h[i][0].v)*>(lds_halo + halo_off((t >> 2) + 64 * i, t & 3)) = pf.h[i][0].v;
            Vec16<T> zv;
            zv.zero();
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                // (slots beyond the halo tile - only possible for i >= 4 - are never read: leave them alone)
                const bool z = !pf.ok[i] && (i < 4 || hyx[i] >= 0);
                if (z) *reinterpret_cast<decltype(zv.v)*>(lds_halo + halo_off((t >> 2) + 64 * i, t & 3)) = zv.v;
            }
        } else if constexpr (pf_halo) {
#pragma unroll
            for (int i = 0; i < kMaxHaloIter; ++i) {
                Vec16<T> v = pf.h[i][0];
                if ((pf.mask >> i) & 1) {
                    if constexpr (NH == 2) {
                        float fa[VEC], fb[VEC];
                        transform_f(pf.h[i][0], fa, p.src[0].mode, pf.sc, pf.sh);
                        transform_f(pf.h[i][1], fb, p.src[1].mode, pf.sc1, pf.sh1);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) v.set(e, blend_a * fa[e] + (1.f - blend_a) * fb[e]);
                    } else {
                        float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
                        transform_f(pf.h[i][0], f0, pf.mode, pf.sc, pf.sh);
                        transform_f(pf.h[i][1], f1, pf.mode, pf.sc, pf.sh);
                        transform_f(pf.h[i][2], f2, pf.mode, pf.sc, pf.sh);
                        transform_f(pf.h[i][3], f3, pf.mode, pf.sc, pf.sh);
                        if constexpr (GSP == MRISR_SP_POOL2) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) v.set(e, fmaxf(fmaxf(f0[e], f1[e]), fmaxf(f2[e], f3[e])));
                        } else {
                            // interpolation weights re-derived from the coordinates (cheaper than 12 live VGPRs)
                            int i0, i1;
                            float wy1, wx1;
                            up2_coord(ty0 + (hyx[i] >> 16) - PAD - p.src[0].off_y, p.src[0].H, i0, i1, wy1);
                            up2_coord(tx0 + (hyx[i] & 0xffff) - PAD - p.src[0].off_x, p.src[0].W, i0, i1, wx1);
                            const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                v.set(e, wy0 * (wx0 * f0[e] + wx1 * f1[e]) + wy1 * (wx0 * f2[e] + wx1 * f3[e]));
                        }
                    }
                }
                if (hyx[i] >= 0) *reinterpret_cast<decltype(v.v)*>(lds_halo + halo_off((t >> 2) + 64 * i, t & 3)) = v.v;
            }
        } else {
            stage_halo<T, GSP>(lds_halo, geom, kc, n, npix_halo, blend_a, p, 0, t);
        }
    };
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int co = bn0 + ni * 32 + 8 * q + 4 * lh;
                // opaque: otherwise the 8 group offsets (co / gs, 64-bit) are hoisted out of the persistent loop
                // and sit in 16 VGPRs for a value needed once per image
                asm volatile("" : "+v"(co));
                const float s = half_wave_sum(st_s[ni][q]), ss = half_wave_sum(st_ss[ni][q]);
                if (lr == 0 && co < p.Cout) {
                    const int g = co / gs;
                    double* sp = p.stats + stat_slot_off_id(bid, p.N, p.groups) + ((size_t)n * p.groups + g) * 2;
                    atomic_add_f64(sp, (double)s);
                    atomic_add_f64(sp + 1, (double)ss);
                }
                st_s[ni][q] = 0.f;
                st_ss[ni][q] = 0.f;
            }
    };
    // epilogue of a finished tile: bias, (relu), NHWC / pixel-shuffled store, per-lane GroupNorm partial sums.
    // Straight-line per (pixel row mi, cout fragment ni): no per-quad branches (the bias comes from LDS, pixels
    // outside the image and channels >= Cout are handled by multiplying the statistics with 0/1 and by predicating
    // only the stores); bias / ReLU sit behind one block-uniform branch per fragment.
    auto epilogue = [&](int n_v, int ty0_v, int tx0_v) {
        if (DBG(p) & 16) return;
        // Addresses: scalar base of (image, tile origin, first channel of this cout block) - SALU after the
        // readfirstlanes - plus one 32-bit per-lane byte offset per pixel row (two 24-bit mads) plus compile-time
        // constants for the fragment / quad: no 64-bit per-lane arithmetic in front of the stores.
        const int n = __builtin_amdgcn_readfirstlane(n_v), ty0 = __builtin_amdgcn_readfirstlane(ty0_v);
        const int tx0 = __builtin_amdgcn_readfirstlane(tx0_v);
        constexpr bool kPSE = EPI == MRISR_OUT_PIXEL_SHUFFLE2;
        const int C4 = p.Cout >> 2;
        // element offset of the tile origin: plain (n, ty0, tx0, bn0); pixel-shuffle (n, 2 ty0, 2 tx0, bn0 / 4)
        const size_t e0 = kPSE ? ((size_t)(n * 2 * p.H + 2 * ty0) * (2 * p.W) + 2 * tx0) * C4 + (bn0 >> 2)
                               : ((size_t)(n * p.H + ty0) * p.W + tx0) * p.Cout + bn0;
        char* obase = (char*)p.out + e0 * sizeof(T);
        const char* mbase = (const char*)p.mask + e0 * sizeof(T);     // (kEpiMask only)
        int lh_e = lh;      // opaque copy: channel-dependent offsets are recomputed per tile, not hoisted (and spilled)
        asm volatile("" : "+v"(lh_e));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int pl = wave * 64 + mi * 32 + lr;
            const int py = pl >> p.tw_log2, px = pl & (TW - 1);
            const int oy = ty0 + py, ox = tx0 + px;
            const bool pv = oy < p.H && ox < p.W;
            const float pvf = pv ? 1.f : 0.f;
            // per-lane byte offset of this pixel relative to the tile origin (+ the lane's channel sub-offset)
            const unsigned loff = kPSE ? mad_u24(mad_u24(2 * py, 2 * p.W, 2 * px), C4 * (unsigned)sizeof(T), 4 * lh_e * (unsigned)sizeof(T))
                                       : mad_u24(mad_u24(py, p.W, px), p.Cout * (unsigned)sizeof(T),
                                                 (sizeof(T) == 2 ? 8 : 4) * lh_e * (unsigned)sizeof(T));
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                if (has_br) {   // block-uniform: bias (from LDS) and ReLU applied in place on the accumulators
                    const float floor_v = p.relu_out ? 0.f : -INFINITY;
                    const float* bl = lds_bias + 4 * lh_e;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // (channels >= Cout: zero weights and a zero LDS bias keep them at exactly 0)
                        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ni * 32 + 8 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ni][mi][4 * q + j] = fmaxf(acc[ni][mi][4 * q + j] + b[j], floor_v);
                    }
                }
                u32x2 packed[4];          // bf16 plain epilogue: the 4 quads of this fragment, packed
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = bn0 + ni * 32 + 8 * q + 4 * lh_e;      // first of 4 consecutive couts
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[ni][mi][4 * q + j];
                        acc[ni][mi][4 * q + j] = 0.f;
                    }
                    if (has_stats) {   // block-uniform: input-gradient and VGG launches carry no GroupNorm statistics
                        const float qs = (v[0] + v[1]) + (v[2] + v[3]);
                        const float qq = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                        st_s[ni][q] = fmaf(pvf, qs, st_s[ni][q]);
                        st_ss[ni][q] = fmaf(pvf, qq, st_ss[ni][q]);
                    }
                    if constexpr (sizeof(T) == 2) {   // bf16: packed, stored after the lane exchange below
                        typedef __attribute__((ext_vector_type(4))) T t4_t;     // bf16x4 / f16x4
                        union { t4_t b; u32x2 u; } cv;
                        cv.b = t4_t{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                        packed[q] = cv.u;
                    } else if (pv && co < p.Cout && !(DBG(p) & 1)) {
                        if constexpr (!kPSE) {
                            const unsigned o = loff + (ni * 32 + 8 * q) * (unsigned)sizeof(T);
                            if constexpr (EPI == kEpiMask) {   // ReLU backward: keep the gradient where the activation is > 0
                                const f32x4 m = gload<f32x4>(mbase + o);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
                            }
                            gstore(obase + o, f32x4{v[0], v[1], v[2], v[3]});
                        } else {   // PixelShuffle(2): channel 4c'+2i+j -> (2y+i, 2x+j, c'); this lane: c' = (ni*32 + 8q)/4 + lh
                            const unsigned o = mad_u24(mad_u24(2 * py, 2 * p.W, 2 * px), C4 * (unsigned)sizeof(T),
                                                       (ni * 8 + 2 * q + lh_e) * (unsigned)sizeof(T));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                gstore(obase + o + ((size_t)(j >> 1) * (2 * p.W) + (j & 1)) * C4 * sizeof(T), from_f32<T>(v[j]));
                        }
                    }
                }
                if constexpr (!kPSE && sizeof(T) == 2) {
                    // lanes l and l+32 hold the two 4-channel halves of each 8-channel group of the same pixel:
                    // exchange so that every lane owns 8 consecutive channels -> 16-byte stores (half the store
                    // instructions).  Quad pair (q, q+1): low half keeps group q, high half keeps group q+1.
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        u32x2 a = packed[q], b = packed[q + 1];
                        auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                        u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
#ifdef MRISR_STAGED_STORES
                        // stage: row = pixel (lr), column = this lane's 8 channels
                        *reinterpret_cast<u32x4*>(lds_stage + lr * kStageRow + (ni * 32 + 8 * (q + lh_e)) * 2) = o;
                        continue;
#endif
                        const int co8 = bn0 + ni * 32 + 8 * (q + lh_e);      // first of the 8 channels this lane now owns
                        const unsigned ob = loff + (ni * 32 + 8 * q) * (unsigned)sizeof(T);
                        if (pv && co8 < p.Cout && !(DBG(p) & 1)) {
                            if constexpr (EPI == kEpiMask) {   // ReLU backward on packed bf16 pairs
                                const u32x4 m = gload<u32x4>(mbase + ob);
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    // (positive and non-zero <=> the 16 bits read as int16 are > 0, for bf16 and fp16 alike)
                                    const unsigned lo = (short)(m[k] & 0xffffu) > 0 ? 0x0000ffffu : 0u;
                                    const unsigned hi = ((int)m[k] >> 16) > 0 ? 0xffff0000u : 0u;
                                    o[k] &= (lo | hi);
                                }
                            }
                            gstore(obase + ob, o);
                        }
                    }
                }
                if constexpr (kPSE && sizeof(T) == 2) {
                    // PixelShuffle(2): conv channel 4c'+2i+j -> pixel (2y+i, 2x+j), channel c'.  packed[q] = the four
                    // (i,j) values of c' = cb + 2q + lh_e.  For one (i,j): this lane holds c' = cb + {0,2,4,6} + lh_e; one
                    // permlane32 swap + a 16-bit interleave give the low half c' = cb..cb+3 and the high half
                    // cb+4..cb+7 -> one 8-byte store per (i,j) instead of four 2-byte stores.
                    const int cb = (bn0 + ni * 32) >> 2;
                    const bool okc = cb + 4 * lh_e < C4;                 // (C4 % 4 == 0: host-checked for this epilogue)
#pragma unroll
                    for (int ij = 0; ij < 4; ++ij) {
                        // 16-bit element ij of packed[q]: dword ij>>1, half ij&1
                        auto pick = [&](int qa, int qb) {   // bf16x2 {c'(qa), c'(qb)} of this (i,j)
                            const unsigned a = packed[qa][ij >> 1], b = packed[qb][ij >> 1];
                            return (ij & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                        };
                        const unsigned P0 = pick(0, 1), P1 = pick(2, 3);          // own c' = {0,2}+lh_e and {4,6}+lh_e
                        auto r = __builtin_amdgcn_permlane32_swap(P0, P1, false, false);   // P0.hi-lanes <-> P1.lo-lanes
                        const unsigned X = r[0], Y = r[1];   // low half: X={0,2} Y={1,3}; high half: X={4,6} Y={5,7}
                        const u32x2 o = {(X & 0xffffu) | (Y << 16), (X >> 16) | (Y & 0xffff0000u)};
                        // scalar: sub-pixel (i, j) and the fragment's first c'
                        char* ob = obase + (((size_t)(ij >> 1) * (2 * p.W) + (ij & 1)) * C4 + ni * 8) * sizeof(T);
                        if (pv && okc && !(DBG(p) & 1)) gstore(ob + loff, o);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // bound the scheduling window: one (mi, ni) group's temporaries live at a time
            }
#ifdef MRISR_STAGED_STORES
            if constexpr (!kPSE && sizeof(T) == 2) {
                constexpr int CH = BN / 8, PPI = 64 / CH, NIT = 32 / PPI;     // 16-B chunks per pixel, pixels per store, stores
                const int chunk = lane & (CH - 1), prow = lane / CH;
                const bool cv = bn0 + chunk * 8 < p.Cout;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int pp = it * PPI + prow;                           // pixel (row of the staging tile) of this lane
                    const u32x4 v = *reinterpret_cast<const u32x4*>(lds_stage + pp * kStageRow + chunk * 16);
                    const int pl2 = wave * 64 + mi * 32 + pp;
                    const int py2 = pl2 >> p.tw_log2, px2 = pl2 & (TW - 1);
                    const unsigned o2 = mad_u24(mad_u24(py2, p.W, px2), p.Cout * 2u, chunk * 16u);
                    if (ty0 + py2 < p.H && tx0 + px2 < p.W && cv && !(DBG(p) & 1)) {
                        u32x4 o = v;
                        if constexpr (EPI == kEpiMask) {   // ReLU backward on packed bf16 pairs
                            const u32x4 m = gload<u32x4>(mbase + o2);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned lo = __uint_as_float(m[k] << 16) > 0.f ? 0x0000ffffu : 0u;
                                const unsigned hi = __uint_as_float(m[k] & 0xffff0000u) > 0.f ? 0xffff0000u : 0u;
                                o[k] &= (lo | hi);
                            }
                        }
                        gstore(obase + o2, o);
                    }
                }
            }
#endif
        }
    };
    // ---- the MFMA block of one work item: halo tile at `hbuf`, weight image at `wl`
    auto run_mma = [&](const char* hbuf, const char* wl, auto&& pace) {
            constexpr bool PACED = !std::is_same<std::decay_t<decltype(pace)>, NoPace>::value;
            if constexpr (PACED) {
                // Symmetric schedule: the same one-step-ahead pipeline, with the LDS-DMA instructions of the NEXT item
                // dealt out one per step (`pace(st)`) instead of issued as a burst: a burst of ~10 DMAs per wave fills
                // the CU's memory queue and stalls each issuing wave for ~2-3 k cycles (measured: 1700-2700 cycles of
                // issue per item and wave = as long as its MFMA block), during which it issues no MFMA either; paced,
                // the SIMD's other wave keeps the matrix pipe busy.  sched_barrier(0) pins the per-step order
                // (reads of step s+1 / DMA -> MFMAs of step s): inline asm is invisible to sched_group_barrier.
                constexpr int NSTEP = 2 * NTAPS;
                frag_t xf[2][2], wf[2][NF];
                auto load_step = [&](int st, int buf) {
                    const int tap = st >> 1, ks = st & 1;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) xf[buf][mi] = *reinterpret_cast<const frag_t*>(hbuf + (xa[mi][tap] ^ (32 * ks)));
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni)
                        wf[buf][ni] = *reinterpret_cast<const frag_t*>(wl + ((wb ^ (32 * ks)) + (tap * BN + ni * 32) * kRowBytes));
                };
                load_step(0, 0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) load_step(st + 1, (st + 1) & 1);
                    pace(st);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG(p) & 8)) {
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mma<T>::run(wf[st & 1][ni], xf[st & 1][mi], acc[ni][mi]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#ifndef MRISR_NO_PIPE_MMA
            if (!(DBG(p) & 8)) {
                // 2 * NTAPS steps of (2 pixel fragments, NF weight fragments, 2 * NF MFMAs), software-pipelined by one step
                // with two fragment sets: the LDS reads of step s+1 are issued before the MFMAs of step s; the order is
                // pinned with sched_group_barrier (hipcc otherwise sinks the reads back next to their use and waits for
                // them at every tap).  Alone the block takes 2.8 k instead of 3.9 k cycles (ideal 72 x 37 = 2.7 k); it
                // paid in the training step (+1.3 %, A/B on one box) only once the vector phase had been trimmed.
                constexpr int NSTEP = 2 * NTAPS;
                frag_t xf[2][2], wf[2][NF];
                auto load_step = [&](int st, int buf) {
                    const int tap = st >> 1, ks = st & 1;
                    const int tapoff = ((tap / KS) * hw + (tap % KS)) * kHaloRowBytes;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        xf[buf][mi] = *reinterpret_cast<const frag_t*>(hbuf + xb[mi] + tapoff + 32 * ks);
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni)
                        wf[buf][ni] = *reinterpret_cast<const frag_t*>(wl + ((wb ^ (32 * ks)) + (tap * BN + ni * 32) * kRowBytes));
                };
                load_step(0, 0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) load_step(st + 1, (st + 1) & 1);
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mma<T>::run(wf[st & 1][ni], xf[st & 1][mi], acc[ni][mi]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + NF, 0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NF, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NF, 0);
                }
            }
#else
            if (!(DBG(p) & 8)) {
#pragma unroll
                for (int tap = 0; tap < NTAPS; ++tap) {
                    const int tapoff = ((tap / KS) * hw + (tap % KS)) * kHaloRowBytes;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        frag_t xf[2], wf[NF];
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
                            xf[mi] = *reinterpret_cast<const frag_t*>(hbuf + xb[mi] + tapoff + 32 * ks);
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni)
                            wf[ni] = *reinterpret_cast<const frag_t*>(wl + ((wb ^ (32 * ks)) + (tap * BN + ni * 32) * kRowBytes));
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mma<T>::run(wf[ni], xf[mi], acc[ni][mi]);
                    }
                }
            }
#endif
    };
    // ---- paced LDS-DMA (symmetric schedule): plan_dma computes what issue_dma / dma_weights would issue for an item,
    // pace_dma(step) issues ONE of those instructions (or the zero fill of a padding slot)
    struct DmaPlan {
        unsigned off[NSLOT];      // byte offset of slot i's 16 bytes from `base`
        unsigned okmask, zmask;   // bit i: slot i is fetched / slot i is zero-filled (conv padding, channels beyond Cin)
        const char* base;         // this lane's source image
        char* hb;                 // destination halo buffer
        const char* wsrc;         // next weight image in global memory + lane * 16
        unsigned wdst;            // LDS address of the weight image to fill
        bool do_halo, do_w;
    };
    auto plan_dma = [&](DmaPlan& pl, int n, int kc, int ty0, int tx0, int buf, int item, bool do_halo, bool do_w) {
        const int ns = __builtin_amdgcn_readfirstlane(n), kcs = __builtin_amdgcn_readfirstlane(kc);
        const int ty0s = __builtin_amdgcn_readfirstlane(ty0), tx0s = __builtin_amdgcn_readfirstlane(tx0);
        const int bufs = __builtin_amdgcn_readfirstlane(buf), its = __builtin_amdgcn_readfirstlane(item);
        const int c0 = kcs * (kRowBytes / (int)sizeof(T)) + ((t & 3) ^ ((t >> 4) & 3)) * VEC;
        const bool w1 = p.nsrc > 1 && c0 >= p.src[0].C;
        const int cs = w1 ? c0 - p.src[0].C : c0;
        const int Cs = w1 ? p.src[1].C : p.src[0].C;
        const bool cok = cs < Cs;
        const char* b0 = image_base(p.src[0].ptr, ns, p.src[0].img_bytes);
        const char* b1 = image_base(p.src[1].ptr, ns, p.src[1].img_bytes);
        pl.base = w1 ? b1 : b0;
        const unsigned Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W;
        const int ys0 = ty0s - PAD - (w1 ? p.src[1].off_y : p.src[0].off_y);
        const int xs0 = tx0s - PAD - (w1 ? p.src[1].off_x : p.src[0].off_x);
        const unsigned C2 = Cs * (unsigned)sizeof(T), cbytes = (cok ? cs : 0) * (unsigned)sizeof(T);
        pl.hb = lds_halo + bufs * kDmaHaloBytes;
        unsigned okm = 0, zm = 0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const unsigned y = ys0 + (hyx[i] >> 16), x = xs0 + (hyx[i] & 0xffff);
            const bool ok = cok & (y < Hs) & (x < Ws);
            pl.off[i] = mad_u24(mad_u24(y, Ws, x), C2, cbytes);
            okm |= (ok ? 1u : 0u) << i;
            zm |= ((!ok && (i < 4 || hyx[i] >= 0)) ? 1u : 0u) << i;
        }
        pl.okmask = okm; pl.zmask = zm;
        pl.do_halo = do_halo; pl.do_w = do_w;
        pl.wsrc = wbase + (size_t)kcs * (WIMG_VECS * 16) + lane * 16;
        pl.wdst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_w + (unsigned)(its & 1) * (WIMG_VECS * 16);
    };
    auto pace_dma = [&](const DmaPlan& pl, int st) {
        if (DBG(p) & 4) return;
        constexpr int NPIECE = WS ? 0 : WIMG_VECS * 16 / 1024, NWJ = (NPIECE + 7) / 8;
        constexpr int PER = KS == 3 ? 1 : 8;           // operations per step (a 1x1 conv has only two steps)
#pragma unroll
        for (int k = st * PER; k < (st + 1) * PER; ++k) {
            if ((k & 1) == 0) {
                const int i = k >> 1;
                if (i < NSLOT && pl.do_halo) {
                    const int w0 = __builtin_amdgcn_readfirstlane(wave);
                    const int row0 = 64 * i + 16 * w0;
                    if ((pl.okmask >> i) & 1) {
                        const unsigned hb_s = (unsigned)(size_t)(__attribute__((address_space(3))) char*)pl.hb;
                        lds_dma16(pl.base + pl.off[i], (unsigned)__builtin_amdgcn_readfirstlane((int)(hb_s + row0 * 64)));
                    } else if ((pl.zmask >> i) & 1) {
                        *reinterpret_cast<u32x4*>(pl.hb + row0 * 64 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
                    }
                }
            } else {
                const int j = k >> 1;
                if (j < NWJ && pl.do_w) {
                    const int piece = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + 8 * j;
                    if (piece < NPIECE) lds_dma16(pl.wsrc + piece * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(pl.wdst + piece * 1024)));
                }
            }
        }
    };
    // ---- DMA variant: SYMMETRIC schedule.  With the halo tile and the weight image both arriving by LDS-DMA a wave has
    // almost no vector work left, so the antiphase roles are dropped: both halves run the same loop - wait for item c's
    // DMAs, ONE workgroup barrier per item, issue the DMAs of item c + 1 into the other buffers, the 72 MFMAs of item c,
    // and the epilogue after a tile's last chunk - and the two waves of a SIMD fill each other's LDS-wait / epilogue
    // gaps on the matrix pipe.  Measured (profiles/r02_conv_dma_schedules.txt, 128 -> 128 at 128^2, batch 16): antiphase
    // with DMA 82-88 us, symmetric with a DMA burst at the top of each item 80-85 us, symmetric with paced DMAs 80-82 us -
    // the schedule hardly matters: the loop is power-limited (shader clock 1.55-1.65 GHz under this load), with the
    // MFMA + LDS-read part alone at 58.6 us (81 % matrix-pipe occupancy) and the DMA path alone at 46 us.  The
    // symmetric form is kept for DMA sources because it has ONE barrier per item and no role state.  Buffers: item c reads
    // halo buffer c & 1 of its half and weight image c & 1; the DMAs for item c + 1 are issued after the barrier that
    // every wave passes once its MFMAs of item c - 1 (the last readers of those buffers) are done.
    if constexpr (DMA) {
        // Optional stagger (-DMRISR_STAGGER, tuning builds): half 1 runs ONE item behind half 0 - both halves use the same
        // chunk per step (the streamed weight image is shared), so half 1's tiles take their chunks in the rotated order
        // 1, 2, ..., n-1, 0 and its epilogues fall on the step after half 0's, while the SIMD's other wave runs MFMAs.
        // Measured neutral in the training step (64-channel weights-stationary kernel 124.3 -> 125.8 us, streamed kernel
        // 84.0 -> 84.6 us, A/B on one box): like the schedule variants before it - the loop is power-limited, what is on
        // the SIMD at the same time does not matter.  Off by default.
#ifdef MRISR_STAGGER
        const int stag = (__builtin_amdgcn_readfirstlane(half) == 1 && p.nchunks >= 2) ? 1 : 0;
        const int nitems1 = (nbt - nh0) * p.nchunks;              // half 1's items (workgroup-uniform)
        const int total = max(nh0 * p.nchunks, nitems1 + ((nitems1 > 0 && p.nchunks >= 2) ? 1 : 0));
#else
        const int stag = 0;
        const int total = nh0 * p.nchunks;                        // steps of the workgroup (half 0 never has fewer tiles)
#endif
        int it_tile = tile0, it_jj = 0, it_n = 0, it_ty0 = 0, it_tx0 = 0;     // current item of this half: tile, chunks done in it
        int gkc = 0;                                                           // chunk of the current step (shared)
        if (stag == 0 && nitems > 0) {
            decode(it_tile, it_n, it_ty0, it_tx0);
            issue_dma(it_n, 0, it_ty0, it_tx0, 0);
        }
        if constexpr (!WS) {
            if (total > 0) dma_weights(0, 0, std::integral_constant<int, 8>{});
        }
        PT_DECL
        for (int c = 0; c < total; ++c) {
            PT_MARK(8)
            __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's pieces of step c (and older stores) have landed
            PT_MARK(0)
            __syncthreads();
            PT_MARK(5)
            const int j = c - stag;                   // this half's item index at this step
            const bool cur_valid = j >= 0 && j < nitems, nx_valid = j + 1 >= 0 && j + 1 < nitems;
            int nx_kc = gkc + 1;
            if (nx_kc == p.nchunks) nx_kc = 0;
            int nx_jj = it_jj, nx_tile = it_tile, nx_n = it_n, nx_ty0 = it_ty0, nx_tx0 = it_tx0;
            if (nx_valid) {
                if (j + 1 == 0) {                     // half 1's first item
                    nx_jj = 0;
                    decode(nx_tile, nx_n, nx_ty0, nx_tx0);
                } else if (++nx_jj == p.nchunks) {
                    nx_jj = 0;
                    nx_tile = it_tile + 1;
                    decode(nx_tile, nx_n, nx_ty0, nx_tx0);
                }
            }
            PT_MARK(2)
            // plan of the next step's DMAs (addresses and validity of this thread's halo slots, this wave's weight
            // pieces); they are issued one per MFMA step by `pace`
            DmaPlan plan;
            plan_dma(plan, nx_n, nx_kc, nx_ty0, nx_tx0, (c + 1) & 1, c + 1, nx_valid, c + 1 < total);
            PT_MARK(3)
            if (cur_valid) {
                asm volatile("" : "+v"(wb));     // (see the antiphase schedule: keeps the 36 weight tap addresses out of VGPRs)
                run_mma(lds_halo + (c & 1) * kDmaHaloBytes, lds_w + (size_t)(WS ? gkc : (c & 1)) * (WIMG_VECS * 16),
                        [&](int st) { pace_dma(plan, st); });
                PT_MARK(6)
                if (it_jj == p.nchunks - 1) {
                    epilogue(it_n, it_ty0, it_tx0);
                    PT_MARK(4)
                    if (p.stats && (!nx_valid || nx_n != it_n)) flush_stats(it_n);
                    PT_MARK(9)
                }
            } else {
                // this half has no item at this step but still owes its DMAs (its share of the next weight image, its
                // own next halo tile)
#pragma unroll
                for (int st = 0; st < 2 * NTAPS; ++st) pace_dma(plan, st);
            }
            gkc = nx_kc;
            it_jj = nx_jj; it_tile = nx_tile; it_n = nx_n; it_ty0 = nx_ty0; it_tx0 = nx_tx0;
        }
#ifdef MRISR_PHASE_TIMING
        if (blockIdx.x == gridDim.x / 2 && lane == 0) {
            pt_acc[10] = __builtin_amdgcn_s_memrealtime() - pt_r0;
#pragma unroll
            for (int k = 0; k < 12; ++k) atomicAdd(&g_phase_cycles[threadIdx.x >> 6][k], pt_acc[k]);
            if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[0][11], 1ull);
        }
#endif
        return;
    }
    // ---- schedule: commit phase c at tick 2c + half, MFMA phase c at tick 2c + 1 + half
    int cur_tile = tile0, cur_kc = 0, cur_n = 0, cur_ty0 = 0, cur_tx0 = 0;       // item c
    int nxt_tile = tile0, nxt_kc = 0, nxt_n = 0, nxt_ty0 = 0, nxt_tx0 = 0;       // item c + 1
    int ep_n = 0, ep_ty0 = 0, ep_tx0 = 0;
    bool ep_pending = false;
    // Streamed weights, shared between the halves: half 0 writes the image of the even items (at its own commit of that
    // item), half 1 the image of the odd items (one item ahead, at its commit of the even item before): an image is
    // written in a tick in which nobody reads it (item c: half 0 reads in tick 2c+1, half 1 in tick 2c+2; image c & 1 is
    // rewritten for item c+2 in tick 2c+3 or 2c+4), and every half loads / stores 9 vectors per thread for every OTHER
    // item instead of for every item (measured: the 9 weight ds_write_b128 were half of the commit time).  The duty runs
    // on half 0's item count, so half 1 keeps serving half 0's last tile when it has one tile less.
    const int nitems0 = nh0 * p.nchunks;
    int wkc = half;            // cin chunk of this half's next weight target (items half, half + 2, ...; nchunks >= 3 here)
    if (nitems > 0) {
        decode(cur_tile, cur_n, cur_ty0, cur_tx0);
        set_geom(cur_n, cur_ty0, cur_tx0);
        issue(cur_n, 0, cur_ty0, cur_tx0);
    }
#ifdef MRISR_DMA_WEIGHTS
    if constexpr (!WS) {
        if (half == 0 && nitems0 > 0) dma_weights(0, 0, std::integral_constant<int, 4>{});
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the image is in LDS before the barrier below
        if constexpr (NH != 1) __syncthreads();
    }
#else
    if constexpr (!WS) {
        if (half < nitems0) issue_weights(wkc);
    }
#endif
    if constexpr (NH == 1) {
        if (wave == 0 && nitems > 0) lds_aff[lane] = pf.aff;     // first item's table; later ones at the end of a matrix phase
        __syncthreads();
    }
    // static priority for the younger half (waves 4-7 lose the VALU arbitration to the older half of their SIMD on every
    // tick: measured 26 k vs 31 k cycles for the same commit work); a provably uniform condition, s_setprio ignores EXEC
#ifndef MRISR_NO_STATIC_PRIO
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
    PT_DECL
    for (int tick = 0; tick < nticks; ++tick) {
        const int phase = tick - half;
        const int c = phase >> 1;
        PT_MARK(8)
        if (phase >= 0 && (phase & 1) == 0) {
            // ------------------------------------------------ vector phase
#ifdef MRISR_DMA_WEIGHTS
            if constexpr (!WS) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): half 0's image DMA of the last matrix phase has landed
#else
            if constexpr (!WS) {
                if (!(c & 1) && c + half < nitems0) store_weights(c + half);
            }
#endif
            // Order: commit item c (its loads were issued one full tick pair ago) -> issue the loads of item c+1
            // right away (the prefetch registers are free again) -> only then the epilogue of the tile that finished
            // in the previous matrix phase.  The loads thus have the rest of this phase plus the whole matrix phase
            // to land; issued at the start of the matrix phase they had half of that and the commit stalled on
            // vmcnt (measured: 20-40 us per launch).
            if (c < nitems) {
                PT_WAIT_LOADS();
                PT_MARK(0)
                commit(cur_n, cur_kc, cur_ty0, cur_tx0);
                PT_MARK(1)
                nxt_tile = cur_tile; nxt_kc = cur_kc + 1; nxt_n = cur_n; nxt_ty0 = cur_ty0; nxt_tx0 = cur_tx0;
                if (nxt_kc == p.nchunks) {
                    nxt_kc = 0;
                    nxt_tile = cur_tile + 1;
                    if (nxt_tile < tile1) {
                        decode(nxt_tile, nxt_n, nxt_ty0, nxt_tx0);
                        set_geom(nxt_n, nxt_ty0, nxt_tx0);
                    }
                }
                PT_MARK(2)
                issue(nxt_n, nxt_kc, nxt_ty0, nxt_tx0);     // unconditional: after the last item this re-loads valid addresses and is never committed
                PT_MARK(3)
            }
            if (ep_pending) {
                epilogue(ep_n, ep_ty0, ep_tx0);
                PT_MARK(4)
                if (p.stats && (c >= nitems || cur_n != ep_n)) flush_stats(ep_n);
                ep_pending = false;
                PT_MARK(9)
            }
        } else if (phase >= 0) {
          // ------------------------------------------------ matrix phase
#ifdef MRISR_DMA_WEIGHTS
          if constexpr (!WS) {
              // half 0, every item: image (c + 1) & 1 (last read two ticks ago) <- weights of item c + 1; it lands during
              // this matrix phase and is waited for at the start of this half's next vector phase
              if (half == 0 && c < nitems && c + 1 < nitems0) dma_weights(nxt_kc, c + 1, std::integral_constant<int, 4>{});
          }
#else
          if constexpr (!WS) {
              // odd item index: load the image this half stores in its next vector phase (target item c + 1 + half)
              if ((c & 1) && c + 1 + half < nitems0) {
                  wkc += 2;
                  if (wkc >= p.nchunks) wkc -= p.nchunks;
                  issue_weights(wkc);
              } else {
                  // not loaded on this path: an empty asm "defines" the registers so that they are not live across the
                  // rest of the loop (36 VGPRs)
#pragma unroll
                  for (int j = 0; j < NW; ++j) asm volatile("" : "=v"(pf.w[j].v));
              }
          }
#endif
          if (c < nitems) {
            const char* wl = lds_w + (size_t)(WS ? cur_kc : (c & 1)) * (WIMG_VECS * 16);
            // keep the bases opaque so the tap addresses are re-derived (one add each) instead of being hoisted
            // out of the persistent loop into 36 VGPRs
            asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(wb));
            run_mma(lds_halo, wl, NoPace{});
            if (cur_kc == p.nchunks - 1) {
                ep_pending = true;
                ep_n = cur_n; ep_ty0 = cur_ty0; ep_tx0 = cur_tx0;
            }
            cur_tile = nxt_tile; cur_kc = nxt_kc; cur_n = nxt_n; cur_ty0 = nxt_ty0; cur_tx0 = nxt_tx0;
            if constexpr (NH == 1) {
                // affine table of the item this half commits in the next tick (loaded by its last issue)
                if (wave == 0) lds_aff[lane] = pf.aff;
            }
          }
            PT_MARK(6)
        }
        if (!(DBG(p) & 64)) __syncthreads();
#ifdef MRISR_PHASE_TIMING
        if (phase >= 0 && (phase & 1) == 0) PT_MARK(5) else PT_MARK(7)
#endif
    }
#ifdef MRISR_PHASE_TIMING
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {   // accumulated over launches (mrisr_debug_phase_reset clears)
        pt_acc[10] = __builtin_amdgcn_s_memrealtime() - pt_r0;     // 100 MHz ticks of the same interval -> shader clock
#pragma unroll
        for (int k = 0; k < 12; ++k) atomicAdd(&g_phase_cycles[threadIdx.x >> 6][k], pt_acc[k]);
        if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[0][11], 1ull);   // launches
    }
#endif
}

#ifndef MRISR_KERNEL_ONLY   // (tuning: a translation unit that instantiates single kernels includes this file with it set)
#ifdef MRISR_PHASE_TIMING
extern "C" int mrisr_debug_phase_reset() {
    static unsigned long long zeros[96];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zeros, sizeof(zeros));
}
extern "C" int mrisr_debug_phase_cycles(unsigned long long* out96) {
    return (int)hipMemcpyFromSymbol(out96, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 96);
}
#endif
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

// All layers of a model in one launch (the optimiser rewrites every master weight each step): blockIdx.y = job.
// Thread = one 16-byte chunk of a packed image (EPC consecutive input channels of one (tap, output channel) row): the index
// arithmetic is paid once per chunk, the fp32 masters are read as EPC consecutive floats (forward operand) or EPC floats
// one output-channel row apart (mirrored input-gradient operand), the chunk is stored with one 16-byte store.
// (one thread per ELEMENT with five integer divisions each ran at 1 TB/s: 62 us per training step.)
struct PackJobDev { const float* w; void* packed; int Cout, Cin, ksize, flip; };
template <typename T>
__global__ void pack_weights_batched_kernel(const PackJobDev* __restrict__ jobs) {
    constexpr int BK = kRowBytes / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);         // elements per 16-B chunk
    const PackJobDev j = jobs[blockIdx.y];
    const int KS = j.ksize, ntaps = KS * KS, Cin = j.Cin, Cout = j.Cout, flip = j.flip & 1;
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;   // logical (output, input) of the image
    const float* __restrict__ w = j.w;
    // source element of (output channel co, tap, input channel k) of the image: forward W[co][tap][k], mirrored W[k][ntaps-1-tap][co]
    auto gather = [&](int co, int tap, int k0, Vec16<T>& v) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int k = k0 + e;
            float x = 0.f;
            if (co < Co && k < Ci) x = !flip ? w[((size_t)co * ntaps + tap) * Cin + k] : w[((size_t)k * ntaps + (ntaps - 1 - tap)) * Cin + co];
            v.set(e, x);
        }
    };
    if (j.flip & MRISR_PACK_RING) {
        // ring layout (conv_ring.hip): [cout block][cin chunk of 16][tap][BN rows][32 B]; the 16-B slot s of row r sits at
        // position s ^ ((r >> 3) & 1), so that a row fragment reads conflict-free and a DMA piece is a linear copy
        const int RBN = conv_ring_bn(TypeTraits<T>::kDtype, Co, Ci, KS);
        if (RBN == 0) return;
        const int rnch = Ci / 16;
        const size_t nchunk = (size_t)(Co / RBN) * rnch * ntaps * RBN * 2;      // 16-B chunks (8 elements of a 16-bit type)
        for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < nchunk; idx += (size_t)gridDim.x * blockDim.x) {
            size_t r = idx;
            const int pos = r & 1; r >>= 1;
            const int row = r % RBN; r /= RBN;
            const int tap = r % ntaps; r /= ntaps;
            const int kc = r % rnch;
            const int cb = r / rnch;
            const int slot = pos ^ ((row >> 3) & 1);
            Vec16<T> v;
            gather(cb * RBN + row, tap, kc * 16 + slot * 8, v);
            store_vec16((T*)j.packed + idx * EPC, v);
        }
        return;
    }
    const int BN = Co >= 64 ? 64 : 32;                          // conv_choose_bn
    const int ncb = (Co + BN - 1) / BN, nchunks = (Ci + BK - 1) / BK;
    const size_t nchunk = (size_t)ncb * nchunks * ntaps * BN * 4;     // four 16-B chunks per 64-B row
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < nchunk; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int pos = r & 3; r >>= 2;            // chunk position inside the row (after the swizzle)
        const int row = r % BN; r /= BN;
        const int tap = r % ntaps; r /= ntaps;
        const int kc = r % nchunks;
        const int cb = r / nchunks;
        const int q = tap * BN + row;
        const int chunk = pos ^ ((q >> 2) & 3);
        Vec16<T> v;
        gather(cb * BN + row, tap, kc * BK + chunk * EPC, v);
        store_vec16((T*)j.packed + idx * EPC, v);
    }
}

extern "C" int mrisr_pack_weights_batched(int dtype, const mrisr_pack_job* jobs_device, int njobs, void* stream) {
    static_assert(sizeof(PackJobDev) == sizeof(mrisr_pack_job), "mrisr_pack_job layout");
    if (!jobs_device || njobs <= 0 || njobs > 65535) MRISR_FAIL(MRISR_E_ARG, "pack_weights_batched: bad job table");
    dim3 grid(256, njobs);     // small jobs leave their surplus blocks immediately; the largest image decides the time
    if (dtype == MRISR_BF16)
        pack_weights_batched_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const PackJobDev*)jobs_device);
    else if (dtype == MRISR_F16)
        pack_weights_batched_kernel<f16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const PackJobDev*)jobs_device);
    else if (dtype == MRISR_F32)
        pack_weights_batched_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const PackJobDev*)jobs_device);
    else
        MRISR_FAIL(MRISR_E_DTYPE, "pack_weights_batched: dtype %d", dtype);
    MRISR_CHECK_LAUNCH("pack_weights_batched");
    return MRISR_OK;
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
    if (transpose_flip & MRISR_PACK_RING) {
        const int flip = transpose_flip & 1;
        if (!mrisr_conv_ring_bn(dtype, flip ? Cin : Cout, flip ? Cout : Cin, ksize))
            MRISR_FAIL(MRISR_E_UNSUPPORTED, "pack_weights: no ring layout for %d -> %d k%d dtype %d", Cin, Cout, ksize, dtype);
        const PackJobDev job{w, packed, Cout, Cin, ksize, transpose_flip};
        PackJobDev* dj = nullptr;      // (stand-alone packing is a test / tool path: the training step uses the batched entry)
        if (hipMalloc(&dj, sizeof(job)) != hipSuccess) MRISR_FAIL(MRISR_E_HIP, "pack_weights: hipMalloc");
        (void)hipMemcpyAsync(dj, &job, sizeof(job), hipMemcpyHostToDevice, (hipStream_t)stream);
        const int rc = mrisr_pack_weights_batched(dtype, (const mrisr_pack_job*)dj, 1, stream);
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(dj);
        return rc;
    }
    transpose_flip &= 1;
    const int Co = transpose_flip ? Cin : Cout, Ci = transpose_flip ? Cout : Cin;
    const int BN = conv_choose_bn(Co), BK = conv_bk(dtype);
    const int ncb = ceil_div(Co, BN), nch = ceil_div(Ci, BK);
    const size_t total = (size_t)ncb * nch * ksize * ksize * BN * BK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == MRISR_BF16)
        pack_weights_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(w, (bf16_t*)packed, Cout, Cin, ksize,
                                                                            transpose_flip, BN, ncb, nch);
    else if (dtype == MRISR_F16)
        pack_weights_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(w, (f16_t*)packed, Cout, Cin, ksize,
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
int num_cus();
int conv_fill_params(const mrisr_conv_desc* d, ConvParams& p, const char* who) {
    if (!d) MRISR_FAIL(MRISR_E_ARG, "%s: null descriptor", who);
    if (!mrisr_dtype_ok(d->dtype)) MRISR_FAIL(MRISR_E_DTYPE, "%s: dtype %d", who, d->dtype);
    if (d->ksize != 1 && d->ksize != 3) MRISR_FAIL(MRISR_E_UNSUPPORTED, "%s: ksize %d", who, d->ksize);
    if (d->nsrc < 1 || d->nsrc > 2) MRISR_FAIL(MRISR_E_ARG, "%s: nsrc %d", who, d->nsrc);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0)
        MRISR_FAIL(MRISR_E_SHAPE, "%s: bad dims N%d H%d W%d Cin%d Cout%d", who, d->N, d->H, d->W, d->Cin, d->Cout);
    const int vec = mrisr_vec(d->dtype);
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
        // ranges of the 24-bit / 32-bit offset arithmetic of the plain loader
        const size_t esz = d->dtype == MRISR_F32 ? 4 : 2;
        if ((size_t)a.H * a.W >= (1u << 24) || a.W >= 32768 || (size_t)a.C * esz >= (1u << 24) || (size_t)a.H * a.W * a.C * esz >= (1ull << 32))
            MRISR_FAIL(MRISR_E_SHAPE, "%s: src%d image %dx%dx%d exceeds the loader's offset range", who, s, a.H, a.W, a.C);
        p.src[s] = SrcDev{a.ptr, a.scale, a.shift, a.C, a.H, a.W, a.mode, a.off_y, a.off_x, (unsigned)((size_t)a.H * a.W * a.C * esz)};
        csum += a.C;
    }
    if (d->combine == MRISR_COMBINE_BLEND) {
        if (d->nsrc != 2 || d->src[0].C != d->src[1].C || !d->blend_alpha) MRISR_FAIL(MRISR_E_ARG, "%s: blend needs 2 equal sources + alpha", who);
        csum = d->src[0].C;
    }
    if (csum != d->Cin) MRISR_FAIL(MRISR_E_SHAPE, "%s: sources carry %d channels, Cin=%d", who, csum, d->Cin);
    const int BK = conv_bk(d->dtype), BN = conv_choose_bn(d->Cout);
    p.blend_alpha = d->blend_alpha; p.wpacked = d->wpacked; p.bias = d->bias; p.out = d->out; p.stats = d->stats;
    p.mask = d->relu_mask;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.nchunks = ceil_div(d->Cin, BK); p.CinP = p.nchunks * BK;
    p.ncb = ceil_div(d->Cout, BN); p.CoutP = p.ncb * BN;
    p.nsrc = d->nsrc; p.combine = d->combine; p.out_mode = d->out_mode; p.groups = d->stats ? d->groups : 0;
    p.relu_out = d->relu_out;
    p.cus = d->cu_limit > 0 && d->cu_limit < num_cus() ? d->cu_limit : num_cus();
#ifdef MRISR_TUNING
    { static const int dbg_env = [] { const char* e = getenv("MRISR_DEBUG"); return e ? atoi(e) : 0; }(); p.dbg = dbg_env; }
#endif
    conv_choose_tile(d->W, p.th, p.tw_log2);
    p.tiles_x = ceil_div(d->W, 1 << p.tw_log2); p.tiles_y = ceil_div(d->H, p.th);
    return MRISR_OK;
}

// CU count of the current device (read once; one process drives one GPU - mrisr.h threading contract)
int num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

int launch_gn_stats(int dtype, const void* x, double* stats, int N, int HW, int C, int groups, hipStream_t s);
// conv_ring.hip: the deep-ring raw-source kernel
bool conv_ring_eligible(const mrisr_conv_desc* d, const ConvParams& p);
bool conv_wgrad_rows_ok(const mrisr_conv_desc* d);
int launch_conv_ring(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s);
// conv_pc.hip: producer / consumer waves (GroupNorm or stored sources, 128-channel output blocks)
bool conv_pc_eligible(const mrisr_conv_desc* d, const ConvParams& p);
int conv_pc_kind(const mrisr_conv_desc* d, const ConvParams& p);
int launch_conv_pc(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s);
// conv1x1.hip: 1x1 convolutions as a plain GEMM
bool conv1x1_gemm_eligible(const mrisr_conv_desc* d, const ConvParams& p);
int launch_conv1x1_gemm(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s);

// every source stored as-is and a plain (single / concat) loader: the halo tile can go global -> LDS by LDS-DMA
static bool conv_dma_halo(const ConvParams& p, int spatial) {
#ifdef MRISR_NO_DMA_HALO
    return false;
#endif
    if (spatial != MRISR_SP_NONE || p.combine == MRISR_COMBINE_BLEND) return false;
    for (int s = 0; s < p.nsrc; ++s)
        if (p.src[s].mode != MRISR_SRC_RAW) return false;
    return true;
}

template <typename T, int BN, int SPATIAL, int KS, bool DMA>
static int launch_conv_v(ConvParams& p, hipStream_t s) {
    const size_t wimg = (size_t)KS * KS * BN * kRowBytes;
    // weights-stationary when every cin chunk fits next to the halo tiles
    p.ws = conv_weights_stationary(p.nchunks, wimg, DMA) ? 1 : 0;
    const size_t lds = conv_halo_total(DMA) + (p.ws ? p.nchunks : 2) * wimg + (BN + 128) * sizeof(float) + conv_stage_bytes();
    p.ntiles = p.N * p.tiles_y * p.tiles_x;
    int per_cb = p.cus / p.ncb;                         // persistent workgroups per cout block, one per CU
    if (per_cb < 1) per_cb = 1;
    const int pairs = ceil_div(p.ntiles, 2);
    if (per_cb > pairs) per_cb = pairs;
    p.tiles_per_block = ceil_div(p.ntiles, per_cb);
    per_cb = ceil_div(p.ntiles, p.tiles_per_block);
    const int grid = per_cb * p.ncb;
    // tiny channel counts (a GroupNorm group narrower than 4 channels): statistics by a separate pass
    double* stats = p.stats;
    const bool stats_sep = stats && ((p.Cout / p.groups) & 3);
    if (stats_sep) p.stats = nullptr;
    constexpr bool kPS = (SPATIAL == MRISR_SP_NONE && KS == 3);   // pixel-shuffle / mask epilogues: plain 3x3 convs only
    static std::once_flag attr_once;   // per instantiation
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, true, 0, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, false, 0, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (kPS) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, true, 1, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, false, 1, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, true, kEpiMask, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BN, SPATIAL, KS, false, kEpiMask, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    });
    if (p.out_mode == MRISR_OUT_PIXEL_SHUFFLE2) {
        if constexpr (kPS) {
            if (p.ws) hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, true, 1, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, false, 1, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
        } else {
            MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: pixel-shuffle epilogue needs a 3x3 conv with a plain source");
        }
    } else if (p.mask) {
        if constexpr (kPS) {
            if (p.ws) hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, true, kEpiMask, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, false, kEpiMask, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
        } else {
            MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: relu_mask epilogue needs a 3x3 conv with a plain source");
        }
    } else if (p.ws) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, true, 0, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, SPATIAL, KS, false, 0, DMA>), dim3(grid), dim3(kFwdThreads), lds, s, p);
    }
    MRISR_CHECK_LAUNCH("conv_forward");
    if (stats_sep) {
        const bool ps = p.out_mode == MRISR_OUT_PIXEL_SHUFFLE2;
        return launch_gn_stats(TypeTraits<T>::kDtype, p.out, stats, p.N, (ps ? 4 : 1) * p.H * p.W, ps ? p.Cout / 4 : p.Cout, p.groups, s);
    }
    return MRISR_OK;
}

template <typename T, int BN, int SPATIAL, int KS>
static int launch_conv(ConvParams& p, hipStream_t s) {
    if constexpr (SPATIAL == MRISR_SP_NONE) {
        if (conv_dma_halo(p, SPATIAL)) return launch_conv_v<T, BN, SPATIAL, KS, true>(p, s);
    }
    return launch_conv_v<T, BN, SPATIAL, KS, false>(p, s);
}

template <typename T, int BN>
static int dispatch_conv_sp(ConvParams& p, int spatial, int ks, hipStream_t s) {
    if (p.combine == MRISR_COMBINE_BLEND) {
        if (ks != 3) MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: blend needs a 3x3 conv");
        return launch_conv<T, BN, kLoaderBlend, 3>(p, s);
    }
    if (ks == 3) {
        if (spatial == MRISR_SP_NONE) return launch_conv<T, BN, MRISR_SP_NONE, 3>(p, s);
        if (spatial == MRISR_SP_POOL2) return launch_conv<T, BN, MRISR_SP_POOL2, 3>(p, s);
        return launch_conv<T, BN, MRISR_SP_UP2, 3>(p, s);
    }
    if (spatial == MRISR_SP_NONE) return launch_conv<T, BN, MRISR_SP_NONE, 1>(p, s);
    if (spatial == MRISR_SP_UP2) return launch_conv<T, BN, MRISR_SP_UP2, 1>(p, s);
    MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: 1x1 conv with pooled source");
}

// Name of the template instantiation the dispatcher picks for a descriptor (the grouping rocprofv3 reports).
extern "C" int mrisr_conv_variant(const mrisr_conv_desc* d, int wgrad, char* out, size_t n) {
    ConvParams p;
    int rc = conv_fill_params(d, p, "conv_variant");
    if (rc) return rc;
    if (!out || n < 8) MRISR_FAIL(MRISR_E_ARG, "conv_variant: bad buffer");
    const char* t = d->dtype == MRISR_BF16 ? "bf16" : (d->dtype == MRISR_F16 ? "f16" : "f32");
    const int loader = d->combine == MRISR_COMBINE_BLEND ? 3 : d->src[0].spatial;
    if (wgrad && conv_wgrad_rows_ok(d)) {
        bool raw = true;
        for (int i = 0; i < d->nsrc; ++i) raw = raw && d->src[i].mode == MRISR_SRC_RAW;
        snprintf(out, n, "conv_wgrad_rows_kernel<%s,%d,%d,%d>", t, d->Cout % 64 ? 1 : 2, d->Cin % 64 ? 1 : 2, raw ? 1 : 0);
    } else if (wgrad) {
        snprintf(out, n, "conv_wgrad_kernel<%s,%d,%d,%d>", t, loader, d->ksize,
                 conv_wgrad_fast(d->dtype, loader, d->ksize, p.tw_log2, d->Cout, d->Cin));
    } else if (conv_ring_eligible(d, p)) {
        snprintf(out, n, "conv_ring_kernel<%s,2,4>", t);
    } else if (conv_pc_eligible(d, p)) {
        // (",64": the 64-channel blocks on tall tiles)
        snprintf(out, n, "conv_pc_kernel<%s,%d%s>", t, d->src[0].mode == MRISR_SRC_NORM ? 1 : 0, conv_pc_kind(d, p) == 2 ? ",64" : "");
    } else if (conv1x1_gemm_eligible(d, p)) {
        snprintf(out, n, "conv1x1_gemm_kernel<%s,%d>", t, d->src[0].mode == MRISR_SRC_NORM ? 1 : 0);
    } else {
        const int BN = conv_choose_bn(d->Cout);
        const size_t wimg = (size_t)d->ksize * d->ksize * BN * kRowBytes;
        const bool dma = conv_dma_halo(p, d->src[0].spatial);
        const int ws = conv_weights_stationary(p.nchunks, wimg, dma) ? 1 : 0;
        snprintf(out, n, "conv_igemm_kernel<%s,%d,%d,%d,%d,%d,%d>", t, BN, loader, d->ksize, ws,
                 d->out_mode == MRISR_OUT_PIXEL_SHUFFLE2 ? 1 : (d->relu_mask ? kEpiMask : 0), dma ? 1 : 0);
    }
    return MRISR_OK;
}

extern "C" int mrisr_conv_forward(const mrisr_conv_desc* d, void* stream) {
    ConvParams p;
    int rc = conv_fill_params(d, p, "conv_forward");
    if (rc) return rc;
    if (!d->wpacked || !d->out) MRISR_FAIL(MRISR_E_ARG, "conv_forward: null weights/out");
    if (d->out_mode == MRISR_OUT_PIXEL_SHUFFLE2 && (d->Cout % 16)) MRISR_FAIL(MRISR_E_SHAPE, "conv_forward: pixel shuffle needs Cout%%16==0");
    if (d->relu_mask && (d->out_mode != MRISR_OUT_PLAIN || d->Cout % (mrisr_vec(d->dtype))))
        MRISR_FAIL(MRISR_E_UNSUPPORTED, "conv_forward: relu_mask needs a plain output with Cout a multiple of the 16-byte vector");
    if (d->stats && (d->groups <= 0 || d->Cout % d->groups)) MRISR_FAIL(MRISR_E_SHAPE, "conv_forward: Cout %d not divisible by groups %d", d->Cout, d->groups);
    if (conv_ring_eligible(d, p)) return launch_conv_ring(d, p, (hipStream_t)stream);
    if (conv_pc_eligible(d, p)) return launch_conv_pc(d, p, (hipStream_t)stream);
    if (conv1x1_gemm_eligible(d, p)) return launch_conv1x1_gemm(d, p, (hipStream_t)stream);
    // square 16 x 16 output tiles (324-pixel halo, 18-pixel rows) instead of 8 x 32 (340, 34): the kernels are bound by the
    // operand bytes they stage (profiles/NOTES.md R2-13/14).  32 -> 32 at 512^2: 167 -> 130 us, 64 -> 32 / 32 -> 64: 3-5 %, wide
    // layers +-2 % each, the training step -0.8 % with every 3x3 layer on 16 x 16 (A/B on one box).  The weight-gradient
    // kernel keeps 8 x 32: its fast paths are built on it.
#ifndef MRISR_NO_TILE16
    if (d->ksize == 3 && d->W > 16 && d->H >= 16) {
        p.th = 16; p.tw_log2 = 4;
        p.tiles_x = ceil_div(d->W, 16); p.tiles_y = ceil_div(d->H, 16);
    }
#endif
    const int sp = d->src[0].spatial;
    hipStream_t s = (hipStream_t)stream;
    const int BN = conv_choose_bn(d->Cout);
    if (d->dtype == MRISR_BF16) return BN == 64 ? dispatch_conv_sp<bf16_t, 64>(p, sp, d->ksize, s) : dispatch_conv_sp<bf16_t, 32>(p, sp, d->ksize, s);
    if (d->dtype == MRISR_F16) return BN == 64 ? dispatch_conv_sp<f16_t, 64>(p, sp, d->ksize, s) : dispatch_conv_sp<f16_t, 32>(p, sp, d->ksize, s);
    return BN == 64 ? dispatch_conv_sp<float, 64>(p, sp, d->ksize, s) : dispatch_conv_sp<float, 32>(p, sp, d->ksize, s);
}
#endif  // MRISR_KERNEL_ONLY
