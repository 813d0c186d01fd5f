// 3x3 convolution on MFMA with PRODUCER / CONSUMER waves (gfx950): forward of the GroupNorm-sourced layers and the layers /
// input gradients with stored sources whose output has a multiple of 128 channels.
//
// Replaces the same aten conv2d / convolution_backward(input) calls as conv_fwd.hip
// (/root/reference/models/unet_model.py:29,34,152,168 and autograd's dgrad of them).
//
// Why a third forward kernel (profiles/NOTES.md R3-2 .. R3-5): in conv_igemm_kernel all eight waves of a workgroup load,
// GroupNorm+LeakyReLU-transform and LDS-store the next chunk BETWEEN their MFMAs - ~36 vector instructions per 16-byte
// vector at ~7 cycles each beside the other wave's MFMAs - and every wave waits at two barriers per chunk.  The
// weight-gradient kernel conv_wgrad_rows.hip showed what the split into roles buys: an MFMA-only wave per SIMD runs at
// 38 cycles per MFMA (84 % of the pipe) as long as the staging waves keep up.  The same split here:
//   * waves 0-3 (one per SIMD, the older ones: they win the issue arbitration) only read fragments and issue MFMAs: a wave
//     owns 64 pixels x 128 output channels (128 accumulator registers, 6 fragment reads per 8 MFMAs), runs its tile's
//     epilogue (GroupNorm statistics, 16-byte stores) itself;
//   * waves 4-7 only stage: buffer loads of the item after next into registers (halo tile 10 x 34 pixels x 16 channels and
//     the item's 36-KiB weight image), GroupNorm + LeakyReLU on the halo vectors that have arrived, LDS stores.
// Work item = 8 x 32 pixels x 128 output channels x 16 input channels (32-byte LDS rows, the ring kernel's layouts: the
// weight image is the ring-packed one, [cout block][cin chunk][tap][128 rows][32 B], slots pre-swizzled).  Three LDS
// buffers + two register sets: an item's loads are issued two items (~3 us) before its commit.  The roles hand items over
// through two LDS counters, not workgroup barriers (see pc_signal / pc_wait_ge).
#include <mutex>
#include <type_traits>

#include "conv_common.h"

// Ablation switches of tuning builds (tools/build_src_variant.sh; results invalid by construction, timing only):
// 1 no MFMA, 2 no GroupNorm / LeakyReLU transform, 4 no epilogue.  0 in the product library.
#ifndef MRISR_PC_DBG
#define MRISR_PC_DBG 0
#endif
// Phase profile (tuning builds only, -DMRISR_PC_PT; tools/conv_bench.py prints it): slot 6 = work, 5 = barrier wait,
// 4 = epilogue, 10 = 100 MHz ticks of the loop.
#ifdef MRISR_PC_PT
__device__ unsigned long long g_pc_cycles[8][12];
#define PPT_DECL unsigned long long pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime(); const unsigned long long pt_r0 = __builtin_amdgcn_s_memrealtime();
#define PPT_MARK(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_now = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_now - pt_t; pt_t = pt_now; __builtin_amdgcn_sched_barrier(0); }
#define PPT_DUMP() if (blockIdx.x == gridDim.x / 2 && lane == 0) { pt_acc[10] = __builtin_amdgcn_s_memrealtime() - pt_r0; for (int k = 0; k < 12; ++k) atomicAdd(&g_pc_cycles[wave][k], pt_acc[k]); if (threadIdx.x == 0) atomicAdd(&g_pc_cycles[0][11], 1ull); }
#else
#define PPT_DECL
#define PPT_MARK(k)
#define PPT_DUMP()
#endif
#ifndef MRISR_PC_AD
#define MRISR_PC_AD 3        // weight-fragment prefetch distance (fragments)
#endif

typedef __attribute__((ext_vector_type(2))) float pc_f32x2;      // pairs for v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

constexpr int kPcThreads = 512;                  // 4 MFMA waves + 4 staging waves
// staging waves (a parameter of the layouts below; 8 for the blend variant was tried: 376 instead of 280 us - more staging
// waves do not make its 32-channel items cheaper)
constexpr int pc_np(bool blend) { return blend ? 4 : 4; }
constexpr int pc_threads(bool blend) { return 256 + 64 * pc_np(blend); }
constexpr int kPcBN = 128;                       // output channels per workgroup (NI = 4; the narrow variant NI = 1 owns 32)
// pixel tile = (4 mi) x 32: an MFMA wave owns mi rows of 32 pixels (mi = 2: 8 x 32 tile, 10 x 34 halo; mi = 4: the tall tile of
// the 64-channel variant, 18 x 34 halo)
constexpr int kPcHaloW = 34;
constexpr int pc_halo_h(int mi) { return 4 * mi + 2; }
constexpr int pc_halo_rows(int mi) { return kPcHaloW * pc_halo_h(mi); }
// 16-byte vectors per staging thread: 340 halo rows x 2 over 64 np threads (the last slot stores unpredicated: whole slots of room)
constexpr int pc_hslots(int np, int mi = 2) { return (2 * pc_halo_rows(mi) + 64 * np - 1) / (64 * np); }
constexpr int pc_hbytes(int np, int mi = 2) { return pc_hslots(np, mi) * 64 * np * 16; }
constexpr int pc_wbytes(int ni) { return 9 * 32 * ni * 32; }                     // weight image of one (cout block, cin chunk)
constexpr int pc_wslots(int ni, int np) { return (pc_wbytes(ni) / 16 + 64 * np - 1) / (64 * np); }
constexpr int pc_buf(int ni, int np, int mi = 2) { return pc_hbytes(np, mi) + pc_wslots(ni, np) * 64 * np * 16; }
constexpr int kPcStages = 3;                     // LDS item buffers (3 x 48 KiB at NI = 4)
constexpr int pc_lds(int ni, int np, int mi = 2) { return kPcStages * pc_buf(ni, np, mi) + kPcBN * 4 + 64; }   // + bias table + the item counters
constexpr unsigned pc_slot_ones(int slots) { return slots <= 0 ? 0u : (pc_slot_ones(slots - 1) | (1u << (4 * (slots - 1)))); }

// Item counters in LDS instead of workgroup barriers: the staging waves may run up to two items ahead of the MFMA waves
// (they keep staging while a tile's epilogue runs) and nobody waits for the slowest wave of the OTHER role at every item.
//   ready[w]: items stored by staging wave w, bumped behind its LDS stores of an item -> item c can be read when all four
//             are >= c + 1;   done[w]: items read by MFMA wave w, bumped behind its last fragment read of an item -> the
//             buffer of item c can be rewritten when all four are >= c + 1
// (one counter per WAVE: the waves of a role are not synchronised with each other, a sum would let a fast wave's count stand
// in for a slow one's).  LDS instructions of one wave execute in issue order, so the increment lands behind the stores / reads
// it publishes; the asm statements are compiler barriers for memory operations.  A wave that waits unreasonably long stops
// waiting for good (results are then wrong, but the kernel ends: a mis-count must never hang the GPU).
__device__ __forceinline__ void pc_signal(unsigned lds_addr) {
    asm volatile("ds_add_u32 %0, %1" ::"v"(lds_addr), "v"(1u) : "memory");
}
// (the polls are volatile LDS loads, not asm: hipcc's own `s_waitcnt lgkmcnt` bookkeeping must see every LDS operation)
__device__ __forceinline__ int pc_min4(const u32x4 v) { return min(min((int)v[0], (int)v[1]), min((int)v[2], (int)v[3])); }
// Four counters.  RELAXED WORKGROUP-scope atomic loads through an explicit LDS (address space 3) pointer: plain `ds_read_b32`s
// that hipcc tracks on lgkmcnt and waits for at their first use.  (A `volatile` load through the generic pointer compiled to
// `flat_load_dwordx4 ... sc0 sc1` + `s_waitcnt vmcnt(0)`: every peek of the staging waves waited for ALL their outstanding
// global loads, and - a FLAT instruction inside the polling loop - made the wait-count pass emit `vmcnt(0)` behind the loop.)
template <int NV>
__device__ __forceinline__ u32x4 pc_peek(const unsigned* cnt) {
    static_assert(NV == 1, "four counters");
    typedef const __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* c3 = (lds_u32*)cnt;
    u32x4 m;
    m[0] = __hip_atomic_load(c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    m[1] = __hip_atomic_load(c3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    m[2] = __hip_atomic_load(c3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    m[3] = __hip_atomic_load(c3 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return m;
}
// `seen`: a peek taken earlier (the answer is usually already there: no LDS round trip on the critical path)
template <int NV>
__device__ __forceinline__ void pc_wait_ge(const unsigned* cnt, u32x4 seen, int target, bool& broken) {
    if (!broken && target > 0) {
        int spin = 0;
        while (__builtin_amdgcn_readfirstlane(pc_min4(seen)) < target) {
            if (++spin > (1 << 21)) { broken = true; break; }
            __builtin_amdgcn_s_sleep(2);
            seen = pc_peek<NV>(cnt);
        }
    }
    asm volatile("" ::: "memory");
}

// NI = 32-channel fragments per MFMA wave (4: 128 output channels per workgroup; 1: the 32-channel layer of the 2x head);
// BLEND: the input is sigmoid(alpha) * act(src0) + (1 - sigmoid(alpha)) * act(src1) (unet_model.py:206-207), formed by the
// staging waves - the blended tensor never goes to HBM (eval forward of final_conv.0)
// MI = 32-pixel rows per MFMA wave.  (NI, MI) = (4, 2): 8 x 32 pixels x 128 channels; (2, 4): 16 x 32 pixels x 64 channels - the
// 64-channel layers: the same 8 accumulators and 6 fragment reads per 8 MFMAs, 38 instead of 47 KiB staged per item
template <typename T, bool NORM, bool STATS, int NI = 4, bool BLEND = false, int MI = 2>
__global__ __launch_bounds__(pc_threads(BLEND), 2) void conv_pc_kernel(const ConvParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams p = pin_params(p_in);
    typedef typename Frag16<T>::type frag_t;
    constexpr int BN = 32 * NI, VEC = 8;
    constexpr int NP = pc_np(BLEND), PT = 64 * NP;                  // staging waves / threads
    constexpr int TR = 4 * MI, kPcHaloH = pc_halo_h(MI), kPcHaloRows = pc_halo_rows(MI);      // tile rows, halo rows
    constexpr int kPcHaloSlots = pc_hslots(NP, MI), kPcHaloBytes = pc_hbytes(NP, MI);
    constexpr int kPcWBytes = pc_wbytes(NI), kPcWSlots = pc_wslots(NI, NP), kPcBuf = pc_buf(NI, NP, MI);
    static_assert(kPcHaloSlots <= 8 && kPcHaloRows + PT / 2 < 1000, "edge flags: 4 bits per slot; row / 34 by multiplication");
    static_assert(!BLEND || NORM, "the blend is of two activated sources");
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool consumer = wave < 4;
    float* lds_bias = reinterpret_cast<float*>(smem + kPcStages * kPcBuf);
    unsigned* lds_cnt = reinterpret_cast<unsigned*>(smem + kPcStages * kPcBuf + kPcBN * 4);
    const unsigned ready_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + kPcStages * kPcBuf + kPcBN * 4);
    const unsigned done_a = ready_a + 32;
    const unsigned* ready_p = lds_cnt;
    const unsigned* done_p = lds_cnt + 8;
    bool broken = false;

    // XCD-aware workgroup order (conv_fwd.hip): the cout blocks of a tile range, then the neighbouring ranges, share an L2
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cb = bid % p.ncb;
    const int bt0 = (bid / p.ncb) * p.tiles_per_block;
    const int bt1 = min(bt0 + p.tiles_per_block, p.ntiles);
    const int total = max(bt1 - bt0, 0) * p.nchunks;     // items of this workgroup
    const int bn0 = cb * BN;
    if (total <= 0) return;
    if (t < BN) lds_bias[t] = p.bias ? gload<float>(p.bias + bn0 + t) : 0.f;
    const float blend_a = BLEND ? 1.f / (1.f + __expf(-gload<float>(p.blend_alpha))) : 1.f;
    if (t < 16) lds_cnt[t] = 0u;
    __syncthreads();                    // the only workgroup barrier: bias table and counters are set up

    auto decode = [&](int tile, int& n, int& ty0, int& tx0) {
        const int tx = tile % p.tiles_x;
        const int r = tile / p.tiles_x;
        n = r / p.tiles_y;
        ty0 = (r - n * p.tiles_y) * TR;
        tx0 = tx * 32;
    };

    // the next tile of the linear order without divisions (decode() costs ~100 vector instructions on uniform values; with two
    // items per tile - the 32-channel layer - that was a third of the staging waves' time)
    auto next_tile = [&](int& n, int& ty0, int& tx0) {
        tx0 += 32;
        if (tx0 == p.tiles_x * 32) {
            tx0 = 0;
            ty0 += TR;
            if (ty0 == p.tiles_y * TR) { ty0 = 0; ++n; }
        }
    };

    // ------------------------------------------------------------------ schedule
    //   consumers:  wait ready(k) -> MFMA(item k) from buffer k % 3 -> signal done (+ the tile's epilogue behind its last item)
    //   producers:  wait done(k - 3) -> commit(item k) into buffer k % 3 -> signal ready -> issue loads(item k+2) into the
    //               registers just freed
    // Two separate loops: neither role's registers are live in the other's code.
    if (!consumer) {
        // ------------------------------------------------------------------ producer (waves 4-7): 256 staging threads
        const int pt = t - 256;
        // halo vector v = pt + 256 j lies at LDS byte 16 v: row v >> 1 = (pt >> 1) + 128 j, physical slot pt & 1, which holds
        // the LOGICAL 8-channel half (pt & 1) ^ ((row >> 3) & 1) = (pt & 1) ^ ((pt >> 4) & 1)
        const int hrow0 = pt >> 1;
        const int lslot = (pt & 1) ^ ((pt >> 4) & 1);
        struct StageSet {
            Vec16<T> h[kPcHaloSlots];
            Vec16<T> h2[BLEND ? kPcHaloSlots : 1];  // second source of the blend
            u32x4 w[kPcWSlots];
            pc_f32x2 sc[VEC / 2], sh[VEC / 2];     // GroupNorm affine of this thread's 8 channels, as pairs (packed fp32 math)
            pc_f32x2 sc2[BLEND ? VEC / 2 : 1], sh2[BLEND ? VEC / 2 : 1];
            int ty0, tx0, w1;
        };
        StageSet S0, S1;
        // per-source geometry (concat: a 16-channel chunk lies in ONE source - host-checked src0.C % 16 == 0)
        const int C0 = p.src[0].C;
        const bool two = p.nsrc > 1;
        const __amdgpu_buffer_rsrc_t x_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.src[0].ptr), 0, (int)((unsigned)p.N * p.src[0].img_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(two ? p.src[1].ptr : p.src[0].ptr), 0, (int)((unsigned)p.N * (two ? p.src[1].img_bytes : p.src[0].img_bytes)), 0x00020000);
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.wpacked), 0, (int)((unsigned)p.ncb * p.nchunks * kPcWBytes), 0x00020000);
        // GroupNorm scale / shift [N][C] floats of each source: buffer loads with a scalar offset (no per-load address arithmetic)
        const __amdgpu_buffer_rsrc_t sc_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.src[0].scale), 0, NORM ? (int)((unsigned)p.N * p.src[0].C * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t sh_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.src[0].shift), 0, NORM ? (int)((unsigned)p.N * p.src[0].C * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t sc_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(two ? p.src[1].scale : p.src[0].scale), 0, NORM ? (int)((unsigned)p.N * (two ? p.src[1].C : p.src[0].C) * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t sh_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(two ? p.src[1].shift : p.src[0].shift), 0, NORM ? (int)((unsigned)p.N * (two ? p.src[1].C : p.src[0].C) * 4u) : 0, 0x00020000);
        unsigned rel0[kPcHaloSlots], rel1[kPcHaloSlots];
        // edge flags of this thread's slots, 4 bits each: halo row 0 / row 9 / column 0 / column 33 (rows >= 340: none)
        unsigned fflags = 0;
#pragma unroll
        for (int j = 0; j < kPcHaloSlots; ++j) {
            const int row = hrow0 + (PT / 2) * j;
            const int hy = (row * 1928) >> 16, hx = row - hy * kPcHaloW;        // row / 34 for row < 1000
            rel0[j] = (unsigned)((hy * p.src[0].W + hx) * p.src[0].C + lslot * VEC) * 2u;
            rel1[j] = two ? (unsigned)((hy * p.src[1].W + hx) * p.src[1].C + lslot * VEC) * 2u : rel0[j];
            if (row < kPcHaloRows)
                fflags |= (unsigned)((hy == 0 ? 1 : 0) | (hy == kPcHaloH - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == kPcHaloW - 1 ? 8 : 0)) << (4 * j);
        }
        const int aff_v = lslot * VEC * 4;         // byte offset of this thread's 8 channels inside a chunk's 16
        const unsigned wv = (unsigned)pt * 16u;

        // Loads of item (tile at n, ty0, tx0; chunk kc) into S.  STRAIGHT-LINE: no branch around a load (conv_wgrad_rows.hip:
        // with branches hipcc gives up counting and waits for the load it has just issued); the source of a chunk is picked
        // with scalar selects.  Halo offsets outside the tensor return zeros (hardware range check); halo pixels beside the
        // plane alias into neighbouring rows and are overwritten with zeros at commit time (edge tiles only).
        auto issue = [&](StageSet& S, int n, int ty0, int tx0, int kc) {
            const bool w1 = !BLEND && two && kc * 16 >= C0;
            const int Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W, Cs = w1 ? p.src[1].C : C0;
            const int offy = w1 ? p.src[1].off_y : p.src[0].off_y, offx = w1 ? p.src[1].off_x : p.src[0].off_x;
            const int cs = kc * 16 - (w1 ? C0 : 0);
            const unsigned tile_off = (unsigned)(((n * Hs + ty0 - 1 - offy) * Ws + tx0 - 1 - offx) * Cs + cs) * 2u;
            const __amdgpu_buffer_rsrc_t rs = w1 ? x_rsrc1 : x_rsrc0;
#pragma unroll
            for (int j = 0; j < kPcHaloSlots; ++j) {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tile_off + (w1 ? rel1[j] : rel0[j])), 0, 0);
                S.h[j].v = __builtin_bit_cast(decltype(S.h[j].v), r);
                if (BLEND) {        // (host-checked: the two sources have the same geometry)
                    const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc1, (int)(tile_off + rel0[j]), 0, 0);
                    S.h2[j].v = __builtin_bit_cast(decltype(S.h2[j].v), r2);
                }
            }
            const int woff = (cb * p.nchunks + kc) * kPcWBytes;
#pragma unroll
            for (int j = 0; j < kPcWSlots; ++j)
                S.w[j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(wv + 16u * PT * j), woff, 0);
            if (NORM) {
                const int aoff = (n * Cs + cs) * 4;
                const __amdgpu_buffer_rsrc_t scr = w1 ? sc_rsrc1 : sc_rsrc0, shr = w1 ? sh_rsrc1 : sh_rsrc0;
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(scr, aff_v + e * 4, aoff, 0));
                    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(shr, aff_v + e * 4, aoff, 0));
                    S.sc[e / 2] = pc_f32x2{a[0], a[1]}; S.sc[e / 2 + 1] = pc_f32x2{a[2], a[3]};
                    S.sh[e / 2] = pc_f32x2{b[0], b[1]}; S.sh[e / 2 + 1] = pc_f32x2{b[2], b[3]};
                    if (BLEND) {
                        const f32x4 a2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc1, aff_v + e * 4, aoff, 0));
                        const f32x4 b2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sh_rsrc1, aff_v + e * 4, aoff, 0));
                        S.sc2[e / 2] = pc_f32x2{a2[0], a2[1]}; S.sc2[e / 2 + 1] = pc_f32x2{a2[2], a2[3]};
                        S.sh2[e / 2] = pc_f32x2{b2[0], b2[1]}; S.sh2[e / 2 + 1] = pc_f32x2{b2[2], b2[3]};
                    }
                }
            }
            S.ty0 = ty0; S.tx0 = tx0; S.w1 = w1 ? 1 : 0;
        };
        PPT_DECL
        // Stores the item held by S into `buf` (halo transformed), then refills S with item (n, ty0, tx0, kc)
        auto commit_issue = [&](StageSet& S, char* buf, int n, int ty0, int tx0, int kc) {
            char* lds_h = buf;
            char* lds_w = buf + kPcHaloBytes;
            int q = pt;
            asm volatile("" : "+v"(q));
            const int pty0 = S.ty0, ptx0 = S.tx0, pw1 = S.w1;
            // Edge tiles: which of this thread's halo slots lie outside the source (conv padding; their loads aliased into
            // neighbouring rows) - 4 bits per slot, applied as a SELECT in front of the one LDS store of the slot.  (A second,
            // exec-masked store of zeros behind the main one was not safe with counter hand-over: an MFMA wave that reacted
            // within ~100 cycles of the ready flag read the aliased value at plane-edge columns - sparse stores of a wave
            // are not guaranteed to be visible to OTHER waves in issue order; profiles/NOTES.md R3-11.)
            unsigned hit = 0;
            {
                const bool w1 = pw1 != 0;
                const int Hs = w1 ? p.src[1].H : p.src[0].H, Ws = w1 ? p.src[1].W : p.src[0].W;
                const int ys0 = pty0 - 1 - (w1 ? p.src[1].off_y : p.src[0].off_y), xs0 = ptx0 - 1 - (w1 ? p.src[1].off_x : p.src[0].off_x);
                const bool interior = ys0 >= 0 && xs0 >= 0 && ys0 + kPcHaloH <= Hs && xs0 + kPcHaloW <= Ws;
                if (!interior) {
                    // at most the outermost halo ring lies outside (every layer of the network: the source covers the plane):
                    // the tile's four edge bits against the slots' precomputed ones
                    const bool ring_only = ys0 >= -1 && xs0 >= -1 && ys0 + kPcHaloH <= Hs + 1 && xs0 + kPcHaloW <= Ws + 1;
                    if (ring_only) {
                        const unsigned tm = (ys0 < 0 ? 1u : 0u) | (ys0 + kPcHaloH > Hs ? 2u : 0u) | (xs0 < 0 ? 4u : 0u) | (xs0 + kPcHaloW > Ws ? 8u : 0u);
                        hit = fflags & (tm * pc_slot_ones(kPcHaloSlots));
                    } else {
#pragma unroll
                        for (int j = 0; j < kPcHaloSlots; ++j) {
                            const int row = (q >> 1) + (PT / 2) * j;
                            const int hy = (row * 1928) >> 16, hx = row - hy * kPcHaloW;
                            const unsigned y = ys0 + hy, x = xs0 + hx;
                            if (row < kPcHaloRows && !(y < (unsigned)Hs && x < (unsigned)Ws)) hit |= 0xfu << (4 * j);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kPcHaloSlots; ++j) {
                Vec16<T> v = S.h[j];
                if (BLEND) {
                    // sigmoid(alpha) * act(src0) + (1 - sigmoid(alpha)) * act(src1) in fp32, rounded once (as the classic blend
                    // loader and norm_blend_kernel)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float y0 = fmaf(v.get(e), S.sc[e / 2][e & 1], S.sh[e / 2][e & 1]);
                        const float y1 = fmaf(S.h2[j].get(e), S.sc2[e / 2][e & 1], S.sh2[e / 2][e & 1]);
                        const float a0 = fmaxf(y0, LRELU_SLOPE * y0), a1 = fmaxf(y1, LRELU_SLOPE * y1);
                        v.set(e, blend_a * a0 + (1.f - blend_a) * a1);
                    }
                } else if (NORM && !(MRISR_PC_DBG & 2)) {
#pragma unroll
                    for (int e = 0; e < VEC; e += 2) {
                        // scalar fp32 math on purpose: v_pk_fma_f32 / v_pk_mul_f32 beside the other wave's MFMAs made this
                        // loop ~3x slower (down1.3 forward 82 -> 105 us, A/B)
                        const float ya = fmaf(v.get(e), S.sc[e / 2][0], S.sh[e / 2][0]), yb = fmaf(v.get(e + 1), S.sc[e / 2][1], S.sh[e / 2][1]);
                        v.set(e, fmaxf(ya, LRELU_SLOPE * ya));
                        v.set(e + 1, fmaxf(yb, LRELU_SLOPE * yb));
                    }
                }
                {
                    const bool zero = (hit & (0xfu << (4 * j))) != 0;
                    u32x4 bits = __builtin_bit_cast(u32x4, v.v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bits[e] = zero ? 0u : bits[e];
                    v.v = __builtin_bit_cast(decltype(v.v), bits);
                }
                *reinterpret_cast<decltype(v.v)*>(lds_h + (q + PT * j) * 16) = v.v;
            }
            PPT_MARK(0)
#pragma unroll
            for (int j = 0; j < kPcWSlots; ++j) *reinterpret_cast<u32x4*>(lds_w + (q + PT * j) * 16) = S.w[j];
            PPT_MARK(1)
            issue(S, n, ty0, tx0, kc);
            PPT_MARK(3)
        };

        // item cursor of the loads (runs three items ahead of the MFMAs; past the end it stays on the last item: the tail
        // re-loads it and stores an image nobody reads)
        int l_idx = 0, l_tile = bt0, l_kc = 0, ln, lty0, ltx0;
        decode(l_tile, ln, lty0, ltx0);
        auto step = [&]() {
            if (l_idx + 1 < total) {
                ++l_idx;
                if (++l_kc == p.nchunks) {
                    l_kc = 0;
                    ++l_tile;
                    next_tile(ln, lty0, ltx0);
                }
            }
        };
        issue(S0, ln, lty0, ltx0, l_kc);
        step();
        issue(S1, ln, lty0, ltx0, l_kc);
        step();
        PPT_MARK(8)
        // two items per trip, no exit between them (with a `break` in the middle hipcc's wait-count pass merges the two halves
        // conservatively: `vmcnt(0)` before the first half's commit, i.e. a wait for the loads issued one item ago); an odd
        // item count is padded with a half that stages the last item again into a buffer nobody reads
        int bi = 0;                 // buffer of the item to commit = item % 3
        u32x4 dseen = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int c = 0; c < total; c += 2) {
            pc_wait_ge<1>(done_p, dseen, c - 2, broken);         // item c - 3 (same buffer) has been read by all four MFMA waves
            PPT_MARK(5)
            commit_issue(S0, smem + bi * kPcBuf, ln, lty0, ltx0, l_kc);
            if (lane == 0) pc_signal(ready_a + 4 * (wave - 4));
            dseen = pc_peek<1>(done_p);
            step();
            bi = bi == kPcStages - 1 ? 0 : bi + 1;
            PPT_MARK(6)
            pc_wait_ge<1>(done_p, dseen, c - 1, broken);
            PPT_MARK(5)
            commit_issue(S1, smem + bi * kPcBuf, ln, lty0, ltx0, l_kc);
            if (lane == 0) pc_signal(ready_a + 4 * (wave - 4));
            dseen = pc_peek<1>(done_p);
            step();
            bi = bi == kPcStages - 1 ? 0 : bi + 1;
            PPT_MARK(6)
        }
        PPT_DUMP()
        return;
    }

    // ------------------------------------------------------------------ consumer (waves 0-3)
    // weight image rows q = tap*BN + ni*32 + lr, 32 B each, 16-B slot s at position s ^ ((q >> 3) & 1) = s ^ ((lr >> 3) & 1)
    const int a_off = kPcHaloBytes + lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    // halo image rows r = (MI*wave + mi + ky)*34 + kx + lr, slot s at position s ^ ((r >> 3) & 1): one offset per (mi+ky, kx)
    int xb[(MI + 2) * 3];
#pragma unroll
    for (int yy = 0; yy < MI + 2; ++yy)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = (MI * wave + yy) * kPcHaloW + kx + lr;
            xb[yy * 3 + kx] = r * 32 + ((lh ^ ((r >> 3) & 1)) << 4);
        }
    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    const bool has_br = p.bias != nullptr || p.relu_out != 0;
    const int gs = p.groups > 0 ? p.Cout / p.groups : 4;
    // GroupNorm partial sums per lane: a group spans >= 16 channels here (host-checked), i.e. the quad pairs {0,1} and {2,3}
    // of a fragment each lie in one group
    // (NI = 1, the 32-channel layer: groups of 4 channels = one quad each -> one accumulator per quad)
    constexpr int QSH = NI == 4 ? 1 : 0;            // quads per accumulator = 1 << QSH (the 64-channel layers: groups of 8 = one quad of both lane halves)
    constexpr int NSQ = STATS ? (4 >> QSH) : 1;
    // pairs: v_pk_add_f32 / v_pk_fma_f32 straight from the accumulator registers (the tall variant: single floats - its 16 pixel-
    // fragment registers more leave no room for 32 statistics registers)
    constexpr bool ST1 = MI > 2;
    typedef typename std::conditional<ST1, float, pc_f32x2>::type st_t;
    st_t st_s[NI][NSQ], st_ss[NI][NSQ];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < NSQ; ++q) { st_s[ni][q] = st_t{}; st_ss[ni][q] = st_t{}; }

    // epilogue of a finished tile (as conv_ring.hip): bias / ReLU, GroupNorm partial sums, packed 16-bit values exchanged
    // between the lane halves (permlane32) -> 16-byte stores.  Whole tiles only (host-checked): no predication.
    auto epilogue = [&](int n, int ty0, int tx0) {
        const size_t e0 = ((size_t)(n * p.H + ty0) * p.W + tx0) * p.Cout + bn0;
        char* obase = (char*)p.out + e0 * sizeof(T);
        int lh_e = lh;
        asm volatile("" : "+v"(lh_e));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
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
                    if constexpr (STATS && ST1) {
                        st_s[ni][q >> QSH] += (v[0] + v[1]) + (v[2] + v[3]);
                        st_ss[ni][q >> QSH] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], st_ss[ni][q >> QSH]))));
                    } else if constexpr (STATS) {
                        const pc_f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
                        st_s[ni][q >> QSH] += v01;
                        st_s[ni][q >> QSH] += v23;
                        st_ss[ni][q >> QSH] = __builtin_elementwise_fma(v01, v01, st_ss[ni][q >> QSH]);
                        st_ss[ni][q >> QSH] = __builtin_elementwise_fma(v23, v23, st_ss[ni][q >> QSH]);
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
                int co = bn0 + ni * 32 + (16 >> (1 - QSH)) * q + 4 * lh;
                asm volatile("" : "+v"(co));
                float s, ss;
                if constexpr (ST1) { s = half_wave_sum(st_s[ni][q]); ss = half_wave_sum(st_ss[ni][q]); }
                else { s = half_wave_sum(st_s[ni][q][0] + st_s[ni][q][1]); ss = half_wave_sum(st_ss[ni][q][0] + st_ss[ni][q][1]); }
                if (lr == 0) {
                    const int g = co / gs;
                    double* sp = p.stats + stat_slot_off_id(bid, p.N, p.groups) + ((size_t)n * p.groups + g) * 2;
                    atomic_add_f64(sp, (double)s);
                    atomic_add_f64(sp + 1, (double)ss);
                }
                st_s[ni][q] = st_t{};
                st_ss[ni][q] = st_t{};
            }
    };

    int tile = bt0, kc = 0, n, ty0, tx0;
    decode(tile, n, ty0, tx0);
    PPT_DECL
    int bi = 0;
    u32x4 rseen = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int c = 0; c < total; ++c) {
        pc_wait_ge<NP / 4>(ready_p, rseen, c + 1, broken);      // all four staging waves have stored item c
        PPT_MARK(5)
        const char* buf = smem + bi * kPcBuf;
        bi = bi == kPcStages - 1 ? 0 : bi + 1;
        const char* wl = buf + a_off;
        // Fragment pipeline (conv_ring.hip): pixel fragments of tap t+1 and weight fragments up to AD steps ahead are requested
        // before the MFMAs that use the current ones are issued
        constexpr int AD = MRISR_PC_AD;
        frag_t af[AD + 1], bf[2][MI];
        auto load_b = [&](int tap, int set) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) bf[set][mi] = *reinterpret_cast<const frag_t*>(buf + xb[(mi + ky) * 3 + kx]);
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
                if (tap == 6 && ni == 0) rseen = pc_peek<NP / 4>(ready_p);      // is the next item there?  (answer used at its start)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (MRISR_PC_DBG & 1) asm volatile("" ::"v"(af[idx % (AD + 1)]), "v"(bf[tap & 1][mi]));
                    else acc[ni][mi] = Frag16<T>::mma(af[idx % (AD + 1)], bf[tap & 1][mi], acc[ni][mi]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (lane == 0) pc_signal(done_a + 4 * wave);       // (behind this wave's last fragment read of the item, in LDS issue order)
        PPT_MARK(6)
        if (++kc == p.nchunks) {
            if (!(MRISR_PC_DBG & 4)) epilogue(n, ty0, tx0);
            const int n_prev = n;
            kc = 0;
            ++tile;
            if (tile < bt1) next_tile(n, ty0, tx0);
            if (STATS && (tile >= bt1 || n != n_prev)) flush_stats(n_prev);
            PPT_MARK(4)
        }
    }
    PPT_DUMP()
}

// ------------------------------------------------------------------------------------------------ host side
#ifndef MRISR_KERNEL_ONLY
#ifdef MRISR_PC_PT
extern "C" int mrisr_debug_phase_reset() {
    static unsigned long long zeros[96];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pc_cycles), zeros, sizeof(zeros));
}
extern "C" int mrisr_debug_phase_cycles(unsigned long long* out96) {
    return (int)hipMemcpyFromSymbol(out96, HIP_SYMBOL(g_pc_cycles), sizeof(unsigned long long) * 96);
}
#endif

// Does this launch take the producer / consumer kernel?  (p: filled by conv_fill_params.)
// 0 = no, 4 = 128-channel blocks, 2 = 64-channel blocks on tall tiles, 1 = the narrow blend variant (Cout = 32, two activated
// sources blended by the staging waves)
int conv_pc_kind(const mrisr_conv_desc* d, const ConvParams& p) {
#ifdef MRISR_NO_PC
    return 0;
#endif
    if (!d->wpacked_ring) return 0;
    if (d->dtype == MRISR_F32 || d->ksize != 3 || d->Cin % 16) return 0;
    if (d->out_mode != MRISR_OUT_PLAIN || d->relu_mask) return 0;
    if (d->nsrc < 1 || d->nsrc > 2) return 0;
    for (int i = 0; i < d->nsrc; ++i) {
        if (d->src[i].spatial != MRISR_SP_NONE) return 0;
        if (d->src[i].mode != d->src[0].mode) return 0;
        if (d->src[i].mode != MRISR_SRC_NORM && d->src[i].mode != MRISR_SRC_RAW) return 0;
    }
    for (int i = 0; i < d->nsrc; ++i)                               // 32-bit byte offsets into each source (buffer loads)
        if ((unsigned long long)d->N * d->src[i].H * d->src[i].W * d->src[i].C * 2ull >= (1ull << 31)) return 0;
    if (d->H % 8 || d->W % 32) return 0;                            // whole 8 x 32 tiles only (unpredicated stores)
    const int tiles = d->N * (d->H / 8) * (d->W / 32);
    if (d->combine == MRISR_COMBINE_CONCAT && d->Cout % kPcBN && d->Cout % 64 == 0) {
        // 64-channel blocks on tall tiles (16 x 32 pixels): the 64-channel layers and the input gradients that end in 64 channels
#ifdef MRISR_NO_PC64
        return 0;
#endif
        if (d->H % 16) return 0;
        if (d->nsrc == 2 && (d->src[0].C % 16)) return 0;
        if (d->src[0].mode == MRISR_SRC_RAW && d->Cin < 128) return 0;  // (64 -> 64 input gradient at 256^2: 111 vs 101 us on the LDS-DMA kernel)
        if (d->stats && ((d->Cout / d->groups) & 3)) return 0;      // a GroupNorm group spans whole accumulator quads
        return (long)(tiles / 2) * (d->Cout / 64) * 4 >= (long)p.cus * 3 ? 2 : 0;
    }
    if (d->combine == MRISR_COMBINE_BLEND) {
        // final_conv.0 without the materialised blend (eval forward): 32 output channels, two activated sources of the conv's
        // own geometry; GroupNorm groups of 4 channels = one accumulator quad each
        if (d->Cout != 32 || d->nsrc != 2 || d->src[0].mode != MRISR_SRC_NORM || !d->blend_alpha) return 0;
        for (int i = 0; i < 2; ++i)
            if (d->src[i].C != d->Cin || d->src[i].H != d->H || d->src[i].W != d->W || d->src[i].off_y || d->src[i].off_x) return 0;
        if (d->stats && (d->Cout / d->groups) % 4) return 0;
        return (long)tiles * 4 >= (long)p.cus * 3 ? 1 : 0;
    }
    if (d->combine != MRISR_COMBINE_CONCAT || d->Cout % kPcBN) return 0;
    if (d->nsrc == 2 && (d->src[0].C % 16)) return 0;               // a 16-channel chunk lies in one source
    // stored sources: the LDS-DMA kernels stage them without any vector work; this kernel wins from 128 input channels on
    // (8 items per tile epilogue; 128 -> 64 input gradient at 256^2: 171 vs 162 us)
    if (d->src[0].mode == MRISR_SRC_RAW && d->Cin < 128) return 0;
    if (d->stats && ((d->Cout / d->groups) & 15)) return 0;         // a GroupNorm group spans whole 16-channel quad pairs
    const int ncb = d->Cout / kPcBN;
    return (long)tiles * ncb * 4 >= (long)p.cus * 3 ? 4 : 0;        // >= 0.75 work items per CU
}
bool conv_pc_eligible(const mrisr_conv_desc* d, const ConvParams& p) { return conv_pc_kind(d, p) != 0; }

template <typename T, bool NORM, bool STATS, int NI, bool BLEND, int MI = 2>
static void launch_pc_k(const ConvParams& p, int grid, hipStream_t s) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_kernel<T, NORM, STATS, NI, BLEND, MI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((conv_pc_kernel<T, NORM, STATS, NI, BLEND, MI>), dim3(grid), dim3(pc_threads(BLEND)), pc_lds(NI, pc_np(BLEND), MI), s, p);
}

template <typename T>
static int launch_pc_t(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s) {
    const int kind = conv_pc_kind(d, cp);
    const int bn = kind == 1 ? 32 : kind == 2 ? 64 : kPcBN;
#ifdef MRISR_PC_BLEND_SHORT
    const bool tall_blend = false;
#else
    const bool tall_blend = kind == 1 && d->H % 16 == 0;            // the blend variant on tall items: 10 % less halo per pixel
#endif
    const int tr = (kind == 2 || tall_blend) ? 16 : 8;
    ConvParams p = cp;
    p.wpacked = d->wpacked_ring;
    p.nchunks = d->Cin / 16; p.ncb = d->Cout / bn;
    p.tiles_x = d->W / 32; p.tiles_y = d->H / tr;
    p.ntiles = d->N * p.tiles_x * p.tiles_y;
    p.groups = d->stats ? d->groups : 0;
    int per_cb = cp.cus / p.ncb;
    if (per_cb < 1) per_cb = 1;
    if (per_cb > p.ntiles) per_cb = p.ntiles;
    p.tiles_per_block = ceil_div(p.ntiles, per_cb);
    per_cb = ceil_div(p.ntiles, p.tiles_per_block);
    const int grid = per_cb * p.ncb;
    const bool norm = d->src[0].mode == MRISR_SRC_NORM, stats = d->stats != nullptr;
    if (kind == 1 && tall_blend) {
        if (stats) launch_pc_k<T, true, true, 1, true, 4>(p, grid, s);
        else launch_pc_k<T, true, false, 1, true, 4>(p, grid, s);
    } else if (kind == 1) {
        if (stats) launch_pc_k<T, true, true, 1, true>(p, grid, s);
        else launch_pc_k<T, true, false, 1, true>(p, grid, s);
    } else if (kind == 2) {
        if (norm && stats) launch_pc_k<T, true, true, 2, false, 4>(p, grid, s);
        else if (norm) launch_pc_k<T, true, false, 2, false, 4>(p, grid, s);
        else if (stats) launch_pc_k<T, false, true, 2, false, 4>(p, grid, s);
        else launch_pc_k<T, false, false, 2, false, 4>(p, grid, s);
    } else if (norm && stats) launch_pc_k<T, true, true, 4, false>(p, grid, s);
    else if (norm) launch_pc_k<T, true, false, 4, false>(p, grid, s);
    else if (stats) launch_pc_k<T, false, true, 4, false>(p, grid, s);
    else launch_pc_k<T, false, false, 4, false>(p, grid, s);
    MRISR_CHECK_LAUNCH("conv_forward(pc)");
    return MRISR_OK;
}

int launch_conv_pc(const mrisr_conv_desc* d, const ConvParams& cp, hipStream_t s) {
    if (d->dtype == MRISR_BF16) return launch_pc_t<bf16_t>(d, cp, s);
    return launch_pc_t<f16_t>(d, cp, s);
}
#endif
