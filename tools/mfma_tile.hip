// Does a larger per-wave output tile (fewer LDS operand bytes per MFMA) buy sustained FLOP/s on this MI355X under DVFS?
// bf16 32x32x16 MFMAs, every operand re-read from LDS with ds_read_b128 each K = 16 step, random operands, one workgroup per CU:
//   64 x 64  per wave (2 A + 2 B fragments per 4 MFMAs: 1.00 KB per MFMA, 64 accumulator registers)  - the convolution kernels
//   64 x 128 per wave (2 A + 4 B per 8 MFMAs: 0.75 KB per MFMA, 128 accumulators), 8 waves per workgroup
//   128 x 128 per wave (4 + 4 per 16: 0.50 KB per MFMA, 256 accumulators), 4 waves per workgroup (one per SIMD)
//   and the first two again with 1 KiB of LDS writes per 8 MFMAs beside the reads (the kernels' operand staging rate)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_tile.hip -o build/mfma_tile && build/mfma_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int NA, int NB, int THREADS, int NW = 0>
__global__ __launch_bounds__(THREADS) void loop(int iters, float* sink, unsigned long long* stamps, const char* gsrc) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 1024];      // 128 KiB of random bf16 in [-1, 1)
    for (int i = threadIdx.x; i < 64 * 1024; i += THREADS) {
        const unsigned h = mix32(i * 2654435761u + blockIdx.x);
        const float v = (float)(h >> 8) * (2.f / 16777216.f) - 1.f;
        lds[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (NW == 100) {
        // loader waves (one per SIMD, waves 8-11 of a 12-wave workgroup): all the workgroup's LDS-DMA pieces, 2 per K = 32
        // step each (= 8 KiB per step per workgroup, as when every MFMA wave issues 1), nothing else
        if (wave >= 8) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)((char*)lds + 98304 + (wave - 8) * 8192 + ((2 * it + w) & 7) * 1024);
                    const char* g = gsrc + (size_t)(((2 * it + w) * 4 + wave) & 1023) * 1024 + lane * 16;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane((int)dst)) : "memory");
                }
                if ((it & 3) == 3) __builtin_amdgcn_s_waitcnt(0x0f70);
            }
            return;
        }
    }
    const char* base = (const char*)lds + (THREADS == 1024 ? wave * 6144 : wave * (NW == 100 ? 12288 : 16384));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NA][NB] = {};
    const int r = lane & 31, lh = lane >> 5;
    const int off = r * 64 + ((lh ^ ((r >> 2) & 3)) << 4);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    char* wbase = (char*)lds + 98304 + wave * 4096 + lane * 16;            // write region: beyond every wave's operands
    for (int it = 0; it < iters; ++it) {
        // NW x 1 KiB per K = 32 step landing in LDS beside the fragment reads (the kernels' operand staging: ~0.135 KB per
        // MFMA).  NW > 0: by LDS-DMA (global_load_lds_dwordx4 from an L2-resident buffer), NW < 0: by ds_write_b128
        if constexpr (NW == 50) {          // 1 KiB per 8 MFMAs with 4 MFMAs per step: every second step
            if (it & 1) {
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)((char*)lds + 98304 + (wave & 7) * 4096 + (it & 3) * 1024);
                const char* g = gsrc + (size_t)((it * 16 + wave) & 1023) * 1024 + lane * 16;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane((int)dst)) : "memory");
            }
            if ((it & 15) == 15) __builtin_amdgcn_s_waitcnt(0x0f70);
        } else if constexpr (NW > 0) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)((char*)lds + 98304 + wave * 4096 + ((it + w) & 3) * 1024);
                const char* g = gsrc + (size_t)(((it + w) * 8 + wave) & 1023) * 1024 + lane * 16;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane((int)dst)) : "memory");
            }
            if ((it & 7) == 7) __builtin_amdgcn_s_waitcnt(0x0f70);
        } else if constexpr (NW < 0) {
#pragma unroll
            for (int w = 0; w < -NW; ++w) *(u32x4*)(wbase + ((it + w) & 3) * 1024) = u32x4{(unsigned)it, 1u, 2u, 3u};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[NA], b[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = *(const bf16x8*)(base + i * 2048 + (off ^ (32 * ks)));
#pragma unroll
            for (int i = 0; i < NB; ++i) b[i] = *(const bf16x8*)(base + NA * 2048 + i * 2048 + (off ^ (32 * ks)));
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    float* sink; unsigned long long* stamps;
    hipMalloc(&sink, 4); hipMalloc(&stamps, 16);
    char* gsrc; hipMalloc(&gsrc, 1 << 20); hipMemset(gsrc, 1, 1 << 20);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int round = 0; round < 4; ++round)
        for (int v = 0; v < 10; ++v) {
            if (v == 7) continue;              // (un-paced loader waves: 48x slower, see profiles/NOTES.md R2-13)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            int waves, na, nb, it = iters;
            if (v == 0) { waves = 8; na = 2; nb = 2; loop<2, 2, 512><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 1) { waves = 8; na = 2; nb = 4; it = iters / 2; loop<2, 4, 512><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 2) { waves = 4; na = 4; nb = 4; it = iters / 2; loop<4, 4, 256><<<cus, 256>>>(it, sink, stamps, gsrc); }
            else if (v == 3) { waves = 8; na = 2; nb = 2; loop<2, 2, 512, 1><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 4) { waves = 8; na = 2; nb = 4; it = iters / 2; loop<2, 4, 512, 2><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 5) { waves = 8; na = 2; nb = 2; loop<2, 2, 512, -1><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 6) { waves = 8; na = 2; nb = 4; it = iters / 2; loop<2, 4, 512, -2><<<cus, 512>>>(it, sink, stamps, gsrc); }
            else if (v == 7) { waves = 8; na = 2; nb = 2; loop<2, 2, 768, 100><<<cus, 768>>>(it, sink, stamps, gsrc); }
            else if (v == 8) { waves = 16; na = 1; nb = 2; it = iters * 2; loop<1, 2, 1024, 0><<<cus, 1024>>>(it, sink, stamps, gsrc); }
            else { waves = 16; na = 1; nb = 2; it = iters * 2; loop<1, 2, 1024, 50><<<cus, 1024>>>(it, sink, stamps, gsrc); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
            const double flops = 2.0 * (32.0 * na) * (32.0 * nb) * 32 * (double)it * waves * cus;
            printf("round %d  %s wave tile %3d x %3d, %d waves: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  %.1f cycles per MFMA per SIMD\n", round, v == 9 ? "+LDS-DMA   " : v == 8 ? "           " : v == 7 ? "+DMA waves " : (v >= 5 ? "+ds_write  " : (v >= 3 ? "+LDS-DMA   " : "           ")), 32 * na,
                   32 * nb, waves, ms, flops / ms / 1e9, h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / it / (2.0 * na * nb) / (waves / 4));
        }
    return 0;
}
