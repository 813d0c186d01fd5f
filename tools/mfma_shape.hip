// Which bf16 MFMA shape sustains more FLOP/s on this MI355X under DVFS when every operand is re-read from LDS at the
// convolution kernels' ratio (64 x 64 output tile per wave, K = 32 per step: 4 + 4 ds_read_b128, then 8 MFMAs 32x32x16 or
// 16 MFMAs 16x16x32)?  Random operands, 8 waves per workgroup (2 per SIMD), one workgroup per CU, ~2 ms per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o build/mfma_shape && build/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void loop(int iters, float* sink, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 1024];      // 128 KiB of random bf16 in [-1, 1)
    for (int i = threadIdx.x; i < 64 * 1024; i += 512) {
        const unsigned h = mix32(i * 2654435761u + blockIdx.x);
        const float v = (float)(h >> 8) * (2.f / 16777216.f) - 1.f;
        lds[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = (const char*)lds + wave * 8192;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        // lane l: row l & 31 (64-B rows, chunk XOR-swizzled by (row >> 2) & 3), k-half l >> 5
        const int r = lane & 31, lh = lane >> 5;
        const int off = r * 64 + ((lh ^ ((r >> 2) & 3)) << 4);
        for (int it = 0; it < iters; ++it) {
            const char* p = base + (it & 7) * 4096 * 0;      // same tiles every step (traffic is what matters)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *(const bf16x8*)(p + i * 2048 + (off ^ (32 * ks)));
                    b[i] = *(const bf16x8*)(p + 4096 + i * 2048 + (off ^ (32 * ks)));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    } else {
        f32x4 acc[4][4] = {};
        // lane l: row l & 15, chunk l >> 4, position XOR 2 * ((row >> 2) & 1): conflict-free for ds_read_b128
        const int r = lane & 15, c = lane >> 4;
        const int off = r * 64 + ((c ^ (((r >> 2) & 1) << 1)) << 4);
        for (int it = 0; it < iters; ++it) {
            const char* p = base;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *(const bf16x8*)(p + i * 1024 + off);
                b[i] = *(const bf16x8*)(p + 4096 + i * 1024 + off);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) s += acc[i][j][k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    float* sink; unsigned long long* stamps;
    hipMalloc(&sink, 4); hipMalloc(&stamps, 16);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int round = 0; round < 4; ++round)
        for (int shape = 32; shape >= 16; shape -= 16) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (shape == 32) loop<32><<<cus, 512>>>(iters, sink, stamps); else loop<16><<<cus, 512>>>(iters, sink, stamps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
            const double flops = 2.0 * 64 * 64 * 32 * (double)iters * 8 * cus;
            printf("round %d  %s: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  %.1f cycles per K=32 step per wave\n", round,
                   shape == 32 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / iters);
        }
    return 0;
}
