// Sustained dense-MFMA rate of this MI355X under DVFS, and what s_memtime counts.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build/mfma_peak && build/mfma_peak
// Every wave issues back-to-back v_mfma_f32_32x32x16_bf16 on 4 independent accumulators (no memory traffic), with
// 1 or 2 waves per SIMD; reports TFLOP/s from HIP events, the s_memtime rate (ticks per microsecond of
// s_memrealtime, 100 MHz) and the implied MFMA issue rate per SIMD in s_memtime ticks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512, 2) void mfma_loop(int iters, unsigned long long* stamps, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* stamps; float* sink;
    hipMalloc(&stamps, 16); hipMalloc(&sink, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int waves = 4; waves <= 8; waves += 4) {       // waves per workgroup = per CU: 1 or 2 per SIMD
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            mfma_loop<<<cus, waves * 64>>>(iters, stamps, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
            const double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * waves * cus;
            const double us_rt = h[1] / 100.0;
            const double mfma_per_simd = 4.0 * iters * (waves / 4);
            printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s; s_memtime %.1f ticks/us (%.3f GHz); %.1f s_memtime ticks per MFMA per SIMD; kernel %.1f us by s_memrealtime\n",
                   waves / 4, ms, flops / ms / 1e9, h[0] / us_rt, h[0] / us_rt / 1e3, h[0] / mfma_per_simd, us_rt);
        }
    }
    return 0;
}
