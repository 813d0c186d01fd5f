// How do an MFMA wave and a VALU wave share one SIMD on gfx950?  (tuning experiment for the forward kernel's
// antiphase halves)   hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/coissue.bin
// One 8-wave workgroup per CU = 2 waves per SIMD (waves w and w+4 share SIMD w).  Roles per half: M = back-to-back
// v_mfma_f32_32x32x16_bf16 on 4 accumulators (optionally one ds_read_b128 per MFMA), V = independent v_fma_f32 chains
// (optionally v_pk_fma_f32), I = idle (exits at once).  Prints s_memtime cycles per MFMA / per VALU instruction for
// each half, for every role combination.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { ROLE_IDLE = 0, ROLE_MFMA = 1, ROLE_MFMA_LDS = 2, ROLE_VALU = 3, ROLE_VALU_PK = 4 };

__global__ __launch_bounds__(512, 2) void k(int role0, int role1, int iters, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int half = threadIdx.x >> 8;
    const int role = half ? role1 : role0;
    for (int i = threadIdx.x; i < 32768 / 4; i += 512) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = t0;
    float s = 0.f;
    if (role == ROLE_MFMA || role == ROLE_MFMA_LDS) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        const char* lp = lds + (threadIdx.x & 255) * 16;
        t0 = __builtin_amdgcn_s_memtime();
        if (role == ROLE_MFMA) {
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        } else {
            // software-pipelined by one step and unrolled by two (two named fragment sets: no register copies)
            bf16x8 a0 = *(const bf16x8*)(lp), a1 = *(const bf16x8*)(lp + 4096), b0 = *(const bf16x8*)(lp + 8192), b1 = *(const bf16x8*)(lp + 12288);
            bf16x8 e0, e1, f0, f1;
            for (int i = 0; i < iters; i += 2) {
                e0 = *(const bf16x8*)(lp + 16384); e1 = *(const bf16x8*)(lp + 16384 + 4096);
                f0 = *(const bf16x8*)(lp + 16384 + 8192); f1 = *(const bf16x8*)(lp + 16384 + 12288);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = *(const bf16x8*)(lp); a1 = *(const bf16x8*)(lp + 4096);
                b0 = *(const bf16x8*)(lp + 8192); b1 = *(const bf16x8*)(lp + 12288);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e0, f0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e0, f1, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e1, f0, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e1, f1, c3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    } else if (role == ROLE_VALU) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = 0.5f + 0.001f * (threadIdx.x + i);
        const float m = 0.999f, ad = 0.001f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], m, ad);        // 32 v_fma_f32 per iteration
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += x[i];
    } else if (role == ROLE_VALU_PK) {
        f32x2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f32x2{0.5f + 0.001f * (threadIdx.x + i), 0.25f};
        const f32x2 m = {0.999f, 0.998f}, ad = {0.001f, 0.002f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], m, ad);   // 32 v_pk_fma_f32 per iteration
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    }
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 7 && (threadIdx.x & 255) == 0) out[half] = t1 - t0;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const char* names[] = {"idle", "mfma", "mfma+lds", "valu", "valu_pk"};
    const int combos[][2] = {{1, 0}, {0, 1}, {2, 0}, {3, 0}, {0, 3}, {4, 0}, {1, 3}, {3, 1}, {2, 3}, {3, 2}, {1, 4}, {4, 1}, {2, 4}, {4, 2}, {1, 1}, {3, 3}};
    const int iters_m = 4000, iters_v = 4000;        // 16000 MFMAs (512k cycles alone) vs 128000 VALU ops (512k cycles alone)
    for (auto& c : combos) {
        hipMemset(out, 0, 16);
        // same iteration count for both halves: an MFMA iteration = 4 MFMAs = 128 cycles alone, a VALU iteration = 32 ops = 128 cycles alone
        k<<<prop.multiProcessorCount, 512>>>(c[0], c[1], iters_m, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("half0=%-9s half1=%-9s :", names[c[0]], names[c[1]]);
        for (int hf = 0; hf < 2; ++hf) {
            const int r = c[hf];
            if (r == 1 || r == 2) printf("  half%d %.1f cycles/MFMA", hf, (double)h[hf] / (4.0 * iters_m));
            else if (r >= 3) printf("  half%d %.2f cycles/VALU-op", hf, (double)h[hf] / (32.0 * iters_v));
        }
        printf("\n");
    }
    return 0;
}
