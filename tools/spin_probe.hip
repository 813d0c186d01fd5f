// Probe for profiles/NOTES.md R3-11: does an inline-asm LDS polling loop wait?  Wave 1 sleeps ~T cycles, then bumps four LDS counters;
// wave 0 polls them with the asm loop and records how long it waited and what it read.   hipcc --offload-arch=gfx950 -O3 tools/spin_probe.hip -o /tmp/spin_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned long long* out, int delay) {
    __shared__ unsigned cnt[16];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (t < 16) cnt[t] = 0u;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)cnt;
    if (wave == 1) {
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(16);
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) asm volatile("ds_add_u32 %0, %1" ::"v"(addr + 4 * k), "v"(1u) : "memory");
        }
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned a, b, c, d;
    int got, left;
    const int target = 1;
    asm volatile(
        "s_mov_b32 %5, 0x200000\n"
        "1:\n\t"
        "ds_read_b32 %0, %6\n\t"
        "ds_read_b32 %1, %6 offset:4\n\t"
        "ds_read_b32 %2, %6 offset:8\n\t"
        "ds_read_b32 %3, %6 offset:12\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_min_i32 %0, %0, %1\n\t"
        "v_min_i32 %2, %2, %3\n\t"
        "v_min_i32 %0, %0, %2\n\t"
        "v_readfirstlane_b32 %4, %0\n\t"
        "s_cmp_ge_i32 %4, %7\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sub_u32 %5, %5, 1\n\t"
        "s_cmp_eq_u32 %5, 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sleep 2\n\t"
        "s_branch 1b\n"
        "2:"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&s"(got), "=&s"(left)
        : "v"(addr), "s"(target)
        : "memory", "scc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)got; out[2] = (unsigned long long)left; out[3] = a; }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    for (int delay : {0, 10, 100, 1000}) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d, delay);
        unsigned long long h[4];
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("delay %5d x s_sleep 16: waited %8llu cycles, got %lld, polls left 0x%llx, min %llu\n", delay, h[0], (long long)(int)h[1], h[2], h[3]);
    }
    return 0;
}
