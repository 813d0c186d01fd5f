// Tuning aid: what does a 16-byte-per-lane global store cost as a function of how its 64 pieces are laid out?
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern.bin tools/store_pattern.hip && tools/store_pattern.bin
// A "tile" is 64 pixels x 256 bytes (one 128-channel 16-bit NHWC row block = what one conv_pc MFMA wave stores per tile: 16 KB, 16
// store instructions).  Patterns (lane l of instruction i writes 16 bytes at):
//   0  pixel l % 32, chunk 2 * (i % 8) + l / 32          (the epilogue's layout today: 64 pieces in 32 lines, 2 adjacent pieces per line)
//   1  pixel 4 * i + l / 16, chunk l % 16                (whole 256-byte pixel rows: 4 rows per instruction, fully contiguous)
//   2  pixel 8 * (i / 2) + l / 8, chunk 8 * (i % 2) + l % 8   (128-byte half rows of 8 pixels, adjacent lanes adjacent)
//   3  8 pixels per instruction, the 8 lanes of a pixel split 4 + 4 over the two lane halves, interleaved chunks (128-byte lines)
//   4  16 pixels per instruction, 2 + 2 lanes per pixel: 64-byte half lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(char* out, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    for (int t = 0; t < tiles_per_wave; ++t) {
        char* base = out + ((size_t)wave * tiles_per_wave + t) * 16384;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int px, ch;
            if (PAT == 0) { px = (lane & 31) + 32 * (i >> 3); ch = 2 * (i & 7) + (lane >> 5); }
            else if (PAT == 1) { px = 4 * i + (lane >> 4); ch = lane & 15; }
            else if (PAT == 2) { px = 8 * (i >> 1) + (lane >> 3); ch = 8 * (i & 1) + (lane & 7); }
            else if (PAT == 3) { px = 8 * (i >> 1) + ((lane & 31) >> 2); ch = 8 * (i & 1) + 2 * (lane & 3) + (lane >> 5); }
            else { px = 16 * (i >> 2) + ((lane & 31) >> 1); ch = 4 * (i & 3) + 2 * (lane & 1) + (lane >> 5); }
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + px * 256 + ch * 16));
        }
    }
}

template <int PAT>
static float run(char* buf, int blocks, int tpw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    store_kernel<PAT><<<blocks, 256>>>(buf, tpw);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) store_kernel<PAT><<<blocks, 256>>>(buf, tpw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const int blocks = 1024, tpw = 16;                      // 4096 waves x 16 tiles x 16 KB = 1 GB
    const size_t bytes = (size_t)blocks * 4 * tpw * 16384;
    char* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float ms[5] = {run<0>(buf, blocks, tpw), run<1>(buf, blocks, tpw), run<2>(buf, blocks, tpw), run<3>(buf, blocks, tpw), run<4>(buf, blocks, tpw)};
    for (int p = 0; p < 5; ++p) printf("pattern %d: %.3f ms  %.2f TB/s\n", p, ms[p], bytes / ms[p] * 1e-9);
    // a compute-like occupancy: 256 workgroups (one per CU), fewer waves in flight, as in the convolution epilogue
    float m2[5] = {run<0>(buf, 256, 64), run<1>(buf, 256, 64), run<2>(buf, 256, 64), run<3>(buf, 256, 64), run<4>(buf, 256, 64)};
    for (int p = 0; p < 5; ++p) printf("256 workgroups, pattern %d: %.3f ms  %.2f TB/s\n", p, m2[p], bytes / m2[p] * 1e-9);
    return 0;
}
