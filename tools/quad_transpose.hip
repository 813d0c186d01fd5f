#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// D = (mask bit of this lane) ? keep : (src taken from the lane XOR 1 / XOR 2 of its quad)       (v_cndmask_b32_dpp: VCC ? src1 : dpp(src0))
__device__ __forceinline__ unsigned sel_x1(unsigned from_partner, unsigned keep, unsigned long long mask) {
    unsigned d;
    asm volatile("s_mov_b64 vcc, %3\n\ts_nop 1\n\tv_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(d) : "v"(from_partner), "v"(keep), "s"(mask) : "vcc");
    return d;
}
__device__ __forceinline__ unsigned sel_x2(unsigned from_partner, unsigned keep, unsigned long long mask) {
    unsigned d;
    asm volatile("s_mov_b64 vcc, %3\n\ts_nop 1\n\tv_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(d) : "v"(from_partner), "v"(keep), "s"(mask) : "vcc");
    return d;
}
// 4x4 transpose of 16-byte elements across the four lanes of a quad: O[i] on lane j = R[j] of lane i
__device__ __forceinline__ void quad_transpose(u32x4 (&R)[4], int lane) {
    const unsigned long long even1 = 0x5555555555555555ull, odd1 = ~even1;      // lanes with bit 0 clear / set
    const unsigned long long even2 = 0x3333333333333333ull, odd2 = ~even2;      // lanes with bit 1 clear / set
#pragma unroll
    for (int k = 0; k < 4; k += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned a = R[k][e], b = R[k + 1][e];
            R[k][e] = sel_x1(b, a, even1);          // even lanes keep a, odd lanes take the even partner's b
            R[k + 1][e] = sel_x1(a, b, odd1);       // odd lanes keep b, even lanes take the odd partner's a
        }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned a = R[k][e], b = R[k + 2][e];
            R[k][e] = sel_x2(b, a, even2);
            R[k + 2][e] = sel_x2(a, b, odd2);
        }
}
__global__ void k(const u32x4* in, u32x4* out) {
    u32x4 R[4];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 4; ++i) R[i] = in[threadIdx.x * 4 + i];
    quad_transpose(R, lane);
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = R[i];
}
int main() {
    u32x4 *in, *out;
    hipMallocManaged(&in, 64 * 4 * 16); hipMallocManaged(&out, 64 * 4 * 16);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) for (int e = 0; e < 4; ++e) ((unsigned*)in)[(l * 4 + r) * 4 + e] = l * 100 + r * 10 + e;
    k<<<1, 64>>>(in, out);
    hipDeviceSynchronize();
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) {
        const int src_lane = (l & ~3) + i, src_reg = l & 3;
        if (((unsigned*)out)[(l * 4 + i) * 4 + e] != (unsigned)(src_lane * 100 + src_reg * 10 + e)) ++bad;
    }
    printf("quad transpose: %d mismatches\n", bad);
    return bad != 0;
}
