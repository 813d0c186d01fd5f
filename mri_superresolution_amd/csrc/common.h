// Shared device/host helpers for libmrisr (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mrisr.h"

// ---------------------------------------------------------------- errors (thread-local)
extern thread_local char g_mrisr_err[512];
#define MRISR_FAIL(code, ...)                                   \
    do {                                                        \
        snprintf(g_mrisr_err, sizeof(g_mrisr_err), __VA_ARGS__); \
        return (code);                                          \
    } while (0)
#define MRISR_CHECK_LAUNCH(name)                                                          \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) MRISR_FAIL(MRISR_E_HIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------- types
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;      // IEEE half (MRISR_F16): the reference's autocast dtype (scripts/train.py:303-306), needs loss scaling
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LRELU_SLOPE 0.2f

template <typename T> struct TypeTraits;
template <> struct TypeTraits<float> {
    static constexpr int kDtype = MRISR_F32;
    static constexpr int kVec = 4;  // elements per 16-byte vector
};
template <> struct TypeTraits<bf16_t> {
    static constexpr int kDtype = MRISR_BF16;
    static constexpr int kVec = 8;
};
template <> struct TypeTraits<f16_t> {
    static constexpr int kDtype = MRISR_F16;
    static constexpr int kVec = 8;
};
// elements per 16-byte vector of a storage dtype
static inline int mrisr_vec(int dtype) { return dtype == MRISR_F32 ? 4 : 8; }
static inline bool mrisr_dtype_ok(int dtype) { return dtype == MRISR_F32 || dtype == MRISR_BF16 || dtype == MRISR_F16; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// A 16-byte register vector of T, with element access as float.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
    __device__ __forceinline__ void zero() { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
};
template <> struct Vec16<bf16_t> {
    bf16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
    __device__ __forceinline__ void zero() {
        v = bf16x8{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f,
                   (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    }
};
template <> struct Vec16<f16_t> {
    f16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (f16_t)x; }
    __device__ __forceinline__ void zero() {
        v = f16x8{(f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f};
    }
};
template <typename T> __device__ __forceinline__ Vec16<T> load_vec16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void store_vec16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// Explicitly GLOBAL-address-space accesses.  Pointers that went through an opaque inline asm (pin_params) lose
// their inferred address space and hipcc falls back to FLAT instructions, which count on lgkmcnt as well as
// vmcnt: every LDS wait then also waits for the outstanding global prefetch loads / epilogue stores.
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ Vec16<T> gload_vec16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const GLOBAL_AS decltype(r.v)*>((const GLOBAL_AS char*)p);
    return r;
}
template <typename V> __device__ __forceinline__ V gload(const void* p) {
    return *reinterpret_cast<const GLOBAL_AS V*>((const GLOBAL_AS char*)p);
}
template <typename V> __device__ __forceinline__ void gstore(void* p, const V& v) {
    *reinterpret_cast<GLOBAL_AS V*>((GLOBAL_AS char*)p) = v;
}

__device__ __forceinline__ float lrelu(float x) { return fmaxf(x, LRELU_SLOPE * x); }   // slope < 1

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {  // over the 32 lanes of each half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
    __hip_atomic_fetch_add((GLOBAL_AS double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    __hip_atomic_fetch_add((GLOBAL_AS float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bilinear x2 align_corners=True source coordinate, as aten's area_pixel_compute_source_index:
// scale = (in-1)/(out-1) in float, src = scale*dst.
// (no fma contraction: aten rounds src to fp32 before taking the fraction; scale * dst - i0 as ONE fma moves the
// weight by up to an ulp of src, 1.5e-5 at row 255)
__device__ __forceinline__ void up2_coord(int dst, int in_size, int& i0, int& i1, float& w1) {
#pragma clang fp contract(off)
    const int out_size = 2 * in_size;
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = src - (float)i0;
}

// slot of a statistics buffer for this workgroup (spreads same-address fp64 atomics)
__device__ __forceinline__ size_t stat_slot_off(int N, int groups) {
    return (size_t)((blockIdx.x + blockIdx.y) & (MRISR_STAT_SLOTS - 1)) * N * groups * 2;
}
// same, from a logical workgroup index (kernels that re-order their workgroups: neighbours in the LOGICAL order work on
// the same image at the same time and must not share a slot)
__device__ __forceinline__ size_t stat_slot_off_id(int id, int N, int groups) {
    return (size_t)(id & (MRISR_STAT_SLOTS - 1)) * N * groups * 2;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
