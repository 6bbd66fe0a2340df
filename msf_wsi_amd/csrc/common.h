// Shared device helpers for the MSF-WSI pre-train step kernels (gfx950 / CDNA4 only).
// Wavefront = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MSFWSI_DT_F32 0
#define MSFWSI_DT_BF16 1

// return codes of the C ABI: 0 ok, <0 invalid argument class, >0 hipError_t
#define MSFWSI_OK 0
#define MSFWSI_EINVAL (-1)
#define MSFWSI_EUNSUPPORTED (-2)

#define MSFWSI_CHECK_ARG(cond) \
    do {                       \
        if (!(cond)) return MSFWSI_EINVAL; \
    } while (0)

static inline int msfwsi_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSFWSI_OK : (int)e;
}

// ---------------------------------------------------------------------------------------------
// element traits: one 16-byte chunk holds VEC elements; BK = k-depth of one LDS stage (64 bytes)
// ---------------------------------------------------------------------------------------------
template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
    static constexpr int VEC = 4;
    static constexpr int BK = 16;
};
template <>
struct ElemTraits<__bf16> {
    static constexpr int VEC = 8;
    static constexpr int BK = 32;
};

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short float_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}

template <typename T>
__device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x);
    f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z);
    f[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void unpack16<__bf16>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16);
    f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16);
    f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16);
    f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16);
    f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    return (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
}

template <typename T>
__device__ __forceinline__ uint4 pack16(const float* f);
template <>
__device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
}
template <>
__device__ __forceinline__ uint4 pack16<__bf16>(const float* f) {
    return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]),
                      pack2_bf16(f[6], f[7]));
}

// value of one element after a round trip through the storage type T
template <typename T>
__device__ __forceinline__ float round_to(float f);
template <>
__device__ __forceinline__ float round_to<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ float round_to<__bf16>(float f) {
    return bf16_bits_to_float(float_to_bf16_bits(f));
}

template <typename T>
__device__ __forceinline__ float load_elem(const T* p, size_t i);
template <>
__device__ __forceinline__ float load_elem<float>(const float* p, size_t i) {
    return p[i];
}
template <>
__device__ __forceinline__ float load_elem<__bf16>(const __bf16* p, size_t i) {
    return bf16_bits_to_float(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void store_elem(T* p, size_t i, float v);
template <>
__device__ __forceinline__ void store_elem<float>(float* p, size_t i, float v) {
    p[i] = v;
}
template <>
__device__ __forceinline__ void store_elem<__bf16>(__bf16* p, size_t i, float v) {
    reinterpret_cast<unsigned short*>(p)[i] = float_to_bf16_bits(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// XCD-aware workgroup renumbering (MI355X: 8 XCDs, each with a private L2; consecutive workgroup ids are dealt
// round-robin over the XCDs).  Maps the hardware id to a logical id such that each XCD works on one CONTIGUOUS
// range of logical ids, so that workgroups sharing operand rows (the n-tiles of one m-tile, the tiles of one pixel
// split) fill the same L2 instead of eight.  Bijective for every grid size; affects speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned hw_id, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = hw_id & 7u, idx = hw_id >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
