// Shared device helpers for the MSF-WSI pre-train step kernels (gfx950 / CDNA4 only).
// Wavefront = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// Raise a kernel's dynamic-LDS limit above the default 64 KiB -- ONCE per kernel (function pointer), not on every launch: the
// hot path issues hundreds of such launches per step from host threads that already feed three streams.  A lock-free set of
// the kernels already raised; two threads racing on a kernel's first launch may both set the attribute (harmless).
inline int msfwsi_raise_lds(const void* kern, int bytes) {
    static std::atomic<uintptr_t> seen[1024];
    const uintptr_t key = reinterpret_cast<uintptr_t>(kern);
    unsigned h = (unsigned)((key >> 4) * 2654435761u) & 1023u;
    for (int probe = 0; probe < 1024; ++probe) {
        const uintptr_t cur = seen[(h + probe) & 1023u].load(std::memory_order_acquire);
        if (cur == key) return 0;
        if (cur == 0) break;
    }
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    for (int probe = 0; probe < 1024; ++probe) {
        uintptr_t expected = 0;
        if (seen[(h + probe) & 1023u].compare_exchange_strong(expected, key, std::memory_order_acq_rel) || expected == key) break;
    }
    return 0;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MSFWSI_DT_F32 0
#define MSFWSI_DT_BF16 1
#define MSFWSI_DT_F16 2

// return codes of the C ABI: 0 ok, <0 invalid argument class, >0 hipError_t
#define MSFWSI_OK 0
#define MSFWSI_EINVAL (-1)
#define MSFWSI_EUNSUPPORTED (-2)

#define MSFWSI_CHECK_ARG(cond) \
    do {                       \
        if (!(cond)) return MSFWSI_EINVAL; \
    } while (0)

// Process-global tuning switches (msfwsi_set_tuning / msfwsi_get_tuning): read by every launch, written from any host thread
// (tests, A/B drivers) -- relaxed atomics, no ordering implied between a switch and launches already enqueued.
struct msfwsi_tunable {
    std::atomic<long> v;
    constexpr msfwsi_tunable(long init) : v(init) {}
    operator long() const { return v.load(std::memory_order_relaxed); }
    msfwsi_tunable& operator=(long x) {
        v.store(x, std::memory_order_relaxed);
        return *this;
    }
};

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
template <>
struct ElemTraits<_Float16> {
    static constexpr int VEC = 8;
    static constexpr int BK = 32;
};

static inline bool msfwsi_dtype_ok(int dt) { return dt == MSFWSI_DT_F32 || dt == MSFWSI_DT_BF16 || dt == MSFWSI_DT_F16; }
static inline int msfwsi_vec_of(int dt) { return dt == MSFWSI_DT_F32 ? 4 : 8; }

// run `...` with T bound to the storage type selected by `dtype` (the statement may contain commas)
#define MSFWSI_WITH_T(dtype, ...)                \
    do {                                         \
        if ((dtype) == MSFWSI_DT_BF16) {         \
            typedef __bf16 T;                    \
            __VA_ARGS__;                         \
        } else if ((dtype) == MSFWSI_DT_F16) {   \
            typedef _Float16 T;                  \
            __VA_ARGS__;                         \
        } else {                                 \
            typedef float T;                     \
            __VA_ARGS__;                         \
        }                                        \
    } while (0)

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

template <>
__device__ __forceinline__ void unpack16<_Float16>(const uint4& v, float* f) {
    const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
}

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    return (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ unsigned short float_to_f16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (_Float16)f);
}
__device__ __forceinline__ unsigned pack2_f16(float lo, float hi) {
    return (unsigned)float_to_f16_bits(lo) | ((unsigned)float_to_f16_bits(hi) << 16);
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

template <>
__device__ __forceinline__ uint4 pack16<_Float16>(const float* f) {
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)f[e];
    return __builtin_bit_cast(uint4, h);
}

// Gate-byte tensor of a [M][cpr chunks] activation (one byte per 16-byte chunk, msfwsi_conv_fwd_post's gate_out): byte offset
// of (pixel m, chunk c).  BLOCKED when a row holds whole dwords of gate bytes (cpr % 4 == 0: every ResNet width):
// [m / 128][c / 4][m % 128][c % 4] -- the four gate bytes of one pixel's 32-channel block form a dword, and the dwords of 128
// consecutive pixels are contiguous (512 bytes).  A wave of the panel kernels owns a 32-channel block of 128 pixels: its
// gate traffic is then 64 contiguous bytes per 16 pixels instead of 16 partial-line requests of 4 bytes each (which cost
// a 14x14 forward launch 0.29 of 1.24 ms: the L2 handles requests, not bytes).  Otherwise linear [m][c].
// The tensor holds msfwsi_gate_numel(M, cpr) bytes (rows padded to 128 in the blocked form).
__host__ __device__ __forceinline__ long gate_off(long m, int c, int cpr) {
    if (cpr & 3) return m * cpr + c;
    return ((m >> 7) * (long)(cpr >> 2) + (c >> 2)) * 512 + (m & 127) * 4 + (c & 3);
}

// ReLU gate of one STORED 16-byte chunk: bit e = (element e > 0), taken from the packed value so that the bits equal
// the sign test of the tensor in memory bit for bit (an fp32 value that rounds to zero in the storage type is closed)
template <typename T>
__device__ __forceinline__ unsigned gate_bits_of(const uint4& v);
template <>
__device__ __forceinline__ unsigned gate_bits_of<float>(const uint4& v) {
    return (__uint_as_float(v.x) > 0.f ? 1u : 0u) | (__uint_as_float(v.y) > 0.f ? 2u : 0u) |
           (__uint_as_float(v.z) > 0.f ? 4u : 0u) | (__uint_as_float(v.w) > 0.f ? 8u : 0u);
}
__device__ __forceinline__ unsigned gate_bits16(const uint4& v) {
    // a 16-bit float is > 0 exactly when its bit pattern, read as a signed 16-bit integer, is > 0 (NaNs aside)
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned b = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b |= ((short)(w[i] & 0xffffu) > 0 ? 1u : 0u) << (2 * i);
        b |= ((int)w[i] > 0xffff ? 1u : 0u) << (2 * i + 1);
    }
    return b;
}
template <>
__device__ __forceinline__ unsigned gate_bits_of<__bf16>(const uint4& v) {
    return gate_bits16(v);
}
template <>
__device__ __forceinline__ unsigned gate_bits_of<_Float16>(const uint4& v) {
    return gate_bits16(v);
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

template <>
__device__ __forceinline__ float round_to<_Float16>(float f) {
    return (float)(_Float16)f;
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
template <>
__device__ __forceinline__ float load_elem<_Float16>(const _Float16* p, size_t i) {
    return (float)p[i];
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

template <>
__device__ __forceinline__ void store_elem<_Float16>(_Float16* p, size_t i, float v) {
    p[i] = (_Float16)v;
}

// MFMA fragment types and the 32x32 step per storage type (16-bit: 32x32x16, fp32: 4 x 32x32x2 exact fp32)
template <typename T>
struct MmaFrag;
template <>
struct MmaFrag<float> {
    typedef f32x4 type;
};
template <>
struct MmaFrag<__bf16> {
    typedef bf16x8 type;
};
template <>
struct MmaFrag<_Float16> {
    typedef f16x8 type;
};
template <typename T>
__device__ __forceinline__ void mma32(f32x16& acc, const typename MmaFrag<T>::type& a, const typename MmaFrag<T>::type& b);
template <>
__device__ __forceinline__ void mma32<__bf16>(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma32<_Float16>(f32x16& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma32<float>(f32x16& acc, const f32x4& a, const f32x4& b) {
    // lane half h supplies k = 4h+e to step e on BOTH operands: a consistent k permutation
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

// exact unsigned division by a launch-time constant (Granlund-Montgomery): 4 VALU instead of ~25
struct FastDiv {
    unsigned mul, sh1, sh2;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned L = 0;
    while ((1ull << L) < d) ++L;  // ceil(log2 d)
    f.mul = (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    f.sh1 = L < 1 ? L : 1;
    f.sh2 = L > 0 ? L - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
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

// Lane index of the calling wave, RE-DERIVED at the point of the call (v_mbcnt) instead of kept in a register since
// kernel entry.  The persistent kernels need tid / lane only before and after a main loop that uses every VGPR the
// launch bounds allow; hipcc parked those two values in scratch across the loop (12-20 bytes per lane, -Rpass-analysis=
// kernel-resource-usage) although one instruction pair re-creates them -- `asm volatile` keeps it from merging this
// computation with the one at kernel entry.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// ---------------------------------------------------------------------------------------------------------------
// 16 bytes per lane global -> LDS through a buffer descriptor without touching VGPRs (`buffer_load_dwordx4 ... lds`):
// per-lane byte offset `voff` + wave-uniform byte offset `soff`, out-of-range -> zeros; lds_wave_base is wave-uniform,
// lane l lands at base + 16 l.
//
// Issued as INLINE ASM, not through __builtin_amdgcn_raw_ptr_buffer_load_lds: with the builtin hipcc (ROCm 7.2) knows of
// the pending LDS write and puts an `s_waitcnt vmcnt(0)` in front of the first `ds_read_b64_tr_b16` after every barrier
// (it does not for plain ds_read_b128) -- in every kernel whose fragments come from the transposed read (all input-
// gradient and weight-gradient kernels) the hand-counted `s_waitcnt vmcnt(N)` pipeline was thereby drained each slab:
// the DMA of slab kt+1 never overlapped the MFMAs of slab kt inside a wave.  The asm statement is opaque to that pass;
// completion is tracked by the kernels' own counted waits (they already were).  M0 (the LDS base of the DMA) is written
// in the same statement that uses it; `s_nop 4` covers an operand fresh from v_readfirstlane (VALU-written SGPR ->
// VMEM: 5 wait states), `s_nop 0` the M0 write -> LDS-DMA wait state.
// RULE: a kernel stages EITHER through this asm path OR through the compiler-tracked builtin
// (__builtin_amdgcn_global_load_lds: wdma16 / dma16 of the register-address kernels), never both in one instance: the
// asm loads are invisible to SIInsertWaitcnts, so a mixed kernel's compiler-inserted waits would under-count.  The
// -DMSFWSI_ASM_DMA=0 build (tools/build_variant.sh asmdma0 "-DMSFWSI_ASM_DMA=0", then MSFWSI_LIB=ab/libmsfwsi_asmdma0.so) is the A/B
// correctness reference of the hand-counted waits: the kernel and production test files pass on it unchanged (round 4).
// ---------------------------------------------------------------------------------------------------------------
#ifndef MSFWSI_ASM_DMA
#define MSFWSI_ASM_DMA 1
#endif
__device__ __forceinline__ void lds_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff) {
#if MSFWSI_ASM_DMA
    const unsigned lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");  // M0 is clobbered: hipcc must not keep a live value in it across the statement
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
#endif
}

