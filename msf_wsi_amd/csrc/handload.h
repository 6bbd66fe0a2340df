// Hand-counted vector-memory loads shared by the activation-stationary kernels (panel.hip, img3x3.hip).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ uint4 as_uint4(const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); }

// Global loads of the block loop, in two forms.  HAND = true (whole panels: every workgroup but possibly the last): inline
// asm, invisible to hipcc's wait-count pass, completion tracked by hand-counted `s_waitcnt vmcnt(N)` statements that name the
// destination "+v" (so that no consumer is scheduled above the wait; form (ii) of the HIP guide's inline-asm section).
// Why: the loop is software-pipelined ACROSS iterations (operands of block j+1 are requested during block j), and for
// loads whose results cross the loop's back edge hipcc falls back to `vmcnt(<small>)` at the first use -- every block
// began by draining the queue, i.e. by waiting for the identity loads issued a moment earlier (1.0 -> 1.25 ms per launch).
// HAND = false (the ragged last panel): plain loads, hipcc's own waits.
template <bool HAND>
__device__ __forceinline__ void pl_load16(u32x4& dst, const void* sbase, unsigned voff) {
    if constexpr (HAND) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    } else {
        dst = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(sbase) + voff);
    }
}
template <bool HAND>
__device__ __forceinline__ void pl_load4(unsigned& dst, const void* sbase, unsigned voff) {
    if constexpr (HAND) {
        asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    } else {
        dst = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(sbase) + voff);
    }
}
// at most N vector-memory operations younger than the one that fills `r` may still be outstanding
template <bool HAND, int N>
__device__ __forceinline__ void pl_wait(u32x4& r) {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}
template <bool HAND, int N>
__device__ __forceinline__ void pl_wait(u32x4& r, unsigned& r2) {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r), "+v"(r2) : "n"(N) : "memory");
}


template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d);
template <>
__device__ __forceinline__ uint2 pack4<__bf16>(float a, float b, float c, float d) {
    return make_uint2(pack2_bf16(a, b), pack2_bf16(c, d));
}
template <>
__device__ __forceinline__ uint2 pack4<_Float16>(float a, float b, float c, float d) {
    return make_uint2(pack2_f16(a, b), pack2_f16(c, d));
}


// wait for EVERYTHING, naming one more in-flight destination: a chain of these after a loop keeps every register that an asm
// load may still write out of hipcc's hands until the data has landed (a register whose value the program no longer needs
// is otherwise free for re-use the moment the loop ends -- and the late load then overwrites whatever was put there)
template <bool HAND>
__device__ __forceinline__ void pl_drain(u32x4& r) {
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r) : : "memory");
}
template <bool HAND>
__device__ __forceinline__ void pl_drain(unsigned& r) {
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r) : : "memory");
}
