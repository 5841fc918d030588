// LDS-DMA helpers shared by conv_dma.hip and conv_wino.hip (gfx950 `buffer_load_dwordx4 ... lds`).
#pragma once
#include <hip/hip_runtime.h>

namespace vr {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (base, num_records = bytes); an offset >= bytes reads as zero.
__device__ __forceinline__ i32x4 make_rsrc(const float* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;      // readfirstlane: the descriptor must live in SGPRs; its inputs are wave-uniform by construction
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffull));
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffull));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// A descriptor fresh from v_readfirstlane must sit 5 wait states before a buffer_* instruction reads it, and hipcc neither sees
// inside an asm string nor pads in front of one: settle it explicitly (the operand pins the readfirstlanes in front of the nop).
__device__ __forceinline__ void settle_rsrc(i32x4& r) { asm volatile("s_nop 4" : "+s"(r)); }

// One 64-lane LDS-DMA: lane l copies 16 B from rsrc.base + voff[l] to LDS byte lds_base + 16*l.
// Inline asm on purpose: the compiler does not track it, so it never inserts a vmcnt(0) in front of
// unrelated LDS reads; the caller waits with dma_wait() before the data is consumed.
__device__ __forceinline__ void dma16(unsigned lds_base, unsigned voff, i32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

// The same with a scalar byte offset added to every lane's address (not part of the range check: gfx9 compares the vector offset).
__device__ __forceinline__ void dma16s(unsigned lds_base, unsigned voff, i32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Dword form: lane l copies 4 B to LDS byte lds_base + 4*l (any 4-B aligned base: odd LDS pitches stay possible).
__device__ __forceinline__ void dma4(unsigned lds_base, unsigned voff, i32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void dma_wait_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void lds_barrier() {      // LDS traffic only (no DMA outstanding)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace vr
