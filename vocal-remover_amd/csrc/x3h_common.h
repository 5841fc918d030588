// Helpers shared by the two fp16-split direct 3x3 kernels (conv_x3h.hip: four waves, every wave does everything in lockstep;
// conv_x3pp.hip: eight waves in two groups that alternate between multiplying and loading / splitting).
#pragma once
#include <type_traits>
#include <utility>

#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

// The pixel loads are inline asm and their waits are placed by hand: hipcc's wait-count pass loses the issue order of loads that
// cross a loop back edge / uniform branches and then waits for (nearly) everything, i.e. also for the loads issued for the chunk
// after next -- the prefetch depth the register sets pay for.
__device__ __forceinline__ float x3h_load(i32x4 rsrc, int voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
// s_waitcnt vmcnt(N) that the uses of the eight registers cannot be scheduled across
template <int N>
__device__ __forceinline__ void x3h_wait8(float (&r)[8]) {
    // (the comment names the registers in the .s file: tools/asm_inflight_audit2.py checks that nothing touches them between the load
    // and this wait)
    asm volatile("s_waitcnt vmcnt(%8) ; landed %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N) : "memory");
}

typedef _Float16 vr_f16x8 __attribute__((ext_vector_type(8)));
typedef float vr_f32x4h __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_f16x16(vr_f16x8 a, vr_f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// (x0, x1) * s -> packed fp16 pairs of the two planes: p1 = rne(x s), p2 = rne(x s - p1); each is ONE fp32 fma rounded once to fp16
__device__ __forceinline__ void split2h_pair(float x0, float x1, float s, int& p1, int& p2) {
    int a, b;                                    // (mixlo leaves the other half of its destination alone; mixhi then fills it: "=v" first)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(a) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(b) : "v"(x0), "v"(s), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(b) : "v"(x1), "v"(s), "v"(a));
    p1 = a; p2 = b;
}
// max(m, |r[0..7]|)   (inline asm: hipcc canonicalises fabsf() with a v_max of its own per value)
__device__ __forceinline__ float x3h_absmax8(float m, const float (&r)[8]) {
#pragma unroll
    for (int cl = 0; cl < 8; cl += 2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(r[cl]), "v"(r[cl + 1]));
    return m;
}
// 2^e as a float, e in [-126, 127]
__device__ __forceinline__ float x3h_pow2(int e) { return __int_as_float((e + 127) << 23); }


// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a TYPE.  `#pragma unroll` gives up above
// -pragma-unroll-threshold, and a loop left rolled indexes its register arrays with a run-time value -- hipcc then keeps them in scratch.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

}  // namespace vr
