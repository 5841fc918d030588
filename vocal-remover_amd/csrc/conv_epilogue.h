// Shared store epilogue of the direct conv kernels (conv_x3.hip, conv_dma.hip): bias, (eval) folded BatchNorm + activation, up to
// three destination segments of the virtual concat, optional accumulation (data-gradient launches add into the gradient of every
// source).  Accumulator layout: the 32x32 MFMA tile -- register r of lane (khalf, l31) is cout (r & 3) + 8 (r >> 2) + 4 khalf of the
// block, pixel l31 of the wave's pixel group ni.
//
// Two things made the old per-element form slow (round 3, tools/x3_proto.hip dbg 4: 5-14 % of a forward launch, more than the
// whole multiply phase of a short-K data gradient):
//   * bias / scale / shift were global loads inside the row loop, each followed by vmcnt(0): one memory round trip per cout.
//     They are loaded once per workgroup at kernel start and parked in LDS (E[3][MT]).
//   * every access sat behind its own guard; hipcc then puts a vmcnt(0) in front of each -- which on gfx9 also waits for the
//     previous STORE -- so the accumulating form was one load round trip plus one store round trip per element.
//     A group of sixteen couts that lies inside the image, inside Cout and inside ONE destination (nearly all of them) now runs
//     straight-line: all old values loaded first, then all stores; the guarded per-element form only handles the edges.
#pragma once
#include <type_traits>
#include <utility>

#include "conv_stage.h"

namespace vr {

// What the store phase needs from ConvArgs travels as individual scalar PARAMETERS, read from `a` in the kernel body.  Handing a
// struct (`const ConvArgs&`, or a by-value copy of the fields) to a helper gives hipcc a stack object, and it turns a select between
// two of its fields into a select of ADDRESSES -- the object then stays in scratch (584 B per thread for ConvArgs; seen in conv_x3
// and conv_dma, with either spelling: a.dst[seg].x or seg == 0 ? a.dst[0].x : ...).  Scalars are SSA values: nothing to address.
#define VR_EPI_PARAMS                                                                                                              \
    float *p0, float *p1, float *p2, long long sN0, long long sN1, long long sN2, long long sC0, long long sC1, long long sC2,        \
        long long sH0, long long sH1, long long sH2, int wshift0, int wshift1, int wshift2, int accumulate0, int accumulate1,       \
        int accumulate2, int e_d1, int e_d2, int e_Cout, int e_Hout, int e_Wout, float eslope
#define VR_EPI_FWD                                                                                                                 \
    p0, p1, p2, sN0, sN1, sN2, sC0, sC1, sC2, sH0, sH1, sH2, wshift0, wshift1, wshift2, accumulate0, accumulate1, accumulate2, e_d1, \
        e_d2, e_Cout, e_Hout, e_Wout, eslope
#define VR_EPI_ARGS(a)                                                                                                             \
    (a).dst[0].p, (a).dst[1].p, (a).dst[2].p, (a).dst[0].sN, (a).dst[1].sN, (a).dst[2].sN, (a).dst[0].sC, (a).dst[1].sC,            \
        (a).dst[2].sC, (a).dst[0].sH, (a).dst[1].sH, (a).dst[2].sH, (a).dst[0].wshift, (a).dst[1].wshift, (a).dst[2].wshift,      \
        (a).dst[0].accumulate, (a).dst[1].accumulate, (a).dst[2].accumulate, (a).d1, (a).d2, (a).Cout, (a).Hout, (a).Wout,       \
        ((a).epi ? (a).epi_slope : 1.f)
#define VR_DST_FIELD(seg, f) ((seg) == 0 ? f##0 : ((seg) == 1 ? f##1 : f##2))

// the three constants of cout co0 + (tid % MT), to be parked with epi_park() once they have landed
template <int MT>
__device__ __forceinline__ void epi_fetch(const float* bias, const float* epi, int Cout, int co0, int tid, float (&ecv)[3]) {
    const int ec = co0 + (tid & (MT - 1));
    const int ecc = ec < Cout ? ec : Cout - 1;
    ecv[0] = bias ? bias[ecc] : 0.f;
    ecv[1] = epi ? epi[2 * ecc] : 1.f;
    ecv[2] = epi ? epi[2 * ecc + 1] : 0.f;
}
template <int MT>
__device__ __forceinline__ void epi_park(float* E, int tid, const float (&ecv)[3]) {
    if (tid < MT) { E[tid] = ecv[0]; E[MT + tid] = ecv[1]; E[2 * MT + tid] = ecv[2]; }
}

// One group of G accumulator rows (IT = mi * (16 / G) + rg / G); a struct template instead of a loop body: `#pragma unroll` gives up above
// -pragma-unroll-threshold, and a loop left rolled here indexes the accumulators with a run-time value -- hipcc then keeps ALL of
// them in scratch (seen on the 128-cout conv_dma tilings).
template <int MT, int WM, int WN, int G, int IT>
struct EpiGroup {
    static __device__ __forceinline__ void run(VR_EPI_PARAMS, f32x16 (&acc)[WM][WN], const float* E, int n, int co0, int khalf, bool tile_in,
                                               const int (&hon)[WN], const int (&won)[WN]) {
        {
            constexpr int mi = IT / (16 / G), rg = (IT % (16 / G)) * G;
            // rows rg .. rg+G-1 of lane half khalf are couts cg0 + (j & 3) + 8 (j >> 2) + 4 khalf: 2 G consecutive couts over both halves
            const int cg0 = co0 + mi * 32 + 2 * rg;                    // (wave-uniform)
            const int seg0 = (cg0 >= e_d1) + (cg0 >= e_d2), seg1 = (cg0 + 2 * G - 1 >= e_d1) + (cg0 + 2 * G - 1 >= e_d2);
            float* const Dp = VR_DST_FIELD(seg0, p);
            float eb[G], esc[G], esh[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int r = rg + j;
                const int cl = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                eb[j] = E[cl]; esc[j] = E[MT + cl]; esh[j] = E[2 * MT + cl];
            }
            // full wait before the first consumer of the LDS reads (hardware fact 5)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < G; ++j) asm volatile("" : "+v"(eb[j]), "+v"(esc[j]), "+v"(esh[j]));
            if (tile_in && cg0 + 2 * G <= e_Cout && seg0 == seg1 && Dp != nullptr) {
                const int cseg = seg0 == 0 ? 0 : (seg0 == 1 ? e_d1 : e_d2);
                const long long sN = VR_DST_FIELD(seg0, sN);
                const long long sC = VR_DST_FIELD(seg0, sC);
                const long long sH = VR_DST_FIELD(seg0, sH);
                const int ws = VR_DST_FIELD(seg0, wshift);
                const int dacc = VR_DST_FIELD(seg0, accumulate);
                float* qb = Dp + (long long)n * sN + (long long)(cg0 - cseg + 4 * khalf) * sC;
                long long offn[WN];
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) offn[ni] = (long long)hon[ni] * sH + ((long long)won[ni] << ws);
                float old[G][WN];
                if (dacc) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni) old[j][ni] = qb[((j & 3) + 8 * (j >> 2)) * sC + offn[ni]];
                } else {
#pragma unroll
                    for (int j = 0; j < G; ++j)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni) old[j][ni] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) {
                        const float v = acc[mi][ni][rg + j] + eb[j];
                        acc[mi][ni][rg + j] = v;
                        qb[((j & 3) + 8 * (j >> 2)) * sC + offn[ni]] = act_apply(fmaf(v, esc[j], esh[j]), eslope) + old[j][ni];
                    }
            } else {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int r = rg + j;
                    const int co = co0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int seg = (co >= e_d1) + (co >= e_d2);
                    const int cod = co - (seg == 0 ? 0 : (seg == 1 ? e_d1 : e_d2));
                    float* dp = VR_DST_FIELD(seg, p);
                    const long long dN = VR_DST_FIELD(seg, sN);
                    const long long dC = VR_DST_FIELD(seg, sC);
                    const long long dH = VR_DST_FIELD(seg, sH);
                    const int dacc = VR_DST_FIELD(seg, accumulate);
                    const int dws = VR_DST_FIELD(seg, wshift);
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) {
                        const float v = acc[mi][ni][r] + eb[j];
                        acc[mi][ni][r] = v;
                        if (co < e_Cout && hon[ni] < e_Hout && won[ni] < e_Wout && dp) {
                            float* q = dp + (long long)n * dN + (long long)cod * dC + (long long)hon[ni] * dH + ((long long)won[ni] << dws);
                            const float y = act_apply(fmaf(v, esc[j], esh[j]), eslope);
                            *q = dacc ? *q + y : y;
                        }
                    }
                }
            }
        }
    }
};

template <int MT, int WM, int WN, int G, int... IT>
__device__ __forceinline__ void epi_groups(std::integer_sequence<int, IT...>, VR_EPI_PARAMS, f32x16 (&acc)[WM][WN], const float* E, int n, int co0,
                                           int khalf, bool tile_in, const int (&hon)[WN], const int (&won)[WN]) {
    (EpiGroup<MT, WM, WN, G, IT>::run(VR_EPI_FWD, acc, E, n, co0, khalf, tile_in, hon, won), ...);
}

// hon / won: output row / column of this lane's pixel in each of the wave's WN pixel groups; tile_in: the whole tile is inside the
// image (wave-uniform).  acc keeps conv + bias afterwards (the BatchNorm statistics of the training forward are taken on it).
// G accumulator rows per group (8: sixteen couts; 4: eight couts, for the tilings whose register budget is tight).
template <int MT, int WM, int WN, int G = 8>
__device__ __forceinline__ void epi_store(VR_EPI_PARAMS, f32x16 (&acc)[WM][WN], const float* E, int n, int co0, int khalf, bool tile_in,
                                          const int (&hon)[WN], const int (&won)[WN]) {
    static_assert(G == 4 || G == 8, "group");
    epi_groups<MT, WM, WN, G>(std::make_integer_sequence<int, WM * (16 / G)>{}, VR_EPI_FWD, acc, E, n, co0, khalf, tile_in, hon, won);
}

}  // namespace vr
