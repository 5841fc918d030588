// Epilogue shared by the Winograd kernels (conv_wino.hip, conv_wino6.hip): the accumulators acc[fi][mi][ni] of wave w hold
// frequencies 2w + fi of a (MT couts) x (64 Winograd tiles = 8 x 32 pixels) block; the 16 frequencies of a (cout, tile) pair
// meet in LDS (32 couts x 32 tiles per pass), A^T M A, bias / folded BatchNorm / activation, stores, BatchNorm partial sums.
#pragma once
#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

constexpr int kWinoMP = 33;                                            // exchange pitch
__host__ __device__ constexpr int wino_epilogue_floats(int MT) { return 16 * 32 * kWinoMP + 2 * MT; }

template <int MT>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, f32x16 (&acc)[2][MT / 32][2], float* smem, int tid, int wave,
                                              int khalf, int l31, int n, int h0, int w0, int co0, int pt) {
    constexpr int WM = MT / 32, MP = kWinoMP;
    // ---------------- epilogue: gather the 16 frequencies per (cout, tile) through LDS, A^T M A ------------
    if (a.dbg == 4) return;                                            // (ablation: no epilogue, no stores)
    float* Mx = smem;                                                  // [16][32][MP]
    float* stat = smem + 16 * 32 * MP;                                 // [MT][2] BatchNorm partial sums (training)
    if (a.part && tid < 2 * MT) stat[tid] = 0.f;                       // (ordered by the first pass's barriers)
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (mi + ni > 0) lds_barrier();                            // previous pass has been read
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    Mx[((2 * wave + fi) * 32 + col) * MP + l31] = acc[fi][mi][ni][r];
                }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = tid + 512 * j;
                const int col = p >> 5, tl = p & 31;
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = Mx[(f * 32 + col) * MP + tl];
                float s0[4], s1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = m[c] + m[4 + c] + m[8 + c];
                    s1[c] = m[4 + c] - m[8 + c] - m[12 + c];
                }
                float y[2][2];
                y[0][0] = s0[0] + s0[1] + s0[2];
                y[0][1] = s0[1] - s0[2] - s0[3];
                y[1][0] = s1[0] + s1[1] + s1[2];
                y[1][1] = s1[1] - s1[2] - s1[3];
                const int co = co0 + mi * 32 + col;
                const int T = ni * 32 + tl;
                const int ho = h0 + 2 * (T >> 4), wo = w0 + 2 * (T & 15);
                const int cc = co < a.Cout ? co : a.Cout - 1;
                const float b = a.bias ? a.bias[cc] : 0.f;
                float esc = 1.f, esh = 0.f, eslope = 1.f;
                if (a.epi) { esc = a.epi[2 * cc]; esh = a.epi[2 * cc + 1]; eslope = a.epi_slope; }
                // destination segment of this cout (the data gradient of a virtual concat has up to three)
                const int seg = (co >= a.d1) + (co >= a.d2);
                const int cod = co - (seg == 0 ? 0 : (seg == 1 ? a.d1 : a.d2));
                float* dp = seg == 0 ? a.dst[0].p : (seg == 1 ? a.dst[1].p : a.dst[2].p);
                const long long dN = seg == 0 ? a.dst[0].sN : (seg == 1 ? a.dst[1].sN : a.dst[2].sN);
                const long long dC = seg == 0 ? a.dst[0].sC : (seg == 1 ? a.dst[1].sC : a.dst[2].sC);
                const long long dH = seg == 0 ? a.dst[0].sH : (seg == 1 ? a.dst[1].sH : a.dst[2].sH);
                const int dacc = seg == 0 ? a.dst[0].accumulate : (seg == 1 ? a.dst[1].accumulate : a.dst[2].accumulate);
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int dr = 0; dr < 2; ++dr) {
#pragma unroll
                    for (int dc = 0; dc < 2; ++dc) {
                        const float v = y[dr][dc] + b;
                        const bool in = ho + dr < a.Hout && wo + dc < a.Wout;
                        if (in) { t1 += v; t2 = fmaf(v, v, t2); }
                        if (in && co < a.Cout && dp) {
                            float* q = dp + (long long)n * dN + (long long)cod * dC + (long long)(ho + dr) * dH + wo + dc;
                            const float o = act_apply(fmaf(v, esc, esh), eslope);
                            *q = dacc ? *q + o : o;
                        }
                    }
                }
                if (a.part) {                                           // sum over the 32 tiles of this half-wave
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        t1 += __shfl_xor(t1, off, 64);
                        t2 += __shfl_xor(t2, off, 64);
                    }
                    if (tl == 0) {                                      // (cout, pass) pairs are unique: no race
                        stat[(mi * 32 + col) * 2 + 0] += t1;
                        stat[(mi * 32 + col) * 2 + 1] += t2;
                    }
                }
            }
        }
    }
    if (a.part) {
        lds_barrier();
        if (tid < MT && co0 + tid < a.Cout) {
            a.part[((long long)pt * a.Cout + co0 + tid) * 2 + 0] = stat[tid * 2 + 0];
            a.part[((long long)pt * a.Cout + co0 + tid) * 2 + 1] = stat[tid * 2 + 1];
        }
    }
}

}  // namespace vr
