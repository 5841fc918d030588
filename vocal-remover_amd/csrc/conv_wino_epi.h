// Epilogue shared by the Winograd kernels (conv_wino.hip, conv_wino6.hip): the accumulators acc[fi][mi][ni] of wave w hold
// frequencies 2w + fi of a (MT couts) x (64 Winograd tiles = 8 x 32 pixels) block; the 16 frequencies of a (cout, tile) pair
// meet in LDS (32 couts x 32 tiles per pass), A^T M A, bias / folded BatchNorm / activation, stores, BatchNorm partial sums.
#pragma once
#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

// Exchange buffer of one pass: [f 16][register pair 8][lane 64] x 8 bytes = 64 KB, + [MT][2] BatchNorm partial sums.
// (Exchanging both pixel halves of a cout half per barrier -- 2 x 64 KB, possible in the 64-cout kernels -- was measured: no gain.)
__host__ __device__ constexpr int wino_epilogue_floats(int MT) { return 16 * 8 * 64 * 2 + 2 * MT; }

// The lanes of all 8 waves own the SAME (cout, tile) pairs -- accumulator register r of lane (khalf, l31) is cout
// (r & 3) + 8 (r >> 2) + 4 khalf, tile l31 -- for different frequencies, so the exchange is wave-to-wave at a fixed lane: every
// wave stores its two frequencies as 8 register PAIRS per lane (ds_write_b64, lane-contiguous), and wave w reads the 16
// frequencies of pair w (ds_read_b64) and finishes the two couts of that pair.  Stores stay coalesced over the tiles (l31).
template <int MT>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, f32x16 (&acc)[2][MT / 32][2], float* smem, int tid, int wave,
                                              int khalf, int l31, int n, int h0, int w0, int co0, int pt) {
    constexpr int WM = MT / 32;
    if (a.dbg == 4) return;                                            // (ablation: no epilogue, no stores)
    vr_f32x2* Mx = reinterpret_cast<vr_f32x2*>(smem);                  // [16][8][64]
    float* stat = smem + 16 * 8 * 64 * 2;                              // [MT][2] BatchNorm partial sums (training)
    const int lane = khalf * 32 + l31;
    if (a.part && tid < 2 * MT) stat[tid] = 0.f;                       // (ordered by the first pass's barriers)
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (mi + ni > 0) lds_barrier();                            // previous pass has been read
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    vr_f32x2 v;
                    v[0] = acc[fi][mi][ni][2 * rp];
                    v[1] = acc[fi][mi][ni][2 * rp + 1];
                    Mx[((2 * wave + fi) * 8 + rp) * 64 + lane] = v;
                }
            lds_barrier();
            vr_f32x2 mm[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) mm[f] = Mx[(f * 8 + wave) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 2 * wave + j;
                const int col = (r & 3) + 8 * (r >> 2) + 4 * khalf;    // cout within the 32 of this pass
                const int tl = l31;
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = mm[f][j];
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
                // 8-byte stores when the row pair (wo, wo + 1) is inside and the destination rows are 8-byte aligned (wo is even)
                const bool al8 = ((reinterpret_cast<size_t>(dp) | (size_t)(dN * 4) | (size_t)(dC * 4) | (size_t)(dH * 4)) & 7) == 0;
                const bool in_c0 = wo < a.Wout, in_c1 = wo + 1 < a.Wout;
                float* qrow = dp + (long long)n * dN + (long long)cod * dC + (long long)ho * dH + wo;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int dr = 0; dr < 2; ++dr) {
                    const bool in_r = ho + dr < a.Hout;
                    const float v0 = y[dr][0] + b, v1 = y[dr][1] + b;
                    if (in_r && in_c0) { t1 += v0; t2 = fmaf(v0, v0, t2); }
                    if (in_r && in_c1) { t1 += v1; t2 = fmaf(v1, v1, t2); }
                    if (in_r && co < a.Cout && dp) {
                        float* q = qrow + (long long)dr * dH;
                        float o0 = act_apply(fmaf(v0, esc, esh), eslope), o1 = act_apply(fmaf(v1, esc, esh), eslope);
                        if (al8 && in_c1) {
                            vr_f32x2* q2 = reinterpret_cast<vr_f32x2*>(q);
                            vr_f32x2 o;
                            o[0] = o0; o[1] = o1;
                            if (dacc) { const vr_f32x2 old = *q2; o[0] += old[0]; o[1] += old[1]; }
                            *q2 = o;
                        } else {
                            if (in_c0) q[0] = dacc ? q[0] + o0 : o0;
                            if (in_c1) q[1] = dacc ? q[1] + o1 : o1;
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
