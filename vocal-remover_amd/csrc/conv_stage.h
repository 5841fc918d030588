// Shared LDS staging of a haloed, transformed input tile (used by the forward/dgrad conv kernel and
// by the weight-gradient kernel).  This is where the fusion of vr_common.h happens: virtual channel
// concat of up to three sources, bilinear x2 upsample (align_corners=True), the producer's
// BatchNorm affine + ReLU/LeakyReLU, the Dropout2d keep-mask, conv zero padding, and -- for the
// data-gradient of stride-2 convs -- zero insertion between the elements of the incoming gradient.
#pragma once
#include "vr_common.h"

namespace vr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, float slope) { return v > 0.f ? v : v * slope; }

// Stage channels [c0, c0+CK) of the virtual input for the tile whose top-left virtual coordinate is
// (hbase, wbase) into Xs[cl*CS + hh*TWp + ww], hh < TH_in, ww < TW_in.  One channel per wave pass,
// lanes over the tile; all global loads of a batch are issued before any is consumed (addresses
// are clamped so the loads are unconditional).  `wave` must be wave-uniform.
template <int TH_in, int TW_in, int TWp, int CS, int CK, int NWAVES>
__device__ __forceinline__ void stage_input_chunk(const ConvArgs& a, float* Xs, int c0, int n, int hbase, int wbase,
                                                  int wave, int lane) {
    constexpr int NE = TH_in * TW_in;
    constexpr int NP = (NE + 63) / 64;
#pragma unroll 1
    for (int cl = wave; cl < CK; cl += NWAVES) {
        const int ci = c0 + cl;               // wave-uniform
        float* dst = Xs + cl * CS;
        if (ci >= a.Cin) {
            for (int e = lane; e < NE; e += 64) dst[(e / TW_in) * TWp + (e % TW_in)] = 0.f;
            continue;
        }
        const int si = (ci >= a.c1) + (ci >= a.c2);
        const ConvSrc& s = a.src[si];
        const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
        const float* base = s.p + (long long)n * s.sN + (long long)clc * s.sC;
        float sc0 = 1.f, sh0 = 0.f, sc1 = 1.f, sh1 = 0.f;
        if (s.aff0) { sc0 = s.aff0[2 * clc]; sh0 = s.aff0[2 * clc + 1]; }
        if (s.aff1) { sc1 = s.aff1[2 * clc]; sh1 = s.aff1[2 * clc + 1]; }
        const float post = s.post ? s.post[n * s.C + clc] : 1.f;
        const float slope = s.slope;
        if (!s.up) {
            constexpr int PB = NP < 6 ? NP : 6;
            const int zs = s.zins;            // 1: source sits on the even virtual coordinates only
#pragma unroll 1
            for (int p0 = 0; p0 < NP; p0 += PB) {
                float raw[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    int e = lane + (p0 + j) * 64;
                    e = e < NE ? e : NE - 1;
                    int hs = (hbase + e / TW_in) >> zs, ws = (wbase + e % TW_in) >> zs;
                    hs = hs < 0 ? 0 : (hs >= s.H ? s.H - 1 : hs);
                    ws = ws < 0 ? 0 : (ws >= s.W ? s.W - 1 : ws);
                    raw[j] = base[(long long)hs * s.sH + ws];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int e = lane + (p0 + j) * 64;
                    const int hh = e / TW_in, ww = e % TW_in;
                    const int hi = hbase + hh, wi = wbase + ww;
                    const bool lo = (hi >> zs) < s.hsplit;
                    float v = act_apply(fmaf(raw[j], lo ? sc0 : sc1, lo ? sh0 : sh1), slope) * post;
                    bool ok = hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
                    if (zs) ok = ok && !((hi | wi) & 1) && (hi >> 1) < s.H && (wi >> 1) < s.W;
                    if (!ok) v = 0.f;
                    if (e < NE) dst[hh * TWp + ww] = v;
                }
            }
        } else {
            // bilinear x2, align_corners=True (torch upsample_bilinear2d): src = dst*(in-1)/(out-1)
            constexpr int PB = NP < 3 ? NP : 3;
#pragma unroll 1
            for (int p0 = 0; p0 < NP; p0 += PB) {
                float r00[PB], r01[PB], r10[PB], r11[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    int e = lane + (p0 + j) * 64;
                    e = e < NE ? e : NE - 1;
                    int hi = hbase + e / TW_in, wi = wbase + e % TW_in;
                    hi = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi);
                    wi = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
                    const int h1 = (int)(s.rh * (float)hi), w1 = (int)(s.rw * (float)wi);
                    const int h1p = (h1 < s.H - 1) ? 1 : 0, w1p = (w1 < s.W - 1) ? 1 : 0;
                    const float* q0 = base + (long long)h1 * s.sH + w1;
                    const float* q1 = q0 + (long long)h1p * s.sH;
                    r00[j] = q0[0]; r01[j] = q0[w1p]; r10[j] = q1[0]; r11[j] = q1[w1p];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int e = lane + (p0 + j) * 64;
                    const int hh = e / TW_in, ww = e % TW_in;
                    const int hi = hbase + hh, wi = wbase + ww;
                    const float h1r = s.rh * (float)hi, w1r = s.rw * (float)wi;
                    const float h1l = h1r - (float)(int)h1r, w1l = w1r - (float)(int)w1r;
                    const float h0l = 1.f - h1l, w0l = 1.f - w1l;
                    const float v00 = act_apply(fmaf(r00[j], sc0, sh0), slope);
                    const float v01 = act_apply(fmaf(r01[j], sc0, sh0), slope);
                    const float v10 = act_apply(fmaf(r10[j], sc0, sh0), slope);
                    const float v11 = act_apply(fmaf(r11[j], sc0, sh0), slope);
                    float v = (h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11)) * post;
                    if (!(hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win)) v = 0.f;
                    if (e < NE) dst[hh * TWp + ww] = v;
                }
            }
        }
    }
}

}  // namespace vr
