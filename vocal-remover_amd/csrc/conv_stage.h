// Shared LDS staging of a haloed, transformed input tile (used by the forward/dgrad conv kernel and
// by the weight-gradient kernel).  This is where the fusion of vr_common.h happens: virtual channel
// concat of up to three sources, bilinear x2 upsample (align_corners=True), the producer's
// BatchNorm affine + ReLU/LeakyReLU, the Dropout2d keep-mask, conv zero padding, and -- for the
// data-gradient of stride-2 convs -- zero insertion between the elements of the incoming gradient.
#pragma once
#include <type_traits>
#include "vr_common.h"

namespace vr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- bf16 matrix pipe with fp32 storage ("mfma_bf16" option, configs[4]): operands are rounded to bf16 (RNE) in
// registers right before the MFMA, products accumulate in fp32.  v_mfma_f32_32x32x8_bf16: lane (row/col = lane & 31,
// kgroup = lane >> 5) supplies the 4 operands k = 4*kgroup .. 4*kgroup+3; 8 passes for 4x the k of the fp32 instruction.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float vr_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 vr_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 vr_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 pack_bf16x4(float a, float b, float c, float d) {
    // (compiler builtin, not inline asm: the hazard recogniser must see the VALU write in front of the MFMA read)
    vr_f32x2 lo, hi;
    lo[0] = a; lo[1] = b; hi[0] = c; hi[1] = d;
    const vr_bf16x2 l = __builtin_convertvector(lo, vr_bf16x2), h = __builtin_convertvector(hi, vr_bf16x2);
    vr_bf16x4 v;
    v[0] = l[0]; v[1] = l[1]; v[2] = h[0]; v[3] = h[1];
    return __builtin_bit_cast(s16x4, v);
}
__device__ __forceinline__ f32x16 mfma_bf16(s16x4 a, s16x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}

// ---- fp32 products on the bf16 matrix pipe ("mfma_mode" 2): every fp32 operand x is written as the exact sum of three
// bf16 numbers  x = x1 + x2 + x3  (x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2; each subtraction is exact, and 3 x 8
// significand bits cover the 24 of fp32), and a*b is accumulated in fp32 as the six products
//     a1 b1 + a2 b1 + a1 b2 + a2 b2 + a1 b3 + a3 b1;
// the three left out (a2 b3, a3 b2, a3 b3) are below 2^-25 |a b|, i.e. under the rounding of one fp32 multiply.
// v_mfma_f32_32x32x16_bf16 takes k = 0..7 from lanes 0-31 and k = 8..15 from lanes 32-63 (row / column = lane & 31): here
// both k halves hold THE SAME 8 input channels, of two different planes, so one instruction adds two of the six products:
//     A = [a1 | a2] x B = [b1 | b1],   A = [a1 | a2] x B = [b2 | b2],   A = [a1 | a3] x B = [b3 | b1]
// = 3 instructions of 32 cycles per 8 channels where v_mfma_f32_32x32x2_f32 needs 4 of 64.
typedef __bf16 vr_bf16x8 __attribute__((ext_vector_type(8)));
typedef int vr_i32x4 __attribute__((ext_vector_type(4)));
typedef int vr_i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned vr_u32x2 __attribute__((ext_vector_type(2)));
// one pair of fp32 values -> packed bf16 pairs of the three planes
__device__ __forceinline__ void split3_pair(float x0, float x1, int& p1, int& p2, int& p3) {
    vr_f32x2 v;
    v[0] = x0; v[1] = x1;
    p1 = __builtin_bit_cast(int, __builtin_convertvector(v, vr_bf16x2));
    v[0] = x0 - __int_as_float(p1 << 16);
    v[1] = x1 - __int_as_float(p1 & (int)0xffff0000);
    p2 = __builtin_bit_cast(int, __builtin_convertvector(v, vr_bf16x2));
    v[0] -= __int_as_float(p2 << 16);
    v[1] -= __int_as_float(p2 & (int)0xffff0000);
    p3 = __builtin_bit_cast(int, __builtin_convertvector(v, vr_bf16x2));
}
__device__ __forceinline__ f32x16 mfma_bf16x16(vr_bf16x8 a, vr_bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Sum over the 32 lanes that share lane >> 5 (one accumulator row of a 32x32 MFMA tile), DPP only: quad swaps, row_half_mirror and
// row_mirror leave every lane of a row of 16 with the row's sum, row_bcast:15 adds the lower row's into the upper one -- the total is
// valid in lanes 16-31 and 48-63 ((lane & 16) != 0).  Five VALU instructions; __shfl_xor is a ds_bpermute (an LDS instruction with
// its own latency) per step: 320 of them per workgroup tile in the BatchNorm partial sums of a 64-cout training conv.
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    auto step = [](float x, auto ctrl, auto rmask) {
        constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value;
        return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, R, 0xF, true));
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});     // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});     // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});    // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});    // row_mirror
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});    // row_bcast:15 into rows 1 and 3
    return v;
}

// Fields of a.src[si] for a wave-uniform si, selected one by one with scalar selects: indexing the
// by-value kernel argument dynamically makes the compiler keep a copy of it in scratch memory.
struct SrcSel {
    const float *p, *aff0, *aff1, *post;
    long long sN, sC, sH;
    int C, H, W, hsplit, up, zins;
    float slope, rh, rw;
};
#define VR_SEL_F(a, si, f) ((si) == 0 ? (a).src[0].f : ((si) == 1 ? (a).src[1].f : (a).src[2].f))
#define VR_SELECT_SRC(s, a, si)                                                                       \
    do {                                                                                              \
        (s).p = VR_SEL_F(a, si, p); (s).aff0 = VR_SEL_F(a, si, aff0); (s).aff1 = VR_SEL_F(a, si, aff1); \
        (s).post = VR_SEL_F(a, si, post); (s).sN = VR_SEL_F(a, si, sN); (s).sC = VR_SEL_F(a, si, sC);   \
        (s).sH = VR_SEL_F(a, si, sH); (s).C = VR_SEL_F(a, si, C); (s).H = VR_SEL_F(a, si, H);           \
        (s).W = VR_SEL_F(a, si, W); (s).hsplit = VR_SEL_F(a, si, hsplit); (s).up = VR_SEL_F(a, si, up); \
        (s).zins = VR_SEL_F(a, si, zins); (s).slope = VR_SEL_F(a, si, slope);                           \
        (s).rh = VR_SEL_F(a, si, rh); (s).rw = VR_SEL_F(a, si, rw);                                     \
    } while (0)

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


// ---------------------------------------------------------------------------------------------------
// Split (issue-early / write-late) staging for the pipelined conv kernel.
//
// The staging loop used to be VALU-bound (index arithmetic, clamps, 64-bit addressing and four-tap
// interpolation set-up per element per channel).  Everything that depends only on the tile geometry
// is now computed ONCE per workgroup into small LDS tables, indexed by the flat tile element
// e = lane + 64*pass (the same for every wave, every channel and every chunk):
//   goff[s][e]  element offset inside a channel plane of plain source s (clamped; zero-inserted
//               sources already halved)
//   meta[e]     LDS offset hh*TWp+ww | valid bit per source (16+s) | "row below hsplit" bit (19+s)
//   lowoff[el]  offset inside a channel plane of the LOW-RES tile element el (upsampled sources)
//   itp[e]      {q0 = scratch offset of the top-left tap, d = w1p | (h1p*LW) << 8, lambda_h, lambda_w}
// Per channel the loader then is: table read + global_load(sgpr base, vgpr offset) in the issue
// phase; fma / max / select / ds_write in the write phase.  Upsampled sources fetch only the
// low-res tile (<= LH x LW values instead of four taps per output element), transform it once
// into a wave-private scratch and interpolate from there.
// All upsampled sources of one conv share H, W and the row stride (checked on the host).
// ---------------------------------------------------------------------------------------------------
template <int TH_in, int TW_in>
struct StageGeom {
    static constexpr int NE = TH_in * TW_in;
    static constexpr int NP = (NE + 63) / 64;
    static constexpr int NEp = NP * 64;
    static constexpr int NPX = NP + 5;            // prefetch registers per channel: NP raw values + (sc0, sh0, sc1, sh1, post)
    static constexpr int LH = TH_in / 2 + 2, LW = TW_in / 2 + 2;     // low-res footprint of the haloed tile
    static constexpr int NL = LH * LW;
    static constexpr int NPL = (NL + 63) / 64;
    static constexpr int NLp = NPL * 64;
    // LDS floats: goff[3][NEp] + meta[NEp] + lowoff[NLp] + itp[NEp][4] + per-wave scratch 4*NL
    static constexpr int TAB = 3 * NEp + NEp + NLp + 4 * NEp;
    static constexpr int SCR = 4 * NL;
    static_assert(NPL <= NP, "low-res tile must fit the prefetch registers");
};

template <int TH_in, int TW_in, int TWp>
__device__ __forceinline__ void build_stage_tables(const ConvArgs& a, int* tab, int hbase, int wbase, int tid) {
    using G = StageGeom<TH_in, TW_in>;
    int* goff = tab;
    int* meta = tab + 3 * G::NEp;
    int* lowoff = meta + G::NEp;
    int* itp = lowoff + G::NLp;
    // which source (if any) is upsampled: they all share geometry
    int us = -1;
    if (a.src[0].up) us = 0;
    if (a.nsrc > 1 && a.src[1].up) us = 1;
    if (a.nsrc > 2 && a.src[2].up) us = 2;
    float rh = 0.f, rw = 0.f; int UH = 1, UW = 1; long long usH = 0;
    if (us == 0) { rh = a.src[0].rh; rw = a.src[0].rw; UH = a.src[0].H; UW = a.src[0].W; usH = a.src[0].sH; }
    if (us == 1) { rh = a.src[1].rh; rw = a.src[1].rw; UH = a.src[1].H; UW = a.src[1].W; usH = a.src[1].sH; }
    if (us == 2) { rh = a.src[2].rh; rw = a.src[2].rw; UH = a.src[2].H; UW = a.src[2].W; usH = a.src[2].sH; }
    const int hb = hbase < 0 ? 0 : (hbase >= a.Hin ? a.Hin - 1 : hbase);
    const int wb = wbase < 0 ? 0 : (wbase >= a.Win ? a.Win - 1 : wbase);
    const int hl0 = (int)(rh * (float)hb), wl0 = (int)(rw * (float)wb);
    for (int e = tid; e < G::NEp; e += 256) {
        const int ec = e < G::NE ? e : G::NE - 1;
        const int hh = ec / TW_in, ww = ec % TW_in;
        const int hi = hbase + hh, wi = wbase + ww;
        const bool inb = hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
        int m = hh * TWp + ww;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int g = 0;
            if (s < a.nsrc && !a.src[s].up) {
                const int zs = a.src[s].zins;
                int hs = hi >> zs, ws = wi >> zs;
                bool ok = inb;
                if (zs) ok = ok && !((hi | wi) & 1) && hs < a.src[s].H && ws < a.src[s].W;
                hs = hs < 0 ? 0 : (hs >= a.src[s].H ? a.src[s].H - 1 : hs);
                ws = ws < 0 ? 0 : (ws >= a.src[s].W ? a.src[s].W - 1 : ws);
                g = (int)(hs * a.src[s].sH) + ws;
                if (ok) m |= 1 << (16 + s);
                if (hs < a.src[s].hsplit) m |= 1 << (19 + s);
            } else if (s < a.nsrc) {
                if (inb) m |= 1 << (16 + s);
                m |= 1 << (19 + s);
            }
            goff[s * G::NEp + e] = g;
        }
        meta[e] = m;
        // interpolation set-up (torch upsample_bilinear2d, align_corners=True: src = dst*(in-1)/(out-1))
        const int hc = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi);
        const int wc = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
        const float h1r = rh * (float)hc, w1r = rw * (float)wc;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = (h1 < UH - 1) ? 1 : 0, w1p = (w1 < UW - 1) ? 1 : 0;
        itp[e * 4 + 0] = (h1 - hl0) * G::LW + (w1 - wl0);
        itp[e * 4 + 1] = w1p | ((h1p * G::LW) << 8);
        itp[e * 4 + 2] = __float_as_int(h1r - (float)h1);
        itp[e * 4 + 3] = __float_as_int(w1r - (float)w1);
    }
    for (int el = tid; el < G::NLp; el += 256) {
        const int ec = el < G::NL ? el : G::NL - 1;
        int hs = hl0 + ec / G::LW, ws = wl0 + ec % G::LW;
        hs = hs >= UH ? UH - 1 : hs;
        ws = ws >= UW ? UW - 1 : ws;
        lowoff[el] = (int)(hs * usH) + ws;
    }
}

template <int TH_in, int TW_in, int CK, int NWAVES>
__device__ __forceinline__ void issue_input_loads(const ConvArgs& a, const int* tab, int c0, int n, int wave, int lane,
                                                  float (&raw)[(CK + NWAVES - 1) / NWAVES][StageGeom<TH_in, TW_in>::NPX]) {
    using G = StageGeom<TH_in, TW_in>;
    constexpr int CPW = (CK + NWAVES - 1) / NWAVES;
    const int* lowoff = tab + 4 * G::NEp;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int cl = wave + i * NWAVES;
        int ci = c0 + cl;
        ci = (cl < CK && ci < a.Cin) ? ci : a.Cin - 1;           // keep the loads unconditional
        const int si = (ci >= a.c1) + (ci >= a.c2);
        const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
        const float* p = VR_SEL_F(a, si, p);
        const long long sN = VR_SEL_F(a, si, sN), sC = VR_SEL_F(a, si, sC);
        const int up = VR_SEL_F(a, si, up);
        const float* base = p + (long long)n * sN + (long long)clc * sC;
        {   // the producer's BatchNorm affine and the dropout keep-mask are prefetched too: fetched in the
            // write stage they would expose one dependent global-load latency per channel
            const float* aff0 = VR_SEL_F(a, si, aff0);
            const float* aff1 = VR_SEL_F(a, si, aff1);
            const float* postp = VR_SEL_F(a, si, post);
            const int srcC = VR_SEL_F(a, si, C);
            float sc0 = 1.f, sh0 = 0.f, sc1 = 1.f, sh1 = 0.f, post = 1.f;
            if (aff0) { sc0 = aff0[2 * clc]; sh0 = aff0[2 * clc + 1]; }
            if (aff1) { sc1 = aff1[2 * clc]; sh1 = aff1[2 * clc + 1]; }
            if (postp) post = postp[n * srcC + clc];
            raw[i][G::NP + 0] = sc0; raw[i][G::NP + 1] = sh0; raw[i][G::NP + 2] = sc1; raw[i][G::NP + 3] = sh1;
            raw[i][G::NP + 4] = post;
        }
        // ONE store site per register: two branches writing raw[i][j] get merged by the optimizer into
        // a store through a pointer phi, which pins the whole array in scratch memory.
        const int* t = up ? (lowoff + lane) : (tab + si * G::NEp + lane);
#pragma unroll
        for (int j = 0; j < G::NP; ++j) {
            int off = t[j * 64];
            if (j >= G::NPL && up) off = 0;      // beyond the low-res tile: harmless in-bounds dummy load
            raw[i][j] = base[off];
        }
    }
}

template <int TH_in, int TW_in, int CS, int CK, int NWAVES>
__device__ __forceinline__ void write_input_stage(const ConvArgs& a, const int* tab, float* Xs, float* scratch /* per-wave [NL] */,
                                                  int c0, int n, int wave, int lane,
                                                  const float (&raw)[(CK + NWAVES - 1) / NWAVES][StageGeom<TH_in, TW_in>::NPX]) {
    using G = StageGeom<TH_in, TW_in>;
    constexpr int CPW = (CK + NWAVES - 1) / NWAVES;
    const int* meta = tab + 3 * G::NEp + lane;
    const int4* itp = reinterpret_cast<const int4*>(tab + 4 * G::NEp + G::NLp) + lane;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int cl = wave + i * NWAVES;
        if (cl >= CK) continue;
        const int ci = c0 + cl;
        float* dst = Xs + cl * CS;
        if (ci >= a.Cin) {
#pragma unroll
            for (int j = 0; j < G::NP; ++j)
                if (lane + j * 64 < G::NE) dst[meta[j * 64] & 0xffff] = 0.f;
            continue;
        }
        const int si = (ci >= a.c1) + (ci >= a.c2);
        const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
        const int up = VR_SEL_F(a, si, up);
        const float slope = VR_SEL_F(a, si, slope);
        const float sc0 = raw[i][G::NP + 0], sh0 = raw[i][G::NP + 1], sc1 = raw[i][G::NP + 2], sh1 = raw[i][G::NP + 3];
        const float post = raw[i][G::NP + 4];
        const int okbit = 1 << (16 + si), lobit = 1 << (19 + si);
        if (!up) {
#pragma unroll
            for (int j = 0; j < G::NP; ++j) {
                const int m = meta[j * 64];
                const bool lo = m & lobit;
                float v = fmaf(raw[i][j], lo ? sc0 : sc1, lo ? sh0 : sh1);
                v = fmaxf(v, v * slope) * post;                  // slope in [0,1]: = v > 0 ? v : v*slope
                v = (m & okbit) ? v : 0.f;
                if (lane + j * 64 < G::NE) dst[m & 0xffff] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::NPL; ++j) {
                const float t = fmaf(raw[i][j], sc0, sh0);
                if (lane + j * 64 < G::NL) scratch[lane + j * 64] = fmaxf(t, t * slope);
            }
            // the same wave wrote and reads the scratch: one wave's LDS operations complete in order
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < G::NP; ++j) {
                const int m = meta[j * 64];
                const int4 u = itp[j * 64];
                const float* q0 = scratch + u.x;
                const int d1 = u.y & 0xff, d2 = u.y >> 8;
                const float h1l = __int_as_float(u.z), w1l = __int_as_float(u.w);
                const float h0l = 1.f - h1l, w0l = 1.f - w1l;
                float v = (h0l * (w0l * q0[0] + w1l * q0[d1]) + h1l * (w0l * q0[d2] + w1l * q0[d2 + d1])) * post;
                v = (m & okbit) ? v : 0.f;
                if (lane + j * 64 < G::NE) dst[m & 0xffff] = v;
            }
            __builtin_amdgcn_wave_barrier();                     // scratch is reused by this wave's next channel
        }
    }
}

}  // namespace vr
