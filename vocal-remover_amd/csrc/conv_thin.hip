// Convs with at most 16 output channels on plain inputs (the full-resolution decoder / stage-tail layers of the two small
// band nets, lib/nets.py:59-66: 49->16, 25->8, 48->16, 10->16, 16->16): the 32-row MFMA of
// conv_dma.hip / conv_wino.hip multiplies 16+ rows of zero padding for them.  v_mfma_f32_16x16x4_f32 has the same rate
// (1024 multiply-adds in 32 cycles) on a 16-cout x 16-pixel tile: rows = couts, columns = 16 consecutive pixels of an image
// row, k = 4 input channels per instruction (lane = (row or column) + 16 * k).
// Everything else is conv_dma.hip's scheme: 4 waves, inputs and the 16-column weight slice by `buffer_load_dwordx4 ... lds`
// (zero padding = out-of-range lanes), two LDS buffers, one barrier per input-channel chunk, operands of the next k-step
// read before the MFMAs of the current one.  The channel pitch is kept = 16 (mod 32) floats so the four channels of a
// k-step sit on disjoint bank halves.
#include <cstdlib>

#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS, int TH, int TW, int CK>
struct ThinCfg {
    static constexpr int KK = KS * KS, MT = 16;
    static constexpr int NG = TH * TW / 16;                    // 16-pixel groups per tile
    static constexpr int GPW = NG / 4;                         // per wave
    static constexpr int TH_in = TH + KS - 1, TW_in = TW + KS - 1;
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int XS0 = (4 - PAD % 4) % 4;
    static constexpr int TWq = ((XS0 + TW_in + 3) / 4) * 4;
    static constexpr int CSX0 = TH_in * TWq;
    static constexpr int CSX = CSX0 + ((CSX0 % 32 == 16) ? 0 : ((48 - CSX0 % 32) % 32));   // = 16 (mod 32), multiple of 4
    static constexpr int XS = CK * CSX;
    static constexpr int WS = KK * CK * MT;
    static constexpr int BUF = XS + WS;
    static constexpr int NPIECE = CSX / 4, NPASS = (NPIECE + 63) / 64;
    static constexpr int NWP = WS / 4, NWPASS = (NWP + 255) / 256;
    static constexpr int CPW = CK / 4;
    static constexpr int NS = KK * (CK / 4);                   // MFMA k-steps per chunk
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    static_assert(NG % 4 == 0 && CK % 4 == 0 && CSX % 32 == 16 && TW % 16 == 0 && LDS_BYTES <= 160 * 1024, "thin conv config");
};

template <int KS, int TH, int TW, int CK>
__global__ __launch_bounds__(256, 4) void conv_thin_kernel(const ConvArgs a) {
    using Cfg = ThinCfg<KS, TH, TW, CK>;
    constexpr int KK = Cfg::KK, MT = 16, GPW = Cfg::GPW, TWq = Cfg::TWq, CSX = Cfg::CSX, CSX0 = Cfg::CSX0, XS0 = Cfg::XS0,
                  NPIECE = Cfg::NPIECE, NPASS = Cfg::NPASS, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS, CPW = Cfg::CPW, NS = Cfg::NS,
                  PAD = Cfg::PAD;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int pt = (a.dbg & 16) ? (id >> 3) * 8 + xcd : xcd * ((a.npt + 7) >> 3) + (id >> 3);      // contiguous tile range per XCD (conv_x3.hip)
    if (pt >= a.npt) return;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int n = pt / tiles_per_img;
    const int trem = pt - n * tiles_per_img;
    const int h0 = (trem / a.tiles_w) * TH;
    const int w0 = (trem % a.tiles_w) * TW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hbase = h0 - PAD, wal0 = w0 - PAD - XS0;
    const int nchunk = (a.Cin + CK - 1) / CK;
    const unsigned lds0 = (unsigned)(size_t)smem;

    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        const int hh = q / (TWq / 4), j = q % (TWq / 4);
        const int hi = hbase + hh, wi = wal0 + 4 * j;
        const bool ok = q < CSX0 / 4 && hi >= 0 && hi < a.Hin && wi >= 0 && wi + 3 < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // weight pieces: LDS order [tap][cl][16 couts], source w[(cl * KK + tap) * CoutPad + m]
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int m4 = q % 4, t2 = q / 4;
        const int cl = t2 % CK, tap = t2 / CK;
        woff[i] = (unsigned)(((cl * KK + tap) * a.CoutPad + m4 * 4) * 4);
    }

    auto issue_chunk = [&](int k) {
        const int c0 = k * CK;
        const unsigned xs_b = lds0 + (unsigned)((k & 1) * Cfg::BUF * 4);
        const unsigned ws_b = xs_b + Cfg::XS * 4;
        {
            const float* wb = a.w + (long long)c0 * KK * a.CoutPad;
            const i32x4 wr = make_rsrc(wb, (unsigned)(((long long)(a.Cin - c0) * KK * a.CoutPad) * 4));   // channels >= Cin: zeros
#pragma unroll
            for (int i = 0; i < NWPASS; ++i) {
                const int pp = wave + 4 * i;
                if (pp * 64 + lane < NWP) dma16(ws_b + pp * 1024, woff[i], wr);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int cl = wave + 4 * cc;
            const int ci = c0 + cl;
            if (ci >= a.Cin) {
                float* z = smem + (k & 1) * Cfg::BUF + cl * CSX;
                for (int e = lane; e < CSX; e += 64) z[e] = 0.f;
                continue;
            }
            const int si = (ci >= a.c1) + (ci >= a.c2);
            const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
            const float* sp = si == 0 ? a.src[0].p : (si == 1 ? a.src[1].p : a.src[2].p);
            const long long sN = si == 0 ? a.src[0].sN : (si == 1 ? a.src[1].sN : a.src[2].sN);
            const long long sC = si == 0 ? a.src[0].sC : (si == 1 ? a.src[1].sC : a.src[2].sC);
            const unsigned sH4 = (unsigned)(si == 0 ? a.src[0].sH : (si == 1 ? a.src[1].sH : a.src[2].sH)) * 4u;
            const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)clc * sC, 0x7FFFFFF0u);
            const unsigned cb = xs_b + (unsigned)(cl * CSX * 4);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const unsigned vo = hrow[p] * sH4 + wcol4[p];
                if ((p + 1) * 64 <= NPIECE) dma16(cb + p * 1024, vo, xr);
                else if (p * 64 + lane < NPIECE) dma16(cb + p * 1024, vo, xr);
            }
        }
    };

    const int kq = lane >> 4, l15 = lane & 15;
    int boff[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int pix = (wave * GPW + g) * 16 + l15;
        boff[g] = kq * CSX + (pix / TW) * TWq + pix % TW + XS0;
    }
    const int aoff = kq * MT + l15;
    f32x4 acc[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = 0.f;

    issue_chunk(0);
    dma_wait_and_barrier();
    for (int k = 0; k < nchunk; ++k) {
        if (k + 1 < nchunk) issue_chunk(k + 1);
        const float* Xs = smem + (k & 1) * Cfg::BUF;
        const float* Ws = Xs + Cfg::XS;
        float av = Ws[aoff], bv[GPW];
#pragma unroll
        for (int g = 0; g < GPW; ++g) bv[g] = Xs[boff[g]];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float avn = 0.f, bvn[GPW];
            if (s + 1 < NS) {
                const int tap = (s + 1) / (CK / 4), kk = (s + 1) % (CK / 4);
                const int toff = (tap / KS) * TWq + tap % KS;
                avn = Ws[(tap * CK + 4 * kk) * MT + aoff];
#pragma unroll
                for (int g = 0; g < GPW; ++g) bvn[g] = Xs[4 * kk * CSX + toff + boff[g]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < GPW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[g], acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) {
                av = avn;
#pragma unroll
                for (int g = 0; g < GPW; ++g) bv[g] = bvn[g];
            }
        }
        dma_wait_and_barrier();
    }

    // epilogue: accumulator register i of lane (kq, l15) = cout 4 kq + i, pixel l15 of the group
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = 4 * kq + i;
        const int cc = co < a.Cout ? co : a.Cout - 1;
        const float b = a.bias ? a.bias[cc] : 0.f;
        float esc = 1.f, esh = 0.f, eslope = 1.f;
        if (a.epi) { esc = a.epi[2 * cc]; esh = a.epi[2 * cc + 1]; eslope = a.epi_slope; }
        const int seg = (co >= a.d1) + (co >= a.d2);
        const int cod = co - (seg == 0 ? 0 : (seg == 1 ? a.d1 : a.d2));
        float* dp = seg == 0 ? a.dst[0].p : (seg == 1 ? a.dst[1].p : a.dst[2].p);
        const long long dN = seg == 0 ? a.dst[0].sN : (seg == 1 ? a.dst[1].sN : a.dst[2].sN);
        const long long dC = seg == 0 ? a.dst[0].sC : (seg == 1 ? a.dst[1].sC : a.dst[2].sC);
        const long long dH = seg == 0 ? a.dst[0].sH : (seg == 1 ? a.dst[1].sH : a.dst[2].sH);
        const int dacc = seg == 0 ? a.dst[0].accumulate : (seg == 1 ? a.dst[1].accumulate : a.dst[2].accumulate);
        const int dws = seg == 0 ? a.dst[0].wshift : (seg == 1 ? a.dst[1].wshift : a.dst[2].wshift);
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            const int pix = (wave * GPW + g) * 16 + l15;
            const int ho = h0 + pix / TW, wo = w0 + pix % TW;
            const float v = acc[g][i] + b;
            if (ho < a.Hout && wo < a.Wout) {
                s1[i] += v;
                s2[i] = fmaf(v, v, s2[i]);
                if (co < a.Cout && dp) {
                    float* q = dp + (long long)n * dN + (long long)cod * dC + (long long)ho * dH + ((long long)wo << dws);
                    const float y = act_apply(fmaf(v, esc, esh), eslope);
                    *q = dacc ? *q + y : y;
                }
            }
        }
    }
    if (a.part) {                                       // BatchNorm partial statistics of this pixel tile (training)
        __syncthreads();
        float* red = smem;                              // [4 waves][16][2]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                s1[i] += __shfl_xor(s1[i], off, 64);
                s2[i] += __shfl_xor(s2[i], off, 64);
            }
            if (l15 == 0) {
                red[(wave * 16 + 4 * kq + i) * 2 + 0] = s1[i];
                red[(wave * 16 + 4 * kq + i) * 2 + 1] = s2[i];
            }
        }
        __syncthreads();
        if (tid < 16 && tid < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += red[(w * 16 + tid) * 2 + 0];
                t2 += red[(w * 16 + tid) * 2 + 1];
            }
            a.part[((long long)pt * a.Cout + tid) * 2 + 0] = t1;
            a.part[((long long)pt * a.Cout + tid) * 2 + 1] = t2;
        }
    }
}

static bool thin_src_plain(const ConvSrc& s) { return !s.aff0 && !s.aff1 && !s.post && !s.up && !s.zins && s.slope == 1.f; }

// The launch can take the 16-cout kernel; fills TH (16, or 8 when the 16-row grid would be small).
bool thin16_pick(const ConvArgs& a, const ConvShape& s, int* TH_out) {
    static const int enabled = getenv("VR_CONV_THIN16") ? atoi(getenv("VR_CONV_THIN16")) : 1;
    if (!enabled || a.Cout > 16 || a.tapmask || a.bf16 == 1) return false;
    // measured per layer: 3x3 with >= 8 input channels 1.35-1.6x faster than the 32-row kernels; the 1x1 tails and the 2-channel
    // first layers (HBM- / latency-bound, one chunk) 10-40 % slower -- those stay where they were
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1) || a.Cin < 8) return false;
    if (a.pad_h != (s.KS - 1) / 2 || a.pad_w != (s.KS - 1) / 2 || a.Wout < 32 || (a.Win & 3)) return false;
    for (int i = 0; i < a.nsrc; ++i) {
        const ConvSrc& c = a.src[i];
        if (!thin_src_plain(c) || c.W != a.Win) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
        if (i < 3 && a.dst[i].wshift) return false;
    }
    if ((long long)a.Cin * s.KS * s.KS * a.CoutPad * 4 >= 0x7FFFFFF0LL) return false;
    const long long tiles16 = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
    *TH_out = tiles16 >= 1024 ? 16 : 8;
    return true;
}

void thin16_fill_tiling(ConvArgs& a, int TH) {
    a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_h = (a.Hout + TH - 1) / TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = 1;
}

template <int KS, int TH, int CK>
static void thin_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = ThinCfg<KS, TH, 32, CK>;
    auto kern = conv_thin_kernel<KS, TH, 32, CK>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

void thin16_launch_conv(const ConvArgs& a, const ConvShape& s, int TH, hipStream_t st) {
    (void)s;
    if (TH == 16) thin_launch<3, 16, 4>(a, st); else thin_launch<3, 8, 4>(a, st);
}

}  // namespace vr
