// conv_x3.hip's convolution (direct 3x3 stride-1, fp32 products from six bf16 products; lib/layers.py:12-20) re-tiled around what the
// round-4 measurements say bounds it: the LDS operand reads and the barriers, not the split VALU.
//
//   * conv_x3_kernel<64,8> reads 10 operands (16 B per lane each) from LDS for every 12 MFMAs; with eight waves per CU on four SIMDs
//     that keeps the LDS pipe ~80 % busy at full matrix rate, and the pixel image P is single-buffered behind two barriers per chunk.
//     conv_x3p.hip (bf16-plane tensors, no split VALU at all) measured NO faster layer for layer except where its 8-wave 64 x 16
//     tile applied: the split pass is not the limit.
//   * Here a workgroup is 8 waves = WMW cout groups (32 couts each) x WNW row groups (WN rows each) of a (32 WMW) x (WN WNW) x 32
//     tile; a wave keeps ONE 32-cout block and WN = 4 pixel rows.  The B operands (pixels) of a tap column tx are read ONCE for the
//     WN + 2 rows the three taps ty = 0..2 need and stay in registers: per 8-channel chunk a wave reads 27 A + 3 * 2 * (WN + 2) = 63
//     operands for 108 MFMAs (0.58 per MFMA, was 0.83).
//   * P is double-buffered: the split pass of chunk k+1 writes the other buffer BEFORE the multiply phase of chunk k, the weight DMA
//     of chunk k+1 and the pixel loads of chunk k+2 are issued in front of it too -- ONE barrier per chunk (a second one only in
//     front of a split pass that interpolates the fused decoder upsample from its low-resolution staging tile).
// Everything else is conv_x3.hip's: per-thread halo pixels loaded with buffer_load_dword and hand-placed waits, split3_pair, weights
// pre-split by x3_weights_kernel ([chunk][tap][plane][cout][8 ch]) arriving by LDS-DMA, the fused bilinear x2 of eval decoders, the
// shared store epilogue (conv_epilogue.h), BatchNorm partial statistics in training, a contiguous tile range per XCD.
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

__device__ __forceinline__ float x3b_load(i32x4 rsrc, int voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void x3b_wait8(float (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N) : "memory");
}

template <int WMW, int WN>
struct X3bCfg {
    static constexpr int NW = 8, NT = 512, WNW = NW / WMW;
    static constexpr int MT = 32 * WMW, TH = WN * WNW;
    static constexpr int TW = 32, CK = 8, KK = 9;
    static constexpr int TH_in = TH + 2, PW = TW + 2;
    static constexpr int NSLOT = TH_in * PW;
    static constexpr int NPASS = (NSLOT + NT - 1) / NT;
    static constexpr int PLANE = NSLOT * 16;
    static constexpr int P_BYTES = 3 * PLANE;
    static constexpr int NWP = KK * 3 * MT;
    static constexpr int W_BYTES = NWP * 16;
    static constexpr int NWPASS = (NWP + NT - 1) / NT;
    static constexpr int LROWS = TH / 2 + 3, LW = 20, LSLOT = LROWS * LW;
    static constexpr int W_OFF = 2 * P_BYTES;
    static constexpr int L_OFF = W_OFF + 2 * W_BYTES;
    static constexpr int L_BYTES = CK * LSLOT * 4;
    static constexpr int E_OFF = L_OFF + L_BYTES;                 // [WMW][3][32] fp32: bias, scale, shift per cout group
    static constexpr int LDS_BYTES = E_OFF + 3 * MT * 4;
    static constexpr int NXL = 8 * NPASS, NWMIN = (NWP / 64) / NW;
    static_assert(NW % WMW == 0 && LDS_BYTES <= 160 * 1024 && 2 * NXL + NWMIN + 4 < 64 && LSLOT <= NT, "tile");
};

template <int WMW, int WN>
__global__ __launch_bounds__(512, 1) void conv_x3b_kernel(const ConvArgs a) {
    using Cfg = X3bCfg<WMW, WN>;
    constexpr int NT = Cfg::NT, NW = Cfg::NW, MT = Cfg::MT, TH = Cfg::TH, TW = Cfg::TW, KK = Cfg::KK, PW = Cfg::PW, NSLOT = Cfg::NSLOT,
                  NPASS = Cfg::NPASS, PLANE = Cfg::PLANE, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS;
    extern __shared__ __attribute__((aligned(16))) char smem_x3b[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    const int per_xcd = (a.npt + 7) >> 3;
    const int pt = xcd * per_xcd + rr / a.nct;                    // a contiguous, row-major range of pixel tiles per XCD (conv_x3.hip)
    if (pt >= a.npt) return;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int n = pt / tiles_per_img;
    const int trem = pt - n * tiles_per_img;
    const int h0 = (trem / a.tiles_w) * TH;
    const int w0 = (trem % a.tiles_w) * TW;
    const int co0 = ct * MT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WMW, wr = wave / WMW;                   // this wave: couts [32 wm, +32), rows [WN wr, +WN) of the tile
    const int nchunk = (a.Cin + 7) >> 3;
    const unsigned lds0 = (unsigned)(size_t)smem_x3b;

    // ---- this thread's pixels of the halo tile ----
    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int s = p * NT + tid;
        const int r = s / PW, c = s - r * PW;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // ---- fused bilinear x2 (eval decoders): low-resolution pixel of the staging tile + interpolation weights per halo pixel ----
    const ConvSrc& us = a.src[0].up ? a.src[0] : (a.src[1].up ? a.src[1] : a.src[2]);
    const bool any_up = a.src[0].up | a.src[1].up | a.src[2].up;
    int lrow = 0, lcol4 = 0;
    int lidx[NPASS];
    float lh[NPASS], lw_[NPASS];
    if (any_up) {
        const int lr0 = (int)(us.rh * (float)(h0 > 0 ? h0 - 1 : 0)), lc0 = (int)(us.rw * (float)(w0 > 0 ? w0 - 1 : 0));
        const int lr = lr0 + tid / Cfg::LW, lc = lc0 + tid % Cfg::LW;
        const bool lok = tid < Cfg::LSLOT && lr < us.H && lc < us.W;
        lrow = lok ? lr : 0;
        lcol4 = lok ? lc * 4 : (int)0x80000000u;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int s = p * NT + tid;
            const int r = s / PW, c = s - r * PW;
            const int hi = h0 - 1 + r, wi = w0 - 1 + c;
            const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
            const float h1r = us.rh * (float)(ok ? hi : 0), w1r = us.rw * (float)(ok ? wi : 0);
            const int h1 = (int)h1r, w1 = (int)w1r;
            lidx[p] = ok ? ((h1 - lr0) * Cfg::LW + (w1 - lc0)) * 4 : -1;
            lh[p] = h1r - (float)h1;
            lw_[p] = w1r - (float)w1;
        }
    }
    // ---- weights: LDS order [tap][plane][m], source x3w[chunk][(tap * 3 + plane) * CoutPad + co0 + m] ----
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + NW * i) * 64 + lane;
        const int m = q % MT, tp = q / MT;
        woff[i] = (unsigned)((tp * a.CoutPad + m) * 16);
    }
    const long long wchunk_bytes = (long long)KK * 3 * a.CoutPad * 16;
    auto issue_w = [&](int k) {
        const char* wb = static_cast<const char*>(a.x3w) + k * wchunk_bytes + (long long)co0 * 16;
        i32x4 wrs = make_rsrc(reinterpret_cast<const float*>(wb), (unsigned)(wchunk_bytes - (long long)co0 * 16));
        settle_rsrc(wrs);
        const unsigned ws_b = lds0 + (unsigned)(Cfg::W_OFF + (k & 1) * Cfg::W_BYTES);
#pragma unroll
        for (int i = 0; i < NWPASS; ++i) {
            const int pp = wave + NW * i;
            if ((pp + 1) * 64 <= NWP) dma16(ws_b + pp * 1024, woff[i], wrs);
            else if (pp * 64 + lane < NWP) dma16(ws_b + pp * 1024, woff[i], wrs);
        }
    };
    // running scalar state of the virtual concat (channels are visited strictly in order)
    const float* xp = a.src[0].p + (long long)n * a.src[0].sN;
    long long xsC = a.src[0].sC;
    unsigned xsH4 = (unsigned)a.src[0].sH * 4u;
    int xend = a.c1, xsi = 0;
    bool xup = a.src[0].up != 0;
    unsigned upm[2] = {0u, 0u};
    int xvo[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) xvo[p] = (int)(hrow[p] * xsH4 + wcol4[p]);
    auto next_source = [&]() {
        ++xsi;
        if (xsi == 1) { xp = a.src[1].p + (long long)n * a.src[1].sN; xsC = a.src[1].sC; xsH4 = (unsigned)a.src[1].sH * 4u; xend = a.c2; xup = a.src[1].up != 0; }
        else { xp = a.src[2].p + (long long)n * a.src[2].sN; xsC = a.src[2].sC; xsH4 = (unsigned)a.src[2].sH * 4u; xend = 1 << 30; xup = a.src[2].up != 0; }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) xvo[p] = (int)(hrow[p] * xsH4 + wcol4[p]);
    };
    float xr[2][NPASS][8];
    auto load_channel = [&](int k, int cl, auto par) {
        constexpr int PAR = decltype(par)::value;
        const int ci = k * 8 + cl;
        const bool live = ci < a.Cin;
        if (live && ci >= xend) next_source();
        if (live && ci >= xend) next_source();
        i32x4 xs = make_rsrc(xp, live ? 0x7FFFFFF0u : 0u);
        const bool up = live && xup;
        if (cl == 0) upm[PAR] = 0u;
        upm[PAR] |= (up ? 1u : 0u) << cl;
        i32x4 xs1 = make_rsrc(xp, (live && !up) ? 0x7FFFFFF0u : 0u);
        settle_rsrc(xs);
        settle_rsrc(xs1);
        xr[PAR][0][cl] = x3b_load(xs, up ? (int)((unsigned)lrow * xsH4) + lcol4 : xvo[0]);
#pragma unroll
        for (int p = 1; p < NPASS; ++p) xr[PAR][p][cl] = x3b_load(xs1, xvo[p]);
        if (live) xp += xsC;
    };
    auto wait_pixels = [&](auto par, auto newer) {
        constexpr int PAR = decltype(par)::value, NEWER = decltype(newer)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) x3b_wait8<NEWER>(xr[PAR][p]);
    };
    auto stage_lowres = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        if (upm[PAR] == 0u) return;
        float* lq = reinterpret_cast<float*>(smem_x3b + Cfg::L_OFF) + tid;
#pragma unroll
        for (int cl = 0; cl < 8; ++cl)
            if (((upm[PAR] >> cl) & 1u) && tid < Cfg::LSLOT) lq[cl * Cfg::LSLOT] = xr[PAR][0][cl];
    };
    // split the pixels of register set PAR into the three bf16 planes of P buffer `buf`
    auto convert = [&](auto par, int buf) {
        constexpr int PAR = decltype(par)::value;
        char* const Pb = smem_x3b + buf * Cfg::P_BYTES;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int s = p * NT + tid;
            if ((p + 1) * NT <= NSLOT || s < NSLOT) {
                vr_i32x4 ph, pm, pl;
                if (upm[PAR] != 0u) {
                    const char* lq = smem_x3b + Cfg::L_OFF + (lidx[p] >= 0 ? lidx[p] : 0);
                    const float h1l = lh[p], h0l = 1.f - h1l, w1l = lw_[p], w0l = 1.f - w1l;
#pragma unroll
                    for (int cl = 0; cl < 8; ++cl) {
                        if ((upm[PAR] >> cl) & 1u) {
                            const float* q = reinterpret_cast<const float*>(lq + cl * Cfg::LSLOT * 4);
                            const float v00 = q[0], v01 = q[1], v10 = q[Cfg::LW], v11 = q[Cfg::LW + 1];
                            const float v = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
                            xr[PAR][p][cl] = lidx[p] >= 0 ? v : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int h, m, l;
                    split3_pair(xr[PAR][p][2 * j], xr[PAR][p][2 * j + 1], h, m, l);
                    ph[j] = h; pm[j] = m; pl[j] = l;
                }
                char* q = Pb + s * 16;
                *reinterpret_cast<vr_i32x4*>(q) = ph;
                *reinterpret_cast<vr_i32x4*>(q + PLANE) = pm;
                *reinterpret_cast<vr_i32x4*>(q + 2 * PLANE) = pl;
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // B operands [b1|b2] and [b1|b3]: lanes 0-31 read plane 0, lanes 32-63 plane 1 resp. 2; pixel (row wr*WN + ri, col l31 + tx)
    const int bb0 = (khalf * NSLOT + wr * WN * PW + l31) * 16;
    const int bb1 = (2 * khalf * NSLOT + wr * WN * PW + l31) * 16;
    // A operands [a1|a1], [a2|a2], [a3|a1] of this wave's cout block
    const int ab0 = (wm * 32 + l31) * 16, ab1 = (MT + wm * 32 + l31) * 16, ab2 = ((khalf ? 0 : 2) * MT + wm * 32 + l31) * 16;

    f32x16 acc[1][WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][ni][r] = 0.f;

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // epilogue constants: E[wm'][3][32]
    float ecv[3];
    {
        const int ec = co0 + (tid & (MT - 1));
        const int ecc = ec < a.Cout ? ec : a.Cout - 1;
        i32x4 rb = make_rsrc(a.bias, a.bias ? 0x7FFFFFF0u : 0u);
        i32x4 re = make_rsrc(a.epi, a.epi ? 0x7FFFFFF0u : 0u);
        settle_rsrc(rb);
        settle_rsrc(re);
        ecv[0] = x3b_load(rb, ecc * 4);
        ecv[1] = x3b_load(re, ecc * 8);
        ecv[2] = x3b_load(re, ecc * 8 + 4);
    }
    // prologue: chunk 0 -> P[0], W[0]; pixels of chunk 1 in flight
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(0, cl, P0{});
    issue_w(0);
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(1, cl, P1{});
    wait_pixels(P0{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});
    asm volatile("" : "+v"(ecv[0]), "+v"(ecv[1]), "+v"(ecv[2]));
    if (tid < MT) {
        float* E = reinterpret_cast<float*>(smem_x3b + Cfg::E_OFF) + (tid >> 5) * 96 + (tid & 31);
        E[0] = ecv[0];
        E[32] = a.epi ? ecv[1] : 1.f;
        E[64] = a.epi ? ecv[2] : 0.f;
    }
    if (any_up) {
        stage_lowres(P0{});
        lds_barrier();
    }
    convert(P0{}, 0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk 0 landed; chunk 1's pixels stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // Iteration k: [split chunk k+1 into the other P buffer] [weight DMA of chunk k+1, pixel loads of chunk k+2] multiply chunk k, barrier.
    auto chunk = [&](int k, auto par) {
        constexpr int PAR = decltype(par)::value;                 // k & 1: P / W buffers of chunk k; pixel registers of chunk k are free
        using Q = std::integral_constant<int, PAR ^ 1>;
        const bool more = k + 1 < nchunk;
        if (more) {
            // chunk k+1's pixels (issued one multiply phase ago) -> P[(k+1)&1], last read in the multiply phase of chunk k-1
            wait_pixels(Q{}, std::integral_constant<int, 0>{});
            if (any_up && upm[PAR ^ 1] != 0u) {                   // (wave-uniform, identical in every wave)
                stage_lowres(Q{});
                lds_barrier();
            }
            convert(Q{}, PAR ^ 1);
            issue_w(k + 1);
            if (k + 2 < nchunk + 1) {                             // chunk k+2's pixels (always the same number of loads: empty descriptors beyond Cin)
#pragma unroll
                for (int cl = 0; cl < 8; ++cl) load_channel(k + 2, cl, par);
            }
        }
        {
            const char* Pb = smem_x3b + PAR * Cfg::P_BYTES;
            const char* Wb = smem_x3b + Cfg::W_OFF + PAR * Cfg::W_BYTES;
            vr_bf16x8 A[2][3], B[WN + 2][2];
            auto read_A = [&](int t, int buf) {
                const char* q = Wb + (t * 3 * MT) * 16;
                A[buf][2] = *reinterpret_cast<const vr_bf16x8*>(q + ab2);
                A[buf][1] = *reinterpret_cast<const vr_bf16x8*>(q + ab1);
                A[buf][0] = *reinterpret_cast<const vr_bf16x8*>(q + ab0);
            };
            read_A(0, 0);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                // the WN + 2 pixel rows the taps (ty, tx), ty = 0..2, need: read once, kept in registers
#pragma unroll
                for (int ri = 0; ri < WN + 2; ++ri) {
                    const int o = (ri * PW + tx) * 16;
                    B[ri][1] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb1 + o);
                    B[ri][0] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb0 + o);
                }
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const int step = tx * 3 + ty, cur = step & 1;
                    // next tap in this order: (ty + 1, tx) or (0, tx + 1)
                    if (step + 1 < KK) read_A(ty < 2 ? (ty + 1) * 3 + tx : tx + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[0][ni] = mfma_bf16x16(A[cur][2], B[ni + ty][1], acc[0][ni]);
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[0][ni] = mfma_bf16x16(A[cur][1], B[ni + ty][0], acc[0][ni]);
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[0][ni] = mfma_bf16x16(A[cur][0], B[ni + ty][0], acc[0][ni]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (more) {
            // outstanding, oldest first: weights of chunk k+1 | chunk k+2's pixels
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");
            __builtin_amdgcn_s_barrier();                        // P(k+1), W(k+1) complete; everyone is done reading P(k), W(k)
            asm volatile("" ::: "memory");
        }
    };
    for (int k = 0; k < nchunk; k += 2) {
        chunk(k, P0{});
        if (k + 1 < nchunk) chunk(k + 1, P1{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (prefetches beyond Cin: empty-descriptor loads nobody else waits for)

    // ---------------- epilogue (conv_epilogue.h) for this wave's 32-cout block ---------------------------------------------------
    {
        int hon[WN], won[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) { hon[ni] = h0 + wr * WN + ni; won[ni] = w0 + l31; }
        epi_store<32, 1, WN>(VR_EPI_ARGS(a), acc, reinterpret_cast<const float*>(smem_x3b + Cfg::E_OFF) + wm * 96, n, co0 + wm * 32, khalf,
                             h0 + TH <= a.Hout && w0 + TW <= a.Wout, hon, won);
    }
    // ---------------- BatchNorm partial statistics (training) -------------------------------------------------------------------
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_x3b);                    // [WNW row groups][MT][2]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int ho = h0 + wr * WN + ni, wo = w0 + l31;
                if (ho < a.Hout && wo < a.Wout) {
                    const float v = acc[0][ni][r];
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                s1 += __shfl_xor(s1, off, 64);
                s2 += __shfl_xor(s2, off, 64);
            }
            if (l31 == 0) {
                const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                red[(wr * MT + m) * 2 + 0] = s1;
                red[(wr * MT + m) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < MT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < Cfg::WNW; ++w) {
                s1 += red[(w * MT + tid) * 2 + 0];
                s2 += red[(w * MT + tid) * 2 + 1];
            }
            const int co = co0 + tid;
            if (co < a.Cout) {
                a.part[((long long)pt * a.Cout + co) * 2 + 0] = s1;
                a.part[((long long)pt * a.Cout + co) * 2 + 1] = s2;
            }
        }
    }
}

template <int WMW, int WN>
static void x3b_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = X3bCfg<WMW, WN>;
    auto kern = conv_x3b_kernel<WMW, WN>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

// MT 64 -> 64 couts x 16 rows (2 cout groups x 4 row groups of 4 rows); MT 32 -> 32 couts x 32 rows (8 row groups of 4 rows)
void x3b_launch_conv(const ConvArgs& a, const X3Tile& t, hipStream_t st) {
    if (t.MT == 64) x3b_launch<2, 4>(a, st);
    else x3b_launch<1, 4>(a, st);
}

}  // namespace vr
