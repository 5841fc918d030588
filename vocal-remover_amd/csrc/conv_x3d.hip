// The 16-column layers on the fp16 matrix pipe (round 6): conv_x3h.hip's arithmetic -- every fp32 product as three fp16 products of
// power-of-two scaled operand planes, fp32 accumulate -- for the convolutions at 1/16 resolution, which conv_x3h's 32-column tile cannot
// take: the three DILATED 3x3 branches of layers.ASPPModule (dilation (4,2) / (8,4) / (12,6), padding = dilation: lib/layers.py:77-85,
// lib/nets.py:10,18), its 1x1 branch conv2 (layers.py:74-76) and Encoder.conv2 of enc5 (3x3, dilation 1: layers.py:34, nets.py:16).
// Until round 5 they ran on the fp32 matrix pipe (conv_dma.hip, 16-column tiles): 0.98 + 0.3 + 0.2 ms of an 8.5 ms inference step for 8 %
// of its FLOPs, at 0.40 - 0.50 of that pipe.
//
// What is different from conv_x3h.hip:
//   * a 32-pixel MFMA column block is TWO rows of 16 columns; a workgroup (4 waves) owns TH = 16 rows x 16 columns x MT couts;
//   * a dilated tap reaches DH rows / DW columns away, so the halo tile is (TH + 2 DH) x (16 + 2 DW) pixels -- for dilation (12,6) 1120
//     slots around 256 outputs.  Only the pixels that EXIST are loaded and split: the image is 16 columns wide, so the 2 DW halo columns
//     are always padding, and so are the halo rows beyond the image.  The two LDS planes are zeroed once; the split pass writes the
//     (TH + 2 DH) x 16 real slots (rows outside the image: zeros again) and the matrix phase reads every tap at a compile-time offset,
//     no bounds arithmetic at all;
//   * the ASPP launch runs the module's FOUR branch convolutions side by side (blockIdx.y = branch: each branch its own ConvArgs and its
//     own instantiation of the tile body): every branch alone fills 44 - 352 of the 512 workgroup slots of the chip.
// Everything else is conv_x3h.hip's schedule: pixel registers two chunks ahead with hand-placed waits, weights by LDS-DMA
// (double-buffered; x3h format, launch_x3h_weights with KK = 9 or 1), the running power-of-two shift, conv_epilogue.h, BatchNorm
// partial sums.
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"
#include "x3h_common.h"

namespace vr {

template <int KK, int DH, int DW, int MT, int TH>
struct X3dCfg {
    static_assert(KK == 9 || KK == 1, "3x3 or 1x1");
    static constexpr int TW = 16, CK = 8;
    static constexpr int HH = KK == 9 ? DH : 0, HW = KK == 9 ? DW : 0;     // halo rows / columns on each side
    static constexpr int TH_in = TH + 2 * HH, PW = TW + 2 * HW;            // halo tile, pixels
    static constexpr int NSLOT = TH_in * PW;                               // LDS slots (zero border included)
    static constexpr int NREAL = TH_in * TW;                               // pixels a workgroup loads and splits per 8-channel chunk
    static constexpr int NPASS = (NREAL + 255) / 256;
    static constexpr int WM = MT / 32, WN = TH / 8;                        // 32-pixel column blocks (two rows each) per wave
    static constexpr int PLANE = NSLOT * 16, P_BYTES = 2 * PLANE;
    static constexpr int NWP = KK * 2 * MT, W_BYTES = NWP * 16, NWPASS = (NWP + 255) / 256;
    static constexpr int E_OFF = P_BYTES + 2 * W_BYTES;                    // epilogue constants [4][MT]: bias, scale, shift, 1 / weight scale
    static constexpr int M_OFF = E_OFF + 4 * MT * 4;                       // the four wave maxima of the chunk being split
    static constexpr int LDS_BYTES = M_OFF + 16;
    static constexpr int NXL = 8 * NPASS, NWMIN = (NWP / 64) / 4;          // pixel loads / (at least) weight DMAs a wave issues per chunk
    static constexpr int NG = KK == 9 ? 14 : 2;                            // matrix-instruction groups per chunk
    static_assert(TH % 8 == 0 && MT % 32 == 0 && LDS_BYTES <= 80 * 1024 && 2 * NXL + NWMIN < 64, "tile");
};

// byte offset of tap t (row-major 3x3) relative to the output pixel's own slot
template <int KK, int DH, int DW, int PW>
__device__ __host__ constexpr int x3d_tap_off(int t) {
    return KK == 1 ? 0 : ((t / 3 - 1) * DH * PW + (t % 3 - 1) * DW) * 16;
}

// One tile: output rows h0 .. h0 + TH - 1 of image n (all 16 columns), couts co0 .. co0 + MT - 1.
template <int KK, int DH, int DW, int MT, int TH>
__device__ __forceinline__ void x3d_tile(const ConvArgs& a, const int pt, const int ct, char* const smem) {
    using Cfg = X3dCfg<KK, DH, DW, MT, TH>;
    constexpr int PW = Cfg::PW, NREAL = Cfg::NREAL, NPASS = Cfg::NPASS, WM = Cfg::WM, WN = Cfg::WN, PLANE = Cfg::PLANE, NWP = Cfg::NWP,
                  NWPASS = Cfg::NWPASS, HH = Cfg::HH, HW = Cfg::HW;
    char* const Pb = smem;
    const int n = pt / a.tiles_h;
    const int h0 = (pt - n * a.tiles_h) * TH;
    const int co0 = ct * MT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = (a.Cin + 7) >> 3;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- the planes start as zeros: border columns and rows beyond the image are never written again ----
    {
        const vr_i32x4 z = {0, 0, 0, 0};
        for (int i = tid; i < 2 * Cfg::NSLOT; i += 256) *reinterpret_cast<vr_i32x4*>(Pb + i * 16) = z;
    }
    // ---- this thread's pixels: slot s = row r of the halo tile (image row h0 - HH + r), column s & 15 ----
    auto pixel_offset = [&](int p, unsigned sH4) -> int {
        const int s = p * 256 + tid;
        const int r = s >> 4, c = s & 15;
        const int hi = h0 - HH + r;
        const bool ok = s < NREAL && hi >= 0 && hi < a.Hin;
        // (product and sum kept apart: fused, hipcc emits v_mad_u64_u32 with a 64-bit addend whose undefined high half it parks in whatever
        // register is free -- including the destination of a pixel load in flight; harmless, the low word is all that is used, but
        // tools/asm_inflight_audit2.py rightly cannot tell)
        unsigned off = (unsigned)hi * sH4;
        asm volatile("" : "+v"(off));
        off += (unsigned)(c * 4);
        return ok ? (int)off : (int)0x80000000u;
    };
    // ---- weight operands: LDS order [tap][plane][m], source x3w[chunk][(tap * 2 + plane) * CoutPad + co0 + m] (conv_x3h.hip) ----
    unsigned woff0;
    {
        const int q = wave * 64 + lane;
        const int m = q % MT, tp = q / MT;
        woff0 = (unsigned)((tp * a.CoutPad + m) * 16);
    }
    const unsigned wstep = (unsigned)((256 / MT) * a.CoutPad * 16);
    const long long wchunk_bytes = (long long)KK * 2 * a.CoutPad * 16;
    auto issue_w = [&](int k) {
        const char* wb = static_cast<const char*>(a.x3w) + k * wchunk_bytes + (long long)co0 * 16;
        const i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), (unsigned)(wchunk_bytes - (long long)co0 * 16));
        const unsigned ws_b = lds0 + (unsigned)(Cfg::P_BYTES + (k & 1) * Cfg::W_BYTES);
#pragma unroll
        for (int i = 0; i < NWPASS; ++i) {
            const int pp = wave + 4 * i;
            if ((pp + 1) * 64 <= NWP) dma16s(ws_b + pp * 1024, woff0, wr, (unsigned)i * wstep);
            else if (pp * 64 + lane < NWP) dma16s(ws_b + pp * 1024, woff0, wr, (unsigned)i * wstep);
        }
    };
    // ---- the channels are visited in order: the source of the virtual concat is a running scalar state (conv_x3h.hip) ----
    const float* xp = a.src[0].p + (long long)n * a.src[0].sN;
    long long xsC = a.src[0].sC;
    unsigned xsH4 = (unsigned)a.src[0].sH * 4u;
    int xend = a.c1, xsi = 0;
    int xvo[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) xvo[p] = pixel_offset(p, xsH4);
    auto next_source = [&]() {
        ++xsi;
        if (xsi == 1) { xp = a.src[1].p + (long long)n * a.src[1].sN; xsC = a.src[1].sC; xsH4 = (unsigned)a.src[1].sH * 4u; xend = a.c2; }
        else { xp = a.src[2].p + (long long)n * a.src[2].sN; xsC = a.src[2].sC; xsH4 = (unsigned)a.src[2].sH * 4u; xend = 1 << 30; }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) xvo[p] = pixel_offset(p, xsH4);
    };
    float xr[2][NPASS][8];
    auto load_channel = [&](int k, int cl, auto par) {
        constexpr int PAR = decltype(par)::value;
        const int ci = k * 8 + cl;                                // wave-uniform
        const bool live = ci < a.Cin;
        if (live && ci >= xend) next_source();
        if (live && ci >= xend) next_source();
        const i32x4 xs = make_rsrc(xp, live ? 0x7FFFFFF0u : 0u);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) xr[PAR][p][cl] = x3h_load(xs, xvo[p]);
        if (live) xp += xsC;
    };
    auto wait_pixels = [&](auto par, auto newer) {
        constexpr int PAR = decltype(par)::value, NEWER = decltype(newer)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) x3h_wait8<NEWER>(xr[PAR][p]);
    };
    // ---- the running power-of-two shift of the pixels (conv_x3h.hip header): x' = x * 2^sh ----
    int sh = 0, shlo = 0;
    float psc = 1.f;
    auto post_max = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) m = x3h_absmax8(m, xr[PAR][p]);
        int b = __float_as_int(m);
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true));
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true));
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true));
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true));
        const int w = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
                          max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
        if (lane == 0) reinterpret_cast<int*>(smem + Cfg::M_OFF)[wave] = w;
    };
    auto read_max_exp = [&]() -> int {
        const vr_i32x4 mm = *reinterpret_cast<const vr_i32x4*>(smem + Cfg::M_OFF);
        const int w = max(max(mm[0], mm[1]), max(mm[2], mm[3]));
        return __builtin_amdgcn_readfirstlane(w) >> 23;
    };
    auto convert = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int s = p * 256 + tid;
            if ((p + 1) * 256 <= NREAL || s < NREAL) {
                vr_i32x4 ph, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int h, l;
                    split2h_pair(xr[PAR][p][2 * j], xr[PAR][p][2 * j + 1], psc, h, l);
                    ph[j] = h; pl[j] = l;
                }
                char* q = Pb + ((s >> 4) * PW + (s & 15) + HW) * 16;
                *reinterpret_cast<vr_i32x4*>(q) = ph;
                *reinterpret_cast<vr_i32x4*>(q + PLANE) = pl;
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // B operands: lane l31 of column block ni is pixel (row 2 (wave WN + ni) + (l31 >> 4), column l31 & 15); bq = its own slot for ni = 0.
    //   X(t): lanes 0-31 plane 0, lanes 32-63 plane 1 of tap t's pixel;   Y(t,t+1): plane 0, lanes 32-63 at tap t+1's pixel: DW columns to
    //   the right (t = 0, 4, 6) or one tap row down and 2 DW columns back (t = 2)
    const int bq = ((2 * wave * WN + (l31 >> 4) + HH) * PW + (l31 & 15) + HW) * 16;
    const int bX = bq + khalf * PLANE, bY1 = bq + khalf * DW * 16, bY2 = bq + khalf * (DH * PW - 2 * DW) * 16;
    const int aX = l31 * 16, aY = (MT + l31 + khalf * 2 * MT) * 16;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    float ecv[4];
    {
        const int ec = co0 + (tid & (MT - 1));
        const int ecc = ec < a.Cout ? ec : a.Cout - 1;
        const i32x4 rb = make_rsrc(a.bias, a.bias ? 0x7FFFFFF0u : 0u);
        const i32x4 re = make_rsrc(a.epi, a.epi ? 0x7FFFFFF0u : 0u);
        const i32x4 rw = make_rsrc(reinterpret_cast<const float*>(static_cast<const char*>(a.x3w) + nchunk * wchunk_bytes), 0x7FFFFFF0u);
        ecv[0] = x3h_load(rb, ecc * 4);
        ecv[1] = x3h_load(re, ecc * 8);
        ecv[2] = x3h_load(re, ecc * 8 + 4);
        ecv[3] = x3h_load(rw, ec * 4);
    }
    auto follow = [&](int e, bool first) {
        const int need = 140 - (e < 14 ? 14 : e);
        shlo = (first || need < shlo) ? need : shlo;
        int nsh = sh;
        if (first || need < sh - 1) nsh = need;
        else if (need > sh + 12) {
            nsh = need < sh + 64 ? need : sh + 64;
            nsh = nsh < shlo + 64 ? nsh : shlo + 64;
            nsh = nsh > sh ? nsh : sh;
        }
        if (nsh != sh) {
            if (!first) {
                const int d = nsh - sh;
                const float f = d < -126 ? 0.f : x3h_pow2(d);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= f;
            }
            sh = nsh;
            psc = x3h_pow2(sh);
        }
    };
    // prologue: pixels of chunk 0 -> P, weights of chunk 0 and pixels of chunk 1 in flight
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(0, cl, P0{});
    issue_w(0);
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(1, cl, P1{});
    wait_pixels(P0{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});
    asm volatile("; landed %0 %1 %2 %3" : "+v"(ecv[0]), "+v"(ecv[1]), "+v"(ecv[2]), "+v"(ecv[3]));
    if (tid < MT) {
        float* E = reinterpret_cast<float*>(smem + Cfg::E_OFF);
        E[tid] = ecv[0];
        E[MT + tid] = a.epi ? ecv[1] : 1.f;
        E[2 * MT + tid] = a.epi ? ecv[2] : 0.f;
        E[3 * MT + tid] = ecv[3];
    }
    post_max(P0{});
    lds_barrier();                                                // (the zeroed planes are complete as well)
    follow(read_max_exp(), true);
    convert(P0{});
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    auto chunk = [&](int k, auto par) {
        constexpr int PAR = decltype(par)::value;
        const bool more = k + 1 < nchunk;
        {
            const char* Wb = smem + Cfg::P_BYTES + PAR * Cfg::W_BYTES;
            vr_f16x8 A[2][WM], B[2][WN];
            // 3x3: group g = 3q + {0, 1} -> X(2q), X(2q + 1); g = 3q + 2 -> Y(2q, 2q + 1); g = 12 -> X(8); g = 13 -> Y(8).  1x1: X(0), Y(0)
            auto read_group = [&](int g, int buf) {
                const bool isY = KK == 1 ? g == 1 : (g == 13 || (g < 12 && g % 3 == 2));
                const int t = KK == 1 ? 0 : (g >= 12 ? 8 : 2 * (g / 3) + (g % 3 == 1 ? 1 : 0));
                const bool lastY = isY && t == KK - 1;            // a single tap: the upper k half meets zeros
                const int to = x3d_tap_off<KK, DH, DW, PW>(t);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) {
                    const char* q = Wb + (t * 2 * MT + mi * 32) * 16;
                    if (!isY) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aX);
                    else if (!lastY) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aY);
                    else {
                        const vr_i32x4 v = *reinterpret_cast<const vr_i32x4*>(q + aX + MT * 16);
                        vr_i32x4 z;
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = khalf ? 0 : v[j];
                        A[buf][mi] = __builtin_bit_cast(vr_f16x8, z);
                    }
                }
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int o = to + ni * 2 * PW * 16;
                    if (!isY) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bX + o);
                    else if (lastY) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bq + o);
                    else if (t == 2) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bY2 + o);
                    else B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bY1 + o);
                }
            };
            auto mfma_group = [&](int buf) {
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = mfma_f16x16(A[buf][mi], B[buf][ni], acc[mi][ni]);
            };
            read_group(0, 0);
#pragma unroll
            for (int g = 0; g < Cfg::NG; ++g) {
                const int cur = g & 1;
                if (g + 1 < Cfg::NG) read_group(g + 1, cur ^ 1);
                if (more) {                                       // weights first (needed a chunk earlier), then the pixels of chunk k + 2
                    if (g == 0) issue_w(k + 1);
                    if (KK == 9) {
                        if (g >= 1 && g <= 4) { load_channel(k + 2, 2 * g - 2, par); load_channel(k + 2, 2 * g - 1, par); }
                    } else {
#pragma unroll
                        for (int cl = 0; cl < 4; ++cl) load_channel(k + 2, 4 * g + cl, par);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            using Q = std::integral_constant<int, PAR ^ 1>;
            // outstanding, oldest first: chunk k+1's pixels | weights of chunk k+1 | chunk k+2's pixels
            wait_pixels(Q{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});
            post_max(Q{});
            lds_barrier();                                       // every wave has read P(k); the maxima of chunk k+1 are in LDS
            follow(read_max_exp(), false);
            convert(Q{});
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk k+1 landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    for (int k = 0; k < nchunk; k += 2) {
        chunk(k, P0{});
        if (k + 1 < nchunk) chunk(k + 1, P1{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the last prefetch (beyond Cin: zeros) has landed before registers are reused

    // ---------------- epilogue (conv_epilogue.h) ----------------
    {
        const float fo = x3h_pow2(-sh);
        const float* Wi = reinterpret_cast<const float*>(smem + Cfg::E_OFF) + 3 * MT;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const vr_f32x4h wi = *reinterpret_cast<const vr_f32x4h*>(Wi + mi * 32 + 8 * rq + 4 * khalf);
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mi][ni][rq * 4 + j] = (acc[mi][ni][rq * 4 + j] * fo) * wi[j];
            }
    }
    int hon[WN], won[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) { hon[ni] = h0 + 2 * (wave * WN + ni) + (l31 >> 4); won[ni] = l31 & 15; }
    epi_store<MT, WM, WN>(VR_EPI_ARGS(a), acc, reinterpret_cast<const float*>(smem + Cfg::E_OFF), n, co0, khalf,
                          h0 + TH <= a.Hout && a.Wout == 16, hon, won);
    // ---------------- BatchNorm partial statistics (training) ----------------
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                         // [4 waves][MT][2]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    if (hon[ni] < a.Hout && won[ni] < a.Wout) {
                        const float v = acc[mi][ni][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
                s1 = half_wave_sum_dpp(s1);
                s2 = half_wave_sum_dpp(s2);
                if (l31 == 16) {
                    const int m = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    red[(wave * MT + m) * 2 + 0] = s1;
                    red[(wave * MT + m) * 2 + 1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < MT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
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

// block -> (pixel tile, cout tile): every XCD walks its own contiguous range of pixel tiles (conv_x3h.hip)
__device__ __forceinline__ bool x3d_block(const ConvArgs& a, int& pt, int& ct) {
    const int id = blockIdx.x;
    const int xcd = id & 7, rr = id >> 3;
    ct = rr % a.nct;
    const int per_xcd = (a.npt + 7) >> 3;
    pt = xcd * per_xcd + rr / a.nct;
    return pt < a.npt;
}

template <int KK, int DH, int DW, int MT>
__global__ __launch_bounds__(256, 2) void conv_x3d_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_x3d[];
    int pt, ct;
    if (!x3d_block(a, pt, ct)) return;
    x3d_tile<KK, DH, DW, MT, 16>(a, pt, ct, smem_x3d);
}

// the four branch convs of one ASPP module: c[0] the 1x1 (conv2), c[1..3] the dilated 3x3 (conv3..conv5); same N, H, Cout, tiling
struct X3dAsppArgs { ConvArgs c[4]; };
template <int MT>
__global__ __launch_bounds__(256, 2) void conv_x3d_aspp_kernel(const X3dAsppArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_x3d[];
    int pt, ct;
    if (!x3d_block(g.c[1], pt, ct)) return;
    // the longest branch first in dispatch order (blockIdx.y = 0: dilation 12)
    if (blockIdx.y == 0) x3d_tile<9, 12, 6, MT, 16>(g.c[3], pt, ct, smem_x3d);
    else if (blockIdx.y == 1) x3d_tile<9, 8, 4, MT, 16>(g.c[2], pt, ct, smem_x3d);
    else if (blockIdx.y == 2) x3d_tile<9, 4, 2, MT, 16>(g.c[1], pt, ct, smem_x3d);
    else x3d_tile<1, 1, 1, MT, 16>(g.c[0], pt, ct, smem_x3d);
}

// True when conv_x3d takes the launch: 16-column image, 3x3 stride 1 with padding = dilation in {1, (4,2), (8,4), (12,6)} or 1x1,
// plain sources (eval: final activations; training: materialised), fp16-plane weights present.
bool x3d_pick(const ConvArgs& a, const ConvShape& s, int* MT_out) {
    static const int enabled = getenv("VR_CONV_X3D") ? atoi(getenv("VR_CONV_X3D")) : 1;
    if (!enabled || a.bf16 != 3 || !a.x3w || a.tapmask) return false;
    if (s.stride != 1 || a.Win != 16 || a.Wout != 16 || a.Hin != a.Hout) return false;
    if (s.KS == 3) {
        const bool dil = (s.dil_h == 1 && s.dil_w == 1) || (s.dil_h == 4 && s.dil_w == 2) || (s.dil_h == 8 && s.dil_w == 4) ||
                         (s.dil_h == 12 && s.dil_w == 6);
        if (!dil || a.pad_h != s.dil_h || a.pad_w != s.dil_w) return false;
    } else if (s.KS == 1) {
        if (a.pad_h != 0 || a.pad_w != 0) return false;
    } else return false;
    for (int i = 0; i < a.nsrc; ++i) {
        const ConvSrc& c = a.src[i];
        if (c.aff0 || c.aff1 || c.post || c.zins || c.up || c.slope != 1.f || c.W != 16 || c.H != a.Hin) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
    }
    static const int force_mt = getenv("VR_X3D_MT") ? atoi(getenv("VR_X3D_MT")) : 0;
    int MT = (a.CoutPad % 64 == 0) ? 64 : 32;
    // fewer than one workgroup per CU with 64 couts: halve the cout tile
    if (MT == 64 && (long long)a.N * ((a.Hout + 15) / 16) * (a.CoutPad / 64) < 256) MT = 32;
    if ((force_mt == 32 || force_mt == 64) && a.CoutPad % force_mt == 0) MT = force_mt;
    *MT_out = MT;
    return true;
}
void x3d_fill_tiling(ConvArgs& a, int MT) {
    a.tiles_w = 1;
    a.tiles_h = (a.Hout + 15) / 16;
    a.npt = a.N * a.tiles_h;
    a.nct = a.CoutPad / MT;
}

template <int KK, int DH, int DW, int MT>
static void x3d_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = X3dCfg<KK, DH, DW, MT, 16>;
    auto kern = conv_x3d_kernel<KK, DH, DW, MT>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    VR_LAUNCH(kern, dim3((unsigned)((a.npt + 7) / 8 * 8 * a.nct)), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}
template <int MT>
static void x3d_launch_mt(const ConvArgs& a, const ConvShape& s, hipStream_t st) {
    if (s.KS == 1) x3d_launch<1, 1, 1, MT>(a, st);
    else if (s.dil_h == 1) x3d_launch<9, 1, 1, MT>(a, st);
    else if (s.dil_h == 4) x3d_launch<9, 4, 2, MT>(a, st);
    else if (s.dil_h == 8) x3d_launch<9, 8, 4, MT>(a, st);
    else x3d_launch<9, 12, 6, MT>(a, st);
}
void x3d_launch_conv(const ConvArgs& a, const ConvShape& s, int MT, hipStream_t st) {
    if (MT == 64) x3d_launch_mt<64>(a, s, st);
    else x3d_launch_mt<32>(a, s, st);
}

// The four branches of an ASPP module in one launch; c[] in concat order (1x1, d = 4, 8, 12), tiling already filled (x3d_fill_tiling, one MT).
template <int MT>
static void x3d_launch_aspp_mt(const X3dAsppArgs& g, hipStream_t st) {
    auto kern = conv_x3d_aspp_kernel<MT>;
    constexpr int lds = X3dCfg<9, 12, 6, MT, 16>::LDS_BYTES;
    static_assert(lds >= X3dCfg<9, 8, 4, MT, 16>::LDS_BYTES && lds >= X3dCfg<1, 1, 1, MT, 16>::LDS_BYTES, "the widest halo sizes the LDS");
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds);
    const ConvArgs& a = g.c[1];
    VR_LAUNCH(kern, dim3((unsigned)((a.npt + 7) / 8 * 8 * a.nct), 4u), dim3(256), lds, st, g);
    VR_HIP(hipGetLastError());
}
bool x3d_aspp_eligible(const ConvArgs* c, const ConvShape* s) {
    static const bool on = !(getenv("VR_ASPP_FUSED") && atoi(getenv("VR_ASPP_FUSED")) == 0);
    if (!on) return false;
    int mt[4];
    for (int j = 0; j < 4; ++j) {
        if (!x3d_pick(c[j], s[j], &mt[j])) return false;
        if (c[j].N != c[0].N || c[j].Hout != c[0].Hout || c[j].CoutPad != c[0].CoutPad) return false;
    }
    return s[0].KS == 1 && s[1].KS == 3 && s[1].dil_h == 4 && s[2].KS == 3 && s[2].dil_h == 8 && s[3].KS == 3 && s[3].dil_h == 12;
}
void x3d_launch_aspp(const ConvArgs* c, hipStream_t st) {
    X3dAsppArgs g;
    // one cout tile for the four: the dilated branches decide (the launch is four times one branch's grid)
    int MT = (c[1].CoutPad % 64 == 0) ? 64 : 32;
    if (MT == 64 && (long long)c[1].N * ((c[1].Hout + 15) / 16) * (c[1].CoutPad / 64) * 4 < 512) MT = 32;
    static const int force_mt = getenv("VR_X3D_MT") ? atoi(getenv("VR_X3D_MT")) : 0;
    if ((force_mt == 32 || force_mt == 64) && c[1].CoutPad % force_mt == 0) MT = force_mt;
    for (int j = 0; j < 4; ++j) { g.c[j] = c[j]; x3d_fill_tiling(g.c[j], MT); }
    if (MT == 64) x3d_launch_aspp_mt<64>(g, st);
    else x3d_launch_aspp_mt<32>(g, st);
}

}  // namespace vr
