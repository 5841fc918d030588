// LDS-DMA convolution: the forward conv of lib/layers.py:12-20 (3x3 stride 1/2, dilated 3x3, 1x1) for
// inputs that are PLAIN tensors (no pending BatchNorm affine / activation / dropout / upsample), which
// is every conv of the eval-mode network once the producer applies BatchNorm+activation in its epilogue.
//
// Why a third conv kernel: on gfx950 the fp32 MFMA and the VALU of a SIMD do not overlap
// (tools/mfma_overlap.hip: MFMA-only 5.0 ms + VALU-only 10.7 ms = 14.7 ms together), so every VALU
// instruction of a loader is paid in full no matter which wave runs it.  A plain input needs no
// arithmetic at all: `buffer_load_dwordx4 ... lds` moves 16 B per lane straight from HBM/L2 into
// LDS (lane-linear destination, per-lane source offset, out-of-range lanes land as zeros = the
// conv's zero padding), issued by the four MFMA waves themselves -- ~4 instructions per wave per
// input-channel chunk, no producer waves, no staging registers, no ds_write.
//
// LDS image per chunk:  Xs[CK][TH_in][TWq]  (row = image columns [w0*S-pad-XS0, ...) so that every
// 4-float piece is 16-B aligned in the image row; a piece is either fully inside the row or fully
// padding because Win % 4 == 0), Ws[tap][CK][MT] (a contiguous re-tiling of w[ci][tap][co0..]).
// Double-buffered, ONE raw s_barrier per chunk; the DMA of chunk k+1 is in flight while chunk k is
// multiplied.  The DMA is inline asm on purpose: the compiler would wait vmcnt(0) before the first
// LDS read that follows a DMA it knows about (cdna_hip_programming.md "Pipelining across barriers").
#include <cstdlib>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

struct DmaTile { int MT, TH, TW; };

template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK>
struct DmaCfg {
    static constexpr int KK = KS * KS;
    static constexpr int NG = TH * TW / 32;
    static constexpr int WM = MT / 32;
    static constexpr int WN = NG / 4;
    static constexpr int TH_in = (TH - 1) * S + (KS - 1) * DH + 1;
    static constexpr int TW_in = (TW - 1) * S + (KS - 1) * DW + 1;
    static constexpr int PADW = DW * (KS - 1) / 2, PADH = DH * (KS - 1) / 2;
    static constexpr int XS0 = (4 - PADW % 4) % 4;            // tile column 0 = image column w0*S - PADW - XS0 (= 0 mod 4)
    static constexpr int TWn = ((XS0 + TW_in + 3) / 4) * 4;   // columns that carry data
    // TW == 16: a 32-lane operand read covers two tile rows -> pitch = 16 (mod 32) keeps them on disjoint banks
    static constexpr int TWq = (TW == 16) ? ((TWn + 15) / 32 * 32 + 16) : TWn;
    static constexpr int CSX = TH_in * TWq;                   // channel pitch
    static constexpr int XS = CK * CSX;
    static constexpr int WS = KK * CK * MT;
    static constexpr int BUF = XS + WS;
    static constexpr int NPIECE = CSX / 4;                    // 16-B pieces per input channel
    static constexpr int NPASS = (NPIECE + 63) / 64;
    static constexpr int NWP = WS / 4;                        // 16-B pieces of the weight slice
    static constexpr int NWPASS = (NWP + 255) / 256;          // per wave
    static constexpr int CPW = CK / 4;                        // input channels per wave
    static constexpr int NS = KK * (CK / 2);                  // MFMA k-steps per chunk
    // epilogue constants of the cout tile (conv_epilogue.h): behind the two buffers -- unless those 3*MT floats would cost a
    // workgroup per CU (the 1x1 tilings with exactly 40 KB); then they are fetched at the end and parked over buffer 0
    static constexpr bool E_TAIL = (160 * 1024) / (2 * BUF * 4) == (160 * 1024) / ((2 * BUF + 3 * MT) * 4);
    static constexpr int E_OFF = E_TAIL ? 2 * BUF : 0;
    static constexpr int LDS_BYTES = (2 * BUF + (E_TAIL ? 3 * MT : 0)) * 4;
    static_assert(WN >= 1 && WN * 4 == NG, "pixel groups must split over the 4 waves");
    static_assert(CK % 4 == 0 && BUF % 4 == 0 && XS % 4 == 0 && CSX % 4 == 0, "16-B LDS slabs");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// Waves per SIMD the register allocation must leave room for: the accumulators need (MT/32) x (pixel groups per wave) x 16
// registers; without a bound hipcc spends 200+ registers on these kernels (one or two workgroups per CU) although their LDS
// footprint allows three to seven -- and the LDS-DMA latency is hidden by other resident workgroups, not inside a wave.
template <int MT, int TH, int TW>
struct DmaOcc {
    static constexpr int ACC = (MT / 32) * (TH * TW / 128) * 16;
    static constexpr int value = ACC <= 32 ? 4 : (ACC <= 64 ? 3 : (ACC <= 96 ? 2 : 1));
};

template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK, bool TM = false>
__global__ __launch_bounds__(256, (DmaOcc<MT, TH, TW>::value)) void conv_dma_kernel(const ConvArgs a) {
    using Cfg = DmaCfg<KS, S, DH, DW, MT, TH, TW, CK>;
    constexpr int KK = Cfg::KK, WM = Cfg::WM, WN = Cfg::WN, TWq = Cfg::TWq, TWn = Cfg::TWn, CSX = Cfg::CSX,
                  XS0 = Cfg::XS0, NPIECE = Cfg::NPIECE, NPASS = Cfg::NPASS, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS,
                  CPW = Cfg::CPW, NS = Cfg::NS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // each XCD walks its own contiguous range of pixel tiles: neighbouring tiles share halo lines through that XCD's L2
    // (conv_x3.hip; measured there: HBM fetch 2.8x -> 1.1x of the input on the full-resolution layers); VR_CONV_DBG=16: interleaved
    const int pt = (a.dbg & 16) ? (rr / a.nct) * 8 + xcd : xcd * ((a.npt + 7) >> 3) + rr / a.nct;
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
    const int hbase = h0 * S - Cfg::PADH, wal0 = w0 * S - Cfg::PADW - XS0;
    const int nchunk = (a.Cin + CK - 1) / CK;
    const int nfull = a.Cin / CK;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- per-lane source coordinates of the input pieces: voffset = hrow * (4*sH of the source) + wcol4 ----
    // (padding / pitch-filler pieces: hrow = 0, wcol4 = 2^31 -> beyond the descriptor -> zeros, no traffic)
    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        const int hh = q / (TWq / 4), j = q % (TWq / 4);
        const int hi = hbase + hh, wi = wal0 + 4 * j;
        const bool ok = q < NPIECE && 4 * j < TWn && hi >= 0 && hi < a.Hin && wi >= 0 && wi + 3 < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // ---- per-lane offsets of the weight pieces: LDS order [tap][cl][m], source w[(cl*KK+tap)*CoutPad + m] ----
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int m4 = q % (MT / 4), t2 = q / (MT / 4);
        const int cl = t2 % CK, tap = t2 / CK;
        woff[i] = (unsigned)(((cl * KK + tap) * a.CoutPad + m4 * 4) * 4);
    }

    const bool uniform_on = !(a.dbg & 8);                 // (VR_CONV_DBG=8: the per-channel form everywhere, for A / B)
    auto issue_chunk = [&](int k) {
        const int c0 = k * CK;
        const unsigned xs_b = lds0 + (unsigned)((k & 1) * Cfg::BUF * 4);
        const unsigned ws_b = xs_b + Cfg::XS * 4;
        // weights: rows of channels >= Cin are out of range of the descriptor -> zeros
        {
            const float* wb = a.w + (long long)c0 * KK * a.CoutPad + co0;
            const i32x4 wr = make_rsrc(wb, (unsigned)(((long long)(a.Cin - c0) * KK * a.CoutPad - co0) * 4));
#pragma unroll
            for (int i = 0; i < NWPASS; ++i) {
                const int pp = wave + 4 * i;
                if (pp * 64 + lane < NWP) dma16(ws_b + pp * 1024, woff[i], wr);
            }
        }
        // The source fields as opaque VALUES: hipcc may turn a select between the loads a.src[si].x into one load from a run-time
        // address inside the kernel-argument struct -- and then keeps the whole struct in scratch (seen on a 1x1 tiling).
        long long f0N = a.src[0].sN, f1N = a.src[1].sN, f2N = a.src[2].sN, f0C = a.src[0].sC, f1C = a.src[1].sC, f2C = a.src[2].sC;
        long long f0H = a.src[0].sH, f1H = a.src[1].sH, f2H = a.src[2].sH;
        unsigned long long f0p = (unsigned long long)a.src[0].p, f1p = (unsigned long long)a.src[1].p, f2p = (unsigned long long)a.src[2].p;
        asm volatile("" : "+s"(f0N), "+s"(f1N), "+s"(f2N), "+s"(f0C), "+s"(f1C), "+s"(f2C));
        asm volatile("" : "+s"(f0H), "+s"(f1H), "+s"(f2H), "+s"(f0p), "+s"(f1p), "+s"(f2p));
        // Round 6: a chunk whose CK channels are all live and come from ONE source (nearly all of them) takes one descriptor -- the plane
        // of its first channel -- and reaches channel cl through the SCALAR offset of the DMA (not part of the range check on gfx9, so
        // padding lanes still read zeros).  The per-channel form below costs ~50 scalar instructions per channel (source select, 64-bit
        // address arithmetic, descriptor, settle): eight channels per wave and chunk in the 1x1 tilings, beside 16 matrix instructions.
        // (1x1 tilings only -- measured: 16-wide 1x1 0.51 -> 0.33 ms, 32-wide 0.28 -> 0.24 ms per inference step; with ONE channel per wave and
        // chunk, the 3x3 stride-2 tilings were 6 - 15 % slower with it)
        if constexpr (KS == 1) {
            const int s0 = (c0 >= a.c1) + (c0 >= a.c2), s1 = (c0 + CK - 1 >= a.c1) + (c0 + CK - 1 >= a.c2);
            if (uniform_on && c0 + CK <= a.Cin && s0 == s1) {
                const int clc0 = c0 - (s0 == 0 ? 0 : (s0 == 1 ? a.c1 : a.c2));
                const float* sp = reinterpret_cast<const float*>(s0 == 0 ? f0p : (s0 == 1 ? f1p : f2p));
                const long long sN = s0 == 0 ? f0N : (s0 == 1 ? f1N : f2N);
                const long long sC = s0 == 0 ? f0C : (s0 == 1 ? f1C : f2C);
                const unsigned sH4 = (unsigned)(s0 == 0 ? f0H : (s0 == 1 ? f1H : f2H)) * 4u;
                if ((unsigned long long)sC * 4ull * (unsigned long long)CK < 0x7FFFFFF0ull) {
                    const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)clc0 * sC, 0x7FFFFFF0u);
                    const unsigned sC4 = (unsigned)(sC * 4);
                    unsigned vo[NPASS];
#pragma unroll
                    for (int p = 0; p < NPASS; ++p) vo[p] = hrow[p] * sH4 + wcol4[p];
#pragma unroll
                    for (int cc = 0; cc < CPW; ++cc) {
                        const int cl = wave + 4 * cc;
                        const unsigned cb = xs_b + (unsigned)(cl * CSX * 4);
                        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)cl * sC4));
#pragma unroll
                        for (int p = 0; p < NPASS; ++p) {
                            if ((p + 1) * 64 <= NPIECE) dma16s(cb + p * 1024, vo[p], xr, so);
                            else if (p * 64 + lane < NPIECE) dma16s(cb + p * 1024, vo[p], xr, so);
                        }
                    }
                    return;
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int cl = wave + 4 * cc;
            const int ci = c0 + cl;                       // wave-uniform
            if (ci >= a.Cin) {
                float* z = smem + (k & 1) * Cfg::BUF + cl * CSX;
                for (int e = lane; e < CSX; e += 64) z[e] = 0.f;
                continue;
            }
            const int si = (ci >= a.c1) + (ci >= a.c2);
            const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
            const float* sp = reinterpret_cast<const float*>(si == 0 ? f0p : (si == 1 ? f1p : f2p));
            const long long sN = si == 0 ? f0N : (si == 1 ? f1N : f2N);
            const long long sC = si == 0 ? f0C : (si == 1 ? f1C : f2C);
            const unsigned sH4 = (unsigned)(si == 0 ? f0H : (si == 1 ? f1H : f2H)) * 4u;
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

    const int khalf = lane >> 5, l31 = lane & 31;
    int boff[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pix = (wave * WN + ni) * 32 + l31;
        const int r = pix / TW, c = pix % TW;
        boff[ni] = khalf * CSX + (r * S) * TWq + c * S + XS0;
    }
    const int aoff = khalf * MT + l31;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if constexpr (Cfg::E_TAIL) {
        float ecv[3];
        epi_fetch<MT>(a.bias, a.epi, a.Cout, co0, tid, ecv);
        if (a.dbg != 1) issue_chunk(0);
        epi_park<MT>(smem + Cfg::E_OFF, tid, ecv);
    } else {
        if (a.dbg != 1) issue_chunk(0);
    }
    dma_wait_and_barrier();

    // ---- full chunks: software-pipelined (the LDS operands of step s+1 are read before the MFMAs of step s)
    for (int k = 0; k < nfull; ++k) {
        if (k + 1 < nchunk && a.dbg != 1) issue_chunk(k + 1);     // buffer (k+1)&1 was released by the last barrier
        const float* Xs = smem + (k & 1) * Cfg::BUF;
        const float* Ws = Xs + Cfg::XS;
        if constexpr (TM) {
            // tap-masked form (parity class of a stride-2 data gradient): only the live taps are multiplied
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                if (!((a.tapmask >> tap) & 1)) continue;                 // wave-uniform
                const int toff = (tap / KS) * DH * TWq + (tap % KS) * DW;
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk) {
                    float av[WM], bv[WN];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[(tap * CK + 2 * kk) * MT + aoff + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[2 * kk * CSX + toff + boff[ni]];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        } else if (a.dbg != 2) {
            float av[WM], bv[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[aoff + mi * 32];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[boff[ni]];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float avn[WM], bvn[WN];
                if (s + 1 < NS) {
                    const int tap = (s + 1) / (CK / 2), kk = (s + 1) % (CK / 2);
                    const int toff = (tap / KS) * DH * TWq + (tap % KS) * DW;
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) avn[mi] = Ws[(tap * CK + 2 * kk) * MT + aoff + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bvn[ni] = Xs[2 * kk * CSX + toff + boff[ni]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NS) {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = avn[mi];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = bvn[ni];
                }
            }
        }
        dma_wait_and_barrier();
    }
    // ---- the partial last chunk (Cin % CK channels; the pair's odd channel is zero-filled) ----------------
    if (nfull < nchunk) {
        const float* Xs = smem + (nfull & 1) * Cfg::BUF;
        const float* Ws = Xs + Cfg::XS;
        const int npair = (a.Cin - nfull * CK + 1) >> 1;
        if (a.dbg != 2) {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                if (TM && !((a.tapmask >> tap) & 1)) continue;
                const int toff = (tap / KS) * DH * TWq + (tap % KS) * DW;
                for (int kk = 0; kk < npair; ++kk) {
                    float av[WM], bv[WN];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[(tap * CK + 2 * kk) * MT + aoff + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[2 * kk * CSX + toff + boff[ni]];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
    }

    // ---------------- epilogue (conv_epilogue.h): bias, (eval) BatchNorm+activation, up to three destination segments -----
    {
        int hon[WN], won[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int pix = (wave * WN + ni) * 32 + l31;
            hon[ni] = h0 + pix / TW; won[ni] = w0 + pix % TW;
        }
        if constexpr (!Cfg::E_TAIL) {                          // (one extra memory round trip per tile, 40-KB tilings only)
            float ecv[3];
            epi_fetch<MT>(a.bias, a.epi, a.Cout, co0, tid, ecv);
            __syncthreads();                                   // every wave is done with buffer 0
            epi_park<MT>(smem, tid, ecv);
            __syncthreads();
        }
        epi_store<MT, WM, WN, (WM * WN >= 4 ? 4 : 8)>(VR_EPI_ARGS(a), acc, smem + Cfg::E_OFF, n, co0, khalf, h0 + TH <= a.Hout && w0 + TW <= a.Wout, hon, won);
    }
    // ---------------- BatchNorm partial statistics (training) -------------------------------------------------
    if (a.part) {
        __syncthreads();
        float* red = smem;                     // [4 waves][MT][2]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int pix = (wave * WN + ni) * 32 + l31;
                    const int ho = h0 + pix / TW, wo = w0 + pix % TW;
                    if (ho < a.Hout && wo < a.Wout) {
                        const float v = acc[mi][ni][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
                s1 = half_wave_sum_dpp(s1);
                s2 = half_wave_sum_dpp(s2);
                if (l31 == 16) {                                   // (the sums are complete in lanes 16-31 / 48-63)
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

// -------------------------------------------------------------------------------------------------------
template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK, bool TM = false>
static void dma_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = DmaCfg<KS, S, DH, DW, MT, TH, TW, CK>;
    auto kern = conv_dma_kernel<KS, S, DH, DW, MT, TH, TW, CK, TM>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    const int grid = groups * 8 * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

static bool src_plain(const ConvSrc& s) {
    return !s.aff0 && !s.aff1 && !s.post && !s.up && !s.zins && s.slope == 1.f;
}

// -------------------------------------------------------------------------------------------------------
// Data gradient of a stride-2 3x3 conv, all four output-parity classes in ONE launch (train.hip: run_conv_backward).
//   dx[2i+ph][2j+pw] = sum over the class's live taps of  w_cls[tap] * dz[i + th - 1][j + tw - 1],   th in {1} (ph = 0) or {1,2} (ph = 1)
// Every (class, tap) pair is one of nine: the launch stages the dz tile and the nine weight slices ONCE per input-channel
// chunk and runs nine MFMA sets into four accumulator sets -- the work of one 3x3 conv -- where the four tap-masked launches
// (conv_dma_kernel<..., TM>) each staged the full tile and all nine slices for 1 / 2 / 2 / 4 live taps and waited on the DMA
// 74 % of the time (profiles/r02_train_sq_pmc.md).  32 couts x (8 x 32) dz pixels per workgroup = 16 x 64 output pixels;
// the two column parities of an output row leave as one 8-byte read-modify-write (the destinations accumulate).
template <int CK>
__global__ __launch_bounds__(256, 3) void conv_dma_s2d_kernel(const ConvArgs a) {
    constexpr int MT = 32, TH = 8, TW = 32;
    using Cfg = DmaCfg<3, 1, 1, 1, MT, TH, TW, CK>;
    constexpr int WN = Cfg::WN, TWq = Cfg::TWq, TWn = Cfg::TWn, CSX = Cfg::CSX, XS0 = Cfg::XS0, NPIECE = Cfg::NPIECE,
                  NPASS = Cfg::NPASS, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS, CPW = Cfg::CPW;
    // the nine (class, tap) pairs: class = ph * 2 + pw, tap = th * 3 + tw
    constexpr int PCLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    constexpr int PTAP[9] = {4, 4, 5, 4, 7, 4, 5, 7, 8};
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // each XCD walks its own contiguous range of pixel tiles: neighbouring tiles share halo lines through that XCD's L2
    // (conv_x3.hip; measured there: HBM fetch 2.8x -> 1.1x of the input on the full-resolution layers); VR_CONV_DBG=16: interleaved
    const int pt = (a.dbg & 16) ? (rr / a.nct) * 8 + xcd : xcd * ((a.npt + 7) >> 3) + rr / a.nct;
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
    const int hbase = h0 - 1, wal0 = w0 - 1 - XS0;
    const int nchunk = a.Cin / CK;                      // (Cin % CK == 0: eligibility)
    const unsigned lds0 = (unsigned)(size_t)smem;

    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        const int hh = q / (TWq / 4), j = q % (TWq / 4);
        const int hi = hbase + hh, wi = wal0 + 4 * j;
        const bool ok = q < NPIECE && 4 * j < TWn && hi >= 0 && hi < a.Hin && wi >= 0 && wi + 3 < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // weight pieces: LDS order [pair][cl][m], source class array PCLS[pair], w_cls[(cl * 9 + PTAP[pair]) * CoutPad + m]
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int m4 = q % (MT / 4), t2 = q / (MT / 4);
        const int cl = t2 % CK, pr = t2 / CK;
        int cls = 0, tap = 4;
#pragma unroll
        for (int e = 0; e < 9; ++e)
            if (pr == e) { cls = PCLS[e]; tap = PTAP[e]; }
        woff[i] = (unsigned)(((long long)cls * a.s2_cls_stride + ((long long)(cl * 9 + tap) * a.CoutPad + m4 * 4)) * 4);
    }
    const unsigned w_bytes = (unsigned)((3 * a.s2_cls_stride + (long long)a.Cin * 9 * a.CoutPad) * 4);

    auto issue_chunk = [&](int k) {
        const int c0 = k * CK;
        const unsigned xs_b = lds0 + (unsigned)((k & 1) * Cfg::BUF * 4);
        const unsigned ws_b = xs_b + Cfg::XS * 4;
        {
            const long long boff = (long long)c0 * 9 * a.CoutPad + co0;
            const i32x4 wr = make_rsrc(a.w + boff, w_bytes - (unsigned)(boff * 4));
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
            const i32x4 xr = make_rsrc(a.src[0].p + (long long)n * a.src[0].sN + (long long)ci * a.src[0].sC, 0x7FFFFFF0u);
            const unsigned sH4 = (unsigned)a.src[0].sH * 4u;
            const unsigned cb = xs_b + (unsigned)(cl * CSX * 4);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const unsigned vo = hrow[p] * sH4 + wcol4[p];
                if ((p + 1) * 64 <= NPIECE) dma16(cb + p * 1024, vo, xr);
                else if (p * 64 + lane < NPIECE) dma16(cb + p * 1024, vo, xr);
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    int boff[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pix = (wave * WN + ni) * 32 + l31;
        boff[ni] = khalf * CSX + (pix / TW) * TWq + pix % TW + XS0;
    }
    const int aoff = khalf * MT + l31;
    f32x16 acc[4][WN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][ni][r] = 0.f;

    issue_chunk(0);
    dma_wait_and_barrier();
    constexpr int NS = 9 * (CK / 2);
    for (int k = 0; k < nchunk; ++k) {
        if (k + 1 < nchunk) issue_chunk(k + 1);
        const float* Xs = smem + (k & 1) * Cfg::BUF;
        const float* Ws = Xs + Cfg::XS;
        float av = Ws[aoff], bv[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[(PTAP[0] / 3) * TWq + PTAP[0] % 3 + boff[ni]];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float avn = 0.f, bvn[WN];
            if (s + 1 < NS) {
                const int pr = (s + 1) / (CK / 2), kk = (s + 1) % (CK / 2);
                const int toff = (PTAP[pr] / 3) * TWq + PTAP[pr] % 3;
                avn = Ws[(pr * CK + 2 * kk) * MT + aoff];
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) bvn[ni] = Xs[2 * kk * CSX + toff + boff[ni]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
                acc[PCLS[s / (CK / 2)]][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[ni], acc[PCLS[s / (CK / 2)]][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) {
                av = avn;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) bv[ni] = bvn[ni];
            }
        }
        dma_wait_and_barrier();
    }

    // epilogue: dx[2 ho + ph][2 wo + pw] += acc[ph * 2 + pw]; the pw pair of a row is one 8-byte access when aligned.
    // Two couts (one accumulator row per lane half) at a time; a group inside the image, inside Cout and inside one 8-byte
    // aligned destination runs straight-line -- all old values loaded before the first store (conv_epilogue.h has the why).
    const bool tile_in = 2 * (h0 + TH) <= a.s2_H && 2 * (w0 + TW) <= a.s2_W;
    long long offq[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pix = (wave * WN + ni) * 32 + l31;
        offq[ni] = ((long long)(h0 + pix / TW) << 32) | (unsigned)(w0 + pix % TW);
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        // row rg of lane half khalf: cout cg0 + 4 khalf
        constexpr int GR = 1;
        const int cg0 = co0 + (rg & 3) + 8 * (rg >> 2);
        const int seg0 = (cg0 >= a.d1) + (cg0 >= a.d2), seg1 = (cg0 + 4 >= a.d1) + (cg0 + 4 >= a.d2);
        float* const Dp = (seg0 == 0 ? a.dst[0].p : (seg0 == 1 ? a.dst[1].p : a.dst[2].p));
        const long long sN = (seg0 == 0 ? a.dst[0].sN : (seg0 == 1 ? a.dst[1].sN : a.dst[2].sN));
        const long long sC = (seg0 == 0 ? a.dst[0].sC : (seg0 == 1 ? a.dst[1].sC : a.dst[2].sC));
        const long long sH = (seg0 == 0 ? a.dst[0].sH : (seg0 == 1 ? a.dst[1].sH : a.dst[2].sH));
        const int gacc = (seg0 == 0 ? a.dst[0].accumulate : (seg0 == 1 ? a.dst[1].accumulate : a.dst[2].accumulate));
        const bool al8g = ((reinterpret_cast<size_t>(Dp) | (size_t)(sN * 4) | (size_t)(sC * 4) | (size_t)(sH * 4)) & 7) == 0;
        if (tile_in && cg0 + 5 <= a.Cout && seg0 == seg1 && Dp != nullptr && al8g) {
            const int cseg = seg0 == 0 ? 0 : (seg0 == 1 ? a.d1 : a.d2);
            float* qb = Dp + (long long)n * sN + (long long)(cg0 - cseg + 4 * khalf) * sC;
            vr_f32x2 old[GR][WN][2];
            if (gacc) {
#pragma unroll
                for (int j = 0; j < GR; ++j)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
                            old[j][ni][ph] = *reinterpret_cast<const vr_f32x2*>(qb + j * sC + (2 * (offq[ni] >> 32) + ph) * sH + 2 * (int)(offq[ni] & 0xffffffffll));
            } else {
#pragma unroll
                for (int j = 0; j < GR; ++j)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) { old[j][ni][ph][0] = 0.f; old[j][ni][ph][1] = 0.f; }
            }
#pragma unroll
            for (int j = 0; j < GR; ++j)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        vr_f32x2 o;
                        o[0] = acc[ph * 2][ni][rg + j] + old[j][ni][ph][0];
                        o[1] = acc[ph * 2 + 1][ni][rg + j] + old[j][ni][ph][1];
                        *reinterpret_cast<vr_f32x2*>(qb + j * sC + (2 * (offq[ni] >> 32) + ph) * sH + 2 * (int)(offq[ni] & 0xffffffffll)) = o;
                    }
            continue;
        }
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const int r = rg + j;
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (co >= a.Cout) continue;
            const int seg = (co >= a.d1) + (co >= a.d2);
            const int cod = co - (seg == 0 ? 0 : (seg == 1 ? a.d1 : a.d2));
            float* dp = seg == 0 ? a.dst[0].p : (seg == 1 ? a.dst[1].p : a.dst[2].p);
            if (!dp) continue;
            const long long dN = seg == 0 ? a.dst[0].sN : (seg == 1 ? a.dst[1].sN : a.dst[2].sN);
            const long long dC = seg == 0 ? a.dst[0].sC : (seg == 1 ? a.dst[1].sC : a.dst[2].sC);
            const long long dH = seg == 0 ? a.dst[0].sH : (seg == 1 ? a.dst[1].sH : a.dst[2].sH);
            const int dacc = seg == 0 ? a.dst[0].accumulate : (seg == 1 ? a.dst[1].accumulate : a.dst[2].accumulate);
            const bool al8 = ((reinterpret_cast<size_t>(dp) | (size_t)(dN * 4) | (size_t)(dC * 4) | (size_t)(dH * 4)) & 7) == 0;
            float* base = dp + (long long)n * dN + (long long)cod * dC;
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int ho = (int)(offq[ni] >> 32), wo = (int)(offq[ni] & 0xffffffffll);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int hf = 2 * ho + ph, wf = 2 * wo;
                    if (hf >= a.s2_H || wf >= a.s2_W) continue;
                    float* q = base + (long long)hf * dH + wf;
                    const float v0 = acc[ph * 2][ni][r], v1 = acc[ph * 2 + 1][ni][r];
                    if (al8 && wf + 1 < a.s2_W) {
                        vr_f32x2* q2 = reinterpret_cast<vr_f32x2*>(q);
                        vr_f32x2 o;
                        o[0] = v0; o[1] = v1;
                        if (dacc) { const vr_f32x2 old = *q2; o[0] += old[0]; o[1] += old[1]; }
                        *q2 = o;
                    } else {
                        q[0] = dacc ? q[0] + v0 : v0;
                        if (wf + 1 < a.s2_W) q[1] = dacc ? q[1] + v1 : v1;
                    }
                }
            }
        }
    }
}

// Fused stride-2 data gradient: dz is a single plain source, weights = the four class arrays of launch_s2_class_weights.
bool s2d_fused_eligible(const ConvArgs& a) {
    static const int enabled = getenv("VR_S2D_FUSED") ? atoi(getenv("VR_S2D_FUSED")) : 1;
    // (16-wide dz, the 1/16-resolution encoders: half of each 32-column tile is padding -- zero-filled by the loader's range check and
    // skipped by the stores -- and the launch still runs ~2x faster than the zero-insertion fallback of conv_ws.hip: 16-23 TFLOP/s)
    static const int min_w = getenv("VR_S2D_MIN_W") ? atoi(getenv("VR_S2D_MIN_W")) : 16;
    if (!enabled || a.nsrc != 1 || (a.Cin & 3) || a.Wout < min_w || (a.Win & 3)) return false;
    const ConvSrc& c = a.src[0];
    if (!src_plain(c) || c.W != a.Win) return false;
    if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
    if ((3 * a.s2_cls_stride + (long long)a.Cin * 9 * a.CoutPad) * 4 >= 0x7FFFFFF0LL) return false;
    return true;
}

void launch_s2d_fused(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_h = (a.Hout + 7) / 8;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / 32;
    using Cfg = DmaCfg<3, 1, 1, 1, 32, 8, 32, 4>;
    auto kern = conv_dma_s2d_kernel<4>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}


// True when the launch can take the LDS-DMA kernel; fills the tile choice (MT, TH, TW).
bool dma_pick(const ConvArgs& a, const ConvShape& s, DmaTile* t) {
    static const int enabled = getenv("VR_CONV_DMA") ? atoi(getenv("VR_CONV_DMA")) : 1;
    if (!enabled) return false;
    const bool dil = s.dil_h != 1 || s.dil_w != 1;
    if (dil) {
        if (!(s.KS == 3 && s.stride == 1)) return false;
        if (!((s.dil_h == 4 && s.dil_w == 2) || (s.dil_h == 8 && s.dil_w == 4) || (s.dil_h == 12 && s.dil_w == 6)))
            return false;
    } else if (!((s.KS == 3 && (s.stride == 1 || s.stride == 2)) || (s.KS == 1 && s.stride == 1))) {
        return false;
    }
    if (a.pad_h != s.dil_h * (s.KS - 1) / 2 || a.pad_w != s.dil_w * (s.KS - 1) / 2) return false;
    if (a.Wout < 16 || (a.Win & 3)) return false;
    for (int i = 0; i < a.nsrc; ++i) {
        const ConvSrc& c = a.src[i];
        if (!src_plain(c) || c.W != a.Win) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
    }
    if ((long long)a.Cin * s.KS * s.KS * a.CoutPad * 4 >= 0x7FFFFFF0LL) return false;
    if (a.tapmask && (a.Wout < 32 || dil || s.KS != 3 || s.stride != 1)) return false;
    if (a.Wout < 32 || dil) {
        // 1/16-resolution layers: 16x16 pixel tiles, 32 couts per workgroup (the grids are small); 8x16 tiles when even
        // that leaves most of the 256 CUs without a second workgroup
        t->TW = 16; t->TH = 16; t->MT = 32;
        static const int min16 = getenv("VR_DMA_MIN16") ? atoi(getenv("VR_DMA_MIN16")) : 800;     // (measured: conv time of the step -2.8 %)
        const long long wgs = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 15) / 16) * (a.CoutPad / 32);
        if (wgs < min16) t->TH = 8;
        return true;
    }
    int MT = (a.CoutPad % 128 == 0) ? 128 : ((a.CoutPad % 64 == 0) ? 64 : 32);
    if (s.stride == 2 && MT == 128) MT = 64;
    const long long tiles = (long long)a.N * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    static const int minblk = getenv("VR_DMA_MINBLK") ? atoi(getenv("VR_DMA_MINBLK")) : 768;
    while (MT > 32 && tiles * (a.CoutPad / MT) < minblk) MT /= 2;
    int TH = 8;
    if (MT == 32 && s.stride == 1) {
        const long long tiles16 = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
        if (tiles16 * (a.CoutPad / 32) >= 1024) TH = 16;
    }
    t->MT = MT; t->TH = TH; t->TW = 32;
    return true;
}

bool conv_dma_eligible(const ConvArgs& a, const ConvShape& s) {
    DmaTile t;
    return dma_pick(a, s, &t);
}

void dma_fill_tiling(ConvArgs& a, const DmaTile& t) {
    a.tiles_w = (a.Wout + t.TW - 1) / t.TW;
    a.tiles_h = (a.Hout + t.TH - 1) / t.TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / t.MT;
}

void dma_launch_conv(const ConvArgs& a, const ConvShape& s, const DmaTile& t, hipStream_t st) {
    const int MT = t.MT, TH = t.TH;
    if (t.TW == 16 && TH == 8) {
        if (s.KS == 1) dma_launch<1, 1, 1, 1, 32, 8, 16, 32>(a, st);
        else if (s.stride == 2) dma_launch<3, 2, 1, 1, 32, 8, 16, 4>(a, st);
        else if (s.dil_h == 1) dma_launch<3, 1, 1, 1, 32, 8, 16, 8>(a, st);
        else if (s.dil_h == 4) dma_launch<3, 1, 4, 2, 32, 8, 16, 4>(a, st);
        else if (s.dil_h == 8) dma_launch<3, 1, 8, 4, 32, 8, 16, 4>(a, st);
        else dma_launch<3, 1, 12, 6, 32, 8, 16, 4>(a, st);
    } else if (t.TW == 16) {
        if (s.KS == 1) dma_launch<1, 1, 1, 1, 32, 16, 16, 32>(a, st);
        else if (s.stride == 2) dma_launch<3, 2, 1, 1, 32, 16, 16, 4>(a, st);
        else if (s.dil_h == 1) dma_launch<3, 1, 1, 1, 32, 16, 16, 8>(a, st);
        else if (s.dil_h == 4) dma_launch<3, 1, 4, 2, 32, 16, 16, 4>(a, st);
        else if (s.dil_h == 8) dma_launch<3, 1, 8, 4, 32, 16, 16, 4>(a, st);
        else dma_launch<3, 1, 12, 6, 32, 16, 16, 4>(a, st);
    } else if (s.KS == 3 && s.stride == 1 && a.tapmask) {
        if (MT == 128) dma_launch<3, 1, 1, 1, 128, 8, 32, 4, true>(a, st);
        else if (MT == 64) dma_launch<3, 1, 1, 1, 64, 8, 32, 4, true>(a, st);
        else if (TH == 16) dma_launch<3, 1, 1, 1, 32, 16, 32, 4, true>(a, st);
        else dma_launch<3, 1, 1, 1, 32, 8, 32, 4, true>(a, st);
    } else if (s.KS == 3 && s.stride == 1) {
        static const int ck4 = getenv("VR_DMA_CK4") ? atoi(getenv("VR_DMA_CK4")) : 7;   // tuning experiment
        if (MT == 128) dma_launch<3, 1, 1, 1, 128, 8, 32, 4>(a, st);
        else if (MT == 64) { if (ck4 & 1) dma_launch<3, 1, 1, 1, 64, 8, 32, 4>(a, st); else dma_launch<3, 1, 1, 1, 64, 8, 32, 8>(a, st); }
        else if (TH == 16) { if (ck4 & 2) dma_launch<3, 1, 1, 1, 32, 16, 32, 4>(a, st); else dma_launch<3, 1, 1, 1, 32, 16, 32, 8>(a, st); }
        else { if (ck4 & 4) dma_launch<3, 1, 1, 1, 32, 8, 32, 4>(a, st); else dma_launch<3, 1, 1, 1, 32, 8, 32, 8>(a, st); }
    } else if (s.KS == 3) {
        if (MT == 64) dma_launch<3, 2, 1, 1, 64, 8, 32, 4>(a, st);
        else dma_launch<3, 2, 1, 1, 32, 8, 32, 4>(a, st);
    } else {
        if (MT == 128) dma_launch<1, 1, 1, 1, 128, 8, 32, 16>(a, st);
        else if (MT == 64) dma_launch<1, 1, 1, 1, 64, 8, 32, 16>(a, st);
        else if (TH == 16) dma_launch<1, 1, 1, 1, 32, 16, 32, 16>(a, st);
        else dma_launch<1, 1, 1, 1, 32, 8, 32, 16>(a, st);
    }
}

}  // namespace vr
