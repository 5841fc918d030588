// Warp-specialised form of the fused-loader convolution (3x3 stride 1/2 and 1x1, >= 32 output columns) for
// inputs that still carry a pending BatchNorm affine / activation / upsample (see conv_mfma.hip for when
// that happens; plain inputs take conv_wino.hip / conv_dma.hip).
//
// A wave issues in order, so a wave that runs the loader and the MFMAs never overlaps the two.  Here the
// workgroup splits the roles:
//   waves 0-3   consumers: nothing but LDS operand reads + MFMAs (+ the epilogue)
//   waves 4-..  producers (8-12): prefetch raw tiles of chunk k+2 into registers, transform chunk k+1 into
//               the other LDS buffer (same table-driven loader as conv_mfma.hip)
// with ONE workgroup barrier per input-channel chunk.  This hides the loader's memory LATENCY; its VALU
// instructions still take SIMD time away from the MFMAs (measured: tools/mfma_overlap.hip), which caps
// this kernel at 88-100 TFLOP/s on the big layers (profiles/README.md).
#include <cstdlib>

#include "conv_stage.h"

namespace vr {

template <int KS, int S, int MT, int TH, int TW, int CK, int NPW>
struct WsCfg {
    static constexpr int KK = KS * KS;
    static constexpr int NG = TH * TW / 32;
    static constexpr int WM = MT / 32;            // every consumer wave owns all MT couts ...
    static constexpr int WN = NG / 4;             // ... for its quarter of the pixel groups
    static constexpr int TH_in = (TH - 1) * S + KS, TW_in = (TW - 1) * S + KS;
    static constexpr int TWp = (TW_in + 1) & ~1;
    static constexpr int XS = CK * TH_in * TWp;
    static constexpr int WS = KK * CK * MT;
    using SG = StageGeom<TH_in, TW_in>;
    static constexpr int BUF = XS + WS;           // one (input tile, weight slice) buffer
    static constexpr int LDS_FLOATS = 2 * BUF + SG::TAB + NPW * SG::NL;
    static constexpr int THREADS = 256 + 64 * NPW;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(WN >= 1 && WN * 4 == NG, "pixel groups must split over 4 consumer waves");
    static_assert(BUF % 4 == 0 && XS % 4 == 0, "16B alignment of the LDS slabs");
};

template <int KS, int S, int MT, int TH, int TW, int CK, int NPW>
__global__ __launch_bounds__(256 + 64 * NPW) void conv_ws_kernel(const ConvArgs a) {
    using Cfg = WsCfg<KS, S, MT, TH, TW, CK, NPW>;
    using SG = typename Cfg::SG;
    constexpr int KK = Cfg::KK, WM = Cfg::WM, WN = Cfg::WN, TH_in = Cfg::TH_in, TW_in = Cfg::TW_in, TWp = Cfg::TWp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab = reinterpret_cast<int*>(smem + 2 * Cfg::BUF);

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    const int pt = (rr / a.nct) * 8 + xcd;
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
    const int hbase = h0 * S - a.pad_h, wbase = w0 * S - a.pad_w;
    const int nchunk = (a.Cin + CK - 1) / CK;

    build_stage_tables<TH_in, TW_in, TWp>(a, tab, hbase, wbase, tid & 255);   // both halves write the same values
    __syncthreads();

    if (wave >= 4) {
        // =========================== producers ===========================================================
        if (a.dbg == 5) __builtin_amdgcn_s_setprio(3);
        const int pw = wave - 4, ptid = tid - 256;
        float* scratch = smem + 2 * Cfg::BUF + SG::TAB + pw * SG::NL;
        constexpr int M4 = MT / 4;
        constexpr int NWV = CK * KK * M4;
        constexpr int PT = 64 * NPW;                      // producer threads
        constexpr int WP = (NWV + PT - 1) / PT;
        constexpr int CPW = (CK + NPW - 1) / NPW;
        float4 wv[WP];
        float raw[CPW][SG::NPX];
        auto issue_weights = [&](int c0) {
#pragma unroll
            for (int j = 0; j < WP; ++j) {
                int idx = ptid + j * PT;
                idx = idx < NWV ? idx : NWV - 1;
                const int m4 = idx % M4;
                const int t2 = idx / M4;
                const int tap = t2 % KK;
                int ci = c0 + t2 / KK;
                ci = ci < a.Cin ? ci : a.Cin - 1;
                wv[j] = *reinterpret_cast<const float4*>(a.w + ((long long)ci * KK + tap) * a.CoutPad + co0 + m4 * 4);
            }
        };
        auto write_stage = [&](int k) {
            float* Xs = smem + (k & 1) * Cfg::BUF;
            float* Ws = Xs + Cfg::XS;
            const int c0 = k * CK;
#pragma unroll
            for (int j = 0; j < WP; ++j) {
                const int idx = ptid + j * PT;
                if (idx < NWV) {
                    const int m4 = idx % M4;
                    const int t2 = idx / M4;
                    const int tap = t2 % KK;
                    const int cl = t2 / KK;
                    const float4 v = (c0 + cl < a.Cin) ? wv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(Ws + (tap * CK + cl) * MT + m4 * 4) = v;
                }
            }
            write_input_stage<TH_in, TW_in, TH_in * TWp, CK, NPW>(a, tab, Xs, scratch, c0, n, pw, lane, raw);
        };
        issue_weights(0);
        issue_input_loads<TH_in, TW_in, CK, NPW>(a, tab, 0, n, pw, lane, raw);
        write_stage(0);
        {
            const int cn = nchunk > 1 ? CK : 0;
            issue_weights(cn);
            issue_input_loads<TH_in, TW_in, CK, NPW>(a, tab, cn, n, pw, lane, raw);
        }
        __syncthreads();                                   // buffer 0 ready
        for (int k = 0; k < nchunk; ++k) {
            if (k + 1 < nchunk && a.dbg != 1) write_stage(k + 1);        // regs hold chunk k+1 (issued one phase ago)
            if (a.dbg != 1) {
                const int kn = (k + 2 < nchunk) ? k + 2 : nchunk - 1;   // unconditional (no register phis)
                issue_weights(kn * CK);
                issue_input_loads<TH_in, TW_in, CK, NPW>(a, tab, kn * CK, n, pw, lane, raw);
            }
            // buffer (k+1)&1 ready, buffer k&1 free.  Raw barrier: __syncthreads() would also wait
            // vmcnt(0), i.e. stall the whole workgroup on the prefetch loads just issued; they are
            // consumed one phase later and the compiler's own counted wait covers them there.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (a.part) { __syncthreads(); __syncthreads(); }  // keep pace with the consumers' statistics epilogue
        return;
    }

    // =============================== consumers ===============================================================
    // static priority: the MFMA waves win issue arbitration over the producer waves sharing their SIMD
    if (a.dbg != 4 && a.dbg != 5) __builtin_amdgcn_s_setprio(2);
    const int khalf = lane >> 5, l31 = lane & 31;
    int boff[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pix = (wave * WN + ni) * 32 + l31;
        const int r = pix / TW, c = pix % TW;
        boff[ni] = (khalf * TH_in + r * S) * TWp + c * S;
    }
    const int aoff = khalf * MT + l31;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    __syncthreads();                                       // buffer 0 ready
    for (int k = 0; k < nchunk; ++k) {
        const float* Xs = smem + (k & 1) * Cfg::BUF;
        const float* Ws = Xs + Cfg::XS;
        const int cleft = a.Cin - k * CK;
        const int npair = ((cleft < CK ? cleft : CK) + 1) >> 1;
        if (npair == CK / 2) {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int toff = (tap / KS) * TWp + (tap % KS);
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk) {
                    float av[WM], bv[WN];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[(tap * CK + 2 * kk) * MT + aoff + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[2 * kk * TH_in * TWp + toff + boff[ni]];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int toff = (tap / KS) * TWp + (tap % KS);
                for (int kk = 0; kk < npair; ++kk) {
                    float av[WM], bv[WN];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[(tap * CK + 2 * kk) * MT + aoff + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = Xs[2 * kk * TH_in * TWp + toff + boff[ni]];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---------------- epilogue: raw store (+bias), up to three destination segments ---------------------
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const float b = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
            float esc = 1.f, esh = 0.f, eslope = 1.f;      // eval: folded BatchNorm + activation
            if (a.epi) {
                const int cc = co < a.Cout ? co : a.Cout - 1;
                esc = a.epi[2 * cc]; esh = a.epi[2 * cc + 1]; eslope = a.epi_slope;
            }
            const int seg = (co >= a.d1) + (co >= a.d2);
            const int cod = co - (seg == 0 ? 0 : (seg == 1 ? a.d1 : a.d2));
            float* dp = seg == 0 ? a.dst[0].p : (seg == 1 ? a.dst[1].p : a.dst[2].p);
            const long long dN = seg == 0 ? a.dst[0].sN : (seg == 1 ? a.dst[1].sN : a.dst[2].sN);
            const long long dC = seg == 0 ? a.dst[0].sC : (seg == 1 ? a.dst[1].sC : a.dst[2].sC);
            const long long dH = seg == 0 ? a.dst[0].sH : (seg == 1 ? a.dst[1].sH : a.dst[2].sH);
            const int dacc = seg == 0 ? a.dst[0].accumulate : (seg == 1 ? a.dst[1].accumulate : a.dst[2].accumulate);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int pix = (wave * WN + ni) * 32 + l31;
                const int ho = h0 + pix / TW, wo = w0 + pix % TW;
                const float v = acc[mi][ni][r] + b;
                acc[mi][ni][r] = v;
                if (co < a.Cout && ho < a.Hout && wo < a.Wout && dp) {
                    float* q = dp + (long long)n * dN + (long long)cod * dC + (long long)ho * dH + wo;
                    const float y = act_apply(fmaf(v, esc, esh), eslope);
                    *q = dacc ? *q + y : y;
                }
            }
        }
    }
    // ---------------- BatchNorm partial statistics (training) -------------------------------------------------
    if (a.part) {
        __syncthreads();
        float* red = smem;                     // [4 consumer waves][MT][2]
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
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    s1 += __shfl_xor(s1, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (l31 == 0) {
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
template <int KS, int S, int MT, int TH, int TW, int CK, int NPW = 4>
static void ws_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = WsCfg<KS, S, MT, TH, TW, CK, NPW>;
    auto kern = conv_ws_kernel<KS, S, MT, TH, TW, CK, NPW>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    const int grid = groups * 8 * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

// Returns false when the shape is not covered (caller falls back to conv_mfma.hip).
bool ws_pick(const ConvArgs& a, const ConvShape& s, int* MT_out, int* TH_out) {
    static const int enabled = getenv("VR_CONV_WS") ? atoi(getenv("VR_CONV_WS")) : 1;
    if (!enabled) return false;
    if (s.dil_h != 1 || s.dil_w != 1) return false;
    if (!((s.KS == 3 && (s.stride == 1 || s.stride == 2)) || (s.KS == 1 && s.stride == 1))) return false;
    if (a.Wout < 32) return false;
    static const int wide = getenv("VR_WS_NPW") ? atoi(getenv("VR_WS_NPW")) : 12;
    int MT = (a.CoutPad % 128 == 0) ? 128 : ((a.CoutPad % 64 == 0) ? 64 : 32);
    if ((s.stride == 2 || (s.KS == 3 && wide == 12)) && MT == 128) MT = 64;
    const long long tiles = (long long)a.N * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    // keep the grid >= ~2 workgroups per CU where the layer allows it
    while (MT > 32 && tiles * (a.CoutPad / MT) < 512) MT /= 2;
    if (tiles * (a.CoutPad / MT) < 128) return false;       // tiny grids: the 256-thread kernel has more blocks
    int TH = 8;
    if (MT == 32 && s.stride == 1) {
        // thin layers: a taller tile doubles the MFMA work per staged chunk (keeps the producers hidden)
        const long long tiles16 = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
        if (tiles16 * (a.CoutPad / 32) >= 1024) TH = 16;
    }
    *MT_out = MT;
    *TH_out = TH;
    return true;
}

void ws_fill_tiling(ConvArgs& a, int MT, int TH) {
    a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_h = (a.Hout + TH - 1) / TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / MT;
}

void ws_launch_conv(const ConvArgs& a, const ConvShape& s, int MT, int TH, hipStream_t st) {
    if (s.KS == 3 && s.stride == 1) {
        static const int npw12 = getenv("VR_WS_NPW") ? atoi(getenv("VR_WS_NPW")) : 12;
        if (npw12 == 12) {
            if (MT == 64) ws_launch<3, 1, 64, 8, 32, 8, 8>(a, st);
            else if (TH == 16) ws_launch<3, 1, 32, 16, 32, 8, 8>(a, st);
            else ws_launch<3, 1, 32, 8, 32, 12, 12>(a, st);
        } else {
            if (MT == 128) ws_launch<3, 1, 128, 8, 32, 4>(a, st);
            else if (MT == 64) ws_launch<3, 1, 64, 8, 32, 8>(a, st);
            else if (TH == 16) ws_launch<3, 1, 32, 16, 32, 8>(a, st);
            else ws_launch<3, 1, 32, 8, 32, 8>(a, st);
        }
    } else if (s.KS == 3) {
        if (MT == 64) ws_launch<3, 2, 64, 8, 32, 4>(a, st);
        else ws_launch<3, 2, 32, 8, 32, 4>(a, st);
    } else {
        if (MT == 128) ws_launch<1, 1, 128, 8, 32, 16>(a, st);
        else if (MT == 64) ws_launch<1, 1, 64, 8, 32, 32>(a, st);
        else if (TH == 16) ws_launch<1, 1, 32, 16, 32, 32>(a, st);
        else ws_launch<1, 1, 32, 8, 32, 32>(a, st);
    }
}

}  // namespace vr
