// Conv dispatcher (launch_conv) + the fused-LOADER convolution for the CascadedNet blocks
// (lib/layers.py:8-64, 67-105) on gfx950.
//
// launch_conv picks, in this order:  conv_wino.hip (Winograd F(2x2,3x3), plain inputs, 3x3 stride 1)  ->
// conv_dma.hip (direct implicit GEMM, plain inputs by LDS-DMA)  ->  conv_ws.hip / the kernel below (inputs that
// still carry a pending BatchNorm affine / activation / dropout / x2 upsample / zero insertion).  Eval mode and
// the materialised training path only produce plain inputs, so the kernel in this file is what remains for
// non-materialised training (VR_NO_TRAIN_MAT), 16-wide stride-2 data gradients and the unit-test hook.
//
// Structure: LDS-staged implicit GEMM.  A workgroup (4 waves) owns MT output channels x a
// TH x TW tile of output pixels of one image and walks the input channels in chunks of CK:
//   stage   the haloed input tile  Xs[CK][TH_in][TWp]   (global -> regs -> transform -> LDS)
//           the weight slice       Ws[KS*KS][CK][MT]
//   compute for each tap, for each channel pair: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain)
//           A[i=cout][k=ci pair] from Ws, B[k][j=pixel] from Xs, accumulators stay in registers.
// The staging step is where the fusion happens (vr_common.h): virtual channel-concat of up to
// three sources, the decoder's bilinear x2 upsample (align_corners=True), the producer's
// BatchNorm affine and ReLU/LeakyReLU, and conv zero padding.
// The epilogue stores the conv output (+ optional bias; eval: folded BatchNorm + activation) and, in
// training mode, per-block (sum, sumsq) partials per output channel for the BatchNorm batch statistics.
//
// Measured later in the round (tools/mfma_overlap.hip): the loader's VALU work is NOT hidden behind the
// fp32 MFMAs on gfx950 -- the reason the plain-input kernels exist.
#include <cstdlib>

#include "conv_stage.h"
#include "kernels.h"

namespace vr {

template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK, int WAVES_M>
struct ConvCfg {
    static constexpr int KK = KS * KS;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int NG = TH * TW / 32;
    static constexpr int WM = MT / 32 / WAVES_M;
    static constexpr int WN = NG / WAVES_N;
    static constexpr int TH_in = (TH - 1) * S + (KS - 1) * DH + 1;
    static constexpr int TW_in = (TW - 1) * S + (KS - 1) * DW + 1;
    // Row pitch in LDS.  TW=32: one row per 32-lane group -> any pitch is conflict-free.
    // TW=16: two rows per 32-lane group -> pitch == 16 (mod 32) keeps the two 16-lane runs on
    // disjoint banks (stride-1 case).
    static constexpr int TWp = (TW == 16) ? ((TW_in + 15) / 32 * 32 + 16) : ((TW_in + 1) & ~1);
    static constexpr int XS = CK * TH_in * TWp;
    static constexpr int WS = KK * CK * MT;
    static constexpr int STG = StageGeom<TH_in, TW_in>::TAB + StageGeom<TH_in, TW_in>::SCR;   // geometry tables + scratch
    static constexpr int LDS_BYTES = (XS + WS + STG) * 4;
    static_assert(WM >= 1 && WN >= 1 && WM * WAVES_M * 32 == MT && WN * WAVES_N == NG, "tile split");
    static_assert(XS % 4 == 0, "weight slab must stay 16B aligned");
    static_assert(TWp >= TW_in, "pitch");
};

template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK, int WAVES_M>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<KS, S, DH, DW, MT, TH, TW, CK, WAVES_M>;
    constexpr int KK = Cfg::KK, WM = Cfg::WM, WN = Cfg::WN, WAVES_N = Cfg::WAVES_N;
    constexpr int TH_in = Cfg::TH_in, TW_in = Cfg::TW_in, TWp = Cfg::TWp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + Cfg::XS;

    // ---- block -> (pixel tile, cout tile).  Blocks land on XCD (id % 8): keep all cout tiles
    // of one pixel tile on the same XCD, back to back, so the input tile is fetched into that
    // XCD's L2 once (placement is a speed assumption only).
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
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int khalf = lane >> 5;
    const int l31 = lane & 31;

    // per-lane B-operand base offsets (pixel of each owned group) inside one channel-pair slab
    int boff[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pix = (wn * WN + ni) * 32 + l31;
        const int r = pix / TW, c = pix % TW;
        boff[ni] = (khalf * TH_in + r * S) * TWp + c * S;
    }
    const int aoff = khalf * MT + wm * WM * 32 + l31;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int hbase = h0 * S - a.pad_h;
    const int wbase = w0 * S - a.pad_w;

    // ---- software pipeline (issue-early / write-late): the raw global loads of chunk k+1 are in
    // flight in registers while the MFMAs of chunk k run; the affine/activation/upsample transform
    // and the LDS writes happen after them.  One LDS buffer, two barriers per chunk.
    using SG = StageGeom<TH_in, TW_in>;
    constexpr int M4 = MT / 4;
    constexpr int NWV = CK * KK * M4;                 // float4 weight vectors per chunk
    constexpr int WP = (NWV + 255) / 256;
    constexpr int CPW = (CK + 3) / 4;
    int* tab = reinterpret_cast<int*>(smem + Cfg::XS + Cfg::WS);
    float* scratch = smem + Cfg::XS + Cfg::WS + SG::TAB + wave * SG::NL;
    float4 wv[WP];
    float raw[(S == 1 && DH == 1 && DW == 1) ? CPW : 1][SG::NPX];
    auto issue_weights = [&](int c0) {
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            int idx = tid + j * 256;
            idx = idx < NWV ? idx : NWV - 1;
            const int m4 = idx % M4;
            const int t2 = idx / M4;
            const int tap = t2 % KK;
            int ci = c0 + t2 / KK;
            ci = ci < a.Cin ? ci : a.Cin - 1;
            wv[j] = *reinterpret_cast<const float4*>(a.w + ((long long)ci * KK + tap) * a.CoutPad + co0 + m4 * 4);
        }
    };
    // Stride-2 and dilated tiles have 3x larger halos (18 load passes per channel): prefetching them
    // costs ~250 VGPRs, so those variants keep the synchronous staging.
    constexpr bool PIPE = (S == 1 && DH == 1 && DW == 1);
    if constexpr (PIPE) {
        issue_weights(0);
        build_stage_tables<TH_in, TW_in, TWp>(a, tab, hbase, wbase, tid);
        __syncthreads();
        issue_input_loads<TH_in, TW_in, CK, 4>(a, tab, 0, n, wave, lane, raw);
    }

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (a.dbg != 3) __syncthreads();   // previous chunk's MFMA reads are done
        if constexpr (!PIPE) issue_weights(c0);
        // ---------------- write stage: Ws[tap][cl][m] <- w[(c0+cl)][tap][co0+m], Xs <- transform(raw) ----
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int idx = tid + j * 256;
            if (idx < NWV) {
                const int m4 = idx % M4;
                const int t2 = idx / M4;
                const int tap = t2 % KK;
                const int cl = t2 / KK;
                const float4 v = (c0 + cl < a.Cin) ? wv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(Ws + (tap * CK + cl) * MT + m4 * 4) = v;
            }
        }
        if constexpr (PIPE) {
            if (a.dbg != 1 && a.dbg != 3) write_input_stage<TH_in, TW_in, TH_in * TWp, CK, 4>(a, tab, Xs, scratch, c0, n, wave, lane, raw);
        } else
            stage_input_chunk<TH_in, TW_in, TWp, TH_in * TWp, CK, 4>(a, Xs, c0, n, hbase, wbase, wave, lane);
        if (a.dbg != 3) __syncthreads();
        // ---------------- prefetch the next chunk (in flight during the MFMAs below) -----------------------
        // (unconditional: a conditional prefetch turns the register arrays into phi nodes and the
        // compiler then waits vmcnt(0) right here to copy them; the last iteration re-fetches its own chunk)
        if constexpr (PIPE) {
            const int cn = (c0 + CK < a.Cin) ? c0 + CK : c0;
            if (a.dbg != 1 && a.dbg != 3) {
                issue_weights(cn);
                issue_input_loads<TH_in, TW_in, CK, 4>(a, tab, cn, n, wave, lane, raw);
            }
        }
        // ---------------- MFMA over this chunk ---------------------------------------------------
        const int cleft = a.Cin - c0;
        const int npair = a.dbg == 2 ? 0 : ((cleft < CK ? cleft : CK) + 1) >> 1;
        if (npair == CK / 2) {
            // full chunk: fully unrolled so the scheduler can hoist the LDS operand reads of later
            // (tap, pair) steps above the MFMAs of earlier ones (hides the LDS latency)
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / KS, kw = tap % KS;
                const int toff = kh * DH * TWp + kw * DW;
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
                const int kh = tap / KS, kw = tap % KS;
                const int toff = kh * DH * TWp + kw * DW;
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
    }

    // ---------------- epilogue: raw store (+bias), up to three destination segments -------------------
    // C/D layout of 32x32 MFMA: col (pixel) = lane & 31, row (cout) = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wm * WM + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const float b = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
            float esc = 1.f, esh = 0.f, eslope = 1.f;      // eval: folded BatchNorm + activation
            if (a.epi) {
                const int cc = co < a.Cout ? co : a.Cout - 1;
                esc = a.epi[2 * cc]; esh = a.epi[2 * cc + 1]; eslope = a.epi_slope;
            }
            const int seg = (co >= a.d1) + (co >= a.d2);
            const int cod = co - (seg == 0 ? 0 : (seg == 1 ? a.d1 : a.d2));
            ConvDst d;      // field-wise select: a dynamic index into the kernarg would go through scratch
            d.p = seg == 0 ? a.dst[0].p : (seg == 1 ? a.dst[1].p : a.dst[2].p);
            d.sN = seg == 0 ? a.dst[0].sN : (seg == 1 ? a.dst[1].sN : a.dst[2].sN);
            d.sC = seg == 0 ? a.dst[0].sC : (seg == 1 ? a.dst[1].sC : a.dst[2].sC);
            d.sH = seg == 0 ? a.dst[0].sH : (seg == 1 ? a.dst[1].sH : a.dst[2].sH);
            d.accumulate = seg == 0 ? a.dst[0].accumulate : (seg == 1 ? a.dst[1].accumulate : a.dst[2].accumulate);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int pix = (wn * WN + ni) * 32 + l31;
                const int ho = h0 + pix / TW, wo = w0 + pix % TW;
                const float v = acc[mi][ni][r] + b;
                acc[mi][ni][r] = v;
                if (co < a.Cout && ho < a.Hout && wo < a.Wout && d.p) {
                    float* q = d.p + (long long)n * d.sN + (long long)cod * d.sC + (long long)ho * d.sH + wo;
                    const float y = act_apply(fmaf(v, esc, esh), eslope);
                    *q = d.accumulate ? *q + y : y;
                }
            }
        }
    }

    // ---------------- BatchNorm partial statistics (training) -----------------------------------------
    if (a.part) {
        __syncthreads();                       // all waves finished reading Xs/Ws
        float* red = smem;                     // [WAVES_N][MT][2]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int pix = (wn * WN + ni) * 32 + l31;
                    const int ho = h0 + pix / TW, wo = w0 + pix % TW;
                    if (ho < a.Hout && wo < a.Wout) {
                        const float v = acc[mi][ni][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {   // stays inside each 32-lane half
                    s1 += __shfl_xor(s1, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (l31 == 0) {
                    const int m = (wm * WM + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    red[(wn * MT + m) * 2 + 0] = s1;
                    red[(wn * MT + m) * 2 + 1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < MT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_N; ++w) {
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

// -------------------------------------------------------------------------------------------------
// Launcher: pick the instantiation from the layer shape.
// -------------------------------------------------------------------------------------------------
struct TileChoice { int MT, TH, TW; };

static TileChoice pick_tile(const ConvArgs& a, const ConvShape& s) {
    TileChoice t;
    const bool dilated = (s.dil_h != 1 || s.dil_w != 1);
    t.TW = (a.Wout >= 32 && !dilated) ? 32 : 16;
    t.TH = (t.TW == 32) ? 8 : 16;
    t.MT = (a.CoutPad % 64 == 0) ? 64 : 32;
    if (t.MT == 64) {
        // small grids: halve the cout tile to double the number of workgroups
        long long tiles = (long long)a.N * ((a.Hout + t.TH - 1) / t.TH) * ((a.Wout + t.TW - 1) / t.TW);
        if (tiles * (a.CoutPad / 64) < 512) t.MT = 32;
    }
    if (t.MT == 32 && t.TW == 32 && s.stride == 1 && !dilated) {
        // thin layers (Cout <= 32, the full-resolution ones): a taller pixel tile doubles the MFMA
        // work per staged chunk and halves the halo overhead
        long long tiles16 = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
        if (tiles16 * (a.CoutPad / 32) >= 1024) t.TH = 16;
    }
    return t;
}

bool ws_pick(const ConvArgs& a, const ConvShape& s, int* MT_out, int* TH_out);
bool thin16_pick(const ConvArgs& a, const ConvShape& s, int* TH);
void thin16_fill_tiling(ConvArgs& a, int TH);
void thin16_launch_conv(const ConvArgs& a, const ConvShape& s, int TH, hipStream_t st);
bool wino_pick(const ConvArgs& a, const ConvShape& s, int* MT_out);
void wino_fill_tiling(ConvArgs& a, int MT);
void wino_launch_conv(const ConvArgs& a, int MT, hipStream_t st);
struct DmaTile { int MT, TH, TW; };
bool dma_pick(const ConvArgs& a, const ConvShape& s, DmaTile* t);
void dma_fill_tiling(ConvArgs& a, const DmaTile& t);
void dma_launch_conv(const ConvArgs& a, const ConvShape& s, const DmaTile& t, hipStream_t st);
void ws_fill_tiling(ConvArgs& a, int MT, int TH);
void ws_launch_conv(const ConvArgs& a, const ConvShape& s, int MT, int TH, hipStream_t st);

void conv_fill_tiling(ConvArgs& a, const ConvShape& s) {
    int wmt, wth;
    int wino_mt, thin_th;
    if (thin16_pick(a, s, &thin_th)) { thin16_fill_tiling(a, thin_th); return; }
    X3Tile x3t;
    if (x3_pick(a, s, &x3t)) { x3_fill_tiling(a, x3t); return; }
    int x3d_mt;
    if (x3d_pick(a, s, &x3d_mt)) { x3d_fill_tiling(a, x3d_mt); return; }
    if (wino_pick(a, s, &wino_mt)) { wino_fill_tiling(a, wino_mt); return; }
    DmaTile dt;
    if (dma_pick(a, s, &dt)) { dma_fill_tiling(a, dt); return; }
    if (ws_pick(a, s, &wmt, &wth)) { ws_fill_tiling(a, wmt, wth); return; }
    TileChoice t = pick_tile(a, s);
    a.tiles_w = (a.Wout + t.TW - 1) / t.TW;
    a.tiles_h = (a.Hout + t.TH - 1) / t.TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / t.MT;
}

size_t conv_part_count(const ConvArgs& a, const ConvShape& s) {
    ConvArgs b = a;
    conv_fill_tiling(b, s);
    return (size_t)b.npt;
}

template <int KS, int S, int DH, int DW, int MT, int TH, int TW, int CK, int WAVES_M>
static void launch_inst(const ConvArgs& a, hipStream_t st) {
    using Cfg = ConvCfg<KS, S, DH, DW, MT, TH, TW, CK, WAVES_M>;
    auto kern = conv_mfma_kernel<KS, S, DH, DW, MT, TH, TW, CK, WAVES_M>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    const int grid = groups * 8 * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int KS, int S, int DH, int DW, int CK>
static void launch_by_tile(const ConvArgs& a, const TileChoice& t, hipStream_t st) {
    if (t.TW == 32) {
        if constexpr (DH == 1 && DW == 1) {
            if (t.MT == 64) launch_inst<KS, S, DH, DW, 64, 8, 32, CK, 1>(a, st);
            else if (t.TH == 16) {
                if constexpr (S == 1) launch_inst<KS, S, DH, DW, 32, 16, 32, CK, 1>(a, st);
                else throw Error(-2, "16x32 tiles are stride-1 only");
            }
            else            launch_inst<KS, S, DH, DW, 32, 8, 32, CK, 1>(a, st);
        } else {
            throw Error(-2, "dilated conv uses TW=16 tiles only");
        }
    } else {
        if (t.MT == 64) launch_inst<KS, S, DH, DW, 64, 16, 16, CK, 1>(a, st);
        else            launch_inst<KS, S, DH, DW, 32, 16, 16, CK, 1>(a, st);
    }
}

double launch_conv(const ConvArgs& a_in, const ConvShape& s, hipStream_t st) {
    ConvArgs a = a_in;
    static const int dbg = getenv("VR_CONV_DBG") ? atoi(getenv("VR_CONV_DBG")) : 0;
    a.dbg = dbg;
    {
        int wmt, wth;
        int wino_mt, thin_th;
        if (a.nsrc >= 1 && a.nsrc <= 3 && thin16_pick(a, s, &thin_th)) {
            thin16_fill_tiling(a, thin_th);
            thin16_launch_conv(a, s, thin_th, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
        X3Tile x3t;
        if (a.nsrc >= 1 && a.nsrc <= 3 && x3_pick(a, s, &x3t)) {
            x3_fill_tiling(a, x3t);
            x3_launch_conv(a, x3t, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
        int x3d_mt;
        if (a.nsrc >= 1 && a.nsrc <= 3 && x3d_pick(a, s, &x3d_mt)) {
            x3d_fill_tiling(a, x3d_mt);
            x3d_launch_conv(a, s, x3d_mt, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
        if (a.nsrc >= 1 && a.nsrc <= 3 && wino_pick(a, s, &wino_mt)) {
            wino_fill_tiling(a, wino_mt);
            wino_launch_conv(a, wino_mt, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
        DmaTile dt;
        if (dma_pick(a, s, &dt)) {
            VR_CHECK(a.nsrc >= 1 && a.nsrc <= 3, -2, "conv takes 1..3 sources");
            dma_fill_tiling(a, dt);
            dma_launch_conv(a, s, dt, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
        VR_CHECK(a.tapmask == 0, -2, "a tap-masked conv needs the LDS-DMA kernel (conv_dma_eligible)");
        if (ws_pick(a, s, &wmt, &wth)) {
            VR_CHECK(a.nsrc >= 1 && a.nsrc <= 3, -2, "conv takes 1..3 sources");
            ws_fill_tiling(a, wmt, wth);
            ws_launch_conv(a, s, wmt, wth, st);
            return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
        }
    }
    conv_fill_tiling(a, s);
    TileChoice t = pick_tile(a, s);
    VR_CHECK(a.CoutPad % t.MT == 0 && a.CoutPad % 32 == 0, -2, "CoutPad must be a multiple of the cout tile");
    VR_CHECK(a.nsrc >= 1 && a.nsrc <= 3, -2, "conv takes 1..3 sources");
    if (s.KS == 1) {
        VR_CHECK(s.stride == 1 && s.dil_h == 1 && s.dil_w == 1, -2, "1x1 conv: stride/dilation must be 1");
        launch_by_tile<1, 1, 1, 1, 32>(a, t, st);
    } else if (s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1) {
        launch_by_tile<3, 1, 1, 1, 8>(a, t, st);
    } else if (s.KS == 3 && s.stride == 2 && s.dil_h == 1 && s.dil_w == 1) {
        launch_by_tile<3, 2, 1, 1, 4>(a, t, st);
    } else if (s.KS == 3 && s.stride == 1 && s.dil_h == 4 && s.dil_w == 2) {
        launch_by_tile<3, 1, 4, 2, 4>(a, t, st);
    } else if (s.KS == 3 && s.stride == 1 && s.dil_h == 8 && s.dil_w == 4) {
        launch_by_tile<3, 1, 8, 4, 4>(a, t, st);
    } else if (s.KS == 3 && s.stride == 1 && s.dil_h == 12 && s.dil_w == 6) {
        launch_by_tile<3, 1, 12, 6, 4>(a, t, st);
    } else {
        throw Error(-2, "unsupported conv shape (kernel/stride/dilation)");
    }
    return 2.0 * a.N * (double)a.Hout * a.Wout * (double)a.Cout * a.Cin * s.KS * s.KS;
}

}  // namespace vr
