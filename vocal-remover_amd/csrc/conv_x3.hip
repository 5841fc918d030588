// Direct 3x3 stride-1 convolution with fp32 products on the bf16 matrix pipe ("mfma_mode" 2; lib/layers.py:12-20).
//
// gfx950 multiplies fp32 operands on v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate (1/16 of the bf16 matrix rate), so an
// fp32-exact product assembled from six bf16 products (conv_stage.h: x = x1 + x2 + x3, three exact bf16 planes per operand)
// costs 3 x 32 cycles per 8 input channels where the fp32 instruction needs 4 x 64: a 2.67x higher multiply roof for the
// DIRECT convolution -- above what Winograd F(2x2,3x3) buys on the fp32 pipe (2.25x), with no input / output transforms,
// 9/16 of the weight bytes and no 16-frequency exchange in the epilogue.
//
// Workgroup = 256 threads / 4 waves, output tile TH x 32 pixels x MT couts, input channels in chunks of 8:
//   * each thread owns up to NPASS pixels ("slots") of the (TH+2) x 34 halo tile: it loads the 8 channels of its pixel with
//     plain buffer_load_dword (one coalesced row segment per wave instruction; conv zero padding = out-of-range offsets),
//     issued BEFORE the MFMA phase of the previous chunk so the latency is covered, splits the four channel pairs into the three
//     bf16 planes (11 VALU per pair) and stores  P[plane][row][col][8 ch]  = one 16-byte MFMA operand per pixel and plane;
//   * the weights arrive pre-split (x3_weights_kernel: [chunk][tap][plane][cout][8 ch]) by LDS-DMA, double-buffered;
//   * multiply phase = ds_read_b128 + MFMA only.  With  A = [a1|a1], [a2|a2], [a3|a1]  and  B = [b1|b2], [b1|b3]  (lanes 0-31
//     hold k = 0..7, lanes 32-63 k = 8..15: both halves carry the SAME 8 channels of two planes)
//         [a1|a1][b1|b2] + [a2|a2][b1|b2] + [a3|a1][b1|b3] = a1b1 + a1b2 + a2b1 + a2b2 + a3b1 + a1b3
//     per tap: 3 WM + 2 WN operand reads feed 3 WM WN instructions; operands of tap t+1 are read before the MFMAs of tap t;
//   * two barriers per chunk (P is single-buffered: 16-29 KB; the two weight buffers take 28-55 KB) -- two or three workgroups
//     per CU overlap one's split pass with the others' multiply phases.
// The epilogue is conv_dma.hip's (same 32x32 accumulator layout): bias, folded BatchNorm + activation (eval), up to three
// destination segments, BatchNorm partial sums (training).
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

// The pixel loads are inline asm and their waits are placed by hand: hipcc's wait-count pass loses the issue order of loads that
// cross a loop back edge / uniform branches and then waits for (nearly) everything, i.e. also for the loads issued for the chunk
// after next -- the prefetch depth the register sets pay for.
__device__ __forceinline__ float x3_load(i32x4 rsrc, int voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
// s_waitcnt vmcnt(N) that the uses of the eight registers cannot be scheduled across
template <int N>
__device__ __forceinline__ void x3_wait8(float (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N) : "memory");
}

template <int MT, int TH>
struct X3Cfg {
    static constexpr int TW = 32, CK = 8, KK = 9;
    static constexpr int TH_in = TH + 2, PW = TW + 2;           // halo tile, pixels
    static constexpr int NSLOT = TH_in * PW;
    static constexpr int NPASS = (NSLOT + 255) / 256;
    static constexpr int WM = MT / 32, WN = TH / 4;
    static constexpr int PLANE = NSLOT * 16;                     // bytes of one bf16 plane (8 channels per pixel)
    static constexpr int P_BYTES = 3 * PLANE;
    static constexpr int NWP = KK * 3 * MT;                      // 16-byte weight operands per chunk
    static constexpr int W_BYTES = NWP * 16;
    static constexpr int NWPASS = (NWP + 255) / 256;
    // fused bilinear x2 (lib/layers.py:52): the low-resolution pixels under the halo tile, [8 ch][LROWS][LW] fp32
    static constexpr int LROWS = TH / 2 + 3, LW = 20, LSLOT = LROWS * LW;
    static constexpr int L_OFF = P_BYTES + 2 * W_BYTES;
    static constexpr int L_BYTES = CK * LSLOT * 4;
    static constexpr int E_OFF = L_OFF + L_BYTES;               // epilogue constants of the cout tile: bias, scale, shift [3][MT] fp32
    static constexpr int LDS_BYTES = E_OFF + 3 * MT * 4;
    static constexpr int OCC = 3 * LDS_BYTES <= 160 * 1024 ? 3 : 2;   // workgroups per CU the register budget must allow
    // vector-memory operations a wave issues per chunk: 8 * NPASS pixel loads (always, also beyond Cin: empty descriptor), and at
    // least NWMIN weight DMAs (the last wave-instruction of the weight slab may be empty for some waves)
    static constexpr int NXL = 8 * NPASS, NWMIN = (NWP / 64) / 4;
    static_assert(TH % 4 == 0 && MT % 32 == 0 && LDS_BYTES <= 80 * 1024 && 2 * NXL + NWMIN < 64 && NXL <= 30 && LSLOT <= 256, "tile");
};

template <int MT, int TH>
__global__ __launch_bounds__(256, (X3Cfg<MT, TH>::OCC)) void conv_x3_kernel(const ConvArgs a) {
    using Cfg = X3Cfg<MT, TH>;
    constexpr int TW = Cfg::TW, KK = Cfg::KK, PW = Cfg::PW, NSLOT = Cfg::NSLOT, NPASS = Cfg::NPASS, WM = Cfg::WM, WN = Cfg::WN,
                  PLANE = Cfg::PLANE, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS;
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];
    char* const Pb = smem_x3;

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // Block b runs on XCD b % 8 (observed dispatch order).  Every XCD walks its OWN contiguous, row-major range of pixel tiles, so
    // the tiles resident on an XCD at any time are neighbours: the 128-byte lines that horizontally adjacent tiles share (a 34-pixel
    // halo row spans three lines) and the halo rows of vertically adjacent ones come from the XCD's L2 (measured: -8 % on the
    // full-resolution layers against tiles interleaved over the XCDs, VR_CONV_DBG=16).
    const int per_xcd = (a.npt + 7) >> 3;
    const int pt = (a.dbg & 16) ? (rr / a.nct) * 8 + xcd : xcd * per_xcd + rr / a.nct;
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
    const int nchunk = (a.Cin + 7) >> 3;
    const unsigned lds0 = (unsigned)(size_t)smem_x3;
    const int dbg = a.dbg & 15;

    // ---- this thread's pixels of the halo tile: byte offset in a channel plane = hrow * (4 * sH) + wcol4 (2^31: padding) ----
    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int s = p * 256 + tid;
        const int r = s / PW, c = s - r * PW;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // ---- sources that arrive through the decoder's bilinear x2 (align_corners=True; eval: the upsample is not materialised):
    // this thread's low-resolution pixel of the staging tile, and for each of its halo pixels the position inside that tile
    // and the two interpolation weights (upsampled sources share their geometry: model.hip) ----
    const ConvSrc& us = a.src[0].up ? a.src[0] : (a.src[1].up ? a.src[1] : a.src[2]);
    const bool any_up = a.src[0].up | a.src[1].up | a.src[2].up;
    int lrow = 0, lcol4 = 0;                       // low-res pixel this thread fetches (byte column; 2^31: none)
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
            const int s = p * 256 + tid;
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
    // ---- weight operands: LDS order [tap][plane][m], source x3w[chunk][(tap * 3 + plane) * CoutPad + co0 + m] ----
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int m = q % MT, tp = q / MT;
        woff[i] = (unsigned)((tp * a.CoutPad + m) * 16);
    }
    const long long wchunk_bytes = (long long)KK * 3 * a.CoutPad * 16;
    auto issue_w = [&](int k) {                                    // the weight DMA of chunk k: NWPASS wave-instructions
        const char* wb = static_cast<const char*>(a.x3w) + k * wchunk_bytes + (long long)co0 * 16;
        const i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), (unsigned)(wchunk_bytes - (long long)co0 * 16));
        const unsigned ws_b = lds0 + (unsigned)(Cfg::P_BYTES + (k & 1) * Cfg::W_BYTES);
#pragma unroll
        for (int i = 0; i < NWPASS; ++i) {
            const int pp = wave + 4 * i;
            if ((pp + 1) * 64 <= NWP) dma16(ws_b + pp * 1024, woff[i], wr);
            else if (pp * 64 + lane < NWP) dma16(ws_b + pp * 1024, woff[i], wr);
        }
    };
    // The channels are visited strictly in order (chunk by chunk), so the source of the virtual concat is a running scalar
    // state: pointer to the current channel's plane, its channel / row strides, the first channel of the next source.
    const float* xp = a.src[0].p + (long long)n * a.src[0].sN;
    long long xsC = a.src[0].sC;
    unsigned xsH4 = (unsigned)a.src[0].sH * 4u;
    int xend = a.c1, xsi = 0;
    bool xup = a.src[0].up != 0;
    unsigned upm[2] = {0u, 0u};                                    // per pixel-register set: which of the 8 channels are upsampled sources
    int xvo[NPASS];                                                // byte offset of this thread's pixels in a channel plane of the current source
#pragma unroll
    for (int p = 0; p < NPASS; ++p) xvo[p] = (int)(hrow[p] * xsH4 + wcol4[p]);
    auto next_source = [&]() {
        ++xsi;
        if (xsi == 1) { xp = a.src[1].p + (long long)n * a.src[1].sN; xsC = a.src[1].sC; xsH4 = (unsigned)a.src[1].sH * 4u; xend = a.c2; xup = a.src[1].up != 0; }
        else { xp = a.src[2].p + (long long)n * a.src[2].sN; xsC = a.src[2].sC; xsH4 = (unsigned)a.src[2].sH * 4u; xend = 1 << 30; xup = a.src[2].up != 0; }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) xvo[p] = (int)(hrow[p] * xsH4 + wcol4[p]);
    };
    // Pixel registers of two chunks: the loads of chunk k+2 are issued during the multiply phase of chunk k and consumed at the end of
    // the multiply phase of chunk k+1 -- one multiply phase (1.4 us of matrix-pipe time) is shorter than the loaded memory latency.
    // Every chunk issues the SAME number of loads (channels beyond Cin read through an empty descriptor), so the hand-placed
    // s_waitcnt counts are compile-time constants.
    float xr[2][NPASS][8];
    auto load_channel = [&](int k, int cl, auto par) {
        constexpr int PAR = decltype(par)::value;
        const int ci = k * 8 + cl;                                // wave-uniform
        const bool live = ci < a.Cin && dbg != 1;
        if (live && ci >= xend) next_source();                    // (a source may be a single channel: two steps at most)
        if (live && ci >= xend) next_source();
        const i32x4 xs = make_rsrc(xp, live ? 0x7FFFFFF0u : 0u);
        const bool up = live && xup;
        if (cl == 0) upm[PAR] = 0u;
        upm[PAR] |= (up ? 1u : 0u) << cl;
        // an upsampled source: ONE low-resolution pixel per thread (register set 0 of the channel); the other sets load nothing
        xr[PAR][0][cl] = x3_load(xs, up ? (int)((unsigned)lrow * xsH4) + lcol4 : xvo[0]);
        const i32x4 xs1 = make_rsrc(xp, (live && !up) ? 0x7FFFFFF0u : 0u);
#pragma unroll
        for (int p = 1; p < NPASS; ++p) xr[PAR][p][cl] = x3_load(xs1, xvo[p]);
        if (live) xp += xsC;
    };
    // the pixel registers of set PAR have landed when at most NEWER younger vector-memory operations are outstanding
    auto wait_pixels = [&](auto par, auto newer) {
        constexpr int PAR = decltype(par)::value, NEWER = decltype(newer)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) x3_wait8<NEWER>(xr[PAR][p]);
    };
    // low-resolution pixels of the upsampled channels -> LDS (before the barrier in front of the split pass)
    auto stage_lowres = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        if (upm[PAR] == 0u) return;
        float* lq = reinterpret_cast<float*>(smem_x3 + Cfg::L_OFF) + tid;
#pragma unroll
        for (int cl = 0; cl < 8; ++cl)
            if (((upm[PAR] >> cl) & 1u) && tid < Cfg::LSLOT) lq[cl * Cfg::LSLOT] = xr[PAR][0][cl];
    };
    auto convert = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int s = p * 256 + tid;
            if ((p + 1) * 256 <= NSLOT || s < NSLOT) {
                vr_i32x4 ph, pm, pl;
                if (upm[PAR] != 0u) {
                    // torch's bilinear, align_corners=True (pointwise.hip: upsample2x_kernel): the +1 neighbours are read even at the
                    // last row / column, where their weight is exactly 0 and the staging tile holds zeros
                    const char* lq = smem_x3 + Cfg::L_OFF + (lidx[p] >= 0 ? lidx[p] : 0);
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
    // B operands [b1|b2] and [b1|b3]: lanes 0-31 read plane 0, lanes 32-63 plane 1 resp. 2; pixel (row wave*WN + ni + ty, col l31 + tx)
    const int bb0 = (khalf * NSLOT + wave * WN * PW + l31) * 16;
    const int bb1 = (2 * khalf * NSLOT + wave * WN * PW + l31) * 16;
    // A operands [a1|a1], [a2|a2], [a3|a1]
    const int ab0 = l31 * 16, ab1 = (MT + l31) * 16, ab2 = ((khalf ? 0 : 2) * MT + l31) * 16;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // epilogue constants of this cout tile (bias; eval: folded BatchNorm scale / shift): loaded FIRST, parked in LDS behind the first
    // pixel wait -- per-row global loads inside the epilogue were one serialised memory round trip per accumulator row
    float ecv[3];
    {
        const int ec = co0 + (tid & (MT - 1));
        const int ecc = ec < a.Cout ? ec : a.Cout - 1;
        const i32x4 rb = make_rsrc(a.bias, a.bias ? 0x7FFFFFF0u : 0u);
        const i32x4 re = make_rsrc(a.epi, a.epi ? 0x7FFFFFF0u : 0u);
        ecv[0] = x3_load(rb, ecc * 4);
        ecv[1] = x3_load(re, ecc * 8);
        ecv[2] = x3_load(re, ecc * 8 + 4);
    }
    // prologue: pixels of chunk 0 -> P, weights of chunk 0 and pixels of chunk 1 in flight
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(0, cl, P0{});
    issue_w(0);
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(1, cl, P1{});
    wait_pixels(P0{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});      // chunk 0's pixels (weights and chunk 1 stay in flight)
    asm volatile("" : "+v"(ecv[0]), "+v"(ecv[1]), "+v"(ecv[2]));                   // (older loads: landed with them)
    if (tid < MT) {
        float* E = reinterpret_cast<float*>(smem_x3 + Cfg::E_OFF);
        E[tid] = ecv[0];
        E[MT + tid] = a.epi ? ecv[1] : 1.f;
        E[2 * MT + tid] = a.epi ? ecv[2] : 0.f;
    }
    if (any_up) {
        stage_lowres(P0{});
        lds_barrier();
    }
    convert(P0{});
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk 0 landed; chunk 1's pixels stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one chunk: multiply P(k) x W(k) while the weights of chunk k+1 and the pixels of chunk k+2 arrive; then split chunk k+1 into P
    auto chunk = [&](int k, auto par) {
        constexpr int PAR = decltype(par)::value;                 // k & 1: the pixel registers chunk k came from (free again)
        const bool more = k + 1 < nchunk;
        {
            const char* Wb = smem_x3 + Cfg::P_BYTES + PAR * Cfg::W_BYTES;
            vr_bf16x8 A[2][3][WM], B[2][2][WN];
            // operand reads of tap t, in the order the MFMA groups consume them: part 0 = [a3|a1] + [b1|b3], part 1 = [a2|a2] + [b1|b2],
            // part 2 = [a1|a1]
            auto read_part = [&](int t, int buf, int part) {
                const int ty = t / 3, tx = t % 3;
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) {
                    const char* q = Wb + (t * 3 * MT + mi * 32) * 16;
                    if (part == 0) A[buf][2][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab2);
                    if (part == 1) A[buf][1][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab1);
                    if (part == 2) A[buf][0][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab0);
                }
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int o = ((ni + ty) * PW + tx) * 16;
                    if (part == 0) B[buf][1][ni] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb1 + o);
                    if (part == 1) B[buf][0][ni] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb0 + o);
                }
            };
            auto mfma_group = [&](int buf, int ja, int jb) {
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = mfma_bf16x16(A[buf][ja][mi], B[buf][jb][ni], acc[mi][ni]);
            };
            read_part(0, 0, 0); read_part(0, 0, 1); read_part(0, 0, 2);
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int cur = t & 1;
                // the operand reads of tap t+1 go out in three bursts between the three MFMA groups of tap t; the vector-memory work
                // for the coming chunks rides behind the first groups of taps 0..4 (weights first: they are needed one chunk earlier)
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 2, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 1);
                if (more) {
                    if (t == 0) issue_w(k + 1);
                    if (t >= 1 && t <= 4) { load_channel(k + 2, 2 * t - 2, par); load_channel(k + 2, 2 * t - 1, par); }
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            using Q = std::integral_constant<int, PAR ^ 1>;
            // outstanding, oldest first: chunk k+1's pixels | weights of chunk k+1 | chunk k+2's pixels
            wait_pixels(Q{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});
            if (any_up) stage_lowres(Q{});
            lds_barrier();                                       // every wave has read P(k); the low-resolution tile of chunk k+1 is in LDS
            if (dbg != 3) convert(Q{});
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk k+1 landed
            __builtin_amdgcn_s_barrier();                        // P(k+1) complete
            asm volatile("" ::: "memory");
        }
    };
    for (int k = 0; k < nchunk; k += 2) {
        chunk(k, P0{});
        if (k + 1 < nchunk) chunk(k + 1, P1{});
    }
    // The last prefetch (chunk nchunk-2, or the prologue when nchunk == 1) targets channels beyond Cin through an empty descriptor:
    // nothing waits for those zero-returning loads inside the loop, and hipcc does not track inline-asm loads -- drain them before
    // the epilogue may reuse the xr[] registers for addresses or old destination values.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---------------- epilogue (conv_epilogue.h): bias, (eval) BatchNorm + activation, up to three destination segments ---------
    if (dbg == 4) return;
    {
        int hon[WN], won[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) { hon[ni] = h0 + wave * WN + ni; won[ni] = w0 + l31; }
        epi_store<MT, WM, WN>(VR_EPI_ARGS(a), acc, reinterpret_cast<const float*>(smem_x3 + Cfg::E_OFF), n, co0, khalf,
                              h0 + TH <= a.Hout && w0 + TW <= a.Wout, hon, won);
    }
    // ---------------- BatchNorm partial statistics (training) -------------------------------------------------
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_x3);                     // [4 waves][MT][2]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int ho = h0 + wave * WN + ni, wo = w0 + l31;
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

// ---- weights as three bf16 planes: w [Cin][KK][CoutPad] fp32 -> x3w [ceil(Cin/8)][KK][3][CoutPad][8 channels] ------------
__device__ __forceinline__ void x3_weights_elem(const float* __restrict__ w, unsigned short* __restrict__ o, int Cin, int KK, int CoutPad,
                                                long long gid) {
    const int cin8 = (Cin + 7) / 8 * 8;
    if (gid >= (long long)cin8 * KK * CoutPad) return;
    const int co = (int)(gid % CoutPad);
    const int t = (int)((gid / CoutPad) % KK);
    const int ci = (int)(gid / ((long long)CoutPad * KK));
    const float v = ci < Cin ? w[((long long)ci * KK + t) * CoutPad + co] : 0.f;
    int p1, p2, p3;
    split3_pair(v, 0.f, p1, p2, p3);
    unsigned short* q = o + ((((long long)(ci >> 3) * KK + t) * 3) * CoutPad + co) * 8 + (ci & 7);
    q[0] = (unsigned short)(p1 & 0xffff);
    q[(long long)CoutPad * 8] = (unsigned short)(p2 & 0xffff);
    q[2LL * CoutPad * 8] = (unsigned short)(p3 & 0xffff);
}
__global__ void x3_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ o, int Cin, int KK, int CoutPad) {
    x3_weights_elem(w, o, Cin, KK, CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void x3_weights_batched_kernel(const X3WDesc* __restrict__ d) {
    const X3WDesc e = d[blockIdx.y];
    x3_weights_elem(e.w, static_cast<unsigned short*>(e.o), e.Cin, e.KK, e.CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

size_t x3_weights_bytes(int Cin, int KK, int CoutPad) { return (size_t)((Cin + 7) / 8) * KK * 3 * CoutPad * 16; }

void launch_x3_weights(const float* w, void* o, int Cin, int KK, int CoutPad, hipStream_t st) {
    const long long n = (long long)((Cin + 7) / 8 * 8) * KK * CoutPad;
    VR_LAUNCH(x3_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, static_cast<unsigned short*>(o), Cin, KK,
                       CoutPad);
    VR_HIP(hipGetLastError());
}
void launch_x3_weights_batched(const X3WDesc* d_descs, int n, long long max_elems, hipStream_t st) {
    if (n <= 0) return;
    VR_LAUNCH(x3_weights_batched_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)n), dim3(256), 0, st, d_descs);
    VR_HIP(hipGetLastError());
}

template <int MT, int TH>
static void x3_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = X3Cfg<MT, TH>;
    auto kern = conv_x3_kernel<MT, TH>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

// True when the launch can take this kernel (3x3 stride-1, plain inputs, split weights available); fills the tile choice.
bool x3_pick(const ConvArgs& a, const ConvShape& s, X3Tile* t) {
    static const int enabled = getenv("VR_CONV_X3") ? atoi(getenv("VR_CONV_X3")) : 1;
    if (!enabled || !a.x3w || a.tapmask) return false;
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1)) return false;
    if (a.pad_h != 1 || a.pad_w != 1 || a.Wout < 32) return false;
    const ConvSrc* u = nullptr;
    for (int i = 0; i < a.nsrc; ++i) {
        const ConvSrc& c = a.src[i];
        if (c.aff0 || c.aff1 || c.post || c.zins || c.slope != 1.f || (c.up ? 2 * c.W : c.W) != a.Win) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
        if (c.up) {                                          // fused bilinear x2: one interpolation geometry per launch
            if (u && (u->H != c.H || u->W != c.W)) return false;
            u = &c;
        }
    }
    int MT = (a.CoutPad % 64 == 0) ? 64 : 32;
    int TH = 8;
    static const int force_mt = getenv("VR_X3_MT") ? atoi(getenv("VR_X3_MT")) : 0;
    static const int force_th = getenv("VR_X3_TH") ? atoi(getenv("VR_X3_TH")) : 0;
    const long long tiles8 = (long long)a.N * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    if (MT == 64 && tiles8 * (a.CoutPad / 64) < 512) MT = 32;            // fewer than two workgroups per CU: halve the cout tile
    if (MT == 32) {
        const long long tiles16 = (long long)a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
        if (tiles16 * (a.CoutPad / 32) >= 1024) TH = 16;
    }
    if (force_mt == 32 || force_mt == 64) MT = (a.CoutPad % force_mt == 0) ? force_mt : MT;
    if (force_th == 8 || force_th == 16) TH = force_th;
    if (MT == 64) TH = 8;
    t->MT = MT; t->TH = TH;
    return true;
}

void x3_fill_tiling(ConvArgs& a, const X3Tile& t) {
    a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_h = (a.Hout + t.TH - 1) / t.TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / t.MT;
}

void x3_launch_conv(const ConvArgs& a, const X3Tile& t, hipStream_t st) {
    if (a.bf16 == 3) { x3h_launch_conv(a, t, st); return; }          // mfma_mode 3: three fp16 products (conv_x3h.hip)
    if (t.MT == 64) x3_launch<64, 8>(a, st);
    else if (t.TH == 16) x3_launch<32, 16>(a, st);
    else x3_launch<32, 8>(a, st);
}

}  // namespace vr
