// Direct 3x3 stride-1 convolution with fp32-grade products from THREE fp16 products ("mfma_mode" 3; lib/layers.py:12-20).
//
// conv_x3.hip forms every fp32 product from six bf16 products (three 8-bit planes per operand).  fp16 carries 11 significand bits,
// so TWO planes hold 22 of fp32's 24 and  a b ~= a1 b1 + a1 b2 + a2 b1  (the omitted a2 b2 and the two dropped operand tails are each
// <= 2^-22 |a b|, a quarter of the fp32 rounding an fp32 MULTIPLY-ADD chain commits per step; measured error against fp64 = the fp32
// direct kernel's, tests/test_gpu_b16.py).  That is 14 matrix instructions per 8-channel chunk instead of 27:
//   X(t):      A = [a1(t) | a1(t)]    x  B = [b1(t) | b2(t)]        one per tap (lanes 0-31 carry k 0..7, lanes 32-63 k 8..15)
//   Y(t,t+1):  A = [a2(t) | a2(t+1)]  x  B = [b1(t) | b1(t+1)]      one per PAIR of taps: the two k halves are two different taps
//                                                                  (tap 8 alone: upper half of A zeroed)
// measured bound (profiles/r04_x3_half_mfma_ablation.txt): conv_x3 with 13 of its 27 instructions is 20-24 % faster.
//
// fp16 has 5 exponent bits, so operands are scaled by exact powers of two:
//   * weights: per output channel, max |w| of the row -> [2^14, 2^15) (x3h_wscale_kernel; the inverse is a fourth epilogue constant);
//   * pixels: per workgroup and 8-channel chunk.  Every thread takes max |x| over the pixels it loaded, the wave maxima meet in LDS
//     at the barrier that already separates the multiply phase from the split pass, and the chunk is scaled so that its maximum lies
//     in [2^13, 2^14).  The accumulators carry ONE running shift `sh`; it follows the chunks with hysteresis (kept while the chunk
//     maximum stays within [2^1, 2^15) of the scaled range, i.e. 2^-26 of the chunk maximum is still resolved by the second plane's
//     subnormals) and when it moves the accumulators are multiplied by the exact power of two -- rare on real activations.
//   Absolute operand error <= max(2^-22 |x'|, 2^-25) in scaled units: relative to the largest pixel of the tile's chunk at least
//   2^-26, so dynamic range ACROSS tiles, chunks and output channels is unlimited (2^+-100 inputs, fp32 subnormals: exact scaling).
// The split is two instructions per value: v_fma_mixlo/hi_f16 computes fma(x, s, 0) resp. fma(x, s, -a1) in fp32 and rounds once to
// fp16 into one half of the destination (4 VALU per channel pair, conv_x3: 11).
//
// Everything else is conv_x3.hip's structure: 4 waves, TH x 32 pixels x MT couts, register pixel loads two chunks ahead with
// hand-placed waits, weights by LDS-DMA (double-buffered), fused bilinear x2, contiguous tile range per XCD, conv_epilogue.h.
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"
#include "x3h_common.h"

namespace vr {

// UP: the launch has upsampled sources (only then the low-resolution staging tile takes LDS: without it three 64 x 8 workgroups fit a CU
// with room to spare -- 146 KB -- where 3 x 53.3 KB = 159.8 KB sat on the edge of the 160 KB, allocation granularity unknown)
template <int MT, int TH, bool UP = true>
struct X3hCfg {
    static constexpr int TW = 32, CK = 8, KK = 9;
    static constexpr int TH_in = TH + 2, PW = TW + 2;           // halo tile, pixels
    static constexpr int NSLOT = TH_in * PW;
    static constexpr int NPASS = (NSLOT + 255) / 256;
    static constexpr int WM = MT / 32, WN = TH / 4;
    static constexpr int PLANE = NSLOT * 16;                     // bytes of one fp16 plane (8 channels per pixel)
    static constexpr int P_BYTES = 2 * PLANE;
    static constexpr int NWP = KK * 2 * MT;                      // 16-byte weight operands per chunk
    static constexpr int W_BYTES = NWP * 16;
    static constexpr int NWPASS = (NWP + 255) / 256;
    // fused bilinear x2 (lib/layers.py:52): the low-resolution pixels under the halo tile, [LROWS][LW][8 ch] fp32 -- channel innermost, so
    // that a staging thread stores its 8 channels as two 16-byte writes and an interpolating thread fetches the 8 channels of a
    // neighbour as two ds_read_b128 (round 5: the [8 ch][LROWS][LW] form cost 32 four-byte reads per pixel, issued channel by channel
    // behind scalar branches with the LDS latency exposed each time -- 3.9 k of the 13.5 k cycles of a chunk, profiles/r05_x3h_phase_trace.txt)
    static constexpr int LROWS = TH / 2 + 3, LW = 20, LSLOT = LROWS * LW;
    static constexpr int L_OFF = P_BYTES + 2 * W_BYTES;
    static constexpr int L_BYTES = UP ? CK * LSLOT * 4 : 0;
    static constexpr int E_OFF = L_OFF + L_BYTES;               // epilogue constants of the cout tile: bias, scale, shift, 1 / weight scale [4][MT] fp32
    static constexpr int M_OFF = E_OFF + 4 * MT * 4;            // the four wave maxima of the chunk being split (uint bits of |x|)
    static constexpr int LDS_BYTES = M_OFF + 16;
    // workgroups per CU the register budget must allow (64 x 8 would fit three by LDS, 53 KB, but needs 175 registers: 36 bytes of
    // scratch under a 168-register cap, and a spilled in-flight pixel register would be silently wrong)
    static constexpr int OCC = (MT == 32 && TH == 8) ? 3 : 2;
    // vector-memory operations a wave issues per chunk: 8 * NPASS pixel loads (always, also beyond Cin: empty descriptor), and at
    // least NWMIN weight DMAs (the last wave-instruction of the weight slab may be empty for some waves)
    static constexpr int NXL = 8 * NPASS, NWMIN = (NWP / 64) / 4;
    static constexpr int NG = 14;                                // matrix-instruction groups per chunk: X0 X1 Y01 X2 X3 Y23 ... X8 Y8
    static_assert(TH % 4 == 0 && MT % 32 == 0 && LDS_BYTES <= 80 * 1024 && 2 * NXL + NWMIN < 64 && NXL <= 30 && LSLOT <= 256, "tile");
};

// ---- phase trace (diagnostics, VR_CONV_DBG bit 64): four workgroups in the middle of the grid stamp s_memtime at the phase boundaries
// of chunks 2..7, wave by wave -- where a chunk's ~14 k cycles go (multiply phase / pixel wait / barriers / split pass / weight wait).
// Layout [workgroup 4][wave 4][chunk 6][point 8]; read back with vr_debug_trace (api.cpp), tools/x3h_trace.py prints it.
__device__ long long g_x3h_trace[4 * 4 * 6 * 8];
void x3h_trace_read(long long* host, int n) {
    VR_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x3h_trace), sizeof(long long) * (size_t)(n < 768 ? n : 768)));
}
void x3h_trace_clear() {
    static long long zeros[768] = {0};
    VR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_x3h_trace), zeros, sizeof zeros));
}

// UP: some source arrives through the fused bilinear x2 (eval); the plain instantiation sheds that path's registers and code
// TRACE: the diagnostic build of the same kernel (phase stamps; never launched unless VR_CONV_DBG has bit 64)
// HI: one more workgroup per CU than the register budget of the UP form allows (plain form only: 153 / 113 registers)
template <int MT, int TH, bool UP, bool HI, bool TRACE = false>
__global__ __launch_bounds__(256, (X3hCfg<MT, TH>::OCC + (HI ? 1 : 0))) void conv_x3h_kernel(const ConvArgs a) {
    using Cfg = X3hCfg<MT, TH, UP>;
    constexpr int TW = Cfg::TW, KK = Cfg::KK, PW = Cfg::PW, NSLOT = Cfg::NSLOT, NPASS = Cfg::NPASS, WM = Cfg::WM, WN = Cfg::WN,
                  PLANE = Cfg::PLANE, NWP = Cfg::NWP, NWPASS = Cfg::NWPASS;
    extern __shared__ __attribute__((aligned(16))) char smem_x3h[];
    char* const Pb = smem_x3h;

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
    const unsigned lds0 = (unsigned)(size_t)smem_x3h;
    const int dbg = a.dbg & 15;
    const bool prio = (a.dbg & 32) != 0;                          // experiment: s_setprio 1 over the matrix instructions of a chunk
    const int trace_wg = TRACE ? (int)blockIdx.x - (int)(gridDim.x / 2) : -1;
    auto stamp = [&](int k, int point) {
        if (TRACE && trace_wg >= 0 && trace_wg < 4 && k >= 2 && k < 8) {
            const long long t = __builtin_readcyclecounter();
            if (lane == 0) g_x3h_trace[((trace_wg * 4 + wave) * 6 + (k - 2)) * 8 + point] = t;
        }
    };

    // ---- this thread's pixels of the halo tile: byte offset in a channel plane = row * (4 * sH) + 4 * column (2^31: padding);
    // recomputed from the thread index whenever the source (= the row pitch) changes, instead of held in registers ----
    auto pixel_offset = [&](int p, unsigned sH4) -> int {
        const int s = p * 256 + tid;
        const int r = s / PW, c = s - r * PW;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
        return ok ? (int)((unsigned)hi * sH4 + (unsigned)(wi * 4)) : (int)0x80000000u;
    };
    // ---- sources that arrive through the decoder's bilinear x2 (align_corners=True; eval: the upsample is not materialised):
    // this thread's low-resolution pixel of the staging tile, and for each of its halo pixels the position inside that tile
    // and the two interpolation weights (upsampled sources share their geometry: model.hip) ----
    const ConvSrc& us = a.src[0].up ? a.src[0] : (a.src[1].up ? a.src[1] : a.src[2]);
    constexpr bool any_up = UP;
    int lrow = 0, lcol4 = 0;                       // low-res pixel this thread fetches (byte column; 2^31: none)
    int lidx[NPASS], lstep[NPASS];                 // (lstep: byte step to the +1 row << 16 | to the +1 column; 0 where that neighbour's weight is exactly 0)
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
            lidx[p] = ok ? ((h1 - lr0) * Cfg::LW + (w1 - lc0)) * 32 : -1;        // byte offset of the pixel's 8 channels in the staging tile
            // A pixel on the last low-resolution row / column has weight exactly 0 on its +1 neighbour, which would lie OUTSIDE the staging
            // tile (up to (LW + 1) * 32 + 16 bytes past it -- epilogue constants, wave maxima or beyond LDS_BYTES; 0 * inf or
            // 0 * nan would poison the sum).  Step 0 there: the neighbour read lands on the pixel itself.
            const int rs = (h1 + 1 < us.H && h1 + 1 - lr0 < Cfg::LROWS) ? Cfg::LW * 32 : 0;
            const int cs = (w1 + 1 < us.W && w1 + 1 - lc0 < Cfg::LW) ? 32 : 0;
            lstep[p] = (rs << 16) | cs;
            lh[p] = h1r - (float)h1;
            lw_[p] = w1r - (float)w1;
        }
    }
    // ---- weight operands: LDS order [tap][plane][m], source x3w[chunk][(tap * 2 + plane) * CoutPad + co0 + m] ----
    // (lane q = (wave + 4 i) * 64 + lane covers operand (tp, m) = (q / MT, q % MT): the part that depends on the pass i is the same
    // for every lane, so it rides on the scalar offset of the DMA -- one vector offset instead of NWPASS)
    unsigned woff0;
    {
        const int q = wave * 64 + lane;
        const int m = q % MT, tp = q / MT;
        woff0 = (unsigned)((tp * a.CoutPad + m) * 16);
    }
    const unsigned wstep = (unsigned)((256 / MT) * a.CoutPad * 16);   // four waves further on
    const long long wchunk_bytes = (long long)KK * 2 * a.CoutPad * 16;
    auto issue_w = [&](int k) {                                    // the weight DMA of chunk k: NWPASS wave-instructions
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
    for (int p = 0; p < NPASS; ++p) xvo[p] = pixel_offset(p, xsH4);
    auto next_source = [&]() {
        ++xsi;
        if (xsi == 1) { xp = a.src[1].p + (long long)n * a.src[1].sN; xsC = a.src[1].sC; xsH4 = (unsigned)a.src[1].sH * 4u; xend = a.c2; xup = a.src[1].up != 0; }
        else { xp = a.src[2].p + (long long)n * a.src[2].sN; xsC = a.src[2].sC; xsH4 = (unsigned)a.src[2].sH * 4u; xend = 1 << 30; xup = a.src[2].up != 0; }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) xvo[p] = pixel_offset(p, xsH4);
    };
    // Pixel registers of two chunks: the loads of chunk k+2 are issued during the multiply phase of chunk k and consumed at the end of
    // the multiply phase of chunk k+1 -- one multiply phase (1.4 us of matrix-pipe time) is shorter than the loaded memory latency.
    // Every chunk issues the SAME number of loads (channels beyond Cin read through an empty descriptor), so the hand-placed
    // s_waitcnt counts are compile-time constants.
    float xr[2][NPASS][8];
    // (Round 5, measured and reverted -- as in round 4: a short path for chunks whose eight channels are live, in one source and not
    // upsampled -- one descriptor per chunk, channel cl at a scalar offset, ~15 instead of ~50 instructions per channel -- made the kernel
    // 6 % SLOWER (conv_x3h 4.90 -> 5.22 ms per inference step): the scalar bookkeeping is not what the waves wait for.)
    auto load_channel = [&](int k, int cl, auto par) {
        constexpr int PAR = decltype(par)::value;
        const int ci = k * 8 + cl;                                // wave-uniform
        const bool live = ci < a.Cin && dbg != 1;
        if (live && ci >= xend) next_source();                    // (a source may be a single channel: two steps at most)
        if (live && ci >= xend) next_source();
        const i32x4 xs = make_rsrc(xp, live ? 0x7FFFFFF0u : 0u);
        const bool up = UP && live && xup;
        if (cl == 0) upm[PAR] = 0u;
        upm[PAR] |= (up ? 1u : 0u) << cl;
        // an upsampled source: ONE low-resolution pixel per thread (register set 0 of the channel); the other sets load nothing
        xr[PAR][0][cl] = x3h_load(xs, up ? (int)((unsigned)lrow * xsH4) + lcol4 : xvo[0]);
        const i32x4 xs1 = make_rsrc(xp, (live && !up) ? 0x7FFFFFF0u : 0u);
#pragma unroll
        for (int p = 1; p < NPASS; ++p) xr[PAR][p][cl] = x3h_load(xs1, xvo[p]);
        if (live) xp += xsC;
    };
    // the pixel registers of set PAR have landed when at most NEWER younger vector-memory operations are outstanding
    auto wait_pixels = [&](auto par, auto newer) {
        constexpr int PAR = decltype(par)::value, NEWER = decltype(newer)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) x3h_wait8<NEWER>(xr[PAR][p]);
    };
    // low-resolution pixels of the upsampled channels -> LDS (before the barrier in front of the split pass)
    auto stage_lowres = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        if (upm[PAR] == 0u) return;
        // (channels of the chunk that are NOT upsampled hold this thread's full-resolution pixel here: stored too, never read back)
        if (tid < Cfg::LSLOT) {
            vr_f32x4h lo4, hi4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo4[j] = xr[PAR][0][j]; hi4[j] = xr[PAR][0][4 + j]; }
            char* lq = smem_x3h + Cfg::L_OFF + tid * 32;
            *reinterpret_cast<vr_f32x4h*>(lq) = lo4;
            *reinterpret_cast<vr_f32x4h*>(lq + 16) = hi4;
        }
    };
    // ---- the running power-of-two shift of the pixels (header): x' = x * 2^sh ----
    int sh = 0, shlo = 0;                                          // (shlo: the shift the largest chunk so far asked for)
    float psc = 1.f;                                               // 2^sh
    // max |x| over this thread's pixel registers of set PAR -> wave maximum -> LDS (read back behind the next barrier)
    auto post_max = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int cl = 0; cl < 8; cl += 2)      // (inline asm: hipcc canonicalises fabsf() with a v_max of its own per value)
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(xr[PAR][p][cl]), "v"(xr[PAR][p][cl + 1]));
        int b = __float_as_int(m);                                 // non-negative floats order like their bit patterns
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true));   // row_half_mirror
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true));   // row_mirror: every lane of a row of 16 holds the row's maximum
        const int w = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
                          max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
        if (lane == 0) reinterpret_cast<int*>(smem_x3h + Cfg::M_OFF)[wave] = w;
    };
    // behind the barrier: the chunk's maximum -> the shift; the accumulators follow when it moves (f32x16 acc[][] is declared below)
    auto read_max_exp = [&]() -> int {
        const vr_i32x4 mm = *reinterpret_cast<const vr_i32x4*>(smem_x3h + Cfg::M_OFF);
        const int w = max(max(mm[0], mm[1]), max(mm[2], mm[3]));
        return __builtin_amdgcn_readfirstlane(w) >> 23;            // biased exponent of the largest |x| (255: inf / nan)
    };
    auto convert = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int s = p * 256 + tid;
            if ((p + 1) * 256 <= NSLOT || s < NSLOT) {
                vr_i32x4 ph, pl;
                if (upm[PAR] != 0u) {
                    // torch's bilinear, align_corners=True (pointwise.hip: upsample2x_kernel); the steps to the +1 neighbours are clamped
                    // (lstep) so that nothing outside the staging tile is ever read
                    const char* lq = smem_x3h + Cfg::L_OFF + (lidx[p] >= 0 ? lidx[p] : 0);
                    const int rs = lstep[p] >> 16, cs = lstep[p] & 0xffff;
                    const float h1l = lh[p], h0l = 1.f - h1l, w1l = lw_[p], w0l = 1.f - w1l;
                    if (upm[PAR] == 0xFFu) {
                        // the whole chunk is upsampled (64 of the 65 upsampled channels of a dec1 layer): eight 16-byte reads up front,
                        // then straight-line arithmetic -- no per-channel branch, the LDS latency is paid once per pixel
                        vr_f32x4h nb[4][2];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                            for (int hf = 0; hf < 2; ++hf)
                                nb[q4][hf] = *reinterpret_cast<const vr_f32x4h*>(lq + (q4 >> 1) * rs + (q4 & 1) * cs + hf * 16);
#pragma unroll
                        for (int cl = 0; cl < 8; ++cl) {
                            const float v00 = nb[0][cl >> 2][cl & 3], v01 = nb[1][cl >> 2][cl & 3], v10 = nb[2][cl >> 2][cl & 3], v11 = nb[3][cl >> 2][cl & 3];
                            const float v = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
                            xr[PAR][p][cl] = lidx[p] >= 0 ? v : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int cl = 0; cl < 8; ++cl) {
                            if ((upm[PAR] >> cl) & 1u) {
                                const char* q = lq + cl * 4;
                                const float v00 = *reinterpret_cast<const float*>(q), v01 = *reinterpret_cast<const float*>(q + cs),
                                            v10 = *reinterpret_cast<const float*>(q + rs), v11 = *reinterpret_cast<const float*>(q + rs + cs);
                                const float v = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
                                xr[PAR][p][cl] = lidx[p] >= 0 ? v : 0.f;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int h, l;
                    split2h_pair(xr[PAR][p][2 * j], xr[PAR][p][2 * j + 1], psc, h, l);
                    ph[j] = h; pl[j] = l;
                }
                char* q = Pb + s * 16;
                *reinterpret_cast<vr_i32x4*>(q) = ph;
                *reinterpret_cast<vr_i32x4*>(q + PLANE) = pl;
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // B operands: pixel (row wave*WN + ni + ty, col l31 + tx) of plane 0 is at bq + ((ni + ty) * PW + tx) * 16.
    //   X(t): lanes 0-31 plane 0 (b1), lanes 32-63 plane 1 (b2) of tap t's pixel;   Y(t,t+1): plane 0, lanes 32-63 at tap t+1's pixel,
    //   which lies one pixel to the right (t = 0, 4, 6) or PW - 2 pixels on (t = 2: from (0,2) to (1,0))
    const int bq = (wave * WN * PW + l31) * 16;
    const int bX = bq + khalf * PLANE, bY1 = bq + khalf * 16, bY2 = bq + khalf * (PW - 2) * 16;
    // A operands, LDS order [tap][plane][m]:  X(t): a1(t) in both halves;  Y(t,t+1): a2(t) | a2(t+1);  Y(8): a2(8) | 0
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
    // epilogue constants of this cout tile (bias; eval: folded BatchNorm scale / shift; 1 / weight scale): loaded FIRST, parked in LDS
    // behind the first pixel wait -- per-row global loads inside the epilogue were one serialised memory round trip per accumulator row
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
        ecv[3] = x3h_load(rw, ec * 4);                                             // (padded couts included: [CoutPad])
    }
    // the shift follows the chunk maxima (header): `e` = biased exponent of the largest |x| of the chunk about to be split
    auto follow = [&](int e, bool first) {
        const int need = 140 - (e < 14 ? 14 : e);                                  // chunk maximum -> [2^13, 2^14)
        // The shift never runs more than 2^64 ahead of the LARGEST chunk seen so far (smallest `need`): that chunk's sums (<= 2^42 in
        // its own scale) must survive every later multiplication of the accumulators -- chunks 2^64 below it do not matter anyway.
        shlo = (first || need < shlo) ? need : shlo;
        int nsh = sh;
        if (first || need < sh - 1) nsh = need;                                    // (larger than 2^15 after scaling: must move)
        else if (need > sh + 12) {                                                 // (maximum below 2: the second plane starts losing bits)
            nsh = need < sh + 64 ? need : sh + 64;
            nsh = nsh < shlo + 64 ? nsh : shlo + 64;
            nsh = nsh > sh ? nsh : sh;
        }
        if (nsh != sh) {
            if (!first) {
                const int d = nsh - sh;                                            // <= 64; a large negative d flushes the old sums
                const float f = d < -126 ? 0.f : x3h_pow2(d);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= f;
            }
            sh = nsh;
            psc = x3h_pow2(sh);                                                    // sh in [-115, 126]
        }
    };
    // prologue: pixels of chunk 0 -> P, weights of chunk 0 and pixels of chunk 1 in flight
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(0, cl, P0{});
    issue_w(0);
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) load_channel(1, cl, P1{});
    wait_pixels(P0{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});      // chunk 0's pixels (weights and chunk 1 stay in flight)
    asm volatile("; landed %0 %1 %2 %3" : "+v"(ecv[0]), "+v"(ecv[1]), "+v"(ecv[2]), "+v"(ecv[3]));      // (older loads: landed with them; the comment is for tools/asm_inflight_audit2.py)
    if (tid < MT) {
        float* E = reinterpret_cast<float*>(smem_x3h + Cfg::E_OFF);
        E[tid] = ecv[0];
        E[MT + tid] = a.epi ? ecv[1] : 1.f;
        E[2 * MT + tid] = a.epi ? ecv[2] : 0.f;
        E[3 * MT + tid] = ecv[3];
    }
    post_max(P0{});
    if (any_up) stage_lowres(P0{});
    lds_barrier();
    follow(read_max_exp(), true);
    convert(P0{});
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk 0 landed; chunk 1's pixels stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one chunk: multiply P(k) x W(k) while the weights of chunk k+1 and the pixels of chunk k+2 arrive; then split chunk k+1 into P
    auto chunk = [&](int k, auto par) {
        constexpr int PAR = decltype(par)::value;                 // k & 1: the pixel registers chunk k came from (free again)
        const bool more = k + 1 < nchunk;
        stamp(k, 0);
        {
            const char* Wb = smem_x3h + Cfg::P_BYTES + PAR * Cfg::W_BYTES;
            vr_f16x8 A[2][WM], B[2][WN];
            // group g of the 14: g = 3q + {0, 1} -> X(2q), X(2q + 1); g = 3q + 2 -> Y(2q, 2q + 1); g = 12 -> X(8); g = 13 -> Y(8)
            auto read_group = [&](int g, int buf) {
                const bool isY = g == 13 || (g < 12 && g % 3 == 2);
                const int t = g >= 12 ? 8 : 2 * (g / 3) + (g % 3 == 1 ? 1 : 0);
                const int ty = t / 3, tx = t % 3;
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) {
                    const char* q = Wb + (t * 2 * MT + mi * 32) * 16;
                    if (!isY) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aX);
                    else if (t < 8) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aY);
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
                    const int o = ((ni + ty) * PW + tx) * 16;
                    if (!isY) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bX + o);
                    else if (t == 8) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bq + o);     // (upper half meets zeros)
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
            if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < Cfg::NG; ++g) {
                const int cur = g & 1;
                // the operand reads of group g+1 go out in front of the matrix instructions of group g; the vector-memory work for the
                // coming chunks rides behind the first groups (weights first: they are needed one chunk earlier)
                if (g + 1 < Cfg::NG) read_group(g + 1, cur ^ 1);
                if (more) {
                    if (g == 0) issue_w(k + 1);
                    if (g >= 1 && g <= 4) { load_channel(k + 2, 2 * g - 2, par); load_channel(k + 2, 2 * g - 1, par); }
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        stamp(k, 1);
        if (more) {
            using Q = std::integral_constant<int, PAR ^ 1>;
            // outstanding, oldest first: chunk k+1's pixels | weights of chunk k+1 | chunk k+2's pixels
            wait_pixels(Q{}, std::integral_constant<int, Cfg::NXL + Cfg::NWMIN>{});
            stamp(k, 2);
            post_max(Q{});
            if (any_up) stage_lowres(Q{});
            stamp(k, 3);
            lds_barrier();                                       // every wave has read P(k); maxima and the low-resolution tile of chunk k+1 are in LDS
            stamp(k, 4);
            follow(read_max_exp(), false);
            if (dbg != 3) convert(Q{});
            stamp(k, 5);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::NXL) : "memory");   // weights of chunk k+1 landed
            stamp(k, 6);
            __builtin_amdgcn_s_barrier();                        // P(k+1) complete
            asm volatile("" ::: "memory");
            stamp(k, 7);
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
        // undo the two scalings: 2^-sh (pixels, this workgroup) and 1 / weight scale (per cout: E[3][m], row m = mi*32 + (r&3) + 8*(r>>2) + 4*khalf)
        const float fo = x3h_pow2(-sh);                           // sh in [-115, 126]
        const float* Wi = reinterpret_cast<const float*>(smem_x3h + Cfg::E_OFF) + 3 * MT;
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
    {
        int hon[WN], won[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) { hon[ni] = h0 + wave * WN + ni; won[ni] = w0 + l31; }
        epi_store<MT, WM, WN>(VR_EPI_ARGS(a), acc, reinterpret_cast<const float*>(smem_x3h + Cfg::E_OFF), n, co0, khalf,
                              h0 + TH <= a.Hout && w0 + TW <= a.Wout, hon, won);
    }
    // ---------------- BatchNorm partial statistics (training) -------------------------------------------------
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_x3h);                     // [4 waves][MT][2]
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

// ---- weights as two fp16 planes, scaled per output channel: w [Cin][KK][CoutPad] fp32 ->
//      x3hw [ceil(Cin/8)][KK][2][CoutPad][8 channels] fp16 | winv [CoutPad] fp32 (1 / scale) | wscl [CoutPad] fp32 (scale) -------------
// (the buffer is sized by x3_weights_bytes(): the third plane's room holds the two tails)
__device__ __forceinline__ float* x3h_tail(void* o, int Cin, int KK, int CoutPad) {
    return reinterpret_cast<float*>(static_cast<char*>(o) + (size_t)((Cin + 7) / 8) * KK * 2 * CoutPad * 16);
}
// one block = 32 output channels x 8 row groups: max |w| of each row -> scale 2^(141 - e) (maximum lands in [2^14, 2^15))
__device__ __forceinline__ void x3h_wscale_block(const float* __restrict__ w, void* o, int Cin, int KK, int CoutPad, int cb) {
    __shared__ float red[8][32];
    if (cb * 32 >= CoutPad) return;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5, co = cb * 32 + c;
    const int rows = Cin * KK;
    float m = 0.f;
    for (int r = g; r < rows; r += 8) m = fmaxf(m, fabsf(w[(long long)r * CoutPad + co]));
    red[g][c] = m;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int j = 1; j < 8; ++j) m = fmaxf(m, red[j][c]);
        int e = __float_as_int(m) >> 23;                           // biased exponent; 0 for an all-zero (padding) row
        e = e < 15 ? 15 : (e > 254 ? 254 : e);
        float* t = x3h_tail(o, Cin, KK, CoutPad);
        t[co] = __int_as_float((e - 14) << 23);                    // 2^(e - 141): what the epilogue multiplies by
        t[CoutPad + co] = __int_as_float((268 - e) << 23);         // 2^(141 - e)
    }
}
__device__ __forceinline__ void x3h_weights_elem(const float* __restrict__ w, void* o, int Cin, int KK, int CoutPad, long long gid) {
    const int cin8 = (Cin + 7) / 8 * 8;
    if (gid >= (long long)cin8 * KK * CoutPad) return;
    const int co = (int)(gid % CoutPad);
    const int t = (int)((gid / CoutPad) % KK);
    const int ci = (int)(gid / ((long long)CoutPad * KK));
    const float sc = x3h_tail(o, Cin, KK, CoutPad)[CoutPad + co];
    const float v = ci < Cin ? w[((long long)ci * KK + t) * CoutPad + co] : 0.f;
    const _Float16 h1 = (_Float16)(v * sc);                        // (v * sc is exact: a power of two)
    const _Float16 h2 = (_Float16)fmaf(v, sc, -(float)h1);         // the residual is exact in fp32, rounded once
    _Float16* q = static_cast<_Float16*>(o) + ((((long long)(ci >> 3) * KK + t) * 2) * CoutPad + co) * 8 + (ci & 7);
    q[0] = h1;
    q[(long long)CoutPad * 8] = h2;
}
__global__ void x3h_wscale_kernel(const float* __restrict__ w, void* o, int Cin, int KK, int CoutPad) {
    x3h_wscale_block(w, o, Cin, KK, CoutPad, blockIdx.x);
}
__global__ void x3h_weights_kernel(const float* __restrict__ w, void* o, int Cin, int KK, int CoutPad) {
    x3h_weights_elem(w, o, Cin, KK, CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void x3h_wscale_batched_kernel(const X3WDesc* __restrict__ d) {
    const X3WDesc e = d[blockIdx.y];
    x3h_wscale_block(e.w, e.o, e.Cin, e.KK, e.CoutPad, blockIdx.x);
}
__global__ void x3h_weights_batched_kernel(const X3WDesc* __restrict__ d) {
    const X3WDesc e = d[blockIdx.y];
    x3h_weights_elem(e.w, e.o, e.Cin, e.KK, e.CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

void launch_x3h_weights(const float* w, void* o, int Cin, int KK, int CoutPad, hipStream_t st) {
    const long long n = (long long)((Cin + 7) / 8 * 8) * KK * CoutPad;
    VR_LAUNCH(x3h_wscale_kernel, dim3((unsigned)((CoutPad + 31) / 32)), dim3(256), 0, st, w, o, Cin, KK, CoutPad);
    VR_LAUNCH(x3h_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, o, Cin, KK, CoutPad);
    VR_HIP(hipGetLastError());
}
void launch_x3h_weights_batched(const X3WDesc* d_descs, int n, long long max_elems, int max_cout_pad, hipStream_t st) {
    if (n <= 0) return;
    VR_LAUNCH(x3h_wscale_batched_kernel, dim3((unsigned)((max_cout_pad + 31) / 32), (unsigned)n), dim3(256), 0, st, d_descs);
    VR_LAUNCH(x3h_weights_batched_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)n), dim3(256), 0, st, d_descs);
    VR_HIP(hipGetLastError());
}

template <int MT, int TH, bool UP, bool HI, bool TRACE = false>
static void x3h_launch_up(const ConvArgs& a, hipStream_t st) {
    using Cfg = X3hCfg<MT, TH, UP>;
    auto kern = conv_x3h_kernel<MT, TH, UP, HI, TRACE>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int MT, int TH>
static void x3h_launch(const ConvArgs& a, hipStream_t st) {
    static const int hi = [] { const char* e = getenv("VR_X3H_HI"); return e ? atoi(e) : 0; }();
    if (a.dbg & 64) {                                             // diagnostic build with phase stamps (tools/x3h_trace.py)
        if (a.src[0].up | a.src[1].up | a.src[2].up) x3h_launch_up<MT, TH, true, false, true>(a, st);
        else x3h_launch_up<MT, TH, false, false, true>(a, st);
        return;
    }
    if (a.src[0].up | a.src[1].up | a.src[2].up) x3h_launch_up<MT, TH, true, false>(a, st);
    else if (hi && TH == 8) x3h_launch_up<MT, TH, false, (TH == 8)>(a, st);
    else x3h_launch_up<MT, TH, false, false>(a, st);
}

// same tile choice as conv_x3.hip (x3_pick / x3_fill_tiling); taken when ConvArgs::bf16 == 3
void x3h_launch_conv(const ConvArgs& a, const X3Tile& t, hipStream_t st) {
    if (t.MT == 64) x3h_launch<64, 8>(a, st);
    else if (t.TH == 16) x3h_launch<32, 16>(a, st);
    else x3h_launch<32, 8>(a, st);
}

}  // namespace vr
