// Weight gradient of the 3x3 stride-1 convolutions with fp32-grade products from THREE fp16 products ("mfma_mode" 3; backward of
// lib/layers.py:12-20 under train.py:92).  Direct form, reduction over PIXELS on v_mfma_f32_32x32x16_f16:
//
//   dW[ci][ty][tx][co] = sum over (n, h, w) of  x[n][ci][h + ty - 1][w + tx - 1] * dz[n][co][h][w]
//
// Both operands are produced inside the kernel: every fp32 value is scaled by an exact power of two and written as two fp16 numbers
// (conv_x3h.hip: 22 significand bits; v_fma_mixlo/hi_f16, 4 VALU per pixel pair) with the PIXELS innermost -- a lane's 16-byte
// matrix operand is 8 consecutive pixels of one channel -- and per 16 pixels, tap and 32-cout block
//     X(h):  A = [x1(h) | x2(h)]    x  B = [z1(h) | z1(h)]      h = 0, 1: both k halves carry THE SAME 8 pixels
//     Y:     A = [x1(0) | x1(1)]    x  B = [z2(0) | z2(1)]      the two k halves are the two 8-pixel groups
// = x1 z1 + x2 z1 + x1 z2: three 16-deep instructions where the fp32 pipe needs 16 x 4 (v_mfma_f32_32x32x2_f32) and the Winograd
// kernel (wgrad_wino.hip) 16 x 4 / 2.25 plus two operand transforms.
//
// Scaling: the x tile and the dz tile of a workgroup each get ONE power-of-two shift, following the tile maxima with hysteresis
// (conv_x3h.hip); the accumulators carry the sum of the two shifts and are multiplied by the exact power of two when one moves.
// A value 2^-16 below the largest of its tile still has 22 good bits (the second plane's fp16 subnormals reach 2^-38 of the tile
// maximum): channels of one tensor that differ by more than that within a 2 x 32-pixel tile lose relative accuracy in the smaller
// one's gradient row -- BatchNorm keeps the channels of every tensor this kernel sees within a few powers of two.
//
// Workgroup = 6 waves, block = (32 input channels, MT couts, one of P contiguous ranges of 4 x 32-pixel dz tiles); wave = (kernel row
// ty, half rh of the tile's four rows): two workgroups = twelve waves per CU cover the memory latency of each other's tile loads (the
// first version, 3 waves on 2 x 32 tiles, waited for its loads most of the time: 128 direct-equivalent TFLOP/s).  Per tile:
//   * every thread loads ~9 pixel PAIRS of x (6 halo rows x 34 columns per channel) and ~6 of dz (per-thread static offsets + one
//     scalar tile offset; only edge tiles compute bounds), for the NEXT tile while the current one is multiplied;
//   * tile maxima (v_max3 |.|, DPP, one LDS slot per wave) meet at the barrier that frees the planes; split pass: one split2h_pair
//     (4 VALU) per pair, two ds_write_b32 into  xP[plane][ci][row][40 px]  /  zP[plane][co][128 px]  (channel pitches 496 / 272
//     bytes = odd multiples of 16: a 32-lane operand read touches every bank once);
//   * wave (ty, rh) multiplies kernel row ty over output rows 2 rh, 2 rh + 1: the x operands of column tap 0 are aligned ds_read_b128 (the LDS row starts at image column
//     w0 - 1), tap 1 is a 16-bit funnel shift of two neighbouring operands (4 v_alignbit_b32), tap 2 a dword renaming of the same
//     registers; accumulators [ci 32][co 32] per (tap, cout block) stay in registers over the whole pixel range;
//   * the two halves of a kernel row meet in LDS; the 9 taps go to the block's partial slab [ci][tap][co] (lanes along co);
//     wgrad_reduce_kernel sums the P slabs.
#include <cstdlib>

#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

typedef _Float16 wxh_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 wxh_mfma(vr_i32x4 a, vr_i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wxh_f16x8, a), __builtin_bit_cast(wxh_f16x8, b), c, 0, 0, 0);
}
// (x0, x1) * s -> packed fp16 pairs of the two planes (conv_x3h.hip): p1 = rne(x s), p2 = rne(x s - p1), one fp32 fma each
__device__ __forceinline__ void wxh_split_pair(float x0, float x1, float s, int& p1, int& p2) {
    int a, b;                                    // (mixlo leaves the other half of its destination alone; mixhi then fills it: "=v" first)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(a) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(b) : "v"(x0), "v"(s), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(b) : "v"(x1), "v"(s), "v"(a));
    p1 = a; p2 = b;
}
__device__ __forceinline__ float wxh_pow2(int e) { return __int_as_float((e + 127) << 23); }      // e in [-126, 127]

template <int MT>
struct WxhCfg {
    static constexpr int CB = 32, TH = 4, TW = 32;
    static constexpr int XR = TH + 2, XPX = 40;                        // x rows of a tile, fp16 elements per row (34 used)
    static constexpr int XCI = XR * XPX * 2 + 16;                      // 496-byte channel pitch
    static constexpr int XPL = CB * XCI;                               // bytes per x plane
    static constexpr int ZCO = TH * TW * 2 + 16;                       // 272-byte cout pitch
    static constexpr int ZPL = MT * ZCO;
    static constexpr int Z_OFF = 2 * XPL;
    static constexpr int M_OFF = 2 * XPL + 2 * ZPL;                    // tile maxima: [x | dz][8 slots] uint bits of |v|
    static constexpr int LDS_BYTES = M_OFF + 64;
    static constexpr int NT = 384;
    // x pairs: 16 per row (tile columns 0..31) as 8 passes of 4 channels x 6 rows x 16 pairs = 384 threads, plus ONE pass for the 17th
    // pair of every row (tile columns 32, 33; threads 0..191 = 32 channels x 6 rows); dz pairs: 6 passes of 6 couts x 4 rows x 16
    static constexpr int NXM = CB / 4, NXP = NXM + 1;
    static constexpr int NZP = (MT + 5) / 6;
    static_assert(MT == 32, "dz passes are laid out for 32-cout blocks");
    static constexpr int WN = MT / 32;
};

// Loads are inline asm with hand-placed waits (see conv_x3.hip).  The per-pass part of an address (channel group i) rides on the
// scalar offset, the second pixel of a pair on the immediate: ONE vector offset per thread serves every pass.  (gfx9 checks
// vector offset + immediate against num_records, not the scalar offset: 2^31 in the vector offset still returns 0.)
__device__ __forceinline__ void wxh_load_pair(i32x4 rsrc, unsigned v0, unsigned v1, unsigned soff, float& a, float& b) {
    asm volatile("buffer_load_dword %0, %2, %4, %5 offen\n\tbuffer_load_dword %1, %3, %4, %5 offen offset:4"
                 : "=&v"(a), "=&v"(b) : "v"(v0), "v"(v1), "s"(rsrc), "s"(soff) : "memory");
}
// (the destination IS the register pair the split pass reads later: a copy out of a temporary would run before the data lands)
__device__ __forceinline__ void wxh_load_x2(i32x4 rsrc, unsigned v, unsigned soff, vr_f32x2& t) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(t) : "v"(v), "s"(rsrc), "s"(soff) : "memory");
}
// after an s_waitcnt: ties the loaded registers to this point so that their uses cannot be scheduled in front of the wait
__device__ __forceinline__ void wxh_tie(float& a, float& b, float& c, float& d) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void wxh_tie2(vr_f32x2& a) { asm volatile("" : "+v"(a)); }
// wave maximum of non-negative float bit patterns (all lanes of the wave participate)
__device__ __forceinline__ int wxh_wave_max(int b) {
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true));   // row_half_mirror
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true));   // row_mirror
    return max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
               max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
}

template <int MT>
__global__ __launch_bounds__(384, 3) void wgrad_x3h_kernel(const WgradArgs a) {
    using Cfg = WxhCfg<MT>;
    constexpr int NXP = Cfg::NXP, NZP = Cfg::NZP, WN = Cfg::WN, XPL = Cfg::XPL, ZPL = Cfg::ZPL, XCI = Cfg::XCI, ZCO = Cfg::ZCO;
    extern __shared__ __attribute__((aligned(16))) char smem_wxh[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int inner = a.nchunks * a.nct;
    const int p = (rr / inner) * 8 + xcd;
    if (p >= a.P) return;
    const int ib = rr % inner;
    const int ct = ib % a.nct, cb = ib / a.nct;
    const int co0 = ct * MT, c0 = cb * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = wave % 3, rh = wave / 3;                          // kernel row; half of the tile's four output rows
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    // ---- the source of the block's 32 input channels: the virtual concat's boundaries are multiples of 32 (wgrad_x3h_pick), so a
    // block reads ONE tensor ----
    const int si = (c0 >= a.in.c1) + (c0 >= a.in.c2);
    const int cbase = si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2);
    const float* const sp = VR_SEL_F(a.in, si, p);
    const long long sN = VR_SEL_F(a.in, si, sN), sC = VR_SEL_F(a.in, si, sC), sH = VR_SEL_F(a.in, si, sH);
    // ---- this thread's pixel pairs.  x, main passes: channel 4 i + xsub, tile row xrow, pair xpr; extra pass: channel tid / 6, row
    // tid % 6, pair 16.  dz: cout 6 i + zsub, row zrow, pair zpr.  Byte offsets relative to the tile's first pixel of the block's
    // first channel; 2^31 = nothing to load ----
    const int xsub = tid / 96, xrem = tid - xsub * 96, xrow = xrem >> 4, xpr = xrem & 15;
    const int esub = tid / 6, erow = tid - esub * 6;
    const int zsub = tid >> 6, zrem = tid & 63, zrow = zrem >> 4, zpr = zrem & 15;
    const int nlive = a.in.Cin - c0, zlive = a.Cout - co0;              // live channels / couts of this block (>= 1)
    const unsigned xoffM = (unsigned)(((long long)xsub * sC + (long long)xrow * sH + 2 * xpr) * 4);
    const unsigned xoffE = (tid < 192 && esub < nlive) ? (unsigned)(((long long)esub * sC + (long long)erow * sH + 32) * 4) : 0x80000000u;
    const unsigned zoffM = (unsigned)(((long long)zsub * a.zC + (long long)zrow * a.zH + 2 * zpr) * 4);
    const unsigned xstep = (unsigned)(4 * sC * 4), zstep = (unsigned)(6 * a.zC * 4);      // scalar offset per pass

    // pixel-pair registers.  EVERY lane issues EVERY load (dead pairs and pixels outside the image: offset 2^31, beyond the
    // descriptor's range, returns 0): a load inside a divergent branch may get a temporary destination that hipcc copies into the
    // merged variable right behind the issue, i.e. before the data has landed.
    float xa[NXP], xb[NXP];
    vr_f32x2 zz[NZP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) { xa[i] = 0.f; xb[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < NZP; ++i) { zz[i][0] = 0.f; zz[i][1] = 0.f; }
    const int dbg = a.in.dbg;                                          // perf ablations (VR_WXH_DBG): 1 no loads, 4 no split, 8 no maxima
    auto issue_tile = [&](int pt) {
        if (dbg & 1) return;
        const int n = pt / tiles_per_img;
        const int trem = pt - n * tiles_per_img;
        const int h0 = (trem / a.tiles_w) * 4, w0 = (trem % a.tiles_w) * 32;
        // x: rows h0 - 1 .. h0 + 4, columns w0 - 1 .. w0 + 32.  The descriptor starts at tile pixel (h0 - 1, w0 - 1) of the block's
        // first channel in image n: in front of the tensor for the first tile row / column, where those pixels are masked.  A thread's
        // position in the tile is the same in every pass, so the image-border masks are formed once per tile.
        {
            const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)(c0 - cbase) * sC + (long long)(h0 - 1) * sH + (w0 - 1), 0x7FFFFFF0u);
            const int hi = h0 - 1 + xrow, wi = w0 - 1 + 2 * xpr;
            const bool okh = hi >= 0 && hi < a.in.Hin;
            const unsigned o0 = (okh && wi >= 0 && wi < a.in.Win) ? xoffM : 0x80000000u;         // (wi + 1 >= 0 always)
            const unsigned o1 = (okh && wi + 1 < a.in.Win) ? xoffM : 0x80000000u;
            // (ONE load site per register; channels beyond Cin in the layer's last block: 4 i + xsub >= nlive)
#pragma unroll
            for (int i = 0; i < Cfg::NXM; ++i) {
                const bool lv = 4 * i + xsub < nlive;
                wxh_load_pair(xr, lv ? o0 : 0x80000000u, lv ? o1 : 0x80000000u, (unsigned)i * xstep, xa[i], xb[i]);
            }
            const int he = h0 - 1 + erow;
            const bool oke = he >= 0 && he < a.in.Hin;
            wxh_load_pair(xr, (oke && w0 + 31 < a.in.Win) ? xoffE : 0x80000000u, (oke && w0 + 32 < a.in.Win) ? xoffE : 0x80000000u, 0u,
                          xa[Cfg::NXM], xb[Cfg::NXM]);
        }
        // dz: rows h0 .. h0 + 3, columns w0 .. w0 + 31 (even width: a pair is inside or outside as a whole)
        {
            const i32x4 zr = make_rsrc(a.dz + (long long)n * a.zN + (long long)co0 * a.zC + (long long)h0 * a.zH + w0, 0x7FFFFFF0u);
            const unsigned oz = (h0 + zrow < a.in.Hout && w0 + 2 * zpr < a.in.Wout) ? zoffM : 0x80000000u;
#pragma unroll
            for (int i = 0; i < NZP; ++i) {
                const bool lv = 6 * i + zsub < (zlive < MT ? zlive : MT);
                wxh_load_x2(zr, lv ? oz : 0x80000000u, (unsigned)i * zstep, zz[i]);
            }
        }
    };
    auto wait_tile = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i + 1 < NXP; i += 2) wxh_tie(xa[i], xb[i], xa[i + 1], xb[i + 1]);
        if (NXP & 1) wxh_tie(xa[NXP - 1], xb[NXP - 1], xa[0], xb[0]);
#pragma unroll
        for (int i = 0; i < NZP; ++i) wxh_tie2(zz[i]);
    };
    // maxima of the loaded tile -> one LDS slot per wave and operand (read back behind the next barrier)
    auto post_max = [&]() {
        if (dbg & 8) return;
        float mx = 0.f, mz = 0.f;
#pragma unroll
        for (int i = 0; i < NXP; ++i) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(xa[i]), "v"(xb[i]));
#pragma unroll
        for (int i = 0; i < NZP; ++i) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mz) : "v"(zz[i][0]), "v"(zz[i][1]));
        const int wx = wxh_wave_max(__float_as_int(mx)), wz = wxh_wave_max(__float_as_int(mz));
        if (lane == 0) {
            int* M = reinterpret_cast<int*>(smem_wxh + Cfg::M_OFF);
            M[wave] = wx;
            M[8 + wave] = wz;
        }
    };
    int shx = 0, shz = 0;                                              // x' = x * 2^shx, dz' = dz * 2^shz
    float scx = 1.f, scz = 1.f;
    // split the loaded pairs into the fp16 planes (the caller has waited for the loads and set the scales)
    auto split_tile = [&]() {
        if (dbg & 4) return;
        int* const xd = reinterpret_cast<int*>(smem_wxh + xsub * XCI) + xrow * 20 + xpr;
#pragma unroll
        for (int i = 0; i < Cfg::NXM; ++i) {
            int h, l;
            wxh_split_pair(xa[i], xb[i], scx, h, l);
            xd[i * XCI] = h; xd[i * XCI + XPL / 4] = l;                  // (4 channels further on: 4 XCI bytes = XCI ints)
        }
        if (tid < 192) {
            int h, l;
            wxh_split_pair(xa[Cfg::NXM], xb[Cfg::NXM], scx, h, l);
            int* d = reinterpret_cast<int*>(smem_wxh + esub * XCI) + erow * 20 + 16;
            d[0] = h; d[XPL / 4] = l;
        }
        int* const zd = reinterpret_cast<int*>(smem_wxh + Cfg::Z_OFF + zsub * ZCO) + zrem;
#pragma unroll
        for (int i = 0; i < NZP; ++i) {
            if (6 * i + zsub < MT) {
                int h, l;
                wxh_split_pair(zz[i][0], zz[i][1], scz, h, l);
                zd[i * 6 * ZCO / 4] = h; zd[i * 6 * ZCO / 4 + ZPL / 4] = l;
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // x operands (rows = input channel l31, tile row ty + r): X-type [x1|x2] -- lanes 0-31 plane 0, lanes 32-63 plane 1, the same 8
    // pixels; Y-type [x1(0)|x1(1)] -- plane 0, lanes 32-63 eight pixels further on
    const char* xqX = smem_wxh + khalf * XPL + l31 * XCI + (ty + 2 * rh) * 80;
    const char* xqY = smem_wxh + l31 * XCI + (ty + 2 * rh) * 80 + khalf * 16;
    // dz operands (columns = cout nj * 32 + l31): X-type [z1|z1] -- plane 0 in both halves; Y-type [z2(0)|z2(1)] -- plane 1
    const char* zqX = smem_wxh + Cfg::Z_OFF + l31 * ZCO + rh * 128;
    const char* zqY = smem_wxh + Cfg::Z_OFF + ZPL + l31 * ZCO + rh * 128 + khalf * 16;

    f32x16 acc[3][WN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nj = 0; nj < WN; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nj][r] = 0.f;

    // the shifts follow the tile maxima (conv_x3h.hip): moved when a maximum would leave [2^1, 2^15) after scaling
    // (lo: the shift the largest tile so far asked for -- the shift never runs more than 2^64 ahead of it, conv_x3h.hip)
    int lox = 0, loz = 0;
    auto follow1 = [&](int e, int sh, int& lo, bool first) -> int {
        const int need = 140 - (e < 14 ? 14 : e);
        lo = (first || need < lo) ? need : lo;
        if (first || need < sh - 1) return need;
        if (need > sh + 12) {
            int n = need < sh + 64 ? need : sh + 64;
            n = n < lo + 64 ? n : lo + 64;
            return n > sh ? n : sh;
        }
        return sh;
    };
    auto follow = [&](bool first) {
        const int* M = reinterpret_cast<const int*>(smem_wxh + Cfg::M_OFF);
        if (dbg & 8) { if (first) { shx = shz = 0; scx = scz = 1.f; } return; }
        const int ex = __builtin_amdgcn_readfirstlane(max(max(max(M[0], M[1]), max(M[2], M[3])), max(M[4], M[5]))) >> 23;
        const int ez = __builtin_amdgcn_readfirstlane(max(max(max(M[8], M[9]), max(M[10], M[11])), max(M[12], M[13]))) >> 23;
        const int nx = follow1(ex, shx, lox, first), nz = follow1(ez, shz, loz, first);
        if (nx != shx || nz != shz) {
            if (!first) {
                const int dx = nx - shx, dz_ = nz - shz;                 // each <= 64; a large negative one flushes the old sums
                const float fx = dx < -126 ? 0.f : wxh_pow2(dx), fz = dz_ < -126 ? 0.f : wxh_pow2(dz_);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int nj = 0; nj < WN; ++nj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][nj][r] = (acc[t][nj][r] * fx) * fz;
            }
            shx = nx; shz = nz;
            scx = wxh_pow2(shx); scz = wxh_pow2(shz);                  // shifts in [-115, 126]
        }
    };

    if (t_begin < t_end) {
        issue_tile(t_begin);
        wait_tile();
        post_max();
        lds_barrier();
        follow(true);
        split_tile();
        lds_barrier();
    }
    for (int pt = t_begin; pt < t_end; ++pt) {
        if (pt + 1 < t_end) issue_tile(pt + 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // X chain: cX = [x1|x2] of 8-pixel group g, nX of group g + 1
            vr_i32x4 cX = *reinterpret_cast<const vr_i32x4*>(xqX + r * 80);
#pragma unroll
            for (int i = 0; i < 2; ++i) {                                  // 16-pixel steps
                const vr_i32x4 mX = *reinterpret_cast<const vr_i32x4*>(xqX + r * 80 + (2 * i + 1) * 16);
                const vr_i32x4 nX = *reinterpret_cast<const vr_i32x4*>(xqX + r * 80 + (2 * i + 2) * 16);
                const vr_i32x4 cY = *reinterpret_cast<const vr_i32x4*>(xqY + r * 80 + (2 * i) * 16);
                const vr_i32x4 nY = *reinterpret_cast<const vr_i32x4*>(xqY + r * 80 + (2 * i + 1) * 16);
                vr_i32x4 Z1a[WN], Z1b[WN], Z2[WN];
#pragma unroll
                for (int nj = 0; nj < WN; ++nj) {
                    const int o = nj * 32 * ZCO + r * 64 + (2 * i) * 16;
                    Z1a[nj] = *reinterpret_cast<const vr_i32x4*>(zqX + o);
                    Z1b[nj] = *reinterpret_cast<const vr_i32x4*>(zqX + o + 16);
                    Z2[nj] = *reinterpret_cast<const vr_i32x4*>(zqY + o);
                }
                // column taps: 0 = the aligned operand, 1 = shifted by one pixel (16 bits), 2 = shifted by one dword
                auto taps = [](const vr_i32x4& c, const vr_i32x4& n, vr_i32x4 (&o)[3]) {
                    o[0] = c;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[1][e] = (int)__builtin_amdgcn_alignbit((unsigned)(e < 3 ? c[e + 1] : n[0]), (unsigned)c[e], 16u);
                    o[2] = vr_i32x4{c[1], c[2], c[3], n[0]};
                };
                vr_i32x4 Xa[3], Xb[3], Yv[3];
                taps(cX, mX, Xa);
                taps(mX, nX, Xb);
                taps(cY, nY, Yv);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int nj = 0; nj < WN; ++nj) {
                        acc[t][nj] = wxh_mfma(Yv[t], Z2[nj], acc[t][nj]);
                        acc[t][nj] = wxh_mfma(Xa[t], Z1a[nj], acc[t][nj]);
                        acc[t][nj] = wxh_mfma(Xb[t], Z1b[nj], acc[t][nj]);
                    }
                cX = nX;
            }
        }
        if (pt + 1 < t_end) {
            wait_tile();
            post_max();
            lds_barrier();                                                 // every wave has read the planes of this tile; maxima posted
            follow(false);
            split_tile();
            lds_barrier();
        }
    }

    // ---------------- the two halves of each kernel row meet in LDS (same shifts in every wave: plain sums) -------------------------
    __syncthreads();                                                      // the planes of the last tile have been read
    {
        float* R = reinterpret_cast<float*>(smem_wxh) + (ty * 3 * WN * 16) * 64 + lane;     // [ty][t][nj][r][lane]
        if (rh == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int nj = 0; nj < WN; ++nj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) R[((t * WN + nj) * 16 + r) * 64] = acc[t][nj][r];
        }
        __syncthreads();
        if (rh == 1) return;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nj = 0; nj < WN; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][nj][r] += R[((t * WN + nj) * 16 + r) * 64];
    }
    // ---------------- the block's partial slab [ci][tap][co]: lane = cout, register = input channel -----------------------------
    const float fx = wxh_pow2(-shx), fz = wxh_pow2(-shz);                 // shifts in [-115, 126]
    float* pp = a.part + (long long)p * a.part_stride;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nj = 0; nj < WN; ++nj) {
            const int co = co0 + nj * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = c0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (ci < a.in.Cin && co < a.CoutPad) pp[((long long)ci * 9 + ty * 3 + t) * a.CoutPad + co] = (acc[t][nj][r] * fx) * fz;
            }
        }
}

// ---- host side -----------------------------------------------------------------------------------------------------
bool wgrad_x3h_pick(const WgradArgs& a, const ConvShape& s, int* MT_out) {
    if (!a.x3h || a.bf16 != 3 || !a.allow_wino) return false;
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1)) return false;
    if (a.in.pad_h != 1 || a.in.pad_w != 1 || a.in.Hout != a.in.Hin || a.in.Wout != a.in.Win) return false;
    if (a.in.Win < 32 || (a.in.Win & 1) || a.in.Hin < 2 || (a.zH & 1) || (a.zC & 1) || (a.zN & 1)) return false;
    if ((reinterpret_cast<size_t>(a.dz) & 7) != 0) return false;                            // dz pairs are 8-byte loads
    if ((a.in.nsrc > 1 && a.in.c1 % 32) || (a.in.nsrc > 2 && a.in.c2 % 32)) return false;   // a 32-channel block reads ONE source
    for (int i = 0; i < a.in.nsrc; ++i) {
        const ConvSrc& c = a.in.src[i];
        if (c.aff0 || c.aff1 || c.post || c.up || c.zins || c.slope != 1.f || c.W != a.in.Win) return false;
        if (((long long)c.C * c.sC + (long long)c.H * (c.sH > 0 ? c.sH : 1)) * 4 >= 0x7FFFFFF0LL) return false;
    }
    if (((long long)a.Cout * a.zC + (long long)a.in.Hout * a.zH) * 4 >= 0x7FFFFFF0LL) return false;
    // (a 64-cout block needs 96 accumulator registers beside ~70 of in-flight pixel pairs and offsets: 188 bytes of scratch under the
    // 256-register cap, and a spilled in-flight register would be silently wrong -- 32-cout blocks only)
    *MT_out = 32;
    return true;
}

void wgrad_x3h_plan(WgradArgs& a, int MT) {
    a.tiles_w = (a.in.Wout + 31) / 32;
    a.tiles_h = (a.in.Hout + 3) / 4;
    a.npt = a.in.N * a.tiles_h * a.tiles_w;
    a.nchunks = (a.in.Cin + 31) / 32;
    a.nct = a.CoutPad / MT;
    a.part_stride = (long long)a.in.Cin * 9 * a.CoutPad;
    long long P = 512 / ((long long)a.nchunks * a.nct);         // two 6-wave workgroups per CU: one round
    if (P < 1) P = 1;
    if (P > a.npt) P = a.npt;
    const long long cap = (64LL << 20) / a.part_stride;         // scratch <= 256 MB
    if (P > cap) P = cap < 1 ? 1 : cap;
    a.P = (int)P;
}

template <int MT>
static void wxh_launch(const WgradArgs& a, hipStream_t st) {
    using Cfg = WxhCfg<MT>;
    auto kern = wgrad_x3h_kernel<MT>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

void wgrad_x3h_launch(const WgradArgs& a, int MT, hipStream_t st) {
    VR_CHECK(MT == 32, -2, "wgrad_x3h: 32-cout blocks only");
    wxh_launch<32>(a, st);
}

}  // namespace vr
