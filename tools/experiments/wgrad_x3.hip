// Weight gradient of the 3x3 stride-1 convolutions with fp32 products from six bf16 products (backward of lib/layers.py:12-20 under
// train.py:92; "mfma_mode" 2).  Direct form, reduction over PIXELS on v_mfma_f32_32x32x16_bf16:
//
//   dW[ci][ty][tx][co] = sum over (n, h, w) of  x[n][ci][h + ty - 1][w + tx - 1] * dz[n][co][h][w]
//
// Both operands are produced inside the kernel: every fp32 value is split into three bf16 planes (conv_stage.h: x = x1 + x2 + x3
// exactly) with the PIXELS innermost -- a lane's 16-byte MFMA operand is 8 consecutive pixels of one channel -- and
//   [x1|x2][z1|z1] + [x1|x2][z2|z2] + [x1|x3][z3|z1] = x1z1 + x2z1 + x1z2 + x2z2 + x1z3 + x3z1
// (lanes 0-31 carry k = 0..7, lanes 32-63 k = 8..15: both halves hold the SAME 8 pixels of two planes).  The MFMA work is 9/4 of
// the Winograd F(3x3,2x2) kernel's (wgrad_wino.hip) at 16/6 of the fp32 instruction's rate, without its two operand transforms.
//
// Workgroup = 3 waves, block = (32 input channels, MT couts, one of P contiguous ranges of 2 x 32-pixel dz tiles).  Per tile:
//   * every thread loads ~12 pixel PAIRS of x (4 halo rows x 34 columns per channel) and ~11 of dz (8-byte loads; per-thread static
//     offsets + one scalar tile offset; only edge tiles compute bounds), for the NEXT tile while the current one is multiplied;
//   * split pass: one split3_pair (11 VALU) per pair, three ds_write_b32 into  xP[plane][ci][row][40 px]  /  dzP[plane][co][64 px]
//     (channel pitches 336 / 144 bytes: a 32-lane operand read touches every bank once);
//   * wave ty multiplies kernel row ty: per 8-pixel step the x operands of column tap 0 are one aligned ds_read_b128 (the LDS row
//     starts at image column w0 - 1), tap 1 is a 16-bit funnel shift of two neighbouring operands (4 v_alignbit_b32), tap 2 is a
//     dword renaming of the same registers; 3 taps x (MT/32) x 3 MFMAs per step; accumulators [ci 32][co 32] per (tap, cout block)
//     stay in registers over the whole pixel range;
//   * the 9 taps go to the block's partial slab [ci][tap][co] (lanes along co); wgrad_reduce_kernel sums the P slabs.
#include <cstdlib>

#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

template <int MT>
struct WxCfg {
    static constexpr int CB = 32, TH = 2, TW = 32;
    static constexpr int XR = TH + 2, XPX = 40;                        // x rows of a tile, bf16 elements per row (34 used)
    static constexpr int XCI = XR * XPX * 2 + 16;                      // 336-byte channel pitch
    static constexpr int XPL = CB * XCI;                               // bytes per x plane
    static constexpr int ZCO = TH * TW * 2 + 16;                       // 144-byte cout pitch
    static constexpr int ZPL = MT * ZCO;
    static constexpr int Z_OFF = 3 * XPL;
    static constexpr int LDS_BYTES = 3 * XPL + 3 * ZPL;
    static constexpr int NT = 192;
    static constexpr int XPAIRS = CB * XR * 17, NXP = (XPAIRS + NT - 1) / NT;      // 2176 pairs, 12 passes
    static constexpr int ZPAIRS = MT * TH * 16, NZP = (ZPAIRS + NT - 1) / NT;      // 2048 / 1024 pairs, 11 / 6 passes
    static constexpr int WN = MT / 32;
    static constexpr int OCC = 3 * LDS_BYTES <= 160 * 1024 ? 3 : 2;
};

// Loads are inline asm with hand-placed waits (see conv_x3.hip); the second pixel of a pair rides on the immediate offset.
__device__ __forceinline__ void wx_load_pair(i32x4 rsrc, unsigned v0, unsigned v1, float& a, float& b) {
    asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_dword %1, %3, %4, 0 offen offset:4"
                 : "=&v"(a), "=&v"(b) : "v"(v0), "v"(v1), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void wx_load_x2(i32x4 rsrc, unsigned v, float& a, float& b) {
    vr_f32x2 t;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(t) : "v"(v), "s"(rsrc) : "memory");
    a = t[0]; b = t[1];
}
// after an s_waitcnt: ties the loaded registers to this point so that their uses cannot be scheduled in front of the wait
__device__ __forceinline__ void wx_tie(float& a, float& b, float& c, float& d) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int MT>
__global__ __launch_bounds__(192) void wgrad_x3_kernel(const WgradArgs a) {
    using Cfg = WxCfg<MT>;
    constexpr int NXP = Cfg::NXP, NZP = Cfg::NZP, WN = Cfg::WN, XPL = Cfg::XPL, ZPL = Cfg::ZPL, XCI = Cfg::XCI, ZCO = Cfg::ZCO;
    extern __shared__ __attribute__((aligned(16))) char smem_wx[];

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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..2 = kernel row ty
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    // ---- sources of the block's 32 input channels (virtual concat of up to three plain tensors) ----
    const int s_first = (c0 >= a.in.c1) + (c0 >= a.in.c2);
    const int c_last = (c0 + 31 < a.in.Cin ? c0 + 31 : a.in.Cin - 1);
    const int s_last = (c_last >= a.in.c1) + (c_last >= a.in.c2);
    // ---- static part of this thread's pixel pairs: byte offset inside the source image / dz image (2^31: nothing to load) ----
    unsigned xoff[NXP], xsrc = 0u;                                     // xsrc: 2 bits per pass = source index
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int q = i * Cfg::NT + tid;
        const int cl = q / 68, rem = q - cl * 68;
        const int row = rem / 17, pr = rem - row * 17;
        const int ci = c0 + cl;
        const bool live = q < Cfg::XPAIRS && ci < a.in.Cin;
        const int cj = live ? ci : 0;
        const int si = (cj >= a.in.c1) + (cj >= a.in.c2);
        const int clc = cj - (si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2));
        const long long sC = VR_SEL_F(a.in, si, sC), sH = VR_SEL_F(a.in, si, sH);
        xoff[i] = live ? (unsigned)(((long long)clc * sC + (long long)row * sH + 2 * pr) * 4) : 0x80000000u;
        xsrc |= (unsigned)si << (2 * i);
    }
    unsigned zoff[NZP];
#pragma unroll
    for (int i = 0; i < NZP; ++i) {
        const int q = i * Cfg::NT + tid;
        const int cl = q >> 5, rem = q & 31;
        const bool live = q < Cfg::ZPAIRS && co0 + cl < a.Cout;
        zoff[i] = live ? (unsigned)(((long long)(co0 + cl) * a.zC + (long long)(rem >> 4) * a.zH + 2 * (rem & 15)) * 4) : 0x80000000u;
    }
    const bool multi = s_first != s_last;

    // pixel-pair registers: lanes without a pair (beyond the block / beyond Cin, Cout) are never loaded and stay 0
    float xa[NXP], xb[NXP], za[NZP], zb[NZP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) { xa[i] = 0.f; xb[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < NZP; ++i) { za[i] = 0.f; zb[i] = 0.f; }
    auto issue_tile = [&](int pt) {
        const int n = pt / tiles_per_img;
        const int trem = pt - n * tiles_per_img;
        const int h0 = (trem / a.tiles_w) * 2, w0 = (trem % a.tiles_w) * 32;
        const bool edge = h0 < 1 || h0 + 2 >= a.in.Hin || w0 < 1 || w0 + 32 >= a.in.Win;      // (wave-uniform)
        // x: rows h0 - 1 .. h0 + 2, columns w0 - 1 .. w0 + 32.  The descriptor starts at tile pixel (h0 - 1, w0 - 1) of the source's
        // image n: in front of the tensor for the first tile row / column, where those lanes are masked (edge path)
        for (int si = s_first; si <= s_last; ++si) {
            const float* sp = VR_SEL_F(a.in, si, p);
            const long long sN = VR_SEL_F(a.in, si, sN), sH = VR_SEL_F(a.in, si, sH);
            const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)(h0 - 1) * sH + (w0 - 1), 0x7FFFFFF0u);
#pragma unroll
            for (int i = 0; i < NXP; ++i) {
                const bool mine = xoff[i] != 0x80000000u && (!multi || (int)((xsrc >> (2 * i)) & 3u) == si);
                if (!edge) {
                    if (mine) wx_load_pair(xr, xoff[i], xoff[i], xa[i], xb[i]);
                } else if (mine) {
                    const int q = i * Cfg::NT + tid;
                    const int rem = q % 68;
                    const int hi = h0 - 1 + rem / 17, wi = w0 - 1 + 2 * (rem % 17);
                    const bool okh = hi >= 0 && hi < a.in.Hin;
                    const bool ok0 = okh && wi >= 0 && wi < a.in.Win, ok1 = okh && wi + 1 < a.in.Win;     // (wi + 1 >= 0 always)
                    wx_load_pair(xr, ok0 ? xoff[i] : 0x80000000u, ok1 ? xoff[i] : 0x80000000u, xa[i], xb[i]);
                }
            }
        }
        // dz: rows h0, h0 + 1, columns w0 .. w0 + 31 (even width: a pair is inside or outside as a whole)
        {
            const i32x4 zr = make_rsrc(a.dz + (long long)n * a.zN + (long long)h0 * a.zH + w0, 0x7FFFFFF0u);
            const bool zedge = h0 + 2 > a.in.Hout || w0 + 32 > a.in.Wout;
#pragma unroll
            for (int i = 0; i < NZP; ++i) {
                if (zoff[i] == 0x80000000u) continue;
                unsigned o = zoff[i];
                if (zedge) {
                    const int rem = (i * Cfg::NT + tid) & 31;
                    if (h0 + (rem >> 4) >= a.in.Hout || w0 + 2 * (rem & 15) >= a.in.Wout) o = 0x80000000u;
                }
                wx_load_x2(zr, o, za[i], zb[i]);
            }
        }
    };
    auto wait_tile = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i + 1 < NXP; i += 2) wx_tie(xa[i], xb[i], xa[i + 1], xb[i + 1]);
        if (NXP & 1) wx_tie(xa[NXP - 1], xb[NXP - 1], xa[0], xb[0]);
#pragma unroll
        for (int i = 0; i + 1 < NZP; i += 2) wx_tie(za[i], zb[i], za[i + 1], zb[i + 1]);
        if (NZP & 1) wx_tie(za[NZP - 1], zb[NZP - 1], za[0], zb[0]);
    };
    // split the loaded pairs into the bf16 planes (the caller has waited for the loads)
    auto split_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int q = i * Cfg::NT + tid;
            if ((i + 1) * Cfg::NT <= Cfg::XPAIRS || q < Cfg::XPAIRS) {
                const int cl = q / 68, rem = q - cl * 68;
                int h, m, l;
                split3_pair(xa[i], xb[i], h, m, l);
                int* d = reinterpret_cast<int*>(smem_wx + cl * XCI) + rem / 17 * 20 + rem % 17;
                d[0] = h; d[XPL / 4] = m; d[2 * XPL / 4] = l;
            }
        }
#pragma unroll
        for (int i = 0; i < NZP; ++i) {
            const int q = i * Cfg::NT + tid;
            if ((i + 1) * Cfg::NT <= Cfg::ZPAIRS || q < Cfg::ZPAIRS) {
                int h, m, l;
                split3_pair(za[i], zb[i], h, m, l);
                int* d = reinterpret_cast<int*>(smem_wx + Cfg::Z_OFF + (q >> 5) * ZCO) + (q & 31);
                d[0] = h; d[ZPL / 4] = m; d[2 * ZPL / 4] = l;
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // x operands [x1|x2], [x1|x3]: lanes 0-31 plane 0, lanes 32-63 plane 1 resp. 2; channel l31, row ty + r
    const char* xq0 = smem_wx + khalf * XPL + l31 * XCI + wave * 80;
    const char* xq1 = smem_wx + 2 * khalf * XPL + l31 * XCI + wave * 80;
    // dz operands [z1|z1], [z2|z2], [z3|z1]: cout nj * 32 + l31
    const char* zq0 = smem_wx + Cfg::Z_OFF + l31 * ZCO;
    const char* zq1 = zq0 + ZPL;
    const char* zq2 = smem_wx + Cfg::Z_OFF + (khalf ? 0 : 2) * ZPL + l31 * ZCO;

    f32x16 acc[3][WN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nj = 0; nj < WN; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nj][r] = 0.f;

    if (t_begin < t_end) {
        issue_tile(t_begin);
        wait_tile();
        split_tile();
        lds_barrier();
    }
    for (int pt = t_begin; pt < t_end; ++pt) {
        if (pt + 1 < t_end) issue_tile(pt + 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            vr_i32x4 c0v = *reinterpret_cast<const vr_i32x4*>(xq0 + r * 80), c1v = *reinterpret_cast<const vr_i32x4*>(xq1 + r * 80);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const vr_i32x4 n0v = *reinterpret_cast<const vr_i32x4*>(xq0 + r * 80 + (j + 1) * 16);
                const vr_i32x4 n1v = *reinterpret_cast<const vr_i32x4*>(xq1 + r * 80 + (j + 1) * 16);
                vr_bf16x8 Z0[WN], Z1[WN], Z2[WN];
#pragma unroll
                for (int nj = 0; nj < WN; ++nj) {
                    const int o = nj * 32 * ZCO + r * 64 + j * 16;
                    Z0[nj] = *reinterpret_cast<const vr_bf16x8*>(zq0 + o);
                    Z1[nj] = *reinterpret_cast<const vr_bf16x8*>(zq1 + o);
                    Z2[nj] = *reinterpret_cast<const vr_bf16x8*>(zq2 + o);
                }
                // column taps: 0 = the aligned operand, 1 = shifted by one pixel (16 bits), 2 = shifted by one dword
                vr_i32x4 s0v, s1v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0v[e] = (int)__builtin_amdgcn_alignbit((unsigned)(e < 3 ? c0v[e + 1] : n0v[0]), (unsigned)c0v[e], 16u);
                    s1v[e] = (int)__builtin_amdgcn_alignbit((unsigned)(e < 3 ? c1v[e + 1] : n1v[0]), (unsigned)c1v[e], 16u);
                }
                const vr_i32x4 d0v = vr_i32x4{c0v[1], c0v[2], c0v[3], n0v[0]}, d1v = vr_i32x4{c1v[1], c1v[2], c1v[3], n1v[0]};
                const vr_bf16x8 X0[3] = {__builtin_bit_cast(vr_bf16x8, c0v), __builtin_bit_cast(vr_bf16x8, s0v), __builtin_bit_cast(vr_bf16x8, d0v)};
                const vr_bf16x8 X1[3] = {__builtin_bit_cast(vr_bf16x8, c1v), __builtin_bit_cast(vr_bf16x8, s1v), __builtin_bit_cast(vr_bf16x8, d1v)};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int nj = 0; nj < WN; ++nj) {
                        acc[t][nj] = mfma_bf16x16(X1[t], Z2[nj], acc[t][nj]);
                        acc[t][nj] = mfma_bf16x16(X0[t], Z1[nj], acc[t][nj]);
                        acc[t][nj] = mfma_bf16x16(X0[t], Z0[nj], acc[t][nj]);
                    }
                c0v = n0v; c1v = n1v;
            }
        }
        if (pt + 1 < t_end) {
            lds_barrier();                                                 // every wave has read the planes of this tile
            wait_tile();
            split_tile();
            lds_barrier();
        }
    }

    // ---------------- the block's partial slab [ci][tap][co]: lane = cout, register = input channel -----------------------------
    float* pp = a.part + (long long)p * a.part_stride;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nj = 0; nj < WN; ++nj) {
            const int co = co0 + nj * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = c0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (ci < a.in.Cin && co < a.CoutPad) pp[((long long)ci * 9 + wave * 3 + t) * a.CoutPad + co] = acc[t][nj][r];
            }
        }
}

// ---- host side -----------------------------------------------------------------------------------------------------
bool wgrad_x3_pick(const WgradArgs& a, const ConvShape& s, int* MT_out) {
    static const bool on = [] { const char* e = getenv("VR_WGRAD_X3"); return !e || atoi(e) != 0; }();
    if (!on || a.bf16 != 2) return false;
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1)) return false;
    if (a.in.pad_h != 1 || a.in.pad_w != 1 || a.in.Hout != a.in.Hin || a.in.Wout != a.in.Win) return false;
    if (a.in.Win < 32 || (a.in.Win & 1) || a.in.Hin < 2) return false;
    for (int i = 0; i < a.in.nsrc; ++i) {
        const ConvSrc& c = a.in.src[i];
        if (c.aff0 || c.aff1 || c.post || c.up || c.zins || c.slope != 1.f || c.W != a.in.Win) return false;
        if (((long long)c.C * c.sC + (long long)c.H * (c.sH > 0 ? c.sH : 1)) * 4 >= 0x7FFFFFF0LL) return false;
    }
    if (((long long)a.Cout * a.zC + (long long)a.in.Hout * a.zH) * 4 >= 0x7FFFFFF0LL) return false;
    *MT_out = a.CoutPad % 64 == 0 ? 64 : 32;
    return true;
}

void wgrad_x3_plan(WgradArgs& a, int MT) {
    a.tiles_w = (a.in.Wout + 31) / 32;
    a.tiles_h = (a.in.Hout + 1) / 2;
    a.npt = a.in.N * a.tiles_h * a.tiles_w;
    a.nchunks = (a.in.Cin + 31) / 32;
    a.nct = a.CoutPad / MT;
    a.part_stride = (long long)a.in.Cin * 9 * a.CoutPad;
    long long P = 1024 / ((long long)a.nchunks * a.nct);        // two or three 3-wave workgroups per CU: ~two rounds
    if (P < 1) P = 1;
    if (P > a.npt) P = a.npt;
    const long long cap = (64LL << 20) / a.part_stride;         // scratch <= 256 MB
    if (P > cap) P = cap < 1 ? 1 : cap;
    a.P = (int)P;
}

template <int MT>
static void wx_launch(const WgradArgs& a, hipStream_t st) {
    using Cfg = WxCfg<MT>;
    auto kern = wgrad_x3_kernel<MT>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

void wgrad_x3_launch(const WgradArgs& a, int MT, hipStream_t st) {
    if (MT == 64) wx_launch<64>(a, st); else wx_launch<32>(a, st);
}

}  // namespace vr
