// Winograd F(3x3, 2x2) weight gradient of the 3x3 stride-1 convolutions (backward of lib/layers.py:12-20 under
// train.py:92; ~70 % of the weight-gradient multiply-adds of the net):
//
//   dW[co][ci] (3x3) = sum over 2x2 output tiles of   A^T [ (G dz G^T) (.) (B^T x B) ] A
//
// dz: the 2x2 tile of the gradient at the conv's raw output, x: the 4x4 input patch around it.  16 multiplies per
// tile instead of the 36 of the direct form, so the MFMA work drops 2.25x -- the direct kernel (wgrad_mfma.hip)
// already sits on the fp32 matrix pipe.  The 16 element-wise products are 16 independent GEMMs with the reduction
// over TILES,   M_f[co][ci] += U_f[co][tile] * V_f[tile][ci],   f = 0..15,
// on v_mfma_f32_32x32x2_f32 (rows = couts, columns = input channels, k = 2 tiles per instruction).
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   (the forward kernel's input transform, conv_wino.hip)
//   G   = [1 0; 1/2 1/2; 1/2 -1/2; 0 1]              A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1]
//
// Workgroup = 512 threads / 8 waves, one per CU; block = (CB input channels, MT couts, one of P contiguous ranges of
// 4x16-pixel chunks = 16 tiles).  Per chunk:
//   * raw input rows (6 x 24 floats per channel) and raw dz rows (4 x 16 per cout) arrive by LDS-DMA; wave w fetches
//     input channels w, w+8, ... and couts 4j..4j+3 for j = w, w+8, ...;
//   * the same wave transforms what it fetched (lane = (tile, one of 4 channels)): no cross-wave dependency between
//     the DMA and the transform;
//   * wave w multiplies frequencies 2w and 2w+1; accumulators stay in registers over the whole pixel range;
//   * at the end the 16 frequencies of a (cout, cin) pair meet in LDS, A^T M A, and the 9 taps go to the block's
//     partial slab [ci][tap][co]; wgrad_reduce_kernel (wgrad_mfma.hip) sums the P slabs deterministically.
// fp32 throughout; the transforms only add and halve.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

template <int CB, int MT>
struct WwCfg {
    static constexpr int TH = 4, TW = 16, KT = 16;                     // chunk: 4 x 16 output pixels = 2 x 8 Winograd tiles
    static constexpr int XR = TH + 2, XS0 = 3, TWq = 24, CSX = XR * TWq + 4;   // raw input rows start 4 columns left of the chunk;
                                                                       // +4: pitch 148 spreads the 4 channels of a transform over the banks
    static constexpr int CSZ = TH * TW;                                // raw dz floats per cout
    static constexpr int KP = KT + 1;                                  // odd k pitch: 32 channels -> 32 banks
    static constexpr int XRAW = CB * CSX, ZRAW = MT * CSZ;
    static constexpr int VS = 16 * CB * KP, US = 16 * MT * KP;
    static constexpr int LDS_FLOATS = XRAW + ZRAW + VS + US;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int WM = MT / 32, WN = CB / 32;
    static constexpr int MP = 33;                                      // epilogue exchange pitch
    static_assert(XR * TWq == 36 * 4 && CSZ == 16 * 4, "piece counts below");
    static_assert(16 * 32 * MP <= LDS_FLOATS && LDS_BYTES <= 160 * 1024, "LDS");
};

template <int CB, int MT, bool BF>
__global__ __launch_bounds__(512) void wgrad_wino_kernel(const WgradArgs a) {
    using Cfg = WwCfg<CB, MT>;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, TWq = Cfg::TWq, CSX = Cfg::CSX, CSZ = Cfg::CSZ, KP = Cfg::KP, XS0 = Cfg::XS0,
                  WM = Cfg::WM, WN = Cfg::WN, MP = Cfg::MP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xraw = smem;
    float* Zraw = smem + Cfg::XRAW;
    float* Vs = Zraw + Cfg::ZRAW;                      // [16][CB][KP]
    float* Us = Vs + Cfg::VS;                          // [16][MT][KP]

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int inner = a.nchunks * a.nct;
    const int p = (rr / inner) * 8 + xcd;
    if (p >= a.P) return;
    const int ib = rr % inner;
    const int ct = ib % a.nct, cb = ib / a.nct;
    const int co0 = ct * MT, c0 = cb * CB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    // ---- DMA of one chunk: this wave's input channels and couts ---------------------------------------------------
    const int xq_row = lane / 6, xq_c4 = lane % 6;                     // lanes 0..35: 16-B piece (row, column quad)
    // dz: lane = piece * 4 + cout (piece = row * 4 + column quad): piece-major in LDS, so that the 4 couts of a transform
    // and the two column halves of a quad land in 16 different bank pairs
    const int zq_cs = lane & 3, zq_row = lane >> 4, zq_c4 = (lane >> 2) & 3;
    auto issue_chunk = [&](int pt) {
        const int n = pt / tiles_per_img;
        const int trem = pt - n * tiles_per_img;
        const int h0 = (trem / a.tiles_w) * TH, w0 = (trem % a.tiles_w) * TW;
        {
            const int hi = h0 - 1 + xq_row, wi = w0 - 1 - XS0 + 4 * xq_c4;
            const bool ok = lane < 36 && hi >= 0 && hi < a.in.Hin && wi >= 0 && wi + 3 < a.in.Win;
#pragma unroll
            for (int i = 0; i < CB / 8; ++i) {
                const int cl = (CB / 8) * wave + i;                    // consecutive channels: the transform's 4 are a pitch apart
                const int ci = c0 + cl;
                const bool live = ci < a.in.Cin;
                const int cj = live ? ci : 0;
                const int si = (cj >= a.in.c1) + (cj >= a.in.c2);
                const int clc = cj - (si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2));
                const float* sp = VR_SEL_F(a.in, si, p);
                const long long sN = VR_SEL_F(a.in, si, sN), sC = VR_SEL_F(a.in, si, sC);
                const unsigned sH4 = (unsigned)VR_SEL_F(a.in, si, sH) * 4u;
                const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)clc * sC, live ? 0x7FFFFFF0u : 0u);
                const unsigned vo = ok ? (unsigned)hi * sH4 + (unsigned)(wi * 4) : 0x80000000u;
                if (lane < 36) dma16(lds0 + (unsigned)(cl * CSX * 4), vo, xr);
            }
        }
        {
            const int h = h0 + zq_row, w = w0 + 4 * zq_c4;
            const bool okp = h < a.in.Hout && w + 3 < a.in.Wout;
            const i32x4 zr = make_rsrc(a.dz + (long long)n * a.zN, 0x7FFFFFF0u);
#pragma unroll
            for (int i = 0; i < MT / 32; ++i) {
                const int j = wave + 8 * i;                            // couts 4j .. 4j+3 of the block
                const int cg = co0 + 4 * j + zq_cs;
                const unsigned vo = (okp && cg < a.Cout)
                                        ? (unsigned)(((long long)cg * a.zC + (long long)h * a.zH + w) * 4)
                                        : 0x80000000u;
                dma16(lds0 + (unsigned)((Cfg::XRAW + 4 * j * CSZ) * 4), vo, zr);
            }
        }
    };

    // ---- transforms of what this wave fetched: lane = (tile k, sub-channel) ---------------------------------------
    const int tk = lane & 15, tsub = lane >> 4;
    const int ti = tk >> 3, tj = tk & 7;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto transform = [&]() {
#pragma unroll
        for (int i = 0; i < CB / 32; ++i) {                            // B^T x B of 4 input channels
            if (a.in.dbg & 8) break;
            // Patch columns 1, 2 of tile tj are one aligned 8-byte LDS read; column 0 is column 2 of tile tj-1 and column 3
            // is column 1 of tile tj+1 (DPP lane shifts, no LDS traffic); the row's first / last tile read their outer
            // column themselves.  The LDS pipe is shared by the whole CU and the transform phase does not overlap MFMAs.
            const int cl = (CB / 8) * wave + 4 * i + tsub;
            const float* xp = Xraw + cl * CSX + (2 * ti) * TWq + 2 * tj + XS0;
            float* V = Vs + cl * KP + tk;                              // + f * CB * KP
            if ((a.in.dbg & 32) && lane >= 1) V = Vs + lane;          // (ablation: all stores of a lane to one address)
            const bool edge = tj == 0 || tj == 7;
            const int eoff = tj == 0 ? 0 : 3;
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 mid = *reinterpret_cast<const f32x2*>(xp + r * TWq + 1);
                const float m1 = mid[0], m2 = mid[1];
                d[r][1] = m1; d[r][2] = m2;
                float e = 0.f;
                if (edge) e = xp[r * TWq + eoff];
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m2), 0x111, 0xf, 0xf, false));    // row_shr:1
                const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m1), 0x101, 0xf, 0xf, false));   // row_shl:1
                d[r][0] = tj == 0 ? e : left;
                d[r][3] = tj == 7 ? e : right;
            }
            float t[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[(r * 4 + 0) * CB * KP] = t[r][0] - t[r][2];
                V[(r * 4 + 1) * CB * KP] = t[r][1] + t[r][2];
                V[(r * 4 + 2) * CB * KP] = t[r][2] - t[r][1];
                V[(r * 4 + 3) * CB * KP] = t[r][1] - t[r][3];
            }
        }
#pragma unroll
        for (int i = 0; i < MT / 32; ++i) {                            // G dz G^T of 4 couts
            if (a.in.dbg & 16) break;
            const int j = wave + 8 * i;
            const int col = 4 * j + tsub;
            // piece-major raw layout of the group: float (piece q, cout, w) at q*16 + cout*4 + w, q = row*4 + (tj>>1)
            const float* zp = Zraw + j * (4 * CSZ) + ((2 * ti) * 4 + (tj >> 1)) * 16 + tsub * 4 + 2 * (tj & 1);
            float* U = Us + col * KP + tk;                             // + f * MT * KP
            const f32x2 g0 = *reinterpret_cast<const f32x2*>(zp);
            const f32x2 g1 = *reinterpret_cast<const f32x2*>(zp + 4 * 16);
            const float g00 = g0[0], g01 = g0[1], g10 = g1[0], g11 = g1[1];
            float t[4][2];
            t[0][0] = g00;                 t[0][1] = g01;
            t[1][0] = 0.5f * (g00 + g10);  t[1][1] = 0.5f * (g01 + g11);
            t[2][0] = 0.5f * (g00 - g10);  t[2][1] = 0.5f * (g01 - g11);
            t[3][0] = g10;                 t[3][1] = g11;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                U[(r * 4 + 0) * MT * KP] = t[r][0];
                U[(r * 4 + 1) * MT * KP] = 0.5f * (t[r][0] + t[r][1]);
                U[(r * 4 + 2) * MT * KP] = 0.5f * (t[r][0] - t[r][1]);
                U[(r * 4 + 3) * MT * KP] = t[r][1];
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][WM][WN];
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[fi][mi][ni][r] = 0.f;

    if (t_begin < t_end) {
        issue_chunk(t_begin);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        transform();
        lds_barrier();
    }
    const float* Ua = Us + (2 * wave) * MT * KP + l31 * KP + khalf;   // + fi*MT*KP + mi*32*KP + 2*s
    const float* Vb = Vs + (2 * wave) * CB * KP + l31 * KP + khalf;   // + fi*CB*KP + ni*32*KP + 2*s
    for (int pt = t_begin; pt < t_end; ++pt) {
        if (pt + 1 < t_end) issue_chunk(pt + 1);       // the raw buffers were last read by this wave's own transform
        if ((a.in.dbg & 64) && pt + 2 < t_end) issue_chunk(pt + 2);   // (timing experiment only: twice the bytes in flight, results are garbage)
        if constexpr (BF) {
            // bf16 operands: two v_mfma_f32_32x32x8_bf16 cover the chunk's 16 tiles of a frequency
            const float* Ub = Us + (2 * wave) * MT * KP + l31 * KP + 4 * khalf;
            const float* Vb2 = Vs + (2 * wave) * CB * KP + l31 * KP + 4 * khalf;
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {
                    s16x4 A[WM], B[WN];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) {
                        const float* q = Ub + fi * MT * KP + mi * 32 * KP + 8 * h8;
                        A[mi] = pack_bf16x4(q[0], q[1], q[2], q[3]);
                    }
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) {
                        const float* q = Vb2 + fi * CB * KP + ni * 32 * KP + 8 * h8;
                        B[ni] = pack_bf16x4(q[0], q[1], q[2], q[3]);
                    }
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < WN; ++ni) acc[fi][mi][ni] = mfma_bf16(A[mi], B[ni], acc[fi][mi][ni]);
                }
        } else if (!(a.in.dbg & 2)) {
            // 16 k-steps (2 frequencies x 8 tile pairs), operands of step s+1 read before the MFMAs of step s
            float av[WM], bv[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) av[mi] = Ua[mi * 32 * KP];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bv[ni] = Vb[ni * 32 * KP];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float avn[WM], bvn[WN];
                if (s + 1 < 16) {
                    const int fi = (s + 1) >> 3, kk = (s + 1) & 7;
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) avn[mi] = Ua[fi * MT * KP + mi * 32 * KP + 2 * kk];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bvn[ni] = Vb[fi * CB * KP + ni * 32 * KP + 2 * kk];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[s >> 3][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[s >> 3][mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 16) {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = avn[mi];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) bv[ni] = bvn[ni];
                }
            }
        }
        if (pt + 1 < t_end) {
            lds_barrier();                                                 // every wave is done reading U, V of this chunk
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's rows of the next chunk have landed
            if (!(a.in.dbg & 1)) transform();
            lds_barrier();
        }
    }

    // ---------------- epilogue: gather the 16 frequencies per (cout, cin) through LDS, A^T M A -> 9 taps ---------------
    float* Mx = smem;                                                  // [16][32 couts][MP]
    float* pp = a.part + (long long)p * a.part_stride;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            lds_barrier();                                             // main loop / previous pass has been read
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;          // cout within the 32-block
                    Mx[((2 * wave + fi) * 32 + row) * MP + l31] = acc[fi][mi][ni][r];
                }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 512 * j;
                const int col = q & 31, cil = q >> 5;                  // lanes along couts: contiguous in the slab
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = Mx[(f * 32 + col) * MP + cil];
                float s0[4], s1[4], s2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = m[c] + m[4 + c] + m[8 + c];
                    s1[c] = m[4 + c] - m[8 + c];
                    s2[c] = m[4 + c] + m[8 + c] - m[12 + c];
                }
                float y[9];
                y[0] = s0[0] + s0[1] + s0[2]; y[1] = s0[1] - s0[2]; y[2] = s0[1] + s0[2] - s0[3];
                y[3] = s1[0] + s1[1] + s1[2]; y[4] = s1[1] - s1[2]; y[5] = s1[1] + s1[2] - s1[3];
                y[6] = s2[0] + s2[1] + s2[2]; y[7] = s2[1] - s2[2]; y[8] = s2[1] + s2[2] - s2[3];
                const int ci = c0 + ni * 32 + cil;
                const int co = co0 + mi * 32 + col;
                if (ci < a.in.Cin && co < a.CoutPad) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) pp[((long long)ci * 9 + t) * a.CoutPad + co] = y[t];
                }
            }
        }
    }
}

// ======================================================================================================================
// Round 4: the same kernel with a REGISTER loader (wgrad_wino_r_kernel).
//
// What bounded the LDS-DMA form (profiles/r04_wgrad_wino_loader.txt): a workgroup has ONE chunk (35-53 KB) in flight, landing in
// the single raw buffer the 16 frequency planes leave room for; its DMA takes ~2.6 us under load and overlaps only the multiply
// phase of the previous chunk, so every chunk pays max(DMA, multiply) + transform -- and 256 CUs with one chunk in flight each draw
// 3.9 TB/s where two in flight draw 5.2.  Here every lane loads the 4 x 4 input patches / 2 x 2 gradient tiles it transforms
// straight into registers (global_load_dwordx2 for the two middle columns of a patch row, one dword for the outer column of the
// chunk's first / last tile; the inner outer columns still come from the neighbour lanes by DPP), TWO chunks ahead: the loads of
// chunk k+2 are issued behind the transform of chunk k and stay in flight across both barriers and the multiply phase.  No raw LDS
// buffers, no DMA wait in front of the transform; zero padding, channels beyond Cin and chunks beyond the block's range read a
// 16-byte zero page instead of being predicated, so every chunk issues the same number of loads and the hand-placed s_waitcnt
// vmcnt(N) are compile-time constants (hipcc does not count inline-asm loads).
// ======================================================================================================================
__device__ __attribute__((aligned(16))) const float kWwZeroPage[4] = {0.f, 0.f, 0.f, 0.f};

typedef float ww_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ww_f32x2 ww_load2(const float* p) {
    ww_f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float ww_load1(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int CB, int MT>
struct WwrCfg {
    static constexpr int TH = 4, TW = 16, KT = 16, KP = KT + 1;
    static constexpr int XI = CB / 32, ZI = MT / 32;                   // transform iterations per wave: 4 channels / 4 couts each
    static constexpr int NL = XI * 8 + ZI * 2;                          // loads per lane and chunk (4 row pairs + 4 edge words per x iteration)
    static constexpr int VS = 16 * CB * KP, US = 16 * MT * KP;
    static constexpr int LDS_FLOATS = VS + US;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int WM = MT / 32, WN = CB / 32, MP = 33;
    static_assert(16 * 32 * MP <= LDS_FLOATS && LDS_BYTES <= 160 * 1024 && 2 * NL < 60, "LDS / vmcnt");
};

template <int CB, int MT>
struct WwRegs {                                                        // what one lane holds of one chunk
    ww_f32x2 xm[WwrCfg<CB, MT>::XI][4];                                // patch rows 0..3, columns 1, 2
    float xe[WwrCfg<CB, MT>::XI][4];                                   // the outer column of the chunk's first / last tile (column 0 / 3)
    ww_f32x2 zg[WwrCfg<CB, MT>::ZI][2];                                // gradient tile rows 0, 1
};

// (32 x 32 blocks leave LDS for two workgroups per CU: one's transform / barrier waits fill with the other's multiply phase)
template <int CB, int MT>
__global__ __launch_bounds__(512, ((CB == 32 && MT == 32) ? 2 : 1)) void wgrad_wino_r_kernel(const WgradArgs a) {
    using Cfg = WwrCfg<CB, MT>;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, KP = Cfg::KP, WM = Cfg::WM, WN = Cfg::WN, MP = Cfg::MP, XI = Cfg::XI, ZI = Cfg::ZI, NL = Cfg::NL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                                  // [16][CB][KP]
    float* Us = Vs + Cfg::VS;                          // [16][MT][KP]

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int inner = a.nchunks * a.nct;
    const int p = (rr / inner) * 8 + xcd;
    if (p >= a.P) return;
    const int ib = rr % inner;
    const int ct = ib % a.nct, cb = ib / a.nct;
    const int co0 = ct * MT, c0 = cb * CB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    // lane = (tile k of the chunk, one of 4 channels / couts)
    const int tk = lane & 15, tsub = lane >> 4;
    const int ti = tk >> 3, tj = tk & 7;
    const float* const zero = kWwZeroPage;

    // ---- per-lane bases: the channels / couts this lane transforms are fixed for the workgroup -------------------------------
    const float* xbase[XI];                            // source plane of the lane's channel (sample 0), or null past Cin
    long long xsN[XI];
    int xsH[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int ci = c0 + (CB / 8) * wave + 4 * i + tsub;
        const bool live = ci < a.in.Cin;
        const int cj = live ? ci : 0;
        const int si = (cj >= a.in.c1) + (cj >= a.in.c2);
        const int cbase = si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2);
        const float* sp = si == 0 ? a.in.src[0].p : (si == 1 ? a.in.src[1].p : a.in.src[2].p);
        const long long sC = si == 0 ? a.in.src[0].sC : (si == 1 ? a.in.src[1].sC : a.in.src[2].sC);
        xsN[i] = si == 0 ? a.in.src[0].sN : (si == 1 ? a.in.src[1].sN : a.in.src[2].sN);
        xsH[i] = (int)(si == 0 ? a.in.src[0].sH : (si == 1 ? a.in.src[1].sH : a.in.src[2].sH));
        xbase[i] = live ? sp + (long long)(cj - cbase) * sC : nullptr;
    }
    const float* zbase[ZI];
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
        const int cg = co0 + 4 * (wave + 8 * i) + tsub;
        zbase[i] = cg < a.Cout ? a.dz + (long long)cg * a.zC : nullptr;
    }

    // ---- loads of one chunk into a register set (always NL load instructions) -------------------------------------------------
    auto load_chunk = [&](int pt, WwRegs<CB, MT>& R) {
        const bool real = pt < t_end && !(a.in.dbg & 8);             // (dbg 8, timing only: every load reads the zero page)
        const int ptc = real ? pt : t_begin;
        const int n = ptc / tiles_per_img;
        const int trem = ptc - n * tiles_per_img;
        const int h0 = (trem / a.tiles_w) * TH, w0 = (trem % a.tiles_w) * TW;
        const int wm = w0 + 2 * tj;                                   // image column of patch column 1
        const int we = tj == 0 ? w0 - 1 : w0 + TW;                    // the outer column this lane may have to fetch itself
        const bool edge_lane = real && (tj == 0 || tj == 7) && we >= 0 && we < a.in.Win;
        const bool mid_ok = real && wm + 1 < a.in.Win;                // (Win % 4 == 0, wm even: the pair is inside or outside together)
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const float* xb = xbase[i] ? xbase[i] + (long long)n * xsN[i] : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hr = h0 - 1 + 2 * ti + r;
                const bool row_ok = xb != nullptr && hr >= 0 && hr < a.in.Hin;
                const float* rowp = xb + (long long)hr * xsH[i];
                R.xm[i][r] = ww_load2((row_ok && mid_ok) ? rowp + wm : zero);
                R.xe[i][r] = ww_load1((row_ok && edge_lane) ? rowp + we : zero);
            }
        }
#pragma unroll
        for (int i = 0; i < ZI; ++i) {
            const float* zb = zbase[i] ? zbase[i] + (long long)n * a.zN : nullptr;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int h = h0 + 2 * ti + r;
                const bool ok = real && zb != nullptr && h < a.in.Hout && wm + 1 < a.in.Wout;
                R.zg[i][r] = ww_load2(ok ? zb + (long long)h * a.zH + wm : zero);
            }
        }
    };
    // the registers of set R have landed when at most NEWER younger loads are outstanding; the asm names every register so that no
    // consumer can be scheduled above the wait
    auto wait_chunk = [&](WwRegs<CB, MT>& R, auto newer) {
        constexpr int NEWER = decltype(newer)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NEWER) : "memory");
#pragma unroll
        for (int i = 0; i < XI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(R.xm[i][r]), "+v"(R.xe[i][r]));
#pragma unroll
        for (int i = 0; i < ZI; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r) asm volatile("" : "+v"(R.zg[i][r]));
    };

    // ---- transforms of what this lane loaded (wgrad_wino_kernel's arithmetic) ---------------------------------------------------
    auto transform = [&](WwRegs<CB, MT>& R) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {                                 // B^T x B of 4 input channels
            const int cl = (CB / 8) * wave + 4 * i + tsub;
            float* V = Vs + cl * KP + tk;                              // + f * CB * KP
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m1 = R.xm[i][r][0], m2 = R.xm[i][r][1];
                d[r][1] = m1; d[r][2] = m2;
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m2), 0x111, 0xf, 0xf, false));    // row_shr:1
                const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m1), 0x101, 0xf, 0xf, false));   // row_shl:1
                d[r][0] = tj == 0 ? R.xe[i][r] : left;
                d[r][3] = tj == 7 ? R.xe[i][r] : right;
            }
            float t[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[(r * 4 + 0) * CB * KP] = t[r][0] - t[r][2];
                V[(r * 4 + 1) * CB * KP] = t[r][1] + t[r][2];
                V[(r * 4 + 2) * CB * KP] = t[r][2] - t[r][1];
                V[(r * 4 + 3) * CB * KP] = t[r][1] - t[r][3];
            }
        }
#pragma unroll
        for (int i = 0; i < ZI; ++i) {                                 // G dz G^T of 4 couts
            const int col = 4 * (wave + 8 * i) + tsub;
            float* U = Us + col * KP + tk;                             // + f * MT * KP
            const float g00 = R.zg[i][0][0], g01 = R.zg[i][0][1], g10 = R.zg[i][1][0], g11 = R.zg[i][1][1];
            float t[4][2];
            t[0][0] = g00;                 t[0][1] = g01;
            t[1][0] = 0.5f * (g00 + g10);  t[1][1] = 0.5f * (g01 + g11);
            t[2][0] = 0.5f * (g00 - g10);  t[2][1] = 0.5f * (g01 - g11);
            t[3][0] = g10;                 t[3][1] = g11;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                U[(r * 4 + 0) * MT * KP] = t[r][0];
                U[(r * 4 + 1) * MT * KP] = 0.5f * (t[r][0] + t[r][1]);
                U[(r * 4 + 2) * MT * KP] = 0.5f * (t[r][0] - t[r][1]);
                U[(r * 4 + 3) * MT * KP] = t[r][1];
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][WM][WN];
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[fi][mi][ni][r] = 0.f;

    const float* Ua = Us + (2 * wave) * MT * KP + l31 * KP + khalf;   // + fi*MT*KP + mi*32*KP + 2*s
    const float* Vb = Vs + (2 * wave) * CB * KP + l31 * KP + khalf;   // + fi*CB*KP + ni*32*KP + 2*s
    auto multiply = [&]() {
        // 16 k-steps (2 frequencies x 8 tile pairs), operands of step s+1 read before the MFMAs of step s
        float av[WM], bv[WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) av[mi] = Ua[mi * 32 * KP];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) bv[ni] = Vb[ni * 32 * KP];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float avn[WM], bvn[WN];
            if (s + 1 < 16) {
                const int fi = (s + 1) >> 3, kk = (s + 1) & 7;
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) avn[mi] = Ua[fi * MT * KP + mi * 32 * KP + 2 * kk];
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) bvn[ni] = Vb[fi * CB * KP + ni * 32 * KP + 2 * kk];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[s >> 3][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[s >> 3][mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < 16) {
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) av[mi] = avn[mi];
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) bv[ni] = bvn[ni];
            }
        }
    };

    // chunk pt: [wait its registers] transform -> U, V | loads of chunk pt+2 | barrier | multiply | barrier
    WwRegs<CB, MT> R0, R1;
    using NLc = std::integral_constant<int, NL>;
    if (t_begin < t_end) {
        load_chunk(t_begin, R0);
        load_chunk(t_begin + 1, R1);
    }
    const int dbg = a.in.dbg;                           // perf experiments only (VR_WW_DBG): 1 no transform, 2 no MFMA, 8 loads hit the zero page only
    for (int pt = t_begin; pt < t_end; pt += 2) {
        wait_chunk(R0, NLc{});                          // chunk pt landed (chunk pt+1 may be in flight)
        if (!(dbg & 1)) transform(R0);
        load_chunk(pt + 2, R0);
        lds_barrier();
        if (!(dbg & 2)) multiply();
        if (pt + 1 < t_end) {
            lds_barrier();                              // every wave is done reading U, V of chunk pt
            wait_chunk(R1, NLc{});
            if (!(dbg & 1)) transform(R1);
            load_chunk(pt + 3, R1);
            lds_barrier();
            if (!(dbg & 2)) multiply();
        }
        if (pt + 2 < t_end) lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the prefetches past the block's range: zero-page loads nobody else waits for)

    // ---------------- epilogue: gather the 16 frequencies per (cout, cin) through LDS, A^T M A -> 9 taps ---------------
    float* Mx = smem;                                                  // [16][32 couts][MP]
    float* pp = a.part + (long long)p * a.part_stride;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            lds_barrier();                                             // main loop / previous pass has been read
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;          // cout within the 32-block
                    Mx[((2 * wave + fi) * 32 + row) * MP + l31] = acc[fi][mi][ni][r];
                }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 512 * j;
                const int col = q & 31, cil = q >> 5;                  // lanes along couts: contiguous in the slab
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = Mx[(f * 32 + col) * MP + cil];
                float s0[4], s1[4], s2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = m[c] + m[4 + c] + m[8 + c];
                    s1[c] = m[4 + c] - m[8 + c];
                    s2[c] = m[4 + c] + m[8 + c] - m[12 + c];
                }
                float y[9];
                y[0] = s0[0] + s0[1] + s0[2]; y[1] = s0[1] - s0[2]; y[2] = s0[1] + s0[2] - s0[3];
                y[3] = s1[0] + s1[1] + s1[2]; y[4] = s1[1] - s1[2]; y[5] = s1[1] + s1[2] - s1[3];
                y[6] = s2[0] + s2[1] + s2[2]; y[7] = s2[1] - s2[2]; y[8] = s2[1] + s2[2] - s2[3];
                const int ci = c0 + ni * 32 + cil;
                const int co = co0 + mi * 32 + col;
                if (ci < a.in.Cin && co < a.CoutPad) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) pp[((long long)ci * 9 + t) * a.CoutPad + co] = y[t];
                }
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------
struct WwPick { int CB, MT; };

static bool ww_enabled() {
    static const bool on = [] { const char* e = getenv("VR_WGRAD_WINO"); return !e || atoi(e) != 0; }();
    return on;
}

// True when the launch can take the Winograd weight-gradient kernel: 3x3 stride-1 undilated conv, every source a plain
// tensor with 16-byte aligned rows (the training executor materialises them), 16-byte aligned dz rows.
bool wgrad_wino_pick(const WgradArgs& a, const ConvShape& s, int* CB_out, int* MT_out) {
    if (!ww_enabled() || !a.allow_wino) return false;
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1)) return false;
    // experiment knobs (round 6): layers with few input / output channels to the direct weight-gradient kernels -- the Winograd form
    // transforms dz for a full 32-cout block and x for a full 32-channel block whatever the layer has
    static const int min_cin = getenv("VR_WW_MIN_CIN") ? atoi(getenv("VR_WW_MIN_CIN")) : 0;
    static const int min_cout = getenv("VR_WW_MIN_COUT") ? atoi(getenv("VR_WW_MIN_COUT")) : 0;
    if (a.in.Cin < min_cin || a.Cout < min_cout) return false;
    if (a.in.pad_h != 1 || a.in.pad_w != 1 || a.in.Hout != a.in.Hin || a.in.Wout != a.in.Win) return false;
    if ((a.in.Win & 3) || a.in.Win < 16 || a.in.Hin < 2) return false;
    for (int i = 0; i < a.in.nsrc; ++i) {
        const ConvSrc& c = a.in.src[i];
        if (c.aff0 || c.aff1 || c.post || c.up || c.zins || c.slope != 1.f || c.W != a.in.Win) return false;
        if ((c.sH & 3) || (c.sC & 3) || (c.sN & 3) || (reinterpret_cast<uintptr_t>(c.p) & 15)) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
    }
    if ((a.zH & 3) || (a.zC & 3) || (a.zN & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15)) return false;
    if ((long long)a.Cout * a.zC * 4 >= 0x7FFFFFF0LL) return false;
    const bool m64 = a.CoutPad % 64 == 0;
    *MT_out = m64 ? 64 : 32;
    *CB_out = (!m64 && a.in.Cin > 32) ? 64 : 32;
    // register loader (round 4): no raw LDS buffers, so 64 x 64 blocks fit beside the 16 frequency planes -- half the dz re-reads,
    // a third fewer transform iterations and half the barriers per multiply-add of the layers with >= 64 input channels
    static const bool b64 = [] { const char* e = getenv("VR_WW_B64"); return !e || atoi(e) != 0; }();
    static const bool regl = [] { const char* e = getenv("VR_WW_REG"); return !e || atoi(e) != 0; }();
    if (b64 && regl && a.bf16 != 1 && m64 && a.in.Cin > 32) *CB_out = 64;
    return true;
}

void wgrad_wino_plan(WgradArgs& a, int CB, int MT) {
    a.tiles_w = (a.in.Wout + 15) / 16;
    a.tiles_h = (a.in.Hout + 3) / 4;
    a.npt = a.in.N * a.tiles_h * a.tiles_w;
    a.nchunks = (a.in.Cin + CB - 1) / CB;
    a.nct = a.CoutPad / MT;
    a.part_stride = (long long)a.in.Cin * 9 * a.CoutPad;
    const long long inner = (long long)a.nchunks * a.nct;
    const long long cap = std::max<long long>(1, (64LL << 20) / a.part_stride);       // scratch <= 256 MB
    static const int ptarget = getenv("VR_WW_PTARGET") ? atoi(getenv("VR_WW_PTARGET")) : 0;
    long long P;
    if (ptarget > 0) {                                        // rounds 2-5 (and the knob): ~ptarget workgroups whatever the shape
        P = ptarget / inner;
        if (P < 1) P = 1;
        if (P > a.npt) P = a.npt;
        if (P > cap) P = cap;
    } else {
        // Round 6: the number of pixel ranges from the dispatch geometry.  One workgroup per CU (139 / 104 KB of LDS; two for the 32 x 32 block), block b on XCD b % 8, 32 CUs
        // per XCD; range p sits on XCD p % 8 with its `inner` (channel block, cout block) workgroups.  The busiest XCD carries
        // ceil(P / 8) * inner workgroups = `rounds` passes over its 32 CUs, each as long as a range (npt / P chunks + the epilogue).  "512
        // workgroups" gave a 192 -> 64 layer (inner = 3) P = 170: 66 workgroups on XCD 0 and 1 = THREE passes of 96 chunks where P = 168
        // takes two (measured with VR_WW_PTARGET: 1905 -> 1400 us at P = 256, three passes of 64).  Ties: the fewest slabs.
        const long long pmax = std::min(std::min<long long>(a.npt, cap), std::max<long long>(8, 1024 / inner));
        const double ovh = 4.0;                               // epilogue + prologue of a workgroup, in chunks
        double best = 1e30;
        P = 1;
        const long long slots = (CB == 32 && MT == 32) ? 64 : 32;     // (the 32 x 32 block's 70 KB of LDS fit a CU twice)
        for (long long c = 1; c <= pmax; ++c) {
            const long long rounds = ((c + 7) / 8 * inner + slots - 1) / slots;
            const double cost = (double)rounds * ((double)((a.npt + c - 1) / c) + ovh);
            if (cost < best * 0.995) { best = cost; P = c; }
        }
    }
    a.P = (int)P;
}

template <int CB, int MT, bool BF>
static void ww_launch(const WgradArgs& a, hipStream_t st) {
    using Cfg = WwCfg<CB, MT>;
    auto kern = wgrad_wino_kernel<CB, MT, BF>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int CB, int MT>
static void wwr_launch(const WgradArgs& a, hipStream_t st) {
    using Cfg = WwrCfg<CB, MT>;
    auto kern = wgrad_wino_r_kernel<CB, MT>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

void wgrad_wino_launch(const WgradArgs& a_in, int CB, int MT, hipStream_t st) {
    WgradArgs a = a_in;
    static const int dbg = [] { const char* e = getenv("VR_WW_DBG"); return e ? atoi(e) : 0; }();   // ablations (perf only)
    a.in.dbg = dbg;
    static const bool reg_loader = [] { const char* e = getenv("VR_WW_REG"); return !e || atoi(e) != 0; }();
    if (a.bf16 != 1 && reg_loader) {
        if (CB == 64 && MT == 64) wwr_launch<64, 64>(a, st);
        else if (CB == 32 && MT == 64) wwr_launch<32, 64>(a, st);
        else if (CB == 64 && MT == 32) wwr_launch<64, 32>(a, st);
        else wwr_launch<32, 32>(a, st);
        return;
    }
    if (a.bf16 == 1) {
        if (CB == 32 && MT == 64) ww_launch<32, 64, true>(a, st);
        else if (CB == 64 && MT == 32) ww_launch<64, 32, true>(a, st);
        else ww_launch<32, 32, true>(a, st);
    } else {
        if (CB == 32 && MT == 64) ww_launch<32, 64, false>(a, st);
        else if (CB == 64 && MT == 32) ww_launch<64, 32, false>(a, st);
        else ww_launch<32, 32, false>(a, st);
    }
}

}  // namespace vr
