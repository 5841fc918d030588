// Winograd F(2x2, 3x3) convolution for the stride-1 3x3 layers with plain inputs (lib/layers.py:12-20,
// the layers that hold ~80 % of the network's multiply-adds).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplies per 2x2 outputs instead of 36: the MFMA work drops 2.25x, which is the only way past the
// fp32 matrix-pipe roof the direct kernel (conv_dma.hip) already sits at on the big layers.  The 16
// element-wise products become 16 independent GEMMs over the input channels,
//   M_f[cout][tile] += U_f[cout][cin] * V_f[cin][tile],        f = 0..15,
// run on v_mfma_f32_32x32x2_f32 exactly like the direct conv (cout = rows, 32 tiles = columns).
//
// Workgroup = 512 threads / 8 waves, output tile 8 x 32 pixels (= 64 Winograd tiles), MT couts,
// input channels in chunks of 8:
//   * raw input rows and the pre-transformed weights U (model.hip keeps G g G^T per layer) arrive by
//     LDS-DMA (lds_dma.h); wave w fetches input channel w of the chunk and 1/8 of the weight slab;
//   * wave w transforms ITS OWN channel (lane = tile; 16 LDS reads, 32 adds, 16 LDS writes) -- no
//     cross-wave dependency between the DMA and the transform, so one barrier per chunk suffices;
//   * wave w multiplies frequencies 2w and 2w+1 (accumulators 2 x MT/32 x 2 tiles of 32x32);
//   * epilogue (conv_wino_epi.h): the 16 frequencies of a (cout, tile) pair live in 8 different waves, so
//     they meet in LDS (32 couts x 32 tiles per pass), A^T M A, bias / folded BatchNorm / activation,
//     8-byte stores, BatchNorm partial sums in training.
// fp32 throughout; the transforms only add and halve, the measured deviation from the direct kernel
// is ~1e-6 relative (tests/test_gpu_parity.py::test_conv_winograd_vs_torch).
// Measured with the SQ counters (profiles/r02_*_sq_pmc.md): the matrix pipe is busy 0.50 (64 couts) / 0.41 (32 couts) of
// the kernel's time; transform, DMA issue, the per-chunk barriers and the epilogue are serial with the MFMA phase.
#include <cstdlib>

#include "conv_wino_epi.h"
#include "kernels.h"

namespace vr {

// MODE: 0 = v_mfma_f32_32x32x2_f32 (exact fp32 products); 1 = bf16 operands (ConvArgs::bf16 == 1); 2 = fp32 products as six
// bf16 products of three-way split operands (conv_stage.h).  Mode 2 keeps BOTH operands in LDS as bf16 planes, 8 input
// channels of a (frequency, plane, row) contiguous, so the multiply phase is ds_read + MFMA only:
//   * U arrives pre-split (wino_weights6_kernel): [f][plane][cout][8 ch] = one 16-byte operand per lane;
//   * V is split by the transform itself: the wave transforms 4 channels x 16 tiles (lane row = channel), a 4x4 register
//     transpose over the lane rows (v_permlane32_swap + v_permlane16_swap, 16 instructions) leaves every lane with four
//     frequencies of all four channels of its tile, which it splits pairwise (11 VALU per pair) and stores as
//     [f][plane][channel group][tile][4 ch] -- 12 ds_write_b64, read back as ds_read_b64.
template <int MT, int MODE>
struct WinoCfg {
    static constexpr int TH = 8, TW = 32, CK = 8, NT = 64;           // pixels, channels per chunk, Winograd tiles
    static constexpr int TH_in = TH + 2, XS0 = 3, TWq = 40, CSX = TH_in * TWq;
    static constexpr int WM = MT / 32;
    static constexpr int XS = CK * CSX;                                // raw input rows (single buffer)
    static constexpr bool X6 = MODE == 2;
    static constexpr int WS = X6 ? 16 * 3 * MT * 4 : 16 * CK * MT;     // U slab   [f][cl][m] floats;  X6: [f][plane][m][8 x bf16]
    static constexpr int VS = X6 ? 16 * 3 * 2 * NT * 2 : 16 * CK * NT;   // V slab   [f][cl][tile] floats;  X6: [f][plane][cg][tile][4 x bf16]
    // 32 couts per workgroup: half the MFMA work per block, so the prologue (first DMA) and the epilogue weigh
    // twice as much and must overlap ANOTHER workgroup's main loop.  One V buffer (at the price of a second
    // barrier per chunk) brings the LDS footprint to 78 KB = two workgroups per CU.
    static constexpr int NVB = (MT == 32 || X6) ? 1 : 2;             // (X6: the U planes are 1.5x the fp32 slab)
    static constexpr int LDS_FLOATS = XS + 2 * WS + NVB * VS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NPIECE = CSX / 4, NPASS = (NPIECE + 63) / 64;
    static constexpr int NWP = WS / 4, NWPASS = NWP / 512;             // 16-B weight pieces per wave-pass
    static_assert(NWP % 512 == 0, "weight slab splits evenly over 8 waves");
    static_assert(wino_epilogue_floats(MT) <= LDS_FLOATS && LDS_BYTES <= 160 * 1024, "LDS");
};

template <int MT, int MODE>     // separate instantiations: a run-time branch around the MFMA loops costs the fp32 kernel
                                // 60 spilled registers
__global__ __launch_bounds__(512) void conv_wino_kernel(const ConvArgs a) {
    using Cfg = WinoCfg<MT, MODE>;
    constexpr bool BF = MODE == 1, X6 = MODE == 2;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, CK = Cfg::CK, NT = Cfg::NT, TWq = Cfg::TWq, CSX = Cfg::CSX, XS0 = Cfg::XS0,
                  WM = Cfg::WM, NPIECE = Cfg::NPIECE, NPASS = Cfg::NPASS, NWPASS = Cfg::NWPASS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xraw = smem;
    float* WsB = smem + Cfg::XS;                       // two U buffers
    float* VsB = WsB + 2 * Cfg::WS;                    // two V buffers

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // each XCD walks its own contiguous range of pixel tiles: neighbouring tiles share halo lines through that XCD's L2
    // (conv_x3.hip; measured: HBM fetch 2.8x -> 1.1x of the input on the full-resolution layers, no time difference on the fp32
    // pipe); VR_CONV_DBG=16: tiles interleaved over the XCDs
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
    const int hbase = h0 - 1, wal0 = w0 - 1 - XS0;
    const int nchunk = (a.Cin + CK - 1) / CK;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // per-lane source coordinates of this wave's input pieces (see conv_dma.hip)
    unsigned hrow[NPASS], wcol4[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        const int hh = q / (TWq / 4), j = q % (TWq / 4);
        const int hi = hbase + hh, wi = wal0 + 4 * j;
        const bool ok = q < NPIECE && hi >= 0 && hi < a.Hin && wi >= 0 && wi + 3 < a.Win;
        hrow[p] = ok ? (unsigned)hi : 0u;
        wcol4[p] = ok ? (unsigned)(wi * 4) : 0x80000000u;
    }
    // weight pieces: LDS order [f][cl][m], source U[(cl*16+f)*CoutPad + m]
    // (X6: LDS order [f][plane][m] in 16-byte operands, source U6[chunk][(f*3+plane)*CoutPad + m])
    unsigned woff[NWPASS];
#pragma unroll
    for (int i = 0; i < NWPASS; ++i) {
        const int q = (wave + 8 * i) * 64 + lane;
        if constexpr (X6) {
            const int m = q % MT, fp = q / MT;
            woff[i] = (unsigned)((fp * a.CoutPad + m) * 16);
        } else {
            const int m4 = q % (MT / 4), t2 = q / (MT / 4);
            const int cl = t2 % CK, f = t2 / CK;
            woff[i] = (unsigned)(((cl * 16 + f) * a.CoutPad + m4 * 4) * 4);
        }
    }

    auto issue_u = [&](int k, int i0, int istep) {                  // weight pieces i0, i0 + istep, ... of this wave, chunk k
        const int c0 = k * CK;
        const unsigned ws_b = lds0 + (unsigned)((Cfg::XS + (k & 1) * Cfg::WS) * 4);
        if (a.dbg == 7) {                                                            // (ablation: no weight DMA)
        } else if constexpr (X6) {
            const long long chunk_bytes = 48LL * a.CoutPad * 16;
            const char* wb = static_cast<const char*>(a.wino6) + k * chunk_bytes + (long long)co0 * 16;
            const i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), (unsigned)(chunk_bytes - (long long)co0 * 16));
#pragma unroll
            for (int i = 0; i < NWPASS; ++i)
                if (i >= i0 && (i - i0) % istep == 0) dma16(ws_b + (wave + 8 * i) * 1024, woff[i], wr);
        } else {
            const float* wb = a.wino + (long long)c0 * 16 * a.CoutPad + co0;
            const i32x4 wr = make_rsrc(wb, (unsigned)(((long long)(a.Cin - c0) * 16 * a.CoutPad - co0) * 4));
#pragma unroll
            for (int i = 0; i < NWPASS; ++i) dma16(ws_b + (wave + 8 * i) * 1024, woff[i], wr);
        }
    };
    auto issue_x = [&](int k) {
        const int c0 = k * CK;
        const int cl = wave;                               // this wave's input channel of the chunk
        const int ci = c0 + cl;
        if (ci >= a.Cin) {
            float* z = Xraw + cl * CSX;
            for (int e = lane; e < CSX; e += 64) z[e] = 0.f;
            return;
        }
        const int si = (ci >= a.c1) + (ci >= a.c2);
        const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
        const float* sp = si == 0 ? a.src[0].p : (si == 1 ? a.src[1].p : a.src[2].p);
        const long long sN = si == 0 ? a.src[0].sN : (si == 1 ? a.src[1].sN : a.src[2].sN);
        const long long sC = si == 0 ? a.src[0].sC : (si == 1 ? a.src[1].sC : a.src[2].sC);
        const unsigned sH4 = (unsigned)(si == 0 ? a.src[0].sH : (si == 1 ? a.src[1].sH : a.src[2].sH)) * 4u;
        const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)clc * sC, 0x7FFFFFF0u);
        const unsigned cb = lds0 + (unsigned)(cl * CSX * 4);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const unsigned vo = hrow[p] * sH4 + wcol4[p];
            if ((p + 1) * 64 <= NPIECE) dma16(cb + p * 1024, vo, xr);
            else if (p * 64 + lane < NPIECE) dma16(cb + p * 1024, vo, xr);
        }
    };
    // X6 issues the input rows first: they are needed by the transform, the weights only one barrier later (split waits below)
    auto issue_chunk = [&](int k) {
        if (a.dbg == 6) return;                                                      // (ablation: no DMA at all)
        if constexpr (X6) { issue_x(k); if (k == 0) issue_u(k, 0, 1); }          // (later chunks: between the MFMA groups)
        else { issue_u(k, 0, 1); issue_x(k); }
    };

    // input transform of this wave's channel: lane = Winograd tile (ti = lane >> 4, tj = lane & 15)
    // (X6: wave = (channel group cg = wave & 1, tile row ti = wave >> 1), lane = (channel 4 cg + (lane >> 4), tj = lane & 15))
    const int xpo = X6 ? (4 * (wave & 1) + (lane >> 4)) * CSX + (2 * (wave >> 1)) * TWq + 2 * (lane & 15) + XS0
                       : wave * CSX + (2 * (lane >> 4)) * TWq + 2 * (lane & 15) + XS0;
    auto transform = [&](int k) {
        const float* xp = Xraw + xpo;
        float* V = VsB + (k % Cfg::NVB) * Cfg::VS + wave * NT + lane;      // + f * CK * NT
        float t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = xp[c], d1 = xp[TWq + c], d2 = xp[2 * TWq + c], d3 = xp[3 * TWq + c];
            t[0][c] = d0 - d2;
            t[1][c] = d1 + d2;
            t[2][c] = d2 - d1;
            t[3][c] = d1 - d3;
        }
        if constexpr (X6) {
            unsigned v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r * 4 + 0] = __float_as_uint(t[r][0] - t[r][2]);
                v[r * 4 + 1] = __float_as_uint(t[r][1] + t[r][2]);
                v[r * 4 + 2] = __float_as_uint(t[r][2] - t[r][1]);
                v[r * 4 + 3] = __float_as_uint(t[r][1] - t[r][3]);
            }
            // 4x4 transpose over the lane rows: afterwards the lane of row rho holds, for i = 0..3, frequency 4 rho + i of the
            // wave's four channels in (v[i], v[4+i], v[8+i], v[12+i])
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = (i & 3) + 4 * (i >> 2);                      // (0..3) with +8, (4..7) with +8
                const vr_u32x2 q = __builtin_amdgcn_permlane32_swap(v[j], v[j + 8], false, false);
                v[j] = q[0]; v[j + 8] = q[1];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = (i & 3) + 8 * (i >> 2);                      // (0..3) with +4, (8..11) with +4
                const vr_u32x2 q = __builtin_amdgcn_permlane16_swap(v[j], v[j + 4], false, false);
                v[j] = q[0]; v[j + 4] = q[1];
            }
            // dword offset of (f = 4 rho, plane 0, cg, tile): ((f * 3 + plane) * 2 + cg) * NT * 2 + tile * 2
            unsigned* V6 = reinterpret_cast<unsigned*>(VsB) + ((4 * (lane >> 4)) * 3 * 2 + (wave & 1)) * NT * 2 +
                           ((wave >> 1) * 16 + (lane & 15)) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int h0, m0, l0, h1, m1, l1;
                split3_pair(__uint_as_float(v[i]), __uint_as_float(v[4 + i]), h0, m0, l0);
                split3_pair(__uint_as_float(v[8 + i]), __uint_as_float(v[12 + i]), h1, m1, l1);
                unsigned* q = V6 + i * 3 * 2 * NT * 2;
                *reinterpret_cast<vr_i32x2*>(q) = vr_i32x2{h0, h1};
                *reinterpret_cast<vr_i32x2*>(q + 2 * NT * 2) = vr_i32x2{m0, m1};
                *reinterpret_cast<vr_i32x2*>(q + 4 * NT * 2) = vr_i32x2{l0, l1};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[(r * 4 + 0) * CK * NT] = t[r][0] - t[r][2];
                V[(r * 4 + 1) * CK * NT] = t[r][1] + t[r][2];
                V[(r * 4 + 2) * CK * NT] = t[r][2] - t[r][1];
                V[(r * 4 + 3) * CK * NT] = t[r][1] - t[r][3];
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][WM][2];
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[fi][mi][ni][r] = 0.f;

    issue_chunk(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (X6) lds_barrier();                     // the transform reads channels other waves fetched
    transform(0);
    lds_barrier();

    const int aoff = khalf * MT + l31;          // + (f*CK + 2kk)*MT + mi*32
    const int boff = khalf * NT + l31;          // + (f*CK + 2kk)*NT + ni*32
    for (int k = 0; k < nchunk; ++k) {
        if (k + 1 < nchunk) issue_chunk(k + 1);
        const float* Ws = WsB + (k & 1) * Cfg::WS + (2 * wave) * CK * MT + aoff;
        const float* Vs = VsB + (k % Cfg::NVB) * Cfg::VS + (2 * wave) * CK * NT + boff;
        if constexpr (X6) {
            // three v_mfma_f32_32x32x16_bf16 per (frequency, cout tile, pixel tile) cover the chunk's 8 input channels
            if (a.dbg != 2) {
                const char* Ub = reinterpret_cast<const char*>(WsB + (k & 1) * Cfg::WS) + (2 * wave) * 3 * MT * 16 + l31 * 16;
                const char* Vb = reinterpret_cast<const char*>(VsB) + (2 * wave) * 3 * 2 * NT * 8 + l31 * 8;
                const int a12 = khalf * MT * 16, a13 = khalf * 2 * MT * 16;       // lanes 32-63: plane 2 resp. plane 3 of U
                const int b3 = khalf ? 0 : 2 * 2 * NT * 8;                        // [b3 | b1]
#pragma unroll
                for (int fi = 0; fi < 2; ++fi) {
                    vr_bf16x8 A12[WM], A13[WM];
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) {
                        const char* q = Ub + (fi * 3 * MT + mi * 32) * 16;
                        A12[mi] = *reinterpret_cast<const vr_bf16x8*>(q + a12);
                        A13[mi] = *reinterpret_cast<const vr_bf16x8*>(q + a13);
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const char* q = Vb + fi * 3 * 2 * NT * 8 + ni * 32 * 8;
                        vr_i32x4 p1, p2, p3;
                        {
                            const vr_i32x2 x0 = *reinterpret_cast<const vr_i32x2*>(q), x1 = *reinterpret_cast<const vr_i32x2*>(q + NT * 8);
                            p1 = vr_i32x4{x0[0], x0[1], x1[0], x1[1]};
                        }
                        {
                            const char* q2 = q + 2 * NT * 8;
                            const vr_i32x2 x0 = *reinterpret_cast<const vr_i32x2*>(q2), x1 = *reinterpret_cast<const vr_i32x2*>(q2 + NT * 8);
                            p2 = vr_i32x4{x0[0], x0[1], x1[0], x1[1]};
                        }
                        {
                            const char* q3 = q + b3;
                            const vr_i32x2 x0 = *reinterpret_cast<const vr_i32x2*>(q3), x1 = *reinterpret_cast<const vr_i32x2*>(q3 + NT * 8);
                            p3 = vr_i32x4{x0[0], x0[1], x1[0], x1[1]};
                        }
                        const vr_bf16x8 B1 = __builtin_bit_cast(vr_bf16x8, p1), B2 = __builtin_bit_cast(vr_bf16x8, p2),
                                        B3 = __builtin_bit_cast(vr_bf16x8, p3);
#pragma unroll
                        for (int mi = 0; mi < WM; ++mi) acc[fi][mi][ni] = mfma_bf16x16(A12[mi], B1, acc[fi][mi][ni]);
#pragma unroll
                        for (int mi = 0; mi < WM; ++mi) acc[fi][mi][ni] = mfma_bf16x16(A12[mi], B2, acc[fi][mi][ni]);
#pragma unroll
                        for (int mi = 0; mi < WM; ++mi) acc[fi][mi][ni] = mfma_bf16x16(A13[mi], B3, acc[fi][mi][ni]);
                        // the weight pieces of chunk k+1 go out between the MFMA groups: one burst of 8 waves x NWPASS DMA
                        // instructions at the top of the phase stalls it (measured: -7 % kernel time this way)
                        if (k + 1 < nchunk && a.dbg != 6) issue_u(k + 1, fi * 2 + ni, 4);
                    }
                }
            }
        } else if constexpr (BF) {
            // bf16 operands: one v_mfma_f32_32x32x8_bf16 covers the chunk's 8 input channels of a frequency
            const float* Wb = WsB + (k & 1) * Cfg::WS + (2 * wave) * CK * MT + (4 * khalf) * MT + l31;
            const float* Vb = VsB + (k % Cfg::NVB) * Cfg::VS + (2 * wave) * CK * NT + (4 * khalf) * NT + l31;
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {
                s16x4 A[WM], B[2];
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) {
                    const float* q = Wb + fi * CK * MT + mi * 32;
                    A[mi] = pack_bf16x4(q[0], q[MT], q[2 * MT], q[3 * MT]);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float* q = Vb + fi * CK * NT + ni * 32;
                    B[ni] = pack_bf16x4(q[0], q[NT], q[2 * NT], q[3 * NT]);
                }
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[fi][mi][ni] = mfma_bf16(A[mi], B[ni], acc[fi][mi][ni]);
            }
        } else if (a.dbg != 2) {
            // 8 k-steps (2 frequencies x 4 channel pairs), operands of step s+1 read before the MFMAs of step s
            float av[WM], bv[2];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) av[mi] = Ws[mi * 32];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bv[ni] = Vs[ni * 32];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float avn[WM], bvn[2];
                if (s + 1 < 8) {
                    const int fi = (s + 1) >> 2, kk = (s + 1) & 3;
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) avn[mi] = Ws[(fi * CK + 2 * kk) * MT + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) bvn[ni] = Vs[(fi * CK + 2 * kk) * NT + ni * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[s >> 2][mi][ni] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[s >> 2][mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 8) {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) av[mi] = avn[mi];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) bv[ni] = bvn[ni];
                }
            }
        }
        if (k + 1 < nchunk) {
            if constexpr (X6) {
                // this wave's channel of chunk k+1 has landed (its NWPASS weight pieces, issued after it, may still be in flight) ...
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NWPASS) : "memory");
                lds_barrier();                                               // ... so has everybody's, and V(k) has been read
            } else {
                if (Cfg::NVB == 1) lds_barrier();                              // every wave is done reading V(k)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's channel of chunk k+1 has landed
            }
            if (a.dbg != 3) transform(k + 1);
            if constexpr (X6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weights of chunk k+1
        }
        lds_barrier();
    }

    // ---------------- epilogue: gather the 16 frequencies per (cout, tile) through LDS, A^T M A ------------
    wino_epilogue<MT>(a, acc, smem, tid, wave, khalf, l31, n, h0, w0, co0, pt);
}

// U = G g G^T per (cin, cout):  w [Cin][9][CoutPad]  ->  u [Cin][16][CoutPad]
__device__ __forceinline__ void wino_weights_elem(const float* __restrict__ w, float* __restrict__ u, int Cin, int CoutPad, long long gid) {
    if (gid >= (long long)Cin * CoutPad) return;
    const int co = (int)(gid % CoutPad);
    const int ci = (int)(gid / CoutPad);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = w[((long long)ci * 9 + i) * CoutPad + co];
    float t[4][3];                                   // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                    // (G g) G^T
        float* o = u + ((long long)ci * 16 + r * 4) * CoutPad + co;
        o[0] = t[r][0];
        o[(long long)CoutPad] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        o[2LL * CoutPad] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        o[3LL * CoutPad] = t[r][2];
    }
}
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int CoutPad) {
    wino_weights_elem(w, u, Cin, CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

void launch_wino_weights(const float* w, float* u, int Cin, int CoutPad, hipStream_t st) {
    const long long n = (long long)Cin * CoutPad;
    VR_LAUNCH(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, u, Cin, CoutPad);
    VR_HIP(hipGetLastError());
}

// The same U as three bf16 planes (conv_stage.h): u6 [ceil(Cin/8)][16][3][CoutPad][8 channels], zero for channels >= Cin.
__device__ __forceinline__ void wino_weights6_elem(const float* __restrict__ w, unsigned short* __restrict__ u6, int Cin, int CoutPad,
                                                   long long gid) {
    const int cin8 = (Cin + 7) / 8 * 8;
    if (gid >= (long long)cin8 * CoutPad) return;
    const int co = (int)(gid % CoutPad);
    const int ci = (int)(gid / CoutPad);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = ci < Cin ? w[((long long)ci * 9 + i) * CoutPad + co] : 0.f;
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
    unsigned short* o = u6 + (((long long)(ci >> 3) * 48) * CoutPad + co) * 8 + (ci & 7);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u[4];
        u[0] = t[r][0];
        u[1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        u[2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        u[3] = t[r][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int p1, p2, p3;
            split3_pair(u[c], 0.f, p1, p2, p3);
            const long long f = r * 4 + c;
            o[(f * 3 + 0) * CoutPad * 8] = (unsigned short)(p1 & 0xffff);
            o[(f * 3 + 1) * CoutPad * 8] = (unsigned short)(p2 & 0xffff);
            o[(f * 3 + 2) * CoutPad * 8] = (unsigned short)(p3 & 0xffff);
        }
    }
}
__global__ void wino_weights6_kernel(const float* __restrict__ w, unsigned short* __restrict__ u6, int Cin, int CoutPad) {
    wino_weights6_elem(w, u6, Cin, CoutPad, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// Every layer of a model in one launch (the training step refreshes ~90 of these per step): blockIdx.y = descriptor
__global__ void wino_weights_batched_kernel(const WinoWDesc* __restrict__ d, int split6) {
    const WinoWDesc e = d[blockIdx.y];
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (split6) wino_weights6_elem(e.w, static_cast<unsigned short*>(e.u), e.Cin, e.CoutPad, gid);
    else wino_weights_elem(e.w, static_cast<float*>(e.u), e.Cin, e.CoutPad, gid);
}

void launch_wino_weights_batched(const WinoWDesc* d_descs, int n, long long max_elems, bool split6, hipStream_t st) {
    if (n <= 0) return;
    VR_LAUNCH(wino_weights_batched_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)n), dim3(256), 0, st, d_descs,
                       split6 ? 1 : 0);
    VR_HIP(hipGetLastError());
}

size_t wino_weights6_bytes(int Cin, int CoutPad) { return (size_t)((Cin + 7) / 8) * 48 * CoutPad * 16; }

void launch_wino_weights6(const float* w, void* u6, int Cin, int CoutPad, hipStream_t st) {
    const long long n = (long long)((Cin + 7) / 8 * 8) * CoutPad;
    VR_LAUNCH(wino_weights6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w,
                       static_cast<unsigned short*>(u6), Cin, CoutPad);
    VR_HIP(hipGetLastError());
}

template <int MT, int MODE>
static void wino_launch(const ConvArgs& a, hipStream_t st) {
    using Cfg = WinoCfg<MT, MODE>;
    auto kern = conv_wino_kernel<MT, MODE>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    const int grid = groups * 8 * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

// True when the launch can take the Winograd kernel (3x3 stride-1 forward or data gradient, plain inputs,
// transformed weights available).
bool wino_pick(const ConvArgs& a, const ConvShape& s, int* MT_out) {
    static const int enabled = getenv("VR_CONV_WINO") ? atoi(getenv("VR_CONV_WINO")) : 1;
    if (!enabled || !a.wino || a.tapmask) return false;
    if (!(s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1)) return false;
    if (a.pad_h != 1 || a.pad_w != 1 || a.Wout < 32 || (a.Win & 3)) return false;
    for (int i = 0; i < a.nsrc; ++i) {
        const ConvSrc& c = a.src[i];
        if (c.aff0 || c.aff1 || c.post || c.up || c.zins || c.slope != 1.f || c.W != a.Win) return false;
        if (i < 3 && a.dst[i].wshift) return false;
        if ((long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 >= 0x7FFFFFF0LL) return false;
    }
    if ((long long)a.Cin * 16 * a.CoutPad * 4 >= 0x7FFFFFF0LL) return false;
    const long long tiles = (long long)a.N * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    int MT = (a.CoutPad % 64 == 0) ? 64 : 32;
    // below this many 64-cout workgroups the 32-cout variant (twice the workgroups, two per CU) fills the 256 CUs better:
    // 528 = 2.06 rounds of 256 was measured 8-10 % slower than the 32-cout form; the split-bf16 64-cout kernel keeps its edge
    static const int min64_env = getenv("VR_WINO_MIN64") ? atoi(getenv("VR_WINO_MIN64")) : 0;
    const int min64 = min64_env ? min64_env : (a.bf16 == 2 && a.wino6 ? 384 : 600);
    if (MT == 64 && tiles * (a.CoutPad / 64) < min64) MT = 32;
    // 32 couts per workgroup amortise the input transform poorly: with few input channels (padded to
    // chunks of 8, no partial-chunk shortcut here) the direct LDS-DMA kernel is the faster one (measured)
    static const int min_cin = getenv("VR_WINO_MINCIN") ? atoi(getenv("VR_WINO_MINCIN")) : 24;
    if (MT == 32 && a.Cin < min_cin) return false;
    *MT_out = MT;
    return true;
}

void wino_fill_tiling(ConvArgs& a, int MT) {
    a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_h = (a.Hout + 7) / 8;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / MT;
}

void wino_launch_conv(const ConvArgs& a, int MT, hipStream_t st) {
    // mode 2 on the 64-cout variant only: the 32-cout one would lose its second workgroup per CU to the bf16 planes (93 KB
    // of LDS) and measured 1.35x SLOWER than its fp32 self (VR_X6_MIN_MT=32 runs it)
    static const int x6_min_mt = getenv("VR_X6_MIN_MT") ? atoi(getenv("VR_X6_MIN_MT")) : 64;
    if (a.bf16 == 2 && a.wino6 && MT >= x6_min_mt) {
        if (MT == 64) wino_launch<64, 2>(a, st); else wino_launch<32, 2>(a, st);
    } else if (a.bf16 == 1) {
        if (MT == 64) wino_launch<64, 1>(a, st); else wino_launch<32, 1>(a, st);
    } else {
        if (MT == 64) wino_launch<64, 0>(a, st); else wino_launch<32, 0>(a, st);
    }
}

}  // namespace vr
