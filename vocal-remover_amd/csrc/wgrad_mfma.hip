// Weight gradient of every conv of the hot path (backward of lib/layers.py:12-20 under train.py:92):
//   dW[co][ci][kh][kw] = sum_{n,h,w} dz[n][co][h][w] * in[n][ci][h*s + kh*d - pad][w*s + kw*d - pad]
// as an fp32-MFMA GEMM with the reduction over PIXELS:  D[i=co][j=ci] += A[i][k=pixel] * B[k][j]
// (one 32x32 accumulator tile per (32-cout block, tap)).  `in` is the conv's virtual input, staged
// through the same fused loader as the forward pass (concat / upsample / BatchNorm affine /
// activation / dropout re-applied on the fly from the RAW saved tensors), `dz` is the gradient at the
// conv's raw output (already through the BatchNorm backward).
//
// Work split: block = (32 input channels, MB*32 output channels, one of P contiguous ranges of
// pixel tiles); accumulators live in registers across the whole pixel range, then one partial
// [ci][tap][co] slab per block goes to scratch and `wgrad_reduce_kernel` sums the P slabs
// (deterministic; no atomics).
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

template <int KS, int S, int DH, int DW, int TH, int TW, int MB>
struct WgCfg {
    static constexpr int KK = KS * KS;
    static constexpr int TP = TH * TW;                 // pixels per tile (= MFMA K extent per tile)
    static constexpr int TH_in = (TH - 1) * S + (KS - 1) * DH + 1;
    static constexpr int TW_in = (TW - 1) * S + (KS - 1) * DW + 1;
    // Dilated layers (round 5): with dilation above the tile height the haloed window is mostly rows no tap of this tile reads
    // (TH = 4, dilation 12: 28 rows for 4); three TH-row windows, one per kernel row, hold everything the nine taps touch --
    // 336 instead of 784 staged elements per channel at dilation 12 (288 / 480 at 8), and three workgroups per CU instead of one.
    static constexpr bool TAPROWS = KS == 3 && S == 1 && (KS - 1) * DH + TH > KS * TH;
    static constexpr int XROWS = TAPROWS ? KS * TH : TH_in;
    static constexpr int CS = (XROWS * TW_in) | 1;     // odd channel pitch: 32 lanes = 32 channels -> 32 banks
    static constexpr int DSs = TP + 1;                 // odd cout pitch
    static constexpr int NT = (KK * MB + 3) / 4;       // accumulator tiles per wave
    static constexpr int XS = 32 * CS;
    static constexpr int DS = MB * 32 * DSs;
    static constexpr int LDS_BYTES = (XS + DS) * 4;
};

template <int KS, int S, int DH, int DW, int TH, int TW, int MB>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradArgs a) {
    using Cfg = WgCfg<KS, S, DH, DW, TH, TW, MB>;
    constexpr int KK = Cfg::KK, TP = Cfg::TP, TH_in = Cfg::TH_in, TW_in = Cfg::TW_in, CS = Cfg::CS, DSs = Cfg::DSs,
                  NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ds = smem + Cfg::XS;

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int inner = a.nchunks * a.nct;
    const int p = (rr / inner) * 8 + xcd;
    if (p >= a.P) return;
    const int ib = rr % inner;
    const int ct = ib % a.nct, cb = ib / a.nct;
    const int co0 = ct * MB * 32, c0 = cb * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;

    int mb_i[NT], tap_i[NT], moff_i[NT], toff_i[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int t = wave + 4 * i;
        t = t < KK * MB ? t : KK * MB - 1;      // surplus slots recompute the last tile, never stored
        mb_i[i] = t / KK;
        tap_i[i] = t % KK;
        moff_i[i] = mb_i[i] * 32 * DSs;
        toff_i[i] = (tap_i[i] / KS) * (Cfg::TAPROWS ? TH : DH) * TW_in + (tap_i[i] % KS) * DW;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);
    for (int pt = t_begin; pt < t_end; ++pt) {
        const int n = pt / tiles_per_img;
        const int trem = pt - n * tiles_per_img;
        const int h0 = (trem / a.tiles_w) * TH, w0 = (trem % a.tiles_w) * TW;
        __syncthreads();
        // ---- dz tile: Ds[co][px] ------------------------------------------------------------------
        {
            constexpr int NEL = MB * 32 * TP;
            constexpr int NPASS = NEL / 256;
            constexpr int PB = NPASS < 8 ? NPASS : 8;
            const float* zb = a.dz + (long long)n * a.zN;
#pragma unroll 1
            for (int p0 = 0; p0 < NPASS; p0 += PB) {
                float v[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int idx = tid + (p0 + j) * 256;
                    const int co = idx / TP, px = idx % TP;
                    int cg = co0 + co; cg = cg < a.Cout ? cg : a.Cout - 1;
                    int h = h0 + px / TW, w = w0 + px % TW;
                    h = h < a.in.Hout ? h : a.in.Hout - 1;
                    w = w < a.in.Wout ? w : a.in.Wout - 1;
                    v[j] = zb[(long long)cg * a.zC + (long long)h * a.zH + w];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int idx = tid + (p0 + j) * 256;
                    const int co = idx / TP, px = idx % TP;
                    const bool ok = (co0 + co < a.Cout) && (h0 + px / TW < a.in.Hout) && (w0 + px % TW < a.in.Wout);
                    Ds[co * DSs + px] = ok ? v[j] : 0.f;
                }
            }
        }
        // ---- input tile: Xs[ci][haloed tile] -------------------------------------------------------------
        if constexpr (Cfg::TAPROWS) {
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
                stage_input_chunk<TH, TW_in, TW_in, CS, 32, 4>(a.in, Xs + kh * TH * TW_in, c0, n, h0 - a.in.pad_h + kh * DH, w0 - a.in.pad_w,
                                                               wave, lane);
        } else {
            stage_input_chunk<TH_in, TW_in, TW_in, CS, 32, 4>(a.in, Xs, c0, n, h0 * S - a.in.pad_h, w0 * S - a.in.pad_w,
                                                               wave, lane);
        }
        __syncthreads();
        // ---- MFMA over the tile's pixels ---------------------------------------------------------------------
#pragma unroll 2
        for (int kp = 0; kp < TP / 2; ++kp) {
            const int px = 2 * kp + khalf;
            const int r = px / TW, c = px % TW;
            const int xoff = l31 * CS + (r * S) * TW_in + c * S;
            const int doff = l31 * DSs + px;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float av = Ds[moff_i[i] + doff];
                const float bv = Xs[xoff + toff_i[i]];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    }
    // ---- partial slab: part[p][ci][tap][co] ----------------------------------------------------------------------
    float* pp = a.part + (long long)p * a.part_stride;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (wave + 4 * i >= KK * MB) continue;
        const int ci = c0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mb_i[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (ci < a.in.Cin && co < a.CoutPad) pp[((long long)ci * KK + tap_i[i]) * a.CoutPad + co] = acc[i][r];
        }
    }
}

// Warp-specialised variant (same split as conv_ws.hip): waves 0..3 only run the MFMA loop on the
// current (dz, input) tile pair, waves 4..4+NPW-1 stage the NEXT pixel tile into the other LDS
// buffer; one workgroup barrier per pixel tile.
template <int KS, int S, int TH, int TW, int MB>
struct WgWsCfg {
    using B = WgCfg<KS, S, 1, 1, TH, TW, MB>;
    static constexpr int BUF = B::XS + B::DS;          // floats per buffer
    static constexpr int LDS_BYTES = 2 * BUF * 4;
};

// Consumer wave of wgrad_ws_kernel.  The four consumer waves are (sel = wave>>1, HALF = wave&1):
//   HALF picks the tap group (taps 0..4 or 5..8, compile time, so every LDS read below is
//   base + immediate), sel picks the 32-cout block (MB == 2) or the half of the tile's rows
//   (MB == 1; the two row-halves are summed through LDS at the end).  One dz fragment feeds all taps
//   of a k-step and the fragments of step k+1 are read before the MFMAs of step k are issued.
template <int KS, int S, int TH, int TW, int MB, int HALF>
__device__ __forceinline__ void wg_consumer(const WgradArgs& a, float* smem, int sel, int lane, int p, int co0, int c0,
                                            int ntiles) {
    using Cfg = WgCfg<KS, S, 1, 1, TH, TW, MB>;
    using Ws = WgWsCfg<KS, S, TH, TW, MB>;
    constexpr int KK = Cfg::KK, TW_in = Cfg::TW_in, CS = Cfg::CS, DSs = Cfg::DSs;
    constexpr int T0 = HALF * 5, NTAP = HALF ? KK - 5 : 5;
    constexpr int ROWS = MB == 2 ? TH : TH / 2;
    static_assert(KS == 3 && (MB == 1 || MB == 2), "tap grouping is written for 3x3 kernels");
    const int khalf = lane >> 5, l31 = lane & 31;
    const int rbeg = MB == 2 ? 0 : sel * ROWS;
    const int doff = (MB == 2 ? sel * 32 * DSs : 0) + l31 * DSs + khalf;
    const int xoff = l31 * CS + khalf * S;

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    __syncthreads();                         // first tile staged
    for (int it = 0; it < ntiles; ++it) {
        const float* Xs = smem + (it & 1) * Ws::BUF;
        const float* Ds = Xs + Cfg::XS;
        if (!(a.in.dbg & 1)) {
            const float* dr = Ds + doff + rbeg * TW;
            const float* xr = Xs + xoff + rbeg * S * TW_in;
            float av = dr[0], bv[NTAP];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) bv[t] = xr[((T0 + t) / KS) * TW_in + (T0 + t) % KS];
#pragma unroll 1
            for (int r = 0; r < ROWS; ++r) {
                const int rn = r + 1 < ROWS ? r + 1 : r;
                const float* drn = Ds + doff + (rbeg + rn) * TW;
                const float* xrn = Xs + xoff + (rbeg + rn) * S * TW_in;
#pragma unroll
                for (int cc = 0; cc < TW / 2; ++cc) {
                    float avn, bvn[NTAP];
                    if (cc + 1 < TW / 2) {
                        avn = dr[2 * (cc + 1)];
#pragma unroll
                        for (int t = 0; t < NTAP; ++t)
                            bvn[t] = xr[2 * (cc + 1) * S + ((T0 + t) / KS) * TW_in + (T0 + t) % KS];
                    } else {
                        avn = drn[0];
#pragma unroll
                        for (int t = 0; t < NTAP; ++t) bvn[t] = xrn[((T0 + t) / KS) * TW_in + (T0 + t) % KS];
                    }
                    __builtin_amdgcn_sched_barrier(0);       // keep the reads of step k+1 ahead of the MFMAs of step k
#pragma unroll
                    for (int t = 0; t < NTAP; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    av = avn;
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) bv[t] = bvn[t];
                }
                dr = drn;
                xr = xrn;
            }
        }
        __syncthreads();                     // done with this tile; the next one is staged
    }
    if (MB == 1) {                           // sum the two row-halves (sel 1 -> LDS -> sel 0)
        float* red = smem;
        if (sel == 1) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((T0 + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (sel == 1) return;
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += red[((T0 + t) * 16 + r) * 64 + lane];
    }
    float* pp = a.part + (long long)p * a.part_stride;
    const int ci = c0 + l31;
    const int mb = MB == 2 ? sel : 0;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (ci < a.in.Cin && co < a.CoutPad) pp[((long long)ci * KK + (T0 + t)) * a.CoutPad + co] = acc[t][r];
        }
    }
}

template <int KS, int S, int TH, int TW, int MB, int NPW>
__global__ __launch_bounds__(256 + 64 * NPW) void wgrad_ws_kernel(const WgradArgs a) {
    using Cfg = WgCfg<KS, S, 1, 1, TH, TW, MB>;
    using Ws = WgWsCfg<KS, S, TH, TW, MB>;
    constexpr int KK = Cfg::KK, TP = Cfg::TP, TH_in = Cfg::TH_in, TW_in = Cfg::TW_in, CS = Cfg::CS, DSs = Cfg::DSs,
                  NT = Cfg::NT, NPT = 64 * NPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int inner = a.nchunks * a.nct;
    const int p = (rr / inner) * 8 + xcd;
    if (p >= a.P) return;
    const int ib = rr % inner;
    const int ct = ib % a.nct, cb = ib / a.nct;
    const int co0 = ct * MB * 32, c0 = cb * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    if (wave >= 4) {
        // ------------------------------ producers ------------------------------
        const int pw = wave - 4, ptid = tid - 256;
        if (a.in.dbg & 4) __builtin_amdgcn_s_setprio(3);
        if (a.dma) {
            // plain sources: both tiles arrive by dword LDS-DMA (per-lane source offset, out-of-range = 0,
            // odd LDS pitches allowed), the loader waves execute almost no VALU instructions
            constexpr int NE = TH_in * TW_in, NPX = (NE + 63) / 64, NPD = TP / 64;
            const unsigned lds0 = (unsigned)(size_t)smem;
            for (int pt = t_begin; pt < t_end; ++pt) {
                const unsigned xs_b = lds0 + (unsigned)(((pt - t_begin) & 1) * Ws::BUF * 4);
                const unsigned ds_b = xs_b + Cfg::XS * 4;
                if (a.in.dbg & 2) { __syncthreads(); continue; }
                const int n = pt / tiles_per_img;
                const int trem = pt - n * tiles_per_img;
                const int h0 = (trem / a.tiles_w) * TH, w0 = (trem % a.tiles_w) * TW;
                const int hbase = h0 * S - a.in.pad_h, wbase = w0 * S - a.in.pad_w;
                unsigned zo[NPD], xh[NPX], xw[NPX];
#pragma unroll
                for (int q = 0; q < NPD; ++q) {
                    const int px = q * 64 + lane;
                    const int h = h0 + px / TW, w = w0 + px % TW;
                    zo[q] = (h < a.in.Hout && w < a.in.Wout) ? (unsigned)(((long long)h * a.zH + w) * 4) : 0x80000000u;
                }
#pragma unroll
                for (int q = 0; q < NPX; ++q) {
                    const int e = q * 64 + lane;
                    const int hi = hbase + e / TW_in, wi = wbase + e % TW_in;
                    const bool ok = e < NE && hi >= 0 && hi < a.in.Hin && wi >= 0 && wi < a.in.Win;
                    xh[q] = ok ? (unsigned)hi : 0u;
                    xw[q] = ok ? (unsigned)(wi * 4) : 0x80000000u;
                }
#pragma unroll 1
                for (int co = pw; co < MB * 32; co += NPW) {
                    const int cg = co0 + co;
                    const i32x4 zr = make_rsrc(a.dz + (long long)n * a.zN + (long long)(cg < a.Cout ? cg : 0) * a.zC,
                                               cg < a.Cout ? 0x7FFFFFF0u : 0u);
#pragma unroll
                    for (int q = 0; q < NPD; ++q) dma4(ds_b + (unsigned)((co * DSs + q * 64) * 4), zo[q], zr);
                }
#pragma unroll 1
                for (int cl = pw; cl < 32; cl += NPW) {
                    const int ci = c0 + cl;
                    const bool live = ci < a.in.Cin;
                    const int cj = live ? ci : 0;
                    const int si = (cj >= a.in.c1) + (cj >= a.in.c2);
                    const int clc = cj - (si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2));
                    const float* sp = si == 0 ? a.in.src[0].p : (si == 1 ? a.in.src[1].p : a.in.src[2].p);
                    const long long sN = si == 0 ? a.in.src[0].sN : (si == 1 ? a.in.src[1].sN : a.in.src[2].sN);
                    const long long sC = si == 0 ? a.in.src[0].sC : (si == 1 ? a.in.src[1].sC : a.in.src[2].sC);
                    const unsigned sH4 =
                        (unsigned)(si == 0 ? a.in.src[0].sH : (si == 1 ? a.in.src[1].sH : a.in.src[2].sH)) * 4u;
                    const i32x4 xr = make_rsrc(sp + (long long)n * sN + (long long)clc * sC, live ? 0x7FFFFFF0u : 0u);
#pragma unroll
                    for (int q = 0; q < NPX; ++q) {
                        const unsigned vo = xh[q] * sH4 + xw[q];
                        if ((q + 1) * 64 <= NE) dma4(xs_b + (unsigned)((cl * CS + q * 64) * 4), vo, xr);
                        else if (q * 64 + lane < NE) dma4(xs_b + (unsigned)((cl * CS + q * 64) * 4), vo, xr);
                    }
                }
                dma_wait();
                __syncthreads();             // tile pt has landed; consumers finished tile pt-1
            }
            __syncthreads();
            if (MB == 1) __syncthreads();
            return;
        }
        for (int pt = t_begin; pt < t_end; ++pt) {
            float* Xs = smem + ((pt - t_begin) & 1) * Ws::BUF;
            float* Ds = Xs + Cfg::XS;
            if (a.in.dbg & 2) { __syncthreads(); continue; }
            const int n = pt / tiles_per_img;
            const int trem = pt - n * tiles_per_img;
            const int h0 = (trem / a.tiles_w) * TH, w0 = (trem % a.tiles_w) * TW;
            {
                constexpr int NEL = MB * 32 * TP;
                constexpr int NPASS = (NEL + NPT - 1) / NPT;
                constexpr int PB = NPASS < 8 ? NPASS : 8;
                const float* zb = a.dz + (long long)n * a.zN;
#pragma unroll 1
                for (int p0 = 0; p0 < NPASS; p0 += PB) {
                    float v[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        int idx = ptid + (p0 + j) * NPT;
                        idx = idx < NEL ? idx : NEL - 1;
                        const int co = idx / TP, px = idx % TP;
                        int cg = co0 + co; cg = cg < a.Cout ? cg : a.Cout - 1;
                        int h = h0 + px / TW, w = w0 + px % TW;
                        h = h < a.in.Hout ? h : a.in.Hout - 1;
                        w = w < a.in.Wout ? w : a.in.Wout - 1;
                        v[j] = zb[(long long)cg * a.zC + (long long)h * a.zH + w];
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        const int idx = ptid + (p0 + j) * NPT;
                        const int co = idx / TP, px = idx % TP;
                        const bool ok = (co0 + co < a.Cout) && (h0 + px / TW < a.in.Hout) && (w0 + px % TW < a.in.Wout);
                        if (idx < NEL) Ds[co * DSs + px] = ok ? v[j] : 0.f;
                    }
                }
            }
            stage_input_chunk<TH_in, TW_in, TW_in, CS, 32, NPW>(a.in, Xs, c0, n, h0 * S - a.in.pad_h,
                                                                 w0 * S - a.in.pad_w, pw, lane);
            __syncthreads();                 // tile pt staged; consumers finished tile pt-1
        }
        __syncthreads();                     // matches the consumers' first barrier of the next (absent) tile
        if (MB == 1) __syncthreads();        // k-split reduction barrier of the consumers
        return;
    }
    // -------------------------------- consumers --------------------------------
    if (wave & 1) wg_consumer<KS, S, TH, TW, MB, 1>(a, smem, wave >> 1, lane, p, co0, c0, t_end - t_begin);
    else wg_consumer<KS, S, TH, TW, MB, 0>(a, smem, wave >> 1, lane, p, co0, c0, t_end - t_begin);
}

// Sums the P partial slabs: 64 elements x 4 slab groups per workgroup (slab p goes to group p & 3, each
// thread keeps 4 loads in flight), groups added in a fixed order -> deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long long stride, int P,
                                                           float* __restrict__ out, long long n, int accumulate) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = q;
        for (; p + 12 < P; p += 16) {
            s0 += part[(long long)p * stride + i];
            s1 += part[(long long)(p + 4) * stride + i];
            s2 += part[(long long)(p + 8) * stride + i];
            s3 += part[(long long)(p + 12) * stride + i];
        }
        for (; p < P; p += 4) s0 += part[(long long)p * stride + i];
    }
    red[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < n) {
        const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        out[i] = accumulate ? out[i] + s : s;
    }
}

struct WgTile { int TH, TW, MB; bool ws; };

static bool wg_ws_enabled() {
    static const bool on = [] { const char* e = getenv("VR_WGRAD_WS"); return !e || atoi(e) != 0; }();
    return on;
}

static WgTile wg_pick(const WgradArgs& a, const ConvShape& s) {
    WgTile t;
    const bool dilated = (s.dil_h != 1 || s.dil_w != 1);
    const int nb = a.CoutPad / 32;
    t.ws = wg_ws_enabled() && s.KS == 3 && !dilated;
    if (t.ws) {
        if (s.stride == 2) { t.TW = 16; t.TH = 4; }
        else { t.TW = a.in.Wout >= 32 ? 32 : 16; t.TH = t.TW == 32 ? 4 : 8; }
        t.MB = (nb % 2 == 0) ? 2 : 1;
        return t;
    }
    t.TW = (a.in.Wout >= 32 && !dilated) ? 32 : 16;
    t.TH = dilated ? 4 : (t.TW == 32 ? 4 : 8);
    t.MB = (nb % 4 == 0) ? 4 : ((nb % 2 == 0) ? 2 : 1);
    return t;
}

bool wgrad_wino_pick(const WgradArgs& a, const ConvShape& s, int* CB_out, int* MT_out);     // wgrad_wino.hip
void wgrad_wino_plan(WgradArgs& a, int CB, int MT);
void wgrad_wino_launch(const WgradArgs& a, int CB, int MT, hipStream_t st);
bool wgrad_gemm_pick(const WgradArgs& a, const ConvShape& s);                                   // wgrad_gemm.hip
void wgrad_gemm_plan(WgradArgs& a);
void wgrad_gemm_launch(const WgradArgs& a, hipStream_t st);

void wgrad_plan(WgradArgs& a, const ConvShape& s) {
    const WgTile t = wg_pick(a, s);
    a.tiles_w = (a.in.Wout + t.TW - 1) / t.TW;
    a.tiles_h = (a.in.Hout + t.TH - 1) / t.TH;
    a.npt = a.in.N * a.tiles_h * a.tiles_w;
    a.nchunks = (a.in.Cin + 31) / 32;
    a.nct = a.CoutPad / (32 * t.MB);
    a.part_stride = (long long)a.in.Cin * s.KS * s.KS * a.CoutPad;
    const long long inner = (long long)a.nchunks * a.nct;
    const long long cap = std::max<long long>(1, (64LL << 20) / a.part_stride);       // scratch <= 256 MB
    // Round 6 (wgrad_wino_plan has the derivation): pixel range p runs on XCD p % 8 with its `inner` workgroups, an XCD has 32 CUs, so the
    // busiest XCD makes ceil(ceil(P / 8) * inner / slots) passes of npt / P tiles each.  slots = 32 x the workgroups a CU holds: one for
    // the warp-specialised kernel (2 x 55 KB of LDS), two for the dilated LDS-DMA kernel (60 - 76 KB).  The other shapes keep "512
    // workgroups" (VR_WG_PTARGET restores it everywhere).
    static const int ptarget = getenv("VR_WG_PTARGET") ? atoi(getenv("VR_WG_PTARGET")) : 0;
    const bool dilated = s.dil_h != 1 || s.dil_w != 1;
    const long long slots = t.ws ? 32 : (dilated ? 64 : 0);
    long long P;
    if (ptarget > 0 || slots == 0) {
        P = (ptarget > 0 ? ptarget : 512) / inner;
        if (P < 1) P = 1;
        if (P > a.npt) P = a.npt;
        if (P > cap) P = cap;
    } else {
        const long long pmax = std::min(std::min<long long>(a.npt, cap), std::max<long long>(8, 1024 / inner));
        const double ovh = 4.0;
        double best = 1e30;
        P = 1;
        for (long long c = 1; c <= pmax; ++c) {
            const long long rounds = ((c + 7) / 8 * inner + slots - 1) / slots;
            const double cost = (double)rounds * ((double)((a.npt + c - 1) / c) + ovh);
            if (cost < best * 0.995) { best = cost; P = c; }
        }
    }
    a.P = (int)P;
}

template <int KS, int S, int DH, int DW, int TH, int TW, int MB>
static void wg_launch_inst(const WgradArgs& a, hipStream_t st) {
    using Cfg = WgCfg<KS, S, DH, DW, TH, TW, MB>;
    auto kern = wgrad_mfma_kernel<KS, S, DH, DW, TH, TW, MB>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int KS, int S, int DH, int DW, int TH, int TW>
static void wg_launch_mb(const WgradArgs& a, int MB, hipStream_t st) {
    if (MB == 4) wg_launch_inst<KS, S, DH, DW, TH, TW, 4>(a, st);
    else if (MB == 2) wg_launch_inst<KS, S, DH, DW, TH, TW, 2>(a, st);
    else wg_launch_inst<KS, S, DH, DW, TH, TW, 1>(a, st);
}

template <int KS, int S, int TH, int TW, int MB>
static void wg_launch_ws_inst(const WgradArgs& a, hipStream_t st) {
    constexpr int NPW = 8;
    using Ws = WgWsCfg<KS, S, TH, TW, MB>;
    auto kern = wgrad_ws_kernel<KS, S, TH, TW, MB, NPW>;
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Ws::LDS_BYTES);
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    VR_LAUNCH(kern, dim3(grid), dim3(256 + 64 * NPW), Ws::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int KS, int S, int TH, int TW>
static void wg_launch_ws(const WgradArgs& a, int MB, hipStream_t st) {
    if (MB == 2) wg_launch_ws_inst<KS, S, TH, TW, 2>(a, st);
    else wg_launch_ws_inst<KS, S, TH, TW, 1>(a, st);
}

// the same sum for MANY layers in one launch: block b belongs to the descriptor d with d.blk0 <= b < next.blk0 (binary search over <= 128).
// VEC = 4: a thread sums FOUR neighbouring elements (16-byte loads, four slabs in flight per accumulator set -- the scalar form of the
// first version streamed at ~0.3 TB/s: 690 us at the end of every train step, in front of Adam); per element the order of the additions
// is wgrad_reduce_kernel's, so the sums stay bit-equal.  VEC = 1: slabs whose size or address is not a multiple of 16 bytes.
template <int VEC>
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgReduceDesc* __restrict__ descs, int nd) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ vec_t red[4][64];
    int lo = 0, hi = nd - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgReduceDesc d = descs[lo];
    const vec_t* __restrict__ part = reinterpret_cast<const vec_t*>(d.part);
    const long long stride = d.stride / VEC, n = d.n / VEC;
    const int P = d.P;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x - d.blk0) * 64 + e;
    vec_t s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = q;
        for (; p + 12 < P; p += 16) {
            const vec_t v0 = part[(long long)p * stride + i], v1 = part[(long long)(p + 4) * stride + i],
                        v2 = part[(long long)(p + 8) * stride + i], v3 = part[(long long)(p + 12) * stride + i];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; p < P; p += 4) s0 += part[(long long)p * stride + i];
    }
    red[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < n) {
        const vec_t s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);        // (the order of wgrad_reduce_kernel: bit-equal results)
        vec_t* o = reinterpret_cast<vec_t*>(d.out) + i;
        *o = d.accumulate ? *o + s : s;
    }
}
// blocks a descriptor takes in the batched launch (the caller lays the descriptors out back to back: WgReduceDesc::blk0)
int wgrad_reduce_vec(const WgReduceDesc* h, int n) {
    for (int j = 0; j < n; ++j)
        if ((h[j].n & 3) || (h[j].stride & 3) || (reinterpret_cast<size_t>(h[j].part) & 15) || (reinterpret_cast<size_t>(h[j].out) & 15)) return 1;
    return 4;
}
void launch_wgrad_reduce_batched(const WgReduceDesc* d_descs, int n, long long total_blocks, int vec, hipStream_t st) {
    if (n <= 0) return;
    if (vec == 4) VR_LAUNCH(wgrad_reduce_batched_kernel<4>, dim3((unsigned)total_blocks), dim3(256), 0, st, d_descs, n);
    else VR_LAUNCH(wgrad_reduce_batched_kernel<1>, dim3((unsigned)total_blocks), dim3(256), 0, st, d_descs, n);
    VR_HIP(hipGetLastError());
}

static thread_local std::vector<WgReduceDesc>* g_wgrad_sink = nullptr;
void wgrad_defer_to(std::vector<WgReduceDesc>* sink) { g_wgrad_sink = sink; }

static void wgrad_reduce(const WgradArgs& a, float* grad_out, int accumulate, hipStream_t st) {
    const long long n = a.part_stride;
    if (g_wgrad_sink) {
        // a later layer of the same step may add into the same gradient (shared weights do not exist in this net, but the debug hooks
        // accumulate on purpose): two deferred sums into one `out` would race inside the batched launch -- the second one runs now
        for (const WgReduceDesc& d : *g_wgrad_sink)
            if (d.out == grad_out) goto immediate;
        g_wgrad_sink->push_back(WgReduceDesc{a.part, a.part_stride, grad_out, n, a.P, accumulate, 0});
        return;
    }
immediate:
    VR_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, a.part, a.part_stride,
                       a.P, grad_out, n, accumulate);
    VR_HIP(hipGetLastError());
}

double launch_wgrad(const WgradArgs& a_in, const ConvShape& s, float* grad_out, int accumulate, hipStream_t st) {
    WgradArgs a = a_in;
    VR_CHECK(a.part != nullptr, -2, "wgrad needs a scratch slab");
    {
        int CB = 0, MT = 0;
        if (wgrad_wino_pick(a, s, &CB, &MT)) {             // Winograd F(3x3,2x2): 2.25x fewer MFMAs
            wgrad_wino_plan(a, CB, MT);
            wgrad_wino_launch(a, CB, MT, st);
            wgrad_reduce(a, grad_out, accumulate, st);
            return 2.0 * a.in.N * (double)a.in.Hout * a.in.Wout * (double)a.Cout * a.in.Cin * 9;
        }
    }
    if (wgrad_gemm_pick(a, s)) {                           // 1x1: pixel-contiguous GEMM, LDS-DMA ring
        wgrad_gemm_plan(a);
        wgrad_gemm_launch(a, st);
        wgrad_reduce(a, grad_out, accumulate, st);
        return 2.0 * a.in.N * (double)a.in.Hout * a.in.Wout * (double)a.Cout * a.in.Cin;
    }
    wgrad_plan(a, s);
    const WgTile t = wg_pick(a, s);
    {
        static const bool dma_on = !getenv("VR_NO_WGRAD_DMA");
        bool plain = dma_on;
        for (int i = 0; i < a.in.nsrc && plain; ++i) {
            const ConvSrc& c = a.in.src[i];
            plain = !c.aff0 && !c.aff1 && !c.post && !c.up && !c.zins && c.slope == 1.f &&
                    (long long)c.H * (c.sH > 0 ? c.sH : 1) * 4 < 0x7FFFFFF0LL;
        }
        plain = plain && (long long)a.in.Hout * a.zH * 4 < 0x7FFFFFF0LL;
        a.dma = plain ? 1 : 0;
    }
    {
        static const int dbg = [] { const char* e = getenv("VR_WG_DBG"); return e ? atoi(e) : 0; }();
        a.in.dbg = dbg;
    }
    if (t.ws) {
        if (s.stride == 2) wg_launch_ws<3, 2, 4, 16>(a, t.MB, st);
        else if (t.TW == 32) wg_launch_ws<3, 1, 4, 32>(a, t.MB, st);
        else wg_launch_ws<3, 1, 8, 16>(a, t.MB, st);
    } else
    if (s.KS == 1) {
        if (t.TW == 32) wg_launch_mb<1, 1, 1, 1, 4, 32>(a, t.MB, st); else wg_launch_mb<1, 1, 1, 1, 8, 16>(a, t.MB, st);
    } else if (s.stride == 1 && s.dil_h == 1 && s.dil_w == 1) {
        if (t.TW == 32) wg_launch_mb<3, 1, 1, 1, 4, 32>(a, t.MB, st); else wg_launch_mb<3, 1, 1, 1, 8, 16>(a, t.MB, st);
    } else if (s.stride == 2) {
        if (t.TW == 32) wg_launch_mb<3, 2, 1, 1, 4, 32>(a, t.MB, st); else wg_launch_mb<3, 2, 1, 1, 8, 16>(a, t.MB, st);
    } else if (s.dil_h == 4 && s.dil_w == 2) {
        wg_launch_mb<3, 1, 4, 2, 4, 16>(a, t.MB, st);
    } else if (s.dil_h == 8 && s.dil_w == 4) {
        wg_launch_mb<3, 1, 8, 4, 4, 16>(a, t.MB, st);
    } else if (s.dil_h == 12 && s.dil_w == 6) {
        wg_launch_mb<3, 1, 12, 6, 4, 16>(a, t.MB, st);
    } else {
        throw Error(-2, "unsupported wgrad shape");
    }
    wgrad_reduce(a, grad_out, accumulate, st);
    return 2.0 * a.in.N * (double)a.in.Hout * a.in.Wout * (double)a.Cout * a.in.Cin * s.KS * s.KS;
}

// Scratch for the P partial slabs.  The kernel choice is re-made at launch time from the real pointers (alignment),
// so the slab is sized for whichever of the two plans needs more.
size_t wgrad_scratch_floats(const WgradArgs& a_in, const ConvShape& s) {
    WgradArgs a = a_in;
    wgrad_plan(a, s);
    size_t need = (size_t)a.P * (size_t)a.part_stride;
    if (s.KS == 1 && s.stride == 1 && (long long)a_in.in.Hout * a_in.in.Wout % 64 == 0) {
        WgradArgs b = a_in;
        wgrad_gemm_plan(b);
        const size_t n = (size_t)b.P * (size_t)b.part_stride;
        if (n > need) need = n;
    }
    if (s.KS == 3 && s.stride == 1 && s.dil_h == 1 && s.dil_w == 1 && a_in.allow_wino) {
        for (int alt = 0; alt < 4; ++alt) {
            WgradArgs b = a_in;
            if (b.CoutPad % ((alt == 0 || alt == 3) ? 64 : 32)) continue;
            wgrad_wino_plan(b, (alt == 1 || alt == 3) ? 64 : 32, (alt == 0 || alt == 3) ? 64 : 32);
            const size_t n = (size_t)b.P * (size_t)b.part_stride;
            if (n > need) need = n;
        }
    }
    return need;
}

}  // namespace vr
