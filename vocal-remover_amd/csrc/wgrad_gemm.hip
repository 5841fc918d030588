// Weight gradient of the 1x1 convolutions (ASPP conv2 / bottleneck, the stage tail convs; backward of
// lib/layers.py:12-20,76-89 under train.py:92):   dW[co][ci] = sum_{n,pixel} dz[n][co][pixel] * x[n][ci][pixel]
// -- a plain GEMM with the reduction over pixels, both operands pixel-contiguous.  Rows of 64 pixels arrive by 16-byte
// LDS-DMA into a ring of 3 stages (no arithmetic in the loader), 8 waves multiply a 64-cout x 128-cin block on
// v_mfma_f32_32x32x2_f32.  The k order inside a 64-pixel chunk is free as long as both operands use the same one:
// lane (row, khalf) reads pixels 8q + 4*khalf .. +3 of its row with ONE 16-byte LDS read and feeds them to four
// successive MFMAs, i.e. one LDS read per operand per four MFMAs (row pitch 68 floats = 17 x 16 B: conflict-free).
// One partial slab [ci][CoutPad] per block, summed by wgrad_reduce_kernel (wgrad_mfma.hip) -- deterministic.
#include <cstdlib>

#include "conv_stage.h"
#include "lds_dma.h"

namespace vr {

struct WgGemmCfg {
    static constexpr int MT = 64, CB = 128, KC = 64, RP = KC + 4;      // couts, input channels, pixels per chunk; 4*RP = pitch of a 4-row group
    static constexpr int D = 3;                                        // stages in flight
    static constexpr int STAGE = (MT + CB) * RP;
    static constexpr int LDS_FLOATS = D * STAGE;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NI = (MT + CB) / 4 / 8;                       // DMA instructions per wave and chunk (4 rows each)
    static_assert(LDS_BYTES <= 160 * 1024 && NI * (D - 1) <= 63, "LDS / vmcnt immediate");
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// KS (round 4): layers with at most 32 couts and 32 input channels (the stage tail convs 32 -> 16 / 16 -> 8 at full resolution: 2.1 M
// pixels per channel) have ONE real 32 x 32 output tile -- in the blocked form seven of the eight waves multiplied zero padding and the
// launch was bound by MFMAs on zeros (346 us for 400 MB).  Here the eight waves split the 64 pixels of a chunk instead (wave w takes
// k-step w) and their accumulators are added in a fixed order through LDS at the end: HBM-bound, as the shape is.
template <bool BF, bool KS = false>
__global__ __launch_bounds__(512, 2) void wgrad_gemm_kernel(const WgradArgs a, int chunks_per_img, long long img_pixels) {
    using Cfg = WgGemmCfg;
    constexpr int MT = Cfg::MT, CB = Cfg::CB, KC = Cfg::KC, RP = Cfg::RP, D = Cfg::D, NI = Cfg::NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];

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
    const int t_begin = (int)((long long)p * a.npt / a.P), t_end = (int)((long long)(p + 1) * a.npt / a.P);

    // ---- DMA: a stage = rows [0, MT) dz, [MT, MT+CB) x.  One instruction = 4 rows x 16 pieces of 16 B, PIECE-major:
    // lane L fetches (row L & 3, piece L >> 2) and lands at 16*L bytes, so that the 32 rows an MFMA operand read touches
    // fall into 8 different bank groups (group pitch 68 quads).  Wave w issues instructions w, w+8, ... (NI of them, always
    // exactly one DMA each: the launcher guarantees that a 4-row group never straddles two tensors of the virtual concat).
    const int q_row = lane & 3, q_piece = lane >> 2;
    auto issue_chunk = [&](int pt) {
        const bool real = pt < t_end;
        const int ptc = real ? pt : t_begin;
        const unsigned stage = lds0 + (unsigned)(((pt - t_begin) % D) * Cfg::STAGE * 4);
        const int n = ptc / chunks_per_img;
        const long long px0 = (long long)(ptc - n * chunks_per_img) * KC + 4 * q_piece;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int inst = wave + 8 * i;                             // rows 4*inst .. 4*inst+3 of the stage (wave-uniform)
            const unsigned dst = stage + (unsigned)(inst * (4 * RP) * 4);
            if (4 * inst < MT) {
                const int cg = co0 + 4 * inst + q_row;
                const bool ok = real && cg < a.Cout;
                const i32x4 zr = make_rsrc(a.dz + (long long)n * a.zN, 0x7FFFFFF0u);
                const unsigned vo = ok ? (unsigned)(((long long)cg * a.zC + px0) * 4) : 0x80000000u;
                dma16(dst, vo, zr);
            } else {
                const int cig = c0 + 4 * inst - MT;                    // first channel of the group
                const bool live = real && cig < a.in.Cin;
                const int cj = live ? cig : 0;
                const int si = (cj >= a.in.c1) + (cj >= a.in.c2);
                const int cbase = si == 0 ? 0 : (si == 1 ? a.in.c1 : a.in.c2);
                const float* sp = VR_SEL_F(a.in, si, p);
                const long long sN = VR_SEL_F(a.in, si, sN), sC = VR_SEL_F(a.in, si, sC);
                const i32x4 xr = make_rsrc(sp + (long long)n * sN, live ? 0x7FFFFFF0u : 0u);
                const unsigned vo = live ? (unsigned)(((long long)(cj - cbase + q_row) * sC + px0) * 4) : 0x80000000u;
                dma16(dst, vo, xr);
            }
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    const int mi = KS ? 0 : (wave & 1), nb = KS ? 0 : (wave >> 1);     // this wave: couts [32*mi, +32) x input channels [32*nb, +32)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // Row r of a stage: group r>>2 at (r>>2) * 4*RP floats, inside the group piece-major: (piece*4 + (r&3)) * 4 floats.
    // The k order inside a chunk is free: at step q lane (row, khalf) reads piece 2q + khalf = pixels 8q+4*khalf .. +3.
    auto row_off = [](int r) { return (r >> 2) * (4 * RP) + (r & 3) * 4; };
    const int a_off = row_off(32 * mi + l31) + 16 * khalf;
    const int b_off = row_off(MT + 32 * nb + l31) + 16 * khalf;

    if (t_begin < t_end) {
#pragma unroll
        for (int i = 0; i < D; ++i) issue_chunk(t_begin + i);
    }
    for (int pt = t_begin; pt < t_end; ++pt) {
        // my part of chunk pt has landed (D-1 younger chunks may be in flight); barrier: everyone's has
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (D - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* st = smem + ((pt - t_begin) % D) * Cfg::STAGE;
#pragma unroll
        for (int q = 0; q < KC / 8; ++q) {
            if (KS && q != wave) continue;                               // (KC / 8 == 8 k-steps == 8 waves)
            const f32x4v av = *reinterpret_cast<const f32x4v*>(st + a_off + 32 * q);
            const f32x4v bv = *reinterpret_cast<const f32x4v*>(st + b_off + 32 * q);
            if constexpr (BF) {
                acc = mfma_bf16(pack_bf16x4(av[0], av[1], av[2], av[3]), pack_bf16x4(bv[0], bv[1], bv[2], bv[3]), acc);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave is done reading this stage
        asm volatile("" ::: "memory");
        issue_chunk(pt + D);                                            // refill it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- partial slab part[p][ci][CoutPad] (KS*KS = 1) -------------------------------------------------------------------
    float* pp = a.part + (long long)p * a.part_stride;
    const int ci = c0 + 32 * nb + l31;
    if constexpr (KS) {
        // the eight waves hold partial sums of the SAME tile: through LDS, added in wave order
        __builtin_amdgcn_s_barrier();                                   // (every wave is past its last stage read)
        float* red = smem;                                              // [8 waves][16 regs][64 lanes]
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave < 2) {                                                 // wave 0: registers 0..7, wave 1: registers 8..15
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = wave * 8 + rr;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += red[(w * 16 + r) * 64 + lane];
                const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (ci < a.in.Cin && co < a.CoutPad) pp[(long long)ci * a.CoutPad + co] = s;
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (ci < a.in.Cin && co < a.CoutPad) pp[(long long)ci * a.CoutPad + co] = acc[r];
    }
}

static bool wg_gemm_enabled() {
    static const bool on = [] { const char* e = getenv("VR_WGRAD_GEMM"); return !e || atoi(e) != 0; }();
    return on;
}

// 1x1 stride-1 conv whose sources and dz are pixel-contiguous per (sample, channel) with 16-byte aligned rows.
bool wgrad_gemm_pick(const WgradArgs& a, const ConvShape& s) {
    if (!wg_gemm_enabled() || s.KS != 1 || s.stride != 1) return false;
    const long long L = (long long)a.in.Hout * a.in.Wout;
    if (a.in.Hin != a.in.Hout || a.in.Win != a.in.Wout || L % 64 != 0) return false;
    for (int i = 0; i < a.in.nsrc; ++i) {
        const ConvSrc& c = a.in.src[i];
        if (c.aff0 || c.aff1 || c.post || c.up || c.zins || c.slope != 1.f) return false;
        if (c.W != a.in.Win || c.H != a.in.Hin || c.sH != c.W) return false;                   // rows back to back
        if ((c.sC & 3) || (c.sN & 3) || (reinterpret_cast<uintptr_t>(c.p) & 15)) return false;
        if ((long long)c.C * c.sC * 4 >= 0x7FFFFFF0LL) return false;
    }
    if ((a.in.Cin & 3) || (a.in.nsrc >= 2 && (a.in.c1 & 3)) || (a.in.nsrc >= 3 && (a.in.c2 & 3))) return false;   // 4-row DMA groups
    if (a.zH != a.in.Wout || (a.zC & 3) || (a.zN & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15)) return false;
    if ((long long)a.Cout * a.zC * 4 >= 0x7FFFFFF0LL) return false;
    return true;
}

void wgrad_gemm_plan(WgradArgs& a) {
    const long long L = (long long)a.in.Hout * a.in.Wout;
    a.tiles_w = (int)(L / 64); a.tiles_h = 1;
    a.npt = a.in.N * a.tiles_w;
    a.nchunks = (a.in.Cin + WgGemmCfg::CB - 1) / WgGemmCfg::CB;
    a.nct = (a.CoutPad + WgGemmCfg::MT - 1) / WgGemmCfg::MT;
    a.part_stride = (long long)a.in.Cin * a.CoutPad;
    long long P = 256 / ((long long)a.nchunks * a.nct);
    if (P < 1) P = 1;
    if (P > a.npt) P = a.npt;
    const long long cap = (64LL << 20) / a.part_stride;
    if (P > cap) P = cap < 1 ? 1 : cap;
    a.P = (int)P;
}

void wgrad_gemm_launch(const WgradArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0}, attr_done_bf{0};          // per device (bit = device index)
    const int grid = ((a.P + 7) / 8) * 8 * a.nchunks * a.nct;
    if (a.bf16 != 1 && a.CoutPad <= 32 && a.in.Cin <= 32) {                       // one real 32 x 32 tile: split the pixels over the waves
        static std::atomic<unsigned long long> attr_done_ks{0};
        ensure_lds_attr(attr_done_ks, reinterpret_cast<const void*>(wgrad_gemm_kernel<false, true>), WgGemmCfg::LDS_BYTES);
        VR_LAUNCH((wgrad_gemm_kernel<false, true>), dim3(grid), dim3(512), WgGemmCfg::LDS_BYTES, st, a, a.tiles_w,
                  (long long)a.in.Hout * a.in.Wout);
        VR_HIP(hipGetLastError());
        return;
    }
    if (a.bf16 == 1) {
        ensure_lds_attr(attr_done_bf, reinterpret_cast<const void*>(wgrad_gemm_kernel<true>), WgGemmCfg::LDS_BYTES);
        VR_LAUNCH(wgrad_gemm_kernel<true>, dim3(grid), dim3(512), WgGemmCfg::LDS_BYTES, st, a, a.tiles_w,
                           (long long)a.in.Hout * a.in.Wout);
    } else {
        ensure_lds_attr(attr_done, reinterpret_cast<const void*>(wgrad_gemm_kernel<false>), WgGemmCfg::LDS_BYTES);
        VR_LAUNCH(wgrad_gemm_kernel<false>, dim3(grid), dim3(512), WgGemmCfg::LDS_BYTES, st, a, a.tiles_w,
                           (long long)a.in.Hout * a.in.Wout);
    }
    VR_HIP(hipGetLastError());
}

}  // namespace vr
