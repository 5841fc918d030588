// Cross-kernel corruption probe (round 3, the lstm weight_hh race): a long-running "canary" kernel holds known patterns in its
// vector registers and in its LDS and re-checks them while a conv kernel runs beside it on a second stream.  Any hit is logged with
// (block, thread, register index, value seen).  Aggressors: conv_x3 (three tilings, with the dbg ablations), conv_wino, conv_dma.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vocal-remover_amd/csrc tools/vgpr_canary.hip -o /tmp/vgpr_canary && /tmp/vgpr_canary
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../vocal-remover_amd/csrc/conv_x3.hip"
#include "../vocal-remover_amd/csrc/conv_wino.hip"
#include "../vocal-remover_amd/csrc/conv_dma.hip"
#include "../vocal-remover_amd/csrc/lstm.hip"

using namespace vr;

constexpr int R = 48;            // canary registers per thread
constexpr int LDSW = 5120;       // canary LDS words (20 KB, the weight_hh kernel's footprint)

__global__ __launch_bounds__(256) void canary_kernel(unsigned* __restrict__ log, unsigned* __restrict__ count, int iters) {
    __shared__ unsigned lds[LDSW];
    unsigned r[R];
    const unsigned tag = ((blockIdx.x & 0xfffu) << 8) | threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) r[i] = 0xC0000000u | ((unsigned)i << 20) | tag;
    for (int i = threadIdx.x; i < LDSW; i += 256) lds[i] = 0xD0000000u | (unsigned)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const unsigned want = 0xC0000000u | ((unsigned)i << 20) | tag;
            if (r[i] != want) {
                const unsigned s = atomicAdd(count, 1u);
                if (s < 4096u) { log[4 * s] = tag; log[4 * s + 1] = (unsigned)i; log[4 * s + 2] = r[i]; log[4 * s + 3] = (unsigned)it; }
                r[i] = want;
            }
        }
        for (int i = threadIdx.x; i < LDSW; i += 256) {
            const unsigned v = lds[i];
            if (v != (0xD0000000u | (unsigned)i)) {
                const unsigned s = atomicAdd(count, 1u);
                if (s < 4096u) { log[4 * s] = tag; log[4 * s + 1] = 0x10000u | (unsigned)i; log[4 * s + 2] = v; log[4 * s + 3] = (unsigned)it; }
                lds[i] = 0xD0000000u | (unsigned)i;
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// Variants of lstm.hip's weight_hh gradient kernel (same arithmetic, same order of additions) for locating the hazard.
//   VAR 0: the production loop.  1: all LDS operands of a step are waited for (lgkmcnt(0)) before the first FMA.
//   2: as 1 plus s_nop 7.  3: scalar LDS reads, one value at a time, volatile.
template <int VAR>
__global__ __launch_bounds__(256) void whh_variant_kernel(const float* __restrict__ dgx, const float* __restrict__ hout, float* dwf, float* dwr,
                                                          int N, int T, int H) {
    __shared__ float hs[64][65];
    __shared__ float dgs[16][64];
    const int g0 = blockIdx.x * 16, dir = blockIdx.y, k0 = blockIdx.z * 64;
    const int G = 4 * H;
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    float* dw = dir ? dwr : dwf;
    const int sh = dir ? 1 : -1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane, th = t + sh;
            const bool ok = t < T && th >= 0 && th < T;
            for (int r = wq; r < 64; r += 4) {
                const int k = k0 + r;
                hs[r][lane] = (ok && k < H) ? hout[((long long)n * 2 * H + (long long)dir * H + k) * T + th] : 0.f;
            }
            for (int r = wq; r < 16; r += 4) {
                const int g = g0 + r;
                dgs[r][lane] = (ok && g < G) ? dgx[((long long)n * 2 * G + (long long)dir * G + g) * T + t] : 0.f;
            }
            __syncthreads();
            if (VAR == 3) {
                for (int tt = 0; tt < 64; ++tt) {
                    const float hv = *(volatile float*)&hs[lane][tt];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(*(volatile float*)&dgs[wq * 4 + q][tt], hv, acc[q]);
                }
            } else {
#pragma unroll 1
                for (int tb = 0; tb < 64; tb += 8) {
                    float hv[8], dg[4][8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = hs[lane][tb + j];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) dg[q][j] = dgs[wq * 4 + q][tb + j];
                    if (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (VAR == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                    if (VAR >= 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(hv[j]));
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(dg[q][j]));
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = fmaf(dg[q][j], hv[j], acc[q]);
                }
            }
            __syncthreads();
        }
    }
    const int k = k0 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = g0 + wq * 4 + q;
        if (g < G && k < H) dw[g * H + k] = acc[q];
    }
}

// Synthetic aggressors (hardware fact 5: which KIND of neighbour disturbs the LDS -> VALU chain of the victim?), each ~64 registers so
// that its waves share SIMDs with the victim's: kind 3 streams global memory into registers (float4 loads, 8 in flight), kind 4
// only moves global memory into LDS by DMA, kind 5 only runs VALU FMAs, kind 6 streams LDS reads.
__global__ __launch_bounds__(256) void agg_vmem_kernel(const float4* __restrict__ src, size_t n4, float* __restrict__ sink, int iters) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(i + (size_t)k * 65536) % n4];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
        i += 8 * 65536 + 17;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}
__global__ __launch_bounds__(256) void agg_dma_kernel(const float* __restrict__ src, size_t nbytes, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const unsigned lds0 = (unsigned)(size_t)lds_dyn;
    const i32x4 r = make_rsrc(src, (unsigned)(nbytes > 0x7FFFFFF0ull ? 0x7FFFFFF0ull : nbytes));
    unsigned off = (blockIdx.x * 256 + threadIdx.x) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma16(lds0 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 4096 + k * 1024, off + k * 4194304u, r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off += 16 * 65536;
    }
}
__global__ __launch_bounds__(256) void agg_valu_kernel(float* __restrict__ sink, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    for (int it = 0; it < iters * 64; ++it) { a0 = fmaf(a0, 1.0001f, a1); a1 = fmaf(a1, 0.9999f, a2); a2 = fmaf(a2, 1.0002f, a3); a3 = fmaf(a3, 0.9998f, a0); }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;
}
__global__ __launch_bounds__(256) void agg_lds_kernel(float* __restrict__ sink, int iters) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = (float)i;
    __syncthreads();
    float s = 0.f;
    for (int it = 0; it < iters * 8; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(&buf[((threadIdx.x * 4 + it * 1024) & 8188)]);
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) sink[0] = s;
}

// kind 7 / 8: nothing but matrix instructions (bf16 32x32x16 / fp32 32x32x2), two independent accumulators per wave
__global__ __launch_bounds__(256) void agg_mfma_bf16_kernel(float* __restrict__ sink, int iters) {
    vr_bf16x8 av, bv;
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = (short)(0x3f80 + threadIdx.x + i); bv[i] = (short)(0x3f00 + i); }
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    for (int it = 0; it < iters * 32; ++it) {
        c0 = mfma_bf16x16(av, bv, c0);
        c1 = mfma_bf16x16(bv, av, c1);
    }
    if (c0[0] + c1[3] == 12345.678f) sink[0] = c0[0];
}
__global__ __launch_bounds__(256) void agg_mfma_f32_kernel(float* __restrict__ sink, int iters) {
    const float av = 1.f + threadIdx.x * 1e-3f, bv = 0.5f;
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    for (int it = 0; it < iters * 16; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 12345.678f) sink[0] = c0[0];
}

static float* dalloc(size_t n) { float* p; VR_HIP(hipMalloc(&p, (n ? n : 1) * 4)); return p; }

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 300;
    // the aggressor layer: a batch-16 stage-2 decoder data gradient (64 -> 64 channels at 128 x 256), pixels ~1e-3 so that a
    // stray pixel is recognisable next to the canary patterns
    const int N = 16, Cin = 64, Cout = 64, H = 128, W = 256, CoutPad = 64;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hw((size_t)Cin * 9 * CoutPad), hx((size_t)N * Cin * H * W);
    for (auto& v : hw) v = nd(rng) * 0.04f;
    for (auto& v : hx) v = nd(rng) * 1e-3f;
    float* dw = dalloc(hw.size());
    float* dx = dalloc(hx.size());
    float* dout = dalloc((size_t)N * Cout * H * W);
    VR_HIP(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.nsrc = 1;
    ConvSrc c{};
    c.p = dx; c.sH = W; c.sC = (long long)H * W; c.sN = c.sC * Cin; c.C = Cin; c.H = H; c.W = W; c.hsplit = 1 << 30; c.slope = 1.f;
    a.src[0] = c;
    a.c1 = a.c2 = Cin; a.Cin = Cin; a.w = dw; a.Cout = Cout; a.CoutPad = CoutPad;
    a.dst[0] = ConvDst{dout, (long long)Cout * H * W, (long long)H * W, (long long)W, 0, 0};
    a.d1 = a.d2 = 1 << 30;
    a.N = N; a.Hout = H; a.Wout = W; a.Hin = H; a.Win = W; a.pad_h = 1; a.pad_w = 1;
    const ConvShape shp{3, 1, 1, 1};
    void* dx3; VR_HIP(hipMalloc(&dx3, x3_weights_bytes(Cin, 9, CoutPad)));
    launch_x3_weights(dw, dx3, Cin, 9, CoutPad, 0);
    float* dwino = dalloc((size_t)Cin * 16 * CoutPad);
    launch_wino_weights(dw, dwino, Cin, CoutPad, 0);
    VR_HIP(hipDeviceSynchronize());

    unsigned *dlog, *dcount;
    VR_HIP(hipMalloc(&dlog, 4096 * 16)); VR_HIP(hipMalloc(&dcount, 4));
    hipStream_t sa, sb;
    VR_HIP(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    VR_HIP(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

    // the victim itself: the LSTM recurrent-weight gradient of stage 2 (batch 16, 256 frames, 64 hidden units) on fixed inputs
    const int LN = 16, LT = 256, LH = 64, LG = 4 * LH;
    std::vector<float> hdg((size_t)LN * 2 * LG * LT), hh((size_t)LN * 2 * LH * LT);
    for (auto& v : hdg) v = nd(rng) * 1e-6f;
    for (auto& v : hh) v = std::tanh(nd(rng));
    float* ddg = dalloc(hdg.size());
    float* dh = dalloc(hh.size());
    float* ddw = dalloc((size_t)2 * LG * LH);
    VR_HIP(hipMemcpy(ddg, hdg.data(), hdg.size() * 4, hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> wref((size_t)2 * LG * LH), wcur(wref.size()), wprod(wref.size());
    float* dpart = dalloc(lstm_whh_grad_scratch_floats(LN, LH));
    hipLaunchKernelGGL(whh_variant_kernel<0>, dim3((4 * LH + 15) / 16, 2, (LH + 63) / 64), dim3(256), 0, 0, ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH);
    VR_HIP(hipMemcpy(wref.data(), ddw, wref.size() * 4, hipMemcpyDeviceToHost));

    struct Agg { const char* name; int kind, MT, TH, dbg; };
    const Agg aggs[] = {
        {"none (canary alone)", -1, 0, 0, 0},
        {"conv_x3<64,8>", 0, 64, 8, 0},  {"conv_x3<32,16>", 0, 32, 16, 0}, {"conv_x3<32,8>", 0, 32, 8, 0},
        {"conv_x3<64,8> dbg1 (no pixel loads)", 0, 64, 8, 1}, {"conv_x3<64,8> dbg3 (no split pass)", 0, 64, 8, 3},
        {"conv_x3<64,8> dbg4 (no epilogue)", 0, 64, 8, 4},
        {"conv_wino", 1, 0, 0, 0}, {"conv_dma", 2, 0, 0, 0},
        {"synthetic: global -> register stream", 3, 0, 0, 0}, {"synthetic: global -> LDS DMA only", 4, 0, 0, 0},
        {"synthetic: VALU only", 5, 0, 0, 0}, {"synthetic: LDS reads only", 6, 0, 0, 0},
        {"synthetic: bf16 MFMA only", 7, 0, 0, 0}, {"synthetic: fp32 MFMA only", 8, 0, 0, 0},
    };
    float* dsink = dalloc(16);
    auto launch_synth = [&](int kind, hipStream_t st) {
        const size_t n4 = hx.size() / 4;
        if (kind == 3) hipLaunchKernelGGL(agg_vmem_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const float4*>(dx), n4, dsink, 24);
        if (kind == 4) hipLaunchKernelGGL(agg_dma_kernel, dim3(2048), dim3(256), 16384, st, dx, hx.size() * 4, 24);
        if (kind == 5) hipLaunchKernelGGL(agg_valu_kernel, dim3(2048), dim3(256), 0, st, dsink, 24);
        if (kind == 6) hipLaunchKernelGGL(agg_lds_kernel, dim3(2048), dim3(256), 0, st, dsink, 24);
        if (kind == 7) hipLaunchKernelGGL(agg_mfma_bf16_kernel, dim3(2048), dim3(256), 0, st, dsink, 24);
        if (kind == 8) hipLaunchKernelGGL(agg_mfma_f32_kernel, dim3(2048), dim3(256), 0, st, dsink, 24);
    };
    for (const Agg& g : aggs) {
        VR_HIP(hipMemset(dcount, 0, 4));
        ConvArgs b = a;
        X3Tile t{g.MT, g.TH};
        int wmt = 0;
        DmaTile dt{};
        bool ok = true;
        if (g.kind == 0) { b.x3w = dx3; b.bf16 = 2; b.dbg = g.dbg; x3_fill_tiling(b, t); }
        if (g.kind == 1) { b.wino = dwino; ok = wino_pick(b, shp, &wmt); if (ok) wino_fill_tiling(b, wmt); }
        if (g.kind == 2) { ok = dma_pick(b, shp, &dt); if (ok) dma_fill_tiling(b, dt); }
        if (!ok) { printf("%-40s not applicable\n", g.name); continue; }
        hipEvent_t e0, e1;
        VR_HIP(hipEventCreate(&e0)); VR_HIP(hipEventCreate(&e1));
        VR_HIP(hipEventRecord(e0, sa));
        for (int l = 0; l < launches; ++l) {
            if (g.kind == 0) x3_launch_conv(b, t, sa);
            if (g.kind == 1) wino_launch_conv(b, wmt, sa);
            if (g.kind == 2) dma_launch_conv(b, shp, dt, sa);
            if (g.kind >= 3) launch_synth(g.kind, sa);
            // a fresh wave of canaries every few conv launches: 32 blocks as the weight_hh kernel, and a chip-filling one
            if (l % 2 == 0) hipLaunchKernelGGL(canary_kernel, dim3(l % 4 == 0 ? 32 : 512), dim3(256), 0, sb, dlog, dcount, 300);
        }
        VR_HIP(hipEventRecord(e1, sa));
        VR_HIP(hipDeviceSynchronize());
        // second pass: the weight_hh gradient itself (production kernel = variant -1, then the variants) beside the same aggressor
        for (int variant = -1; variant <= 3; ++variant) {
        int wbad = 0, wruns = 0;
        if (g.kind < 0 && variant == -1) {                    // the production kernel sums in another order than the variants: own reference
            launch_lstm_whh_grad(ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH, 0, dpart, 0);
            VR_HIP(hipDeviceSynchronize());
            VR_HIP(hipMemcpy(wprod.data(), ddw, wprod.size() * 4, hipMemcpyDeviceToHost));
        }
        const std::vector<float>& wref_v = variant < 0 ? wprod : wref;
        int rowq[4] = {}, colq[4] = {};
        for (int l = 0; l < launches; ++l) {
            if (g.kind == 0) x3_launch_conv(b, t, sa);
            if (g.kind == 1) wino_launch_conv(b, wmt, sa);
            if (g.kind == 2) dma_launch_conv(b, shp, dt, sa);
            if (g.kind >= 3) launch_synth(g.kind, sa);
            if (l % 4 == 1) {
                const dim3 wgrid((4 * LH + 15) / 16, 2, (LH + 63) / 64);
                if (variant < 0) launch_lstm_whh_grad(ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH, 0, dpart, sb);
                if (variant == 0) hipLaunchKernelGGL(whh_variant_kernel<0>, wgrid, dim3(256), 0, sb, ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH);
                if (variant == 1) hipLaunchKernelGGL(whh_variant_kernel<1>, wgrid, dim3(256), 0, sb, ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH);
                if (variant == 2) hipLaunchKernelGGL(whh_variant_kernel<2>, wgrid, dim3(256), 0, sb, ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH);
                if (variant == 3) hipLaunchKernelGGL(whh_variant_kernel<3>, wgrid, dim3(256), 0, sb, ddg, dh, ddw, ddw + (size_t)LG * LH, LN, LT, LH);
                VR_HIP(hipMemcpyAsync(wcur.data(), ddw, wcur.size() * 4, hipMemcpyDeviceToHost, sb));
                VR_HIP(hipStreamSynchronize(sb));
                ++wruns;
                int nd_ = 0, first = -1;
                for (size_t i = 0; i < wcur.size(); ++i) if (std::memcmp(&wcur[i], &wref_v[i], 4)) { if (first < 0) first = (int)i; ++nd_; }
                for (size_t i = 0; i < wcur.size(); ++i) if (std::memcmp(&wcur[i], &wref_v[i], 4)) { ++rowq[((i % (LG * LH)) / LH) & 3]; ++colq[(i % LH) >> 4]; }
                if (nd_) { ++wbad; if (wbad <= 1 && variant < 0) printf("    weight_hh run %d: %d elements differ, first at row %d col %d (%.6e vs %.6e)\n", wruns, nd_,
                                                         (first % (LG * LH)) / LH, first % LH, (double)wcur[first], (double)wref_v[first]); }
            }
        }
        VR_HIP(hipDeviceSynchronize());
        printf("%-40s weight_hh variant %2d: %d of %d runs differ from the solo run; differing elements by row%%4: %d %d %d %d, by column/16: %d %d %d %d\n",
               g.name, variant, wbad, wruns, rowq[0], rowq[1], rowq[2], rowq[3], colq[0], colq[1], colq[2], colq[3]);
        }
        float ms; VR_HIP(hipEventElapsedTime(&ms, e0, e1));
        unsigned cnt;
        VR_HIP(hipMemcpy(&cnt, dcount, 4, hipMemcpyDeviceToHost));
        printf("%-40s %d launches in %.1f ms: %u canary hits\n", g.name, launches, ms, cnt);
        if (cnt) {
            std::vector<unsigned> lg(4 * std::min(cnt, 4096u));
            VR_HIP(hipMemcpy(lg.data(), dlog, lg.size() * 4, hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < std::min(cnt, 24u); ++i) {
                float f; std::memcpy(&f, &lg[4 * i + 2], 4);
                printf("    block %u thread %u (lane %u) %s %u: saw 0x%08x (%.4e) at iteration %u\n", lg[4 * i] >> 8, lg[4 * i] & 255u, lg[4 * i] & 63u,
                       (lg[4 * i + 1] & 0x10000u) ? "LDS word" : "register", lg[4 * i + 1] & 0xffffu, lg[4 * i + 2], (double)f, lg[4 * i + 3]);
            }
            // histogram over register index and lane quarter
            int byreg[R] = {}, byq[4] = {}, ldsn = 0;
            for (unsigned i = 0; i < std::min(cnt, 4096u); ++i) {
                if (lg[4 * i + 1] & 0x10000u) { ++ldsn; continue; }
                ++byreg[lg[4 * i + 1]]; ++byq[(lg[4 * i] & 63u) >> 4];
            }
            printf("    lane quarters: %d %d %d %d; LDS hits %d; registers:", byq[0], byq[1], byq[2], byq[3], ldsn);
            for (int i = 0; i < R; ++i) if (byreg[i]) printf(" r%d:%d", i, byreg[i]);
            printf("\n");
        }
        fflush(stdout);
    }
    // ---- third part: the conv kernels themselves as victims: each tiling alone, then beside every aggressor, bit-compared ----
    {
        printf("conv kernels as victims (output of %d launches beside an aggressor vs the solo output, bit-compared):\n", launches / 4);
        const size_t nout = (size_t)N * Cout * H * W;
        float* dout2 = dalloc(nout);
        std::vector<float> ref(nout), cur(nout);
        struct Vic { const char* name; int kind, MT, TH, flavour; };      // flavour 1: bias + folded BatchNorm epilogue; 2: + the source seen through the fused x2 upsample
        const Vic vics[] = {{"conv_x3<64,8>", 0, 64, 8, 0}, {"conv_x3<32,16>", 0, 32, 16, 0}, {"conv_x3<32,8>", 0, 32, 8, 0}, {"conv_dma", 2, 0, 0, 0},
                            {"x3<64,8> bias+epi", 0, 64, 8, 1}, {"x3<32,16> bias+epi", 0, 32, 16, 1}, {"x3<32,8> bias+epi", 0, 32, 8, 1}, {"dma bias+epi", 2, 0, 0, 1},
                            {"x3<64,8> up", 0, 64, 8, 2}, {"x3<32,16> up", 0, 32, 16, 2}, {"x3<32,8> up", 0, 32, 8, 2}};
        std::vector<float> hbias(Cout), hepi(2 * Cout);
        for (auto& x : hbias) x = nd(rng) * 1e-4f;
        for (int i = 0; i < Cout; ++i) { hepi[2 * i] = 1.f + 0.1f * nd(rng); hepi[2 * i + 1] = nd(rng) * 1e-4f; }
        float* dbias = dalloc(Cout);
        float* depi = dalloc(2 * Cout);
        VR_HIP(hipMemcpy(dbias, hbias.data(), Cout * 4, hipMemcpyHostToDevice));
        VR_HIP(hipMemcpy(depi, hepi.data(), 2 * Cout * 4, hipMemcpyHostToDevice));
        for (const Vic& v : vics) {
            ConvArgs vb = a;
            vb.dst[0].p = dout2;
            if (v.flavour >= 1) { vb.bias = dbias; vb.epi = depi; vb.epi_slope = 0.01f; }
            if (v.flavour == 2) {                             // the same buffer read as a half-resolution source under the bilinear x2
                ConvSrc& c2 = vb.src[0];
                c2.H = H / 2; c2.W = W / 2; c2.sH = W / 2; c2.sC = (long long)(H / 2) * (W / 2); c2.sN = c2.sC * Cin; c2.up = 1;
                c2.rh = (float)(H / 2 - 1) / (float)(H - 1); c2.rw = (float)(W / 2 - 1) / (float)(W - 1);
            }
            X3Tile vt{v.MT, v.TH};
            DmaTile vdt{};
            if (v.kind == 0) { vb.x3w = dx3; vb.bf16 = 2; x3_fill_tiling(vb, vt); }
            if (v.kind == 2) { if (!dma_pick(vb, shp, &vdt)) continue; dma_fill_tiling(vb, vdt); }
            auto launch_v = [&](hipStream_t st) { if (v.kind == 0) x3_launch_conv(vb, vt, st); else dma_launch_conv(vb, shp, vdt, st); };
            launch_v(0);
            VR_HIP(hipDeviceSynchronize());
            VR_HIP(hipMemcpy(ref.data(), dout2, nout * 4, hipMemcpyDeviceToHost));
            for (const Agg& g : aggs) {
                if (g.dbg) continue;
                ConvArgs b = a;
                X3Tile t{g.MT, g.TH};
                int wmt = 0;
                DmaTile dt{};
                if (g.kind == 0) { b.x3w = dx3; b.bf16 = 2; x3_fill_tiling(b, t); }
                if (g.kind == 1) { b.wino = dwino; if (!wino_pick(b, shp, &wmt)) continue; wino_fill_tiling(b, wmt); }
                if (g.kind == 2) { if (!dma_pick(b, shp, &dt)) continue; dma_fill_tiling(b, dt); }
                int bad = 0, runs = 0;
                size_t nbad = 0, firstbad = 0;
                for (int l = 0; l < launches / 4; ++l) {
                    for (int rep = 0; rep < 3; ++rep) {
                        if (g.kind == 0) x3_launch_conv(b, t, sa);
                        if (g.kind == 1) wino_launch_conv(b, wmt, sa);
                        if (g.kind == 2) dma_launch_conv(b, shp, dt, sa);
            if (g.kind >= 3) launch_synth(g.kind, sa);
                    }
                    VR_HIP(hipMemsetAsync(dout2, 0xff, nout * 4, sb));
                    launch_v(sb);
                    VR_HIP(hipMemcpyAsync(cur.data(), dout2, nout * 4, hipMemcpyDeviceToHost, sb));
                    VR_HIP(hipStreamSynchronize(sb));
                    ++runs;
                    size_t nb = 0;
                    for (size_t i = 0; i < nout; ++i) if (std::memcmp(&cur[i], &ref[i], 4)) { if (!nb && !bad) firstbad = i; ++nb; }
                    if (nb) { ++bad; nbad += nb; }
                }
                VR_HIP(hipDeviceSynchronize());
                printf("  %-16s beside %-22s: %d of %d runs differ (%zu elements)", v.name, g.name, bad, runs, nbad);
                if (bad) printf("; first at element %zu (n %zu, cout %zu, h %zu, w %zu)", firstbad, firstbad / ((size_t)Cout * H * W), (firstbad / ((size_t)H * W)) % Cout,
                                (firstbad / W) % H, firstbad % W);
                printf("\n");
                fflush(stdout);
            }
        }
    }
    // ---- fourth part: the bidirectional LSTM (registers + LDS broadcast of h, compiler-scheduled) as victim ----
    for (int LHv : {64, 32}) {
        const int BN = 6, BT = 256, BG = 4 * LHv;
        std::vector<float> hgx((size_t)BN * 2 * BG * BT), hwf((size_t)BG * LHv), hwr((size_t)BG * LHv);
        for (auto& v : hgx) v = nd(rng);
        for (auto& v : hwf) v = nd(rng) * 0.1f;
        for (auto& v : hwr) v = nd(rng) * 0.1f;
        float* dgx = dalloc(hgx.size());
        float* dwf = dalloc(hwf.size());
        float* dwr = dalloc(hwr.size());
        float* dho = dalloc((size_t)BN * 2 * LHv * BT);
        VR_HIP(hipMemcpy(dgx, hgx.data(), hgx.size() * 4, hipMemcpyHostToDevice));
        VR_HIP(hipMemcpy(dwf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice));
        VR_HIP(hipMemcpy(dwr, hwr.data(), hwr.size() * 4, hipMemcpyHostToDevice));
        const size_t nh = (size_t)BN * 2 * LHv * BT;
        std::vector<float> ref(nh), cur(nh);
        launch_bilstm(dgx, dwf, dwr, dho, BN, BT, LHv, 0);
        VR_HIP(hipDeviceSynchronize());
        VR_HIP(hipMemcpy(ref.data(), dho, nh * 4, hipMemcpyDeviceToHost));
        for (const Agg& g : aggs) {
            if (g.dbg) continue;
            ConvArgs b = a;
            X3Tile t{g.MT, g.TH};
            int wmt = 0;
            DmaTile dt{};
            if (g.kind == 0) { b.x3w = dx3; b.bf16 = 2; x3_fill_tiling(b, t); }
            if (g.kind == 1) { b.wino = dwino; if (!wino_pick(b, shp, &wmt)) continue; wino_fill_tiling(b, wmt); }
            if (g.kind == 2) { if (!dma_pick(b, shp, &dt)) continue; dma_fill_tiling(b, dt); }
            int bad = 0, runs = 0;
            size_t nbad = 0;
            for (int l = 0; l < launches / 4; ++l) {
                for (int rep = 0; rep < 3; ++rep) {
                    if (g.kind == 0) x3_launch_conv(b, t, sa);
                    if (g.kind == 1) wino_launch_conv(b, wmt, sa);
                    if (g.kind == 2) dma_launch_conv(b, shp, dt, sa);
            if (g.kind >= 3) launch_synth(g.kind, sa);
                }
                launch_bilstm(dgx, dwf, dwr, dho, BN, BT, LHv, sb);
                VR_HIP(hipMemcpyAsync(cur.data(), dho, nh * 4, hipMemcpyDeviceToHost, sb));
                VR_HIP(hipStreamSynchronize(sb));
                ++runs;
                size_t nb = 0;
                for (size_t i = 0; i < nh; ++i) if (std::memcmp(&cur[i], &ref[i], 4)) ++nb;
                if (nb) { ++bad; nbad += nb; }
            }
            VR_HIP(hipDeviceSynchronize());
            printf("  bilstm H=%-3d     beside %-22s: %d of %d runs differ (%zu elements)\n", LHv, g.name, bad, runs, nbad);
            fflush(stdout);
        }
    }
    return 0;
}
