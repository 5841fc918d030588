// Stand-alone bench + check of the two fp16-split direct 3x3 kernels (mfma_mode 3) on one MI355X: conv_x3h.hip (four waves in lockstep,
// round 4 / 5) against conv_x3pp.hip (round 6: ping-pong of two wave groups) in every tile shape that fits the layer, on the same seeded
// tensors, against a double-precision reference evaluated at sampled output points.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vocal-remover_amd/csrc tools/x3pp_proto.hip -o tools/_build/x3pp_proto
//   tools/_build/x3pp_proto [mode] [dbg]   0 = correctness shapes + the S30 inference layers (N = 11), 1 = correctness only,
//                                          2 = three layers (profiling), 3 = the batch-16 training shapes
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../vocal-remover_amd/csrc/conv_x3.hip"
#include "../vocal-remover_amd/csrc/conv_x3h.hip"
#include "experiments/conv_x3pp.hip"
#include "../vocal-remover_amd/csrc/profile.hip"

using namespace vr;

struct SrcDef { int C; int halo; int up = 0; };     // a source of C channels stored with `halo` extra columns/rows around it (strided view);
                                                   // up: stored at half resolution, seen through the bilinear x2 (align_corners=True)

__global__ void ref_points_kernel(const ConvArgs a, const int* __restrict__ idx, int npts, double* __restrict__ ref) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    long long o = idx[i];
    const int wo = (int)(o % a.Wout); o /= a.Wout;
    const int ho = (int)(o % a.Hout); o /= a.Hout;
    const int co = (int)(o % a.Cout);
    const int n = (int)(o / a.Cout);
    double s = 0.0;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const int si = (ci >= a.c1) + (ci >= a.c2);
        const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
        const ConvSrc& c = a.src[si];
        const float* base = c.p + (long long)n * c.sN + (long long)clc * c.sC;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int hi = ho + ky - 1, wi = wo + kx - 1;
                if (hi < 0 || hi >= a.Hin || wi < 0 || wi >= a.Win) continue;
                double v;
                if (c.up) {
                    const double hr = (double)(c.H - 1) / (double)(2 * c.H - 1) * hi, wr = (double)(c.W - 1) / (double)(2 * c.W - 1) * wi;
                    const int h1 = (int)hr, w1 = (int)wr;
                    const int h1p = h1 < c.H - 1, w1p = w1 < c.W - 1;
                    const double hl = hr - h1, wl = wr - w1;
                    const float* r0 = base + (long long)h1 * c.sH;
                    const float* r1 = r0 + (long long)h1p * c.sH;
                    v = (1 - hl) * ((1 - wl) * r0[w1] + wl * r0[w1 + w1p]) + hl * ((1 - wl) * r1[w1] + wl * r1[w1 + w1p]);
                } else {
                    v = (double)base[(long long)hi * c.sH + wi];
                }
                s += v * (double)a.w[((long long)ci * 9 + ky * 3 + kx) * a.CoutPad + co];
            }
    }
    if (a.bias) s += a.bias[co];
    ref[i] = s;
}

static float* dalloc(size_t n) { float* p; VR_HIP(hipMalloc(&p, (n ? n : 1) * 4)); return p; }

struct Result { double us, err, scale; };

template <class F>
static double time_us(F&& f, int iters) {
    hipEvent_t e0, e1;
    VR_HIP(hipEventCreate(&e0)); VR_HIP(hipEventCreate(&e1));
    f(); f();
    VR_HIP(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    VR_HIP(hipEventRecord(e1));
    VR_HIP(hipEventSynchronize(e1));
    float ms; VR_HIP(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / iters;
}

static int g_dbg = 0;
static void run_shape(const char* name, int N, std::vector<SrcDef> srcs, int Cout, int H, int W, float xscale, int special, int dbg = 0) {
    dbg |= g_dbg;
    int Cin = 0;
    for (auto& s : srcs) Cin += s.C;
    const int CoutPad = (Cout + 31) / 32 * 32;
    std::mt19937 rng(1234 + Cin * 7 + Cout);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hw((size_t)Cin * 9 * CoutPad, 0.f);
    const float wsc = 1.f / std::sqrt((float)Cin * 9.f);
    for (int ci = 0; ci < Cin; ++ci)
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < Cout; ++co) hw[((size_t)ci * 9 + t) * CoutPad + co] = nd(rng) * wsc;
    float* dw = dalloc(hw.size());
    VR_HIP(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hb(Cout);
    for (auto& b : hb) b = nd(rng) * 0.1f * xscale;
    float* dbias = dalloc(Cout);
    VR_HIP(hipMemcpy(dbias, hb.data(), Cout * 4, hipMemcpyHostToDevice));

    ConvArgs a{};
    a.nsrc = (int)srcs.size();
    std::vector<float*> bufs;
    for (int i = 0; i < a.nsrc; ++i) {
        const int halo = srcs[i].halo, C = srcs[i].C, up = srcs[i].up;
        const int Hl = up ? H / 2 : H, Wl = up ? W / 2 : W;
        const int Hs = Hl + 2 * halo, Ws = Wl + 2 * halo;
        std::vector<float> hx((size_t)N * C * Hs * Ws);
        for (auto& v : hx) {
            float x = nd(rng) * std::exp(nd(rng)) * xscale;
            if (special == 1) {                               // subnormals, exact bf16 values, values one ulp off a bf16 boundary
                const int r = (int)(rng() % 6);
                if (r == 0) x = std::ldexp(nd(rng), -140);
                else if (r == 1) { unsigned u; std::memcpy(&u, &x, 4); u &= 0xffff0000u; std::memcpy(&x, &u, 4); }
                else if (r == 2) { unsigned u; std::memcpy(&u, &x, 4); u = (u & 0xffff0000u) | 0x8000u; std::memcpy(&x, &u, 4); }
                else if (r == 3) { unsigned u; std::memcpy(&u, &x, 4); u = (u & 0xffff0000u) | 0x7fffu; std::memcpy(&x, &u, 4); }
                else if (r == 4) { unsigned u; std::memcpy(&u, &x, 4); u = (u & 0xffffff00u) | 0x80u; std::memcpy(&x, &u, 4); }
            }
            if (special == 2) x *= std::ldexp(1.f, (int)(rng() % 5) * 12 - 24);      // per-value scale swings
            if (special == 4) x = 0.f;                                                // zero pixels: how much of the time is the chip's power limit
            v = x;
        }
        if (special == 3) {                                   // per-channel-chunk scale swings: the running shift has to move between chunks
            for (int n = 0; n < N; ++n)
                for (int c = 0; c < C; ++c) {
                    const float f = std::ldexp(1.f, ((c / 8) % 5) * 17 - 30);
                    float* q = hx.data() + ((size_t)n * C + c) * Hs * Ws;
                    for (size_t e = 0; e < (size_t)Hs * Ws; ++e) q[e] *= f;
                }
        }
        float* dx = dalloc(hx.size());
        VR_HIP(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        bufs.push_back(dx);
        ConvSrc c{};
        c.p = dx + (size_t)halo * Ws + halo;
        c.sH = Ws; c.sC = (long long)Hs * Ws; c.sN = c.sC * C;
        c.C = C; c.H = Hl; c.W = Wl; c.hsplit = 1 << 30; c.slope = 1.f; c.up = up;
        c.rh = (float)(Hl - 1) / (float)(2 * Hl - 1); c.rw = (float)(Wl - 1) / (float)(2 * Wl - 1);
        a.src[i] = c;
    }
    a.c1 = a.nsrc >= 2 ? srcs[0].C : Cin;
    a.c2 = a.nsrc >= 3 ? srcs[0].C + srcs[1].C : Cin;
    a.Cin = Cin; a.w = dw; a.bias = dbias; a.Cout = Cout; a.CoutPad = CoutPad;
    const size_t nout = (size_t)N * Cout * H * W;
    float* dout = dalloc(nout);
    a.dst[0] = ConvDst{dout, (long long)Cout * H * W, (long long)H * W, (long long)W, 0, 0};
    a.d1 = a.d2 = 1 << 30;
    a.N = N; a.Hout = H; a.Wout = W; a.Hin = H; a.Win = W; a.pad_h = 1; a.pad_w = 1;
    a.dbg = dbg;
    const double flops = 2.0 * N * (double)H * W * (double)Cout * Cin * 9;

    const int npts = 16384;
    std::vector<int> hidx(npts);
    for (auto& i : hidx) i = (int)(rng() % nout);
    hidx[0] = 0; hidx[1] = (int)(nout - 1); hidx[2] = W - 1; hidx[3] = (H - 1) * W;
    int* didx; VR_HIP(hipMalloc(&didx, npts * 4));
    VR_HIP(hipMemcpy(didx, hidx.data(), npts * 4, hipMemcpyHostToDevice));
    double* dref; VR_HIP(hipMalloc(&dref, npts * 8));
    hipLaunchKernelGGL(ref_points_kernel, dim3((npts + 63) / 64), dim3(64), 0, 0, a, didx, npts, dref);
    std::vector<double> href(npts);
    VR_HIP(hipMemcpy(href.data(), dref, npts * 8, hipMemcpyDeviceToHost));
    double scale = 0;
    for (double r : href) scale = std::fmax(scale, std::fabs(r));
    std::vector<float> hout(nout);
    auto check = [&]() -> double {
        VR_HIP(hipMemcpy(hout.data(), dout, nout * 4, hipMemcpyDeviceToHost));
        double e = 0;
        for (int i = 0; i < npts; ++i) { const double d = std::fabs((double)hout[hidx[i]] - href[i]); e = std::fmax(e, d != d ? 1e30 : d); }
        return e / scale;
    };
    printf("%-38s N%-2d %3d->%3d %4dx%-3d %7.2f GF |", name, N, Cin, Cout, H, W, flops * 1e-9);

    void* dx3; VR_HIP(hipMalloc(&dx3, x3_weights_bytes(Cin, 9, CoutPad)));
    launch_x3h_weights(dw, dx3, Cin, 9, CoutPad, 0);
    double best_h = 1e30, best_p = 1e30;
    // ---- conv_x3h (round 5) ----
    const int mts[3] = {64, 32, 32}, ths[3] = {8, 16, 8};
    for (int v = 0; v < 3; ++v) {
        if (CoutPad % mts[v]) continue;
        ConvArgs b = a;
        b.x3w = dx3; b.bf16 = 3;
        X3Tile t{mts[v], ths[v]};
        x3_fill_tiling(b, t);
        VR_HIP(hipMemset(dout, 0xff, nout * 4));
        const double us = time_us([&] { x3h_launch_conv(b, t, 0); }, 5);
        const double e = (dbg & 15) == 0 ? check() : 0.0;
        printf(" h<%d,%d> %7.1f us %5.0f TF %.1e |", t.MT, t.TH, us, flops / us * 1e-6, e);
        best_h = std::fmin(best_h, us);
    }
    // ---- conv_x3pp (round 6) ----
    for (int pp = 1; pp <= 5; ++pp) {
        if (pp == 4) continue;
        int couts, rows;
        x3pp_tile_dims(pp, &couts, &rows);
        if (CoutPad % (couts >= 64 ? 64 : 32)) continue;
        if (couts == 128 && CoutPad < 128) continue;
        ConvArgs b = a;
        b.x3w = dx3; b.bf16 = 3;
        X3ppTile t{pp};
        b.tiles_w = (b.Wout + 31) / 32;
        b.tiles_h = (b.Hout + rows - 1) / rows;
        b.npt = b.N * b.tiles_h * b.tiles_w;
        b.nct = (b.CoutPad + couts - 1) / couts;
        VR_HIP(hipMemset(dout, 0xff, nout * 4));
        const double us = time_us([&] { x3pp_launch_conv(b, t, 0); }, 5);
        const double e = (dbg & 15) == 0 ? check() : 0.0;
        printf(" pp%d %7.1f us %5.0f TF %.1e |", pp, us, flops / us * 1e-6, e);
        if (dbg & 64) {
            static long long tr[8 * 48 * 3];
            x3pp_trace_read(tr);
            printf("\n  trace pp%d: phase: per wave body+barrier cycles (waves 0-3 = group A, 4-7 = group B)\n", pp);
            for (int ph = 2; ph < 14; ++ph) {
                printf("   ph%2d", ph);
                for (int w = 0; w < 8; ++w) {
                    const long long* q = tr + (w * 48 + ph) * 3;
                    printf(" | %5lld+%-5lld", q[1] - q[0], q[2] - q[1]);
                }
                printf("\n");
            }
        }
        best_p = std::fmin(best_p, us);
    }
    printf(" best h %.1f pp %.1f  %+.0f%%\n", best_h, best_p, (best_p / best_h - 1) * 100);
    fflush(stdout);
    for (float* p : bufs) hipFree(p);
    hipFree(dw); hipFree(dbias); hipFree(dout); hipFree(didx); hipFree(dref); hipFree(dx3);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    g_dbg = argc > 2 ? atoi(argv[2]) : 0;
    try {
        if (mode == 0 || mode == 1) {
            // correctness: odd sizes, partial tiles, partial channel chunks, strided sources, fused upsample, scale extremes
            run_shape("small odd", 1, {{10, 0}}, 20, 37, 48, 1.f, 0);
            run_shape("one chunk", 1, {{5, 0}}, 32, 40, 64, 1.f, 0);
            run_shape("two chunks", 2, {{16, 0}}, 64, 24, 32, 1.f, 0);
            run_shape("3 strided sources 13+8+1", 2, {{13, 2}, {8, 0}, {1, 3}}, 40, 50, 70, 1.f, 0);
            run_shape("3 aligned strided sources", 2, {{13, 4}, {8, 0}, {1, 8}}, 40, 50, 72, 1.f, 0);
            run_shape("64ch single tile", 1, {{64, 0}}, 64, 8, 32, 1.f, 0);
            run_shape("128 couts", 1, {{24, 0}}, 128, 20, 64, 1.f, 0);
            run_shape("192 couts", 1, {{40, 0}}, 192, 33, 32, 1.f, 0);
            run_shape("upsampled 16 + skip 8", 2, {{16, 0, 1}, {8, 0}}, 32, 40, 64, 1.f, 0);
            run_shape("upsampled 13 (strided) + up 1 + skip 10", 2, {{13, 2, 1}, {1, 0, 1}, {10, 3}}, 40, 36, 96, 1.f, 0);
            run_shape("upsampled only, odd tiles", 1, {{24, 0, 1}}, 64, 20, 40, 1.f, 0);
            run_shape("upsampled 64 -> 128 couts", 1, {{64, 0, 1}}, 128, 34, 64, 1.f, 0);
            run_shape("scale 2^-100", 1, {{24, 0}}, 32, 32, 64, std::ldexp(1.f, -100), 0);
            run_shape("scale 2^+100", 1, {{24, 0}}, 32, 32, 64, std::ldexp(1.f, 100), 0);
            run_shape("subnormal / bf16-boundary inputs", 1, {{24, 0}}, 64, 32, 64, 1.f, 1);
            run_shape("per-value scale swings", 1, {{40, 0}}, 64, 32, 64, 1.f, 2);
            run_shape("per-chunk scale swings", 1, {{72, 0}}, 64, 32, 64, 1.f, 3);
            run_shape("per-chunk scale swings, 32 couts", 2, {{72, 0}}, 32, 48, 64, 1.f, 3);
        }
        if (mode == 1) return 0;
        if (mode == 2) {
            run_shape("stg3 dec1 up64+up1+32 ->32 @1024x256", 11, {{64, 0, 1}, {1, 0, 1}, {32, 0}}, 32, 1024, 256, 1.f, 0);
            run_shape("stg3 enc2b 64->64 @512x128", 11, {{64, 0}}, 64, 512, 128, 1.f, 0);
            run_shape("stg3 dec3 320->128 @256x64", 11, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 0);
            return 0;
        }
        if (mode == 5) {                                    // phase trace of one workgroup (dbg bit 64): per wave and phase, body / barrier cycles
            g_dbg |= 64;
            const char* nm = argc > 3 ? argv[3] : "enc3b";
            if (!strcmp(nm, "enc3b")) run_shape("TRACE stg3 enc3b 128->128 @256x64", 11, {{128, 0}}, 128, 256, 64, 1.f, 0);
            else if (!strcmp(nm, "enc2b")) run_shape("TRACE stg3 enc2b 64->64 @512x128", 11, {{64, 0}}, 64, 512, 128, 1.f, 0);
            else run_shape("TRACE stg3 dec1 up ->32 @1024x256", 11, {{64, 0, 1}, {1, 0, 1}, {32, 0}}, 32, 1024, 256, 1.f, 0);
            return 0;
        }
        if (mode == 4) {
            run_shape("ZERO stg3 enc2b 64->64 @512x128", 11, {{64, 0}}, 64, 512, 128, 1.f, 4);
            run_shape("ZERO stg3 dec3 320->128 @256x64", 11, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 4);
            run_shape("ZERO stg3 enc1 26->32 @1024x256", 11, {{26, 0}}, 32, 1024, 256, 1.f, 4);
            return 0;
        }
        if (mode == 3) {
            run_shape("T stg3 dec1 97->32 @1024x256", 16, {{64, 0}, {32, 0}, {1, 0}}, 32, 1024, 256, 1.f, 0);
            run_shape("T stg3 dec2 192->64 @512x128", 16, {{128, 0}, {64, 0}}, 64, 512, 128, 1.f, 0);
            run_shape("T stg3 dec3 320->128 @256x64", 16, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 0);
            run_shape("T stg3 dec4 448->192 @128x32", 16, {{256, 0}, {192, 0}}, 192, 128, 32, 1.f, 0);
            run_shape("T stg3 enc2b 64->64 @512x128", 16, {{64, 0}}, 64, 512, 128, 1.f, 0);
            run_shape("T stg3 enc3b 128->128 @256x64", 16, {{128, 0}}, 128, 256, 64, 1.f, 0);
            run_shape("T stg3 dgrad dec1 32->97 @1024x256", 16, {{32, 0}}, 97, 1024, 256, 1.f, 0);
            run_shape("T stg3 dgrad dec2 64->192 @512x128", 16, {{64, 0}}, 192, 512, 128, 1.f, 0);
            run_shape("T stg2l dec1 97->32 @512x256", 16, {{64, 0}, {32, 0}, {1, 0}}, 32, 512, 256, 1.f, 0);
            run_shape("T stg1l dec1 49->16 @512x256", 16, {{32, 0}, {16, 0}, {1, 0}}, 16, 512, 256, 1.f, 0);
            return 0;
        }
        // the stride-1 3x3 layers of one S30 inference call (N = 11 crops) -- SURVEY.md 8(a-detail); eval: decoders >= 512x128 interpolate in the kernel
        run_shape("stg3 dec1 up64+up1+32 ->32 @1024x256", 11, {{64, 0, 1}, {1, 0, 1}, {32, 0}}, 32, 1024, 256, 1.f, 0);
        run_shape("stg3 dec2 up128+64 ->64 @512x128", 11, {{128, 0, 1}, {64, 0}}, 64, 512, 128, 1.f, 0);
        run_shape("stg3 dec3 320->128 @256x64", 11, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 0);
        run_shape("stg3 dec4 448->192 @128x32", 11, {{256, 0}, {192, 0}}, 192, 128, 32, 1.f, 0);
        run_shape("stg3 enc1 26->32 @1024x256", 11, {{26, 0}}, 32, 1024, 256, 1.f, 0);
        run_shape("stg3 enc2b 64->64 @512x128", 11, {{64, 0}}, 64, 512, 128, 1.f, 0);
        run_shape("stg3 enc3b 128->128 @256x64", 11, {{128, 0}}, 128, 256, 64, 1.f, 0);
        run_shape("stg3 enc4b 192->192 @128x32", 11, {{192, 0}}, 192, 128, 32, 1.f, 0);
        run_shape("stg2l dec1 up64+up1+32 ->32 @512x256", 11, {{64, 0, 1}, {1, 0, 1}, {32, 0}}, 32, 512, 256, 1.f, 0);
        run_shape("stg2l dec2 192->64 @256x128", 11, {{128, 0}, {64, 0}}, 64, 256, 128, 1.f, 0);
        run_shape("stg2l dec3 320->128 @128x64", 11, {{192, 0}, {128, 0}}, 128, 128, 64, 1.f, 0);
        run_shape("stg2l dec4 448->192 @64x32", 11, {{256, 0}, {192, 0}}, 192, 64, 32, 1.f, 0);
        run_shape("stg2l enc1 10->32 @512x256", 11, {{10, 0}}, 32, 512, 256, 1.f, 0);
        run_shape("stg2l enc2b 64->64 @256x128", 11, {{64, 0}}, 64, 256, 128, 1.f, 0);
        run_shape("stg2l enc3b 128->128 @128x64", 11, {{128, 0}}, 128, 128, 64, 1.f, 0);
        run_shape("stg1l dec1 up32+up1+16 ->16 @512x256", 11, {{32, 0, 1}, {1, 0, 1}, {16, 0}}, 16, 512, 256, 1.f, 0);
        run_shape("stg1l dec2 96->32 @256x128", 11, {{64, 0}, {32, 0}}, 32, 256, 128, 1.f, 0);
        run_shape("stg1l enc2b 32->32 @256x128", 11, {{32, 0}}, 32, 256, 128, 1.f, 0);
        run_shape("stg1l dec3 160->64 @128x64", 11, {{96, 0}, {64, 0}}, 64, 128, 64, 1.f, 0);
    } catch (const vr::Error& e) {
        printf("ERROR %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
