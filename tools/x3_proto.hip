// Stand-alone bench + check of the 3x3 stride-1 conv kernels on one MI355X (no Python, no torch): conv_x3.hip (split-bf16 direct),
// conv_wino.hip (Winograd, fp32 MFMA) and conv_dma.hip (direct, fp32 MFMA) on the same seeded tensors, against a double-precision
// reference evaluated at sampled output points.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vocal-remover_amd/csrc tools/x3_proto.hip -o /tmp/x3_proto && /tmp/x3_proto
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../vocal-remover_amd/csrc/conv_x3.hip"
#include "../vocal-remover_amd/csrc/conv_wino.hip"
#include "../vocal-remover_amd/csrc/conv_dma.hip"

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

static void run_shape(const char* name, int N, std::vector<SrcDef> srcs, int Cout, int H, int W, float xscale, int special, int dbg = 0) {
    int Cin = 0;
    for (auto& s : srcs) Cin += s.C;
    const int CoutPad = (Cout + 31) / 32 * 32;
    std::mt19937 rng(1234 + Cin * 7 + Cout);
    std::normal_distribution<float> nd(0.f, 1.f);
    // weights [Cin][9][CoutPad]
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
            v = x;
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
    const ConvShape shp{3, 1, 1, 1};
    const double flops = 2.0 * N * (double)H * W * (double)Cout * Cin * 9;

    // reference points
    const int npts = 16384;
    std::vector<int> hidx(npts);
    for (auto& i : hidx) i = (int)(rng() % nout);
    // always include the four corners of image 0 / last cout
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
    auto check = [&](const char* what) -> double {
        VR_HIP(hipMemcpy(hout.data(), dout, nout * 4, hipMemcpyDeviceToHost));
        double e = 0, e2 = 0;
        for (int i = 0; i < npts; ++i) { const double d = std::fabs((double)hout[hidx[i]] - href[i]); e = std::fmax(e, d); e2 += d * d; }
        (void)what;
        printf(" max %.2e rms %.2e |", e / scale, std::sqrt(e2 / npts) / scale);
        return e / scale;
    };
    printf("%-34s N%d %3d->%3d %4dx%-3d %7.2f GF |", name, N, Cin, Cout, H, W, flops * 1e-9);

    // ---- x3 variants ----
    void* dx3; VR_HIP(hipMalloc(&dx3, x3_weights_bytes(Cin, 9, CoutPad)));
    launch_x3_weights(dw, dx3, Cin, 9, CoutPad, 0);
    const int mts[3] = {64, 32, 32}, ths[3] = {8, 16, 8};
    for (int v = 0; v < 3; ++v) {
        if (CoutPad % mts[v]) continue;
        ConvArgs b = a;
        b.x3w = dx3; b.bf16 = 2;
        X3Tile t{mts[v], ths[v]};
        if (b.Wout < 32) continue;
        x3_fill_tiling(b, t);
        VR_HIP(hipMemset(dout, 0xff, nout * 4));
        const double us = time_us([&] { x3_launch_conv(b, t, 0); }, 5);
        printf(" x3<%d,%d> %7.1f us %6.1f TF", t.MT, t.TH, us, flops / us * 1e-6);
        if ((dbg & 15) == 0) check("x3"); else printf(" |");
    }
    // ---- Winograd fp32 ----
    bool has_up = false;
    for (auto& sd : srcs) has_up = has_up || sd.up;
    if (dbg == 0 && !has_up) {
        float* dwino = dalloc((size_t)Cin * 16 * CoutPad);
        launch_wino_weights(dw, dwino, Cin, CoutPad, 0);
        ConvArgs b = a;
        b.wino = dwino; b.bf16 = 0;
        int MT;
        if (wino_pick(b, shp, &MT)) {
            wino_fill_tiling(b, MT);
            VR_HIP(hipMemset(dout, 0xff, nout * 4));
            const double us = time_us([&] { wino_launch_conv(b, MT, 0); }, 5);
            printf(" wino<%d> %7.1f us %6.1f TF", MT, us, flops / us * 1e-6);
            check("wino");
        }
        ConvArgs c = a;
        DmaTile dt;
        if (dma_pick(c, shp, &dt)) {
            dma_fill_tiling(c, dt);
            VR_HIP(hipMemset(dout, 0xff, nout * 4));
            const double us = time_us([&] { dma_launch_conv(c, shp, dt, 0); }, 5);
            printf(" dma<%d,%d> %7.1f us %6.1f TF", dt.MT, dt.TH, us, flops / us * 1e-6);
            check("dma");
        }
        hipFree(dwino);
    }
    printf("\n");
    fflush(stdout);
    for (float* p : bufs) hipFree(p);
    hipFree(dw); hipFree(dbias); hipFree(dout); hipFree(didx); hipFree(dref); hipFree(dx3);
}

int main(int argc, char** argv) {
    const int quick = argc > 1 ? atoi(argv[1]) : 0;
    try {
        if (quick != 2) {
        // correctness: odd sizes, partial tiles, partial channel chunks, three strided sources
        run_shape("small odd", 1, {{10, 0}}, 20, 37, 48, 1.f, 0);
        run_shape("3 strided sources 13+8+1", 2, {{13, 2}, {8, 0}, {1, 3}}, 40, 50, 70, 1.f, 0);
        run_shape("3 aligned strided sources", 2, {{13, 4}, {8, 0}, {1, 8}}, 40, 50, 72, 1.f, 0);
        run_shape("64ch single tile", 1, {{64, 0}}, 64, 8, 32, 1.f, 0);
        run_shape("upsampled 16 + skip 8", 2, {{16, 0, 1}, {8, 0}}, 32, 40, 64, 1.f, 0);
        run_shape("upsampled 13 (strided) + up 1 + skip 10", 2, {{13, 2, 1}, {1, 0, 1}, {10, 3}}, 40, 36, 96, 1.f, 0);
        run_shape("upsampled only, odd tiles", 1, {{24, 0, 1}}, 64, 20, 40, 1.f, 0);
        run_shape("scale 2^-100", 1, {{24, 0}}, 32, 32, 64, std::ldexp(1.f, -100), 0);
        run_shape("scale 2^+100", 1, {{24, 0}}, 32, 32, 64, std::ldexp(1.f, 100), 0);
        run_shape("subnormal / bf16-boundary inputs", 1, {{24, 0}}, 64, 32, 64, 1.f, 1);
        }
        if (quick == 1) return 0;
        if (quick == 2) {                                   // profiling run (rocprofv3 --pmc): the two layers the tuning is about
            run_shape("stg3 dec1 97->32 @1024x256", 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 1024, 256, 1.f, 0);
            run_shape("stg3 dec2 192->64 @512x128", 6, {{128, 0}, {64, 0}}, 64, 512, 128, 1.f, 0);
            run_shape("dbg1 dec1", 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 1024, 256, 1.f, 0, 1);
            return 0;
        }
        // the stride-1 3x3 layers of one inference lane (N = 6 crops) -- SURVEY.md §8(a-detail)
        run_shape("stg3 dec1 97->32 @1024x256", 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 1024, 256, 1.f, 0);
        run_shape("stg3 dec2 192->64 @512x128", 6, {{128, 0}, {64, 0}}, 64, 512, 128, 1.f, 0);
        run_shape("stg3 dec3 320->128 @256x64", 6, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 0);
        run_shape("stg3 dec4 448->192 @128x32", 6, {{256, 0}, {192, 0}}, 192, 128, 32, 1.f, 0);
        run_shape("stg3 enc2b 64->64 @512x128", 6, {{64, 0}}, 64, 512, 128, 1.f, 0);
        run_shape("stg3 enc3b 128->128 @256x64", 6, {{128, 0}}, 128, 256, 64, 1.f, 0);
        run_shape("stg3 enc1 26->32 @1024x256", 6, {{26, 0}}, 32, 1024, 256, 1.f, 0);
        run_shape("stg2l dec1 97->32 @512x256", 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 512, 256, 1.f, 0);
        run_shape("stg2l dec2 192->64 @256x128", 6, {{128, 0}, {64, 0}}, 64, 256, 128, 1.f, 0);
        run_shape("stg1l dec1 49->16 @512x256", 6, {{32, 0}, {16, 0}, {1, 0}}, 16, 512, 256, 1.f, 0);
        // the same decoder layers with the x2 upsample fused (eval mode: the upsampled tensor is never materialised)
        run_shape("stg3 dec1 up64+up1+32 ->32 @1024x256", 6, {{64, 0, 1}, {1, 0, 1}, {32, 0}}, 32, 1024, 256, 1.f, 0);
        run_shape("stg3 dec2 up128+64 ->64 @512x128", 6, {{128, 0, 1}, {64, 0}}, 64, 512, 128, 1.f, 0);
        run_shape("stg3 dec3 up192+128 ->128 @256x64", 6, {{192, 0, 1}, {128, 0}}, 128, 256, 64, 1.f, 0);
        // ablations of the x3 kernel on two layers: 16 = tiles interleaved over the XCDs (the other kernels' mapping) instead of a
        // contiguous range per XCD; 2 = no MFMA, 3 = no split pass, 1 = no pixel loads, 4 = no epilogue
        for (int dbg : {16, 3, 1, 4}) {
            char nm[64];
            snprintf(nm, sizeof nm, "dbg%d stg3 dec1 97->32", dbg);
            run_shape(nm, 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 1024, 256, 1.f, 0, dbg);
            snprintf(nm, sizeof nm, "dbg%d stg3 dec2 192->64", dbg);
            run_shape(nm, 6, {{128, 0}, {64, 0}}, 64, 512, 128, 1.f, 0, dbg);
            if (dbg == 16) {
                run_shape("dbg16 stg3 dec3 320->128", 6, {{192, 0}, {128, 0}}, 128, 256, 64, 1.f, 0, dbg);
                run_shape("dbg16 stg3 enc2b 64->64", 6, {{64, 0}}, 64, 512, 128, 1.f, 0, dbg);
                run_shape("dbg16 stg2l dec1 97->32", 6, {{64, 0}, {32, 0}, {1, 0}}, 32, 512, 256, 1.f, 0, dbg);
            }
        }
    } catch (const vr::Error& e) {
        printf("ERROR %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
