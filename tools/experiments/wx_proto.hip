// Stand-alone bench + check of the 3x3 stride-1 weight-gradient kernels on one MI355X: wgrad_x3.hip (split-bf16 direct) against
// wgrad_wino.hip (Winograd F(3x3,2x2), fp32 MFMA) through the library's own launch_wgrad, on seeded tensors, against a
// double-precision reference at sampled (cout, cin, tap) entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vocal-remover_amd/csrc tools/experiments/wx_proto.hip -o /tmp/wx_proto
// (experiment of round 3, NOT part of the library: see profiles/README.md -- slower than wgrad_wino.hip and not yet correct)
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "wgrad_x3.hip"
#include "../../vocal-remover_amd/csrc/wgrad_wino.hip"
#include "../../vocal-remover_amd/csrc/wgrad_gemm.hip"
#include "../../vocal-remover_amd/csrc/wgrad_mfma.hip"

using namespace vr;

struct SrcDef { int C; int halo; };

// one workgroup per sampled entry (co, ci, tap): sum over all pixels in double
__global__ void ref_entries_kernel(const ConvArgs a, const float* __restrict__ dz, long long zN, long long zC, long long zH, int Cout,
                                   const int* __restrict__ ent, double* __restrict__ ref) {
    const int e = blockIdx.x;
    const int co = ent[3 * e], ci = ent[3 * e + 1], tap = ent[3 * e + 2];
    const int ty = tap / 3, tx = tap % 3;
    const int si = (ci >= a.c1) + (ci >= a.c2);
    const int clc = ci - (si == 0 ? 0 : (si == 1 ? a.c1 : a.c2));
    const ConvSrc& c = a.src[si];
    double s = 0.0;
    const long long npx = (long long)a.N * a.Hout * a.Wout;
    for (long long i = threadIdx.x; i < npx; i += blockDim.x) {
        const int w = (int)(i % a.Wout);
        const int h = (int)((i / a.Wout) % a.Hout);
        const int n = (int)(i / ((long long)a.Wout * a.Hout));
        const int hi = h + ty - 1, wi = w + tx - 1;
        if (hi < 0 || hi >= a.Hin || wi < 0 || wi >= a.Win) continue;
        s += (double)c.p[(long long)n * c.sN + (long long)clc * c.sC + (long long)hi * c.sH + wi] *
             (double)dz[(long long)n * zN + (long long)co * zC + (long long)h * zH + w];
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ref[e] = red[0];
}

static float* dalloc(size_t n) { float* p; VR_HIP(hipMalloc(&p, (n ? n : 1) * 4)); return p; }

template <class F>
static double time_us(F&& f, int iters) {
    hipEvent_t e0, e1;
    VR_HIP(hipEventCreate(&e0)); VR_HIP(hipEventCreate(&e1));
    f();
    VR_HIP(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    VR_HIP(hipEventRecord(e1));
    VR_HIP(hipEventSynchronize(e1));
    float ms; VR_HIP(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / iters;
}

static void run_shape(const char* name, int N, std::vector<SrcDef> srcs, int Cout, int H, int W, int zhalo = 0) {
    int Cin = 0;
    for (auto& s : srcs) Cin += s.C;
    const int CoutPad = (Cout + 31) / 32 * 32;
    std::mt19937 rng(77 + Cin * 3 + Cout);
    std::normal_distribution<float> nd(0.f, 1.f);
    WgradArgs a{};
    a.in.nsrc = (int)srcs.size();
    std::vector<float*> bufs;
    for (int i = 0; i < a.in.nsrc; ++i) {
        const int halo = srcs[i].halo, C = srcs[i].C;
        const int Hs = H + 2 * halo, Ws = W + 2 * halo;
        std::vector<float> hx((size_t)N * C * Hs * Ws);
        for (auto& v : hx) v = nd(rng) * std::exp(0.5f * nd(rng));
        float* dx = dalloc(hx.size());
        VR_HIP(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        bufs.push_back(dx);
        ConvSrc c{};
        c.p = dx + (size_t)halo * Ws + halo;
        c.sH = Ws; c.sC = (long long)Hs * Ws; c.sN = c.sC * C;
        c.C = C; c.H = H; c.W = W; c.hsplit = 1 << 30; c.slope = 1.f;
        a.in.src[i] = c;
    }
    a.in.c1 = a.in.nsrc >= 2 ? srcs[0].C : Cin;
    a.in.c2 = a.in.nsrc >= 3 ? srcs[0].C + srcs[1].C : Cin;
    a.in.Cin = Cin; a.in.Cout = Cout; a.in.CoutPad = CoutPad;
    a.in.N = N; a.in.Hout = H; a.in.Wout = W; a.in.Hin = H; a.in.Win = W; a.in.pad_h = 1; a.in.pad_w = 1;
    a.in.d1 = a.in.d2 = 1 << 30;
    const int Hz = H + 2 * zhalo, Wz = W + 2 * zhalo;
    std::vector<float> hz((size_t)N * Cout * Hz * Wz);
    for (auto& v : hz) v = nd(rng) * 0.01f * std::exp(0.5f * nd(rng));
    float* dzb = dalloc(hz.size());
    VR_HIP(hipMemcpy(dzb, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    a.dz = dzb + (size_t)zhalo * Wz + zhalo;
    a.zH = Wz; a.zC = (long long)Hz * Wz; a.zN = a.zC * Cout;
    a.Cout = Cout; a.CoutPad = CoutPad;
    a.allow_wino = 1;
    const ConvShape shp{3, 1, 1, 1};
    const double flops = 2.0 * N * (double)H * W * (double)Cout * Cin * 9;
    const size_t nw = (size_t)Cin * 9 * CoutPad;
    float* dgrad = dalloc(nw);

    // sampled reference entries
    const int nent = 96;
    std::vector<int> hent(3 * nent);
    for (int e = 0; e < nent; ++e) { hent[3 * e] = (int)(rng() % Cout); hent[3 * e + 1] = (int)(rng() % Cin); hent[3 * e + 2] = (int)(rng() % 9); }
    hent[0] = Cout - 1; hent[1] = Cin - 1; hent[2] = 8;      // last cout / cin / tap
    hent[3] = 0; hent[4] = 0; hent[5] = 0;
    int* dent; VR_HIP(hipMalloc(&dent, hent.size() * 4));
    VR_HIP(hipMemcpy(dent, hent.data(), hent.size() * 4, hipMemcpyHostToDevice));
    double* dref; VR_HIP(hipMalloc(&dref, nent * 8));
    hipLaunchKernelGGL(ref_entries_kernel, dim3(nent), dim3(256), 0, 0, a.in, a.dz, a.zN, a.zC, a.zH, Cout, dent, dref);
    std::vector<double> href(nent);
    VR_HIP(hipMemcpy(href.data(), dref, nent * 8, hipMemcpyDeviceToHost));
    double scale = 0;
    for (double r : href) scale = std::fmax(scale, std::fabs(r));

    printf("%-36s N%-2d %3d->%3d %4dx%-3d %7.1f GF |", name, N, Cin, Cout, H, W, flops * 1e-9);
    std::vector<float> hg(nw);
    for (int mode : {2, 0}) {
        WgradArgs b = a;
        b.bf16 = mode;
        const size_t scratch = wgrad_scratch_floats(b, shp);
        float* dpart = dalloc(scratch);
        b.part = dpart;
        VR_HIP(hipMemset(dgrad, 0xff, nw * 4));
        const double us = time_us([&] { launch_wgrad(b, shp, dgrad, 0, 0); }, 3);
        VR_HIP(hipMemcpy(hg.data(), dgrad, nw * 4, hipMemcpyDeviceToHost));
        double e = 0, e2 = 0;
        for (int i = 0; i < nent; ++i) {
            const double d = std::fabs((double)hg[((size_t)hent[3 * i + 1] * 9 + hent[3 * i + 2]) * CoutPad + hent[3 * i]] - href[i]);
            e = std::fmax(e, d); e2 += d * d;
        }
        printf(" %s %8.1f us %6.1f TF max %.2e rms %.2e |", mode == 2 ? "x3  " : "wino", us, flops / us * 1e-6, e / scale, std::sqrt(e2 / nent) / scale);
        hipFree(dpart);
    }
    printf("\n");
    fflush(stdout);
    for (float* p : bufs) hipFree(p);
    hipFree(dzb); hipFree(dgrad); hipFree(dent); hipFree(dref);
}

int main(int argc, char** argv) {
    const int quick = argc > 1 ? atoi(argv[1]) : 0;
    try {
        run_shape("small odd", 1, {{10, 0}}, 20, 37, 48);
        run_shape("3 strided sources 13+8+1", 2, {{13, 2}, {8, 0}, {1, 3}}, 40, 50, 70, 1);
        run_shape("block straddles sources 30+5+40", 2, {{30, 0}, {5, 1}, {40, 0}}, 64, 24, 64);
        run_shape("single tile row", 1, {{64, 0}}, 64, 2, 32);
        if (quick) return 0;
        // the 3x3 stride-1 layers of the benched train step (batch 16)
        run_shape("stg3 dec1 97->32 @1024x256", 16, {{64, 0}, {1, 0}, {32, 0}}, 32, 1024, 256);
        run_shape("stg3 dec2 192->64 @512x128", 16, {{128, 0}, {64, 0}}, 64, 512, 128);
        run_shape("stg3 dec3 320->128 @256x64", 16, {{192, 0}, {128, 0}}, 128, 256, 64);
        run_shape("stg3 dec4 448->192 @128x32", 16, {{256, 0}, {192, 0}}, 192, 128, 32);
        run_shape("stg3 enc2b 64->64 @512x128", 16, {{64, 0}}, 64, 512, 128);
        run_shape("stg3 enc3b 128->128 @256x64", 16, {{128, 0}}, 128, 256, 64);
        run_shape("stg3 enc1 26->32 @1024x256", 16, {{26, 0}}, 32, 1024, 256);
        run_shape("stg2l dec1 97->32 @512x256", 16, {{64, 0}, {1, 0}, {32, 0}}, 32, 512, 256);
        run_shape("stg2l dec2 192->64 @256x128", 16, {{128, 0}, {64, 0}}, 64, 256, 128);
    } catch (const vr::Error& e) {
        printf("ERROR %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
