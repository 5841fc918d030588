// Test hooks (tests/ only): single kernels of the training path behind vr_debug_kernel, host pointers in and out,
// so that every backward kernel has an isolated parity test against torch autograd (tests/test_gpu_kernels.py).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "model.h"

namespace vr {

namespace {

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n_) : n(n_) {
        VR_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(float)));
        // hipMemset on device memory returns before the fill has run, and the fill is on the NULL stream: the kernels of these hooks
        // run on the handle's non-blocking stream, which does not order against it -- wait here, or a late fill wipes a kernel's output
        // (seen once in round 5: test_gpu_hazard 'rows' victim off by whole values beside a busy aggressor)
        VR_HIP(hipMemset(p, 0, (n ? n : 1) * sizeof(float)));
        VR_HIP(hipStreamSynchronize(nullptr));
    }
    DevBuf(const float* host, size_t n_) : n(n_) {
        VR_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(float)));
        if (host && n) VR_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
    }
    ~DevBuf() { if (p) hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void download(float* host) const { if (host && n) VR_HIP(hipMemcpy(host, p, n * sizeof(float), hipMemcpyDeviceToHost)); }
};

Tensor dense(float* p, int N, int C, int H, int W) {
    Tensor t;
    t.p = p; t.N = N; t.C = C; t.H = H; t.W = W;
    t.sH = W; t.sC = (long long)H * W; t.sN = t.sC * C; t.slope = 1.f;
    return t;
}

}  // namespace

// name            dims                  fparams          inputs                                              outputs
// bn_backward     N,C,H,W               slope,eps,mom    z, G, gamma, beta, post[N][C]|null, rm[C], rv[C]    dz, dgamma, dbeta, affine[C][2], rm, rv
// lstm            N,T,H                 -                gx[N][8H][T], whh_f[4H][H], whh_r, dh[N][2H][T]     h[N][2H][T], dgx[N][8H][T], dwhh_f, dwhh_r
// upsample        N,C,H,W               -                x[N,C,H,W], dhi[N,C,2H,2W]                          up[N,C,2H,2W], glo[N,C,H,W]
// pool            N,C,H,W               -                x, gp[N,C,W], d[N,C,H,W]                            pooled[N,C,W], g[N,C,H,W], sumh[N,C,W]
// thin            N,C,H,W,CO            slope            x, aff[C][2]|null, w[CO][C], dz[N,CO,H,W]           g[N,C,H,W], dw[CO][C], z[N,H,W] (CO=1: forward)
// head_loss       N,C,H,W,bins          slope,gscale     x, aff|null, w[2][C], X[N,2,bins,W], Y              dlogit[N,2,H,W], mask[N,2,bins,W], loss[1]
// rows            N,R,W                 -                x[N,R,W], aff[R][2], d[N,R,W]                       relu(x*a+b), channel sums of d [R]
// adam            n                     lr,b1,b2,eps,gscale,step   p, g, m, v                                p, m, v
void Model::debug_kernel(const std::string& name, const int64_t* dims, int ndims, const float* fp, int nfp,
                         const float* const* in, int nin, float* const* out, int nout) {
    DeviceGuard dev_guard(device);
    auto need = [&](int nd, int nf, int ni, int no) {
        VR_CHECK(ndims >= nd && nfp >= nf && nin >= ni && nout >= no, -2, "vr_debug_kernel(" + name + "): too few arguments");
    };
    hipStream_t st = stream;
    if (name == "x3h_trace") {
        // conv_x3h.hip phase stamps (VR_CONV_DBG bit 64).  dims[0]: 0 = clear, 1 = read into out[0] (768 floats: cycles relative to the
        // earliest stamp, -1 where nothing was stamped); layout [workgroup 4][wave 4][chunk 6][point 8]
        need(1, 0, 0, dims[0] ? 1 : 0);
        VR_HIP(hipDeviceSynchronize());
        if (!dims[0]) { x3h_trace_clear(); return; }
        long long raw[768];
        x3h_trace_read(raw, 768);
        long long lo = 0;
        for (long long v : raw) if (v && (!lo || v < lo)) lo = v;
        for (int i = 0; i < 768; ++i) out[0][i] = raw[i] ? (float)(raw[i] - lo) : -1.f;
        return;
    }
    if (name == "bn_backward") {
        need(4, 3, 7, 6);
        const int N = (int)dims[0], C = (int)dims[1], H = (int)dims[2], W = (int)dims[3];
        const size_t n = (size_t)N * C * H * W;
        DevBuf z(in[0], n), g(in[1], n), gamma(in[2], C), beta(in[3], C), rm(in[5], C), rv(in[6], C);
        DevBuf post(in[4], in[4] ? (size_t)N * C : 0);
        // forward statistics: one partial row of (sum, sumsq) per channel, then the library's own finalize
        std::vector<float> part((size_t)C * 2);
        for (int c = 0; c < C; ++c) {
            double s1 = 0, s2 = 0;
            for (int b = 0; b < N; ++b) {
                const float* q = in[0] + ((size_t)b * C + c) * H * W;
                for (size_t i = 0; i < (size_t)H * W; ++i) { s1 += q[i]; s2 += (double)q[i] * q[i]; }
            }
            part[2 * c] = (float)s1; part[2 * c + 1] = (float)s2;
        }
        DevBuf dpart(part.data(), part.size()), aff((size_t)C * 2), smean(C), sinv(C), dgamma(C), dbeta(C);
        BNFinalizeArgs f{};
        f.part = dpart.p; f.nparts = 1; f.pstride = C * 2; f.count = (double)N * H * W;
        f.w = gamma.p; f.b = beta.p; f.rm = rm.p; f.rv = rv.p; f.affine = aff.p; f.save_mean = smean.p; f.save_invstd = sinv.p;
        f.C = C; f.eps = fp[1]; f.momentum = fp[2]; f.broadcast = 0;
        launch_bn_finalize(f, st);
        BnBwdArgs a{};
        a.g = g.p; a.z = z.p; a.N = N; a.C = C; a.H = H; a.W = W; a.sH = W; a.sC = (long long)H * W; a.sN = a.sC * C;
        a.aff = aff.p; a.aff_bcast = 0; a.slope = fp[0]; a.post = in[4] ? post.p : nullptr;
        a.gamma = gamma.p; a.save_mean = smean.p; a.save_invstd = sinv.p; a.dgamma = dgamma.p; a.dbeta = dbeta.p; a.acc_grads = 1;
        DevBuf coef((size_t)C * 3), bpart((size_t)bn_bwd_chunks(a) * C * 2);
        a.coef = coef.p; a.part = bpart.p;
        launch_bn_bwd(a, st);
        VR_HIP(hipStreamSynchronize(st));
        g.download(out[0]); dgamma.download(out[1]); dbeta.download(out[2]); aff.download(out[3]); rm.download(out[4]); rv.download(out[5]);
    } else if (name == "lstm") {
        need(3, 0, 4, 4);
        const int N = (int)dims[0], T = (int)dims[1], H = (int)dims[2], G = 4 * H;
        DevBuf gx(in[0], (size_t)N * 2 * G * T), wf(in[1], (size_t)G * H), wr(in[2], (size_t)G * H), dh(in[3], (size_t)N * 2 * H * T);
        DevBuf h((size_t)N * 2 * H * T), save((size_t)N * 2 * T * 5 * H), dgx((size_t)N * 2 * G * T), dwf((size_t)G * H), dwr((size_t)G * H);
        launch_bilstm_train(gx.p, wf.p, wr.p, h.p, save.p, N, T, H, st);
        launch_bilstm_bwd(dh.p, save.p, wf.p, wr.p, dgx.p, N, T, H, st);
        DevBuf wpart(lstm_whh_grad_scratch_floats(N, H));
        launch_lstm_whh_grad(dgx.p, h.p, dwf.p, dwr.p, N, T, H, 1, wpart.p, st);
        VR_HIP(hipStreamSynchronize(st));
        h.download(out[0]); dgx.download(out[1]); dwf.download(out[2]); dwr.download(out[3]);
    } else if (name == "upsample") {
        need(4, 0, 2, 2);
        const int N = (int)dims[0], C = (int)dims[1], H = (int)dims[2], W = (int)dims[3];
        const size_t n = (size_t)N * C * H * W;
        DevBuf x(in[0], n), dhi(in[1], 4 * n), up(4 * n), glo(n);
        launch_upsample2x(dense(x.p, N, C, H, W), up.p, st);
        launch_upsample_bwd(dhi.p, N, C, H, W, glo.p, (long long)C * H * W, (long long)H * W, W, 1, st);
        VR_HIP(hipStreamSynchronize(st));
        up.download(out[0]); glo.download(out[1]);
    } else if (name == "pool") {
        need(4, 0, 3, 3);
        const int N = (int)dims[0], C = (int)dims[1], H = (int)dims[2], W = (int)dims[3];
        const size_t n = (size_t)N * C * H * W, m = (size_t)N * C * W;
        DevBuf x(in[0], n), gp(in[1], m), d(in[2], n), pooled(m), g(n), sumh(m);
        launch_avgpool_h(dense(x.p, N, C, H, W), pooled.p, st);
        launch_avgpool_bwd(gp.p, g.p, N, C, H, W, (long long)C * H * W, (long long)H * W, W, 1, st);
        launch_sum_h(d.p, N, C, H, W, sumh.p, st);
        VR_HIP(hipStreamSynchronize(st));
        pooled.download(out[0]); g.download(out[1]); sumh.download(out[2]);
    } else if (name == "thin") {
        need(5, 1, 4, 3);
        const int N = (int)dims[0], C = (int)dims[1], H = (int)dims[2], W = (int)dims[3], CO = (int)dims[4];
        VR_CHECK(CO == 1 || CO == 2, -2, "thin: CO must be 1 or 2");
        const size_t n = (size_t)N * C * H * W, nz = (size_t)N * CO * H * W;
        DevBuf x(in[0], n), aff(in[1], in[1] ? (size_t)C * 2 : 0), w(in[2], (size_t)CO * C), dz(in[3], nz), g(n), dw((size_t)CO * C);
        Tensor t = dense(x.p, N, C, H, W);
        t.slope = fp[0];
        if (in[1]) t.aff0 = aff.p;
        DevBuf part((size_t)thin_wgrad_blocks(t) * CO * C);
        launch_thin_dgrad(t, CO, w.p, dz.p, g.p, 0, st);          // store, then accumulate once more: the caller expects 2 x the gradient
        launch_thin_dgrad(t, CO, w.p, dz.p, g.p, 1, st);
        launch_thin_wgrad(t, CO, dz.p, part.p, dw.p, 0, st);
        DevBuf zf((size_t)N * H * W);
        if (CO == 1) launch_squeeze_conv(t, w.p, zf.p, nullptr, false, st);
        VR_HIP(hipStreamSynchronize(st));
        g.download(out[0]); dw.download(out[1]);
        if (CO == 1) zf.download(out[2]);
    } else if (name == "head_loss") {
        need(5, 2, 5, 3);
        const int N = (int)dims[0], C = (int)dims[1], H = (int)dims[2], W = (int)dims[3], bins = (int)dims[4];
        const size_t n = (size_t)N * C * H * W, nx = (size_t)N * 2 * bins * W;
        DevBuf x(in[0], n), aff(in[1], in[1] ? (size_t)C * 2 : 0), w(in[2], (size_t)2 * C), X(in[3], nx), Y(in[4], nx);
        Tensor t = dense(x.p, N, C, H, W);
        t.slope = fp[0];
        if (in[1]) t.aff0 = aff.p;
        DevBuf dlogit((size_t)N * 2 * H * W), mask(nx), lpart((size_t)head_loss_blocks(t)), loss(4);
        launch_head_loss(t, w.p, X.p, Y.p, bins, fp[1], dlogit.p, mask.p, lpart.p, loss.p, (float)(1.0 / (double)nx), st);
        VR_HIP(hipStreamSynchronize(st));
        dlogit.download(out[0]); mask.download(out[1]);
        VR_HIP(hipMemcpy(out[2], loss.p, sizeof(float), hipMemcpyDeviceToHost));
    } else if (name == "rows") {
        need(3, 0, 3, 2);
        const int N = (int)dims[0], R = (int)dims[1], W = (int)dims[2];
        const size_t n = (size_t)N * R * W;
        DevBuf x(in[0], n), aff(in[1], (size_t)R * 2), d(in[2], n), o(n), sums(R);
        launch_rows_affine_relu(x.p, o.p, aff.p, N, R, W, st);
        launch_channel_sum(d.p, N, R, W, sums.p, 0, st);
        VR_HIP(hipStreamSynchronize(st));
        o.download(out[0]); sums.download(out[1]);
    } else if (name == "adam") {
        need(1, 6, 4, 3);
        const size_t n = (size_t)dims[0];
        // the float parameters arrive as fp32; snap them back to the decimal doubles the optimizer was built with
        double dp[6];
        for (int i = 0; i < 6; ++i) {
            char buf[48];
            std::snprintf(buf, sizeof buf, "%.7g", (double)fp[i]);
            dp[i] = std::strtod(buf, nullptr);
        }
        DevBuf p(in[0], n), g(in[1], n), m(in[2], n), v(in[3], n);
        launch_adam(p.p, g.p, m.p, v.p, (long long)n, dp[0], dp[1], dp[2], dp[3], (long long)dp[5], dp[4], st);
        VR_HIP(hipStreamSynchronize(st));
        p.download(out[0]); m.download(out[1]); v.download(out[2]);
    } else {
        throw Error(-2, "vr_debug_kernel: unknown kernel name: " + name);
    }
}

}  // namespace vr
