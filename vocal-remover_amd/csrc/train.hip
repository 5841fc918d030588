// Training executor: forward in train mode (model.hip) records a tape; this file walks it backwards
// (train.py:81-96: loss = L1(mask*X, y); (loss/accumulation_steps).backward(); optimizer.step()).
#include <cmath>
#include <cstring>

#include "model.h"

namespace vr {

ConvSrc make_src(const Tensor& t, bool up, int bcastH);

static inline int round_up32(int v) { return (v + 31) / 32 * 32; }

template <class F>
void Model::for_each_conv(F&& f) {
    for (int i = 0; i < 5; ++i) {
        BaseNetL& B = nets_[i];
        f(B.enc1);
        for (int j = 0; j < 4; ++j) { f(B.enc_a[j]); f(B.enc_b[j]); }
        f(B.aspp_pool); f(B.aspp_c2);
        for (int j = 0; j < 3; ++j) f(B.aspp_d[j]);
        f(B.aspp_bott);
        for (int j = 0; j < 4; ++j) f(B.dec[j]);
        f(B.lstm.proj); f(B.lstm.dense);
    }
    f(tail1); f(tail2);
}

void Model::ensure_train_state() {
    if (g_arena) return;
    VR_HIP(hipMalloc(&g_arena, p_floats * sizeof(float)));
    VR_HIP(hipMalloc(&m_arena, p_floats * sizeof(float)));
    VR_HIP(hipMalloc(&v_arena, p_floats * sizeof(float)));
    VR_HIP(hipMemset(g_arena, 0, p_floats * sizeof(float)));
    VR_HIP(hipMemset(m_arena, 0, p_floats * sizeof(float)));
    VR_HIP(hipMemset(v_arena, 0, p_floats * sizeof(float)));
    // flipped + transposed copies of the conv weights for the data-gradient launches
    size_t off = 0;
    std::vector<std::pair<Conv*, size_t>> slots;
    for_each_conv([&](Conv& L) {
        const size_t n = (size_t)L.Cout * L.KS * L.KS * round_up32(L.Cin);
        slots.push_back({&L, off});
        off += (n + 63) & ~size_t(63);
    });
    wt_floats = off;
    VR_HIP(hipMalloc(&wt_arena, wt_floats * sizeof(float)));
    VR_HIP(hipMemset(wt_arena, 0, wt_floats * sizeof(float)));
    std::vector<FlipDesc> descs;
    for (auto& sl : slots) {
        Conv& L = *sl.first;
        wt_of[L.w] = wt_arena + sl.second;
        descs.push_back(FlipDesc{L.w->dev, wt_arena + sl.second, L.Cin, L.Cout, L.KS * L.KS, round_up32(L.Cin), L.CoutPad});
    }
    {   // stride-2 3x3 convs: four parity-class weight tensors each (refreshed per step next to the flip)
        size_t tot = 0;
        for_each_conv([&](Conv& L) {
            if (L.KS == 3 && L.stride == 2 && L.dh == 1 && L.dw == 1) { s2_list.push_back(&L); tot += (size_t)4 * L.Cout * 9 * round_up32(L.Cin); }
        });
        if (tot) {
            VR_HIP(hipMalloc(&s2w_arena, tot * sizeof(float)));
            size_t o = 0;
            std::vector<S2WDesc> sd;
            for (Conv* L : s2_list) {
                s2w_of[L->w] = s2w_arena + o;
                sd.push_back(S2WDesc{L->w->dev, s2w_arena + o, L->Cin, L->Cout, L->CoutPad, round_up32(L->Cin)});
                s2w_max = std::max(s2w_max, (long long)4 * L->Cout * 9 * round_up32(L->Cin));
                o += (size_t)4 * L->Cout * 9 * round_up32(L->Cin);
            }
            VR_HIP(hipMalloc(reinterpret_cast<void**>(&d_s2w), sd.size() * sizeof(S2WDesc)));
            VR_HIP(hipMemcpy(d_s2w, sd.data(), sd.size() * sizeof(S2WDesc), hipMemcpyHostToDevice));
        }
    }
    n_flip = (int)descs.size();
    VR_HIP(hipMalloc(&d_flip, descs.size() * sizeof(FlipDesc)));
    VR_HIP(hipMemcpy(d_flip, descs.data(), descs.size() * sizeof(FlipDesc), hipMemcpyHostToDevice));
}

void Model::zero_grad_api() {
    DeviceGuard dev_guard(device);
    ensure_train_state();
    prof_memset_async(g_arena, 0, p_floats * sizeof(float), stream);
    VR_HIP(hipStreamSynchronize(stream));
}

void Model::grad_arena(float** ptr, int64_t* numel) {
    DeviceGuard dev_guard(device);
    ensure_train_state();
    *ptr = g_arena;
    *numel = (int64_t)p_floats;
}

void Model::set_dropout(int mode, unsigned long long seed, const float* masks, int B) {
    VR_CHECK(mode >= 0 && mode <= 2, -2, "dropout mode must be 0 (off), 1 (native RNG) or 2 (injected)");
    dropout_mode = mode;
    dropout_seed = seed;
    dropout_host.clear();
    if (mode == 2) {
        VR_CHECK(masks && B > 0, -2, "injected dropout needs masks [5][B][8*nout]");
        dropout_host.assign(masks, masks + (size_t)5 * B * 8 * nout);
    }
}

// nn.Dropout2d(0.1) keep-masks for one train-mode forward (lib/layers.py:90: the five ASPP outputs), [5][B][8*nout]
// holding 0 or 1/0.9.  Mode 1 draws them on the device from a counter-based generator keyed on (seed, number of
// train-mode forwards so far, element): every forward -- each micro-batch of an accumulation window too -- gets a
// fresh draw, as torch does, and nothing restarts when the optimizer is re-created.
__global__ void dropout_mask_kernel(float* __restrict__ m, long long n, unsigned long long seed, unsigned long long call) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (call + 1) + 0xD1B54A32D192ED03ull * (unsigned long long)(i + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;            // splitmix64 finalizer
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.f / 16777216.f);
    m[i] = (u >= 0.1f) ? (1.f / 0.9f) : 0.f;
}

void Model::prepare_dropout(int B) {
    dropout_dev = nullptr;
    train_calls += 1;
    if (dropout_mode == 0) return;
    const size_t n = (size_t)5 * B * 8 * nout;
    if (n > dropout_cap) {
        VR_HIP(hipStreamSynchronize(stream));
        if (dropout_buf) VR_HIP(hipFree(dropout_buf));
        dropout_buf = nullptr; dropout_cap = 0;
        VR_HIP(hipMalloc(&dropout_buf, n * sizeof(float)));
        dropout_cap = n;
    }
    if (dropout_mode == 2) {
        VR_CHECK(dropout_host.size() == n, -2, "injected dropout masks were given for a different batch size");
        VR_HIP(hipMemcpyAsync(dropout_buf, dropout_host.data(), n * sizeof(float), hipMemcpyHostToDevice, stream));
    } else {
        VR_LAUNCH(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dropout_buf, (long long)n,
                           dropout_seed, train_calls);
        VR_HIP(hipGetLastError());
    }
    dropout_dev = dropout_buf;
}

// BatchNorm backward of a conv record: G -> dz in place, d(gamma), d(beta) into the gradient arena.
void Model::bwd_bn_of(const Tensor& out, Conv& L) {
    BN* b = L.bn;
    BnBwdArgs a{};
    a.g = out.g; a.z = out.p; a.N = out.N; a.C = out.C; a.H = out.H; a.W = out.W;
    a.sN = out.sN; a.sC = out.sC; a.sH = out.sH;
    a.aff = b->affine; a.aff_bcast = b->bcast ? 1 : 0; a.slope = L.slope; a.post = out.post;
    a.gamma = b->w->dev; a.save_mean = b->save_mean; a.save_invstd = b->save_invstd;
    a.dgamma = dry ? nullptr : grad_of(b->w); a.dbeta = dry ? nullptr : grad_of(b->b);
    a.acc_grads = 1;
    a.coef = ws.allocf((size_t)out.C * 3);
    a.part = ws.allocf((size_t)bn_bwd_chunks(a) * out.C * 2);
    if (!dry && g_first(out.g))            // no consumer wrote a gradient into this tensor (does not happen in this net): it is zero
        prof_memset_async(out.g, 0, (size_t)out.N * out.C * out.H * out.W * sizeof(float), stream);
    if (!dry) launch_bn_bwd(a, stream);
}

void Model::bwd_conv(TapeRec& r) {
    Conv& L = *r.L;
    const Tensor& out = r.out;
    const int N = r.N;
    // 1. gradient at the raw conv output
    if (L.bn) bwd_bn_of(out, L);
    // (no BatchNorm: the only such convs -- LSTM input projection -- have identity activation: dz = G)
    // 2. bias gradient (Linear bias): channel sums of dz laid out [N][C][W]
    if (r.bias_param && !dry) launch_channel_sum(out.g, N, out.C, out.W, grad_of(r.bias_param), 1, stream);
    // 3. weight gradient
    ConvArgs f;
    build_fwd_args(L, r.srcs, N, r.batch_as_h, f);
    const ConvShape shp{L.KS, L.stride, L.dh, L.dw};
    {
        WgradArgs w{};
        w.in = f;
        w.dz = out.g;
        if (r.batch_as_h) { w.zN = 0; w.zC = out.sC; w.zH = out.sN; }
        else { w.zN = out.sN; w.zC = out.sC; w.zH = out.sH; }
        w.Cout = L.Cout; w.CoutPad = L.CoutPad;
        w.allow_wino = train_wino ? 1 : 0;
        w.bf16 = mfma_mode;                      // 1: bf16 MFMA operands (wgrad_wino / wgrad_gemm); 2: split-bf16 products (wgrad_x3.hip)
        w.part = ws.allocf(wgrad_scratch_floats(w, shp));
        if (!dry) {
            // The weight gradient is off the critical path (dz -> data gradient -> previous layer): it runs on the
            // side stream beside the data-gradient / BatchNorm-backward chain, so the element-wise passes and the
            // tails of one stream's kernels fill the other's bubbles.  backward() joins before returning.
            static const bool overlap = !getenv("VR_NO_WGRAD_OVERLAP");
            hipStream_t wst = stream;
            if (overlap && !serial && !profiling && side_stream) {
                VR_HIP(hipEventRecord(ev_fork, stream));          // dz (and every forward tensor) is ready here
                VR_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
                wst = side_stream;
                wgrad_on_side = true;
            }
            record_begin(0, 2.0 * N * (double)(r.batch_as_h ? 1 : f.Hout) * f.Wout * (double)L.Cout * L.Cin * L.KS * L.KS);
            record_note(conv_alg_bytes(L, f, N, r.batch_as_h), ("wgrad " + L.name).c_str());
            launch_wgrad(w, shp, grad_of(L.w), 1, wst);
            record_end();
        }
    }
    // 4. data gradient, split back to the sources of the virtual concat
    bool any = false;
    for (auto& sp : r.srcs) any = any || sp.t.g != nullptr;
    if (!any) return;
    ConvArgs d{};
    Tensor dzt = out;                       // dz as a plain source
    dzt.p = out.g; dzt.aff0 = dzt.aff1 = nullptr; dzt.slope = 1.f; dzt.post = nullptr; dzt.hsplit = 1 << 30;
    d.nsrc = 1;
    d.src[0] = make_src(dzt, false, 0);
    if (r.batch_as_h) { d.src[0].sH = d.src[0].sN; d.src[0].sN = 0; d.src[0].H = N; }
    d.src[0].zins = (L.stride == 2) ? 1 : 0;
    d.c1 = d.c2 = L.Cout; d.Cin = L.Cout;
    d.w = dry ? nullptr : wt_of[L.w];
    {
        auto it = winot_of.find(L.w);
        d.wino = (!dry && train_wino && it != winot_of.end()) ? it->second : nullptr;
        auto it6 = winot6_of.find(L.w);
        d.wino6 = (d.wino && mfma_mode == 2 && it6 != winot6_of.end()) ? it6->second : nullptr;
    }
    {
        auto itx = x3t_of.find(L.w);
        d.x3w = (!dry && train_wino && x3_mode() && itx != x3t_of.end()) ? itx->second : nullptr;
        auto itd = x3dt_of.find(L.w);                     // dilated 3x3 / 1x1 ASPP branches (conv_x3d.hip)
        if (!dry && train_wino && mfma_mode == 3 && itd != x3dt_of.end()) d.x3w = itd->second;
        if (!x3d_mode && f.Win == 16) d.x3w = nullptr;
    }
    d.bias = nullptr;
    d.bf16 = mfma_mode;
    d.Cout = L.Cin; d.CoutPad = round_up32(L.Cin);
    d.N = f.N; d.Hin = f.Hin; d.Win = f.Win; d.Hout = f.Hin; d.Wout = f.Win;
    d.pad_h = (L.KS == 1) ? 0 : L.dh; d.pad_w = (L.KS == 1) ? 0 : L.dw;
    struct Post { int kind; float* tmp; const SrcSpec* sp; };
    std::vector<Post> posts;
    int cbase = 0;
    for (size_t i = 0; i < r.srcs.size(); ++i) {
        const SrcSpec& sp = r.srcs[i];
        const Tensor& t = sp.t;
        ConvDst ds{};
        if (!t.g) {
            ds.p = nullptr;
        } else if (sp.up || sp.bcastH) {
            const size_t n = (size_t)N * t.C * f.Hin * f.Win;
            float* tmp = ws.allocf(n);
            ds = ConvDst{tmp, (long long)t.C * f.Hin * f.Win, (long long)f.Hin * f.Win, (long long)f.Win, 0};
            posts.push_back(Post{sp.up ? 1 : 2, tmp, &sp});
        } else if (r.batch_as_h) {
            ds = ConvDst{t.g, 0, t.sC, t.sN, 1};
        } else {
            ds = ConvDst{t.g, t.sN, t.sC, t.sH, (!dry && g_first(t.g)) ? 0 : 1};      // the tensor's first backward writer stores
        }
        d.dst[i] = ds;
        cbase += t.C;
        if (i == 0) d.d1 = cbase;
        if (i == 1) d.d2 = cbase;
    }
    if (r.srcs.size() == 1) d.d1 = d.d2 = 1 << 30;
    if (r.srcs.size() == 2) d.d2 = 1 << 30;
    if (dry) return;
    // Stride-2 3x3: four parity-class stride-1 convs over dz (9 tap evaluations) instead of one conv over the
    // zero-inserted gradient (36), when the LDS-DMA kernel covers the shape.
    {
        static const bool s2_classes = !getenv("VR_NO_S2_CLASSES");
        auto it = s2w_of.find(L.w);
        bool ok = s2_classes && L.stride == 2 && L.KS == 3 && it != s2w_of.end() && posts.empty() && !r.batch_as_h;
        if (ok) {
            ConvArgs c = d;
            c.src[0].zins = 0;
            c.Hin = out.H; c.Win = out.W;
            const ConvShape cshp{3, 1, 1, 1};
            const size_t cls_stride = (size_t)L.Cout * 9 * round_up32(L.Cin);
            // all four classes in one launch (one staging of dz and of the nine live (class, tap) weight slices per chunk)
            {
                ConvArgs k = c;
                k.Hout = (f.Hin + 1) / 2; k.Wout = (f.Win + 1) / 2;
                k.w = it->second; k.wino = nullptr; k.wino6 = nullptr; k.x3w = nullptr;
                k.s2_cls_stride = (long long)cls_stride; k.s2_H = f.Hin; k.s2_W = f.Win;
                if (s2d_fused_eligible(k)) {
                    record_begin(0, 2.0 * N * (double)f.Hout * f.Wout * (double)L.Cout * L.Cin * L.KS * L.KS);
                    record_note(conv_alg_bytes(L, f, N, false), ("dgrad " + L.name).c_str());
                    launch_s2d_fused(k, stream);
                    record_end();
                    return;
                }
            }
            const int masks[2] = {1 << 1, (1 << 1) | (1 << 2)};          // live taps along one axis: parity 0 / 1
            for (int cls = 0; cls < 4 && ok; ++cls) {
                const int ph = cls >> 1, pw = cls & 1;
                ConvArgs k = c;
                k.Hout = (f.Hin - ph + 1) / 2; k.Wout = (f.Win - pw + 1) / 2;
                k.w = it->second + cls * cls_stride;
                k.wino = nullptr; k.wino6 = nullptr; k.x3w = nullptr;
                int tm = 0;
                for (int th = 0; th < 3; ++th)
                    for (int tw = 0; tw < 3; ++tw)
                        if (((masks[ph] >> th) & 1) && ((masks[pw] >> tw) & 1)) tm |= 1 << (th * 3 + tw);
                k.tapmask = tm;
                for (int i = 0; i < 3; ++i) {
                    if (!k.dst[i].p) continue;
                    k.dst[i].p += (long long)ph * k.dst[i].sH + pw;
                    k.dst[i].sH *= 2;
                    k.dst[i].wshift = 1;
                }
                if (cls == 0 && !conv_dma_eligible(k, cshp)) { ok = false; break; }
                if (cls == 0) {
                    record_begin(0, 2.0 * N * (double)f.Hout * f.Wout * (double)L.Cout * L.Cin * L.KS * L.KS);
                    record_note(conv_alg_bytes(L, f, N, false), ("dgrad " + L.name).c_str());
                }
                launch_conv(k, cshp, stream);
            }
            if (ok) { record_end(); return; }
        }
    }
    const ConvShape dshp{L.KS, 1, L.dh, L.dw};
    record_begin(0, 2.0 * N * (double)(r.batch_as_h ? 1 : f.Hout) * f.Wout * (double)L.Cout * L.Cin * L.KS * L.KS);
    record_note(conv_alg_bytes(L, f, N, r.batch_as_h), ("dgrad " + L.name).c_str());
    launch_conv(d, dshp, stream);
    record_end();
    for (auto& p : posts) {
        const Tensor& t = p.sp->t;
        if (p.kind == 1) launch_upsample_bwd(p.tmp, N, t.C, t.H, t.W, t.g, t.sN, t.sC, t.sH, g_first(t.g) ? 0 : 1, stream);
        else launch_sum_h(p.tmp, N, t.C, f.Hin, f.Win, t.g, stream);
    }
}

void Model::backward() {
    struct Join {                                      // the side stream's weight gradients must land before Adam
        Model* m;
        ~Join() {
            if (!m->wgrad_on_side) return;
            m->wgrad_on_side = false;
            hipEventRecord(m->ev_join, m->side_stream);
            hipStreamWaitEvent(m->stream, m->ev_join, 0);
        }
    } join{this};
    // The 107 wgrad_reduce launches of a step (~10 us each, launch-bound) become ONE launch behind the last weight gradient: launch_wgrad
    // only records what it would have summed (vr_common.h); flush_wgrad_sums() at the end of the tape launches the sum.
    static const bool defer_on = !getenv("VR_NO_WGRAD_DEFER");
    struct Unsink { ~Unsink() { wgrad_defer_to(nullptr); } } unsink;      // (also when a launch throws)
    wred_host.clear();
    if (defer_on && !dry) wgrad_defer_to(&wred_host);
    // Stages 1-2: the high-band records (chain 1) are independent of the low-band ones; in the reversed tape
    // they come first after stage 3, so they are enqueued on a second stream and the low chain follows on the main one.
    static const bool bwd_fork = !getenv("VR_NO_BWD_FORK");
    hipStream_t main_stream = stream;
    hipStream_t hi_stream = (bwd_fork && !serial && !dry && !profiling && !lanes.empty()) ? lanes[0].main : nullptr;
    int cur_chain = 0;
    bool hi_used = false;
    struct Restore {
        Model* m; hipStream_t s; hipStream_t hi; hipEvent_t ev; bool* used;
        ~Restore() {
            m->stream = s;
            if (*used) { hipEventRecord(ev, hi); hipStreamWaitEvent(s, ev, 0); }
        }
    } restore{this, main_stream, hi_stream, hi_stream ? lanes[0].done : nullptr, &hi_used};
    for (size_t k = tape.size(); k-- > 0;) {
        TapeRec& r = tape[k];
        if (hi_stream && r.chain != cur_chain) {
            if (r.chain == 1) {                    // entering the high chain: everything so far (stage 3) must be done
                VR_HIP(hipEventRecord(lanes[0].start, main_stream));
                VR_HIP(hipStreamWaitEvent(hi_stream, lanes[0].start, 0));
                stream = hi_stream;
                hi_used = true;
            } else {
                stream = main_stream;
            }
            cur_chain = r.chain;
        }
        switch (r.kind) {
        case TK_CONV:
            bwd_conv(r);
            break;
        case TK_AVGPOOL: {
            const Tensor& x5 = r.srcs[0].t;
            if (!dry && x5.g) launch_avgpool_bwd(r.out.g, x5.g, r.N, x5.C, x5.H, x5.W, x5.sN, x5.sC, x5.sH, g_first(x5.g) ? 0 : 1, stream);
            break;
        }
        case TK_SQUEEZE: {
            // 1x1 conv 2c->1 + BatchNorm(1) + ReLU (lib/layers.py:112): out is the 1-channel image z
            LSTMMod& M = *r.M;
            bwd_bn_of(r.out, M.squeeze);
            const Tensor& h = r.srcs[0].t;
            float* part = ws.allocf((size_t)thin_wgrad_blocks(h) * h.C);
            if (!dry) {
                launch_thin_wgrad(h, 1, r.out.g, part, grad_of(M.squeeze.w), 1, stream);
                launch_thin_dgrad(h, 1, M.squeeze.w->dev, r.out.g, h.g, g_first(h.g) ? 0 : 1, stream);
            }
            break;
        }
        case TK_LSTM: {
            LSTMMod& M = *r.M;
            const int H = M.hid, G = 4 * H, T = r.out.W, N = r.N;
            float* wpart = ws.allocf(lstm_whh_grad_scratch_floats(N, H));
            if (!dry) {
                launch_bilstm_bwd(r.out.g, r.save, M.whh_f->dev, M.whh_r->dev, r.aux.g, N, T, H, stream);
                launch_lstm_whh_grad(r.aux.g, r.out.p, grad_of(M.whh_f), grad_of(M.whh_r), N, T, H, 1, wpart, stream);
                // b_ih and b_hh enter the gates as a sum: both receive the channel sums of dgx
                float* tmp = ws.allocf((size_t)2 * G);
                launch_channel_sum(r.aux.g, N, 2 * G, T, tmp, 0, stream);
                launch_add(grad_of(M.b_ih_f), tmp, grad_of(M.b_ih_f), G, stream);
                launch_add(grad_of(M.b_hh_f), tmp, grad_of(M.b_hh_f), G, stream);
                launch_add(grad_of(M.b_ih_r), tmp + G, grad_of(M.b_ih_r), G, stream);
                launch_add(grad_of(M.b_hh_r), tmp + G, grad_of(M.b_hh_r), G, stream);
            } else {
                ws.allocf((size_t)2 * G);
            }
            break;
        }
        default:
            break;
        }
    }
    wgrad_defer_to(nullptr);
    flush_wgrad_sums();
}

// ONE launch for every deferred weight-gradient slab sum of this backward pass, behind the last weight gradient on its stream
void Model::flush_wgrad_sums() {
    if (wred_host.empty()) return;
    long long blocks = 0;
    const int vec = wgrad_reduce_vec(wred_host.data(), (int)wred_host.size());
    for (WgReduceDesc& d : wred_host) { d.blk0 = blocks; blocks += (d.n / vec + 63) / 64; }
    const size_t bytes = wred_host.size() * sizeof(WgReduceDesc);
    hipStream_t st = wgrad_on_side ? side_stream : stream;
    if (wred_cap < wred_host.size()) {
        if (wred_dev) { VR_HIP(hipDeviceSynchronize()); hipFree(wred_dev); }
        wred_dev = nullptr; wred_cap = 0; wred_sent.clear();
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&wred_dev), 2 * bytes));
        wred_cap = 2 * wred_host.size();
    }
    // the table is the same in every step of a plan (bump-allocated slabs, fixed gradient arena): uploaded only when it changed
    if (wred_sent.size() != wred_host.size() || memcmp(wred_sent.data(), wred_host.data(), bytes) != 0) {
        VR_HIP(hipStreamSynchronize(st));                         // (the previous step's launch may still be reading the old table)
        VR_HIP(hipMemcpy(wred_dev, wred_host.data(), bytes, hipMemcpyHostToDevice));
        wred_sent = wred_host;
    }
    launch_wgrad_reduce_batched(wred_dev, (int)wred_host.size(), blocks, vec, st);
    wred_host.clear();
}

void Model::train_fwd_bwd_api(const float* X, const float* Y, bool on_dev, int B, int T, int accumulation_steps,
                              float* loss_out, float* mask_out, bool mask_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(training, -2, "train step needs train mode: call vr_set_mode(h, 1) (model.train(), train.py:69)");
    graph_valid = false;
    VR_CHECK(B > 0 && accumulation_steps > 0, -2, "batch and accumulation_steps must be positive");
    VR_CHECK(T > 0 && T % 16 == 0, -5, "h1_shape[3] must be greater than h2_shape[3] (frames must be a multiple of 16)");
    ensure_train_state();
    // flipped/transposed weights for the data gradients + Winograd-domain copies of both, once per step
    // (before the planning dry run: the kernel choice, hence the partial-statistics layout, depends on them)
    launch_flip_transpose(d_flip, n_flip, stream);
    launch_s2_class_weights(d_s2w, (int)s2_list.size(), s2w_max, stream);
    refresh_wino(true);
    const size_t io_floats = (size_t)B * 2 * output_bin * T;
    const int Hm = max_bin;
    // ---- plan: dry run of forward + backward sizes both arenas ----------------------------------------
    auto run_all = [&](const float* xd, const float* yd, float* maskd, float* lossd) {
        tape.clear();
        g_fresh.clear(); gs_zero.clear();
        Tensor xt;
        xt.p = const_cast<float*>(xd); xt.N = B; xt.C = 2; xt.H = Hm; xt.W = T;
        xt.sH = T; xt.sC = (long long)output_bin * T; xt.sN = 2 * xt.sC; xt.slope = 1.f;
        Tensor f3 = run_net(xt);
        // head + loss (train.py:81,89) and its backward into f3.g / out.weight
        float* dlogit = ws.allocf((size_t)B * 2 * Hm * T);
        float* lpart = ws.allocf((size_t)head_loss_blocks(f3));
        float* wpart = ws.allocf((size_t)thin_wgrad_blocks(f3) * 2 * f3.C);
        if (!dry) {
            if (gs_clear_pending) { VR_HIP(hipStreamWaitEvent(stream, lanes[0].join, 0)); gs_clear_pending = false; }
            const double ntot = (double)B * 2 * output_bin * T;
            launch_head_loss(f3, out_w->dev, xd, yd, output_bin, (float)(1.0 / (ntot * accumulation_steps)), dlogit, maskd,
                             lpart, lossd, (float)(1.0 / ntot), stream);
            launch_thin_wgrad(f3, 2, dlogit, wpart, grad_of(out_w), 1, stream);
            launch_thin_dgrad(f3, 2, out_w->dev, dlogit, f3.g, g_first(f3.g) ? 0 : 1, stream);
        }
        backward();
    };
    {
        Arena sws = ws, sgs = gs;
        ws.dry = gs.dry = true; ws.base = gs.base = nullptr; ws.off = gs.off = 0; ws.peak = gs.peak = 0;
        dry = true;
        try { run_all(reinterpret_cast<float*>(uintptr_t(256)), nullptr, nullptr, nullptr); }
        catch (...) { dry = false; ws = sws; gs = sgs; tape.clear(); throw; }
        dry = false;
        gs_zero_plan = gs_zero;                      // which buffers of gs need a zero fill (the real pass allocates in the same order)
        const size_t need_ws = ws.peak + (3 * io_floats + 64) * sizeof(float) + 8192, need_gs = gs.peak + 4096;
        ws = sws; gs = sgs;
        ensure_ws(need_ws);
        if (need_gs > gs.cap) {
            VR_HIP(hipStreamSynchronize(stream));
            if (gs.base) VR_HIP(hipFree(gs.base));
            gs.base = nullptr; gs.cap = 0;
            VR_HIP(hipMalloc(reinterpret_cast<void**>(&gs.base), need_gs + (need_gs >> 4)));
            gs.cap = need_gs + (need_gs >> 4);
        }
        ws.reset(); gs.reset();
        // the activation-gradient arena is only touched from the loss kernels on: clear it beside the forward pass
        if (!lanes.empty()) {
            VR_HIP(hipEventRecord(lanes[0].fork, stream));            // (previous step's consumers of gs are done)
            VR_HIP(hipStreamWaitEvent(lanes[0].main, lanes[0].fork, 0));
            clear_gs_zero_ranges(lanes[0].main);
            VR_HIP(hipEventRecord(lanes[0].join, lanes[0].main));
            gs_clear_pending = true;
        } else {
            clear_gs_zero_ranges(stream);
        }
    }
    prepare_dropout(B);
    // ---- inputs ---------------------------------------------------------------------------------------------------
    const float *xd = X, *yd = Y;
    if (!on_dev) {
        float* tx = ws.allocf(io_floats);
        float* ty = ws.allocf(io_floats);
        VR_HIP(hipMemcpyAsync(tx, X, io_floats * sizeof(float), hipMemcpyHostToDevice, stream));
        VR_HIP(hipMemcpyAsync(ty, Y, io_floats * sizeof(float), hipMemcpyHostToDevice, stream));
        xd = tx; yd = ty;
    }
    float* maskd = nullptr;
    if (mask_out) maskd = mask_on_dev ? mask_out : ws.allocf(io_floats);
    float* lossd = ws.allocf(16);

    run_all(xd, yd, maskd, lossd);
    float loss_h = 0.f;
    VR_HIP(hipMemcpyAsync(&loss_h, lossd, sizeof(float), hipMemcpyDeviceToHost, stream));
    if (mask_out && !mask_on_dev) VR_HIP(hipMemcpyAsync(mask_out, maskd, io_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
    if (loss_out) *loss_out = loss_h;
    tape.clear();
    affine_dirty = true;
    dropout_dev = nullptr;
}

// ---- the autograd split of the same step: `mask = model(X)` keeps the graph, `loss.backward()` comes later with
// dLoss/dmask (train.py:81,92 as two calls, so that the reference's own loss expression and torch's own optimizer can
// sit in between).  The graph lives in the workspace arenas: any other call on the handle invalidates it. ------------
void Model::forward_train_api(const float* X, bool on_dev, int B, int T, float* mask_out, bool mask_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(training, -2, "forward with a graph needs train mode: call vr_set_mode(h, 1) (model.train(), train.py:69)");
    VR_CHECK(B > 0, -2, "batch must be positive");
    VR_CHECK(T > 0 && T % 16 == 0, -5, "h1_shape[3] must be greater than h2_shape[3] (frames must be a multiple of 16)");
    VR_CHECK(mask_out != nullptr, -2, "null argument");
    graph_valid = false;
    ensure_train_state();
    launch_flip_transpose(d_flip, n_flip, stream);
    launch_s2_class_weights(d_s2w, (int)s2_list.size(), s2w_max, stream);
    refresh_wino(true);
    const size_t io_floats = (size_t)B * 2 * output_bin * T;
    const int Hm = max_bin;
    auto fwd = [&](const float* xd) {
        tape.clear();
        g_fresh.clear(); gs_zero.clear();
        Tensor xt;
        xt.p = const_cast<float*>(xd); xt.N = B; xt.C = 2; xt.H = Hm; xt.W = T;
        xt.sH = T; xt.sC = (long long)output_bin * T; xt.sN = 2 * xt.sC; xt.slope = 1.f;
        return run_net(xt);
    };
    auto bwd_scratch = [&](const Tensor& f3) {       // the allocations backward_api makes, in its order
        ws.allocf((size_t)B * 2 * Hm * T);
        ws.allocf((size_t)thin_wgrad_blocks(f3) * 2 * f3.C);
    };
    {   // plan: dry run of forward + backward sizes both arenas
        Arena sws = ws, sgs = gs;
        ws.dry = gs.dry = true; ws.base = gs.base = nullptr; ws.off = gs.off = 0; ws.peak = gs.peak = 0;
        dry = true;
        try {
            ws.allocf(3 * io_floats + 64);
            Tensor f3 = fwd(reinterpret_cast<float*>(uintptr_t(256)));
            bwd_scratch(f3);
            backward();
        } catch (...) { dry = false; ws = sws; gs = sgs; tape.clear(); throw; }
        dry = false;
        gs_zero_plan = gs_zero;
        const size_t need_ws = ws.peak + 8192, need_gs = gs.peak + 4096;
        ws = sws; gs = sgs;
        ensure_ws(need_ws);
        if (need_gs > gs.cap) {
            VR_HIP(hipStreamSynchronize(stream));
            if (gs.base) VR_HIP(hipFree(gs.base));
            gs.base = nullptr; gs.cap = 0;
            VR_HIP(hipMalloc(reinterpret_cast<void**>(&gs.base), need_gs + (need_gs >> 4)));
            gs.cap = need_gs + (need_gs >> 4);
        }
        ws.reset(); gs.reset();
        if (gs_clear_pending) { VR_HIP(hipStreamWaitEvent(stream, lanes[0].join, 0)); gs_clear_pending = false; }
        clear_gs_zero_ranges(stream);
    }
    prepare_dropout(B);
    float* xd = ws.allocf(io_floats);
    graph_mask = ws.allocf(io_floats);
    ws.allocf(io_floats + 64);                                     // (keeps the dry run's offsets)
    VR_HIP(hipMemcpyAsync(xd, X, io_floats * sizeof(float), on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    graph_f3 = fwd(xd);
    HeadDst d{};
    d.p = graph_mask; d.dH = T; d.dC = (long long)output_bin * T; d.dN = 2 * d.dC;
    d.w_lo = 0; d.w_hi = T; d.pad_rows = output_bin - max_bin;
    launch_head_sigmoid(graph_f3, out_w->dev, d, stream);
    VR_HIP(hipMemcpyAsync(mask_out, graph_mask, io_floats * sizeof(float), mask_on_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
    graph_valid = true; graph_B = B; graph_T = T; ++graph_gen;
    affine_dirty = true;
}

void Model::backward_api(const float* dmask, bool on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(graph_valid, -2, "vr_backward: no graph -- call vr_forward_train first (another call on the handle in between frees it)");
    VR_CHECK(dmask != nullptr, -2, "null argument");
    graph_valid = false;
    const int B = graph_B, T = graph_T, Hm = max_bin;
    const size_t io_floats = (size_t)B * 2 * output_bin * T;
    const Tensor& f3 = graph_f3;
    const float* dm = dmask;
    if (!on_dev) {                                                 // the input copy of the forward is dead by now: reuse nothing, stage after the tape
        float* tmp = nullptr;
        VR_HIP(hipMalloc(&tmp, io_floats * sizeof(float)));
        struct Free { float* p; hipStream_t s; ~Free() { hipStreamSynchronize(s); hipFree(p); } } fr{tmp, stream};
        VR_HIP(hipMemcpyAsync(tmp, dmask, io_floats * sizeof(float), hipMemcpyHostToDevice, stream));
        float* dlogit = ws.allocf((size_t)B * 2 * Hm * T);
        float* wpart = ws.allocf((size_t)thin_wgrad_blocks(f3) * 2 * f3.C);
        launch_head_bwd(tmp, graph_mask, B, Hm, T, output_bin, dlogit, stream);
        launch_thin_wgrad(f3, 2, dlogit, wpart, grad_of(out_w), 1, stream);
        launch_thin_dgrad(f3, 2, out_w->dev, dlogit, f3.g, g_first(f3.g) ? 0 : 1, stream);
        backward();
        VR_HIP(hipStreamSynchronize(stream));
    } else {
        float* dlogit = ws.allocf((size_t)B * 2 * Hm * T);
        float* wpart = ws.allocf((size_t)thin_wgrad_blocks(f3) * 2 * f3.C);
        launch_head_bwd(dm, graph_mask, B, Hm, T, output_bin, dlogit, stream);
        launch_thin_wgrad(f3, 2, dlogit, wpart, grad_of(out_w), 1, stream);
        launch_thin_dgrad(f3, 2, out_w->dev, dlogit, f3.g, g_first(f3.g) ? 0 : 1, stream);
        backward();
        VR_HIP(hipStreamSynchronize(stream));
    }
    tape.clear();
    dropout_dev = nullptr;
}

// Training input pipeline (lib/dataset.py:105-120 after the random draws and the file reads), see augment.hip.
void Model::augment_api(const float* Xc, const float* yc, const float* Xi, const float* yi, const void* desc, const float* rw,
                        int B, int T, int bins, bool in_on_dev, float* Xmag, float* ymag, bool out_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(B > 0 && T > 0 && bins > 0, -2, "augment: empty batch");
    const size_t crop_b = (size_t)B * T * 2 * bins * sizeof(float2), out_b = (size_t)B * 2 * bins * T * sizeof(float);
    const size_t desc_b = (size_t)B * sizeof(AugDesc), rw_b = (size_t)bins * sizeof(float);
    auto up256 = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t need = up256(desc_b) + up256(rw_b);
    if (!in_on_dev) need += 4 * up256(crop_b);
    if (!out_on_dev) need += 2 * up256(out_b);
    if (need > aug_cap) {
        VR_HIP(hipStreamSynchronize(stream));
        if (aug_buf) VR_HIP(hipFree(aug_buf));
        aug_buf = nullptr; aug_cap = 0;
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&aug_buf), need + (need >> 3)));
        aug_cap = need + (need >> 3);
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = aug_buf + off; off += up256(bytes); return p; };
    AugDesc* dd = reinterpret_cast<AugDesc*>(take(desc_b));
    float* drw = reinterpret_cast<float*>(take(rw_b));
    VR_HIP(hipMemcpyAsync(dd, desc, desc_b, hipMemcpyHostToDevice, stream));
    if (rw) VR_HIP(hipMemcpyAsync(drw, rw, rw_b, hipMemcpyHostToDevice, stream));
    const float* src[4] = {Xc, yc, Xi, yi};
    const float2* dev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        if (!src[i]) continue;
        if (in_on_dev) { dev[i] = reinterpret_cast<const float2*>(src[i]); continue; }
        char* p = take(crop_b);
        VR_HIP(hipMemcpyAsync(p, src[i], crop_b, hipMemcpyHostToDevice, stream));
        dev[i] = reinterpret_cast<const float2*>(p);
    }
    float* ox = out_on_dev ? Xmag : reinterpret_cast<float*>(take(out_b));
    float* oy = out_on_dev ? ymag : reinterpret_cast<float*>(take(out_b));
    launch_augment(dev[0], dev[1], dev[2] ? dev[2] : dev[0], dev[3] ? dev[3] : dev[1], dd, rw ? drw : nullptr, B, T, bins, ox, oy,
                   stream);
    if (!out_on_dev) {
        VR_HIP(hipMemcpyAsync(Xmag, ox, out_b, hipMemcpyDeviceToHost, stream));
        VR_HIP(hipMemcpyAsync(ymag, oy, out_b, hipMemcpyDeviceToHost, stream));
    }
    VR_HIP(hipStreamSynchronize(stream));
}

void Model::reset_adam_state() {
    DeviceGuard dev_guard(device);
    adam_step = 0;
    if (m_arena) {
        prof_memset_async(m_arena, 0, p_floats * sizeof(float), stream);
        prof_memset_async(v_arena, 0, p_floats * sizeof(float), stream);
        VR_HIP(hipStreamSynchronize(stream));
    }
}

void Model::adam_step_api(double lr, double b1, double b2, double eps, double grad_scale) {
    DeviceGuard dev_guard(device);
    ensure_train_state();
    adam_step += 1;
    launch_adam(p_arena, g_arena, m_arena, v_arena, (long long)p_floats, lr, b1, b2, eps, adam_step, grad_scale, stream);
    VR_HIP(hipStreamSynchronize(stream));
    affine_dirty = true;
}

void Model::adam_state(float* m_host, float* v_host, int64_t numel, int64_t* step, bool set) {
    DeviceGuard dev_guard(device);
    ensure_train_state();
    VR_CHECK(numel == (int64_t)p_floats, -2, "adam state: element count must equal vr_grad_arena's");
    VR_HIP(hipStreamSynchronize(stream));
    if (set) {
        VR_HIP(hipMemcpy(m_arena, m_host, p_floats * sizeof(float), hipMemcpyHostToDevice));
        VR_HIP(hipMemcpy(v_arena, v_host, p_floats * sizeof(float), hipMemcpyHostToDevice));
        adam_step = *step;
    } else {
        VR_HIP(hipMemcpy(m_host, m_arena, p_floats * sizeof(float), hipMemcpyDeviceToHost));
        VR_HIP(hipMemcpy(v_host, v_arena, p_floats * sizeof(float), hipMemcpyDeviceToHost));
        *step = adam_step;
    }
}

void Model::get_grad(const std::string& key, float* host, int64_t cap_bytes) {
    auto it = by_key.find(key);
    VR_CHECK(it != by_key.end(), -2, "unknown parameter key: " + key);
    Param& p = *it->second;
    VR_CHECK(p.trainable, -2, key + " is a buffer, it has no gradient");
    DeviceGuard dev_guard(device);
    ensure_train_state();
    VR_HIP(hipStreamSynchronize(stream));
    VR_CHECK((size_t)cap_bytes >= p.numel * sizeof(float), -2, "buffer too small for " + key);
    const float* g = g_arena + (p.dev - p_arena);
    if (p.kind == PK_CONV) {
        std::vector<float> tmp((size_t)p.Cin * p.KK * p.CoutPad);
        VR_HIP(hipMemcpy(tmp.data(), g, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci)
                for (int k = 0; k < p.KK; ++k)
                    host[((size_t)co * p.Cin + ci) * p.KK + k] = tmp[((size_t)ci * p.KK + k) * p.CoutPad + co];
    } else if (p.kind == PK_LSTM_IH) {
        std::vector<float> tmp((size_t)p.Cin * p.Cout);
        VR_HIP(hipMemcpy2D(tmp.data(), (size_t)p.Cout * sizeof(float), g + p.co_off, (size_t)p.CoutPad * sizeof(float),
                           (size_t)p.Cout * sizeof(float), (size_t)p.Cin, hipMemcpyDeviceToHost));
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci) host[(size_t)co * p.Cin + ci] = tmp[(size_t)ci * p.Cout + co];
    } else {
        VR_HIP(hipMemcpy(host, g, p.numel * sizeof(float), hipMemcpyDeviceToHost));
    }
}

}  // namespace vr

// =====================================================================================================
// unit-test hook: backward of ONE conv (no BatchNorm) through the MFMA dgrad / wgrad kernels
// =====================================================================================================
namespace vr {

void Model::debug_conv_bwd(const float* x, int N, int Cin, int H, int W, const float* w_oihw, int Cout, int KS, int stride,
                           int dh, int dw, int up, const float* aff, float slope, const float* dz, float* dx_out,
                           float* dw_out) {
    DeviceGuard dev_guard(device);
    ensure_train_state();
    const int KK = KS * KS, CoutPad = (Cout + 31) / 32 * 32, CinPad = round_up32(Cin);
    Conv L;
    L.name = "debug"; L.Cin = Cin; L.Cout = Cout; L.CoutPad = CoutPad; L.KS = KS; L.stride = stride; L.dh = dh; L.dw = dw;
    L.pad_h = KS == 1 ? 0 : dh; L.pad_w = KS == 1 ? 0 : dw; L.bn = nullptr; L.slope = 1.f;
    Param P;
    P.kind = PK_CONV; P.Cin = Cin; P.Cout = Cout; P.KK = KK; P.CoutPad = CoutPad;
    L.w = &P;
    const int Hin = up ? 2 * H : H, Win = up ? 2 * W : W;
    const int Hout = (Hin + 2 * L.pad_h - dh * (KS - 1) - 1) / stride + 1, Wout = (Win + 2 * L.pad_w - dw * (KS - 1) - 1) / stride + 1;
    std::vector<float> wk((size_t)Cin * KK * CoutPad, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < KK; ++k) wk[((size_t)ci * KK + k) * CoutPad + co] = w_oihw[((size_t)co * Cin + ci) * KK + k];
    const size_t xin = (size_t)N * Cin * H * W, xout = (size_t)N * Cout * Hout * Wout;
    float *dx, *dgx, *dwk, *dwt, *dgw, *dzd, *daff = nullptr;
    VR_HIP(hipMalloc(&dx, xin * 4)); VR_HIP(hipMalloc(&dgx, xin * 4)); VR_HIP(hipMalloc(&dwk, wk.size() * 4));
    VR_HIP(hipMalloc(&dwt, (size_t)Cout * KK * CinPad * 4)); VR_HIP(hipMalloc(&dgw, wk.size() * 4)); VR_HIP(hipMalloc(&dzd, xout * 4));
    VR_HIP(hipMemcpy(dx, x, xin * 4, hipMemcpyHostToDevice));
    VR_HIP(hipMemset(dgx, 0, xin * 4)); VR_HIP(hipMemset(dgw, 0, wk.size() * 4)); VR_HIP(hipMemset(dwt, 0, (size_t)Cout * KK * CinPad * 4));
    VR_HIP(hipMemcpy(dwk, wk.data(), wk.size() * 4, hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dzd, dz, xout * 4, hipMemcpyHostToDevice));
    if (aff) { VR_HIP(hipMalloc(&daff, (size_t)Cin * 8)); VR_HIP(hipMemcpy(daff, aff, (size_t)Cin * 8, hipMemcpyHostToDevice)); }
    P.dev = dwk; P.grad_override = dgw;
    wt_of[&P] = dwt;
    FlipDesc fd{dwk, dwt, Cin, Cout, KK, CinPad, CoutPad};
    FlipDesc* dfd;
    VR_HIP(hipMalloc(&dfd, sizeof(FlipDesc)));
    VR_HIP(hipMemcpy(dfd, &fd, sizeof(FlipDesc), hipMemcpyHostToDevice));
    launch_flip_transpose(dfd, 1, stream);
    // 3x3 stride-1: the data gradient takes the same kernels as in a train step (Winograd / split-bf16 direct over the flipped weights)
    float* dwinot = nullptr;
    void* dx3t = nullptr;
    if (KS == 3 && stride == 1 && dh == 1 && dw == 1 && train_wino) {
        VR_HIP(hipMalloc(&dwinot, (size_t)Cout * 16 * CinPad * sizeof(float)));
        launch_wino_weights(dwt, dwinot, Cout, CinPad, stream);
        winot_of[&P] = dwinot;
        if (mfma_mode == 2) {
            VR_HIP(hipMalloc(&dx3t, x3_weights_bytes(Cout, 9, CinPad)));
            launch_x3_weights(dwt, dx3t, Cout, 9, CinPad, stream);
            x3t_of[&P] = dx3t;
        }
        if (mfma_mode == 3) {
            VR_HIP(hipMalloc(&dx3t, x3_weights_bytes(Cout, 9, CinPad)));
            launch_x3h_weights(dwt, dx3t, Cout, 9, CinPad, stream);
            x3t_of[&P] = dx3t;
        }
    }
    if (stride == 1 && (KS == 1 || dh > 1) && train_wino && mfma_mode == 3) {      // conv_x3d.hip's layers (16-column images)
        VR_HIP(hipMalloc(&dx3t, x3_weights_bytes(Cout, KK, CinPad)));
        launch_x3h_weights(dwt, dx3t, Cout, KK, CinPad, stream);
        x3dt_of[&P] = dx3t;
    }
    float* ds2w = nullptr;                    // stride-2 3x3: also exercise the parity-class data gradient
    S2WDesc* ds2d = nullptr;
    if (KS == 3 && stride == 2 && dh == 1 && dw == 1) {
        VR_HIP(hipMalloc(&ds2w, (size_t)4 * Cout * 9 * CinPad * sizeof(float)));
        const S2WDesc sd{dwk, ds2w, Cin, Cout, CoutPad, CinPad};
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&ds2d), sizeof(S2WDesc)));
        VR_HIP(hipMemcpy(ds2d, &sd, sizeof(S2WDesc), hipMemcpyHostToDevice));
        launch_s2_class_weights(ds2d, 1, 4LL * Cout * 9 * CinPad, stream);
        s2w_of[&P] = ds2w;
    }
    Tensor t;
    t.p = dx; t.g = dgx; t.N = N; t.C = Cin; t.H = H; t.W = W; t.sH = W; t.sC = (long long)H * W; t.sN = t.sC * Cin;
    t.aff0 = daff; t.slope = slope;
    TapeRec r;
    r.kind = TK_CONV; r.L = &L; r.N = N;
    SrcSpec sp{t}; sp.up = up != 0;
    r.srcs = {sp};
    r.out.p = nullptr; r.out.g = dzd; r.out.N = N; r.out.C = Cout; r.out.H = Hout; r.out.W = Wout;
    r.out.sH = Wout; r.out.sC = (long long)Hout * Wout; r.out.sN = r.out.sC * Cout; r.out.slope = 1.f;
    {   // size the workspace with a dry pass
        Arena sws = ws;
        ws.dry = true; ws.base = nullptr; ws.off = 0; ws.peak = 0; dry = true;
        try { bwd_conv(r); } catch (...) { dry = false; ws = sws; throw; }
        dry = false;
        const size_t need = ws.peak + 4096;
        ws = sws;
        ensure_ws(need);
        ws.reset();
    }
    bwd_conv(r);
    VR_HIP(hipStreamSynchronize(stream));
    if (wgrad_on_side) { VR_HIP(hipStreamSynchronize(side_stream)); wgrad_on_side = false; }   // the weight gradient ran there
    VR_HIP(hipMemcpy(dx_out, dgx, xin * 4, hipMemcpyDeviceToHost));
    std::vector<float> gk(wk.size());
    VR_HIP(hipMemcpy(gk.data(), dgw, gk.size() * 4, hipMemcpyDeviceToHost));
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < KK; ++k) dw_out[((size_t)co * Cin + ci) * KK + k] = gk[((size_t)ci * KK + k) * CoutPad + co];
    wt_of.erase(&P);
    s2w_of.erase(&P);
    winot_of.erase(&P);
    x3t_of.erase(&P);
    x3dt_of.erase(&P);
    hipFree(ds2w); hipFree(ds2d); hipFree(dwinot); hipFree(dx3t);
    hipFree(dx); hipFree(dgx); hipFree(dwk); hipFree(dwt); hipFree(dgw); hipFree(dzd); hipFree(daff); hipFree(dfd);
}

}  // namespace vr
