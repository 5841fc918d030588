// Host side of libvr_mi355.so: see model.h.  Topology follows lib/nets.py:8-141 and
// lib/layers.py:8-133 of the reference; parameter keys are the reference's state_dict keys.
#include "model.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace vr {

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// =====================================================================================================
// construction
// =====================================================================================================
Param* Model::add_param(const std::string& key, std::vector<int64_t> shape, ParamKind kind, bool trainable) {
    params.emplace_back();
    Param* p = &params.back();
    p->key = key;
    p->shape = std::move(shape);
    p->kind = kind;
    p->trainable = trainable;
    p->numel = 1;
    for (auto d : p->shape) p->numel *= (size_t)d;
    p->dev_numel = p->numel;
    if (kind == PK_CONV) {
        p->Cout = (int)p->shape[0];
        p->Cin = (int)p->shape[1];
        p->KK = p->shape.size() == 4 ? (int)(p->shape[2] * p->shape[3]) : 1;
        p->CoutPad = round_up(p->Cout, 32);
        p->dev_numel = (size_t)p->Cin * p->KK * p->CoutPad;
    }
    if (kind == PK_NBT) p->dev_numel = 0;
    by_key[key] = p;
    return p;
}

BN* Model::add_bn(const std::string& prefix, int C, int bcast) {
    bns.emplace_back();
    BN* b = &bns.back();
    b->C = C;
    b->bcast = bcast;
    b->w = add_param(prefix + ".weight", {C}, PK_PLAIN, true);
    b->b = add_param(prefix + ".bias", {C}, PK_PLAIN, true);
    b->rm = add_param(prefix + ".running_mean", {C}, PK_BUFFER, false);
    b->rv = add_param(prefix + ".running_var", {C}, PK_BUFFER, false);
    b->nbt = add_param(prefix + ".num_batches_tracked", {}, PK_NBT, false);
    bn_list.push_back(b);
    return b;
}

// layers.Conv2DBNActiv (lib/layers.py:8-26)
void Model::build_cba(Conv& L, const std::string& prefix, int nin, int nout_, int ks, int stride, int pad_h, int pad_w,
                      int dh, int dw, float slope) {
    L.name = prefix;
    L.Cin = nin; L.Cout = nout_; L.CoutPad = round_up(nout_, 32);
    L.KS = ks; L.stride = stride; L.pad_h = pad_h; L.pad_w = pad_w; L.dh = dh; L.dw = dw; L.slope = slope;
    L.w = add_param(prefix + ".conv.0.weight", {nout_, nin, ks, ks}, PK_CONV, true);
    L.bn = add_bn(prefix + ".conv.1", nout_, 0);
    if (ks == 3 && stride == 1 && dh == 1 && dw == 1) wino_list.push_back(&L);
    if (ks == 3 && stride == 1 && dh > 1) x3d_list.push_back(&L);         // the dilated ASPP branches (conv_x3d.hip)
}

// nets.BaseNet (lib/nets.py:10-24)
void Model::build_basenet(BaseNetL& B, const std::string& p, int nin, int c, int nin_lstm, int nout_lstm_) {
    B.prefix = p; B.c = c;
    const float RELU = 0.f, LEAKY = 0.01f;
    build_cba(B.enc1, p + ".enc1", nin, c, 3, 1, 1, 1, 1, 1, RELU);
    const int ch[5] = {c, 2 * c, 4 * c, 6 * c, 8 * c};
    for (int i = 0; i < 4; ++i) {
        const std::string e = p + ".enc" + std::to_string(i + 2);
        build_cba(B.enc_a[i], e + ".conv1", ch[i], ch[i + 1], 3, 2, 1, 1, 1, 1, LEAKY);
        build_cba(B.enc_b[i], e + ".conv2", ch[i + 1], ch[i + 1], 3, 1, 1, 1, 1, 1, LEAKY);
    }
    const int C8 = 8 * c;
    build_cba(B.aspp_pool, p + ".aspp.conv1.1", C8, C8, 1, 1, 0, 0, 1, 1, RELU);
    build_cba(B.aspp_c2, p + ".aspp.conv2", C8, C8, 1, 1, 0, 0, 1, 1, RELU);
    x3d_list.push_back(&B.aspp_c2);                                        // (runs beside the dilated branches in their launch)
    const int dil[3][2] = {{4, 2}, {8, 4}, {12, 6}};
    for (int i = 0; i < 3; ++i)
        build_cba(B.aspp_d[i], p + ".aspp.conv" + std::to_string(i + 3), C8, C8, 3, 1, dil[i][0], dil[i][1], dil[i][0],
                  dil[i][1], RELU);
    build_cba(B.aspp_bott, p + ".aspp.bottleneck", 5 * C8, C8, 1, 1, 0, 0, 1, 1, RELU);
    build_cba(B.dec[0], p + ".dec4.conv1", 14 * c, 6 * c, 3, 1, 1, 1, 1, 1, RELU);
    build_cba(B.dec[1], p + ".dec3.conv1", 10 * c, 4 * c, 3, 1, 1, 1, 1, 1, RELU);
    build_cba(B.dec[2], p + ".dec2.conv1", 6 * c, 2 * c, 3, 1, 1, 1, 1, 1, RELU);
    // layers.LSTMModule (lib/layers.py:110-122)
    LSTMMod& M = B.lstm;
    const std::string q = p + ".lstm_dec2";
    const int hid = nout_lstm_ / 2;
    M.nin = nin_lstm; M.hid = hid;
    M.squeeze.name = q + ".conv";
    M.squeeze.Cin = 2 * c; M.squeeze.Cout = 1; M.squeeze.KS = 1; M.squeeze.slope = RELU;
    M.squeeze.w = add_param(q + ".conv.conv.0.weight", {1, 2 * c, 1, 1}, PK_PLAIN, true);
    M.squeeze.bn = add_bn(q + ".conv.conv.1", 1, nin_lstm);
    M.proj.name = q + ".lstm";
    M.proj.Cin = nin_lstm; M.proj.Cout = 8 * hid; M.proj.CoutPad = round_up(8 * hid, 32);
    M.proj.KS = 1; M.proj.stride = 1; M.proj.pad_h = M.proj.pad_w = 0; M.proj.bn = nullptr; M.proj.slope = 1.f;
    Param* ihf = add_param(q + ".lstm.weight_ih_l0", {4 * hid, nin_lstm}, PK_LSTM_IH, true);
    M.whh_f = add_param(q + ".lstm.weight_hh_l0", {4 * hid, hid}, PK_PLAIN, true);
    M.b_ih_f = add_param(q + ".lstm.bias_ih_l0", {4 * hid}, PK_PLAIN, true);
    M.b_hh_f = add_param(q + ".lstm.bias_hh_l0", {4 * hid}, PK_PLAIN, true);
    Param* ihr = add_param(q + ".lstm.weight_ih_l0_reverse", {4 * hid, nin_lstm}, PK_LSTM_IH, true);
    M.whh_r = add_param(q + ".lstm.weight_hh_l0_reverse", {4 * hid, hid}, PK_PLAIN, true);
    M.b_ih_r = add_param(q + ".lstm.bias_ih_l0_reverse", {4 * hid}, PK_PLAIN, true);
    M.b_hh_r = add_param(q + ".lstm.bias_hh_l0_reverse", {4 * hid}, PK_PLAIN, true);
    ihf->Cout = 4 * hid; ihf->Cin = nin_lstm; ihf->KK = 1; ihf->CoutPad = M.proj.CoutPad; ihf->co_off = 0;
    ihf->dev_numel = (size_t)nin_lstm * M.proj.CoutPad;
    ihr->Cout = 4 * hid; ihr->Cin = nin_lstm; ihr->KK = 1; ihr->CoutPad = M.proj.CoutPad; ihr->co_off = 4 * hid;
    ihr->dev_numel = 0; ihr->alias_of = ihf;
    M.proj.w = ihf;
    M.dense.name = q + ".dense";
    M.dense.Cin = 2 * hid; M.dense.Cout = nin_lstm; M.dense.CoutPad = round_up(nin_lstm, 32);
    M.dense.KS = 1; M.dense.stride = 1; M.dense.pad_h = M.dense.pad_w = 0; M.dense.slope = 0.f;
    M.dense.w = add_param(q + ".dense.0.weight", {nin_lstm, 2 * hid}, PK_CONV, true);
    M.dense_b = add_param(q + ".dense.0.bias", {nin_lstm}, PK_PLAIN, true);
    M.dense.bn = add_bn(q + ".dense.1", nin_lstm, 0);
    build_cba(B.dec[3], p + ".dec1.conv1", 3 * c + 1, c, 3, 1, 1, 1, 1, 1, RELU);
}

Model::Model(int device_, int n_fft_, int hop_, int nout_, int nout_lstm_)
    : device(device_), n_fft(n_fft_), hop(hop_), nout(nout_), nout_lstm(nout_lstm_) {
    VR_CHECK(n_fft >= 64 && (n_fft & (n_fft - 1)) == 0 && n_fft <= 8192, -2, "n_fft must be a power of two in [64, 8192]");
    VR_CHECK(hop > 0, -2, "hop_length must be positive");
    VR_CHECK(nout % 4 == 0 && nout >= 4 && nout_lstm % 4 == 0 && nout_lstm >= 4, -2, "nout / nout_lstm must be multiples of 4");
    max_bin = n_fft / 2;
    output_bin = n_fft / 2 + 1;
    VR_CHECK((max_bin / 2) % 16 == 0, -2, "n_fft/4 must be a multiple of 16 (four stride-2 encoders)");
    DeviceGuard dev_guard(device);
    if (const char* e = getenv("VR_MFMA_MODE")) { mfma_mode = atoi(e); VR_CHECK(mfma_mode >= 0 && mfma_mode <= 3, -2, "VR_MFMA_MODE: 0, 1, 2 or 3"); }
    default_mfma_mode = mfma_mode;
    VR_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (!getenv("VR_NO_SIDE_STREAM")) {
        VR_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
        VR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        VR_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        if (!getenv("VR_NO_SPLIT_BATCH")) {
            int k = getenv("VR_LANES") ? atoi(getenv("VR_LANES")) : 2;
            k = k < 1 ? 1 : (k > 4 ? 4 : k);
            lanes.resize(k - 1);
            for (Lane& l : lanes) {
                VR_HIP(hipStreamCreateWithFlags(&l.main, hipStreamNonBlocking));
                VR_HIP(hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking));
                VR_HIP(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
                VR_HIP(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
                VR_HIP(hipEventCreateWithFlags(&l.start, hipEventDisableTiming));
                VR_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
            }
        }
    }
    const int nin = 2;
    const int nin_lstm = max_bin / 2;
    // lib/nets.py:59-80
    build_basenet(nets_[0], "stg1_low_band_net.0", nin, nout / 2, nin_lstm / 2, nout_lstm);
    build_cba(tail1, "stg1_low_band_net.1", nout / 2, nout / 4, 1, 1, 0, 0, 1, 1, 0.f);
    build_basenet(nets_[1], "stg1_high_band_net", nin, nout / 4, nin_lstm / 2, nout_lstm / 2);
    build_basenet(nets_[2], "stg2_low_band_net.0", nout / 4 + nin, nout, nin_lstm / 2, nout_lstm);
    build_cba(tail2, "stg2_low_band_net.1", nout, nout / 2, 1, 1, 0, 0, 1, 1, 0.f);
    build_basenet(nets_[3], "stg2_high_band_net", nout / 4 + nin, nout / 2, nin_lstm / 2, nout_lstm / 2);
    build_basenet(nets_[4], "stg3_full_band_net", 3 * nout / 4 + nin, nout, nin_lstm, nout_lstm);
    out_w = add_param("out.weight", {nin, nout, 1, 1}, PK_PLAIN, true);
    aux_out_w = add_param("aux_out.weight", {nin, 3 * nout / 4, 1, 1}, PK_PLAIN, true);   // never used (nets.py:80)
    finalize_layout();

    // FFT plan: twiddles and periodic Hann window computed in double on the host
    plan.n_fft = n_fft;
    plan.log2n = 0;
    while ((1 << plan.log2n) < n_fft) ++plan.log2n;
    std::vector<float2> tw(n_fft / 2);
    std::vector<float> win(n_fft);
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < n_fft / 2; ++k) {
        const double a = -2.0 * PI * k / n_fft;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int i = 0; i < n_fft; ++i) win[i] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * i / n_fft));
    VR_HIP(hipMalloc(&plan.twiddle, tw.size() * sizeof(float2)));
    VR_HIP(hipMalloc(&plan.window, win.size() * sizeof(float)));
    VR_HIP(hipMemcpy(plan.twiddle, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(plan.window, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
}

void Model::finalize_layout() {
    size_t poff = 0, boff = 0;
    auto bump = [](size_t& o, size_t n) { size_t a = (o + 63) & ~size_t(63); o = a + n; return a; };
    for (auto& p : params) {
        if (p.kind == PK_NBT || p.alias_of) continue;
        if (p.trainable) p.off = bump(poff, p.dev_numel);
        else p.off = bump(boff, p.dev_numel);
    }
    // BatchNorm affine tables + saved batch statistics live in the buffer arena
    std::vector<size_t> aff_off(bn_list.size()), sm_off(bn_list.size()), si_off(bn_list.size());
    for (size_t i = 0; i < bn_list.size(); ++i) {
        BN* b = bn_list[i];
        const int rows = b->bcast ? b->bcast : b->C;
        aff_off[i] = bump(boff, (size_t)rows * 2);
        sm_off[i] = bump(boff, (size_t)b->C);
        si_off[i] = bump(boff, (size_t)b->C);
    }
    size_t aspp_off[5];
    for (int i = 0; i < 5; ++i) aspp_off[i] = bump(boff, (size_t)4 * 8 * nets_[i].c * 2);
    p_floats = poff + 64; b_floats = boff + 64;
    VR_HIP(hipMalloc(&p_arena, p_floats * sizeof(float)));
    VR_HIP(hipMalloc(&b_arena, b_floats * sizeof(float)));
    VR_HIP(hipMemset(p_arena, 0, p_floats * sizeof(float)));
    VR_HIP(hipMemset(b_arena, 0, b_floats * sizeof(float)));
    for (auto& p : params) {
        if (p.kind == PK_NBT || p.alias_of) continue;
        p.dev = (p.trainable ? p_arena : b_arena) + p.off;
    }
    for (auto& p : params)
        if (p.alias_of) p.dev = p.alias_of->dev;
    for (size_t i = 0; i < bn_list.size(); ++i) {
        bn_list[i]->affine = b_arena + aff_off[i];
        bn_list[i]->save_mean = b_arena + sm_off[i];
        bn_list[i]->save_invstd = b_arena + si_off[i];
    }
    for (int i = 0; i < 5; ++i) {
        BaseNetL& B = nets_[i];
        B.aspp_aff = b_arena + aspp_off[i];
        const size_t blk = (size_t)8 * B.c * 2;
        B.aspp_c2.bn->affine = B.aspp_aff;
        for (int j = 0; j < 3; ++j) B.aspp_d[j].bn->affine = B.aspp_aff + (j + 1) * blk;
    }
    // default BatchNorm state = torch's fresh module: weight 1, running_var 1
    for (BN* b : bn_list) {
        std::vector<float> ones(b->C, 1.f);
        VR_HIP(hipMemcpy(b->w->dev, ones.data(), b->C * sizeof(float), hipMemcpyHostToDevice));
        VR_HIP(hipMemcpy(b->rv->dev, ones.data(), b->C * sizeof(float), hipMemcpyHostToDevice));
    }
    std::vector<BNFoldDesc> descs;
    for (BN* b : bn_list) descs.push_back(BNFoldDesc{b->w->dev, b->b->dev, b->rm->dev, b->rv->dev, b->affine, b->C, b->bcast});
    VR_HIP(hipMalloc(&d_fold, descs.size() * sizeof(BNFoldDesc)));
    VR_HIP(hipMemcpy(d_fold, descs.data(), descs.size() * sizeof(BNFoldDesc), hipMemcpyHostToDevice));
    affine_dirty = true;
}

Model::~Model() {
    if (launch_prof && !prof_owned_by_this_thread(launch_prof)) prof_abandon(launch_prof);   // the opening thread still holds the pointer
    else { if (g_launch_prof == launch_prof) g_launch_prof = nullptr; prof_destroy(launch_prof); }
    int prev_dev = -1;
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    hipSetDevice(device);
    struct Back { int d; ~Back() { if (d >= 0) hipSetDevice(d); } } back{prev_dev};
    try { comm_destroy(); } catch (...) {}
    hipFree(wire_buf);
    if (stream) hipStreamSynchronize(stream);
    hipFree(p_arena); hipFree(b_arena); hipFree(d_fold);
    hipFree(ws.base); hipFree(io.base); hipFree(gs.base); hipFree(wino_arena); hipFree(winot_arena); hipFree(wino6_arena); hipFree(winot6_arena); hipFree(x3_arena); hipFree(x3t_arena); hipFree(xb_fwd.dev); hipFree(xb_bwd.dev); hipFree(wb_fwd.dev); hipFree(wb_bwd.dev); hipFree(wb_fwd6.dev); hipFree(wb_bwd6.dev); hipFree(aug_buf); hipFree(wred_dev); hipFree(d_s2w);
    for (BaseNetL& B : nets_) hipFree(B.lstm.bias_sum);
    hipFree(g_arena); hipFree(m_arena); hipFree(v_arena); hipFree(wt_arena); hipFree(d_flip); hipFree(dropout_buf);
    hipFree(plan.twiddle); hipFree(plan.window);
    for (Lane& l : lanes) {
        hipStreamSynchronize(l.main); hipStreamSynchronize(l.side);
        hipStreamDestroy(l.main); hipStreamDestroy(l.side);
        hipEventDestroy(l.fork); hipEventDestroy(l.join); hipEventDestroy(l.start); hipEventDestroy(l.done);
        hipFree(l.ws.base);
    }
    if (side_stream) { hipStreamSynchronize(side_stream); hipStreamDestroy(side_stream); hipEventDestroy(ev_fork); hipEventDestroy(ev_join); }
    if (stream) hipStreamDestroy(stream);
}

// =====================================================================================================
// parameters
// =====================================================================================================
void Model::set_param(const std::string& key, const void* host, const int64_t* shape, int ndim) {
    auto it = by_key.find(key);
    VR_CHECK(it != by_key.end(), -2, "unknown parameter key: " + key);
    Param& p = *it->second;
    VR_CHECK((size_t)ndim == p.shape.size(), -2, "shape rank mismatch for " + key);
    for (int i = 0; i < ndim; ++i) VR_CHECK(shape[i] == p.shape[i], -2, "shape mismatch for " + key);
    DeviceGuard dev_guard(device);
    VR_HIP(hipStreamSynchronize(stream));
    if (p.kind == PK_NBT) { p.nbt = *static_cast<const int64_t*>(host); return; }
    const float* src = static_cast<const float*>(host);
    if (p.kind == PK_CONV) {
        std::vector<float> tmp((size_t)p.Cin * p.KK * p.CoutPad, 0.f);
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci)
                for (int k = 0; k < p.KK; ++k)
                    tmp[((size_t)ci * p.KK + k) * p.CoutPad + co] = src[((size_t)co * p.Cin + ci) * p.KK + k];
        VR_HIP(hipMemcpy(p.dev, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    } else if (p.kind == PK_LSTM_IH) {
        std::vector<float> tmp((size_t)p.Cin * p.Cout);
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci) tmp[(size_t)ci * p.Cout + co] = src[(size_t)co * p.Cin + ci];
        VR_HIP(hipMemcpy2D(p.dev + p.co_off, (size_t)p.CoutPad * sizeof(float), tmp.data(), (size_t)p.Cout * sizeof(float),
                           (size_t)p.Cout * sizeof(float), (size_t)p.Cin, hipMemcpyHostToDevice));
    } else {
        VR_HIP(hipMemcpy(p.dev, src, p.numel * sizeof(float), hipMemcpyHostToDevice));
    }
    affine_dirty = true;
}

void Model::get_param(const std::string& key, void* host, int64_t cap_bytes) {
    auto it = by_key.find(key);
    VR_CHECK(it != by_key.end(), -2, "unknown parameter key: " + key);
    Param& p = *it->second;
    DeviceGuard dev_guard(device);
    VR_HIP(hipStreamSynchronize(stream));
    if (p.kind == PK_NBT) {
        VR_CHECK(cap_bytes >= 8, -2, "buffer too small");
        *static_cast<int64_t*>(host) = p.nbt;
        return;
    }
    VR_CHECK((size_t)cap_bytes >= p.numel * sizeof(float), -2, "buffer too small for " + key);
    float* dst = static_cast<float*>(host);
    if (p.kind == PK_CONV) {
        std::vector<float> tmp((size_t)p.Cin * p.KK * p.CoutPad);
        VR_HIP(hipMemcpy(tmp.data(), p.dev, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci)
                for (int k = 0; k < p.KK; ++k)
                    dst[((size_t)co * p.Cin + ci) * p.KK + k] = tmp[((size_t)ci * p.KK + k) * p.CoutPad + co];
    } else if (p.kind == PK_LSTM_IH) {
        std::vector<float> tmp((size_t)p.Cin * p.Cout);
        VR_HIP(hipMemcpy2D(tmp.data(), (size_t)p.Cout * sizeof(float), p.dev + p.co_off, (size_t)p.CoutPad * sizeof(float),
                           (size_t)p.Cout * sizeof(float), (size_t)p.Cin, hipMemcpyDeviceToHost));
        for (int co = 0; co < p.Cout; ++co)
            for (int ci = 0; ci < p.Cin; ++ci) dst[(size_t)co * p.Cin + ci] = tmp[(size_t)ci * p.Cout + co];
    } else {
        VR_HIP(hipMemcpy(dst, p.dev, p.numel * sizeof(float), hipMemcpyDeviceToHost));
    }
}

void Model::set_training(bool t) {
    if (training && !t) affine_dirty = true;     // batch-stat affines must be replaced by running-stat ones
    training = t;
}

void Model::fold_eval_affines() {
    if (!affine_dirty) return;
    int maxC = 1;
    for (BN* b : bn_list) maxC = std::max(maxC, std::max(b->C, b->bcast));
    launch_bn_fold_eval(d_fold, (int)bn_list.size(), maxC, 1e-5f, stream);
    refresh_wino(false);        // the weights may have changed too (set_param / Adam)
    for (BaseNetL& B : nets_) {  // LSTM gate biases b_ih + b_hh, once per parameter change instead of two launches per LSTM and forward
        LSTMMod& M = B.lstm;
        const int G = 4 * M.hid;
        if (!M.bias_sum) VR_HIP(hipMalloc(reinterpret_cast<void**>(&M.bias_sum), (size_t)2 * G * sizeof(float)));
        launch_add(M.b_ih_f->dev, M.b_hh_f->dev, M.bias_sum, G, stream);
        launch_add(M.b_ih_r->dev, M.b_hh_r->dev, M.bias_sum + G, G, stream);
    }
    affine_dirty = false;
}

void Model::set_option(const std::string& name, int value) {
    plan_peak = 0;                                           // kernel choice and scratch layout depend on the options: re-plan the next forward
    if (name == "train_winograd") train_wino = value != 0;
    else if (name == "serial_exec") serial = value != 0;     // every kernel on the handle's one stream (race detector of the tests)
    else if (name == "params_dirty") affine_dirty = true;    // the parameter arena was written from outside (vr_param_arena)
    else if (name == "mfma_bf16") { mfma_mode = value != 0 ? 1 : default_mfma_mode; affine_dirty = true; }   // bf16 operands on the matrix pipe (0: back to the handle's default mode)
    else if (name == "mfma_mode") {                          // 0 fp32 MFMA, 1 bf16 operands, 2 fp32 via 6 bf16 products, 3 via 3 fp16 products (model.h); -1 = the default
        if (value < -1 || value > 3) throw Error(-2, "mfma_mode: 0, 1, 2, 3 or -1 (default)");
        mfma_mode = value < 0 ? default_mfma_mode : value; affine_dirty = true;   // (the next eval forward refreshes the derived weight copies)
    }
    else if (name == "conv_x3d") {                           // the 16-column layers on the fp16 pipe (conv_x3d.hip, mfma_mode 3): 0 off (conv_dma.hip as in round 5),
        if (value < -1 || value > 2) throw Error(-2, "conv_x3d: 0, 1, 2 or -1 (default)");     // 1 one launch per conv, 2 (default) + the four ASPP branches in one launch
        x3d_mode = value < 0 ? 2 : value;
    }
    else if (name == "adam_reset") reset_adam_state();      // a freshly constructed torch.optim.Adam has no moments
    else if (name == "hip_graph" || name == "conv_x3p" || name == "wgrad_x3h" || name == "conv_x3b") {
        // options of round 4 whose kernels moved to tools/experiments in round 5: accepted and ignored, so that an older caller keeps
        // working; said once per option name
        static std::string told;
        if (told.find("|" + name + "|") == std::string::npos) {
            told += "|" + name + "|";
            fprintf(stderr, "libvr_mi355: option '%s' was retired with its kernel (tools/experiments/README.md); ignored\n", name.c_str());
        }
    }
    else throw Error(-2, "unknown option: " + name);
}

// Winograd-domain copies (G g G^T) of every 3x3 stride-1 weight; with_dgrad: also of the flipped/transposed
// weights the data gradient convolves with (train.hip keeps those in wt_of, refreshed once per step).
void Model::refresh_wino(bool with_dgrad) {
    if (!wino_arena) {
        size_t total = 0;
        for (Conv* L : wino_list) total += (size_t)L->Cin * 16 * L->CoutPad;
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&wino_arena), total * sizeof(float)));
        size_t off = 0;
        for (Conv* L : wino_list) { L->wino = wino_arena + off; off += (size_t)L->Cin * 16 * L->CoutPad; }
    }
    {
        std::vector<WinoWDesc> d;
        for (Conv* L : wino_list) d.push_back(WinoWDesc{L->w->dev, L->wino, L->Cin, L->CoutPad});
        run_wino_batch(wb_fwd, d, false);
    }
    auto cin_pad = [](const Conv* L) { return (L->Cin + 31) / 32 * 32; };
    static const bool x3_on = !(getenv("VR_CONV_X3") && atoi(getenv("VR_CONV_X3")) == 0);
    if (x3_mode() && x3_on) {
        // split-bf16 (mode 2) / split-fp16 (mode 3) copies of the DIRECT weights (conv_x3.hip takes every 3x3 stride-1 launch wide enough for its tiles)
        // (x3d_list: the dilated 3x3 and 1x1 branch convs of the ASPP modules, conv_x3d.hip -- fp16 planes only, filled in mode 3)
        if (!x3_arena) {
            size_t total = 0;
            for (Conv* L : wino_list) total += x3_weights_bytes(L->Cin, 9, L->CoutPad);
            for (Conv* L : x3d_list) total += x3_weights_bytes(L->Cin, L->KS * L->KS, L->CoutPad);
            VR_HIP(hipMalloc(reinterpret_cast<void**>(&x3_arena), total ? total : 16));
            size_t off = 0;
            for (Conv* L : wino_list) { L->x3w = x3_arena + off; off += x3_weights_bytes(L->Cin, 9, L->CoutPad); }
            for (Conv* L : x3d_list) { L->x3w = x3_arena + off; off += x3_weights_bytes(L->Cin, L->KS * L->KS, L->CoutPad); }
        }
        {
            std::vector<X3WDesc> d;
            for (Conv* L : wino_list) d.push_back(X3WDesc{L->w->dev, L->x3w, L->Cin, 9, L->CoutPad});
            if (mfma_mode == 3)
                for (Conv* L : x3d_list) d.push_back(X3WDesc{L->w->dev, L->x3w, L->Cin, L->KS * L->KS, L->CoutPad});
            run_x3_batch(xb_fwd, d);
        }
        if (with_dgrad) {
            if (!x3t_arena) {
                size_t total = 0;
                for (Conv* L : wino_list) total += x3_weights_bytes(L->Cout, 9, cin_pad(L));
                for (Conv* L : x3d_list) total += x3_weights_bytes(L->Cout, L->KS * L->KS, cin_pad(L));
                VR_HIP(hipMalloc(reinterpret_cast<void**>(&x3t_arena), total ? total : 16));
                size_t off = 0;
                for (Conv* L : wino_list) { x3t_of[L->w] = x3t_arena + off; off += x3_weights_bytes(L->Cout, 9, cin_pad(L)); }
                for (Conv* L : x3d_list) { x3dt_of[L->w] = x3t_arena + off; off += x3_weights_bytes(L->Cout, L->KS * L->KS, cin_pad(L)); }
            }
            std::vector<X3WDesc> d;
            for (Conv* L : wino_list) {
                auto it = wt_of.find(L->w);
                if (it != wt_of.end()) d.push_back(X3WDesc{it->second, x3t_of[L->w], L->Cout, 9, cin_pad(L)});
            }
            if (mfma_mode == 3)
                for (Conv* L : x3d_list) {
                    auto it = wt_of.find(L->w);
                    if (it != wt_of.end()) d.push_back(X3WDesc{it->second, x3dt_of[L->w], L->Cout, L->KS * L->KS, cin_pad(L)});
                }
            run_x3_batch(xb_bwd, d);
        }
    }
    if (mfma_mode == 2 && !x3_on) {
        // bf16-plane copies for the launches that can take the split kernel: the 64-cout Winograd variant only (conv_wino.hip)
        auto fwd6 = [](const Conv* L) { return L->CoutPad % 64 == 0; };
        auto bwd6 = [&](const Conv* L) { return cin_pad(L) % 64 == 0; };
        if (!wino6_arena) {
            size_t total = 0;
            for (Conv* L : wino_list) if (fwd6(L)) total += wino_weights6_bytes(L->Cin, L->CoutPad);
            VR_HIP(hipMalloc(reinterpret_cast<void**>(&wino6_arena), total ? total : 16));
            size_t off = 0;
            for (Conv* L : wino_list) if (fwd6(L)) { L->wino6 = wino6_arena + off; off += wino_weights6_bytes(L->Cin, L->CoutPad); }
        }
        {
            std::vector<WinoWDesc> d;
            for (Conv* L : wino_list) if (fwd6(L)) d.push_back(WinoWDesc{L->w->dev, L->wino6, L->Cin, L->CoutPad});
            run_wino_batch(wb_fwd6, d, true);
        }
        if (with_dgrad) {
            if (!winot6_arena) {
                size_t total = 0;
                for (Conv* L : wino_list) if (bwd6(L)) total += wino_weights6_bytes(L->Cout, cin_pad(L));
                VR_HIP(hipMalloc(reinterpret_cast<void**>(&winot6_arena), total ? total : 16));
                size_t off = 0;
                for (Conv* L : wino_list) if (bwd6(L)) { winot6_of[L->w] = winot6_arena + off; off += wino_weights6_bytes(L->Cout, cin_pad(L)); }
            }
            std::vector<WinoWDesc> d;
            for (Conv* L : wino_list) {
                auto it = wt_of.find(L->w);
                if (bwd6(L) && it != wt_of.end()) d.push_back(WinoWDesc{it->second, winot6_of[L->w], L->Cout, cin_pad(L)});
            }
            run_wino_batch(wb_bwd6, d, true);
        }
    }
    if (!with_dgrad) return;
    if (!winot_arena) {
        size_t total = 0;
        for (Conv* L : wino_list) total += (size_t)L->Cout * 16 * cin_pad(L);
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&winot_arena), total * sizeof(float)));
        size_t off = 0;
        for (Conv* L : wino_list) { winot_of[L->w] = winot_arena + off; off += (size_t)L->Cout * 16 * cin_pad(L); }
    }
    std::vector<WinoWDesc> d;
    for (Conv* L : wino_list) {
        auto it = wt_of.find(L->w);
        if (it != wt_of.end()) d.push_back(WinoWDesc{it->second, winot_of[L->w], L->Cout, cin_pad(L)});
    }
    run_wino_batch(wb_bwd, d, false);
}

void Model::run_x3_batch(X3Batch& b, std::vector<X3WDesc>& descs) {
    if (descs.empty()) return;
    bool same = b.dev && b.host.size() == descs.size();
    for (size_t i = 0; same && i < descs.size(); ++i)
        same = b.host[i].w == descs[i].w && b.host[i].o == descs[i].o && b.host[i].Cin == descs[i].Cin && b.host[i].CoutPad == descs[i].CoutPad;
    if (!same) {
        VR_HIP(hipStreamSynchronize(stream));
        if (b.dev) VR_HIP(hipFree(b.dev));
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&b.dev), descs.size() * sizeof(X3WDesc)));
        VR_HIP(hipMemcpy(b.dev, descs.data(), descs.size() * sizeof(X3WDesc), hipMemcpyHostToDevice));
        b.host = descs;
        b.max_elems = 0;
        for (const X3WDesc& e : descs) b.max_elems = std::max(b.max_elems, (long long)((e.Cin + 7) / 8 * 8) * e.KK * e.CoutPad);
    }
    if (mfma_mode == 3) {
        int mc = 0;
        for (const X3WDesc& e : descs) mc = std::max(mc, e.CoutPad);
        launch_x3h_weights_batched(b.dev, (int)descs.size(), b.max_elems, mc, stream);
    } else {
        launch_x3_weights_batched(b.dev, (int)descs.size(), b.max_elems, stream);
    }
}

// One launch for a whole descriptor table; the device copy is re-uploaded only when the table changed.
void Model::run_wino_batch(WinoBatch& b, std::vector<WinoWDesc>& descs, bool split6) {
    if (descs.empty()) return;
    bool same = b.dev && b.host.size() == descs.size();
    for (size_t i = 0; same && i < descs.size(); ++i)
        same = b.host[i].w == descs[i].w && b.host[i].u == descs[i].u && b.host[i].Cin == descs[i].Cin && b.host[i].CoutPad == descs[i].CoutPad;
    if (!same) {
        VR_HIP(hipStreamSynchronize(stream));                       // (a launch may still read the old table)
        if (b.dev) VR_HIP(hipFree(b.dev));
        VR_HIP(hipMalloc(reinterpret_cast<void**>(&b.dev), descs.size() * sizeof(WinoWDesc)));
        VR_HIP(hipMemcpy(b.dev, descs.data(), descs.size() * sizeof(WinoWDesc), hipMemcpyHostToDevice));
        b.host = descs;
        b.max_elems = 0;
        for (const WinoWDesc& e : descs) {
            const long long cin = split6 ? (e.Cin + 7) / 8 * 8 : e.Cin;
            b.max_elems = std::max(b.max_elems, cin * e.CoutPad);
        }
    }
    launch_wino_weights_batched(b.dev, (int)descs.size(), b.max_elems, split6, stream);
}

// =====================================================================================================
// workspace
// =====================================================================================================
void Model::ensure_ws(size_t bytes) {
    if (bytes <= ws.cap) return;
    VR_HIP(hipStreamSynchronize(stream));
    if (ws.base) VR_HIP(hipFree(ws.base));
    ws.base = nullptr; ws.cap = 0;
    const size_t want = bytes + (bytes >> 4) + (1 << 20);
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&ws.base), want));
    ws.cap = want;
}

// Exchange lane A (stream, side_stream, ws) with lane B: run_net() and the launch helpers only know the
// member names, so the second half-batch is enqueued by swapping, running, swapping back.
void Model::swap_lane(int i) {
    Lane& l = lanes[i];
    std::swap(stream, l.main);
    std::swap(side_stream, l.side);
    std::swap(ev_fork, l.fork);
    std::swap(ev_join, l.join);
    std::swap(ws, l.ws);
}

void Model::ensure_io(size_t bytes) {
    if (bytes <= io.cap) return;
    VR_HIP(hipStreamSynchronize(stream));
    if (io.base) VR_HIP(hipFree(io.base));
    io.base = nullptr; io.cap = 0;
    const size_t want = bytes + (bytes >> 4) + (1 << 20);
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&io.base), want));
    io.cap = want;
}

// Algorithmic HBM bytes of one convolution launch -- forward, data gradient or weight gradient alike: the (virtual) input-sized
// tensor, the output-sized tensor and the weights, each crossing HBM once.
double Model::conv_alg_bytes(const Conv& L, const ConvArgs& a, int N, bool batch_as_h) const {
    return 4.0 * ((double)a.N * L.Cin * a.Hin * a.Win + (double)N * L.Cout * (batch_as_h ? 1 : a.Hout) * a.Wout +
                  (double)L.Cin * L.KS * L.KS * L.Cout);
}

// The executor's note for the next launch: a convolution's algorithmic FLOPs (bytes / tag follow through record_note).
void Model::record_begin(int kind, double flops) {
    if (!profiling || dry) return;
    (void)kind;
    prof_note(flops, 0.0, true, "conv");
}
void Model::record_note(double bytes, const char* tag) {
    if (!profiling || dry || !g_launch_prof) return;
    prof_note_update(bytes, tag);
}
void Model::record_end() {
    if (!profiling || dry) return;
    prof_note_clear();                                   // (a dispatcher that launched nothing leaves no stale note)
}

void Model::profile_begin() {
    VR_CHECK(!launch_prof || prof_owned_by_this_thread(launch_prof), -3, "vr_profile_begin: a profile opened by another thread is still open");
    prof_destroy(launch_prof);
    launch_prof = prof_create();
    g_launch_prof = launch_prof;
    profiling = true;
}

void Model::profile_end(double* conv_ms, double* conv_flops, double* conv_bytes, int* launches) {
    DeviceGuard dev_guard(device);
    VR_CHECK(!launch_prof || prof_owned_by_this_thread(launch_prof), -3, "vr_profile_end must be called from the thread that called vr_profile_begin");
    VR_HIP(hipDeviceSynchronize());                      // every stream the profiled step used
    g_launch_prof = nullptr;
    profiling = false;
    static const bool dump = getenv("VR_PROFILE_DUMP") != nullptr;           // per-launch table on stderr
    if (launch_prof) prof_collect(launch_prof, conv_ms, conv_flops, conv_bytes, launches, &profile_report, dump);
    prof_destroy(launch_prof);
    launch_prof = nullptr;
}

float* Model::galloc(size_t n, bool plain) {
    const size_t off = (gs.off + 255) & ~size_t(255);            // (Arena::alloc's own rounding)
    float* p = gs.allocf(n);
    if (!plain) gs_zero.push_back({off, n * sizeof(float)});
    else if (!dry) g_fresh[p] = true;
    return p;
}

bool Model::g_first(const float* g) {
    auto it = g_fresh.find(g);
    if (it == g_fresh.end() || !it->second) return false;
    it->second = false;
    return true;
}

// zero the buffers of `gs` that need it (ranges recorded by the planning dry run; adjacent ones merged)
void Model::clear_gs_zero_ranges(hipStream_t st) {
    // VR_GS_POISON=1 (debug): every byte of the arena is first set to 0xFF (a NaN pattern), so a backward writer that ACCUMULATES into a
    // buffer nobody stored to first (first-writer-stores keys g_fresh by the exact base pointer: an offset or partial view would get
    // g_first() == false) turns its tensor into NaNs instead of silently adding to last step's values -- the train-parity tests run
    // once under it (tests/test_gpu_train.py)
    static const bool poison = [] { const char* e = getenv("VR_GS_POISON"); return e && atoi(e) != 0; }();
    if (poison && gs.base && gs.cap) VR_HIP(hipMemsetAsync(gs.base, 0xFF, gs.cap, st));
    size_t i = 0;
    while (i < gs_zero_plan.size()) {
        size_t b = gs_zero_plan[i].first, e = b + gs_zero_plan[i].second;
        size_t j = i + 1;
        while (j < gs_zero_plan.size() && gs_zero_plan[j].first <= ((e + 255) & ~size_t(255))) {
            e = std::max(e, gs_zero_plan[j].first + gs_zero_plan[j].second);
            ++j;
        }
        prof_memset_async(gs.base + b, 0, e - b, st);
        i = j;
    }
}

void Model::tap(const std::string& name, const Tensor& t) {
    if (record_taps && !dry) taps[name] = t;
}

int64_t Model::get_tap(const std::string& name, float* host, int64_t cap_floats, int64_t* shape4) {
    DeviceGuard dev_guard(device);
    auto it = taps.find(name);
    VR_CHECK(it != taps.end(), -2, "no such tap: " + name);
    const Tensor& t = it->second;
    const int64_t n = (int64_t)t.N * t.C * t.H * t.W;
    if (shape4) { shape4[0] = t.N; shape4[1] = t.C; shape4[2] = t.H; shape4[3] = t.W; }
    if (!host) return n;
    VR_CHECK(cap_floats >= n, -2, "tap buffer too small");
    float* tmp = nullptr;
    VR_HIP(hipMalloc(&tmp, n * sizeof(float)));
    launch_materialize(t, tmp, stream);
    VR_HIP(hipStreamSynchronize(stream));
    VR_HIP(hipMemcpy(host, tmp, n * sizeof(float), hipMemcpyDeviceToHost));
    VR_HIP(hipFree(tmp));
    return n;
}

// =====================================================================================================
// executor
// =====================================================================================================
ConvSrc make_src(const Tensor& t, bool up, int bcastH) {
    ConvSrc s{};
    s.p = t.p; s.aff0 = t.aff0; s.aff1 = t.aff1 ? t.aff1 : t.aff0;
    s.sN = t.sN; s.sC = t.sC; s.sH = t.sH;
    s.C = t.C; s.H = t.H; s.W = t.W;
    s.hsplit = t.hsplit; s.slope = t.slope; s.up = up ? 1 : 0; s.post = t.post; s.zins = 0;
    s.rh = (t.H > 0) ? (float)(t.H - 1) / (float)(2 * t.H - 1) : 0.f;
    s.rw = (t.W > 0) ? (float)(t.W - 1) / (float)(2 * t.W - 1) : 0.f;
    if (bcastH) { s.sH = 0; s.H = bcastH; }
    return s;
}

// Forward launch description of a conv layer (shared by the forward pass and the weight gradient).
void Model::build_fwd_args(Conv& L, const std::vector<SrcSpec>& srcs, int N, bool batch_as_h, ConvArgs& a) {
    VR_CHECK(!srcs.empty() && srcs.size() <= 3, -2, "conv " + L.name + ": 1..3 sources");
    a = ConvArgs{};
    a.nsrc = (int)srcs.size();
    int Hin = -1, Win = -1, ctot = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        const SrcSpec& sp = srcs[i];
        a.src[i] = make_src(sp.t, sp.up, sp.bcastH);
        const int vh = sp.up ? 2 * sp.t.H : (sp.bcastH ? sp.bcastH : sp.t.H);
        const int vw = sp.up ? 2 * sp.t.W : sp.t.W;
        if (sp.plain) {                       // materialised view of the same values
            ConvSrc& c = a.src[i];
            c = ConvSrc{};
            c.p = sp.plain; c.C = sp.t.C; c.H = vh; c.W = vw;
            c.sH = vw; c.sC = (long long)vh * vw; c.sN = c.sC * sp.t.C;
            c.hsplit = 1 << 30; c.slope = 1.f;
        }
        if (Hin < 0) { Hin = vh; Win = vw; }
        // spec_utils.crop_center (lib/spec_utils.py:8-23) is the identity for every valid shape;
        // anything else is the reference's ValueError.
        VR_CHECK(vh == Hin && vw == Win, -5, "h1_shape[3] must be greater than h2_shape[3] (decoder skip/upsample size mismatch in " + L.name + ")");
        ctot += sp.t.C;
    }
    VR_CHECK(ctot == L.Cin, -2, "conv " + L.name + ": channel count mismatch");
    {   // the loader keeps ONE interpolation table per workgroup: upsampled sources must share geometry
        const ConvSrc* u = nullptr;
        for (int i = 0; i < a.nsrc; ++i) {
            if (!a.src[i].up) continue;
            if (u) VR_CHECK(u->H == a.src[i].H && u->W == a.src[i].W && u->sH == a.src[i].sH, -2,
                            "conv " + L.name + ": upsampled sources must share H, W and row stride");
            u = &a.src[i];
        }
    }
    a.c1 = a.nsrc >= 2 ? srcs[0].t.C : L.Cin;
    a.c2 = a.nsrc >= 3 ? srcs[0].t.C + srcs[1].t.C : L.Cin;
    a.Cin = L.Cin;
    a.w = L.w->dev;
    a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.d1 = a.d2 = 1 << 30;
    int Nk = N;
    if (batch_as_h) {
        VR_CHECK(L.KS == 1 && Hin == 1, -2, "batch-as-rows view needs a 1x1 conv on H=1 input");
        for (int i = 0; i < a.nsrc; ++i) {
            VR_CHECK(!a.src[i].post && !a.src[i].up, -2, "batch-as-rows: unsupported source flags");
            a.src[i].sH = a.src[i].sN; a.src[i].sN = 0; a.src[i].H = N;
        }
        Hin = N; Nk = 1;
    }
    a.N = Nk; a.Hin = Hin; a.Win = Win;
    a.Hout = (Hin + 2 * L.pad_h - L.dh * (L.KS - 1) - 1) / L.stride + 1;
    a.Wout = (Win + 2 * L.pad_w - L.dw * (L.KS - 1) - 1) / L.stride + 1;
    a.pad_h = L.pad_h; a.pad_w = L.pad_w;
}

Tensor Model::run_conv(Conv& L, const std::vector<SrcSpec>& srcs_in, int N, const Tensor* out_view, const float* bias,
                       bool batch_as_h) {
    // Training: give the conv plain inputs (one element-wise / upsample pass per source) -- the forward conv
    // then takes the LDS-DMA kernel and the weight gradient re-reads the same buffers without arithmetic.
    std::vector<SrcSpec> srcs = srcs_in;
    static const bool mat_enabled = !getenv("VR_NO_TRAIN_MAT");
    if (training && mat_enabled) {
        for (SrcSpec& sp : srcs) {
            const Tensor& t = sp.t;
            if (!(t.aff0 || t.aff1 || t.post || t.slope != 1.f || sp.up || sp.bcastH)) continue;
            const int vh = sp.up ? 2 * t.H : (sp.bcastH ? sp.bcastH : t.H), vw = sp.up ? 2 * t.W : t.W;
            float* buf = ws.allocf((size_t)t.N * t.C * vh * vw);
            if (!dry) {
                if (sp.up) launch_upsample2x(t, buf, stream);
                else {
                    Tensor v = t;
                    if (sp.bcastH) { v.H = sp.bcastH; v.sH = 0; }
                    launch_materialize(v, buf, stream);
                }
            }
            sp.plain = buf;
        }
    }
    ConvArgs a;
    build_fwd_args(L, srcs, N, batch_as_h, a);
    if (!training) {
        // fused-upsample sources are for conv_x3.hip only: anything else gets the materialised tensor after all
        bool any_up = false;
        for (const SrcSpec& sp : srcs) any_up = any_up || sp.up;
        if (any_up) {
            ConvArgs probe = a;
            probe.x3w = x3_mode() ? L.x3w : nullptr;
            probe.bf16 = mfma_mode;
            X3Tile xt;
            if (!x3_pick(probe, ConvShape{L.KS, L.stride, L.dh, L.dw}, &xt)) {
                for (SrcSpec& sp : srcs) {
                    if (!sp.up) continue;
                    const Tensor& t = sp.t;
                    Tensor u;
                    u.N = t.N; u.C = t.C; u.H = 2 * t.H; u.W = 2 * t.W;
                    u.sH = u.W; u.sC = (long long)u.H * u.W; u.sN = u.sC * u.C; u.slope = 1.f;
                    u.p = ws.allocf((size_t)u.N * u.C * u.H * u.W);
                    if (!dry) launch_upsample2x(t, u.p, stream);
                    sp = SrcSpec{u};
                }
                build_fwd_args(L, srcs, N, batch_as_h, a);
            }
        }
    }
    a.bias = bias;
    // Eval: the folded BatchNorm + activation go into the conv's epilogue, so the stored tensor is the
    // final activation and its consumers load it with no arithmetic (conv_dma.hip).  Training keeps
    // the raw tensor + pending affine: the batch statistics only exist after the whole conv has run.
    const bool fuse_epi = !training && L.bn != nullptr;
    if (fuse_epi) { a.epi = L.bn->affine; a.epi_slope = L.slope; }
    a.wino = (training && !train_wino) ? nullptr : L.wino;   // (null until the first refresh_wino())
    a.wino6 = (a.wino && mfma_mode == 2) ? L.wino6 : nullptr;
    a.x3w = (x3_mode() && !(training && !train_wino)) ? L.x3w : nullptr;
    if (!x3d_mode && a.Win == 16) a.x3w = nullptr;          // (16-column layers: only conv_x3d.hip reads the planes)
    a.bf16 = mfma_mode;
    Tensor o;
    if (batch_as_h) {
        o.N = N; o.C = L.Cout; o.H = 1; o.W = a.Wout;
        if (out_view) { o.p = out_view->p; o.g = out_view->g; o.sN = out_view->sN; o.sC = out_view->sC; o.sH = out_view->sH; }
        else {
            o.p = ws.allocf((size_t)N * L.Cout * a.Wout); o.sN = (long long)L.Cout * a.Wout; o.sC = a.Wout; o.sH = a.Wout;
            if (taping()) o.g = galloc((size_t)N * L.Cout * a.Wout, false);      // (written through the batch-as-rows view: keep it zeroed)
        }
        a.dst[0] = ConvDst{o.p, 0, o.sC, o.sN, 0};
    } else {
        o.N = N; o.C = L.Cout; o.H = a.Hout; o.W = a.Wout;
        if (out_view) {
            VR_CHECK(out_view->H == a.Hout && out_view->W == a.Wout && out_view->C == L.Cout, -2, "conv " + L.name + ": output view shape mismatch");
            o.p = out_view->p; o.g = out_view->g; o.sN = out_view->sN; o.sC = out_view->sC; o.sH = out_view->sH;
        } else {
            o.p = ws.allocf((size_t)N * L.Cout * a.Hout * a.Wout);
            o.sH = a.Wout; o.sC = (long long)a.Hout * a.Wout; o.sN = o.sC * L.Cout;
            if (taping()) o.g = galloc((size_t)N * L.Cout * a.Hout * a.Wout, true);
        }
        a.dst[0] = ConvDst{o.p, o.sN, o.sC, o.sH, 0};
    }
    const ConvShape shp{L.KS, L.stride, L.dh, L.dw};
    const bool stats = training && L.bn;
    size_t npt = 0;
    if (stats) {
        npt = conv_part_count(a, shp);
        a.part = ws.allocf(npt * L.Cout * 2);
    }
    if (!dry && conv_sink) {
        // the caller launches this conv together with its siblings (ASPP branches: one conv_x3d launch for the four) and, in training,
        // finalises the BatchNorm statistics behind it
        PendingConv pc{};
        pc.a = a; pc.shp = shp;
        pc.flops = 2.0 * N * (double)a.Hout * a.Wout * (double)L.Cout * L.Cin * L.KS * L.KS;
        pc.bytes = conv_alg_bytes(L, a, N, batch_as_h);
        pc.stats = stats; pc.bn = L.bn;
        if (stats) {
            BNFinalizeArgs& f = pc.fin;
            f.part = a.part; f.nparts = (int)npt; f.pstride = L.Cout * 2;
            f.count = (double)N * a.Hout * a.Wout;
            f.w = L.bn->w->dev; f.b = L.bn->b->dev; f.rm = L.bn->rm->dev; f.rv = L.bn->rv->dev;
            f.affine = L.bn->affine; f.save_mean = L.bn->save_mean; f.save_invstd = L.bn->save_invstd;
            f.C = L.Cout; f.eps = 1e-5f; f.momentum = 0.1f; f.broadcast = 0;
        }
        conv_sink->push_back(pc);
    } else if (!dry) {
        const double flops = 2.0 * N * (double)(batch_as_h ? 1 : a.Hout) * a.Wout * (double)L.Cout * L.Cin * L.KS * L.KS;
        record_begin(0, flops);
        if (profiling) {
            // algorithmic HBM bytes of the launch: the virtual input, the 3x3/1x1 weights and the output, once each
            char tag[160];
            snprintf(tag, sizeof tag, "%s k%d s%d d%d ci%d co%d %dx%dx%d", L.name.c_str(), L.KS, L.stride, L.dh, L.Cin,
                     L.Cout, a.N, a.Hout, a.Wout);
            record_note(conv_alg_bytes(L, a, N, batch_as_h), tag);
        }
        launch_conv(a, shp, stream);
        record_end();
        if (stats) {
            BNFinalizeArgs f{};
            f.part = a.part; f.nparts = (int)npt; f.pstride = L.Cout * 2;
            f.count = (double)N * (batch_as_h ? 1 : a.Hout) * a.Wout;
            f.w = L.bn->w->dev; f.b = L.bn->b->dev; f.rm = L.bn->rm->dev; f.rv = L.bn->rv->dev;
            f.affine = L.bn->affine; f.save_mean = L.bn->save_mean; f.save_invstd = L.bn->save_invstd;
            f.C = L.Cout; f.eps = 1e-5f; f.momentum = 0.1f; f.broadcast = 0;
            launch_bn_finalize(f, stream);
            L.bn->nbt->nbt += 1;
        }
    }
    if (L.bn && !fuse_epi) { o.aff0 = L.bn->affine; o.slope = L.slope; } else { o.aff0 = nullptr; o.slope = 1.f; }
    if (taping()) {
        TapeRec r;
        r.kind = TK_CONV; r.L = &L; r.srcs = srcs; r.out = o; r.N = N; r.batch_as_h = batch_as_h; r.bias = bias;
        tape.push_back(std::move(r));
    }
    return o;
}

// layers.LSTMModule.forward (lib/layers.py:124-133)
Tensor Model::run_lstm(LSTMMod& M, const Tensor& h) {
    const int N = h.N, nb = h.H, nf = h.W;
    VR_CHECK(nb == M.nin, -2, "LSTM input size does not match the number of bins at dec2");
    // 1x1 conv 2c -> 1 (+BN+ReLU): raw z [N][nb][nf]
    float* z = ws.allocf((size_t)N * nb * nf);
    const int nblk = launch_squeeze_conv(h, M.squeeze.w->dev, z, nullptr, true, stream);
    float* part = training ? ws.allocf((size_t)nblk * 2) : nullptr;
    if (!dry) {
        // eval: BatchNorm (running statistics, folded) + ReLU in the kernel's store, so the LSTM input projection reads a plain tensor
        launch_squeeze_conv(h, M.squeeze.w->dev, z, part, false, stream, training ? nullptr : M.squeeze.bn->affine);
        if (training) {
            BN* b = M.squeeze.bn;
            BNFinalizeArgs f{};
            f.part = part; f.nparts = nblk; f.pstride = 2; f.count = (double)N * nb * nf;
            f.w = b->w->dev; f.b = b->b->dev; f.rm = b->rm->dev; f.rv = b->rv->dev;
            f.affine = b->affine; f.save_mean = b->save_mean; f.save_invstd = b->save_invstd;
            f.C = 1; f.eps = 1e-5f; f.momentum = 0.1f; f.broadcast = b->bcast;
            launch_bn_finalize(f, stream);
            b->nbt->nbt += 1;
        }
    }
    // z as [N, C=nb, H=1, W=nf] (bins become channels): the LSTM input projection is a 1x1 conv
    Tensor zt;
    zt.p = z; zt.N = N; zt.C = nb; zt.H = 1; zt.W = nf;
    zt.sN = (long long)nb * nf; zt.sC = nf; zt.sH = nf;
    if (training) { zt.aff0 = M.squeeze.bn->affine; zt.slope = 0.f; }
    else zt.slope = 1.f;
    if (taping()) {
        zt.g = galloc((size_t)N * nb * nf, false);
        TapeRec r;
        r.kind = TK_SQUEEZE; r.M = &M; r.srcs = {SrcSpec{h}}; r.N = N;
        r.out = zt;                      // as a 1-channel image [N,1,nb,nf] for its own BatchNorm backward
        r.out.C = 1; r.out.H = nb; r.out.sC = (long long)nb * nf; r.out.sH = nf;
        tape.push_back(std::move(r));
    }
    const int G = 4 * M.hid;
    float* bias = ws.allocf((size_t)2 * G);               // (allocated in every mode: the planning dry run must see one sequence)
    if (!training && M.bias_sum) bias = M.bias_sum;       // eval: refreshed by fold_eval_affines()
    else if (!dry) {
        launch_add(M.b_ih_f->dev, M.b_hh_f->dev, bias, G, stream);
        launch_add(M.b_ih_r->dev, M.b_hh_r->dev, bias + G, G, stream);
    }
    Tensor gx = run_conv(M.proj, {SrcSpec{zt}}, N, nullptr, bias, true);       // [N][8H][nf]
    float* hc = ws.allocf((size_t)N * 2 * M.hid * nf);                          // [N][2H][nf]
    float* save = taping() ? ws.allocf((size_t)N * 2 * nf * 5 * M.hid) : nullptr;
    if (!dry) launch_bilstm_train(gx.p, M.whh_f->dev, M.whh_r->dev, hc, save, N, nf, M.hid, stream);
    Tensor ht;
    ht.p = hc; ht.N = N; ht.C = 2 * M.hid; ht.H = 1; ht.W = nf;
    ht.sN = (long long)2 * M.hid * nf; ht.sC = nf; ht.sH = nf; ht.slope = 1.f;
    if (taping()) {
        ht.g = galloc((size_t)N * 2 * M.hid * nf, false);
        TapeRec r;
        r.kind = TK_LSTM; r.M = &M; r.N = N; r.out = ht; r.aux = gx; r.save = save;
        tape.push_back(std::move(r));
    }
    Tensor lin = run_conv(M.dense, {SrcSpec{ht}}, N, nullptr, M.dense_b->dev, true);   // [N][nb][nf] raw
    if (taping()) tape.back().bias_param = M.dense_b;
    // BatchNorm1d + ReLU (lib/layers.py:120-121).  Eval: in place.  Train: the raw values are what the
    // BatchNorm backward needs, so the activated copy goes to its own buffer and shares lin's gradient.
    float* act = training ? ws.allocf((size_t)N * nb * nf) : lin.p;
    if (!dry && training) launch_rows_affine_relu(lin.p, act, M.dense.bn->affine, N, nb, nf, stream);   // eval: fused into the conv epilogue
    Tensor o;                                                                   // [N,1,nb,nf], already activated
    o.p = act; o.g = lin.g; o.N = N; o.C = 1; o.H = nb; o.W = nf;
    o.sN = (long long)nb * nf; o.sC = (long long)nb * nf; o.sH = nf; o.slope = 1.f;
    return o;
}

// Decoder input (lib/layers.py:52): training fuses the x2 bilinear upsample into the consuming conv's
// loader; eval materialises it once (HBM-bound, small) so the conv reads a plain tensor by LDS-DMA.
Model::SrcSpec Model::upsampled(const Tensor& t) {
    if (training) {
        SrcSpec s{t};
        s.up = true;
        return s;
    }
    // Eval, split-bf16 mode: conv_x3.hip interpolates while it splits the pixels into bf16 planes, so the full-resolution decoder
    // layers read the LOW-resolution tensor (a quarter of the bytes) and nothing is materialised.  Measured per layer (tools/
    // x3_proto.hip): it pays from 512 x 128 output pixels on (dec1, stage-3 dec2); below, the interpolation VALU of the many-channel
    // layers costs more than the HBM-bound upsample pass.
    static const bool fuse_on = !(getenv("VR_X3_FUSE_UP") && atoi(getenv("VR_X3_FUSE_UP")) == 0);
    static const bool x3_on = !(getenv("VR_CONV_X3") && atoi(getenv("VR_CONV_X3")) == 0);
    if (fuse_on && x3_on && x3_mode() && 4LL * t.H * t.W >= 65536 && 2 * t.W >= 32 && !t.aff0 && !t.aff1 && !t.post && t.slope == 1.f) {
        SrcSpec s{t};
        s.up = true;
        return s;
    }
    Tensor u;
    u.N = t.N; u.C = t.C; u.H = 2 * t.H; u.W = 2 * t.W;
    u.sH = u.W; u.sC = (long long)u.H * u.W; u.sN = u.sC * u.C; u.slope = 1.f;
    u.p = ws.allocf((size_t)u.N * u.C * u.H * u.W);
    if (!dry) launch_upsample2x(t, u.p, stream);
    return SrcSpec{u};
}

// nets.BaseNet.__call__ (lib/nets.py:26-41)
Tensor Model::run_basenet(BaseNetL& B, const std::vector<SrcSpec>& in, int N, const Tensor* out_view) {
    const std::string& p = B.prefix;
    Tensor e[5];
    e[0] = run_conv(B.enc1, in, N, nullptr, nullptr, false);
    tap(p + ".e1", e[0]);
    for (int i = 0; i < 4; ++i) {
        Tensor t = run_conv(B.enc_a[i], {SrcSpec{e[i]}}, N, nullptr, nullptr, false);
        e[i + 1] = run_conv(B.enc_b[i], {SrcSpec{t}}, N, nullptr, nullptr, false);
        tap(p + ".e" + std::to_string(i + 2), e[i + 1]);
    }
    // layers.ASPPModule.forward (lib/layers.py:92-105)
    const Tensor& x5 = e[4];
    const int C8 = 8 * B.c;
    float* pooled = ws.allocf((size_t)N * C8 * x5.W);
    if (!dry) launch_avgpool_h(x5, pooled, stream);
    Tensor pt;
    pt.p = pooled; pt.N = N; pt.C = C8; pt.H = 1; pt.W = x5.W;
    pt.sN = (long long)C8 * x5.W; pt.sC = x5.W; pt.sH = x5.W; pt.slope = 1.f;
    if (taping()) {
        pt.g = galloc((size_t)N * C8 * x5.W, false);
        TapeRec r;
        r.kind = TK_AVGPOOL; r.srcs = {SrcSpec{x5}}; r.out = pt; r.N = N;
        tape.push_back(std::move(r));
    }
    Tensor f1 = run_conv(B.aspp_pool, {SrcSpec{pt}}, N, nullptr, nullptr, true);
    // conv2..conv5 write channel slices of one [N, 4*C8, H, W] buffer (the concat is never built)
    Tensor cat4;
    cat4.N = N; cat4.C = 4 * C8; cat4.H = x5.H; cat4.W = x5.W;
    cat4.sH = x5.W; cat4.sC = (long long)x5.H * x5.W; cat4.sN = cat4.sC * cat4.C;
    cat4.p = ws.allocf((size_t)N * cat4.C * x5.H * x5.W);
    if (taping()) cat4.g = galloc((size_t)N * cat4.C * x5.H * x5.W, false);
    if (training) { cat4.aff0 = B.aspp_aff; cat4.slope = 0.f; }      // eval: the branch convs store final activations
    Conv* branch[4] = {&B.aspp_c2, &B.aspp_d[0], &B.aspp_d[1], &B.aspp_d[2]};
    // Training: the four branches read the same activated tensor -- materialised ONCE here (run_conv would do it once per branch); the plain
    // copy shares x5's gradient buffer, so the four data gradients still meet in x5.g
    Tensor x5b = x5;
    {
        static const bool mat_enabled = !getenv("VR_NO_TRAIN_MAT");
        if (training && mat_enabled && (x5.aff0 || x5.aff1 || x5.post || x5.slope != 1.f)) {
            float* buf = ws.allocf((size_t)N * C8 * x5.H * x5.W);
            if (!dry) launch_materialize(x5, buf, stream);
            x5b.p = buf; x5b.aff0 = x5b.aff1 = nullptr; x5b.post = nullptr; x5b.slope = 1.f; x5b.hsplit = 1 << 30;
            x5b.sH = x5.W; x5b.sC = (long long)x5.H * x5.W; x5b.sN = x5b.sC * C8;
        }
    }
    // Eval, stage 3: the four branch convs are independent and each fills < 256 CUs at 1/16 resolution --
    // two of them go to the idle side stream.
    static const bool aspp_fork = !getenv("VR_NO_ASPP_FORK");
    // Eval, mfma_mode 3 (round 6): the four branch convs go out as ONE launch on the fp16 matrix pipe (conv_x3d.hip: blockIdx.y = branch).
    // They are collected first; if any of them does not qualify (another mode, VR_ASPP_FUSED=0) each is launched on its own as before.
    std::vector<PendingConv> pend;
    const bool group = !dry && mfma_mode == 3 && x3d_mode == 2 && !(training && !train_wino);
    const bool afk = aspp_fork && !serial && !training && !dry && !profiling && side_stream != nullptr && !band_fork_active && !group;
    hipStream_t aspp_main = stream;
    // (Round 5, measured and removed: every branch on its own stream -- three auxiliary streams per lane, in every stage -- made the
    // inference step 12.4 ms instead of 8.65, also with 8 or 16 hardware queues: each extra fork / join pair costs more than the overlap
    // of four ~60 us kernels returns.  Two streams, stage 3 only, is the measured optimum.)
    if (afk) {
        hipEvent_t ef = ev_fork;
        VR_HIP(hipEventRecord(ef, aspp_main));
        VR_HIP(hipStreamWaitEvent(side_stream, ef, 0));
    }
    for (int j = 0; j < 4; ++j) {
        Tensor v = cat4;
        v.C = C8;
        v.p = dry ? cat4.p : cat4.p + (long long)j * C8 * cat4.sC;
        if (taping() && !dry) v.g = cat4.g + (long long)j * C8 * cat4.sC;
        if (afk) stream = (j & 1) ? side_stream : aspp_main;
        if (group) conv_sink = &pend;
        try { run_conv(*branch[j], {SrcSpec{x5b}}, N, &v, nullptr, false); } catch (...) { stream = aspp_main; conv_sink = nullptr; throw; }
        conv_sink = nullptr;
    }
    if (group) {
        VR_CHECK(pend.size() == 4, -3, "ASPP: four branch convs expected");
        ConvArgs c4[4];
        ConvShape s4[4];
        double flops = 0.0, bytes = 0.0;
        for (int j = 0; j < 4; ++j) { c4[j] = pend[j].a; s4[j] = pend[j].shp; flops += pend[j].flops; bytes += pend[j].bytes; }
        if (x3d_aspp_eligible(c4, s4)) {
            record_begin(0, flops);
            if (profiling) {
                char tag[160];
                snprintf(tag, sizeof tag, "%s.aspp.conv2-5 k1+3x(k3 d4/8/12) ci%d co%d %dx%dx%d", p.c_str(), C8, C8, N, x5.H, x5.W);
                // (the four read the same input: counted once)
                record_note(bytes - 3.0 * 4.0 * (double)N * C8 * x5.H * x5.W, tag);
            }
            x3d_launch_aspp(c4, stream);
            record_end();
        } else {
            for (int j = 0; j < 4; ++j) {
                record_begin(0, pend[j].flops);
                if (profiling) record_note(pend[j].bytes, branch[j]->name.c_str());
                launch_conv(c4[j], s4[j], stream);
                record_end();
            }
        }
        for (int j = 0; j < 4; ++j)
            if (pend[j].stats) {
                launch_bn_finalize(pend[j].fin, stream);
                pend[j].bn->nbt->nbt += 1;
            }
    }
    if (afk) {
        stream = aspp_main;
        hipEvent_t ej = ev_join;
        VR_HIP(hipEventRecord(ej, side_stream));
        VR_HIP(hipStreamWaitEvent(aspp_main, ej, 0));
    }
    SrcSpec s1{f1};
    s1.bcastH = x5.H;          // bilinear from H=1 with align_corners=True is a broadcast along H
    Tensor h = run_conv(B.aspp_bott, {s1, SrcSpec{cat4}}, N, nullptr, nullptr, false);
    if (training && dropout_dev) {
        int idx = (int)(&B - nets_);
        h.post = dropout_dev + (size_t)idx * N * 8 * nout;   // [5][N][8*nout] slots, row pitch 8c
        if (taping()) tape.back().out.post = h.post;          // the bottleneck's BatchNorm backward needs it
    }
    tap(p + ".aspp", h);
    // decoders (lib/layers.py:51-64): upsample x2 + skip concat + conv, all inside the conv's loader
    for (int i = 0; i < 3; ++i) {
        SrcSpec up = upsampled(h);
        h = run_conv(B.dec[i], {up, SrcSpec{e[3 - i]}}, N, nullptr, nullptr, false);
        tap(p + ".dec" + std::to_string(4 - i), h);
    }
    // Eval, stage 3 (the side stream is idle there): the x2 upsample of h (HBM-bound) runs beside the LSTM branch
    // (latency-bound); both only need h, and dec1 needs both.
    static const bool lstm_fork = !getenv("VR_NO_LSTM_FORK");
    const bool fk = lstm_fork && !serial && !training && !dry && !profiling && side_stream != nullptr && !band_fork_active;
    SrcSpec uh;
    hipEvent_t lstm_join = nullptr;
    if (fk) {
        hipStream_t ms = stream;
        hipEvent_t ef = ev_fork;
        VR_HIP(hipEventRecord(ef, ms));
        VR_HIP(hipStreamWaitEvent(side_stream, ef, 0));
        stream = side_stream;
        try { uh = upsampled(h); } catch (...) { stream = ms; throw; }
        lstm_join = ev_join;
        VR_HIP(hipEventRecord(lstm_join, side_stream));
        stream = ms;
    }
    Tensor l = run_lstm(B.lstm, h);
    tap(p + ".lstm", l);
    if (fk) VR_HIP(hipStreamWaitEvent(stream, lstm_join, 0));
    else uh = upsampled(h);
    SrcSpec ul = upsampled(l);
    Tensor o = run_conv(B.dec[3], {uh, ul, SrcSpec{e[0]}}, N, out_view, nullptr, false);
    tap(p + ".dec1", o);
    return o;
}

// CascadedNet.forward up to (not including) `out` (lib/nets.py:86-102)
Tensor Model::run_net(const Tensor& x) {
    const int B = x.N, T = x.W;
    const int bandw = max_bin / 2;
    Tensor xl = x; xl.H = bandw;
    Tensor xh = x; xh.H = bandw; xh.p = dry ? x.p : x.p + (long long)bandw * x.sH;
    auto make_aux = [&](int C) {
        Tensor t;
        t.N = B; t.C = C; t.H = max_bin; t.W = T;
        t.sH = T; t.sC = (long long)max_bin * T; t.sN = t.sC * C;
        t.p = ws.allocf((size_t)B * C * max_bin * T);
        if (taping()) t.g = galloc((size_t)B * C * max_bin * T, false);          // written through the full view and the band halves
        t.slope = training ? 0.f : 1.f;     // eval: dec1 / the tail convs store final activations
        return t;
    };
    Tensor aux1 = make_aux(nout / 4), aux2 = make_aux(nout / 2);
    auto half = [&](const Tensor& a, int which) {
        Tensor v = a; v.H = bandw;
        if (which && !dry) { v.p = a.p + (long long)bandw * a.sH; if (a.g) v.g = a.g + (long long)bandw * a.sH; }
        return v;
    };
    Tensor v;
    // The low-band chain (stg1_low -> stg2_low) and the high-band chain (stg1_high -> stg2_high) are
    // independent until stage 3 (lib/nets.py:91-98).  They run on two HIP streams (forward of both modes) so the
    // small 1/16-resolution layers of one chain fill the CUs the other leaves idle.
    // (not while per-kernel HIP-event timing is on: overlapping kernels would inflate each other's time)
    static const bool train_fork = !getenv("VR_NO_TRAIN_FORK");
    static const bool band_fork = !getenv("VR_NO_BAND_FORK");
    const bool fork = band_fork && !serial && !dry && (!training || train_fork) && !profiling && side_stream != nullptr;
    hipStream_t main_stream = stream;
    if (fork) {
        hipEvent_t ef = ev_fork;
        VR_HIP(hipEventRecord(ef, main_stream));
        VR_HIP(hipStreamWaitEvent(side_stream, ef, 0));
        band_fork_active = true;                 // until the join: the side stream belongs to the high-band chain
    }
    Tensor l1r = run_basenet(nets_[0], {SrcSpec{xl}}, B, nullptr);
    v = half(aux1, 0);
    Tensor l1 = run_conv(tail1, {SrcSpec{l1r}}, B, &v, nullptr, false);
    Tensor l2r = run_basenet(nets_[2], {SrcSpec{xl}, SrcSpec{l1}}, B, nullptr);
    v = half(aux2, 0);
    Tensor l2 = run_conv(tail2, {SrcSpec{l2r}}, B, &v, nullptr, false);
    if (fork) stream = side_stream;
    const size_t tape_hi0 = tape.size();
    v = half(aux1, 1);
    Tensor h1 = run_basenet(nets_[1], {SrcSpec{xh}}, B, &v);
    v = half(aux2, 1);
    Tensor h2 = run_basenet(nets_[3], {SrcSpec{xh}, SrcSpec{h1}}, B, &v);
    for (size_t i = tape_hi0; i < tape.size(); ++i) tape[i].chain = 1;    // backward may run these beside the low chain
    if (fork) {
        hipEvent_t ej = ev_join;
        VR_HIP(hipEventRecord(ej, side_stream));
        stream = main_stream;
        VR_HIP(hipStreamWaitEvent(main_stream, ej, 0));
        band_fork_active = false;
    }
    tap("l1", l1); tap("h1", h1);
    tap("l2", l2); tap("h2", h2);
    aux1.aff0 = l1.aff0; aux1.aff1 = h1.aff0; aux1.hsplit = bandw;
    aux2.aff0 = l2.aff0; aux2.aff1 = h2.aff0; aux2.hsplit = bandw;
    Tensor xf = x; xf.H = max_bin;
    Tensor f3 = run_basenet(nets_[4], {SrcSpec{xf}, SrcSpec{aux1}, SrcSpec{aux2}}, B, nullptr);
    return f3;
}

void Model::plan_and_reserve(int B, int T, size_t extra_bytes) {
    graph_valid = false;                         // the workspace is about to be rewound: a kept training graph dies here
    // the dry run is pure host work (~0.3 ms for the full net): remember its result per (B, T, mode)
    if (plan_B == B && plan_T == T && plan_training == training && plan_peak > 0) {
        ensure_ws(plan_peak + extra_bytes + 4096);
        ws.reset();
        return;
    }
    Arena saved = ws;
    ws.dry = true; ws.base = nullptr; ws.off = 0; ws.peak = 0;
    dry = true;
    Tensor x;
    x.p = reinterpret_cast<float*>(uintptr_t(256));
    x.N = B; x.C = 2; x.H = max_bin; x.W = T; x.sH = T; x.sC = (long long)output_bin * T; x.sN = 2 * x.sC;
    try { run_net(x); } catch (...) { dry = false; ws = saved; throw; }
    dry = false;
    const size_t need = ws.peak + extra_bytes + 4096;
    plan_B = B; plan_T = T; plan_training = training; plan_peak = ws.peak;
    ws = saved;
    ensure_ws(need);
    ws.reset();
}

static void check_T(int T, int offset, int mode) {
    VR_CHECK(T > 0 && T % 16 == 0, -5, "h1_shape[3] must be greater than h2_shape[3] (frames must be a multiple of 16)");
    if (mode != 0) VR_CHECK(T - 2 * offset > 0, -6, "assert mask.size()[3] > 0 (frames must exceed 2*offset)");
}

__global__ void mul_crop_kernel(const float* __restrict__ x, float* __restrict__ m, int T, int Wm, int off, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % Wm);
    const long long row = gid / Wm;
    m[gid] *= x[row * T + off + w];
}

void Model::forward_api(const float* x, bool x_on_device, int B, int T, int mode, float* out, bool out_on_device) {
    DeviceGuard dev_guard(device);
    VR_CHECK(B > 0, -2, "batch must be positive");
    check_T(T, offset, mode);
    const size_t in_floats = (size_t)B * 2 * output_bin * T;
    const int Wm = mode == 0 ? T : T - 2 * offset;
    const size_t out_floats = (size_t)B * 2 * output_bin * Wm;
    // Train mode (the reference's `model(X)` under model.train(), train.py:81 without the backward): BatchNorm uses
    // and updates batch statistics, Dropout2d is live, but no tape and no gradient buffers are kept.
    struct FwdOnly { Model* m; ~FwdOnly() { m->fwd_only = false; m->dropout_dev = nullptr; } } fwd_scope{this};
    fwd_only = training;
    if (training) refresh_wino(false);      // before planning: the kernel choice (and its partial-statistics layout) depends on them
    if (!training) fold_eval_affines();     // (eval: the bf16-plane weight tables decide conv_x3 vs the materialised-upsample plan)
    plan_and_reserve(B, T, (in_floats + out_floats) * sizeof(float) + 1024);
    if (training) prepare_dropout(B);
    float* xd = ws.allocf(in_floats);
    float* od = out_on_device ? out : ws.allocf(out_floats);
    if (x_on_device) VR_HIP(hipMemcpyAsync(xd, x, in_floats * sizeof(float), hipMemcpyDeviceToDevice, stream));
    else VR_HIP(hipMemcpyAsync(xd, x, in_floats * sizeof(float), hipMemcpyHostToDevice, stream));
    Tensor xt;
    xt.p = xd; xt.N = B; xt.C = 2; xt.H = max_bin; xt.W = T;
    xt.sH = T; xt.sC = (long long)output_bin * T; xt.sN = 2 * xt.sC; xt.slope = 1.f;
    if (record_taps) taps.clear();
    Tensor f3 = run_net(xt);
    HeadDst d{};
    d.p = od; d.dH = Wm; d.dC = (long long)output_bin * Wm; d.dN = 2 * d.dC;
    d.w_lo = mode == 0 ? 0 : offset; d.w_hi = mode == 0 ? T : T - offset; d.pad_rows = output_bin - max_bin;
    launch_head_sigmoid(f3, out_w->dev, d, stream);
    if (mode == 2) {
        const long long total = (long long)out_floats;
        VR_LAUNCH(mul_crop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xd, od, T, Wm,
                           offset, total);
        VR_HIP(hipGetLastError());
    }
    if (!out_on_device) VR_HIP(hipMemcpyAsync(out, od, out_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
    if (training) affine_dirty = true;
}

// sum |pred - crop_center(y)| per block; pred [rows][Wm] dense, y [rows][T] dense, columns [off, off+Wm) of y
__global__ __launch_bounds__(256) void l1_crop_kernel(const float* __restrict__ pred, const float* __restrict__ y, int T, int Wm,
                                                      int off, long long total, float* __restrict__ part) {
    float s = 0.f;
    for (long long gid = (long long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long long)gridDim.x * 256) {
        const int w = (int)(gid % Wm);
        const long long row = gid / Wm;
        s += fabsf(pred[gid] - y[row * T + off + w]);
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One batch of train.validate_epoch (train.py:117-127): y_pred = model.predict(X); y = crop_center(y, y_pred);
// loss = L1Loss()(y_pred, y) -- forward, crop and the mean-absolute-error reduction all on the device.
void Model::validate_api(const float* X, const float* Y, bool on_dev, int B, int T, float* loss_out) {
    DeviceGuard dev_guard(device);
    VR_CHECK(!training, -2, "validate step runs in eval mode (train.py:109 model.eval()); call vr_set_mode(h, 0) first");
    VR_CHECK(B > 0, -2, "batch must be positive");
    check_T(T, offset, 2);
    const size_t in_floats = (size_t)B * 2 * output_bin * T;
    const int Wm = T - 2 * offset;
    const size_t out_floats = (size_t)B * 2 * output_bin * Wm;
    const int nblk = 1024;
    fold_eval_affines();                    // before planning, as in forward_api
    plan_and_reserve(B, T, (2 * in_floats + out_floats + nblk + 64) * sizeof(float) + 8192);
    float* xd = ws.allocf(in_floats);
    const float* yd = Y;
    const hipMemcpyKind kind = on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    VR_HIP(hipMemcpyAsync(xd, X, in_floats * sizeof(float), kind, stream));
    if (!on_dev) {
        float* ty = ws.allocf(in_floats);
        VR_HIP(hipMemcpyAsync(ty, Y, in_floats * sizeof(float), hipMemcpyHostToDevice, stream));
        yd = ty;
    }
    float* od = ws.allocf(out_floats);
    float* part = ws.allocf(nblk);
    float* lossd = ws.allocf(16);
    Tensor xt;
    xt.p = xd; xt.N = B; xt.C = 2; xt.H = max_bin; xt.W = T;
    xt.sH = T; xt.sC = (long long)output_bin * T; xt.sN = 2 * xt.sC; xt.slope = 1.f;
    Tensor f3 = run_net(xt);
    HeadDst d{};
    d.p = od; d.dH = Wm; d.dC = (long long)output_bin * Wm; d.dN = 2 * d.dC;
    d.w_lo = offset; d.w_hi = T - offset; d.pad_rows = output_bin - max_bin;
    launch_head_sigmoid(f3, out_w->dev, d, stream);
    const long long total = (long long)out_floats;
    VR_LAUNCH(mul_crop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xd, od, T, Wm, offset, total);
    VR_HIP(hipGetLastError());
    VR_LAUNCH(l1_crop_kernel, dim3(nblk), dim3(256), 0, stream, od, yd, T, Wm, offset, total, part);
    VR_HIP(hipGetLastError());
    launch_reduce_rows(part, 1, nblk, lossd, 1, 0, (float)(1.0 / (double)total), stream);
    float loss_h = 0.f;
    VR_HIP(hipMemcpyAsync(&loss_h, lossd, sizeof(float), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
    if (loss_out) *loss_out = loss_h;
}

// =====================================================================================================
// signal path
// =====================================================================================================
void Model::stft_api(const float* wave, bool on_dev, long long L, float* spec, bool spec_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(L > 0, -2, "empty wave");
    const int T = 1 + (int)(L / hop);
    const size_t spec_f = (size_t)2 * output_bin * T * 2;
    ensure_io(((size_t)2 * L + spec_f) * sizeof(float) + 4096);
    io.reset();
    const float* wd = wave;
    if (!on_dev) {
        float* tmp = io.allocf((size_t)2 * L);
        VR_HIP(hipMemcpyAsync(tmp, wave, (size_t)2 * L * sizeof(float), hipMemcpyHostToDevice, stream));
        wd = tmp;
    }
    float* sd = spec_on_dev ? spec : io.allocf(spec_f);
    launch_stft(plan, wd, L, hop, T, reinterpret_cast<float2*>(sd), stream);
    if (!spec_on_dev) VR_HIP(hipMemcpyAsync(spec, sd, spec_f * sizeof(float), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
}

void Model::istft_api(const float* spec, bool on_dev, int T, float* wave, bool wave_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(T > 0, -2, "empty spectrogram");
    const size_t spec_f = (size_t)2 * output_bin * T * 2;
    const size_t out_f = (size_t)2 * hop * (T - 1);
    const size_t frames_f = (size_t)2 * T * n_fft;
    ensure_io((spec_f + out_f + frames_f) * sizeof(float) + 8192);
    io.reset();
    const float* sd = spec;
    if (!on_dev) {
        float* tmp = io.allocf(spec_f);
        VR_HIP(hipMemcpyAsync(tmp, spec, spec_f * sizeof(float), hipMemcpyHostToDevice, stream));
        sd = tmp;
    }
    float* frames = io.allocf(frames_f);
    float* wd = wave_on_dev ? wave : io.allocf(out_f + 4);
    launch_istft(plan, reinterpret_cast<const float2*>(sd), hop, T, frames, wd, stream);
    if (!wave_on_dev && out_f) VR_HIP(hipMemcpyAsync(wave, wd, out_f * sizeof(float), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
}

// spec_utils.merge_artifacts (lib/spec_utils.py:60-93) reduced to its per-frame weight vector
// (the reference builds a [2, bins, T] weight that is constant over channel and bin).  numpy slice
// semantics are kept: negative starts wrap, a slice/linspace length mismatch is numpy's ValueError,
// an empty above-threshold set is the IndexError the reference raises at `idx[0]`.
void merge_artifacts_weight(const std::vector<float>& fmin, std::vector<float>& weight, float thres, int min_range,
                            int fade) {
    const int T = (int)fmin.size();
    VR_CHECK(min_range >= fade * 2, -2, "min_range must be >= fade_size * 2");
    weight.assign((size_t)T, 0.f);
    std::vector<int> idx;
    for (int t = 0; t < T; ++t) if (fmin[t] > thres) idx.push_back(t);
    VR_CHECK(!idx.empty(), -7, "index 0 is out of bounds for axis 0 with size 0");
    std::vector<int> starts{idx[0]}, ends;
    for (size_t i = 1; i < idx.size(); ++i)
        if (idx[i] - idx[i - 1] != 1) { ends.push_back(idx[i - 1]); starts.push_back(idx[i]); }
    ends.push_back(idx.back());
    auto norm = [T](int i) { if (i < 0) i += T; return i < 0 ? 0 : (i > T ? T : i); };
    auto assign_ramp = [&](int a, int b, bool up) {       // weight[a:b] = linspace(0,1,fade) or linspace(1,0,fade)
        const int lo = norm(a), hi = norm(b);
        const int n = hi > lo ? hi - lo : 0;
        VR_CHECK(n == fade, -2, "could not broadcast input array from shape (" + std::to_string(fade) + ",) into shape (" + std::to_string(n) + ",)");
        for (int i = 0; i < fade; ++i) {
            const double x = (double)i / (fade - 1);
            weight[lo + i] = (float)(up ? x : 1.0 - x);
        }
    };
    bool have_old = false;
    int old_e = 0;
    for (size_t k = 0; k < starts.size(); ++k) {
        int s0 = starts[k], e0 = ends[k];
        if (!(e0 - s0 > min_range)) continue;
        if (have_old && s0 - old_e < fade) s0 = old_e - fade * 2;
        if (s0 != 0) assign_ramp(s0, s0 + fade, true); else s0 -= fade;
        if (e0 != T) assign_ramp(e0 - fade, e0, false); else e0 += fade;
        const int lo = norm(s0 + fade), hi = norm(e0 - fade);
        for (int i = lo; i < hi; ++i) weight[i] = 1.f;
        old_e = e0;
        have_old = true;
    }
}

// dataset.make_padding (lib/dataset.py:198-205)
static void make_padding(int width, int cropsize, int offset, int& left, int& right, int& roi) {
    left = offset;
    roi = cropsize - offset * 2;
    if (roi == 0) roi = cropsize;
    right = roi - (width % roi) + left;
}

// Separator.separate / separate_tta (inference.py:70-102) on device-resident spectrograms.
// spec_d, y_d, v_d: device [2][bins][T] complex64.  scratch comes from `io` (caller reserved).
static size_t separate_scratch_floats(int bins, int T, int cropsize, int offset, int tta) {
    int l, r, roi;
    make_padding(T, cropsize, offset, l, r, roi);
    const size_t Wpad2 = (size_t)T + l + r + roi;
    return 2 * (size_t)2 * bins * Wpad2 * (tta ? 2 : 1) + 2 * (size_t)T + (size_t)8 * bins + 4096;
}

void Model::separate_api(const float* spec, bool on_dev, int T, int tta, int batchsize, int cropsize, float* y_spec,
                         float* v_spec, bool out_on_dev, bool io_reserved, float* y_wave_d, float* v_wave_d) {
    DeviceGuard dev_guard(device);
    const bool post = (tta & 2) != 0;       // flags: bit 0 = --tta, bit 1 = --postprocess
    tta &= 1;
    VR_CHECK(T > 0, -2, "empty spectrogram");
    VR_CHECK(!training, -2, "separate() runs in eval mode (inference.py:52); call vr_set_mode(h, 0) first");
    check_T(cropsize, offset, 1);
    VR_CHECK(cropsize - 2 * offset > 0, -6, "cropsize must exceed 2*offset");
    const int bins = output_bin;
    const size_t spec_f = (size_t)2 * bins * T * 2;
    int pad_l, pad_r, roi;
    make_padding(T, cropsize, offset, pad_l, pad_r, roi);
    const size_t scratch = separate_scratch_floats(bins, T, cropsize, offset, tta);
    if (!io_reserved) {
        // direct call (host or device pointers): size and rewind the staging arena here; only the wave-level entry
        // point, which has already carved its own buffers out of `io`, passes io_reserved
        ensure_io((3 * spec_f + scratch) * sizeof(float) + 65536);
        io.reset();
    }
    const float* sd = spec;
    if (!on_dev) {
        float* tmp = io.allocf(spec_f);
        VR_HIP(hipMemcpyAsync(tmp, spec, spec_f * sizeof(float), hipMemcpyHostToDevice, stream));
        sd = tmp;
    }
    float* yd = out_on_dev ? y_spec : io.allocf(spec_f);
    float* vd = out_on_dev ? v_spec : io.allocf(spec_f);
    unsigned* stats = static_cast<unsigned*>(io.alloc(16 + (size_t)2 * bins * 16));
    float* in_aff = io.allocf(16);
    const int npass = tta ? 2 : 1;
    float* mask[2] = {nullptr, nullptr};
    int Wm[2] = {0, 0};
    int max_patches = 0;
    for (int ps = 0; ps < npass; ++ps) {
        const int Wpad = T + pad_l + pad_r + (ps ? roi : 0);
        max_patches = std::max(max_patches, (Wpad - 2 * offset) / roi);
    }
    const int bs = (batchsize <= 0) ? max_patches : std::min(batchsize, max_patches);
    fold_eval_affines();                    // before planning, as in forward_api
    plan_and_reserve(bs, cropsize, 0);
    for (int ps = 0; ps < npass; ++ps) {
        const int pl = pad_l + (ps ? roi / 2 : 0), pr = pad_r + (ps ? roi / 2 : 0);
        const int Wpad = T + pl + pr;
        const int patches = (Wpad - 2 * offset) / roi;
        float* mag = io.allocf((size_t)2 * bins * Wpad);
        Wm[ps] = patches * roi;
        mask[ps] = io.allocf((size_t)2 * bins * Wm[ps]);
        prof_memset_async(mag, 0, (size_t)2 * bins * Wpad * sizeof(float), stream);
        launch_mag_pad(reinterpret_cast<const float2*>(sd), bins, T, mag, Wpad, pl, stats, stream);
        launch_coef_affine(stats, 2 * bins, tta ? 1 : 0, in_aff, stream);
        {   // X_mag / coef once, so that the first conv of every BaseNet reads a plain tensor (LDS-DMA path)
            Tensor m;
            m.p = mag; m.N = 1; m.C = 2; m.H = bins; m.W = Wpad;
            m.sH = Wpad; m.sC = (long long)bins * Wpad; m.sN = 2 * m.sC;
            m.aff0 = in_aff; m.slope = 1.f;
            launch_materialize(m, mag, stream);          // element-wise, in place
        }
        auto run_crops = [&](int first, int count) {
            ws.reset();
            Tensor x;
            x.p = mag + (size_t)first * roi; x.N = count; x.C = 2; x.H = max_bin; x.W = cropsize;
            x.sN = roi; x.sC = (long long)bins * Wpad; x.sH = Wpad;
            x.slope = 1.f;
            Tensor f3 = run_net(x);
            HeadDst d{};
            d.p = mask[ps] + (size_t)first * roi; d.dN = roi; d.dC = (long long)bins * Wm[ps]; d.dH = Wm[ps];
            d.w_lo = offset; d.w_hi = cropsize - offset; d.pad_rows = output_bin - max_bin;
            launch_head_sigmoid(f3, out_w->dev, d, stream);
        };
        for (int i = 0; i < patches; i += bs) {
            const int nb = std::min(bs, patches - i);
            const int K = std::min((int)lanes.size() + 1, nb);
            if (K < 2 || profiling || serial) {
                run_crops(i, nb);
                continue;
            }
            // crops [i, i+nb) in K contiguous parts; part 0 on the handle's own streams, part j on lane j-1
            for (Lane& l : lanes) {
                if (l.ws.cap >= ws.cap) continue;            // every lane plans for the same (bs, cropsize)
                VR_HIP(hipDeviceSynchronize());
                if (l.ws.base) VR_HIP(hipFree(l.ws.base));
                l.ws.base = nullptr; l.ws.cap = 0;
                VR_HIP(hipMalloc(reinterpret_cast<void**>(&l.ws.base), ws.cap));
                l.ws.cap = ws.cap;
            }
            // The host enqueues part 0 completely before part 1 (~0.5 ms of launches: profiles/README.md), so the later lanes start late:
            // VR_LANE0_EXTRA = n gives part 0 n crops more than an even share.  Measured in round 6 (tools/gpu_r6_call7.sh, S30, two lanes):
            // 6 + 5 crops 8.63 - 8.78 ms, 7 + 4 8.69 - 8.73, 8 + 3 8.90; three lanes 10.4, one lane 9.24 -- the even split stays.
            static const int extra_env = getenv("VR_LANE0_EXTRA") ? atoi(getenv("VR_LANE0_EXTRA")) : -1;
            const int even = (nb + K - 1) / K;
            int head = extra_env >= 0 ? extra_env : 0;
            if (even + head > nb - (K - 1)) head = std::max(0, nb - (K - 1) - even);      // every part keeps at least one crop
            const int per0 = even + head;
            const int per = K > 1 ? (nb - per0 + K - 2) / (K - 1) : nb;
            for (int j = 1; j < K; ++j) {                    // `mag` is ready at this point of the main stream
                hipEvent_t es = lanes[j - 1].start;
                VR_HIP(hipEventRecord(es, stream));
                VR_HIP(hipStreamWaitEvent(lanes[j - 1].main, es, 0));
            }
            run_crops(i, std::min(per0, nb));
            for (int j = 1; j < K; ++j) {
                const int first = per0 + (j - 1) * per, count = std::min(per, nb - first);
                if (count <= 0) break;
                swap_lane(j - 1);
                try { run_crops(i + first, count); } catch (...) { swap_lane(j - 1); throw; }
                hipEvent_t done = lanes[j - 1].done;
                VR_HIP(hipEventRecord(done, stream));
                swap_lane(j - 1);
                VR_HIP(hipStreamWaitEvent(stream, done, 0));
            }
        }
    }
    const float* wgt = nullptr;
    if (post) {
        // spec_utils.merge_artifacts (lib/spec_utils.py:60-93): frames whose mask minimum exceeds the
        // threshold for more than min_range frames are pulled towards 1 with linear fades.  The per-frame
        // minimum is reduced on the GPU, the O(T) run logic runs on the host, the blend in apply_mask.
        float* fmin_d = io.allocf((size_t)T);
        float* wgt_d = io.allocf((size_t)T);
        launch_frame_min(bins, T, mask[0], Wm[0], tta ? mask[1] : nullptr, Wm[1], roi / 2, fmin_d, stream);
        std::vector<float> fmin((size_t)T), w;
        VR_HIP(hipMemcpyAsync(fmin.data(), fmin_d, (size_t)T * sizeof(float), hipMemcpyDeviceToHost, stream));
        VR_HIP(hipStreamSynchronize(stream));
        merge_artifacts_weight(fmin, w, 0.05f, 64, 32);
        VR_HIP(hipMemcpyAsync(wgt_d, w.data(), (size_t)T * sizeof(float), hipMemcpyHostToDevice, stream));
        VR_HIP(hipStreamSynchronize(stream));
        wgt = wgt_d;
    }
    if (y_wave_d && v_wave_d) {
        // wave-level caller: mask application, inverse FFT, window and overlap-add in one pass per stem -- the y / v
        // spectrograms (inference.py:32-38) are never materialised
        for (int which = 0; which < 2; ++which)
            launch_istft_masked(plan, reinterpret_cast<const float2*>(sd), hop, T, mask[0], Wm[0], tta ? mask[1] : nullptr, Wm[1],
                                roi / 2, wgt, which, which ? v_wave_d : y_wave_d, stream);
        enq_t1 = std::chrono::steady_clock::now(); enq_have = true;       // (VR_ENQ_TIMING: everything is enqueued at this point)
        VR_HIP(hipStreamSynchronize(stream));
        return;
    }
    launch_apply_mask(reinterpret_cast<const float2*>(sd), bins, T, mask[0], Wm[0], tta ? mask[1] : nullptr, Wm[1], roi / 2,
                      wgt, reinterpret_cast<float2*>(yd), reinterpret_cast<float2*>(vd), stream);
    if (!out_on_dev) {
        VR_HIP(hipMemcpyAsync(y_spec, yd, spec_f * sizeof(float), hipMemcpyDeviceToHost, stream));
        VR_HIP(hipMemcpyAsync(v_spec, vd, spec_f * sizeof(float), hipMemcpyDeviceToHost, stream));
    }
    VR_HIP(hipStreamSynchronize(stream));
}

void Model::separate_wave_api(const float* wave, bool on_dev, long long L, int tta, int batchsize, int cropsize,
                              float* y_wave, float* v_wave, bool out_on_dev) {
    DeviceGuard dev_guard(device);
    VR_CHECK(L >= hop, -2, "wave shorter than one hop");
    separate_wave_body(wave, on_dev, L, tta, batchsize, cropsize, y_wave, v_wave, out_on_dev);
}

void Model::separate_wave_body(const float* wave, bool on_dev, long long L, int tta, int batchsize, int cropsize,
                               float* y_wave, float* v_wave, bool out_on_dev) {
    // VR_ENQ_TIMING=1 (diagnostics): host time to ENQUEUE the whole call vs the time until the device has drained it
    static const bool enq_timing = getenv("VR_ENQ_TIMING") != nullptr;
    const auto enq_t0 = std::chrono::steady_clock::now();
    struct EnqReport {
        Model* m; bool on; std::chrono::steady_clock::time_point t0;
        ~EnqReport() {
            if (!on || !m->enq_have) return;
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[vr-enq] enqueue %.3f ms, total %.3f ms\n", std::chrono::duration<double, std::milli>(m->enq_t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t0).count());
            m->enq_have = false;
        }
    } enq_report{this, enq_timing, enq_t0};
    const int T = 1 + (int)(L / hop);
    const int bins = output_bin;
    const size_t spec_f = (size_t)2 * bins * T * 2;
    const size_t out_f = (size_t)2 * hop * (T - 1);
    const size_t frames_f = (size_t)2 * T * n_fft;
    const size_t scratch = separate_scratch_floats(bins, T, cropsize, offset, tta);
    ensure_io(((size_t)2 * L + 3 * spec_f + 2 * out_f + frames_f + scratch) * sizeof(float) + 65536);
    io.reset();
    float* stage_in = io.allocf((size_t)2 * L);
    const float* wd = wave;
    if (!on_dev) {
        VR_HIP(hipMemcpyAsync(stage_in, wave, (size_t)2 * L * sizeof(float), hipMemcpyHostToDevice, stream));
        wd = stage_in;
    }
    float* spec = io.allocf(spec_f);
    float* ys = io.allocf(spec_f);
    float* vs = io.allocf(spec_f);
    float* frames = io.allocf(frames_f);
    float* stage_y = io.allocf(out_f + 4);
    float* stage_v = io.allocf(out_f + 4);
    float* yw = out_on_dev ? y_wave : stage_y;
    float* vw = out_on_dev ? v_wave : stage_v;
    launch_stft(plan, wd, L, hop, T, reinterpret_cast<float2*>(spec), stream);
    if (istft_masked_available(plan, hop) && out_f) {
        separate_api(spec, true, T, tta, batchsize, cropsize, ys, vs, true, /*io_reserved=*/true, yw, vw);
    } else {
        separate_api(spec, true, T, tta, batchsize, cropsize, ys, vs, true, /*io_reserved=*/true);
        launch_istft(plan, reinterpret_cast<const float2*>(ys), hop, T, frames, yw, stream);
        launch_istft(plan, reinterpret_cast<const float2*>(vs), hop, T, frames, vw, stream);
    }
    if (!out_on_dev && out_f) {
        VR_HIP(hipMemcpyAsync(y_wave, yw, out_f * sizeof(float), hipMemcpyDeviceToHost, stream));
        VR_HIP(hipMemcpyAsync(v_wave, vw, out_f * sizeof(float), hipMemcpyDeviceToHost, stream));
    }
    VR_HIP(hipStreamSynchronize(stream));
}

// =====================================================================================================
// unit-test hook: one conv through the MFMA kernel with a single dense source
// =====================================================================================================
void Model::debug_conv(const float* x, int N, int Cin, int H, int W, const float* w_oihw, int Cout, int KS, int stride,
                       int dh, int dw, int flags, const float* aff, float slope, const float* bias, float* out,
                       float* stats_out) {
    DeviceGuard dev_guard(device);
    // flags: bit 0 = fused x2 upsample; bit 1 = give the launch Winograd-domain weights (conv_wino.hip);
    //        bit 2 = `aff`/`slope` describe the EPILOGUE ([Cout][2] folded BatchNorm + activation, eval mode)
    const int up = flags & 1;
    const bool want_wino = (flags & 2) != 0, epi = (flags & 4) != 0;
    const int KK = KS * KS, CoutPad = round_up(Cout, 32);
    const int Hin = up ? 2 * H : H, Win = up ? 2 * W : W;
    const int pad_h = KS == 1 ? 0 : dh, pad_w = KS == 1 ? 0 : dw;
    const int Hout = (Hin + 2 * pad_h - dh * (KS - 1) - 1) / stride + 1;
    const int Wout = (Win + 2 * pad_w - dw * (KS - 1) - 1) / stride + 1;
    std::vector<float> wk((size_t)Cin * KK * CoutPad, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < KK; ++k) wk[((size_t)ci * KK + k) * CoutPad + co] = w_oihw[((size_t)co * Cin + ci) * KK + k];
    float *dx, *dw_, *dout, *daff = nullptr, *dbias = nullptr, *dpart = nullptr;
    const size_t xin = (size_t)N * Cin * H * W, xout = (size_t)N * Cout * Hout * Wout;
    VR_HIP(hipMalloc(&dx, xin * 4)); VR_HIP(hipMalloc(&dw_, wk.size() * 4)); VR_HIP(hipMalloc(&dout, xout * 4));
    VR_HIP(hipMemcpy(dx, x, xin * 4, hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dw_, wk.data(), wk.size() * 4, hipMemcpyHostToDevice));
    const size_t naff = epi ? Cout : Cin;
    if (aff) { VR_HIP(hipMalloc(&daff, naff * 8)); VR_HIP(hipMemcpy(daff, aff, naff * 8, hipMemcpyHostToDevice)); }
    if (bias) { VR_HIP(hipMalloc(&dbias, (size_t)Cout * 4)); VR_HIP(hipMemcpy(dbias, bias, (size_t)Cout * 4, hipMemcpyHostToDevice)); }
    Tensor t;
    t.p = dx; t.N = N; t.C = Cin; t.H = H; t.W = W; t.sH = W; t.sC = (long long)H * W; t.sN = t.sC * Cin;
    if (!epi) { t.aff0 = daff; t.slope = slope; }
    ConvArgs a{};
    a.nsrc = 1; a.src[0] = make_src(t, up != 0, 0); a.c1 = a.c2 = Cin; a.Cin = Cin;
    a.w = dw_; a.bias = dbias; a.Cout = Cout; a.CoutPad = CoutPad;
    if (epi && daff) { a.epi = daff; a.epi_slope = slope; }
    a.bf16 = mfma_mode;
    float* dwino = nullptr;
    void* dwino6 = nullptr;
    void* dx3w = nullptr;
    if (want_wino && mfma_mode == 3 && stride == 1 && (KS == 1 || dh > 1)) {
        // conv_x3d.hip's layers (dilated 3x3, 1x1 at 16 columns): fp16-plane weights only
        VR_HIP(hipMalloc(&dx3w, x3_weights_bytes(Cin, KK, CoutPad)));
        launch_x3h_weights(dw_, dx3w, Cin, KK, CoutPad, stream);
        a.x3w = dx3w;
    } else if (want_wino) {
        VR_CHECK(KS == 3 && stride == 1 && dh == 1 && dw == 1, -2, "Winograd weights exist for 3x3 stride-1 convs only");
        VR_HIP(hipMalloc(&dwino, (size_t)Cin * 16 * CoutPad * 4));
        launch_wino_weights(dw_, dwino, Cin, CoutPad, stream);
        a.wino = dwino;
        if (mfma_mode == 2) {
            VR_HIP(hipMalloc(&dwino6, wino_weights6_bytes(Cin, CoutPad)));
            launch_wino_weights6(dw_, dwino6, Cin, CoutPad, stream);
            a.wino6 = dwino6;
            VR_HIP(hipMalloc(&dx3w, x3_weights_bytes(Cin, 9, CoutPad)));
            launch_x3_weights(dw_, dx3w, Cin, 9, CoutPad, stream);
            a.x3w = dx3w;
        }
        if (mfma_mode == 3) {
            VR_HIP(hipMalloc(&dx3w, x3_weights_bytes(Cin, 9, CoutPad)));
            launch_x3h_weights(dw_, dx3w, Cin, 9, CoutPad, stream);
            a.x3w = dx3w;
        }
    }
    if (!x3d_mode && Win == 16) a.x3w = nullptr;            // option conv_x3d 0: the fp32-pipe kernels for the 16-column layers
    a.dst[0] = ConvDst{dout, (long long)Hout * Wout * Cout, (long long)Hout * Wout, (long long)Wout, 0};
    a.d1 = a.d2 = 1 << 30;
    a.N = N; a.Hout = Hout; a.Wout = Wout; a.Hin = Hin; a.Win = Win; a.pad_h = pad_h; a.pad_w = pad_w;
    const ConvShape shp{KS, stride, dh, dw};
    size_t npt = 0;
    if (stats_out) { npt = conv_part_count(a, shp); VR_HIP(hipMalloc(&dpart, npt * Cout * 8)); a.part = dpart; }
    launch_conv(a, shp, stream);
    VR_HIP(hipStreamSynchronize(stream));
    VR_HIP(hipMemcpy(out, dout, xout * 4, hipMemcpyDeviceToHost));
    if (stats_out) {
        std::vector<float> part(npt * Cout * 2);
        VR_HIP(hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < Cout; ++c) {
            double s1 = 0, s2 = 0;
            for (size_t i = 0; i < npt; ++i) { s1 += part[(i * Cout + c) * 2]; s2 += part[(i * Cout + c) * 2 + 1]; }
            stats_out[2 * c] = (float)s1; stats_out[2 * c + 1] = (float)s2;
        }
    }
    hipFree(dx); hipFree(dw_); hipFree(dout); hipFree(daff); hipFree(dbias); hipFree(dpart); hipFree(dwino); hipFree(dwino6); hipFree(dx3w);
}

}  // namespace vr
