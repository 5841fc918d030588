// extern "C" surface of libvr_mi355.so (include/vr_mi355.h).  Exceptions stop here.
#include "../../include/vr_mi355.h"

#include <cstring>
#include <new>

#include "model.h"

namespace vr {
void resample_api(int device, const float* x, int channels, long long n_in, int sr_in, int sr_out, float* y, long long n_out);
void xcorr_argmax_api(int device, const float* a, long long na, const float* b, long long nb, long long* argmax_out);
}  // namespace vr

struct vr_model {
    vr::Model m;
    vr_model(int d, int n, int h, int o, int l) : m(d, n, h, o, l) {}
};

static thread_local std::string g_err;

template <class F>
static int guard(F&& f) {
    try {
        f();
        return VR_OK;
    } catch (const vr::Error& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_err = "host allocation failed";
        return VR_ERR_OOM;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VR_ERR_UNKNOWN;
    } catch (...) {
        g_err = "unknown error";
        return VR_ERR_UNKNOWN;
    }
}

#define NEED(h)                                                  \
    if (!(h)) {                                                  \
        g_err = "null handle";                                   \
        return VR_ERR_BAD_ARGUMENT;                              \
    }

extern "C" {

const char* vr_last_error(void) { return g_err.c_str(); }

int vr_create(int device, int n_fft, int hop_length, int nout, int nout_lstm, vr_handle* out) {
    if (!out) { g_err = "null out pointer"; return VR_ERR_BAD_ARGUMENT; }
    *out = nullptr;
    return guard([&] {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            throw vr::Error(VR_ERR_HIP, "no HIP device visible: libvr_mi355 has no CPU fallback");
        if (device < 0 || device >= count) throw vr::Error(VR_ERR_BAD_ARGUMENT, "device index out of range");
        *out = new vr_model(device, n_fft, hop_length, nout, nout_lstm);
    });
}

int vr_destroy(vr_handle h) {
    NEED(h);
    return guard([&] { delete h; });
}

int vr_num_params(vr_handle h) {
    if (!h) return VR_ERR_BAD_ARGUMENT;
    return (int)h->m.params.size();
}

int vr_param_info(vr_handle h, int index, char* key_buf, int key_cap, int64_t* shape4, int* ndim, int* is_int64,
                  int* trainable) {
    NEED(h);
    return guard([&] {
        VR_CHECK(index >= 0 && index < (int)h->m.params.size(), VR_ERR_BAD_ARGUMENT, "param index out of range");
        const vr::Param& p = h->m.params[index];
        if (key_buf) {
            VR_CHECK((int)p.key.size() + 1 <= key_cap, VR_ERR_BAD_ARGUMENT, "key buffer too small");
            std::memcpy(key_buf, p.key.c_str(), p.key.size() + 1);
        }
        if (ndim) *ndim = (int)p.shape.size();
        if (shape4)
            for (size_t i = 0; i < p.shape.size() && i < 4; ++i) shape4[i] = p.shape[i];
        if (is_int64) *is_int64 = p.kind == vr::PK_NBT;
        if (trainable) *trainable = p.trainable;
    });
}

int vr_set_param(vr_handle h, const char* key, const void* host, const int64_t* shape, int ndim) {
    NEED(h);
    return guard([&] {
        VR_CHECK(key && host && (shape || ndim == 0), VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.set_param(key, host, shape, ndim);
    });
}

int vr_get_param(vr_handle h, const char* key, void* host, int64_t capacity_bytes) {
    NEED(h);
    return guard([&] {
        VR_CHECK(key && host, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.get_param(key, host, capacity_bytes);
    });
}

int vr_set_mode(vr_handle h, int training) {
    NEED(h);
    return guard([&] { h->m.set_training(training != 0); });
}

int vr_set_option(vr_handle h, const char* name, int value) {
    NEED(h);
    return guard([&] {
        VR_CHECK(name, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.set_option(name, value);
    });
}

int vr_forward(vr_handle h, const float* x, int x_on_device, int B, int T, int mode, float* out, int out_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(x && out, VR_ERR_BAD_ARGUMENT, "null argument");
        VR_CHECK(mode >= 0 && mode <= 2, VR_ERR_BAD_ARGUMENT, "mode must be 0 (forward), 1 (predict_mask) or 2 (predict)");
        h->m.forward_api(x, x_on_device != 0, B, T, mode, out, out_on_device != 0);
    });
}

int vr_stft(vr_handle h, const float* wave, int wave_on_device, int64_t L, float* spec, int spec_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(wave && spec, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.stft_api(wave, wave_on_device != 0, L, spec, spec_on_device != 0);
    });
}

int vr_istft(vr_handle h, const float* spec, int spec_on_device, int T, float* wave, int wave_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(spec && wave, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.istft_api(spec, spec_on_device != 0, T, wave, wave_on_device != 0);
    });
}

int vr_separate(vr_handle h, const float* spec, int spec_on_device, int T, int tta, int batchsize, int cropsize,
                float* y_spec, float* v_spec, int out_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(spec && y_spec && v_spec, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.separate_api(spec, spec_on_device != 0, T, tta, batchsize, cropsize, y_spec, v_spec, out_on_device != 0);
    });
}

int vr_separate_wave(vr_handle h, const float* wave, int wave_on_device, int64_t L, int tta, int batchsize,
                     int cropsize, float* y_wave, float* v_wave, int out_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(wave && y_wave && v_wave, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.separate_wave_api(wave, wave_on_device != 0, L, tta, batchsize, cropsize, y_wave, v_wave, out_on_device != 0);
    });
}

int vr_train_step(vr_handle h, const float* X, const float* y, int on_device, int B, int T, int accumulation_steps,
                  float* loss_out, float* mask_out, int mask_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(X && y, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.train_fwd_bwd_api(X, y, on_device != 0, B, T, accumulation_steps, loss_out, mask_out, mask_on_device != 0);
    });
}

int vr_forward_train(vr_handle h, const float* X, int on_device, int B, int T, float* mask_out, int mask_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(X && mask_out, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.forward_train_api(X, on_device != 0, B, T, mask_out, mask_on_device != 0);
    });
}

int vr_backward(vr_handle h, const float* dmask, int on_device) {
    NEED(h);
    return guard([&] { h->m.backward_api(dmask, on_device != 0); });
}

int vr_graph_generation(vr_handle h, int64_t* generation, int* valid) {
    NEED(h);
    return guard([&] {
        VR_CHECK(generation, VR_ERR_BAD_ARGUMENT, "null argument");
        *generation = h->m.graph_gen;
        if (valid) *valid = h->m.graph_valid ? 1 : 0;
    });
}

int vr_param_arena(vr_handle h, float** device_ptr, int64_t* numel) {
    NEED(h);
    return guard([&] {
        VR_CHECK(device_ptr && numel, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.param_arena(device_ptr, numel);
    });
}

int vr_augment_batch(vr_handle h, const float* X, const float* y, const float* X_mix, const float* y_mix, const vr_aug* desc,
                     const float* reduction_weight, int B, int T, int bins, int in_on_device, float* X_mag, float* y_mag,
                     int out_on_device) {
    NEED(h);
    return guard([&] {
        VR_CHECK(X && y && desc && X_mag && y_mag, VR_ERR_BAD_ARGUMENT, "null argument");
        bool mix = false, red = false;
        for (int b = 0; b < B; ++b) {
            mix = mix || (desc[b].flags & 8);
            red = red || (desc[b].flags & (1 | 16));
        }
        VR_CHECK(!mix || (X_mix && y_mix), VR_ERR_BAD_ARGUMENT, "mixup flagged but no partner crops given");
        VR_CHECK(!red || reduction_weight, VR_ERR_BAD_ARGUMENT, "vocal reduction flagged but no reduction_weight given");
        h->m.augment_api(X, y, X_mix, y_mix, desc, reduction_weight, B, T, bins, in_on_device != 0, X_mag, y_mag,
                         out_on_device != 0);
    });
}

int vr_adam_step(vr_handle h, double lr, double b1, double b2, double eps, double grad_scale) {
    NEED(h);
    return guard([&] { h->m.adam_step_api(lr, b1, b2, eps, grad_scale); });
}

int vr_get_adam_state(vr_handle h, float* exp_avg, float* exp_avg_sq, int64_t numel, int64_t* step) {
    NEED(h);
    return guard([&] {
        VR_CHECK(exp_avg && exp_avg_sq && step, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.adam_state(exp_avg, exp_avg_sq, numel, step, false);
    });
}

int vr_set_adam_state(vr_handle h, const float* exp_avg, const float* exp_avg_sq, int64_t numel, int64_t step) {
    NEED(h);
    return guard([&] {
        VR_CHECK(exp_avg && exp_avg_sq, VR_ERR_BAD_ARGUMENT, "null argument");
        int64_t st = step;
        h->m.adam_state(const_cast<float*>(exp_avg), const_cast<float*>(exp_avg_sq), numel, &st, true);
    });
}

int vr_zero_grad(vr_handle h) {
    NEED(h);
    return guard([&] { h->m.zero_grad_api(); });
}

int vr_get_grad(vr_handle h, const char* key, float* host, int64_t capacity_bytes) {
    NEED(h);
    return guard([&] {
        VR_CHECK(key && host, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.get_grad(key, host, capacity_bytes);
    });
}

int vr_set_dropout(vr_handle h, int mode, uint64_t seed, const float* masks, int B) {
    NEED(h);
    return guard([&] { h->m.set_dropout(mode, seed, masks, B); });
}

int vr_grad_arena(vr_handle h, float** device_ptr, int64_t* numel) {
    NEED(h);
    return guard([&] {
        VR_CHECK(device_ptr && numel, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.grad_arena(device_ptr, numel);
    });
}

int vr_validate_step(vr_handle h, const float* X, const float* y, int on_device, int B, int T, float* loss_out) {
    NEED(h);
    return guard([&] {
        VR_CHECK(X && y, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.validate_api(X, y, on_device != 0, B, T, loss_out);
    });
}

int vr_comm_unique_id(void* id_out) {
    return guard([&] {
        VR_CHECK(id_out, VR_ERR_BAD_ARGUMENT, "null argument");
        vr::comm_unique_id(id_out);
    });
}

int vr_comm_init(vr_handle h, int rank, int world_size, const void* id) {
    NEED(h);
    return guard([&] { h->m.comm_init(rank, world_size, id); });
}

int vr_comm_destroy(vr_handle h) {
    NEED(h);
    return guard([&] { h->m.comm_destroy(); });
}

int vr_allreduce_grads(vr_handle h, int wire_dtype) {
    NEED(h);
    return guard([&] { h->m.allreduce_grads(wire_dtype); });
}

int vr_broadcast_params(vr_handle h, int root, int with_optimizer) {
    NEED(h);
    return guard([&] { h->m.broadcast_params(root, with_optimizer != 0); });
}

int vr_debug_kernel(vr_handle h, const char* name, const int64_t* dims, int ndims, const float* fparams, int nfparams,
                    const float* const* inputs, int ninputs, float* const* outputs, int noutputs) {
    NEED(h);
    return guard([&] {
        VR_CHECK(name && dims && inputs && outputs, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.debug_kernel(name, dims, ndims, fparams, nfparams, inputs, ninputs, outputs, noutputs);
    });
}

static void need_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw vr::Error(VR_ERR_HIP, "no HIP device visible: libvr_mi355 has no CPU fallback");
    if (device < 0 || device >= count) throw vr::Error(VR_ERR_BAD_ARGUMENT, "device index out of range");
}

int vr_resample(int device, const float* x, int channels, int64_t n_in, int sr_in, int sr_out, float* y, int64_t n_out) {
    return guard([&] {
        VR_CHECK(x && y, VR_ERR_BAD_ARGUMENT, "null argument");
        need_device(device);
        vr::resample_api(device, x, channels, n_in, sr_in, sr_out, y, n_out);
    });
}

int vr_xcorr_argmax(int device, const float* a, int64_t na, const float* b, int64_t nb, int64_t* argmax_out) {
    return guard([&] {
        VR_CHECK(a && b && argmax_out, VR_ERR_BAD_ARGUMENT, "null argument");
        need_device(device);
        long long best = 0;
        vr::xcorr_argmax_api(device, a, na, b, nb, &best);
        *argmax_out = best;
    });
}

int vr_profile_begin(vr_handle h) {
    NEED(h);
    return guard([&] { h->m.profile_begin(); });
}

int vr_profile_end(vr_handle h, double* conv_ms, double* conv_flops, int* conv_launches, double* conv_bytes) {
    NEED(h);
    return guard([&] { h->m.profile_end(conv_ms, conv_flops, conv_bytes, conv_launches); });
}

int64_t vr_profile_report(vr_handle h, char* buf, int64_t capacity) {
    if (!h) { g_err = "null handle"; return VR_ERR_BAD_ARGUMENT; }
    const std::string& r = h->m.profile_report;
    if (buf && capacity > 0) {
        const size_t n = std::min<size_t>(r.size(), (size_t)capacity - 1);
        std::memcpy(buf, r.data(), n);
        buf[n] = 0;
    }
    return (int64_t)r.size() + 1;
}

int vr_debug_conv2d(vr_handle h, const float* x, int N, int Cin, int H, int W, const float* w, int Cout, int ksize,
                    int stride, int dil_h, int dil_w, int upsample, const float* affine, float slope, const float* bias,
                    float* out, float* stats_out) {
    NEED(h);
    return guard([&] {
        VR_CHECK(x && w && out, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.debug_conv(x, N, Cin, H, W, w, Cout, ksize, stride, dil_h, dil_w, upsample, affine, slope, bias, out, stats_out);
    });
}

int vr_debug_conv2d_backward(vr_handle h, const float* x, int N, int Cin, int H, int W, const float* w, int Cout,
                             int ksize, int stride, int dil_h, int dil_w, int upsample, const float* affine, float slope,
                             const float* dz, float* dx_out, float* dw_out) {
    NEED(h);
    return guard([&] {
        VR_CHECK(x && w && dz && dx_out && dw_out, VR_ERR_BAD_ARGUMENT, "null argument");
        h->m.debug_conv_bwd(x, N, Cin, H, W, w, Cout, ksize, stride, dil_h, dil_w, upsample, affine, slope, dz, dx_out, dw_out);
    });
}

int vr_debug_merge_artifacts_weight(const float* frame_min, int T, float thres, int min_range, int fade_size,
                                    float* weight_out) {
    return guard([&] {
        VR_CHECK(frame_min && weight_out && T > 0, VR_ERR_BAD_ARGUMENT, "null argument");
        std::vector<float> f(frame_min, frame_min + T), w;
        vr::merge_artifacts_weight(f, w, thres, min_range, fade_size);
        std::memcpy(weight_out, w.data(), (size_t)T * sizeof(float));
    });
}

int vr_debug_record_taps(vr_handle h, int enable) {
    NEED(h);
    return guard([&] { h->m.record_taps = enable != 0; if (!enable) h->m.taps.clear(); });
}

int64_t vr_debug_get_tap(vr_handle h, const char* name, float* host, int64_t capacity_floats, int64_t* shape4) {
    if (!h || !name) { g_err = "null argument"; return VR_ERR_BAD_ARGUMENT; }
    int64_t n = 0;
    const int rc = guard([&] { n = h->m.get_tap(name, host, capacity_floats, shape4); });
    return rc == VR_OK ? n : rc;
}

}  // extern "C"
