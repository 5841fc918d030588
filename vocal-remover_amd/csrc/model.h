// Host-side model: parameter registry (reference state_dict keys), network topology of
// lib/nets.py / lib/layers.py, forward executor, and the device-resident Separator pipeline.
#pragma once
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

namespace vr {

// Bump allocator over one device slab; `dry` mode only measures.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    void* alloc(size_t bytes) {
        const size_t a = (off + 255) & ~size_t(255);
        off = a + bytes;
        if (off > peak) peak = off;
        if (dry) return reinterpret_cast<void*>(uintptr_t(256));   // non-null dummy, never dereferenced
        if (off > cap) throw Error(-4, "workspace arena overflow (planning bug)");
        return base + a;
    }
    float* allocf(size_t n) { return static_cast<float*>(alloc(n * sizeof(float))); }
    void reset() { off = 0; }
};

enum ParamKind { PK_PLAIN = 0, PK_CONV = 1, PK_LSTM_IH = 2, PK_BUFFER = 3, PK_NBT = 4 };

struct Param {
    std::string key;
    std::vector<int64_t> shape;     // torch shape
    ParamKind kind = PK_PLAIN;
    size_t numel = 0;               // torch numel
    size_t dev_numel = 0;           // device footprint (floats) incl. padding
    size_t off = 0;                 // offset (floats) into its arena
    bool trainable = true;
    // PK_CONV / PK_LSTM_IH layout info: device layout [Cin][KK][CoutPad]
    int Cout = 0, Cin = 0, KK = 1, CoutPad = 0, co_off = 0;
    Param* alias_of = nullptr;      // PK_LSTM_IH reverse half lives inside the forward half's buffer
    float* dev = nullptr;
    int64_t nbt = 0;                // PK_NBT value (host)
    float* grad_override = nullptr; // test hook: gradient slot outside the arena
};

struct BN {
    int C = 0;
    Param *w = nullptr, *b = nullptr, *rm = nullptr, *rv = nullptr, *nbt = nullptr;
    float* affine = nullptr;        // [rows][2] scale, shift (device)
    int bcast = 0;                  // >0: C==1, affine replicated to this many rows
    float* save_mean = nullptr;
    float* save_invstd = nullptr;
};

struct Conv {
    std::string name;
    int Cin = 0, Cout = 0, CoutPad = 0, KS = 3, stride = 1, dh = 1, dw = 1, pad_h = 1, pad_w = 1;
    Param* w = nullptr;
    BN* bn = nullptr;
    float slope = 0.f;              // activation after the BatchNorm
    float* wino = nullptr;          // eval: Winograd-transformed weights [Cin][16][CoutPad] (3x3 stride-1 layers)
    void* wino6 = nullptr;          // mfma_mode 2: the same as three bf16 planes (conv_wino.hip: wino_weights6_kernel)
    void* x3w = nullptr;            // mfma_mode 2: the direct weights as three bf16 planes (conv_x3.hip: x3_weights_kernel)
};

struct LSTMMod {
    Conv squeeze;                   // 1x1 conv 2c -> 1 (+BN+ReLU), run by the thin-conv kernel
    int nin = 0, hid = 0;           // LSTM input size (= nbins at dec2 level), hidden per direction
    Conv proj;                      // W_ih of both directions as one 1x1 conv (nin -> 8*hid), no BN
    Param *b_ih_f = nullptr, *b_hh_f = nullptr, *b_ih_r = nullptr, *b_hh_r = nullptr;   // [4H] each
    Param *whh_f = nullptr, *whh_r = nullptr;
    Conv dense;                     // Linear(2*hid -> nin) as 1x1 conv + BatchNorm1d + ReLU
    Param* dense_b = nullptr;
    float* bias_sum = nullptr;      // eval: b_ih + b_hh of both directions [2][4H], refreshed with the folded affines (saves two launches per LSTM and call)
};

struct BaseNetL {
    std::string prefix;
    int c = 0;
    Conv enc1;
    Conv enc_a[4], enc_b[4];        // enc2..enc5: conv1 (stride 2), conv2
    Conv aspp_pool, aspp_c2, aspp_d[3], aspp_bott;
    float* aspp_aff = nullptr;      // [4*8c][2]: affine tables of conv2..conv5 laid out contiguously
    Conv dec[4];                    // dec4, dec3, dec2, dec1
    LSTMMod lstm;
};

void comm_unique_id(void* out128);                       // ncclGetUniqueId (comm.hip)

// host half of spec_utils.merge_artifacts (lib/spec_utils.py:60-93): per-frame mask minimum -> blend weight
void merge_artifacts_weight(const std::vector<float>& fmin, std::vector<float>& weight, float thres, int min_range,
                            int fade);


class Model {
public:
    Model(int device, int n_fft, int hop, int nout, int nout_lstm);
    ~Model();

    int device, n_fft, hop, nout, nout_lstm, max_bin, output_bin, offset = 64;

    // ---- parameters ----
    std::deque<Param> params;
    std::map<std::string, Param*> by_key;
    void set_param(const std::string& key, const void* host, const int64_t* shape, int ndim);
    void get_param(const std::string& key, void* host, int64_t cap_bytes);
    void set_training(bool t);
    bool training = false;
    bool fwd_only = false;                               // train-mode vr_forward: batch statistics, no tape, no gradient buffers
    bool taping() const { return training && !fwd_only; }

    // ---- forward over host or device input ----
    // x: [B,2,output_bin,T] fp32 magnitudes.  mode 0: forward (full width), 1: predict_mask
    // (offset crop), 2: predict (x*mask, offset crop).  out sized accordingly.
    void forward_api(const float* x, bool x_on_device, int B, int T, int mode, float* out, bool out_on_device);

    // train.validate_epoch body for one batch (train.py:117-127): predict + crop_center(y) + L1, on the device
    void validate_api(const float* X, const float* Y, bool on_dev, int B, int T, float* loss_out);
    // tests only: one kernel of the training path, host pointers (debug.hip)
    void debug_kernel(const std::string& name, const int64_t* dims, int ndims, const float* fp, int nfp,
                      const float* const* in, int nin, float* const* out, int nout);

    // ---- signal path ----
    void stft_api(const float* wave, bool on_dev, long long L, float* spec, bool spec_on_dev);
    void istft_api(const float* spec, bool on_dev, int T, float* wave, bool wave_on_dev);
    // spec [2,bins,T] complex64 -> y_spec, v_spec (same shape)
    void separate_api(const float* spec, bool on_dev, int T, int tta, int batchsize, int cropsize,
                      float* y_spec, float* v_spec, bool out_on_dev, bool io_reserved = false,
                      float* y_wave_d = nullptr, float* v_wave_d = nullptr);
    // wave [2,L] -> y_wave, v_wave [2, hop*(T-1)]: whole inference.py pipeline, device resident
    void separate_wave_api(const float* wave, bool on_dev, long long L, int tta, int batchsize, int cropsize,
                           float* y_wave, float* v_wave, bool out_on_dev);
    void separate_wave_body(const float* wave, bool on_dev, long long L, int tta, int batchsize, int cropsize,
                            float* y_wave, float* v_wave, bool out_on_dev);

    // ---- debug / test hooks ----
    void debug_conv(const float* x, int N, int Cin, int H, int W, const float* w_oihw, int Cout, int KS, int stride,
                    int dh, int dw, int up, const float* aff, float slope, const float* bias, float* out,
                    float* stats_out);
    bool record_taps = false;
    std::map<std::string, Tensor> taps;
    int64_t get_tap(const std::string& name, float* host, int64_t cap_floats, int64_t* shape4);

    // ---- profiling ----
    bool profiling = false;
    LaunchProfiler* launch_prof = nullptr;               // profile.hip: every VR_LAUNCH of a profiled step, timed on its own stream
    std::string profile_report;                          // per-kernel-name totals of the last profiled step (vr_profile_report)
    void profile_begin();
    void augment_api(const float* Xc, const float* yc, const float* Xi, const float* yi, const void* desc, const float* rw,
                     int B, int T, int bins, bool in_on_dev, float* Xmag, float* ymag, bool out_on_dev);
    char* aug_buf = nullptr; size_t aug_cap = 0;         // staging of the training input pipeline
    bool train_wino = true;                              // vr_set_option("train_winograd"): Winograd kernels in train mode
    // vr_set_option("mfma_mode"): how the 3x3 stride-1 convs multiply.
    //   2 (round-3 default) = fp32 products as six bf16 products of three-way split operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation
    //       (conv_x3.hip: direct conv, error vs fp64 = an fp32 direct convolution's; forward and data-gradient convs -- weight
    //       gradients stay on the fp32 MFMA);
    //   0 = v_mfma_f32_32x32x2_f32 throughout (Winograd F(2x2,3x3) / direct kernels, round-1/2 default);
    //   1 = operands rounded to bf16 (configs[4] arithmetic; "mfma_bf16" 1 is the same);
    //   3 (default since round 4) = the layers of mode 2 with fp32-grade products from THREE fp16 products of two-way split,
    //       power-of-two scaled operands on v_mfma_f32_32x32x16_f16 (conv_x3h.hip: 14 instead of 27 matrix instructions per 8-channel
    //       chunk; measured error vs fp64 at or below mode 2's).
    int mfma_mode = 3;
    bool x3_mode() const { return mfma_mode == 2 || mfma_mode == 3; }
    int default_mfma_mode = 3;                           // (VR_MFMA_MODE overrides; "mfma_mode" -1 / "mfma_bf16" 0 return to it)
    bool serial = false;                                 // vr_set_option("serial_exec"): no lanes / side streams (tests: race detector)
    void set_option(const std::string& name, int value);
    void reset_adam_state();
    void profile_end(double* conv_ms, double* conv_flops, double* conv_bytes, int* launches);

    std::chrono::steady_clock::time_point enq_t1; bool enq_have = false;   // VR_ENQ_TIMING diagnostics (model.hip: separate_wave_body)
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // eval mode: the high-band chain runs here
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool band_fork_active = false;                       // run_net: the side stream currently carries the high-band chain
    bool gs_clear_pending = false;                       // training: the gs memset runs on lanes[0].main beside the forward
    bool wgrad_on_side = false;                          // training: weight gradients in flight on side_stream
    // eval mode, second lane: the crops of a batch are independent, so separate() runs the two halves
    // of the batch as two concurrent chains (own streams, own workspace); the tail of one chain's
    // kernels and its memory-bound kernels (upsample, LSTM, copies) overlap the other chain's convs.
    struct Lane { hipStream_t main = nullptr, side = nullptr; hipEvent_t fork = nullptr, join = nullptr, start = nullptr, done = nullptr; Arena ws; };
    std::vector<Lane> lanes;                             // the additional lanes (VR_LANES - 1, default 1)
    void swap_lane(int i);

private:
    // arenas
    float* p_arena = nullptr; size_t p_floats = 0;      // trainable parameters (kernel layouts)
    float* b_arena = nullptr; size_t b_floats = 0;      // buffers (running stats), affine tables, saves
    Arena ws;                                            // activations workspace
    Arena io;                                            // persistent I/O staging (spec, mask, waves)
    void ensure_ws(size_t bytes);
    int plan_B = -1, plan_T = -1; bool plan_training = false; size_t plan_peak = 0;    // memo of plan_and_reserve
    void ensure_io(size_t bytes);

    std::deque<BN> bns;
    std::vector<BN*> bn_list;
    std::vector<Conv*> wino_list;                        // 3x3 stride-1 layers (conv_wino.hip)
    struct PendingConv { ConvArgs a; ConvShape shp; double flops, bytes; bool stats; BNFinalizeArgs fin; BN* bn; };
    std::vector<PendingConv>* conv_sink = nullptr;       // set: run_conv hands the launch to its caller instead of launching (ASPP branch group)
    int x3d_mode = 2;                                    // option "conv_x3d": 0 off, 1 single launches, 2 + the ASPP branch group
    std::vector<Conv*> x3d_list;                         // the ASPP branch convs conv_x3d.hip takes in mfma_mode 3: dilated 3x3, conv2 (1x1)
    float* wino_arena = nullptr;
    float* winot_arena = nullptr;                        // training: Winograd copies of the flipped/transposed weights
    std::map<const Param*, float*> winot_of;
    char* wino6_arena = nullptr;                         // mfma_mode 2: bf16-plane copies of both
    char* winot6_arena = nullptr;
    std::map<const Param*, void*> winot6_of;
    char* x3_arena = nullptr;                            // mfma_mode 2: bf16-plane copies of the direct 3x3 stride-1 weights (conv_x3.hip)
    char* x3t_arena = nullptr;                           //              and of their flipped / transposed forms (data gradient)
    std::map<const Param*, void*> x3t_of;
    std::map<const Param*, void*> x3dt_of;                // the same for x3d_list (fp16 planes, valid in mfma_mode 3 only)
    // deferred weight-gradient slab sums of one backward pass (launch_wgrad_reduce_batched, round 6)
    std::vector<WgReduceDesc> wred_host, wred_sent;
    WgReduceDesc* wred_dev = nullptr;
    size_t wred_cap = 0;
    void flush_wgrad_sums();
    struct X3Batch { std::vector<X3WDesc> host; X3WDesc* dev = nullptr; long long max_elems = 0; };
    X3Batch xb_fwd, xb_bwd;
    void run_x3_batch(X3Batch& b, std::vector<X3WDesc>& descs);
    void refresh_wino(bool with_dgrad);
    // batched refresh: descriptor tables (host copy + device copy, re-uploaded only when a pointer changed)
    struct WinoBatch { std::vector<WinoWDesc> host; WinoWDesc* dev = nullptr; long long max_elems = 0; };
    WinoBatch wb_fwd, wb_bwd, wb_fwd6, wb_bwd6;
    void run_wino_batch(WinoBatch& b, std::vector<WinoWDesc>& descs, bool split6);
    BNFoldDesc* d_fold = nullptr;
    bool affine_dirty = true;
    void fold_eval_affines();

    BaseNetL nets_[5];
    Conv tail1, tail2;                                   // stg{1,2}_low_band_net.1
    Param *out_w = nullptr, *aux_out_w = nullptr;

    FFTPlan plan{};

    // builders
    Param* add_param(const std::string& key, std::vector<int64_t> shape, ParamKind kind, bool trainable);
    BN* add_bn(const std::string& prefix, int C, int bcast);
    void build_cba(Conv& L, const std::string& prefix, int nin, int nout_, int ks, int stride, int pad_h, int pad_w,
                   int dh, int dw, float slope);
    void build_basenet(BaseNetL& B, const std::string& prefix, int nin, int c, int nin_lstm, int nout_lstm_);
    void finalize_layout();

    // executor
    // `plain`: training only -- dense [N][C][Hv][Wv] copy of the values the conv sees (BatchNorm, activation,
    // dropout, upsample / broadcast applied), so that forward and weight gradient load it without arithmetic
    struct SrcSpec { Tensor t; bool up = false; int bcastH = 0; float* plain = nullptr; };
    // ---- training tape: forward order == topological order, backward walks it in reverse ----
    enum TapeKind { TK_CONV, TK_AVGPOOL, TK_SQUEEZE, TK_LSTM, TK_DENSE_ACT };
    struct TapeRec {
        TapeKind kind;
        Conv* L = nullptr;
        std::vector<SrcSpec> srcs;
        Tensor out;                  // raw output (+ g)
        int N = 0;
        bool batch_as_h = false;
        const float* bias = nullptr;
        Param* bias_param = nullptr;           // dense bias (gradient = channel sums of dz)
        LSTMMod* M = nullptr;                  // TK_LSTM / TK_SQUEEZE / TK_DENSE_ACT
        Tensor aux;                            // kind-specific second tensor
        float* save = nullptr;                 // LSTM gate/cell record
        float* buf0 = nullptr;                 // kind-specific buffers
        float* buf1 = nullptr;
        int chain = 0;                         // 1: high-band chain of stages 1-2 (independent of the low-band chain)
    };
    std::vector<TapeRec> tape;
    Arena gs;                                            // gradients of activations
    // First writer stores (round 4): the gradient buffer of a conv output is owned by that tensor alone -- every consumer's backward
    // addresses the whole tensor -- so the first backward writer STORES and the later ones accumulate: no zero fill, and the first
    // data-gradient epilogue writes instead of read-modify-writing zeros.  Buffers that are written through partial views (the
    // stage outputs aux1 / aux2: full tensor by stage 3, band halves by stage 2) or by kernels that only accumulate (pooled / LSTM
    // internals) stay zero-initialised: their byte ranges inside `gs` are recorded by the planning dry run (gs_zero).
    std::vector<std::pair<size_t, size_t>> gs_zero, gs_zero_plan;      // (offset, bytes); _plan = the dry run's list, what gets cleared
    std::map<const float*, bool> g_fresh;                // plain buffers of this forward: true until their first backward writer
    float* galloc(size_t n, bool plain);
    bool g_first(const float* g);                        // true exactly once per plain buffer: that writer must store, not accumulate
    void clear_gs_zero_ranges(hipStream_t st);
    float* g_arena = nullptr;                            // gradients of parameters (mirrors p_arena)
    float* m_arena = nullptr; float* v_arena = nullptr;  // Adam moments
    float* wt_arena = nullptr; size_t wt_floats = 0;     // flipped/transposed conv weights for dgrad
    FlipDesc* d_flip = nullptr; int n_flip = 0;
    std::map<const Param*, float*> wt_of;
    float* s2w_arena = nullptr;                          // parity-class weights of the stride-2 convs' data gradients
    std::map<const Param*, float*> s2w_of;
    std::vector<Conv*> s2_list;
    S2WDesc* d_s2w = nullptr;                            // descriptor table of launch_s2_class_weights (one launch for all of them)
    long long s2w_max = 0;
    long long adam_step = 0;
    float* grad_of(const Param* p) { return p->grad_override ? p->grad_override : g_arena + (p->dev - p_arena); }
public:
    void debug_conv_bwd(const float* x, int N, int Cin, int H, int W, const float* w_oihw, int Cout, int KS, int stride,
                        int dh, int dw, int up, const float* aff, float slope, const float* dz, float* dx_out,
                        float* dw_out);
private:
    void ensure_train_state();
    void backward();
    void bwd_conv(TapeRec& r);
    void bwd_bn_of(const Tensor& out, Conv& L);
    std::vector<float> dropout_host;                     // injected keep-masks [5][N][8*nout] or empty
    int dropout_mode = 1;                                // 0 off, 1 native RNG (default: nn.Dropout2d is active in train mode), 2 injected
    unsigned long long dropout_seed = 0x5DEECE66Dull;
    unsigned long long train_calls = 0;                  // counter of train-mode forwards: a fresh dropout draw for each (lib/layers.py:90)
    void prepare_dropout(int B);                         // fills dropout_dev for a train-mode forward of batch B
    float* dropout_buf = nullptr; size_t dropout_cap = 0;
public:
    // train.py:77-96: forward (train mode) + L1 loss + backward; gradients accumulate in the arena.
    void train_fwd_bwd_api(const float* X, const float* Y, bool on_dev, int B, int T, int accumulation_steps,
                           float* loss_out, float* mask_out, bool mask_on_dev);
    // the same step as two calls (autograd split): forward that keeps the graph, backward from dLoss/dmask
    void forward_train_api(const float* X, bool on_dev, int B, int T, float* mask_out, bool mask_on_dev);
    void backward_api(const float* dmask, bool on_dev);
    bool graph_valid = false; int graph_B = 0, graph_T = 0; Tensor graph_f3; float* graph_mask = nullptr;
    int64_t graph_gen = 0;                               // bumped by every vr_forward_train: the caller's backward names the graph it wants
    void param_arena(float** ptr, int64_t* numel) { *ptr = p_arena; *numel = (int64_t)p_floats; }
    void mark_params_dirty() { affine_dirty = true; }
    void adam_step_api(double lr, double b1, double b2, double eps, double grad_scale);
    void zero_grad_api();
    void adam_state(float* m_host, float* v_host, int64_t numel, int64_t* step, bool set);
    void get_grad(const std::string& key, float* host, int64_t cap_bytes);
    void set_dropout(int mode, unsigned long long seed, const float* masks, int B);
    void grad_arena(float** ptr, int64_t* numel);
    // ---- data-parallel exchange (comm.hip): RCCL on the handle's stream ----
    void comm_init(int rank, int world, const void* id128);
    void comm_destroy();
    void allreduce_grads(int wire_dtype);
    void broadcast_params(int root, bool with_optimizer);
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    void* wire_buf = nullptr;                            // bf16 copy of the gradient bucket (wire_dtype 1)
private:
    void build_fwd_args(Conv& L, const std::vector<SrcSpec>& srcs, int N, bool batch_as_h, ConvArgs& a);
    Tensor run_conv(Conv& L, const std::vector<SrcSpec>& srcs, int N, const Tensor* out_view, const float* bias,
                    bool batch_as_h);
    template <class F> void for_each_conv(F&& f);
    Tensor run_basenet(BaseNetL& B, const std::vector<SrcSpec>& in, int N, const Tensor* out_view);
    Tensor run_lstm(LSTMMod& M, const Tensor& h);
    SrcSpec upsampled(const Tensor& t);
    Tensor run_net(const Tensor& x);                     // -> stg3 dec1 output (raw + affine)
    void tap(const std::string& name, const Tensor& t);
    bool dry = false;
    const float* dropout_dev = nullptr;                  // [5][N][Cmax] keep-masks (training), or null
    void plan_and_reserve(int B, int T, size_t extra_bytes);
    void record_begin(int kind, double flops);
    void record_note(double bytes, const char* tag);
    double conv_alg_bytes(const Conv& L, const ConvArgs& a, int N, bool batch_as_h) const;
    void record_end();
};

}  // namespace vr
