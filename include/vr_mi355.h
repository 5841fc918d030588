/* libvr_mi355.so -- C ABI of the MI355X-native vocal-remover hot path.
 *
 * The reference (tsurumeso/vocal-remover @ 2024_08_07) has no FFI layer: its boundary is the Python
 * API that inference.py / train.py / pseudo.py call.  Each entry point below names the reference
 * interface it replaces; vocal-remover_amd/ binds them with ctypes and re-exposes the reference's
 * class / function names.
 *
 * Conventions
 *   - every function returns 0 on success or a negative vr_status; the message of the last failure
 *     on the calling thread is vr_last_error().  Nothing aborts, nothing calls back into the host.
 *   - `*_on_device` flags say whether a data pointer is host memory (numpy / torch CPU,
 *     C-contiguous) or device memory on the handle's GPU.  The caller owns every pointer it passes;
 *     the library copies.  The library owns all device memory it allocates (weights, workspace).
 *   - a handle is bound to one GPU and one HIP stream and is not thread-safe; use one handle per
 *     process per GPU.  Calls return after the handle's stream has drained.
 *   - tensors are fp32; spectrograms are complex64 stored as interleaved (re, im) floats with the
 *     reference's layout [2, n_fft/2+1, frames]; waves are [2, samples].
 */
#ifndef VR_MI355_H
#define VR_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vr_model* vr_handle;

enum vr_status {
    VR_OK = 0,
    VR_ERR_UNKNOWN = -1,
    VR_ERR_BAD_ARGUMENT = -2,   /* shape / key / flag errors                                      */
    VR_ERR_HIP = -3,            /* a HIP runtime call failed                                       */
    VR_ERR_OOM = -4,            /* workspace planning / allocation failure                         */
    VR_ERR_CROP_CENTER = -5,    /* reference ValueError of spec_utils.crop_center (spec_utils.py:15) */
    VR_ERR_EMPTY_MASK = -6,     /* reference `assert mask.size()[3] > 0` (nets.py:129,139)         */
    VR_ERR_INDEX = -7,          /* reference IndexError in merge_artifacts (spec_utils.py:65, empty idx) */
    VR_ERR_COMM = -8            /* RCCL could not be loaded, or a collective failed                        */
};

const char* vr_last_error(void);

/* nets.CascadedNet(n_fft, hop_length, nout=32, nout_lstm=128)        lib/nets.py:46-80
 * (is_complex=False, the only configuration any reference caller uses).                           */
int vr_create(int device, int n_fft, int hop_length, int nout, int nout_lstm, vr_handle* out);
int vr_destroy(vr_handle h);

/* nn.Module.state_dict() / load_state_dict()                  inference.py:131, train.py:209,290
 * One call per state-dict key (689 keys for the default net), torch shapes and layouts
 * (conv OIHW, LSTM [4H, I], gate order i,f,g,o).  num_batches_tracked entries are int64.        */
int vr_num_params(vr_handle h);
int vr_param_info(vr_handle h, int index, char* key_buf, int key_cap, int64_t* shape4, int* ndim,
                  int* is_int64, int* trainable);
int vr_set_param(vr_handle h, const char* key, const void* host, const int64_t* shape, int ndim);
int vr_get_param(vr_handle h, const char* key, void* host, int64_t capacity_bytes);

/* nn.Module.train() / eval()                                          inference.py:52, train.py:69,109 */
int vr_set_mode(vr_handle h, int training);
/* Numerical options.  "train_winograd" (default 1): train-mode forward and data-gradient 3x3 stride-1
 * convolutions may use the transformed-weight kernels (mfma_mode 0: Winograd F(2x2,3x3), fp32, rounding differs from the direct
 * kernel by ~1e-6 relative per conv -- the same class of difference as cuDNN's algorithm choice in the reference; mfma_mode 3 / 2:
 * the split direct kernels conv_x3h.hip / conv_x3.hip); 0 = the fp32 direct kernels only.  Eval follows "mfma_mode" alone.
 * "adam_reset": zero the Adam moments and the step counter (what constructing a new
 * torch.optim.Adam does; train.py:215-218).  "serial_exec" (default 0): 1 = every kernel on the handle's one
 * stream, no lanes / side streams (tests: results must not depend on the concurrent executor).
 * "mfma_mode" (default 3): how the 3x3 stride-1 convolutions (84 % of the multiply-adds) form their products.
 *   3 = fp32-grade products from THREE fp16 products: every operand is scaled by an exact power of two (weights per output channel,
 *       pixels per workgroup tile and 8-channel chunk, the accumulators follow) and written as two fp16 numbers (22 significand
 *       bits), a*b ~= a1b1 + a1b2 + a2b1 with fp32 accumulation on v_mfma_f32_32x32x16_f16 -- conv_x3h.hip, forward and data
 *       gradient, 14 matrix instructions per 8-channel chunk; measured error against fp64 at or below mode 2's and an fp32 direct
 *       convolution's (tests), any fp32 dynamic range (2^+-100 scales, subnormals) included.
 *   2 = fp32 products assembled from six bf16 products of three-way split operands (x = x1 + x2 + x3 exactly,
 *       a*b = a1b1 + a2b1 + a1b2 + a2b2 + a1b3 + a3b1, fp32 accumulation) on v_mfma_f32_32x32x16_bf16 -- the direct kernel
 *       conv_x3.hip, forward and data gradient; error against fp64 = an fp32 direct convolution's (tests); in eval the
 *       decoder's bilinear x2 is fused into the full-resolution layers.  Storage, the other kernels and the weight gradients
 *       are unchanged (fp32).
 *   0 = v_mfma_f32_32x32x2_f32 (fp32 operands) throughout: Winograd F(2x2,3x3) / direct kernels (the round-1/2 default).
 *   1 = bf16 MFMA operands (configs[4] arithmetic): the Winograd convolutions (forward, data gradient, weight gradient) and
 *       the 1x1 weight-gradient GEMM round their operands to bf16 (RNE, in registers); accumulation, every stored tensor, the
 *       master weights and Adam stay fp32.   -1 = back to the handle's default (3, or VR_MFMA_MODE).
 * "mfma_bf16": 1 = "mfma_mode" 1; 0 = back to the handle's default mode.
 * "params_dirty": the parameter arena was written from outside (vr_param_arena).
 * "conv_x3d" (round 6, "mfma_mode" 3 only): the 16-column layers of 256-frame crops -- the ASPP branch convs (lib/layers.py:74-85) and
 *   Encoder.conv2 of enc5 -- on the fp16 matrix pipe: 2 (default, also -1) with the four ASPP branches of a module in one launch,
 *   1 one launch per conv, 0 the fp32-pipe kernels.   */
int vr_set_option(vr_handle h, const char* name, int value);

/* CascadedNet.forward (mode 0) / predict_mask (mode 1) / predict (mode 2)   lib/nets.py:82-141
 * x:   [B, 2, n_fft/2+1, T] fp32 magnitudes
 * out: mode 0 [B,2,bins,T];  modes 1,2 [B,2,bins,T-128]                                          */
int vr_forward(vr_handle h, const float* x, int x_on_device, int B, int T, int mode, float* out,
               int out_on_device);

/* spec_utils.wave_to_spectrogram(wave, hop_length, n_fft)          lib/spec_utils.py:26-31
 * wave [2, L] -> spec [2, bins, 1 + L/hop] complex64                                              */
int vr_stft(vr_handle h, const float* wave, int wave_on_device, int64_t L, float* spec, int spec_on_device);

/* spec_utils.spectrogram_to_wave(spec, hop_length)                 lib/spec_utils.py:157-165
 * spec [2, bins, T] -> wave [2, hop*(T-1)]                                                        */
int vr_istft(vr_handle h, const float* spec, int spec_on_device, int T, float* wave, int wave_on_device);

/* Separator(model, device, batchsize, cropsize).separate / separate_tta     inference.py:70-102
 * spec [2,bins,T] complex64 -> y_spec (instruments), v_spec (vocals), same shape.
 * `tta` is a flag word: bit 0 = --tta (separate_tta), bit 1 = --postprocess (spec_utils.merge_artifacts,
 * lib/spec_utils.py:60-93, inference.py:27-30).  batchsize <= 0: all crops of a pass in one device batch. */
int vr_separate(vr_handle h, const float* spec, int spec_on_device, int T, int tta, int batchsize,
                int cropsize, float* y_spec, float* v_spec, int out_on_device);

/* The whole of inference.py:147-176 in one device-resident call:
 * wave_to_spectrogram -> Separator.separate[_tta] -> spectrogram_to_wave x2.
 * wave [2, L] -> y_wave, v_wave [2, hop*(L/hop)]                                                  */
int vr_separate_wave(vr_handle h, const float* wave, int wave_on_device, int64_t L, int tta, int batchsize,
                     int cropsize, float* y_wave, float* v_wave, int out_on_device);

/* ---- training: the body of train.train_epoch (train.py:77-96) ------------------------------------ */
/* mask = model(X); loss = L1Loss()(mask * X, y); (loss / accumulation_steps).backward()
 * X, y: [B, 2, bins, T] fp32.  Gradients ACCUMULATE in the library's gradient arena until vr_zero_grad
 * (model.zero_grad()).  *loss_out = the un-scaled mean L1 loss (loss.item()).  mask_out (optional,
 * may be NULL): the full-width mask [B,2,bins,T] that model(X) returns.  Needs vr_set_mode(h, 1).    */
int vr_train_step(vr_handle h, const float* X, const float* y, int on_device, int B, int T, int accumulation_steps,
                  float* loss_out, float* mask_out, int mask_on_device);
/* The same step as TWO calls, for callers that keep the reference's own loss expression and optimizer between them
 * (train.py:81-95 unmodified: `mask = model(X)` ... `loss.backward()` ... `optimizer.step()`):
 *   vr_forward_train   mask = model(X) under model.train(): batch-statistics BatchNorm (running buffers updated), live
 *                      Dropout2d, the graph (raw activations) kept inside the handle.  mask_out [B,2,bins,T].
 *   vr_backward        dmask = dLoss/dmask [B,2,bins,T] -> gradients ACCUMULATE in the gradient arena exactly like
 *                      vr_train_step's.  Consumes the graph; any other call on the handle in between frees it (-2).
 * vr_param_arena: the flat fp32 parameter arena (device pointer, element count; same indexing as vr_grad_arena), so
 * that an element-wise optimizer from outside (torch.optim.Adam on a zero-copy view) can update the weights in place;
 * call vr_set_option(h, "params_dirty", 1) after writing it so that eval-mode folded tables are rebuilt.            */
int vr_forward_train(vr_handle h, const float* X, int on_device, int B, int T, float* mask_out, int mask_on_device);
int vr_backward(vr_handle h, const float* dmask, int on_device);
/* Identity of the graph the handle currently holds: *generation counts the vr_forward_train calls so far, *valid (may be
 * NULL) says whether that graph is still alive.  A caller that keeps several forward results around (autograd) records the
 * generation after vr_forward_train and refuses to call vr_backward for any other one -- the handle keeps ONE graph. */
int vr_graph_generation(vr_handle h, int64_t* generation, int* valid);
int vr_param_arena(vr_handle h, float** device_ptr, int64_t* numel);

/* Training input pipeline on the device: replaces the numeric part of
 * lib/dataset.py VocalRemoverTrainingSet.__getitem__ (dataset.py:105-120) for a whole batch.
 *   X, y           [B][T][2][bins] complex64 (re,im interleaved): the cropsize rows read from the cached .npy files
 *                  (their on-disk row order; dataset.py:33-46,58-66), host or device
 *   X_mix, y_mix   the mixup partners' rows (dataset.py:85-103), same layout; may be NULL when no sample mixes
 *   desc[b]        coef (dataset.py:109), coef_mix (dataset.py:90-91), lam (np.random.beta, dataset.py:97) and flags:
 *                  bit 0 aggressively_remove_vocal, 1 channel swap, 2 inst-only, 3 mixup,
 *                  bits 4-6 = bits 0-2 for the partner (dataset.py:68-83 is applied to both independently)
 *   reduction_weight [bins]  (train.py:197-205), NULL if no flag needs it
 * Output: X_mag, y_mag [B][2][bins][T] fp32 = np.abs of the augmented crops, the tensors train_epoch consumes. */
typedef struct vr_aug { float coef; float coef_mix; float lam; int flags; } vr_aug;
int vr_augment_batch(vr_handle h, const float* X, const float* y, const float* X_mix, const float* y_mix, const vr_aug* desc,
                     const float* reduction_weight, int B, int T, int bins, int in_on_device, float* X_mag, float* y_mag,
                     int out_on_device);

/* torch.optim.Adam(lr, betas=(b1,b2), eps, weight_decay=0).step()   train.py:215-218,95
 * grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce).  Hyper-parameters are doubles
 * like torch's python floats: 1 - beta, the bias corrections and lr / bias_correction1 are formed in double and only
 * then rounded to the fp32 the parameters live in (1 - 0.999f differs from 0.001f by 1.3e-5 relative).          */
int vr_adam_step(vr_handle h, double lr, double b1, double b2, double eps, double grad_scale);
int vr_zero_grad(vr_handle h);                                          /* model.zero_grad(), train.py:96 */
/* optimizer.state_dict() / load_state_dict() for a resumable checkpoint (the reference saves the model only,
 * train.py:290): the Adam moments as flat host arrays of vr_grad_arena's element count (arena order) + the step. */
int vr_get_adam_state(vr_handle h, float* exp_avg, float* exp_avg_sq, int64_t numel, int64_t* step);
int vr_set_adam_state(vr_handle h, const float* exp_avg, const float* exp_avg_sq, int64_t numel, int64_t step);
int vr_get_grad(vr_handle h, const char* key, float* host, int64_t capacity_bytes);   /* param.grad, torch layout */
/* nn.Dropout2d(0.1) on the five ASPP outputs (lib/layers.py:90), live in train mode.  mode 1 (the DEFAULT, as in
 * the reference): device-side counter-based generator keyed on (seed, number of train-mode forwards so far), a
 * fresh draw per forward; 0: off (explicit opt-out, parity tests);
 * 2: injected keep-masks [5][B][8*nout] holding 0 or 1/0.9 (parity tests), nets in the order
 * stg1_low, stg1_high, stg2_low, stg2_high, stg3_full, row pitch 8*c of each net.                   */
int vr_set_dropout(vr_handle h, int mode, uint64_t seed, const float* masks, int B);
/* The single flat fp32 gradient bucket (device pointer + element count) for the data-parallel
 * all-reduce: vr_allreduce_grads sums it in place with the library's own RCCL communicator (or a caller may reduce this view
 * through torch.distributed -- Trainer(backend='torch' | 'staged')), then vr_adam_step.                                      */
int vr_grad_arena(vr_handle h, float** device_ptr, int64_t* numel);

/* One batch of train.validate_epoch (train.py:117-127), eval mode:
 *   y_pred = model.predict(X); y = spec_utils.crop_center(y, y_pred); loss = nn.L1Loss()(y_pred, y)
 * X, y: [B, 2, bins, T] fp32.  *loss_out = loss.item().  Forward, crop and the L1 reduction run on the device. */
int vr_validate_step(vr_handle h, const float* X, const float* y, int on_device, int B, int T, float* loss_out);

/* ---- data-parallel exchange.  The reference has none: train.py:211-213 takes one --gpu. ----
 * One process per GPU, one handle per process.  Rank 0 draws an id (vr_comm_unique_id, 128 bytes = ncclUniqueId),
 * hands it to the other ranks by any host channel (INTEGRATION.md uses torch.distributed's store), every rank calls
 * vr_comm_init.  Per optimizer step: vr_train_step -> vr_allreduce_grads -> vr_adam_step(grad_scale = 1/world).
 * Parity definition: N ranks == the reference's gradient accumulation with accumulation_steps = N (train.py:91-96).
 * RCCL is dlopen()ed on first use (the copy already mapped in the process, e.g. torch's, else the system's).      */
#define VR_COMM_ID_BYTES 128
int vr_comm_unique_id(void* id_out);
int vr_comm_init(vr_handle h, int rank, int world_size, const void* id);
int vr_comm_destroy(vr_handle h);
/* In-place SUM all-reduce of the flat gradient arena (vr_grad_arena) over RCCL/xGMI, enqueued on the handle's stream
 * (no host synchronisation; vr_adam_step follows on the same stream).  wire_dtype 0: fp32 bucket; 1: bf16 bucket
 * (rounded to bf16, summed in bf16 on the wire, widened back to the fp32 arena: half the bytes per link).         */
int vr_allreduce_grads(vr_handle h, int wire_dtype);
/* Rank `root`'s parameters, BatchNorm buffers and num_batches_tracked (and, with_optimizer != 0, the Adam moments and
 * step count) replace every rank's: replicas start identical (what DistributedDataParallel does at construction). */
int vr_broadcast_params(vr_handle h, int root, int with_optimizer);

/* ---- audio front end (no model handle, `device` = GPU index, host pointers) ----------------
 * vr_resample: the resampling step of librosa.load(path, sr=sr_out, res_type='kaiser_fast')   inference.py:136-138,
 * lib/spec_utils.py:139-142.  x [channels][n_in] at sr_in -> y [channels][n_out], n_out = ceil(n_in * sr_out / sr_in)
 * (librosa's fix_length: resampy yields int(n_in * ratio) samples, the rest is zero).  resampy~=0.4 is a third-party
 * dependency of the reference that is not vendored: its published 'kaiser_fast' filter is restated (parity unpinned).
 * vr_xcorr_argmax: np.argmax(np.correlate(a, b, 'full'))        lib/spec_utils.py:107-108 (align_wave_head_and_tail). */
int vr_resample(int device, const float* x, int channels, int64_t n_in, int sr_in, int sr_out, float* y, int64_t n_out);
int vr_xcorr_argmax(int device, const float* a, int64_t na, const float* b, int64_t nb, int64_t* argmax_out);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------- */
/* Bracket subsequent calls: every kernel launch is timed with HIP events on the stream it is launched on (the executor runs
 * every kernel on ONE stream while profiling); conv_* aggregate the convolution launches.
 * conv_flops = 2 x multiply-adds of the direct convolutions (the Winograd kernel performs fewer),
 * conv_bytes = algorithmic HBM bytes (virtual input + weights + output of each launch, once each). */
int vr_profile_begin(vr_handle h);
int vr_profile_end(vr_handle h, double* conv_ms, double* conv_flops, int* conv_launches, double* conv_bytes);
/* Per-kernel totals of the step bracketed by the last vr_profile_begin / vr_profile_end: EVERY kernel the library launched
 * (HIP events on the stream each launch went to), one text line per kernel name
 *     name \t calls \t milliseconds \t algorithmic FLOPs \t algorithmic HBM bytes \t calls that carried figures \n
 * (FLOPs / bytes: convolutions = 2 x multiply-adds of the direct form and input + output + weights once each -- forward,
 * data-gradient and weight-gradient launches alike; element-wise kernels = bytes read + written once; 0 where a kernel has no
 * noted figure).  Returns the size needed incl. the terminator; copies at most `capacity` bytes.  bench.py builds
 * `roofline.classes` from it. */
int64_t vr_profile_report(vr_handle h, char* buf, int64_t capacity);

/* ---- test hooks (tests/ only) ---------------------------------------------------------------- */
/* One convolution through the library's conv dispatcher: x [N,Cin,H,W], w OIHW, padding = dilation
 * (3x3) or 0 (1x1).  `upsample` is a flag word: bit 0 = fused bilinear x2 upsample of x; bit 1 = also
 * hand the launch Winograd-domain weights (3x3 stride-1 only; taken when the input is plain and
 * stats_out is null); bit 2 = `affine`/`slope` are the EPILOGUE ([Cout][2] folded BatchNorm +
 * activation, the eval-mode form) instead of a pending affine [Cin][2] on the input.
 * stats_out [Cout][2] receives (sum, sumsq) of the raw output per channel when non-null.          */
int vr_debug_conv2d(vr_handle h, const float* x, int N, int Cin, int H, int W, const float* w, int Cout,
                    int ksize, int stride, int dil_h, int dil_w, int upsample, const float* affine,
                    float slope, const float* bias, float* out, float* stats_out);
/* Backward of the same single convolution through the MFMA data-gradient / weight-gradient kernels:
 * dz [N,Cout,Hout,Wout] -> dx_out [N,Cin,H,W] (gradient w.r.t. the activated, pre-upsample input
 * values) and dw_out (OIHW).                                                                       */
int vr_debug_conv2d_backward(vr_handle h, const float* x, int N, int Cin, int H, int W, const float* w, int Cout,
                             int ksize, int stride, int dil_h, int dil_w, int upsample, const float* affine,
                             float slope, const float* dz, float* dx_out, float* dw_out);
/* Host half of --postprocess (no GPU involved): per-frame mask minimum [T] -> merge_artifacts blend
 * weight [T] (lib/spec_utils.py:64-87), incl. the reference's IndexError / ValueError cases.        */
int vr_debug_merge_artifacts_weight(const float* frame_min, int T, float thres, int min_range, int fade_size,
                                    float* weight_out);
/* One kernel of the training path in isolation, host pointers in and out (csrc/debug.hip lists the names, their
 * dims / float parameters / inputs / outputs): bn_backward, lstm, upsample, pool, thin, head_loss, rows, adam.      */
int vr_debug_kernel(vr_handle h, const char* name, const int64_t* dims, int ndims, const float* fparams, int nfparams,
                    const float* const* inputs, int ninputs, float* const* outputs, int noutputs);
/* Record intermediate activations of the next vr_forward and read them back (post-activation). */
int vr_debug_record_taps(vr_handle h, int enable);
int64_t vr_debug_get_tap(vr_handle h, const char* name, float* host, int64_t capacity_floats, int64_t* shape4);

#ifdef __cplusplus
}
#endif
#endif /* VR_MI355_H */
