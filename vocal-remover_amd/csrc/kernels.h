// Launch wrappers for the non-MFMA kernels of libvr_mi355.so (definitions in *.hip).
#pragma once
#include "vr_common.h"

namespace vr {

// ---- pointwise.hip -----------------------------------------------------------------------------
// Thin-output 1x1 convs that are HBM-bound (Cout = 1 or 2): the LSTM squeeze conv
// (lib/layers.py:112) and the mask head `out` + sigmoid + replicate-pad + offset crop
// (lib/nets.py:79,109-115,127-128).
struct HeadDst {
    float* p;                 // destination base
    long long dN, dC, dH;     // strides of the destination (elements)
    int w_lo, w_hi;           // keep input columns [w_lo, w_hi); column w lands at w - w_lo
    int pad_rows;             // replicate the last input row this many extra times (1025 - 1024)
};
void launch_head_sigmoid(const Tensor& x, const float* w /*[2][C]*/, const HeadDst& d, hipStream_t st);
// out[n][0][h][w] = sum_c w[c] * act(x);  part: [nblocks][2] (sum, sumsq) or null; returns nblocks
// epi (eval): device [2] = folded (scale, shift) of the single-channel BatchNorm, applied with the ReLU before the store
int launch_squeeze_conv(const Tensor& x, const float* w /*[C]*/, float* out, float* part, bool dry, hipStream_t st,
                        const float* epi = nullptr);

// mean over H of act(x) -> out [N][C][W]  (AdaptiveAvgPool2d((1, None)), lib/layers.py:72)
void launch_avgpool_h(const Tensor& x, float* out, hipStream_t st);

// BatchNorm bookkeeping ---------------------------------------------------------------------------
struct BNFoldDesc { const float *w, *b, *rm, *rv; float* affine; int C; int bcast; };  // bcast>0: C==1, replicate to bcast rows
void launch_bn_fold_eval(const BNFoldDesc* d_descs, int ndesc, int maxC, float eps, hipStream_t st);
struct BNFinalizeArgs {
    const float* part; int nparts; int pstride;   // partial rows: part[i*pstride + c*2 + {0,1}]
    double count;                                  // elements per channel
    const float *w, *b; float *rm, *rv;           // running stats updated in place (momentum)
    float* affine; float* save_mean; float* save_invstd;
    int C; float eps, momentum;
    int broadcast;                                 // >0: C==1 stats, affine replicated to `broadcast` rows
};
void launch_bn_finalize(const BNFinalizeArgs& a, hipStream_t st);

// in-place rows affine + relu on [N][R][W]: v = relu(v*aff[r][0] + aff[r][1])  (BatchNorm1d + ReLU
// of lib/layers.py:120-121 applied to the Linear output laid out [N, nbins, nframes])
void launch_rows_affine_relu(const float* x, float* out, const float* aff, int N, int R, int W, hipStream_t st);

// out[i] = a[i] + b[i]
void launch_add(const float* a, const float* b, float* out, int n, hipStream_t st);

// dense post-activation copy of a Tensor (debug taps / tests)
void launch_materialize(const Tensor& x, float* out, hipStream_t st);
// training input pipeline (augment.hip); layout-compatible with vr_aug in include/vr_mi355.h
struct AugDesc { float coef, coef_mix, lam; int flags; };
void launch_augment(const float2* X, const float2* Y, const float2* Xi, const float2* Yi, const AugDesc* desc, const float* rw,
                    int B, int T, int bins, float* Xmag, float* Ymag, hipStream_t st);
bool thin16_pick(const ConvArgs& a, const ConvShape& s, int* TH);  // conv_thin.hip: <= 16 couts on v_mfma_f32_16x16x4_f32
void thin16_fill_tiling(ConvArgs& a, int TH);
void thin16_launch_conv(const ConvArgs& a, const ConvShape& s, int TH, hipStream_t st);
bool s2d_fused_eligible(const ConvArgs& a);                       // conv_dma.hip: stride-2 data gradient, four parity classes in one launch
void launch_s2d_fused(const ConvArgs& a, hipStream_t st);
struct S2WDesc { const float* w; float* wc; int Cin, Cout, CoutPad, CinPad; };          // one stride-2 layer; max_elems = max over layers of 4 * Cout * 9 * CinPad
void launch_s2_class_weights(const S2WDesc* d_descs, int n, long long max_elems, hipStream_t st);
void launch_wino_weights(const float* w, float* u, int Cin, int CoutPad, hipStream_t st);   // U = G g G^T
struct WinoWDesc { const float* w; void* u; int Cin, CoutPad; };                              // one layer of a batched refresh
void launch_wino_weights_batched(const WinoWDesc* d_descs, int n, long long max_elems, bool split6, hipStream_t st);
size_t wino_weights6_bytes(int Cin, int CoutPad);                                           // U as three bf16 planes (mfma_mode 2)
void launch_wino_weights6(const float* w, void* u6, int Cin, int CoutPad, hipStream_t st);
// conv_x3.hip: direct 3x3 stride-1 conv, fp32 products from six bf16 products (mfma_mode 2)
struct X3Tile { int MT, TH; };
struct X3WDesc { const float* w; void* o; int Cin, KK, CoutPad; };                            // one layer of a batched weight split
bool x3_pick(const ConvArgs& a, const ConvShape& s, X3Tile* t);
void x3_fill_tiling(ConvArgs& a, const X3Tile& t);
void x3_launch_conv(const ConvArgs& a, const X3Tile& t, hipStream_t st);
size_t x3_weights_bytes(int Cin, int KK, int CoutPad);
void launch_x3_weights(const float* w, void* o, int Cin, int KK, int CoutPad, hipStream_t st);
void launch_x3_weights_batched(const X3WDesc* d_descs, int n, long long max_elems, hipStream_t st);
// conv_x3h.hip: the same convs with fp32-grade products from three fp16 products (mfma_mode 3); weights in the SAME buffers
// (x3_weights_bytes), format [chunk][tap][2][CoutPad][8] fp16 + per-cout scale tails
void x3h_launch_conv(const ConvArgs& a, const X3Tile& t, hipStream_t st);
void x3h_trace_read(long long* host, int n);          // diagnostics: phase stamps of the TRACE build (VR_CONV_DBG bit 64)
void x3h_trace_clear();
void launch_x3h_weights(const float* w, void* o, int Cin, int KK, int CoutPad, hipStream_t st);
void launch_x3h_weights_batched(const X3WDesc* d_descs, int n, long long max_elems, int max_cout_pad, hipStream_t st);
// conv_x3d.hip (round 6): conv_x3h's arithmetic for the 16-column layers -- dilated 3x3 (ASPP), 3x3 dilation 1 (enc5.conv2), 1x1 (ASPP conv2);
// weights in x3h format (launch_x3h_weights with KK = 9 / 1); mfma_mode 3 only
bool x3d_pick(const ConvArgs& a, const ConvShape& s, int* MT);
void x3d_fill_tiling(ConvArgs& a, int MT);
void x3d_launch_conv(const ConvArgs& a, const ConvShape& s, int MT, hipStream_t st);
bool x3d_aspp_eligible(const ConvArgs* c4, const ConvShape* s4);   // c4 / s4 in concat order: 1x1, dilation (4,2), (8,4), (12,6)
void x3d_launch_aspp(const ConvArgs* c4, hipStream_t st);          // the four branch convs of an ASPP module in ONE launch
void launch_upsample2x(const Tensor& x, float* out, hipStream_t st);   // dense [N][C][2H][2W], activated

// ---- lstm.hip -----------------------------------------------------------------------------------
// gx: [N][2*4H][T] input projections (+bias) for both directions; whh: [2][4H][H];
// out: [N][2H][T] (forward hidden in channels [0,H), reverse in [H,2H)).
void launch_bilstm(const float* gx, const float* whh_f, const float* whh_r, float* out,
                   int N, int T, int H, hipStream_t st);

// training variants: `save` [N][2][T][5H] keeps (i, f, g, o, c) per step for the backward pass
void launch_bilstm_train(const float* gx, const float* whh_f, const float* whh_r, float* out, float* save,
                         int N, int T, int H, hipStream_t st);
// dh [N][2H][T] (gradient at the LSTM output) -> dgx [N][8H][T] (gradient at the input projections)
void launch_bilstm_bwd(const float* dh, const float* save, const float* whh_f, const float* whh_r, float* dgx,
                       int N, int T, int H, hipStream_t st);
// dW_hh[dir][g][k] = sum_{n,t} dgx[n][dir*4H+g][t] * h_prev[n][dir*H+k][t]
size_t lstm_whh_grad_scratch_floats(int N, int H);      // floats of `part` below (one slab per sample slice and direction)
void launch_lstm_whh_grad(const float* dgx, const float* hout, float* dwhh_f, float* dwhh_r, int N, int T, int H,
                          int accumulate, float* part, hipStream_t st);

// ---- backward.hip -------------------------------------------------------------------------------
struct BnBwdArgs {
    float* g;                       // in: G (grad wrt post-activation value); out: dz (in place)
    const float* z;                 // raw conv output, same strides as g
    int N, C, H, W;
    long long sN, sC, sH;
    const float* aff;               // [C][2] scale, shift used in the forward (null = identity)
    int aff_bcast;                  // 1: C == 1 and the table is a broadcast copy (use row 0)
    float slope;
    const float* post;              // [N][C] dropout keep-mask or null
    const float *gamma, *save_mean, *save_invstd;
    float *dgamma, *dbeta;          // gradient arena slots
    int acc_grads;
    float* coef;                    // [C][3] scratch (kA, kB, kC); null = no BatchNorm (dz = dy)
    float* part;                    // [chunks][C][2] scratch
};
int bn_bwd_chunks(const BnBwdArgs& a);
void launch_bn_bwd(const BnBwdArgs& a, hipStream_t st);

void launch_upsample_bwd(const float* dhi, int N, int C, int H, int W, float* glo, long long gN, long long gC,
                         long long gH, int accumulate, hipStream_t st);
void launch_sum_h(const float* d, int N, int C, int H, int W, float* out, hipStream_t st);
void launch_avgpool_bwd(const float* gp, float* g, int N, int C, int H, int W, long long sN, long long sC, long long sH,
                        int accumulate, hipStream_t st);
void launch_thin_dgrad(const Tensor& x, int CO, const float* w, const float* dz, float* g, int accumulate, hipStream_t st);
int thin_wgrad_blocks(const Tensor& x);
void launch_thin_wgrad(const Tensor& x, int CO, const float* dz, float* part, float* dw, int accumulate, hipStream_t st);
void launch_reduce_rows(const float* part, long long stride, int P, float* out, long long n, int accumulate, float scale,
                        hipStream_t st);
int head_loss_blocks(const Tensor& x);
void launch_head_loss(const Tensor& x, const float* w, const float* X, const float* Y, int bins, float gscale,
                      float* dlogit, float* mask_out, float* loss_part, float* loss_out, float loss_scale, hipStream_t st);
void launch_head_bwd(const float* dmask, const float* mask, int N, int H, int W, int bins, float* dlogit, hipStream_t st);
struct FlipDesc { const float* w; float* wt; int Cin, Cout, KK, CinPad, CoutPad; };
void launch_flip_transpose(const FlipDesc* d_descs, int n, hipStream_t st);
void launch_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2, double eps,
                 long long step, double gscale, hipStream_t st);
void launch_channel_sum(const float* d, int N, int C, int W, float* out, int accumulate, hipStream_t st);
void launch_f32_to_bf16(const float* x, unsigned short* y, long long n, hipStream_t st);
void launch_bf16_to_f32(const unsigned short* x, float* y, long long n, hipStream_t st);

// ---- stft.hip -----------------------------------------------------------------------------------
struct FFTPlan { int n_fft; int log2n; float2* twiddle; float* window; };
// wave [2][L] -> spec [2][bins][T] complex64
void launch_stft(const FFTPlan& pl, const float* wave, long long L, int hop, int T, float2* spec, hipStream_t st);
// spec [2][bins][T] -> frames scratch [2][T][n_fft] -> wave [2][hop*(T-1)]
void launch_istft(const FFTPlan& pl, const float2* spec, int hop, int T, float* frames, float* wave, hipStream_t st);
// Fused form for hop == n_fft/2: wave = istft(m * spec) (which 0) or istft(spec - m * spec) (which 1); mask_a null = plain
// istft.  No frame buffer, no materialised y / v spectrograms (inference.py:26-40 + lib/spec_utils.py:157-165 in one pass).
bool istft_masked_available(const FFTPlan& pl, int hop);
void launch_istft_masked(const FFTPlan& pl, const float2* spec, int hop, int T, const float* mask_a, int Wa, const float* mask_b,
                         int Wb, int shift, const float* wgt, int which, float* wave, hipStream_t st);
// mag_pad [2][bins][Wpad] (pre-zeroed) <- |spec| at column pad_l + t; maxima into stats:
// per-row partial maxima into stats (16 B header + 2 x bins rows of (max |X| bits, 64-bit lexicographic complex key))
void launch_mag_pad(const float2* spec, int bins, int T, float* mag_pad, int Wpad, int pad_l,
                    unsigned* stats, hipStream_t st);
// aff[0..3] = (1/coef, 0, 1/coef, 0), coef = max|X| (mode 0) or |lexicographic max| (mode 1)
void launch_coef_affine(const unsigned* stats, int rows, int mode, float* aff, hipStream_t st);   // rows = 2 * bins partials
// y = m*X, v = (1-m)*X with m = mask_a[.., t] (tta=0) or 0.5*(mask_a[.., t] + mask_b[.., t + shift])
// wgt [T] (or null): per-frame merge_artifacts weight, m += wgt[t] * (1 - m)
void launch_apply_mask(const float2* spec, int bins, int T, const float* mask_a, int Wa,
                       const float* mask_b, int Wb, int shift, const float* wgt, float2* y, float2* v, hipStream_t st);
// fmin[t] = min over (channel, bin) of the final mask at frame t
void launch_frame_min(int bins, int T, const float* mask_a, int Wa, const float* mask_b, int Wb, int shift, float* fmin,
                      hipStream_t st);

}  // namespace vr
