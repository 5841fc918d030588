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
int launch_squeeze_conv(const Tensor& x, const float* w /*[C]*/, float* out, float* part, bool dry, hipStream_t st);

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
void launch_rows_affine_relu(float* x, const float* aff, int N, int R, int W, hipStream_t st);

// out[i] = a[i] + b[i]
void launch_add(const float* a, const float* b, float* out, int n, hipStream_t st);

// dense post-activation copy of a Tensor (debug taps / tests)
void launch_materialize(const Tensor& x, float* out, hipStream_t st);

// ---- lstm.hip -----------------------------------------------------------------------------------
// gx: [N][2*4H][T] input projections (+bias) for both directions; whh: [2][4H][H];
// out: [N][2H][T] (forward hidden in channels [0,H), reverse in [H,2H)).
void launch_bilstm(const float* gx, const float* whh_f, const float* whh_r, float* out,
                   int N, int T, int H, hipStream_t st);

// ---- stft.hip -----------------------------------------------------------------------------------
struct FFTPlan { int n_fft; int log2n; float2* twiddle; float* window; };
// wave [2][L] -> spec [2][bins][T] complex64
void launch_stft(const FFTPlan& pl, const float* wave, long long L, int hop, int T, float2* spec, hipStream_t st);
// spec [2][bins][T] -> frames scratch [2][T][n_fft] -> wave [2][hop*(T-1)]
void launch_istft(const FFTPlan& pl, const float2* spec, int hop, int T, float* frames, float* wave, hipStream_t st);
// mag_pad [2][bins][Wpad] (pre-zeroed) <- |spec| at column pad_l + t; maxima into stats:
// stats[0] = max |X| as float bits (uint), stats[2..3] = 64-bit lexicographic complex max key
void launch_mag_pad(const float2* spec, int bins, int T, float* mag_pad, int Wpad, int pad_l,
                    unsigned* stats, hipStream_t st);
void launch_stats_init(unsigned* stats, hipStream_t st);
// aff[0..3] = (1/coef, 0, 1/coef, 0), coef = max|X| (mode 0) or |lexicographic max| (mode 1)
void launch_coef_affine(const unsigned* stats, int mode, float* aff, hipStream_t st);
// y = m*X, v = (1-m)*X with m = mask_a[.., t] (tta=0) or 0.5*(mask_a[.., t] + mask_b[.., t + shift])
void launch_apply_mask(const float2* spec, int bins, int T, const float* mask_a, int Wa,
                       const float* mask_b, int Wb, int shift, float2* y, float2* v, hipStream_t st);

}  // namespace vr
