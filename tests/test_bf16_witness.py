"""configs[4] ("bf16 data parallel"): is the collapse of the gradient direction under bf16 MFMA operands the arithmetic or a kernel bug?
(VERDICT r4 "next round" item 6.)  The same train step (train.py:77-96 through oracle/train_step.py) is evaluated on the CPU in fp32 and
in three bf16 forms that share no code with the library (oracle/bf16_witness.py):

    operands / s1    the two operands of every 3x3 stride-1 convolution rounded to bf16, forward, data gradient and weight gradient;
                     products and sums fp32, every tensor stored in fp32 -- the arithmetic of the library's `mfma_mode` 1
    operands / all   the same for every convolution
    autocast         torch.autocast('cpu', dtype=torch.bfloat16): torch's own mixed precision (bf16 activations as well)

CPU part (here): at batch 2 x [2,1025,256] on the full CascadedNet(2048,1024,32,128) ALL THREE lose the fp32 gradient direction
(measured global cosine 0.26 / 0.18 / 0.08, per-tensor medians 0.31 / 0.24 / 0.11; batch 8: see DESIGN.md section 6) while loss and
gradient norm stay right -- so the 0.37 the GPU's mode 1 shows at batch 16 is what bf16 operands do to a randomly initialised net
behind ~100 batch-statistics BatchNorms, not a defect of the bf16 kernels.  GPU part: `mfma_mode` 1 on the same inputs lands in the
same place as the `operands / s1` witness."""
import numpy as np
import pytest
import torch

from oracle import bf16_witness as bw
from oracle import train_step, weights

N_FFT, HOP = 2048, 1024


@pytest.fixture(scope='module')
def step2():
    torch.manual_seed(0)
    sd = weights.make_state_dict(1234)
    X, y = train_step.synth_batch(2, T=256, n_fft=N_FFT, seed=0)
    masks = train_step.dropout_masks(2, 7)
    loss, grads = bw.loss_and_grads(sd, X, y, N_FFT, masks, 'fp32')
    return sd, X, y, masks, loss, grads


@pytest.mark.parametrize('form,which', [('operands', 's1'), ('operands', 'all'), ('autocast', 'all')])
def test_bf16_operands_lose_the_fp32_gradient_direction_on_the_cpu_too(step2, form, which):
    sd, X, y, masks, loss0, g0 = step2
    loss, g = bw.loss_and_grads(sd, X, y, N_FFT, masks, form, which)
    s = bw.summary(g0, g)
    print('%s/%s vs fp32 at batch 2: loss %.8f vs %.8f; gradient global cosine %.4f, |g| ratio %.4f, per-tensor cosine min %.4f median %.4f'
          % (form, which, loss, loss0, s['global_cosine'], s['norm_ratio'], s['tensor_cosine_min'], s['tensor_cosine_median']))
    # what stays right: the loss (to 1e-3) and the size of the gradient (to 10 %) ...
    assert abs(loss - loss0) <= 1e-3 * abs(loss0)
    assert 0.9 <= s['norm_ratio'] <= 1.1
    # ... and what does not: the direction.  An implementation-independent property of 8-bit operands on this net at random initialisation.
    assert s['global_cosine'] < 0.6 and s['tensor_cosine_median'] < 0.6


@pytest.mark.gpu
def test_gpu_bf16_mode_matches_the_cpu_witness(vr, step2):
    """`mfma_mode` 1 (bf16 MFMA operands in the Winograd forward / data-gradient / weight-gradient kernels and the 1x1 weight-gradient
    GEMM) against the library's own fp32 step on the same batch-2 inputs: it loses the direction like the CPU witness does -- global
    cosine within 0.25 of the witness's, same loss, same gradient norm -- and the fp32 GPU step agrees with the fp32 CPU step."""
    sd, X, y, masks, loss0, g0 = step2
    w = bw.summary(g0, bw.loss_and_grads(sd, X, y, N_FFT, masks, 'operands', 's1')[1])
    model = vr.nets.CascadedNet(N_FFT, HOP, 32, 128)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0'))
    Xd, yd = X.to('cuda:0'), y.to('cuda:0')

    def step(**opt):
        try:
            model.load_state_dict(sd)
            for k, v in opt.items():
                model.set_option(k, v)
            model.train()
            model.set_dropout_masks(masks)
            model.zero_grad()
            loss = model.train_step(Xd, yd, 1)
            return loss, {k: v.cpu() for k, v in model.grads().items()}
        finally:
            model.set_dropout_masks(None)
            model.set_option('mfma_mode', -1)
            model.eval()

    loss_f, g_f = step(mfma_mode=0)
    loss_b, g_b = step(mfma_mode=1)
    fp32 = bw.summary(g0, g_f)
    gpu = bw.summary(g_f, g_b)
    print('GPU fp32 (mode 0) vs CPU fp32: global cosine %.5f; GPU bf16 operands (mode 1) vs GPU fp32: global cosine %.4f, |g| ratio %.4f, '
          'per-tensor median %.4f -- CPU witness (operands/s1 vs CPU fp32): %.4f, %.4f, %.4f'
          % (fp32['global_cosine'], gpu['global_cosine'], gpu['norm_ratio'], gpu['tensor_cosine_median'], w['global_cosine'], w['norm_ratio'],
             w['tensor_cosine_median']))
    assert abs(loss_f - loss0) <= 2e-6 * abs(loss0) + 2e-6 and fp32['global_cosine'] >= 0.995
    assert abs(loss_b - loss_f) <= 1e-3 * abs(loss_f) and 0.9 <= gpu['norm_ratio'] <= 1.1
    assert abs(gpu['global_cosine'] - w['global_cosine']) <= 0.25 and gpu['global_cosine'] < 0.7
