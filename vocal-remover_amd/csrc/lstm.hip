// Bidirectional LSTM recurrence of layers.LSTMModule (lib/layers.py:113-117,128).
//
// The input projection W_ih x_t + b_ih + b_hh for all time steps and both directions is hoisted
// into one 1x1-conv launch of the MFMA kernel (gx below), so what remains here is the strictly
// sequential part: 128 steps of  gates = gx_t + W_hh h_{t-1}  followed by the cell update.
// One workgroup per (sample, direction); W_hh^T lives in LDS for the whole sequence (4H*H floats,
// 64 KB at H=64), thread g owns gate row g, h is broadcast from LDS.  Latency-bound by design:
// the work per step is 4H*H FMAs; all (sample, direction) pairs run concurrently.
#include "kernels.h"

namespace vr {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void bilstm_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                              const float* __restrict__ whh_r, float* __restrict__ out, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = 4 * H;
    float* WT = lds;               // [H][G]
    float* hbuf = WT + H * G;      // [H]
    float* abuf = hbuf + H;        // [G]
    const int n = blockIdx.x, dir = blockIdx.y;
    const int g = threadIdx.x;
    const float* whh = dir ? whh_r : whh_f;
    for (int i = threadIdx.x; i < G * H; i += blockDim.x) {
        const int row = i / H, k = i % H;          // whh[row][k]
        WT[k * G + row] = whh[i];
    }
    if (g < H) hbuf[g] = 0.f;
    float c = 0.f;
    const float* gxp = gx + ((long long)n * 2 * G + (long long)dir * G + g) * T;
    float* outp = out + ((long long)n * 2 * H + (long long)dir * H + g) * T;
    __syncthreads();
    float pre = (g < G) ? gxp[dir ? T - 1 : 0] : 0.f;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        float nxt = 0.f;
        if (g < G && step + 1 < T) nxt = gxp[dir ? t - 1 : t + 1];   // prefetch next step's projection
        if (g < G) {
            float a = pre;
            for (int k = 0; k < H; ++k) a = fmaf(WT[k * G + g], hbuf[k], a);
            abuf[g] = a;
        }
        __syncthreads();
        if (g < H) {
            const float ig = sigmoidf_(abuf[g]);
            const float fg = sigmoidf_(abuf[H + g]);
            const float gg = tanhf(abuf[2 * H + g]);
            const float og = sigmoidf_(abuf[3 * H + g]);
            c = fg * c + ig * gg;
            const float h = og * tanhf(c);
            hbuf[g] = h;
            outp[t] = h;
        }
        __syncthreads();
        pre = nxt;
    }
}

void launch_bilstm(const float* gx, const float* whh_f, const float* whh_r, float* out,
                   int N, int T, int H, hipStream_t st) {
    const int G = 4 * H;
    VR_CHECK(G <= 1024, -2, "LSTM hidden size per direction must be <= 256");
    const int threads = ((G + 63) / 64) * 64;
    const size_t lds = (size_t)(H * G + H + G) * sizeof(float);
    VR_CHECK(lds <= 160 * 1024, -2, "LSTM W_hh does not fit LDS");
    static bool attr_set = false;
    if (!attr_set) {
        VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(bilstm_kernel, dim3(N, 2), dim3(threads), lds, st, gx, whh_f, whh_r, out, T, H);
    VR_HIP(hipGetLastError());
}

}  // namespace vr
