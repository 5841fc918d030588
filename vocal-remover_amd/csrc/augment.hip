// Training input pipeline on the device: everything VocalRemoverTrainingSet.__getitem__
// (lib/dataset.py:105-120) does AFTER the random decisions and the file reads --
//   X /= coef, y /= coef                                   (dataset.py:108-110)
//   aggressively_remove_vocal                               (dataset.py:48-56)
//   channel swap / inst-only                                (dataset.py:68-83)
//   mixup with a second, independently augmented crop       (dataset.py:85-103)
//   np.abs, and the [T,2,bins] -> [2,bins,T] transpose      (dataset.py:63-64,116-117)
// -- as ONE HBM-bound kernel over a whole batch.  The host keeps what is inherently host work: drawing
// the numpy random numbers in the reference's order and seek-reading cropsize rows of the cached .npy.
#include "kernels.h"

namespace vr {

__device__ __forceinline__ float2 cdivf(float2 a, float c) { return make_float2(a.x / c, a.y / c); }

// One (frame t, channel c, bin) element of one crop after /coef and the per-sample augmentations.
__device__ __forceinline__ void aug_element(const float2* __restrict__ X, const float2* __restrict__ Y, long long base, int bins,
                                            int t, int c, int bin, float coef, int reduce, int swap, int inst, float rw,
                                            float2& xo, float2& yo) {
    const int cs = swap ? 1 - c : c;
    const long long i = base + ((long long)t * 2 + cs) * bins + bin;
    float2 x = cdivf(X[i], coef), y = cdivf(Y[i], coef);
    if (reduce) {
        const float xm = hypotf(x.x, x.y), ym = hypotf(y.x, y.y);
        float v = xm - ym;
        v = v > ym ? v : 0.f;
        const float ym2 = fmaxf(ym - v * rw, 0.f);
        const float ux = ym > 0.f ? y.x / ym : 1.f, uy = ym > 0.f ? y.y / ym : 0.f;      // exp(1j * angle(y))
        y = make_float2(ym2 * ux, ym2 * uy);
    }
    if (inst) x = y;
    xo = x;
    yo = y;
}

// grid: (ceil(T/32), ceil(bins/32), B*2); block (32, 8).  32x32 tile transposed through LDS so that both the
// reads (bin-contiguous) and the writes (frame-contiguous) are coalesced.
__global__ __launch_bounds__(256) void augment_kernel(const float2* __restrict__ X, const float2* __restrict__ Y,
                                                      const float2* __restrict__ Xi, const float2* __restrict__ Yi,
                                                      const AugDesc* __restrict__ desc, const float* __restrict__ rw, int T,
                                                      int bins, float* __restrict__ Xmag, float* __restrict__ Ymag) {
    __shared__ float tx[32][33], ty[32][33];
    const int b = blockIdx.z >> 1, c = blockIdx.z & 1;
    const int t0 = blockIdx.x * 32, bin0 = blockIdx.y * 32;
    const AugDesc d = desc[b];
    const long long base = (long long)b * T * 2 * bins;
    const int bin = bin0 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int t = t0 + r;
        float xm = 0.f, ym = 0.f;
        if (t < T && bin < bins) {
            const float w = rw ? rw[bin] : 0.f;
            float2 x, y;
            aug_element(X, Y, base, bins, t, c, bin, d.coef, d.flags & 1, d.flags & 2, d.flags & 4, w, x, y);
            if (d.flags & 8) {
                float2 xi, yi;
                aug_element(Xi, Yi, base, bins, t, c, bin, d.coef_mix, d.flags & 16, d.flags & 32, d.flags & 64, w, xi, yi);
                const float l = d.lam, m = 1.f - d.lam;
                x = make_float2(l * x.x + m * xi.x, l * x.y + m * xi.y);
                y = make_float2(l * y.x + m * yi.x, l * y.y + m * yi.y);
            }
            xm = hypotf(x.x, x.y);
            ym = hypotf(y.x, y.y);
        }
        tx[r][threadIdx.x] = xm;
        ty[r][threadIdx.x] = ym;
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int bo = bin0 + r;
        if (t < T && bo < bins) {
            const long long o = (((long long)b * 2 + c) * bins + bo) * T + t;
            Xmag[o] = tx[threadIdx.x][r];
            Ymag[o] = ty[threadIdx.x][r];
        }
    }
}

void launch_augment(const float2* X, const float2* Y, const float2* Xi, const float2* Yi, const AugDesc* desc, const float* rw,
                    int B, int T, int bins, float* Xmag, float* Ymag, hipStream_t st) {
    const dim3 grid((T + 31) / 32, (bins + 31) / 32, B * 2), block(32, 8);
    VR_LAUNCH(augment_kernel, grid, block, 0, st, X, Y, Xi, Yi, desc, rw, T, bins, Xmag, Ymag);
    VR_HIP(hipGetLastError());
}

}  // namespace vr
