// Audio front end of inference.py / dataset preparation on the device:
//
//   vr_resample       the resampling step of librosa.load(path, sr=44100, res_type='kaiser_fast')
//                     (inference.py:136-138, lib/spec_utils.py:139-142).  librosa delegates to resampy 0.4
//                     (requirements.txt: resampy~=0.4.0, NOT vendored in the reference => parity unpinned): band-limited
//                     sinc interpolation with a Kaiser-windowed filter table -- 'kaiser_fast' = 16 zero crossings,
//                     2^9 table samples per crossing, roll-off 0.85, Kaiser beta 8.555504641634386 -- linear interpolation
//                     between table entries, left wing + right wing per output sample (resampy/interpn.py).
//   vr_xcorr_argmax   argmax of np.correlate(a, b, 'full') in spec_utils.align_wave_head_and_tail
//                     (lib/spec_utils.py:107-108): one workgroup per lag.
// Both are tiny next to the network; they exist so that the whole of inference.py / cache_or_load stays on the device
// path and needs neither librosa nor resampy.
#include <cmath>
#include <vector>

#include "kernels.h"

namespace vr {

// y[c][t] = sum over both filter wings (resampy.interpn._resample_loop), fp32 accumulator like the float32 output array
__global__ void resample_kernel(const float* __restrict__ x, long long n_in, float* __restrict__ y, long long n_out,
                                const double* __restrict__ win, const double* __restrict__ delta, int nwin, int precision,
                                double sample_ratio) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= n_out) return;
    const float* xc = x + (long long)c * n_in;
    const double scale = sample_ratio < 1.0 ? sample_ratio : 1.0;
    const double time_increment = 1.0 / sample_ratio;
    const int index_step = (int)(scale * precision);
    const double time_register = (double)t * time_increment;
    const long long n = (long long)time_register;
    double frac = scale * (time_register - (double)n);
    double index_frac = frac * precision;
    int offset = (int)index_frac;
    double eta = index_frac - offset;
    float acc = 0.f;
    long long i_max = (nwin - offset) / index_step;
    if (i_max > n + 1) i_max = n + 1;
    for (long long i = 0; i < i_max; ++i) {
        const int k = offset + (int)i * index_step;
        const double w = win[k] + eta * delta[k];
        acc = (float)((double)acc + w * (double)xc[n - i]);
    }
    frac = scale - frac;
    index_frac = frac * precision;
    offset = (int)index_frac;
    eta = index_frac - offset;
    long long k_max = (nwin - offset) / index_step;
    if (k_max > n_in - n - 1) k_max = n_in - n - 1;
    for (long long k2 = 0; k2 < k_max; ++k2) {
        const int k = offset + (int)k2 * index_step;
        const double w = win[k] + eta * delta[k];
        acc = (float)((double)acc + w * (double)xc[n + k2 + 1]);
    }
    y[(long long)c * n_out + t] = acc;
}

// resampy.filters.sinc_window(num_zeros, precision, kaiser(beta), rolloff): the right half of the windowed sinc
static void kaiser_sinc_table(int num_zeros, int precision_bits, double rolloff, double beta, std::vector<double>& win) {
    const int num_bits = 1 << precision_bits;
    const int n = num_bits * num_zeros;
    win.resize((size_t)n + 1);
    const double PI = 3.14159265358979323846;
    const double i0b = std::cyl_bessel_i(0.0, beta);
    for (int k = 0; k <= n; ++k) {
        const double xs = rolloff * ((double)num_zeros * k / n);                    // linspace(0, num_zeros, n+1) * rolloff
        const double sinc = xs == 0.0 ? 1.0 : std::sin(PI * xs) / (PI * xs);
        const double r = (double)k / n;                                             // kaiser(2n+1, beta)[n + k]
        const double taper = std::cyl_bessel_i(0.0, beta * std::sqrt(1.0 - r * r > 0.0 ? 1.0 - r * r : 0.0)) / i0b;
        win[k] = taper * rolloff * sinc;
    }
}

void resample_api(int device, const float* x, int channels, long long n_in, int sr_in, int sr_out, float* y, long long n_out) {
    VR_CHECK(channels > 0 && n_in > 0 && sr_in > 0 && sr_out > 0 && n_out > 0, -2, "resample: bad argument");
    DeviceGuard dev_guard(device);
    const double ratio = (double)sr_out / (double)sr_in;
    VR_CHECK(n_out <= (long long)std::ceil((double)n_in * ratio) + 1, -2, "resample: n_out larger than ceil(n_in * ratio)");
    std::vector<double> win, delta;
    kaiser_sinc_table(16, 9, 0.85, 8.555504641634386, win);                           // 'kaiser_fast'
    if (ratio < 1.0) for (double& v : win) v *= ratio;
    delta.resize(win.size());
    for (size_t i = 0; i + 1 < win.size(); ++i) delta[i] = win[i + 1] - win[i];
    delta.back() = 0.0;
    float *dx = nullptr, *dy = nullptr;
    double *dw = nullptr, *dd = nullptr;
    struct Free { void* p[4]; ~Free() { for (void* q : p) hipFree(q); } } fr{{nullptr, nullptr, nullptr, nullptr}};
    VR_HIP(hipMalloc(&dx, (size_t)channels * n_in * sizeof(float))); fr.p[0] = dx;
    VR_HIP(hipMalloc(&dy, (size_t)channels * n_out * sizeof(float))); fr.p[1] = dy;
    VR_HIP(hipMalloc(&dw, win.size() * sizeof(double))); fr.p[2] = dw;
    VR_HIP(hipMalloc(&dd, win.size() * sizeof(double))); fr.p[3] = dd;
    VR_HIP(hipMemcpy(dx, x, (size_t)channels * n_in * sizeof(float), hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dw, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(dd, delta.data(), win.size() * sizeof(double), hipMemcpyHostToDevice));
    // resampy produces int(n_in * ratio) samples; librosa's fix_length pads / trims to ceil(n_in * ratio) with zeros
    const long long n_core = (long long)((double)n_in * ratio);
    VR_HIP(hipMemset(dy, 0, (size_t)channels * n_out * sizeof(float)));
    const long long n_run = n_core < n_out ? n_core : n_out;
    if (n_run > 0) {
        float* ytmp = dy;
        // rows of dy are n_out long; the kernel writes the first n_run samples of each
        hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_run + 255) / 256), channels), dim3(256), 0, 0, dx, n_in, ytmp, n_out,
                           dw, dd, (int)win.size(), 1 << 9, ratio);
        VR_HIP(hipGetLastError());
    }
    VR_HIP(hipDeviceSynchronize());
    // samples [n_run, n_out) of every row must stay zero: the kernel guards t < n_out only, so clear the tail again
    if (n_run < n_out)
        for (int c = 0; c < channels; ++c)
            VR_HIP(hipMemset(dy + (size_t)c * n_out + n_run, 0, (size_t)(n_out - n_run) * sizeof(float)));
    VR_HIP(hipMemcpy(y, dy, (size_t)channels * n_out * sizeof(float), hipMemcpyDeviceToHost));
}

// full[k] = sum_n a[n + k - (nb - 1)] * b[n],  k = 0 .. na + nb - 2   (np.correlate(a, b, 'full'), real input)
__global__ __launch_bounds__(256) void xcorr_full_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b,
                                                         long long nb, float* __restrict__ full) {
    const long long k = blockIdx.x;
    const long long shift = k - (nb - 1);
    long long lo = shift < 0 ? -shift : 0;                 // n with 0 <= n + shift < na
    long long hi = nb < na - shift ? nb : na - shift;
    float s = 0.f;
    for (long long n = lo + threadIdx.x; n < hi; n += 256) s = fmaf(a[n + shift], b[n], s);
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) full[k] = red[0] + red[1] + red[2] + red[3];
}

void xcorr_argmax_api(int device, const float* a, long long na, const float* b, long long nb, long long* argmax_out) {
    VR_CHECK(na > 0 && nb > 0 && na + nb - 1 < 0x7FFFFFFFLL, -2, "xcorr: bad lengths");
    DeviceGuard dev_guard(device);
    const long long nf = na + nb - 1;
    float *da = nullptr, *db = nullptr, *df = nullptr;
    struct Free { void* p[3]; ~Free() { for (void* q : p) hipFree(q); } } fr{{nullptr, nullptr, nullptr}};
    VR_HIP(hipMalloc(&da, (size_t)na * sizeof(float))); fr.p[0] = da;
    VR_HIP(hipMalloc(&db, (size_t)nb * sizeof(float))); fr.p[1] = db;
    VR_HIP(hipMalloc(&df, (size_t)nf * sizeof(float))); fr.p[2] = df;
    VR_HIP(hipMemcpy(da, a, (size_t)na * sizeof(float), hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(db, b, (size_t)nb * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(xcorr_full_kernel, dim3((unsigned)nf), dim3(256), 0, 0, da, na, db, nb, df);
    VR_HIP(hipGetLastError());
    std::vector<float> full((size_t)nf);
    VR_HIP(hipMemcpy(full.data(), df, (size_t)nf * sizeof(float), hipMemcpyDeviceToHost));
    long long best = 0;
    for (long long k = 1; k < nf; ++k) if (full[(size_t)k] > full[(size_t)best]) best = k;       // np.argmax: first maximum
    *argmax_out = best;
}

}  // namespace vr
