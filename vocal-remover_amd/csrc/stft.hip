// STFT / iSTFT and the spectrogram-side glue of inference.Separator, all device-resident.
//
//  K1  stft         : librosa.stft semantics at the reference call site lib/spec_utils.py:26-31
//                     (periodic Hann, centre zero-padding n_fft/2, frames 1 + L//hop, complex64).
//  K2  mag_pad      : |X| into the zero-padded crop source + the two normalisers the reference
//                     uses (inference.py:74 max|X|; inference.py:87,94 numpy's lexicographic
//                     complex max).
//  K14 istft        : librosa.istft at lib/spec_utils.py:157-165 (irfft * window, overlap-add,
//                     / window-sum-square where > tiny, trim n_fft/2) -> hop*(T-1) samples.
//      apply_mask   : inference.py:26-40 (y = mask*X, v = (1-mask)*X; TTA average :97-98).
// One workgroup per frame; radix-2 FFT in LDS.  These stages are HBM-bound streaming work that is
// <1 % of the pipeline, kept simple and exact-ordered.
#include "kernels.h"

#include <cfloat>
#include <cstdlib>

namespace vr {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place radix-2 DIT FFT on x[0..n) (already in bit-reversed order). tw[k] = exp(-2*pi*i*k/n).
__device__ __forceinline__ void fft_lds(float2* x, const float2* __restrict__ tw, int n, int log2n) {
    const int half_n = n >> 1;
    for (int s = 0; s < log2n; ++s) {
        const int half = 1 << s;
        __syncthreads();
        for (int b = threadIdx.x; b < half_n; b += blockDim.x) {
            const int pos = b & (half - 1);
            const int i = ((b >> s) << (s + 1)) + pos;
            const int j = i + half;
            const float2 w = tw[pos << (log2n - 1 - s)];
            const float2 t = cmul(w, x[j]);
            const float2 u = x[i];
            x[i] = make_float2(u.x + t.x, u.y + t.y);
            x[j] = make_float2(u.x - t.x, u.y - t.y);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void stft_kernel(FFTPlan pl, const float* __restrict__ wave, long long L,
                                                   int hop, int T, float2* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) float2 xs[];
    const int n = pl.n_fft, t = blockIdx.x, ch = blockIdx.y;
    const float* wv = wave + (long long)ch * L;
    const long long start = (long long)t * hop - n / 2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long long p = start + i;
        const float v = (p >= 0 && p < L) ? wv[p] * pl.window[i] : 0.f;
        const int r = __brev((unsigned)i) >> (32 - pl.log2n);
        xs[r] = make_float2(v, 0.f);
    }
    fft_lds(xs, pl.twiddle, n, pl.log2n);
    const int bins = n / 2 + 1;
    for (int k = threadIdx.x; k < bins; k += blockDim.x)
        spec[((long long)ch * bins + k) * T + t] = xs[k];
}

void launch_stft(const FFTPlan& pl, const float* wave, long long L, int hop, int T, float2* spec, hipStream_t st);
static int tile_frames(const FFTPlan& pl, int extra_floats);
static bool tiled_signal_path(const FFTPlan& pl, int hop);
void launch_stft_tiled(const FFTPlan& pl, const float* wave, long long L, int T, float2* spec, hipStream_t st);

void launch_stft(const FFTPlan& pl, const float* wave, long long L, int hop, int T, float2* spec, hipStream_t st) {
    if (tiled_signal_path(pl, hop)) { launch_stft_tiled(pl, wave, L, T, spec, st); return; }
    VR_LAUNCH(stft_kernel, dim3(T, 2), dim3(256), pl.n_fft * sizeof(float2), st, pl, wave, L, hop, T, spec);
    VR_HIP(hipGetLastError());
}

// =====================================================================================================
// Frame-tiled STFT / iSTFT for hop == n_fft/2 (every reference call site: n_fft 2048, hop 1024).
//
// One workgroup = F consecutive frames of one channel.  Each frame is ONE n_fft/2-point complex FFT of the packed real
// signal z[m] = x[2m] + i x[2m+1]; the spectrum is unpacked with E[k] = (Z[k] + conj Z[M-k]) / 2,
// O[k] = -i (Z[k] - conj Z[M-k]) / 2, X[k] = E[k] + e^{-2 pi i k / N} O[k].  The F spectra are collected in an LDS tile
// [bins][F] and written as rows of F complex values (128 B for F = 16): the reference layout [ch][bins][T] is stored
// coalesced instead of one 8-byte element per 10 KB stride (measured before: 4.3x the algorithmic write traffic).
// The inverse reads the same tiles (optionally times the mask: inference.py:26-40 fused into the load), runs the packed
// inverse FFT, applies the window and overlap-adds inside the workgroup (hop = n_fft/2: a sample = second half of frame s
// + first half of frame s+1), so the [ch][T][n_fft] frame buffer and its 13x gather traffic are gone.
// =====================================================================================================
// radix-2 DIT on n points (bit-reversed input) by the `nt` threads of one frame group (tid = 0..nt-1); tw is the table
// of a 2^twshift times larger transform.  Barriers are workgroup-wide: every group runs the same number of stages.
__device__ __forceinline__ void fft_lds_sub(float2* x, const float2* __restrict__ tw, int n, int log2n, int twshift, int tid,
                                            int nt) {
    const int half_n = n >> 1;
    for (int s = 0; s < log2n; ++s) {
        const int half = 1 << s;
        __syncthreads();
        for (int b = tid; b < half_n; b += nt) {
            const int pos = b & (half - 1);
            const int i = ((b >> s) << (s + 1)) + pos;
            const int j = i + half;
            const float2 w = tw[(pos << (log2n - 1 - s)) << twshift];
            const float2 t = cmul(w, x[j]);
            const float2 u = x[i];
            x[i] = make_float2(u.x + t.x, u.y + t.y);
            x[j] = make_float2(u.x - t.x, u.y - t.y);
        }
    }
    __syncthreads();
}

constexpr int TG = 4;                        // frames in flight per workgroup (1024 threads = 4 groups of 256)

__global__ __launch_bounds__(1024) void stft_tile_kernel(FFTPlan pl, const float* __restrict__ wave, long long L, int T, int F,
                                                         float2* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    const int n = pl.n_fft, M = n >> 1, bins = M + 1, hop = M, logM = pl.log2n - 1;
    const int g = threadIdx.x >> 8, tid = threadIdx.x & 255;
    float2* zs = lds2 + (size_t)g * M;       // this group's FFT buffer
    float2* tile = lds2 + (size_t)TG * M;    // [bins][F]
    const int t0 = blockIdx.x * F, ch = blockIdx.y;
    const float* wv = wave + (long long)ch * L;
    const int nf = (T - t0) < F ? (T - t0) : F;
    for (int f0 = 0; f0 < nf; f0 += TG) {
        const int f = f0 + g;
        const bool live = f < nf;
        if (live) {
            const long long start = (long long)(t0 + f) * hop - M;      // centre = True: n_fft/2 zeros in front
            for (int m = tid; m < M; m += 256) {
                const long long p = start + 2 * m;
                const float v0 = (p >= 0 && p < L) ? wv[p] * pl.window[2 * m] : 0.f;
                const float v1 = (p + 1 >= 0 && p + 1 < L) ? wv[p + 1] * pl.window[2 * m + 1] : 0.f;
                zs[__brev((unsigned)m) >> (32 - logM)] = make_float2(v0, v1);
            }
        }
        fft_lds_sub(zs, pl.twiddle, M, logM, 1, tid, 256);
        if (live) {
            for (int k = tid; k < bins; k += 256) {
                const float2 zk = zs[k & (M - 1)], zm = zs[(M - k) & (M - 1)];
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 b = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                const float2 w = k < M ? pl.twiddle[k] : make_float2(-1.f, 0.f);
                const float2 wb = cmul(w, b);                              // X = E - i * w * B
                tile[k * F + f] = make_float2(e.x + wb.y, e.y - wb.x);
            }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < bins * F; idx += 1024) {
        const int k = idx / F, f = idx - k * F;
        if (f < nf) spec[((long long)ch * bins + k) * T + t0 + f] = tile[idx];
    }
}

// which: 0 = plain spectrogram (mask_a null) / instruments y = m X, 1 = vocals v = (1 - m) X
__global__ __launch_bounds__(1024) void istft_tile_kernel(FFTPlan pl, const float2* __restrict__ spec, int T, int S,
                                                          const float* __restrict__ ma, int Wa, const float* __restrict__ mb, int Wb,
                                                          int shift, const float* __restrict__ wgt, int which,
                                                          float* __restrict__ wave, long long out_len) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    const int n = pl.n_fft, M = n >> 1, bins = M + 1, logM = pl.log2n - 1, F = S + 1;
    const int g = threadIdx.x >> 8, tid = threadIdx.x & 255;
    float2* zs = lds2 + (size_t)g * M;                              // this group's FFT buffer = its time-domain frame afterwards
    float2* tile = lds2 + (size_t)TG * M;                           // [bins][F]
    float* prev = reinterpret_cast<float*>(tile + (size_t)bins * F);   // [M] windowed second half of the last frame of the previous round
    const int t0 = blockIdx.x * S, ch = blockIdx.y;
    const int nf = (T - t0) < F ? (T - t0) : F;
    for (int idx = threadIdx.x; idx < bins * F; idx += 1024) {
        const int k = idx / F, f = idx - k * F;
        float2 v = make_float2(0.f, 0.f);
        if (f < nf) {
            const long long row = (long long)ch * bins + k;
            const int t = t0 + f;
            v = spec[row * T + t];
            if (ma) {
                float m = ma[row * Wa + t];
                if (mb) m = (m + mb[row * Wb + t + shift]) * 0.5f;
                if (wgt) m += wgt[t] * (1.f - m);
                const float gm = which ? 1.f - m : m;                // y = m X, v = (1 - m) X  (inference.py:32-38; same form as apply_mask)
                v = make_float2(gm * v.x, gm * v.y);
            }
        }
        tile[idx] = v;
    }
    __syncthreads();
    const float invM = 1.f / (float)M;
    for (int f0 = 0; f0 < nf; f0 += TG) {
        const int f = f0 + g;
        const bool live = f < nf;
        if (live) {
            // Z[k] = E[k] + i O[k];  loaded as conj(Z) so that a forward FFT gives conj(M * z)
            for (int k = tid; k < M; k += 256) {
                float2 xk = tile[k * F + f], xm = tile[(M - k) * F + f];
                if (k == 0) { xk.y = 0.f; xm.y = 0.f; }             // numpy's irfft ignores the imaginary parts of DC and Nyquist
                const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
                const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
                const float2 w = pl.twiddle[k];                     // e^{-2 pi i k / N}; O = conj(w) * d
                const float2 o = make_float2(w.x * d.x + w.y * d.y, w.x * d.y - w.y * d.x);
                const float2 z = make_float2(e.x - o.y, e.y + o.x); // E + i O
                zs[__brev((unsigned)k) >> (32 - logM)] = make_float2(z.x, -z.y);
            }
        }
        fft_lds_sub(zs, pl.twiddle, M, logM, 1, tid, 256);
        // in place: zs[m] = (x[2m], x[2m+1]) windowed -> the group's buffer is the windowed frame, as floats [n_fft]
        if (live) {
            for (int m = tid; m < M; m += 256) {
                const float2 r = zs[m];
                zs[m] = make_float2(r.x * invM * pl.window[2 * m], -r.y * invM * pl.window[2 * m + 1]);
            }
        }
        __syncthreads();
        // segment s = t - 1 = second half of frame t-1 (previous group's buffer, or `prev` for group 0) + first half of frame t
        if (live && f > 0) {
            const int s = t0 + f - 1;
            const float* cur = reinterpret_cast<const float*>(zs);
            const float* before = g == 0 ? prev : reinterpret_cast<const float*>(zs - M) + M;
            for (int i = tid; i < M; i += 256) {
                const float w0 = pl.window[i], w2 = pl.window[i + M];
                const float ws = fmaf(w0, w0, w2 * w2);
                const float a = before[i] + cur[i];
                const long long p = (long long)s * M + i;
                if (p < out_len) wave[(long long)ch * out_len + p] = ws > FLT_MIN ? a / ws : a;
            }
        }
        __syncthreads();
        // carry: the last live frame of this round leaves its second half for the next round's group 0
        {
            const int last = (nf - f0 < TG ? nf - f0 : TG) - 1;
            if (g == last) {
                const float* cur = reinterpret_cast<const float*>(zs);
                for (int i = tid; i < M; i += 256) prev[i] = cur[M + i];
            }
        }
        __syncthreads();
    }
}

static int tile_frames(const FFTPlan& pl, int extra_floats) {
    // largest F <= 17 with TG FFT buffers + tile[bins][F] (+ extra) inside 150 KB of LDS
    const int M = pl.n_fft / 2, bins = M + 1;
    long long budget = 150 * 1024 - (long long)TG * M * 8 - (long long)extra_floats * 4;
    int F = (int)(budget / ((long long)bins * 8));
    return F > 17 ? 17 : F;
}

static bool tiled_signal_path(const FFTPlan& pl, int hop) {
    static const bool on = !getenv("VR_NO_TILED_STFT");
    return on && hop * 2 == pl.n_fft && pl.n_fft >= 128 && tile_frames(pl, pl.n_fft / 2) >= 3;
}

void launch_istft_masked(const FFTPlan& pl, const float2* spec, int hop, int T, const float* mask_a, int Wa, const float* mask_b,
                         int Wb, int shift, const float* wgt, int which, float* wave, hipStream_t st) {
    const int M = pl.n_fft / 2, bins = M + 1;
    const long long out_len = (long long)hop * (T - 1);
    if (out_len <= 0) return;
    const int F = tile_frames(pl, M);
    const int S = F - 1;
    const size_t lds = (size_t)TG * M * 8 + (size_t)bins * F * 8 + (size_t)M * 4;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(istft_tile_kernel), 160 * 1024);
    // per stem: the complex spectrogram (8 B per bin-frame), the mask(s) (4 B), hop samples written per frame, two channels
    prof_note(0.0, 2.0 * ((double)bins * T * (8.0 + 4.0 * (mask_b ? 2 : 1)) + 4.0 * (double)out_len));
    VR_LAUNCH(istft_tile_kernel, dim3((unsigned)((T - 1 + S - 1) / S), 2), dim3(1024), lds, st, pl, spec, T, S, mask_a, Wa,
                       mask_b, Wb, shift, wgt, which, wave, out_len);
    VR_HIP(hipGetLastError());
}

// irfft(spec[:, t]) * window -> frames[ch][t][0..n)
__global__ __launch_bounds__(256) void istft_frame_kernel(FFTPlan pl, const float2* __restrict__ spec, int T,
                                                          float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float2 xs[];
    const int n = pl.n_fft, t = blockIdx.x, ch = blockIdx.y;
    const int bins = n / 2 + 1;
    const float2* sp = spec + (long long)ch * bins * T + t;
    // inverse via forward FFT of the conjugate spectrum (Hermitian-extended); imaginary parts of
    // the DC and Nyquist bins are ignored like numpy's irfft does.
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        float2 v;
        if (k < bins) {
            v = sp[(long long)k * T];
            v.y = -v.y;
            if (k == 0 || k == n / 2) v.y = 0.f;
        } else {
            v = sp[(long long)(n - k) * T];      // conj(conj(X[n-k])) = X[n-k]
        }
        const int r = __brev((unsigned)k) >> (32 - pl.log2n);
        xs[r] = v;
    }
    fft_lds(xs, pl.twiddle, n, pl.log2n);
    const float inv = 1.f / (float)n;
    float* fr = frames + ((long long)ch * T + t) * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) fr[i] = xs[i].x * inv * pl.window[i];
}

__global__ void istft_ola_kernel(FFTPlan pl, const float* __restrict__ frames, int hop, int T, long long out_len,
                                 float* __restrict__ wave) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = blockIdx.y;
    if (p >= out_len) return;
    const int n = pl.n_fft;
    const long long q = p + n / 2;
    long long t_hi = q / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    long long t_lo = (q - n + hop) / hop;          // smallest t with t*hop + n > q
    if (q - n + 1 <= 0) t_lo = 0;
    if (t_lo < 0) t_lo = 0;
    float acc = 0.f, wss = 0.f;
    for (long long t = t_lo; t <= t_hi; ++t) {
        const int i = (int)(q - t * hop);
        if (i < 0 || i >= n) continue;
        acc += frames[((long long)ch * T + t) * n + i];
        const float w = pl.window[i];
        wss = fmaf(w, w, wss);
    }
    wave[(long long)ch * out_len + p] = (wss > FLT_MIN) ? acc / wss : acc;
}

void launch_stft_tiled(const FFTPlan& pl, const float* wave, long long L, int T, float2* spec, hipStream_t st) {
    const int M = pl.n_fft / 2, bins = M + 1;
    int F = tile_frames(pl, 0);
    if (F > 16) F = 16;
    F = F / TG * TG;                                     // whole rounds of TG frames
    const size_t lds = (size_t)TG * M * 8 + (size_t)bins * F * 8;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(stft_tile_kernel), 160 * 1024);
    prof_note(0.0, 2.0 * (4.0 * (double)L + 8.0 * (double)bins * T));              // unique audio read once, complex64 spectrogram written
    VR_LAUNCH(stft_tile_kernel, dim3((unsigned)((T + F - 1) / F), 2), dim3(1024), lds, st, pl, wave, L, T, F, spec);
    VR_HIP(hipGetLastError());
}

bool istft_masked_available(const FFTPlan& pl, int hop) { return tiled_signal_path(pl, hop); }

void launch_istft(const FFTPlan& pl, const float2* spec, int hop, int T, float* frames, float* wave, hipStream_t st) {
    if (tiled_signal_path(pl, hop)) {
        launch_istft_masked(pl, spec, hop, T, nullptr, 0, nullptr, 0, 0, nullptr, 0, wave, st);
        return;
    }
    VR_LAUNCH(istft_frame_kernel, dim3(T, 2), dim3(256), pl.n_fft * sizeof(float2), st, pl, spec, T, frames);
    VR_HIP(hipGetLastError());
    const long long out_len = (long long)hop * (T - 1);
    if (out_len > 0) {
        VR_LAUNCH(istft_ola_kernel, dim3((unsigned)((out_len + 255) / 256), 2), dim3(256), 0, st, pl, frames,
                           hop, T, out_len, wave);
        VR_HIP(hipGetLastError());
    }
}

// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ord32(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord32(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}


// One workgroup per (channel, bin) row: |X| into the padded crop source (row-contiguous loads and stores, no 64-bit
// division per element) and the row's two maxima into part[row] -- no atomics; coef_affine_kernel reduces the 2 x bins
// partials.  (Round 1: 16 k same-address atomics and a flat 64-bit index made this 190 us for 34 MB.)
__global__ __launch_bounds__(256) void mag_pad_kernel(const float2* __restrict__ spec, int T, float* __restrict__ mag_pad,
                                                      int Wpad, int pad_l, unsigned long long* __restrict__ part) {
    const int row = blockIdx.x;
    const float2* sp = spec + (long long)row * T;
    float* dst = mag_pad + (long long)row * Wpad + pad_l;
    float mx = 0.f;
    unsigned long long key = ((unsigned long long)ord32(0.f) << 32) | ord32(0.f);   // the zero padding is part of the reduced array
    for (int t = threadIdx.x; t < T; t += 256) {
        const float2 z = sp[t];
        const float m = sqrtf(z.x * z.x + z.y * z.y);
        dst[t] = m;
        mx = fmaxf(mx, m);
        const unsigned long long k = ((unsigned long long)ord32(z.x) << 32) | ord32(z.y);
        key = k > key ? k : key;
    }
    __shared__ float rmx[4];
    __shared__ unsigned long long rkey[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) { rmx[threadIdx.x >> 6] = mx; rkey[threadIdx.x >> 6] = key; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { mx = fmaxf(mx, rmx[i]); key = rkey[i] > key ? rkey[i] : key; }
        part[2 * row] = (unsigned long long)__float_as_uint(mx);
        part[2 * row + 1] = key;
    }
}

void launch_mag_pad(const float2* spec, int bins, int T, float* mag_pad, int Wpad, int pad_l, unsigned* stats,
                    hipStream_t st) {
    prof_note(0.0, 2.0 * (double)bins * (8.0 * T + 4.0 * Wpad));
    VR_LAUNCH(mag_pad_kernel, dim3(2 * bins), dim3(256), 0, st, spec, T, mag_pad, Wpad, pad_l,
                       reinterpret_cast<unsigned long long*>(stats) + 2);
    VR_HIP(hipGetLastError());
}

// stats layout: [0..1] legacy words, then 2 x bins rows of (max |X| bits, lexicographic complex key) partials
__global__ __launch_bounds__(256) void coef_affine_kernel(const unsigned* stats, int rows, int mode, float* aff) {
    const unsigned long long* part = reinterpret_cast<const unsigned long long*>(stats) + 2;
    unsigned mxb = 0u;
    unsigned long long key = ((unsigned long long)ord32(0.f) << 32) | ord32(0.f);
    for (int r = threadIdx.x; r < rows; r += 256) {
        const unsigned b = (unsigned)part[2 * r];
        mxb = b > mxb ? b : mxb;                               // non-negative floats order like their bit patterns
        key = part[2 * r + 1] > key ? part[2 * r + 1] : key;
    }
    __shared__ unsigned rm[256];
    __shared__ unsigned long long rk[256];
    rm[threadIdx.x] = mxb; rk[threadIdx.x] = key;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            rm[threadIdx.x] = rm[threadIdx.x + off] > rm[threadIdx.x] ? rm[threadIdx.x + off] : rm[threadIdx.x];
            rk[threadIdx.x] = rk[threadIdx.x + off] > rk[threadIdx.x] ? rk[threadIdx.x + off] : rk[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float coef;
        if (mode == 0) {
            coef = __uint_as_float(rm[0]);
        } else {
            const float re = unord32((unsigned)(rk[0] >> 32)), im = unord32((unsigned)(rk[0] & 0xffffffffu));
            coef = sqrtf(re * re + im * im);
        }
        const float s = 1.f / coef;
        aff[0] = s; aff[1] = 0.f; aff[2] = s; aff[3] = 0.f;
    }
}
void launch_coef_affine(const unsigned* stats, int rows, int mode, float* aff, hipStream_t st) {
    VR_LAUNCH(coef_affine_kernel, dim3(1), dim3(256), 0, st, stats, rows, mode, aff);
    VR_HIP(hipGetLastError());
}

// per-frame minimum of the final mask over (channel, bin): input of spec_utils.merge_artifacts
// (lib/spec_utils.py:64).  One workgroup per 64 frames; lanes along time (coalesced rows).
__global__ __launch_bounds__(256) void frame_min_kernel(int rows, int T, const float* __restrict__ ma, int Wa,
                                                        const float* __restrict__ mb, int Wb, int shift,
                                                        float* __restrict__ fmin) {
    __shared__ float red[4][64];
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    float m = 3.4e38f;
    if (t < T) {
        for (int r = part; r < rows; r += 4) {
            float v = ma[(long long)r * Wa + t];
            if (mb) v = (v + mb[(long long)r * Wb + t + shift]) * 0.5f;
            m = fminf(m, v);
        }
    }
    red[part][threadIdx.x & 63] = m;
    __syncthreads();
    if (part == 0 && t < T) fmin[t] = fminf(fminf(red[0][threadIdx.x], red[1][threadIdx.x]), fminf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

void launch_frame_min(int bins, int T, const float* mask_a, int Wa, const float* mask_b, int Wb, int shift, float* fmin,
                      hipStream_t st) {
    VR_LAUNCH(frame_min_kernel, dim3((T + 63) / 64), dim3(256), 0, st, 2 * bins, T, mask_a, Wa, mask_b, Wb, shift, fmin);
    VR_HIP(hipGetLastError());
}

__global__ void apply_mask_kernel(const float2* __restrict__ spec, int bins, int T, const float* __restrict__ ma,
                                  int Wa, const float* __restrict__ mb, int Wb, int shift, const float* __restrict__ wgt,
                                  float2* __restrict__ y, float2* __restrict__ v) {
    const long long total = 2LL * bins * T;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int t = (int)(gid % T);
    const long long row = gid / T;
    float m = ma[row * Wa + t];
    if (mb) m = (m + mb[row * Wb + t + shift]) * 0.5f;
    if (wgt) m += wgt[t] * (1.f - m);            // merge_artifacts: y_mask += weight * (1 - y_mask)
    const float2 z = spec[gid];
    y[gid] = make_float2(m * z.x, m * z.y);
    const float im = 1.f - m;
    v[gid] = make_float2(im * z.x, im * z.y);
}

void launch_apply_mask(const float2* spec, int bins, int T, const float* mask_a, int Wa, const float* mask_b, int Wb,
                       int shift, const float* wgt, float2* y, float2* v, hipStream_t st) {
    const long long total = 2LL * bins * T;
    VR_LAUNCH(apply_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, spec, bins, T, mask_a,
                       Wa, mask_b, Wb, shift, wgt, y, v);
    VR_HIP(hipGetLastError());
}

}  // namespace vr
