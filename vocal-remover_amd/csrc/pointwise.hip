// HBM-bound kernels around the conv stack: thin 1x1 convs (Cout 1/2), the mask head, frequency
// average pool, BatchNorm statistics bookkeeping.  All are streaming kernels: 16-byte loads along
// the contiguous time axis, one pass over the data, nothing re-read.
#include <cstdlib>

#include "kernels.h"

namespace vr {

__device__ __forceinline__ float act1(float v, float slope) { return v > 0.f ? v : v * slope; }

// (scale, shift) = (1, 0) and a post factor of 1 for tensors without them: lets a kernel load its constants unconditionally, in the
// same burst as its pixels (a load behind `if (aff)` is followed by a vmcnt(0): one extra memory round trip per constant)
__device__ const float kIdentityAffine[2] = {1.f, 0.f};
__device__ const float kOne[1] = {1.f};

__device__ __forceinline__ void load_aff(const Tensor& x, int h, int c, float& sc, float& sh) {
    const float* aff = (h < x.hsplit) ? x.aff0 : x.aff1;
    sc = 1.f; sh = 0.f;
    if (aff) { sc = aff[2 * c]; sh = aff[2 * c + 1]; }
}

// ---------------------------------------------------------------------------------------------------
// Thin 1x1 conv over channels, CO outputs, 4 consecutive frames per thread.
//   FINAL: sigmoid + window/crop + replicate rows (mask head, lib/nets.py:109-115,127-128)
//   else : raw store to out[n][h][w] (+ per-block sum/sumsq partials for BatchNorm batch stats)
// ---------------------------------------------------------------------------------------------------
template <int CO, bool FINAL>
__global__ __launch_bounds__(256) void thin_conv_kernel(Tensor x, const float* __restrict__ w, HeadDst d,
                                                        float* out, float* part, const float* __restrict__ epi) {
    const int W4 = x.W >> 2;
    const long long total = (long long)x.N * x.H * W4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    if (gid < total) {
        const int w4 = (int)(gid % W4);
        const long long t = gid / W4;
        const int h = (int)(t % x.H);
        const int n = (int)(t / x.H);
        float acc[CO][4];
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
        const float* base = x.p + (long long)n * x.sN + (long long)h * x.sH + w4 * 4;
        for (int c = 0; c < x.C; ++c) {
            float sc, sh;
            load_aff(x, h, c, sc, sh);
            const float4 r = *reinterpret_cast<const float4*>(base + (long long)c * x.sC);
            const float v[4] = {act1(fmaf(r.x, sc, sh), x.slope), act1(fmaf(r.y, sc, sh), x.slope),
                                act1(fmaf(r.z, sc, sh), x.slope), act1(fmaf(r.w, sc, sh), x.slope)};
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float wc = w[o * x.C + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(wc, v[j], acc[o][j]);
            }
        }
        if constexpr (FINAL) {
#pragma unroll
            for (int o = 0; o < CO; ++o) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int wcol = w4 * 4 + j;
                    if (wcol < d.w_lo || wcol >= d.w_hi) continue;
                    const float m = 1.f / (1.f + __expf(-acc[o][j]));
                    float* dst = d.p + (long long)n * d.dN + (long long)o * d.dC + (wcol - d.w_lo);
                    dst[(long long)h * d.dH] = m;
                    if (h == x.H - 1)
                        for (int e = 1; e <= d.pad_rows; ++e) dst[(long long)(h + e) * d.dH] = m;
                }
            }
        } else {
            if (epi) {           // eval: the single-channel BatchNorm (folded scale, shift) + ReLU here, so the stored row is final
                const float esc = epi[0], esh = epi[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][j] = fmaxf(fmaf(acc[0][j], esc, esh), 0.f);
            }
            float4 o4 = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            *reinterpret_cast<float4*>(out + ((long long)n * x.H + h) * x.W + w4 * 4) = o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += acc[0][j]; s2 = fmaf(acc[0][j], acc[0][j], s2); }
        }
    }
    if constexpr (!FINAL) {
        if (part) {
            __shared__ float red[8];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
            const int wave = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                part[blockIdx.x * 2 + 0] = red[0] + red[2] + red[4] + red[6];
                part[blockIdx.x * 2 + 1] = red[1] + red[3] + red[5] + red[7];
            }
        }
    }
}

static void check_vec4(const Tensor& x) {
    VR_CHECK(x.W % 4 == 0 && x.sN % 4 == 0 && x.sC % 4 == 0 && x.sH % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(x.p) & 15) == 0,
             -2, "thin conv needs 16-byte aligned rows (frames % 4 == 0)");
}

void launch_head_sigmoid(const Tensor& x, const float* w, const HeadDst& d, hipStream_t st) {
    check_vec4(x);
    const long long total = (long long)x.N * x.H * (x.W / 4);
    const int grid = (int)((total + 255) / 256);
    prof_note(2.0 * 2 * x.C * (double)x.N * x.H * x.W, 4.0 * ((double)x.N * x.C * x.H * x.W + 2.0 * x.N * x.H * x.W));   // C -> 2 head
    VR_LAUNCH((thin_conv_kernel<2, true>), dim3(grid), dim3(256), 0, st, x, w, d, nullptr, nullptr, nullptr);
    VR_HIP(hipGetLastError());
}

int launch_squeeze_conv(const Tensor& x, const float* w, float* out, float* part, bool dry, hipStream_t st, const float* epi) {
    const long long total = (long long)x.N * x.H * (x.W / 4);
    const int grid = (int)((total + 255) / 256);
    if (dry) return grid;
    check_vec4(x);
    HeadDst d{};
    prof_note(2.0 * x.C * (double)x.N * x.H * x.W, 4.0 * ((double)x.N * x.C * x.H * x.W + (double)x.N * x.H * x.W));
    VR_LAUNCH((thin_conv_kernel<1, false>), dim3(grid), dim3(256), 0, st, x, w, d, out, part, epi);
    VR_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------------------------------
__global__ void avgpool_h_kernel(Tensor x, float* out) {
    const int total = x.N * x.C * x.W;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = gid % x.W;
    const int c = (gid / x.W) % x.C;
    const int n = gid / (x.W * x.C);
    const float* base = x.p + (long long)n * x.sN + (long long)c * x.sC + w;
    float s = 0.f;
    for (int h = 0; h < x.H; ++h) {
        float sc, sh;
        load_aff(x, h, c, sc, sh);
        s += act1(fmaf(base[(long long)h * x.sH], sc, sh), x.slope);
    }
    out[gid] = s / (float)x.H;
}

void launch_avgpool_h(const Tensor& x, float* out, hipStream_t st) {
    const int total = x.N * x.C * x.W;
    VR_LAUNCH(avgpool_h_kernel, dim3((total + 255) / 256), dim3(256), 0, st, x, out);
    VR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// Eval mode: affine = (w / sqrt(running_var + eps), b - running_mean * scale) for every BatchNorm.
__global__ void bn_fold_eval_kernel(const BNFoldDesc* descs, float eps) {
    const BNFoldDesc d = descs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = d.bcast ? d.bcast : d.C;
    if (i >= rows) return;
    const int c = d.bcast ? 0 : i;
    const float scale = d.w[c] / sqrtf(d.rv[c] + eps);
    d.affine[2 * i] = scale;
    d.affine[2 * i + 1] = d.b[c] - d.rm[c] * scale;
}

void launch_bn_fold_eval(const BNFoldDesc* d_descs, int ndesc, int maxC, float eps, hipStream_t st) {
    VR_LAUNCH(bn_fold_eval_kernel, dim3((maxC + 127) / 128, ndesc), dim3(128), 0, st, d_descs, eps);
    VR_HIP(hipGetLastError());
}

// Train mode: reduce per-block partials (fp64), produce the affine, update running stats with the
// unbiased variance (torch BatchNorm semantics, momentum 0.1), save mean / invstd for backward.
__global__ __launch_bounds__(256) void bn_finalize_kernel(BNFinalizeArgs a) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < a.nparts; i += 256) {
        s1 += (double)a.part[(long long)i * a.pstride + c * 2 + 0];
        s2 += (double)a.part[(long long)i * a.pstride + c * 2 + 1];
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { r1[threadIdx.x] += r1[threadIdx.x + off]; r2[threadIdx.x] += r2[threadIdx.x + off]; }
        __syncthreads();
    }
    __shared__ float sh_scale, sh_shift;
    if (threadIdx.x == 0) {
        const double mean = r1[0] / a.count;
        double var = r2[0] / a.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float scale = a.w[c] * invstd;
        const float shift = a.b[c] - (float)mean * scale;
        sh_scale = scale; sh_shift = shift;
        if (!a.broadcast) { a.affine[2 * c] = scale; a.affine[2 * c + 1] = shift; }
        if (a.save_mean) { a.save_mean[c] = (float)mean; a.save_invstd[c] = invstd; }
        const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
        a.rm[c] = (1.f - a.momentum) * a.rm[c] + a.momentum * (float)mean;
        a.rv[c] = (1.f - a.momentum) * a.rv[c] + a.momentum * (float)unbiased;
    }
    if (a.broadcast) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.broadcast; i += 256) { a.affine[2 * i] = sh_scale; a.affine[2 * i + 1] = sh_shift; }
    }
}

void launch_bn_finalize(const BNFinalizeArgs& a, hipStream_t st) {
    VR_LAUNCH(bn_finalize_kernel, dim3(a.C), dim3(256), 0, st, a);
    VR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
__global__ void rows_affine_relu_kernel(const float* x, float* out, const float* aff, int R, int W, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int r = (int)((gid / W) % R);
    const float v = fmaf(x[gid], aff[2 * r], aff[2 * r + 1]);
    out[gid] = v > 0.f ? v : 0.f;
}

void launch_rows_affine_relu(const float* x, float* out, const float* aff, int N, int R, int W, hipStream_t st) {
    const long long total = (long long)N * R * W;
    VR_LAUNCH(rows_affine_relu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, out, aff, R, W, total);
    VR_HIP(hipGetLastError());
}

__global__ void add_kernel(const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

void launch_add(const float* a, const float* b, float* out, int n, hipStream_t st) {
    VR_LAUNCH(add_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, out, n);
    VR_HIP(hipGetLastError());
}

__global__ void materialize_kernel(Tensor x, float* out, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % x.W);
    long long t = gid / x.W;
    const int h = (int)(t % x.H); t /= x.H;
    const int c = (int)(t % x.C);
    const int n = (int)(t / x.C);
    float sc, sh;
    load_aff(x, h, c, sc, sh);
    const float raw = x.p[(long long)n * x.sN + (long long)c * x.sC + (long long)h * x.sH + w];
    const float post = x.post ? x.post[n * x.C + c] : 1.f;        // Dropout2d keep-mask / 0.9
    out[gid] = act1(fmaf(raw, sc, sh), x.slope) * post;
}

// Bilinear x2 upsample, align_corners=True (lib/layers.py:52 F.interpolate), of the activated tensor
// into a dense [N][C][2H][2W] buffer: eval mode materialises the decoder's upsampled input once so
// that the consuming conv can take the LDS-DMA path (plain input, no arithmetic in the loader).
// Same arithmetic (tap weights and summation order) as the fused loader in conv_stage.h.
template <int V>      // V consecutive output columns per thread (4: one 16-B store; 2: odd source widths)
__global__ __launch_bounds__(256) void upsample2x_kernel(Tensor x, float* __restrict__ out, float rh, float rw,
                                                         long long total4) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total4) return;
    const int W2 = 2 * x.W, H2 = 2 * x.H, Q = W2 / V;
    const int wq = (int)(gid % Q);
    long long t = gid / Q;
    const int hi = (int)(t % H2); t /= H2;
    const int c = (int)(t % x.C);
    const int n = (int)(t / x.C);
    const float h1r = rh * (float)hi;
    const int h1 = (int)h1r;
    const int h1p = (h1 < x.H - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float* r0 = x.p + (long long)n * x.sN + (long long)c * x.sC + (long long)h1 * x.sH;
    const float* r1 = r0 + (long long)h1p * x.sH;
    float sc0, sh0, sc1, sh1;
    load_aff(x, h1, c, sc0, sh0);
    load_aff(x, h1 + h1p, c, sc1, sh1);
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int wi = wq * V + j;
        const float w1r = rw * (float)wi;
        const int w1 = (int)w1r;
        const int w1p = (w1 < x.W - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const float v00 = act1(fmaf(r0[w1], sc0, sh0), x.slope);
        const float v01 = act1(fmaf(r0[w1 + w1p], sc0, sh0), x.slope);
        const float v10 = act1(fmaf(r1[w1], sc1, sh1), x.slope);
        const float v11 = act1(fmaf(r1[w1 + w1p], sc1, sh1), x.slope);
        o[j] = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
    }
    const float post = x.post ? x.post[n * x.C + c] : 1.f;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] *= post;
    if constexpr (V == 4) reinterpret_cast<float4*>(out)[gid] = make_float4(o[0], o[1], o[2], o[3]);
    else reinterpret_cast<float2*>(out)[gid] = make_float2(o[0], o[1]);
}

// Row-tiled form (even source widths): a thread group = one output row of one (n, c) plane, so the row weights, the two
// source rows, the affine and the dropout factor are wave-uniform and no per-element division is left; a thread makes four
// consecutive output columns (one 16-byte store) from at most four consecutive source columns per row.
__global__ __launch_bounds__(256) void upsample2x_rows_kernel(Tensor x, float* __restrict__ out, float rh, float rw, int qp_log2,
                                                              long long nrows) {
    const int W2 = 2 * x.W, H2 = 2 * x.H;
    // 2^qp_log2 threads per output row (>= quads of 4 columns, capped at 256), 256 >> qp_log2 rows per workgroup
    const int q = (qp_log2 >= 8 ? blockIdx.y * 256 : 0) + (threadIdx.x & ((1 << qp_log2) - 1));
    const long long row = (long long)blockIdx.x * (256 >> qp_log2) + (threadIdx.x >> qp_log2);
    if (4 * q >= W2 || row >= nrows) return;
    const int hi = (int)(row % H2);
    const int pc = (int)(row / H2);                            // n * C + c
    const int c = pc % x.C, n = pc / x.C;
    const float h1r = rh * (float)hi;
    const int h1 = (int)h1r;
    const int h1p = (h1 < x.H - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float* r0 = x.p + (long long)n * x.sN + (long long)c * x.sC + (long long)h1 * x.sH;
    const float* r1 = r0 + (long long)h1p * x.sH;
    // the affine of the two source rows and the dropout factor through unconditional pointers: one burst with the eight pixels
    const float* af0 = (h1 < x.hsplit) ? x.aff0 : x.aff1;
    const float* af1 = (h1 + h1p < x.hsplit) ? x.aff0 : x.aff1;
    const float* ap0 = af0 ? af0 + 2 * c : kIdentityAffine;
    const float* ap1 = af1 ? af1 + 2 * c : kIdentityAffine;
    const float* pp = x.post ? x.post + n * x.C + c : kOne;
    // source columns of the quad: floor(rw * wi) for wi = 4q .. 4q+3 lie in [wb, wb + 2] with wb = floor(rw * 4q); + 1 neighbour
    const int wb = (int)(rw * (float)(4 * q));
    float ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ws = wb + j < x.W ? wb + j : x.W - 1;
        ra[j] = r0[ws];
        rb[j] = r1[ws];
    }
    const float sc0 = ap0[0], sh0 = ap0[1], sc1 = ap1[0], sh1 = ap1[1], post = pp[0];
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = act1(fmaf(ra[j], sc0, sh0), x.slope);
        b[j] = act1(fmaf(rb[j], sc1, sh1), x.slope);
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wi = 4 * q + j;
        const float w1r = rw * (float)wi;
        const int w1 = (int)w1r;
        const int w1p = (w1 < x.W - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const int d = w1 - wb;                                   // 0, 1 or 2
        const float v00 = d == 0 ? a[0] : (d == 1 ? a[1] : a[2]);
        const float v01 = w1p ? (d == 0 ? a[1] : (d == 1 ? a[2] : a[3])) : v00;
        const float v10 = d == 0 ? b[0] : (d == 1 ? b[1] : b[2]);
        const float v11 = w1p ? (d == 0 ? b[1] : (d == 1 ? b[2] : b[3])) : v10;
        o[j] = (h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11)) * post;
    }
    reinterpret_cast<float4*>(out + ((long long)pc * H2 + hi) * W2)[q] = make_float4(o[0], o[1], o[2], o[3]);
}

// LDS form of the row-tiled kernel (round 5): a workgroup makes 256 >> qp consecutive output rows of ONE plane; the source rows under
// them (at most half as many + 2) are fetched once with aligned 16-byte loads, BatchNorm affine + activation applied on the way in, and
// every thread then takes its eight source values from LDS -- one global load per thread at most instead of eight 4-byte ones.
// Same arithmetic, same order as upsample2x_rows_kernel (bit-equal results).  Needs W % 4 == 0, 16-byte aligned rows, 2 H % rows == 0.
__global__ __launch_bounds__(256) void upsample2x_lds_kernel(Tensor x, float* __restrict__ out, float rh, float rw, int qp_log2) {
    __shared__ __attribute__((aligned(16))) float L[1024];                   // (128 >> qp + 2) rows x W <= 768 floats
    const int W2 = 2 * x.W, H2 = 2 * x.H;
    const int rpb = 256 >> qp_log2;
    const long long row0 = (long long)blockIdx.x * rpb;
    const int pc = (int)(row0 / H2);                                        // n * C + c
    const int hi0 = (int)(row0 - (long long)pc * H2);
    const int c = pc % x.C, n = pc / x.C;
    const int hmin = (int)(rh * (float)hi0);
    int hmax = (int)(rh * (float)(hi0 + rpb - 1)) + 1;
    hmax = hmax < x.H - 1 ? hmax : x.H - 1;
    const int nsrc = hmax - hmin + 1, Q = x.W >> 2;
    const float* base = x.p + (long long)n * x.sN + (long long)c * x.sC;
    for (int e = threadIdx.x; e < nsrc * Q; e += 256) {
        const int r = e / Q, q = e - r * Q;
        const int h = hmin + r;
        const float* af = (h < x.hsplit) ? x.aff0 : x.aff1;
        const float sc = af ? af[2 * c] : 1.f, sh = af ? af[2 * c + 1] : 0.f;
        const float4 v = *reinterpret_cast<const float4*>(base + (long long)h * x.sH + 4 * q);
        *reinterpret_cast<float4*>(L + r * x.W + 4 * q) = make_float4(act1(fmaf(v.x, sc, sh), x.slope), act1(fmaf(v.y, sc, sh), x.slope),
                                                                      act1(fmaf(v.z, sc, sh), x.slope), act1(fmaf(v.w, sc, sh), x.slope));
    }
    __syncthreads();
    const int q = threadIdx.x & ((1 << qp_log2) - 1);
    const int hi = hi0 + (threadIdx.x >> qp_log2);
    if (4 * q >= W2) return;
    const float h1r = rh * (float)hi;
    const int h1 = (int)h1r;
    const int h1p = (h1 < x.H - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float* r0 = L + (h1 - hmin) * x.W;
    const float* r1 = r0 + h1p * x.W;
    const float post = x.post ? x.post[n * x.C + c] : 1.f;
    const int wb = (int)(rw * (float)(4 * q));
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ws = wb + j < x.W ? wb + j : x.W - 1;
        a[j] = r0[ws];
        b[j] = r1[ws];
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wi = 4 * q + j;
        const float w1r = rw * (float)wi;
        const int w1 = (int)w1r;
        const int w1p = (w1 < x.W - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const int d = w1 - wb;                                   // 0, 1 or 2
        const float v00 = d == 0 ? a[0] : (d == 1 ? a[1] : a[2]);
        const float v01 = w1p ? (d == 0 ? a[1] : (d == 1 ? a[2] : a[3])) : v00;
        const float v10 = d == 0 ? b[0] : (d == 1 ? b[1] : b[2]);
        const float v11 = w1p ? (d == 0 ? b[1] : (d == 1 ? b[2] : b[3])) : v10;
        o[j] = (h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11)) * post;
    }
    reinterpret_cast<float4*>(out + ((long long)pc * H2 + hi) * W2)[q] = make_float4(o[0], o[1], o[2], o[3]);
}

void launch_upsample2x(const Tensor& x, float* out, hipStream_t st) {
    const float rh = (x.H > 0) ? (float)(x.H - 1) / (float)(2 * x.H - 1) : 0.f;
    const float rw = (x.W > 0) ? (float)(x.W - 1) / (float)(2 * x.W - 1) : 0.f;
    const long long total = (long long)x.N * x.C * x.H * x.W * 4;
    static const bool rows = !getenv("VR_NO_UP_ROWS");
    const long long nrows = (long long)x.N * x.C * 2 * x.H;
    prof_note(0.0, 4.0 * 5.0 * (double)x.N * x.C * x.H * x.W);          // reads the low-resolution tensor, writes 4x as much
    if (rows && (x.W & 1) == 0 && x.W >= 8 && nrows < 0x7FFFFFFFLL && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int quads = 2 * x.W / 4;
        int qp = 2;                                            // threads per row: power of two >= quads, 4 .. 256
        while ((1 << qp) < quads && qp < 8) ++qp;
        const int rpb = 256 >> qp;
        static const bool lds_on = !(getenv("VR_UP_LDS") && atoi(getenv("VR_UP_LDS")) == 0);
        if (lds_on && qp <= 7 && (x.W & 3) == 0 && (x.sH & 3) == 0 && (x.sC & 3) == 0 && (x.sN & 3) == 0 && (2 * x.H) % rpb == 0 &&
            (reinterpret_cast<uintptr_t>(x.p) & 15) == 0 && (128 / (1 << qp) + 3) * x.W <= 1024) {
            VR_LAUNCH(upsample2x_lds_kernel, dim3((unsigned)(nrows / rpb)), dim3(256), 0, st, x, out, rh, rw, qp);
            VR_HIP(hipGetLastError());
            return;
        }
        VR_LAUNCH(upsample2x_rows_kernel, dim3((unsigned)((nrows + rpb - 1) / rpb), qp >= 8 ? (quads + 255) / 256 : 1), dim3(256),
                           0, st, x, out, rh, rw, qp, nrows);
        VR_HIP(hipGetLastError());
        return;
    }
    if ((x.W & 1) == 0) {
        const long long tv = total / 4;
        VR_LAUNCH(upsample2x_kernel<4>, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, st, x, out, rh, rw, tv);
    } else {
        const long long tv = total / 2;
        VR_LAUNCH(upsample2x_kernel<2>, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, st, x, out, rh, rw, tv);
    }
    VR_HIP(hipGetLastError());
}

// Four consecutive frames per thread (16-B load and store) when the rows allow it.
__global__ __launch_bounds__(256) void materialize4_kernel(Tensor x, float* __restrict__ out, long long total4) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total4) return;
    const int Q = x.W >> 2;
    const int wq = (int)(gid % Q);
    long long t = gid / Q;
    const int h = (int)(t % x.H); t /= x.H;
    const int c = (int)(t % x.C);
    const int n = (int)(t / x.C);
    float sc, sh;
    load_aff(x, h, c, sc, sh);
    const float post = x.post ? x.post[n * x.C + c] : 1.f;
    const float4 r = *reinterpret_cast<const float4*>(x.p + (long long)n * x.sN + (long long)c * x.sC + (long long)h * x.sH + 4 * wq);
    float4 o;
    o.x = act1(fmaf(r.x, sc, sh), x.slope) * post;
    o.y = act1(fmaf(r.y, sc, sh), x.slope) * post;
    o.z = act1(fmaf(r.z, sc, sh), x.slope) * post;
    o.w = act1(fmaf(r.w, sc, sh), x.slope) * post;
    reinterpret_cast<float4*>(out)[gid] = o;
}

// The same with the plane index (n, c) on blockIdx.y: one division per thread (by the quads of a row) instead of three by run-time values
// -- gid % Q, / H, % C cost ~80 VALU instructions for the 32 bytes a thread moves (round 5).
__global__ __launch_bounds__(256) void materialize4p_kernel(Tensor x, float* __restrict__ out, int HQ) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HQ) return;
    const int Q = x.W >> 2;
    const int h = idx / Q, wq = idx - h * Q;
    const int pc = blockIdx.y;                                  // n * C + c (wave-uniform: scalar arithmetic)
    const int n = pc / x.C, c = pc - n * x.C;
    float sc, sh;
    load_aff(x, h, c, sc, sh);
    const float post = x.post ? x.post[n * x.C + c] : 1.f;
    const float4 r = *reinterpret_cast<const float4*>(x.p + (long long)n * x.sN + (long long)c * x.sC + (long long)h * x.sH + 4 * wq);
    float4 o;
    o.x = act1(fmaf(r.x, sc, sh), x.slope) * post;
    o.y = act1(fmaf(r.y, sc, sh), x.slope) * post;
    o.z = act1(fmaf(r.z, sc, sh), x.slope) * post;
    o.w = act1(fmaf(r.w, sc, sh), x.slope) * post;
    reinterpret_cast<float4*>(out)[(long long)pc * HQ + idx] = o;
}

void launch_materialize(const Tensor& x, float* out, hipStream_t st) {
    const long long total = (long long)x.N * x.C * x.H * x.W;
    prof_note(0.0, 4.0 * (double)((x.sH == 0 && x.H > 0 ? total / x.H : total) + total));     // one read (H-broadcast: one row), one write
    const bool vec = (x.W & 3) == 0 && (x.sH & 3) == 0 && (x.sC & 3) == 0 && (x.sN & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(x.p) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    if (vec) {
        const long long t4 = total / 4;
        const long long HQ = (long long)x.H * (x.W >> 2), planes = (long long)x.N * x.C;
        static const bool planes_on = !(getenv("VR_MAT_PLANES") && atoi(getenv("VR_MAT_PLANES")) == 0);
        if (planes_on && planes <= 65535 && HQ >= 256 && HQ < (1LL << 30))
            VR_LAUNCH(materialize4p_kernel, dim3((unsigned)((HQ + 255) / 256), (unsigned)planes), dim3(256), 0, st, x, out, (int)HQ);
        else
            VR_LAUNCH(materialize4_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, st, x, out, t4);
    } else {
        VR_LAUNCH(materialize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, out, total);
    }
    VR_HIP(hipGetLastError());
}

}  // namespace vr
