// Backward-pass kernels around the MFMA dgrad / wgrad launches (train.py:91-96 of the reference:
// loss.backward(); optimizer.step()).  All HBM-bound streaming / reduction kernels.
//
// Gradient convention: for every activation tensor T the buffer G_T holds dLoss/d(value consumers
// see) = gradient w.r.t. post-BatchNorm, post-activation, post-dropout values.  The producer's
// BatchNorm backward turns G_T (in place) into dz, the gradient at the RAW conv output:
//     dy = G * post * act'(y),  y = z*scale + shift
//     dz = scale * (dy - mean(dy) - xhat * mean(dy*xhat))  =  kA*dy + kB*z + kC      (per channel)
#include <cmath>
#include <cstdlib>

#include <algorithm>
#include "conv_stage.h"
#include "kernels.h"

namespace vr {

__device__ __forceinline__ float dact(float y, float slope) { return y > 0.f ? 1.f : slope; }

// ---------------------------------------------------------------------------------------------------
// BatchNorm backward, pass 1: per-channel partial sums S1 = sum dy, S2 = sum dy*z.
// grid = (chunks of N*H rows, C).  part[(chunk*C + c)*2 + {0,1}]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
    const int c = blockIdx.y;
    const int rows = a.N * a.H;
    const int r0 = (int)((long long)blockIdx.x * rows / gridDim.x), r1 = (int)((long long)(blockIdx.x + 1) * rows / gridDim.x);
    const float sc = a.aff ? a.aff[2 * (a.aff_bcast ? 0 : c)] : 1.f, sh = a.aff ? a.aff[2 * (a.aff_bcast ? 0 : c) + 1] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const bool vec = (a.W & 3) == 0 && (a.sH & 3) == 0 && (a.sC & 3) == 0 && (a.sN & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.g)) & 15) == 0;
    if (vec) {                                   // four frames per thread, 16-byte loads
        const int Q = a.W >> 2;
        auto accum = [&](const float4& z4, const float4& g4, float pm) {
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = gg[j] * pm * dact(fmaf(zz[j], sc, sh), a.slope);
                s1 += dy;
                s2 = fmaf(dy, zz[j], s2);
            }
        };
        if (Q <= 256 && (256 % Q) == 0) {
            // a thread keeps its column quad and walks the rows in steps of 256 / Q: no per-element divisions, and the loads of
            // two row steps are issued before either is consumed (round 4: 3.9 -> HBM-speed streaming)
            const int rstep = 256 / Q, w = (threadIdx.x % Q) * 4;
            int r = r0 + threadIdx.x / Q;
            // (round 5: four row steps per iteration -- eight 16-byte loads in flight per thread; with two the pass read at 4 TB/s where
            // the flat apply pass reaches 5.6)
            for (; r + 3 * rstep < r1; r += 4 * rstep) {
                long long o[4]; float pq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rj = r + j * rstep;
                    const int nj = rj / a.H, hj = rj - nj * a.H;
                    o[j] = (long long)nj * a.sN + (long long)c * a.sC + (long long)hj * a.sH + w;
                    pq[j] = a.post ? a.post[nj * a.C + c] : 1.f;
                }
                float4 zq[4], gq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { zq[j] = *reinterpret_cast<const float4*>(a.z + o[j]); gq[j] = *reinterpret_cast<const float4*>(a.g + o[j]); }
#pragma unroll
                for (int j = 0; j < 4; ++j) accum(zq[j], gq[j], pq[j]);
            }
            for (; r + rstep < r1; r += 2 * rstep) {
                const int ra = r, rb = r + rstep;
                const int na = ra / a.H, ha = ra - na * a.H, nb = rb / a.H, hb = rb - nb * a.H;
                const long long oa = (long long)na * a.sN + (long long)c * a.sC + (long long)ha * a.sH + w;
                const long long ob = (long long)nb * a.sN + (long long)c * a.sC + (long long)hb * a.sH + w;
                const float4 za = *reinterpret_cast<const float4*>(a.z + oa), ga = *reinterpret_cast<const float4*>(a.g + oa);
                const float4 zb = *reinterpret_cast<const float4*>(a.z + ob), gb = *reinterpret_cast<const float4*>(a.g + ob);
                const float pa = a.post ? a.post[na * a.C + c] : 1.f, pb = a.post ? a.post[nb * a.C + c] : 1.f;
                accum(za, ga, pa);
                accum(zb, gb, pb);
            }
            if (r < r1) {
                const int n = r / a.H, h = r - n * a.H;
                const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
                accum(*reinterpret_cast<const float4*>(a.z + off), *reinterpret_cast<const float4*>(a.g + off), a.post ? a.post[n * a.C + c] : 1.f);
            }
        } else {
            const int total4 = (r1 - r0) * Q;
            for (int e = threadIdx.x; e < total4; e += 256) {
                const int r = r0 + e / Q, w = (e % Q) * 4;
                const int n = r / a.H, h = r % a.H;
                const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
                accum(*reinterpret_cast<const float4*>(a.z + off), *reinterpret_cast<const float4*>(a.g + off), a.post ? a.post[n * a.C + c] : 1.f);
            }
        }
    }
    const int total = vec ? 0 : (r1 - r0) * a.W;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = r0 + e / a.W, w = e % a.W;
        const int n = r / a.H, h = r % a.H;
        const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
        const float z = a.z[off];
        float g = a.g[off];
        if (a.post) g *= a.post[n * a.C + c];
        const float dy = g * dact(fmaf(z, sc, sh), a.slope);
        s1 += dy;
        s2 = fmaf(dy, z, s2);
    }
    __shared__ float red[8];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.part[((long long)blockIdx.x * a.C + c) * 2 + 0] = red[0] + red[2] + red[4] + red[6];
        a.part[((long long)blockIdx.x * a.C + c) * 2 + 1] = red[1] + red[3] + red[5] + red[7];
    }
}

// pass 1 in the flat form of the apply pass (round 6): the plane (n, c) on blockIdx.y, ONE 16-byte load of z and of g per thread, 1024
// elements per block -> one partial row per (sample, 1024-element piece) of a channel: part[((n * pieces + piece) * C + c) * 2 + {0,1}].
// (The row-looped kernel above streams at ~4.2 TB/s, the flat apply pass below at 6.2: tools/hbm_rw.hip saw the same between a
// grid-stride copy and one float4 per thread.)
__global__ __launch_bounds__(256) void bn_bwd_reduce4p_kernel(BnBwdArgs a, int HQ) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int pc = blockIdx.y;
    const int n = pc / a.C, c = pc - n * a.C;
    float s1 = 0.f, s2 = 0.f;
    if (idx < HQ) {
        const int Q = a.W >> 2;
        const int h = idx / Q, w = (idx - h * Q) * 4;
        const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
        const int ca = a.aff_bcast ? 0 : c;
        const float sc = a.aff ? a.aff[2 * ca] : 1.f, sh = a.aff ? a.aff[2 * ca + 1] : 0.f;
        const float pm = a.post ? a.post[n * a.C + c] : 1.f;
        const float4 z4 = *reinterpret_cast<const float4*>(a.z + off);
        const float4 g4 = *reinterpret_cast<const float4*>(a.g + off);
        const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dy = gg[j] * pm * dact(fmaf(zz[j], sc, sh), a.slope);
            s1 += dy;
            s2 = fmaf(dy, zz[j], s2);
        }
    }
    // wave sums by DPP (valid in lanes 31 and 63 of the two halves), the four waves in a fixed order through LDS
    s1 = half_wave_sum_dpp(s1);
    s2 = half_wave_sum_dpp(s2);
    __shared__ float red[8];
    const int wave = threadIdx.x >> 6;
    const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), 63));
    const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2), 63));
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = t1; red[wave * 2 + 1] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long row = (long long)n * gridDim.x + blockIdx.x;
        a.part[(row * a.C + c) * 2 + 0] = red[0] + red[2] + red[4] + red[6];
        a.part[(row * a.C + c) * 2 + 1] = red[1] + red[3] + red[5] + red[7];
    }
}

// pass 1b: finalize -> d(gamma), d(beta) into the gradient arena, coefficients (kA,kB,kC) for pass 2
// (Round 6, built and measured: the channel's LAST reduce block doing this -- per-channel block counter, __threadfence() + atomicAdd per
// block, 107 launches fewer per train step, results bit-equal -- made the step 55.0 -> 61.0 ms: an agent-scope release / acquire on
// gfx950 writes back and invalidates the XCD's WHOLE L2 (buffer_wbl2 / buffer_inv sc1), once per block, under the streaming reads of this
// pass and of the weight gradients on the other stream.  The counters must also be per stream: backward() runs two chains.  Reverted.)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(BnBwdArgs a, int nchunks) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) {
        s1 += (double)a.part[((long long)i * a.C + c) * 2 + 0];
        s2 += (double)a.part[((long long)i * a.C + c) * 2 + 1];
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { r1[threadIdx.x] += r1[threadIdx.x + off]; r2[threadIdx.x] += r2[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double M = (double)a.N * a.H * a.W;
        const double mean = a.save_mean[c], invstd = a.save_invstd[c];
        const double dbeta = r1[0];
        const double dgamma = invstd * (r2[0] - mean * r1[0]);
        const double scale = (double)a.gamma[c] * invstd;
        if (a.acc_grads) { a.dgamma[c] += (float)dgamma; a.dbeta[c] += (float)dbeta; }
        else { a.dgamma[c] = (float)dgamma; a.dbeta[c] = (float)dbeta; }
        a.coef[3 * c + 0] = (float)scale;
        a.coef[3 * c + 1] = (float)(-scale * invstd * dgamma / M);
        a.coef[3 * c + 2] = (float)(-scale * dbeta / M + scale * invstd * mean * dgamma / M);
    }
}

// pass 2: G <- dz in place.  coef == null: no BatchNorm (dz = dy).
// four frames per thread (16-byte loads / store); chosen by launch_bn_bwd when the rows allow it
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(BnBwdArgs a) {
    const long long total4 = (long long)a.N * a.C * a.H * (a.W >> 2);
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total4) return;
    const int Q = a.W >> 2;
    const int w = (int)(gid % Q) * 4;
    long long t = gid / Q;
    const int h = (int)(t % a.H); t /= a.H;
    const int c = (int)(t % a.C);
    const int n = (int)(t / a.C);
    const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
    const int ca = a.aff_bcast ? 0 : c;
    const float sc = a.aff ? a.aff[2 * ca] : 1.f, sh = a.aff ? a.aff[2 * ca + 1] : 0.f;
    const float pm = a.post ? a.post[n * a.C + c] : 1.f;
    float kA = 1.f, kB = 0.f, kC = 0.f;
    if (a.coef) { kA = a.coef[3 * c]; kB = a.coef[3 * c + 1]; kC = a.coef[3 * c + 2]; }
    const float4 z4 = *reinterpret_cast<const float4*>(a.z + off);
    const float4 g4 = *reinterpret_cast<const float4*>(a.g + off);
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dy = gg[j] * pm * dact(fmaf(zz[j], sc, sh), a.slope);
        o[j] = a.coef ? fmaf(kA, dy, fmaf(kB, zz[j], kC)) : dy;
    }
    *reinterpret_cast<float4*>(a.g + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// pass 2 with the plane index (n, c) on blockIdx.y (round 5): one division per thread instead of three by run-time values
__global__ __launch_bounds__(256) void bn_bwd_apply4p_kernel(BnBwdArgs a, int HQ) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HQ) return;
    const int Q = a.W >> 2;
    const int h = idx / Q, w = (idx - h * Q) * 4;
    const int pc = blockIdx.y;
    const int n = pc / a.C, c = pc - n * a.C;
    const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
    const int ca = a.aff_bcast ? 0 : c;
    const float sc = a.aff ? a.aff[2 * ca] : 1.f, sh = a.aff ? a.aff[2 * ca + 1] : 0.f;
    const float pm = a.post ? a.post[n * a.C + c] : 1.f;
    float kA = 1.f, kB = 0.f, kC = 0.f;
    if (a.coef) { kA = a.coef[3 * c]; kB = a.coef[3 * c + 1]; kC = a.coef[3 * c + 2]; }
    const float4 z4 = *reinterpret_cast<const float4*>(a.z + off);
    const float4 g4 = *reinterpret_cast<const float4*>(a.g + off);
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dy = gg[j] * pm * dact(fmaf(zz[j], sc, sh), a.slope);
        o[j] = a.coef ? fmaf(kA, dy, fmaf(kB, zz[j], kC)) : dy;
    }
    *reinterpret_cast<float4*>(a.g + off) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
    const long long total = (long long)a.N * a.C * a.H * a.W;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % a.W);
    long long t = gid / a.W;
    const int h = (int)(t % a.H); t /= a.H;
    const int c = (int)(t % a.C);
    const int n = (int)(t / a.C);
    const long long off = (long long)n * a.sN + (long long)c * a.sC + (long long)h * a.sH + w;
    const int ca = a.aff_bcast ? 0 : c;
    const float sc = a.aff ? a.aff[2 * ca] : 1.f, sh = a.aff ? a.aff[2 * ca + 1] : 0.f;
    const float z = a.z[off];
    float g = a.g[off];
    if (a.post) g *= a.post[n * a.C + c];
    const float dy = g * dact(fmaf(z, sc, sh), a.slope);
    a.g[off] = a.coef ? fmaf(a.coef[3 * c], dy, fmaf(a.coef[3 * c + 1], z, a.coef[3 * c + 2])) : dy;
}

static int bn_bwd_chunks_rows(const BnBwdArgs& a) {
    const long long per_c = (long long)a.N * a.H * a.W;
    long long ch = per_c / 16384;        // (round 5: 4096 / 8192 / 32768 elements per block measured the same: 2.64 - 2.76 ms per step)
    if (ch < 1) ch = 1;
    if (ch > 256) ch = 256;
    if (ch > (long long)a.N * a.H) ch = (long long)a.N * a.H;
    return (int)ch;
}
// the flat reduce (bn_bwd_reduce4p_kernel) by SHAPE: rows of whole 16-byte quads, planes on blockIdx.y, at least one full block per plane
static bool bn_bwd_flat_shape(const BnBwdArgs& a) {
    static const bool on = !(getenv("VR_BN_REDUCE_FLAT") && atoi(getenv("VR_BN_REDUCE_FLAT")) == 0);
    const long long HQ = (long long)a.H * (a.W >> 2), planes = (long long)a.N * a.C;
    return on && (a.W & 3) == 0 && (a.sH & 3) == 0 && (a.sC & 3) == 0 && (a.sN & 3) == 0 && planes <= 65535 && HQ >= 256 && HQ < (1LL << 30);
}
// partial rows per channel the scratch must hold (the kernel choice is re-made at launch time from the real pointers: the larger of the two)
int bn_bwd_chunks(const BnBwdArgs& a) {
    const int rows = bn_bwd_chunks_rows(a);
    if (!bn_bwd_flat_shape(a)) return rows;
    const long long flat = (long long)a.N * (((long long)a.H * (a.W >> 2) + 255) / 256);
    return (int)std::max<long long>(rows, flat);
}

void launch_bn_bwd(const BnBwdArgs& a, hipStream_t st) {
    const double elems = (double)a.N * a.C * a.H * a.W;
    if (a.coef) {
        int nch = bn_bwd_chunks_rows(a);
        prof_note(0.0, 8.0 * elems);                         // reads z and g
        const bool aligned = ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.g)) & 15) == 0;
        if (bn_bwd_flat_shape(a) && aligned) {
            const long long HQ = (long long)a.H * (a.W >> 2);
            const unsigned pieces = (unsigned)((HQ + 255) / 256);
            nch = (int)(a.N * (long long)pieces);
            VR_LAUNCH(bn_bwd_reduce4p_kernel, dim3(pieces, (unsigned)((long long)a.N * a.C)), dim3(256), 0, st, a, (int)HQ);
        } else {
            VR_LAUNCH(bn_bwd_reduce_kernel, dim3(nch, a.C), dim3(256), 0, st, a);
        }
        VR_HIP(hipGetLastError());
        VR_LAUNCH(bn_bwd_finalize_kernel, dim3(a.C), dim3(256), 0, st, a, nch);
        VR_HIP(hipGetLastError());
    }
    const long long total = (long long)a.N * a.C * a.H * a.W;
    const bool vec = (a.W & 3) == 0 && (a.sH & 3) == 0 && (a.sC & 3) == 0 && (a.sN & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.g)) & 15) == 0;
    prof_note(0.0, 12.0 * elems);                            // reads z and g, writes g
    const long long HQ = (long long)a.H * (a.W >> 2), planes = (long long)a.N * a.C;
    static const bool planes_on = !(getenv("VR_MAT_PLANES") && atoi(getenv("VR_MAT_PLANES")) == 0);
    if (vec && planes_on && planes <= 65535 && HQ >= 256 && HQ < (1LL << 30))
        VR_LAUNCH(bn_bwd_apply4p_kernel, dim3((unsigned)((HQ + 255) / 256), (unsigned)planes), dim3(256), 0, st, a, (int)HQ);
    else if (vec) VR_LAUNCH(bn_bwd_apply4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, a);
    else VR_LAUNCH(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    VR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// Transposed bilinear x2 (align_corners=True): G_lo += U^T D_hi, gather form (no atomics).
// ---------------------------------------------------------------------------------------------------
__global__ void upsample_bwd_kernel(const float* __restrict__ dhi, int N, int C, int H, int W, float rh, float rw,
                                    float* __restrict__ glo, long long gN, long long gC, long long gH, int accumulate) {
    const long long total = (long long)N * C * H * W;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int j = (int)(gid % W);
    long long t = gid / W;
    const int i = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const int H2 = 2 * H, W2 = 2 * W;
    const float* src = dhi + ((long long)n * C + c) * H2 * W2;
    // rows/columns that can touch low-res index i: floor(r*h) in {i-1, i} with r = (H-1)/(2H-1) < 1/2  =>  h in [2i-2, 2i+2];
    // the five row weights and five column weights are computed once, then a 5x5 weighted sum
    float wh[5], ww[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const int h = 2 * i - 2 + d;
        float v = 0.f;
        if (h >= 0 && h < H2) {
            const float h1r = rh * (float)h;
            const int h1 = (int)h1r;
            const int h1p = (h1 < H - 1) ? 1 : 0;
            const float l1 = h1r - (float)h1;
            if (h1 == i) v += 1.f - l1;
            if (h1 + h1p == i) v += l1;
        }
        wh[d] = v;
        const int w = 2 * j - 2 + d;
        float u = 0.f;
        if (w >= 0 && w < W2) {
            const float w1r = rw * (float)w;
            const int w1 = (int)w1r;
            const int w1p = (w1 < W - 1) ? 1 : 0;
            const float m1 = w1r - (float)w1;
            if (w1 == j) u += 1.f - m1;
            if (w1 + w1p == j) u += m1;
        }
        ww[d] = u;
    }
    float acc = 0.f;
#pragma unroll
    for (int dh = 0; dh < 5; ++dh) {
        if (wh[dh] == 0.f) continue;
        const int h = 2 * i - 2 + dh;
        const float* row = src + (long long)h * W2 + 2 * j - 2;
        float r = 0.f;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw)
            if (ww[dw] != 0.f) r = fmaf(ww[dw], row[dw], r);
        acc = fmaf(wh[dh], r, acc);
    }
    float* const gq = glo + (long long)n * gN + (long long)c * gC + (long long)i * gH + j;
    *gq = accumulate ? *gq + acc : acc;
}

// LDS-tiled form for the large tensors: one workgroup = an 8 x 32 tile of low-resolution outputs of one (n, c) plane;
// the (2*8+3) x (2*32+3) patch of the high-resolution gradient that can touch it is staged once (coalesced rows), the
// 5 x 5 separable gather then runs out of LDS -- every dhi element is fetched from HBM once per tile instead of up to
// 25 times through the caches.
// VEC (round 5): the patch is staged with ALIGNED 16-byte loads -- columns [2 j0 - 4, 2 j0 + 2 TJ + 4), 18 quads per row, 342 loads per
// tile instead of 1273 four-byte ones on rows that start two floats off a 16-byte boundary (needs 2 W % 4 == 0 and an aligned dhi).
template <bool VEC>
__global__ __launch_bounds__(256) void upsample_bwd_tiled_kernel(const float* __restrict__ dhi, int C, int H, int W, float rh,
                                                                 float rw, int tiles_w, int tiles_per_plane,
                                                                 float* __restrict__ glo, long long gN, long long gC,
                                                                 long long gH, int accumulate) {
    constexpr int TI = 8, TJ = 32, PH = 2 * TI + 3, PW = 2 * TJ + 3, PP = VEC ? 2 * TJ + 12 : PW + 1, XO = VEC ? 2 : 0;
    __shared__ __attribute__((aligned(16))) float patch[PH * PP];
    const int plane = blockIdx.x / tiles_per_plane;          // n * C + c
    const int trem = blockIdx.x - plane * tiles_per_plane;
    const int i0 = (trem / tiles_w) * TI, j0 = (trem % tiles_w) * TJ;
    const int H2 = 2 * H, W2 = 2 * W;
    const float* src = dhi + (long long)plane * H2 * W2;
    const int hb = 2 * i0 - 2, wb = 2 * j0 - 2;
    if constexpr (VEC) {
        constexpr int QR = (2 * TJ + 8) / 4, NV = (PH * QR + 255) / 256;       // 18 quads per row: patch column 4 q <-> w = wb - 2 + 4 q
        float4 pv4[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int ec = e < PH * QR ? e : PH * QR - 1;
            const int r = ec / QR, q = ec - r * QR;
            int h = hb + r, w = wb - 2 + 4 * q;
            h = h < 0 ? 0 : (h >= H2 ? H2 - 1 : h);
            w = w < 0 ? 0 : (w > W2 - 4 ? W2 - 4 : w);
            pv4[k] = *reinterpret_cast<const float4*>(src + (long long)h * W2 + w);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int r = e / QR, q = e - r * QR;
            const int h = hb + r, w = wb - 2 + 4 * q;
            if (e < PH * QR)
                *reinterpret_cast<float4*>(patch + r * PP + 4 * q) = (h >= 0 && h < H2 && w >= 0 && w < W2) ? pv4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
    // all loads of the thread issued before the first use: clamped addresses, masked afterwards (a load behind its own guard gets
    // a vmcnt(0) from hipcc -- five serial memory round trips per thread)
    constexpr int NL = (PH * PW + 255) / 256;
    float pv[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int ec = e < PH * PW ? e : PH * PW - 1;
        const int r = ec / PW, q = ec - r * PW;
        int h = hb + r, w = wb + q;
        h = h < 0 ? 0 : (h >= H2 ? H2 - 1 : h);
        w = w < 0 ? 0 : (w >= W2 ? W2 - 1 : w);
        pv[k] = src[(long long)h * W2 + w];
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int r = e / PW, q = e - r * PW;
        const int h = hb + r, w = wb + q;
        if (e < PH * PW) patch[r * PP + q] = (h >= 0 && h < H2 && w >= 0 && w < W2) ? pv[k] : 0.f;
    }
    }
    // the 5 row weights of the tile's 8 rows and the 5 column weights of its 32 columns: 200 values computed once per workgroup and
    // shared through LDS (round 5: every thread used to derive its own ten -- ~80 VALU instructions per output element)
    __shared__ float whs[TI * 5], wws[TJ * 5];
    if (threadIdx.x < TI * 5 + TJ * 5) {
        const bool isrow = threadIdx.x < TI * 5;
        const int e = isrow ? threadIdx.x : threadIdx.x - TI * 5;
        const int t = e / 5, d = e - t * 5;
        const int o = (isrow ? i0 : j0) + t;                    // low-resolution row / column
        const int hh = 2 * o - 2 + d;                          // high-resolution row / column that may contribute
        const int L = isrow ? H : W, L2 = 2 * L;
        const float rr = isrow ? rh : rw;
        float v = 0.f;
        if (hh >= 0 && hh < L2) {
            const float h1r = rr * (float)hh;
            const int h1 = (int)h1r;
            const int h1p = (h1 < L - 1) ? 1 : 0;
            const float l1 = h1r - (float)h1;
            if (h1 == o) v += 1.f - l1;
            if (h1 + h1p == o) v += l1;
        }
        (isrow ? whs : wws)[e] = v;
    }
    __syncthreads();
    const int ti = threadIdx.x >> 5, tj = threadIdx.x & 31;
    const int i = i0 + ti, j = j0 + tj;
    if (i >= H || j >= W) return;
    float wh[5], ww[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) { wh[d] = whs[ti * 5 + d]; ww[d] = wws[tj * 5 + d]; }
    const float* pr = patch + (2 * ti) * PP + 2 * tj + XO;
    float acc = 0.f;
#pragma unroll
    for (int dh = 0; dh < 5; ++dh) {
        float r = 0.f;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) r = fmaf(ww[dw], pr[dh * PP + dw], r);
        acc = fmaf(wh[dh], r, acc);
    }
    const int n = plane / C, c = plane - n * C;
    float* const gq = glo + (long long)n * gN + (long long)c * gC + (long long)i * gH + j;
    *gq = accumulate ? *gq + acc : acc;
}

void launch_upsample_bwd(const float* dhi, int N, int C, int H, int W, float* glo, long long gN, long long gC,
                         long long gH, int accumulate, hipStream_t st) {
    const long long total = (long long)N * C * H * W;
    const float rh = (float)(H - 1) / (float)(2 * H - 1), rw = (float)(W - 1) / (float)(2 * W - 1);
    static const bool tiled = !getenv("VR_NO_UPBWD_TILED");
    prof_note(0.0, 4.0 * (accumulate ? 6.0 : 5.0) * (double)total);   // reads the 4x larger gradient, writes (accumulating: read-modify-writes) the low-resolution one
    if (tiled && W >= 16 && H >= 4) {
        const int tiles_w = (W + 31) / 32, tiles_h = (H + 7) / 8;
        const long long blocks = (long long)N * C * tiles_h * tiles_w;
        if (blocks < 0x7FFFFFFFLL) {
            static const bool vec_on = !(getenv("VR_UPBWD_VEC") && atoi(getenv("VR_UPBWD_VEC")) == 0);
            if (vec_on && (W & 1) == 0 && W >= 2 && (reinterpret_cast<uintptr_t>(dhi) & 15) == 0)
                VR_LAUNCH(upsample_bwd_tiled_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, dhi, C, H, W, rh, rw, tiles_w,
                                   tiles_h * tiles_w, glo, gN, gC, gH, accumulate);
            else
                VR_LAUNCH(upsample_bwd_tiled_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, dhi, C, H, W, rh, rw, tiles_w,
                                   tiles_h * tiles_w, glo, gN, gC, gH, accumulate);
            VR_HIP(hipGetLastError());
            return;
        }
    }
    VR_LAUNCH(upsample_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dhi, N, C, H, W, rh, rw,
                       glo, gN, gC, gH, accumulate);
    VR_HIP(hipGetLastError());
}

// out[n][c][w] += sum_h d[n][c][h][w]     (backward of the broadcast along H, lib/layers.py:94)
__global__ void sum_h_kernel(const float* __restrict__ d, int N, int C, int H, int W, float* __restrict__ out) {
    const int total = N * C * W;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = gid % W;
    const int nc = gid / W;
    const float* p = d + (long long)nc * H * W + w;
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += p[(long long)h * W];
    out[gid] += s;
}
void launch_sum_h(const float* d, int N, int C, int H, int W, float* out, hipStream_t st) {
    const int total = N * C * W;
    VR_LAUNCH(sum_h_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d, N, C, H, W, out);
    VR_HIP(hipGetLastError());
}

// g[n][c][h][w] += gp[n][c][w] / H     (backward of AdaptiveAvgPool2d((1,None)), lib/layers.py:72)
__global__ void avgpool_bwd_kernel(const float* __restrict__ gp, float* __restrict__ g, int N, int C, int H, int W,
                                   long long sN, long long sC, long long sH, int accumulate) {
    const long long total = (long long)N * C * H * W;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % W);
    long long t = gid / W;
    const int h = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    float* const q = g + (long long)n * sN + (long long)c * sC + (long long)h * sH + w;
    const float v = gp[((long long)n * C + c) * W + w] / (float)H;
    *q = accumulate ? *q + v : v;
}
void launch_avgpool_bwd(const float* gp, float* g, int N, int C, int H, int W, long long sN, long long sC, long long sH,
                        int accumulate, hipStream_t st) {
    const long long total = (long long)N * C * H * W;
    VR_LAUNCH(avgpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gp, g, N, C, H, W, sN,
                       sC, sH, accumulate);
    VR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// Thin convs (Cout = CO in {1,2}) backward.
//   dgrad: G_x[n][c][h][w] (+)= post-chain is the producer's job; here  sum_o W[o][c] * dz[n][o][h][w]
//   wgrad: dW[o][c] = sum_{n,h,w} dz[n][o][h][w] * act(x)[n][c][h][w]   (8 channels per blockIdx.y)
// dz is dense [N][CO][H][W].
// ---------------------------------------------------------------------------------------------------

// float4 form (round 4): one thread = four consecutive columns x 8 channels; dz is fetched once per eight channels instead of once per
// element, the three 64-bit divisions per element are gone (the scalar form above ran at 1.2-2.4 TB/s: 445 us for the head's 2 -> 32
// data gradient at batch 16); rows must be 16-byte aligned.
template <int CO>
__global__ __launch_bounds__(256) void thin_dgrad4_kernel(Tensor x, const float* __restrict__ w, const float* __restrict__ dz,
                                                          float* __restrict__ g, int accumulate) {
    const int W4 = x.W >> 2;
    const int total4 = x.N * x.H * W4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const int c0 = blockIdx.y * 8;
    const int w4 = e % W4;
    const int t = e / W4;
    const int h = t % x.H;
    const int n = t / x.H;
    float4 d[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) d[o] = *reinterpret_cast<const float4*>(dz + (((long long)n * CO + o) * x.H + h) * x.W + 4 * w4);
    float* gb = g + (long long)n * x.sN + (long long)h * x.sH + 4 * w4;
    float4 old[8];
    if (accumulate) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k < x.C ? c0 + k : x.C - 1;
            old[k] = *reinterpret_cast<const float4*>(gb + (long long)c * x.sC);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        if (c < x.C) {
            float4 v = accumulate ? old[k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float wv = w[o * x.C + c];
                v.x = fmaf(wv, d[o].x, v.x); v.y = fmaf(wv, d[o].y, v.y); v.z = fmaf(wv, d[o].z, v.z); v.w = fmaf(wv, d[o].w, v.w);
            }
            *reinterpret_cast<float4*>(gb + (long long)c * x.sC) = v;
        }
    }
}


// The same with four consecutive frames per thread (16-byte loads of x and dz) and 32-bit index arithmetic: the scalar form above spends
// its time on two 64-bit divisions and nine 4-byte loads per element (0.9 TB/s measured); rows must be 16-byte aligned.
template <int CO>
__global__ __launch_bounds__(256) void thin_wgrad4_kernel(Tensor x, const float* __restrict__ dz, float* __restrict__ part) {
    const int c0 = blockIdx.y * 8;
    float acc[CO][8];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[o][k] = 0.f;
    const int W4 = x.W >> 2;
    const int total4 = x.N * x.H * W4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total4; e += gridDim.x * 256) {
        const int w4 = e % W4;
        const int t = e / W4;
        const int h = t % x.H;
        const int n = t / x.H;
        float4 d[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) d[o] = *reinterpret_cast<const float4*>(dz + (((long long)n * CO + o) * x.H + h) * x.W + 4 * w4);
        const float* aff = (h < x.hsplit) ? x.aff0 : x.aff1;
        const float* xb = x.p + (long long)n * x.sN + (long long)h * x.sH + 4 * w4;
        float4 xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k < x.C ? c0 + k : x.C - 1;
            xv[k] = *reinterpret_cast<const float4*>(xb + (long long)c * x.sC);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            if (c < x.C) {
                float sc = 1.f, sh = 0.f;
                if (aff) { sc = aff[2 * c]; sh = aff[2 * c + 1]; }
                const float post = x.post ? x.post[n * x.C + c] : 1.f;
                float v[4] = {fmaf(xv[k].x, sc, sh), fmaf(xv[k].y, sc, sh), fmaf(xv[k].z, sc, sh), fmaf(xv[k].w, sc, sh)};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (v[j] > 0.f ? v[j] : v[j] * x.slope) * post;
#pragma unroll
                for (int o = 0; o < CO; ++o)
                    acc[o][k] = fmaf(d[o].x, v[0], fmaf(d[o].y, v[1], fmaf(d[o].z, v[2], fmaf(d[o].w, v[3], acc[o][k]))));
            }
        }
    }
    __shared__ float red[4][CO * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = acc[o][k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wave][o * 8 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < CO * 8) {
        const int o = threadIdx.x / 8, k = threadIdx.x % 8;
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (c0 + k < x.C) part[(long long)blockIdx.x * CO * x.C + o * x.C + c0 + k] = v;
    }
}

// out[i] (+)= scale * sum_p part[p*stride + i]   (double accumulation, fixed order: deterministic).
// One workgroup per output element: 256 threads stride over the P partials, tree-reduce in LDS.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, long long stride, int P,
                                                          float* __restrict__ out, long long n, int accumulate, float scale) {
    __shared__ double red[256];
    const long long i = blockIdx.x;
    double s = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) s += (double)part[(long long)p * stride + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)red[0] * scale;
        out[i] = accumulate ? out[i] + v : v;
    }
}
void launch_reduce_rows(const float* part, long long stride, int P, float* out, long long n, int accumulate, float scale,
                        hipStream_t st) {
    VR_LAUNCH(reduce_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, part, stride, P, out, n, accumulate, scale);
    VR_HIP(hipGetLastError());
}

void launch_thin_dgrad(const Tensor& x, int CO, const float* w, const float* dz, float* g, int accumulate, hipStream_t st) {
    const long long total = (long long)x.N * x.C * x.H * x.W;
    // reads x (activation derivative), dz; writes (accumulating: read-modify-writes) g
    prof_note(2.0 * CO * (double)total, 4.0 * ((accumulate ? 3.0 : 2.0) * (double)total + (double)CO * x.N * x.H * x.W));
    const bool vec = (x.W & 3) == 0 && (x.sH & 3) == 0 && (x.sC & 3) == 0 && (x.sN & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0 &&
                     (long long)x.N * x.H * x.W < 0x7FFFFFFFLL;
    if (vec) {
        const dim3 g4((unsigned)(((long long)x.N * x.H * (x.W >> 2) + 255) / 256), (unsigned)((x.C + 7) / 8));
        if (CO == 1) VR_LAUNCH((thin_dgrad4_kernel<1>), g4, dim3(256), 0, st, x, w, dz, g, accumulate);
        else VR_LAUNCH((thin_dgrad4_kernel<2>), g4, dim3(256), 0, st, x, w, dz, g, accumulate);
    } else {
        // (round 6: the scalar per-element form is gone -- frames % 16 == 0 (lib/nets.py:129 / spec_utils.crop_center) makes every width
        // the model can produce a multiple of 4, and nothing ever reached it)
        throw Error(-2, "thin conv data gradient: widths and strides must be multiples of 4 floats");
    }
    VR_HIP(hipGetLastError());
}

int thin_wgrad_blocks(const Tensor& x) {
    const long long total = (long long)x.N * x.H * x.W;
    long long b = (total + 256 * 32 - 1) / (256 * 32);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}
void launch_thin_wgrad(const Tensor& x, int CO, const float* dz, float* part, float* dw, int accumulate, hipStream_t st) {
    const int nb = thin_wgrad_blocks(x);
    const dim3 grid(nb, (x.C + 7) / 8);
    prof_note(2.0 * CO * (double)x.N * x.C * x.H * x.W, 4.0 * ((double)x.N * x.C * x.H * x.W + (double)CO * x.N * x.H * x.W));
    const bool vec = (x.W & 3) == 0 && (x.sH & 3) == 0 && (x.sC & 3) == 0 && (x.sN & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x.p) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0 &&
                     (long long)x.N * x.H * x.W < 0x7FFFFFFFLL;
    if (vec) {
        if (CO == 1) VR_LAUNCH((thin_wgrad4_kernel<1>), grid, dim3(256), 0, st, x, dz, part);
        else VR_LAUNCH((thin_wgrad4_kernel<2>), grid, dim3(256), 0, st, x, dz, part);
    } else {
        throw Error(-2, "thin conv weight gradient: widths and strides must be multiples of 4 floats");
    }
    VR_HIP(hipGetLastError());
    launch_reduce_rows(part, (long long)CO * x.C, nb, dw, (long long)CO * x.C, accumulate, 1.f, st);
}

// ---------------------------------------------------------------------------------------------------
// Head + loss (train.py:81,89): mask = sigmoid(out(f3)) (replicate-padded row), loss = mean|mask*X - y|.
// Writes dlogit [N][2][H][W] = dLoss/d(pre-sigmoid) * gscale and per-block |.| sums.
// X, y: [N][2][bins][T] dense; bins = H + pad_rows.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_loss_kernel(Tensor x, const float* __restrict__ w, const float* __restrict__ X,
                                                        const float* __restrict__ Y, int bins, float gscale,
                                                        float* __restrict__ dlogit, float* __restrict__ mask_out,
                                                        float* __restrict__ loss_part) {
    const long long total = (long long)x.N * x.H * x.W;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f;
    if (gid < total) {
        const int wq = (int)(gid % x.W);
        const long long t = gid / x.W;
        const int h = (int)(t % x.H);
        const int n = (int)(t / x.H);
        float a0 = 0.f, a1 = 0.f;
        const float* aff = (h < x.hsplit) ? x.aff0 : x.aff1;
        for (int c = 0; c < x.C; ++c) {
            float sc = 1.f, sh = 0.f;
            if (aff) { sc = aff[2 * c]; sh = aff[2 * c + 1]; }
            float v = fmaf(x.p[(long long)n * x.sN + (long long)c * x.sC + (long long)h * x.sH + wq], sc, sh);
            v = v > 0.f ? v : v * x.slope;
            a0 = fmaf(w[c], v, a0);
            a1 = fmaf(w[x.C + c], v, a1);
        }
        const float lg[2] = {a0, a1};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float m = 1.f / (1.f + expf(-lg[o]));
            float dm = 0.f;
            const int reps = (h == x.H - 1) ? bins - x.H + 1 : 1;      // replicate-pad rows share the last logit
            for (int e = 0; e < reps; ++e) {
                const long long off = (((long long)n * 2 + o) * bins + h + e) * x.W + wq;
                const float xv = X[off];
                const float diff = m * xv - Y[off];
                lsum += fabsf(diff);
                dm += (diff > 0.f ? xv : (diff < 0.f ? -xv : 0.f));
                if (mask_out) mask_out[off] = m;
            }
            dlogit[(((long long)n * 2 + o) * x.H + h) * x.W + wq] = dm * gscale * m * (1.f - m);
        }
    }
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Backward of the mask head alone (the autograd split of train.py:81,92): dmask [N][2][bins][W] = dLoss/d(mask), mask
// [N][2][bins][W] = the forward's sigmoid output  ->  dlogit [N][2][H][W] = m (1 - m) * (sum over the replicate-padded
// rows that share the logit of row H-1).
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ mask, int N,
                                                       int H, int W, int bins, float* __restrict__ dlogit) {
    const long long total = (long long)N * 2 * H * W;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % W);
    long long t = gid / W;
    const int h = (int)(t % H);
    const long long no = t / H;                  // n * 2 + o
    const long long off = (no * bins + h) * W + w;
    const float m = mask[off];
    float d = dmask[off];
    if (h == H - 1)
        for (int e = 1; e <= bins - H; ++e) d += dmask[off + (long long)e * W];
    dlogit[gid] = d * m * (1.f - m);
}
void launch_head_bwd(const float* dmask, const float* mask, int N, int H, int W, int bins, float* dlogit, hipStream_t st) {
    const long long total = (long long)N * 2 * H * W;
    VR_LAUNCH(head_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dmask, mask, N, H, W, bins, dlogit);
    VR_HIP(hipGetLastError());
}

int head_loss_blocks(const Tensor& x) { return (int)(((long long)x.N * x.H * x.W + 255) / 256); }

void launch_head_loss(const Tensor& x, const float* w, const float* X, const float* Y, int bins, float gscale,
                      float* dlogit, float* mask_out, float* loss_part, float* loss_out, float loss_scale, hipStream_t st) {
    const int nb = head_loss_blocks(x);
    // reads the C-channel feature map, X, Y; writes the 2-channel logit gradient (+ the mask when asked)
    prof_note(4.0 * x.C * (double)x.N * x.H * x.W, 4.0 * ((double)x.N * x.C * x.H * x.W + (mask_out ? 8.0 : 6.0) * x.N * x.H * x.W));
    VR_LAUNCH(head_loss_kernel, dim3(nb), dim3(256), 0, st, x, w, X, Y, bins, gscale, dlogit, mask_out, loss_part);
    VR_HIP(hipGetLastError());
    launch_reduce_rows(loss_part, 1, nb, loss_out, 1, 0, loss_scale, st);
}

// ---------------------------------------------------------------------------------------------------
// Weights for the data-gradient: WT[co][KK-1-tap][ci] (CinPad) <- W[ci][tap][co] (CoutPad)
// ---------------------------------------------------------------------------------------------------
__global__ void flip_transpose_kernel(const FlipDesc* descs) {
    const FlipDesc d = descs[blockIdx.y];
    const int total = d.Cin * d.KK * d.Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % d.Cout;
        const int t = i / d.Cout;
        const int tap = t % d.KK, ci = t / d.KK;
        d.wt[((long long)co * d.KK + (d.KK - 1 - tap)) * d.CinPad + ci] = d.w[((long long)ci * d.KK + tap) * d.CoutPad + co];
    }
}
// Parity-class weights of a stride-2 3x3 conv's data gradient (see ConvArgs::tapmask): for output parity
// (ph, pw) the gradient is a stride-1 conv over dz whose tap t = (th, tw) (input offset th-1, tw-1) carries the
// forward weight w[.][.][kh][kw]:   parity 0: th = 1 <- k = 1;   parity 1: th = 1 <- k = 2, th = 2 <- k = 0.
//   w  [Cin][9][CoutPad] (forward layout)   ->   wc [4][Cout][9][CinPad]  (dz channel, tap, input channel)
// (one launch for all stride-2 layers of the net: blockIdx.y = layer -- 20 launches of ~5 us per train step until round 6)
__global__ void s2_class_weights_kernel(const S2WDesc* __restrict__ descs) {
    const S2WDesc d = descs[blockIdx.y];
    const float* __restrict__ w = d.w;
    const int Cin = d.Cin, Cout = d.Cout, CoutPad = d.CoutPad, CinPad = d.CinPad;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)Cout * 9 * CinPad;
    if (gid >= 4 * per) return;
    const int cls = (int)(gid / per);
    long long r = gid - cls * per;
    const int ci = (int)(r % CinPad); r /= CinPad;
    const int t = (int)(r % 9);
    const int co = (int)(r / 9);
    const int ph = cls >> 1, pw = cls & 1, th = t / 3, tw = t % 3;
    const int kh = ph == 0 ? (th == 1 ? 1 : -1) : (th == 1 ? 2 : (th == 2 ? 0 : -1));
    const int kw = pw == 0 ? (tw == 1 ? 1 : -1) : (tw == 1 ? 2 : (tw == 2 ? 0 : -1));
    float v = 0.f;
    if (kh >= 0 && kw >= 0 && ci < Cin) v = w[((long long)ci * 9 + kh * 3 + kw) * CoutPad + co];
    d.wc[gid] = v;
}

void launch_s2_class_weights(const S2WDesc* d_descs, int n, long long max_elems, hipStream_t st) {
    if (n <= 0) return;
    VR_LAUNCH(s2_class_weights_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)n), dim3(256), 0, st, d_descs);
    VR_HIP(hipGetLastError());
}

void launch_flip_transpose(const FlipDesc* d_descs, int n, hipStream_t st) {
    VR_LAUNCH(flip_transpose_kernel, dim3(64, n), dim3(256), 0, st, d_descs);
    VR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// torch.optim.Adam defaults (train.py:215-218) over the flat parameter arena, one launch.
// ---------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float b1, float b2, float omb1, float omb2, float step_size, float bc2_sqrt, float eps,
                            float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + omb1 * gi;              // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + omb2 * gi * gi;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);                    // param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
}
void launch_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2, double eps,
                 long long step, double gscale, hipStream_t st) {
    // scalars in double, as torch's python floats are (torch/optim/adam.py _single_tensor_adam)
    const double bc1 = 1.0 - std::pow(b1, (double)step);
    const double bc2s = std::sqrt(1.0 - std::pow(b2, (double)step));
    prof_note(0.0, 28.0 * (double)n);                        // p, g, m, v read; p, m, v written
    VR_LAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, (float)b1, (float)b2,
                       (float)(1.0 - b1), (float)(1.0 - b2), (float)(lr / bc1), (float)bc2s, (float)eps, (float)gscale);
    VR_HIP(hipGetLastError());
}

// bf16 wire format of the gradient bucket (round-to-nearest-even; NaN stays NaN)
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned u = __float_as_uint(x[i]);
    const unsigned r = ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) ? (u | 0x00400000u) : u + 0x7FFFu + ((u >> 16) & 1u);
    y[i] = (unsigned short)(r >> 16);
}
__global__ void bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = __uint_as_float((unsigned)x[i] << 16);
}
void launch_f32_to_bf16(const float* x, unsigned short* y, long long n, hipStream_t st) {
    VR_LAUNCH(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    VR_HIP(hipGetLastError());
}
void launch_bf16_to_f32(const unsigned short* x, float* y, long long n, hipStream_t st) {
    VR_LAUNCH(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    VR_HIP(hipGetLastError());
}

// sum over (n, w) per channel of d [N][C][W]  ->  out[c]      (bias gradients)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ d, int N, int C, int W,
                                                          float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int e = threadIdx.x; e < N * W; e += 256) s += d[((long long)(e / W) * C + c) * W + e % W];
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = red[0] + red[1] + red[2] + red[3];
        out[c] = accumulate ? out[c] + v : v;
    }
}
void launch_channel_sum(const float* d, int N, int C, int W, float* out, int accumulate, hipStream_t st) {
    VR_LAUNCH(channel_sum_kernel, dim3(C), dim3(256), 0, st, d, N, C, W, out, accumulate);
    VR_HIP(hipGetLastError());
}

}  // namespace vr
