// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 with register operands,
// zeros vs random data (power/clock sensitivity of the matrix pipe).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[256 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a = -a;                       // keep the sums bounded, operands toggling
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 matrix pipe: v_mfma_f32_32x32x16_bf16 (8x the K of the fp32 instruction per issue)
__global__ __launch_bounds__(256) void mfma_bf16_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(threadIdx.x + j) & 255]; b[j] = (__bf16)in[256 + ((threadIdx.x + 3 * j) & 255)]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        a[0] = -a[0];
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    float *in, *out;
    hipMalloc(&in, 512 * 4);
    hipMalloc(&out, blocks * 256 * 4);
    float h[512];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 512; ++i) h[i] = mode ? (float)rand() / RAND_MAX - 0.5f : 0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wpb = 0; wpb < 1; ++wpb) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int threads = wpb ? 512 : 256;    // 1 or 2 MFMA waves per SIMD
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)blocks * (threads / 64) * iters * 4 * 4096.0;
                if (rep == 2)
                    printf("data=%s waves/block=%d: %.2f ms  %.1f TFLOP/s\n", mode ? "random" : "zeros", threads / 64, ms,
                           flops / ms * 1e-9);
            }
        }
    }
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_bf16_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 16);
        printf("bf16 32x32x16 (random data): %.2f ms  %.1f TFLOP/s  -> fp32 emulated by 6 / 3 bf16 MFMAs: %.0f / %.0f TFLOP/s\n", ms,
               flops / ms * 1e-9, flops / ms * 1e-9 / 6, flops / ms * 1e-9 / 3);
    }
    return 0;
}
