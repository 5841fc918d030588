// HBM streaming rates by read : write mix (DESIGN.md hardware fact 7 says the train step's element-wise kernels all end up at 2-2.8 TB/s of
// WRITES whatever they read).  Plain grid-stride float4 kernels over 2 GiB buffers, HIP-event timed, best of 5:
//   read-only (sum into one float per block), write-only (fill), copy 1:1, 1 read : 4 writes (the upsample's mix), 2 reads : 1 write
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_rw.hip -o tools/_build/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_read(const float4* __restrict__ a, long long n, float* __restrict__ out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[blockIdx.x] = s;          // (keeps the loads alive)
}
__global__ void k_fill(float4* __restrict__ a, long long n, float v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a[i] = make_float4(v, v, v, v);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_1r4w(const float4* __restrict__ a, float4* __restrict__ b, long long n) {       // reads n, writes 4 n
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        b[i] = v; b[i + n] = v; b[i + 2 * n] = v; b[i + 3 * n] = v;
    }
}
__global__ void k_2r1w(const float4* __restrict__ a, const float4* __restrict__ c, float4* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = a[i], w = c[i];
        b[i] = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
    }
}
// one float4 per thread, no loop (the shape of bn_bwd_apply4 / materialize4)
__global__ void k_copy_flat(const float4* __restrict__ a, float4* __restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main() {
    const long long bytes = 2LL << 30, n = bytes / 16;
    float4 *a, *b, *c; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 3, bytes)); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    auto timeit = [&](const char* name, double rbytes, double wbytes, auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r && ms < best) best = ms;
        }
        printf("%-34s %7.3f ms   total %5.2f TB/s   reads %5.2f TB/s   writes %5.2f TB/s\n", name, best, (rbytes + wbytes) / best / 1e9,
               rbytes / best / 1e9, wbytes / best / 1e9);
    };
    timeit("read only (2 GiB)", bytes, 0, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
    timeit("write only (2 GiB)", 0, bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n, 1.f); });
    timeit("copy 1:1 (2 + 2 GiB), grid-stride", bytes, bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    timeit("copy 1:1, one float4 per thread", bytes, bytes, [&] { hipLaunchKernelGGL(k_copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); });
    timeit("1 read : 4 writes (0.5 + 2 GiB)", bytes / 4, bytes, [&] { hipLaunchKernelGGL(k_1r4w, dim3(grid), dim3(256), 0, 0, a, b, n / 4); });
    timeit("2 reads : 1 write (4 + 2 GiB)", 2.0 * bytes, bytes, [&] { hipLaunchKernelGGL(k_2r1w, dim3(grid), dim3(256), 0, 0, a, c, b, n); });
    timeit("hipMemsetAsync (2 GiB)", 0, bytes, [&] { hipMemsetAsync(b, 0, bytes, 0); });
    return 0;
}
