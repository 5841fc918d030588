// Probe: raw_buffer_load_lds on gfx950 -- per-lane source offsets, lane-linear LDS destination,
// out-of-range lanes must land as 0.0f in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, int n_valid, float* __restrict__ out) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 256) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_valid * 4, 0x00020000);
    // lane l of wave w reads element (w*64 + l)*3 (strided gather); offsets past n_valid*4 are OOB;
    // every 5th lane is forced OOB with a huge offset
    unsigned off = (unsigned)((wave * 64 + lane) * 3) * 4u;
    if (lane % 5 == 4) off = 0x7FFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + wave * 64), 4, off, 0, 0, 0);
    // second instruction with an SGPR offset (soffset) of 4 bytes -> element +1, into the upper half
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256 + wave * 64), 4, off, 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = lds[i];
}

int main() {
    const int N = 4096, n_valid = 600;
    float *src, *out, h[N], o[1024];
    for (int i = 0; i < N; ++i) h[i] = (float)(i + 1);
    (void)hipMalloc(&src, N * 4);
    (void)hipMalloc(&out, 1024 * 4);
    (void)hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, src, n_valid, out);
    (void)hipMemcpy(o, out, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
        const int l = i & 63;
        const int e = i * 3;
        float want = (l % 5 == 4 || e >= n_valid) ? 0.f : h[e];
        float want2 = (l % 5 == 4 || e >= n_valid) ? 0.f : (e + 1 < N ? h[e + 1] : 0.f);
        if (o[i] != want) { if (bad < 8) printf("A i=%d got %g want %g\n", i, o[i], want); ++bad; }
        if (o[256 + i] != want2) { if (bad < 8) printf("B i=%d got %g want %g (soffset included in range check?)\n", i, o[256 + i], want2); ++bad; }
    }
    printf("untouched region intact: %d\n", o[512] == -7.f && o[1023] == -7.f);
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return 0;
}
