// Probe 2: inline-asm `buffer_load_dwordx4 ... offen lds` (16 B per lane, lane-linear LDS destination,
// M0 = LDS base), out-of-range pieces must land as zeros; partially out-of-range pieces per dword.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned lds_byte, unsigned voff, i32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, int n_valid, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2048; i += 256) lds[i] = -7.f;
    __syncthreads();
    const unsigned long long base = (unsigned long long)src;
    i32x4 rsrc;
    rsrc[0] = (int)(base & 0xffffffffu);
    rsrc[1] = (int)((base >> 32) & 0xffffu);
    rsrc[2] = n_valid * 4;
    rsrc[3] = 0x00020000;
    // lane l of wave w reads 4 floats starting at element (w*64+l)*4 + 1  (NOT 16B aligned on purpose)
    unsigned off = (unsigned)((wave * 64 + lane) * 4 + 1) * 4u;
    if (lane % 5 == 4) off = 0x7FFFFFF0u;
    const unsigned lds0 = (unsigned)(size_t)(lds) + wave * 1024;     // shared pointers are 32-bit LDS offsets
    dma16(lds0, off, rsrc, 0);
    dma16(lds0 + 4096, off, rsrc, 8);       // soffset = 8 bytes -> +2 elements
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048; i += 256) out[i] = lds[i];
}

int main() {
    const int N = 8192, n_valid = 602;
    float *src, *out, h[N], o[2048];
    for (int i = 0; i < N; ++i) h[i] = (float)(i + 1);
    (void)hipMalloc(&src, N * 4);
    (void)hipMalloc(&out, 2048 * 4);
    (void)hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 2048 * 4, 0, src, n_valid, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    (void)hipMemcpy(o, out, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0, badB_in = 0, badB_soff = 0;
    for (int q = 0; q < 256; ++q) {
        const int l = q & 63;
        for (int j = 0; j < 4; ++j) {
            const int e = q * 4 + 1 + j;
            const float want = (l % 5 == 4 || e >= n_valid) ? 0.f : h[e];
            if (o[q * 4 + j] != want) { if (bad < 6) printf("A q=%d j=%d got %g want %g\n", q, j, o[q * 4 + j], want); ++bad; }
            const int e2 = e + 2;
            const float got2 = o[1024 + q * 4 + j];
            if (l % 5 == 4) { if (got2 != 0.f) ++badB_in; continue; }
            if (e2 < n_valid) { if (got2 != h[e2]) ++badB_in; }
            else if (e < n_valid) {        // in range by voffset alone, out of range with soffset added
                printf("soffset probe: e=%d e2=%d got %g (0 => soffset is range checked; %g => not)\n", e, e2, got2, h[e2]);
                ++badB_soff;
            } else if (got2 != 0.f) ++badB_in;
        }
    }
    printf("A mismatches %d, B in-range mismatches %d\n", bad, badB_in);
    return 0;
}
