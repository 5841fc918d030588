#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned x = 100 + lane, y = 200 + lane;
    u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    u32x2 q = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 x(=100+lane)", "swap32 y(=200+lane)", "swap16 x", "swap16 y"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]); printf("\n"); }
    return 0;
}
