// micro-test: direction of DPP row_shr:1 / row_shl:1 (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, false);        // row_shr:1
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x101, 0xf, 0xf, false);   // row_shl:1
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("row_shr:1 "); for (int i = 0; i < 20; ++i) printf("%d ", h[i]); printf("\nrow_shl:1 "); for (int i = 0; i < 20; ++i) printf("%d ", h[64 + i]); printf("\n");
    return 0;
}
