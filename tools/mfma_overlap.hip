// Microbenchmark: do MFMA waves and VALU / LDS / global-load waves of the same workgroup overlap
// on gfx950?  Block = 4 MFMA waves (one per SIMD) + 8 worker waves.  Two matrix instructions: v_mfma_f32_32x32x2_f32 (runs at
// the fp32 vector rate) and v_mfma_f32_32x32x16_bf16 (the bf16 matrix pipe); worker kind 5 = the three-way bf16 split of
// conv_stage.h (v_cvt_pk_bf16_f32 + shifts + subtracts).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o tools/mfma_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dma4(unsigned lds_base, unsigned voff, i32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

// kind: 0 VALU fma chains, 1 LDS writes, 2 global loads (L2-resident), 3 LDS reads, 4 LDS-DMA (no VALU in the loop)
__global__ __launch_bounds__(768) void overlap(const float* __restrict__ in, float* __restrict__ out, int mfma_iters,
                                               int work_iters, int kind, int mtype) {
    __shared__ float lds[16384];
    const int tid = threadIdx.x, wave = tid >> 6;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = in[tid], b = in[256 + tid];
        if (mtype == 0) {
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                a = -a;
            }
        } else {
            bf16x8 av, bv;
            for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b - j); }
            for (int it = 0; it < 2 * mfma_iters; ++it) {          // (half the duration per instruction: same phase length)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
                av[0] = -av[0];
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else if (kind == 0) {
        float x0 = in[tid], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        const float m = in[tid & 255] * 1e-3f;
        for (int it = 0; it < work_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = fmaf(x0, m, 1.f); x1 = fmaf(x1, m, 1.f); x2 = fmaf(x2, m, 1.f); x3 = fmaf(x3, m, 1.f);
            }
        }
        s = x0 + x1 + x2 + x3;
    } else if (kind == 5) {
        float x0 = in[tid], x1 = x0 + 1.5f;
        int accp = 0;
        for (int it = 0; it < work_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                f32x2 v; v[0] = x0; v[1] = x1;
                const int p1 = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
                v[0] = x0 - __int_as_float(p1 << 16); v[1] = x1 - __int_as_float(p1 & (int)0xffff0000);
                const int p2 = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
                v[0] -= __int_as_float(p2 << 16); v[1] -= __int_as_float(p2 & (int)0xffff0000);
                const int p3 = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
                accp ^= p1 + p2 + p3;
                x0 += 1.25f; x1 -= 0.75f;
            }
        }
        s = (float)accp;
    } else if (kind == 1) {
        for (int it = 0; it < work_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) lds[(tid - 256 + u * 512 + it) & 16383] = (float)it;
        }
        __builtin_amdgcn_s_waitcnt(0);
        s = lds[tid];
    } else if (kind == 2) {
        const float* p = in + (tid - 256);
        for (int it = 0; it < work_iters; ++it) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[((it * 8 + u) * 512 + blockIdx.x * 4096) & ((1 << 22) - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    } else if (kind == 4) {
        const unsigned long long b = (unsigned long long)in;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffull));
        r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffull));
        r[2] = 1 << 24;
        r[3] = 0x00020000;
        const unsigned base = (unsigned)(size_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
        const unsigned vo = (unsigned)((tid - 256) * 4 + (blockIdx.x & 63) * 4096);
        for (int it = 0; it < work_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) dma4(base + u * 256, vo + u * 32768, r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int it = 0; it < work_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s += lds[(tid - 256 + u * 512 + it) & 16383];
        }
    }
    out[blockIdx.x * 768 + tid] = s;
}

int main() {
    const int blocks = 256 * 4;
    float *in, *out;
    (void)hipMalloc(&in, (1 << 22) * 4 + 4096);
    (void)hipMalloc(&out, blocks * 768 * 4);
    float* h = (float*)malloc((1 << 22) * 4);
    for (int i = 0; i < (1 << 22); ++i) h[i] = (float)rand() / (float)RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h, (1 << 22) * 4, hipMemcpyHostToDevice);
    const char* names[6] = {"VALU fma", "LDS write", "global load", "LDS read", "LDS-DMA", "bf16 split"};
    const int witers[6] = {40000, 20000, 3000, 20000, 3000, 30000};
    for (int mtype = 0; mtype < 2; ++mtype)
    for (int kind = 0; kind < 6; ++kind) {
        float t[3];
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int mi = cfg == 1 ? 0 : 10000, wi = cfg == 0 ? 0 : witers[kind];
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(overlap, dim3(blocks), dim3(768), 0, 0, in, out, mi, wi, kind, mtype);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&t[cfg], e0, e1);
            }
        }
        printf("%s %-12s mfma-only %.2f ms | work-only %.2f ms | both %.2f ms (sum %.2f, max %.2f)\n", mtype ? "bf16 32x32x16" : "f32 32x32x2  ", names[kind], t[0], t[1],
               t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
    }
    return 0;
}
