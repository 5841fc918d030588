// Direct 3x3 stride-1 convolution over activations STORED as bf16 planes (eval path of mfma_mode 2; lib/layers.py:12-20).
//
// conv_x3.hip forms every fp32 product from six bf16 products, but pays for it in the consumer: every input pixel is loaded
// into registers (buffer_load_dword + hand-placed waits), split into its three bf16 planes (11 VALU per channel pair) and stored to
// LDS once per cout tile and per halo overlap -- and on gfx950 VALU work is not hidden behind the matrix pipe.
// Here the split happens ONCE, where it is free or cheap:
//   * a tensor that feeds 3x3 stride-1 convs is stored as   [N][G = ceil(C/8)][3 planes][H][W]   of 16-byte units
//     (unit = the 8 channels of one pixel in one plane, bf16; x = p1 + p2 + p3 exactly, conv_stage.h) -- 6 B per element instead
//     of 4: inference runs at ~1.2 of 8 TB/s, the bandwidth is there;
//   * producers write it: this kernel's own epilogue (planes and / or fp32), upsample2x_planes_kernel (the decoder's bilinear x2,
//     HBM-bound: its split VALU costs nothing), to_planes_kernel (thin tensors: network input, stage outputs, LSTM branch);
//   * the consumer pulls units by LDS-DMA straight into the MFMA operand image P[plane][halo pixel] -- no pixel registers, no
//     split pass, no per-cout-tile re-split; conv zero padding = out-of-range DMA offsets.
// Every source of the virtual concat occupies whole 8-channel groups (channels beyond its C are zero units with zero weights):
// the weight table [chunk][tap][plane][cout][8 ch] is built per layer for that padded channel order (x3p_weights_kernel).
//
// Workgroup = 512 threads / 8 waves, output tile TH x 32 pixels x MT couts, one per CU: wave w owns rows w*WN .. w*WN+WN-1 of the
// tile (WN = TH / 8) and all MT couts.  P and the weight slab of a chunk are BOTH double-buffered: the DMA of chunk k+1 is issued
// before the multiply phase of chunk k and waited for after it, ONE barrier per chunk.  The multiply phase is conv_x3.hip's:
//     [a1|a1][b1|b2] + [a2|a2][b1|b2] + [a3|a1][b1|b3] = a1b1 + a1b2 + a2b1 + a2b2 + a3b1 + a1b3
// three v_mfma_f32_32x32x16_bf16 per tap and (cout block, row) pair, operands of tap t+1 read between the MFMA groups of tap t.
#include <cstdlib>

#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"

namespace vr {

template <int MT, int TH>
struct X3pCfg {
    static constexpr int TW = 32, KK = 9, NWAVE = 8, NT = 512;
    static constexpr int TH_in = TH + 2, PW = TW + 2;
    static constexpr int NSLOT = TH_in * PW;                      // halo pixels
    static constexpr int PLANE = NSLOT * 16;
    static constexpr int NPU = 3 * NSLOT;                         // 16-byte units of one chunk's pixel image
    static constexpr int P_BYTES = NPU * 16;
    static constexpr int NPI = (NPU + NT - 1) / NT;               // DMA rounds (one unit per thread and round)
    static constexpr int NWP = KK * 3 * MT;                       // 16-byte weight operands per chunk
    static constexpr int W_BYTES = NWP * 16;
    static constexpr int NWI = (NWP + NT - 1) / NT;
    static constexpr int WM = MT / 32, WN = TH / 8;
    static constexpr int W_OFF = 2 * P_BYTES;
    static constexpr int E_OFF = W_OFF + 2 * W_BYTES;             // bias, scale, shift of the cout tile [3][MT] fp32
    static constexpr int LDS_BYTES = E_OFF + 3 * MT * 4;
    static_assert(TH % 8 == 0 && MT % 32 == 0 && LDS_BYTES <= 160 * 1024, "tile");
};

template <int MT, int TH>
__global__ __launch_bounds__(512, 1) void conv_x3p_kernel(const X3pArgs a) {
    using Cfg = X3pCfg<MT, TH>;
    constexpr int TW = Cfg::TW, KK = Cfg::KK, PW = Cfg::PW, NSLOT = Cfg::NSLOT, WM = Cfg::WM, WN = Cfg::WN, NPI = Cfg::NPI, NWI = Cfg::NWI,
                  NPU = Cfg::NPU, NWP = Cfg::NWP;
    extern __shared__ __attribute__((aligned(16))) char smem_x3p[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // every XCD walks its own contiguous, row-major range of pixel tiles (conv_x3.hip: halo rows and shared 128-byte lines then
    // come from that XCD's L2)
    const int per_xcd = (a.npt + 7) >> 3;
    const int pt = xcd * per_xcd + rr / a.nct;
    if (pt >= a.npt || rr / a.nct >= per_xcd) return;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int n = pt / tiles_per_img;
    const int trem = pt - n * tiles_per_img;
    const int h0 = (trem / a.tiles_w) * TH;
    const int w0 = (trem % a.tiles_w) * TW;
    const int co0 = ct * MT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)smem_x3p;
    const long long HW = (long long)a.H * a.W;

    // ---- this thread's units of the pixel image: byte offset inside one (sample, channel group) = ((plane*H + hi)*W + wi)*16 ----
    unsigned poff[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int u = (wave + 8 * i) * 64 + lane;
        const int plane = u / NSLOT, s = u - plane * NSLOT;
        const int r = s / PW, c = s - r * PW;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = u < NPU && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
        poff[i] = ok ? (unsigned)(((long long)plane * HW + (long long)hi * a.W + wi) * 16) : 0x80000000u;
    }
    unsigned woff[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int q = (wave + 8 * i) * 64 + lane;
        const int m = q % MT, tp = q / MT;
        woff[i] = (unsigned)((tp * a.CoutPad + m) * 16);
    }
    const unsigned group_bytes = (unsigned)(3 * HW * 16);
    const long long wchunk_bytes = (long long)KK * 3 * a.CoutPad * 16;
    // the channel groups are visited strictly in order: running scalar state of the virtual concat
    const char* gp = a.src[0].p + (long long)n * a.src[0].sN;
    long long gstep = a.src[0].sG;
    int gleft = a.src[0].ngroups, gsi = 0;
    auto issue = [&](int k) {                                      // pixel image + weight slab of chunk k -> buffers k & 1
        if (gleft == 0) {                                          // (a source has at least one group: one step at most)
            ++gsi;
            if (gsi == 1) { gp = a.src[1].p + (long long)n * a.src[1].sN; gstep = a.src[1].sG; gleft = a.src[1].ngroups; }
            else { gp = a.src[2].p + (long long)n * a.src[2].sN; gstep = a.src[2].sG; gleft = a.src[2].ngroups; }
        }
        i32x4 pr = make_rsrc(reinterpret_cast<const float*>(gp), group_bytes);
        gp += gstep; --gleft;
        const char* wb = static_cast<const char*>(a.w) + k * wchunk_bytes + (long long)co0 * 16;
        i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), (unsigned)(wchunk_bytes - (long long)co0 * 16));
        settle_rsrc(pr);
        settle_rsrc(wr);
        const unsigned pb = lds0 + (unsigned)((k & 1) * Cfg::P_BYTES);
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int j = wave + 8 * i;
            if ((j + 1) * 64 <= NPU) dma16(pb + j * 1024, poff[i], pr);
            else if (j * 64 + lane < NPU) dma16(pb + j * 1024, poff[i], pr);
        }
        const unsigned ws_b = lds0 + (unsigned)(Cfg::W_OFF + (k & 1) * Cfg::W_BYTES);
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int j = wave + 8 * i;
            if ((j + 1) * 64 <= NWP) dma16(ws_b + j * 1024, woff[i], wr);
            else if (j * 64 + lane < NWP) dma16(ws_b + j * 1024, woff[i], wr);
        }
    };

    const int khalf = lane >> 5, l31 = lane & 31;
    // B operands [b1|b2] and [b1|b3]: lanes 0-31 read plane 0, lanes 32-63 plane 1 resp. 2; pixel (row wave*WN + ni + ty, col l31 + tx)
    const int bb0 = (khalf * NSLOT + wave * WN * PW + l31) * 16;
    const int bb1 = (2 * khalf * NSLOT + wave * WN * PW + l31) * 16;
    // A operands [a1|a1], [a2|a2], [a3|a1]
    const int ab0 = l31 * 16, ab1 = (MT + l31) * 16, ab2 = ((khalf ? 0 : 2) * MT + l31) * 16;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // epilogue constants of this cout tile, parked in LDS behind the first wait
    float ecv[3] = {0.f, 1.f, 0.f};
    if (tid < MT) {
        const int ec = co0 + tid;
        const int ecc = ec < a.Cout ? ec : a.Cout - 1;
        ecv[0] = a.bias ? a.bias[ecc] : 0.f;
        ecv[1] = a.epi ? a.epi[2 * ecc] : 1.f;
        ecv[2] = a.epi ? a.epi[2 * ecc + 1] : 0.f;
    }
    issue(0);
    if (tid < MT) {
        float* E = reinterpret_cast<float*>(smem_x3p + Cfg::E_OFF);
        E[tid] = ecv[0]; E[MT + tid] = ecv[1]; E[2 * MT + tid] = ecv[2];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int k = 0; k < a.nchunk; ++k) {
        const bool more = k + 1 < a.nchunk;
        if (more) issue(k + 1);                        // buffers (k+1)&1 were last read in chunk k-1: the barrier below closed that
        {
            const char* Pb = smem_x3p + (k & 1) * Cfg::P_BYTES;
            const char* Wb = smem_x3p + Cfg::W_OFF + (k & 1) * Cfg::W_BYTES;
            vr_bf16x8 A[2][3][WM], B[2][2][WN];
            auto read_part = [&](int t, int buf, int part) {
                const int ty = t / 3, tx = t % 3;
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) {
                    const char* q = Wb + (t * 3 * MT + mi * 32) * 16;
                    if (part == 0) A[buf][2][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab2);
                    if (part == 1) A[buf][1][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab1);
                    if (part == 2) A[buf][0][mi] = *reinterpret_cast<const vr_bf16x8*>(q + ab0);
                }
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int o = ((ni + ty) * PW + tx) * 16;
                    if (part == 0) B[buf][1][ni] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb1 + o);
                    if (part == 1) B[buf][0][ni] = *reinterpret_cast<const vr_bf16x8*>(Pb + bb0 + o);
                }
            };
            auto mfma_group = [&](int buf, int ja, int jb) {
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = mfma_bf16x16(A[buf][ja][mi], B[buf][jb][ni], acc[mi][ni]);
            };
            read_part(0, 0, 0); read_part(0, 0, 1); read_part(0, 0, 2);
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int cur = t & 1;
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 2, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < KK) read_part(t + 1, cur ^ 1, 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(cur, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of chunk k+1 has landed
            __builtin_amdgcn_s_barrier();                                  // everyone's has; everyone is done reading chunk k
            asm volatile("" ::: "memory");
        }
    }

    // ---------------- epilogue: bias, folded BatchNorm + activation; fp32 and / or bf16-plane destinations -----------------------
    const float* E = reinterpret_cast<const float*>(smem_x3p + Cfg::E_OFF);
    const int Gout = (a.Cout + 7) >> 3;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                               // registers 4j .. 4j+3: couts cb + 0..3 of this lane half
            const int cl = mi * 32 + 8 * j + 4 * khalf;             // first of this lane's four couts, within the tile
            const int cb = co0 + cl;
            float eb[4], esc[4], esh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { eb[q] = E[cl + q]; esc[q] = E[MT + cl + q]; esh[q] = E[2 * MT + cl + q]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (hardware fact 5: full wait before the first consumer)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(eb[q]), "+v"(esc[q]), "+v"(esh[q]));
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int ho = h0 + wave * WN + ni, wo = w0 + l31;
                if (ho >= a.H || wo >= a.W) continue;
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = act_apply(fmaf(acc[mi][ni][4 * j + q] + eb[q], esc[q], esh[q]), a.slope);
                    y[q] = (cb + q < a.Cout) ? v : 0.f;             // padded couts of the last group are stored as zeros
                }
                if (a.out) {
                    float* q0 = a.out + (long long)n * a.oN + (long long)cb * a.oC + (long long)ho * a.oH + wo;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (cb + q < a.Cout) q0[q * a.oC] = y[q];
                }
                if (a.opl && (cb >> 3) < Gout) {
                    int p1a, p2a, p3a, p1b, p2b, p3b;
                    split3_pair(y[0], y[1], p1a, p2a, p3a);
                    split3_pair(y[2], y[3], p1b, p2b, p3b);
                    char* u = a.opl + (((long long)n * Gout + (cb >> 3)) * 3 * HW + (long long)ho * a.W + wo) * 16 + khalf * 8;
                    vr_i32x2 v1, v2, v3;
                    v1[0] = p1a; v1[1] = p1b; v2[0] = p2a; v2[1] = p2b; v3[0] = p3a; v3[1] = p3b;
                    *reinterpret_cast<vr_i32x2*>(u) = v1;
                    *reinterpret_cast<vr_i32x2*>(u + HW * 16) = v2;
                    *reinterpret_cast<vr_i32x2*>(u + 2 * HW * 16) = v3;
                }
            }
        }
    }
}

// ---- weights for the padded channel order: w [Cin][9][CoutPad] fp32 -> [nchunk][9][3][CoutPad][8 channels] bf16 ---------------
// chunk c, channel slot e (0..7): source s = the segment c falls in, channel = seg_start[s] + 8 * (c - chunk_start[s]) + e when that is
// below seg_start[s] + seg_len[s], else a zero weight (the slot is padding of the source's last group).
__global__ void x3p_weights_kernel(const X3pWDesc* __restrict__ d) {
    const X3pWDesc e = d[blockIdx.y];
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nslots = e.nchunk * 8;
    if (gid >= (long long)nslots * 9 * e.CoutPad) return;
    const int co = (int)(gid % e.CoutPad);
    const int t = (int)((gid / e.CoutPad) % 9);
    const int slot = (int)(gid / ((long long)e.CoutPad * 9));
    const int c = slot >> 3, el = slot & 7;
    const int g0 = (e.seg[0] + 7) >> 3, g1 = g0 + ((e.seg[1] + 7) >> 3);
    const int s = (c >= g0) + (c >= g1);
    const int cstart = s == 0 ? 0 : (s == 1 ? e.seg[0] : e.seg[0] + e.seg[1]);
    const int gstart = s == 0 ? 0 : (s == 1 ? g0 : g1);
    const int local = 8 * (c - gstart) + el;
    const int clen = s == 0 ? e.seg[0] : (s == 1 ? e.seg[1] : e.seg[2]);
    const float v = local < clen ? e.w[((long long)(cstart + local) * 9 + t) * e.CoutPad + co] : 0.f;
    int p1, p2, p3;
    split3_pair(v, 0.f, p1, p2, p3);
    unsigned short* q = static_cast<unsigned short*>(e.o) + ((((long long)c * 9 + t) * 3) * e.CoutPad + co) * 8 + el;
    q[0] = (unsigned short)(p1 & 0xffff);
    q[(long long)e.CoutPad * 8] = (unsigned short)(p2 & 0xffff);
    q[2LL * e.CoutPad * 8] = (unsigned short)(p3 & 0xffff);
}

size_t x3p_weights_bytes(int nchunk, int CoutPad) { return (size_t)nchunk * 9 * 3 * CoutPad * 16; }

void launch_x3p_weights(const X3pWDesc* d_descs, int n, long long max_elems, hipStream_t st) {
    if (n <= 0) return;
    VR_LAUNCH(x3p_weights_kernel, dim3((unsigned)((max_elems + 255) / 256), (unsigned)n), dim3(256), 0, st, d_descs);
    VR_HIP(hipGetLastError());
}

// ---- producers of plane tensors that are not a conv epilogue -------------------------------------------------------------------
// fp32 tensor view (plain values: eval tensors carry no pending affine) -> planes [N][G][3][H][W] units
__global__ __launch_bounds__(256) void to_planes_kernel(Tensor x, char* __restrict__ out, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % x.W);
    long long t = gid / x.W;
    const int h = (int)(t % x.H); t /= x.H;
    const int G = (x.C + 7) >> 3;
    const int g = (int)(t % G);
    const int n = (int)(t / G);
    const float* p = x.p + (long long)n * x.sN + (long long)(8 * g) * x.sC + (long long)h * x.sH + w;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (8 * g + e < x.C) ? p[e * x.sC] : 0.f;
    vr_i32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int a, b, c;
        split3_pair(v[2 * j], v[2 * j + 1], a, b, c);
        ph[j] = a; pm[j] = b; pl[j] = c;
    }
    const long long HW = (long long)x.H * x.W;
    char* u = out + (((long long)n * G + g) * 3 * HW + (long long)h * x.W + w) * 16;
    *reinterpret_cast<vr_i32x4*>(u) = ph;
    *reinterpret_cast<vr_i32x4*>(u + HW * 16) = pm;
    *reinterpret_cast<vr_i32x4*>(u + 2 * HW * 16) = pl;
}

void launch_to_planes(const Tensor& x, void* out, hipStream_t st) {
    const long long total = (long long)x.N * ((x.C + 7) / 8) * x.H * x.W;
    prof_note(0.0, (double)x.N * x.H * x.W * (4.0 * x.C + 48.0 * ((x.C + 7) / 8)));
    VR_LAUNCH(to_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, static_cast<char*>(out), total);
    VR_HIP(hipGetLastError());
}

// decoder F.interpolate(x2, bilinear, align_corners=True) (lib/layers.py:52) of a plain fp32 tensor, written as planes of the
// [2H][2W] result; the interpolation is upsample2x_kernel's (pointwise.hip), term for term
__global__ __launch_bounds__(256) void upsample2x_planes_kernel(Tensor x, char* __restrict__ out, float rh, float rw, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int W2 = 2 * x.W, H2 = 2 * x.H;
    const int wi = (int)(gid % W2);
    long long t = gid / W2;
    const int hi = (int)(t % H2); t /= H2;
    const int G = (x.C + 7) >> 3;
    const int g = (int)(t % G);
    const int n = (int)(t / G);
    const float h1r = rh * (float)hi, w1r = rw * (float)wi;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < x.H - 1) ? 1 : 0, w1p = (w1 < x.W - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float* p = x.p + (long long)n * x.sN + (long long)(8 * g) * x.sC + (long long)h1 * x.sH + w1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (8 * g + e < x.C) {
            const float* q = p + e * x.sC;
            const float v00 = q[0], v01 = q[w1p], v10 = q[h1p * x.sH], v11 = q[h1p * x.sH + w1p];
            v[e] = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
        } else {
            v[e] = 0.f;
        }
    }
    vr_i32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int a, b, c;
        split3_pair(v[2 * j], v[2 * j + 1], a, b, c);
        ph[j] = a; pm[j] = b; pl[j] = c;
    }
    const long long HW = (long long)H2 * W2;
    char* u = out + (((long long)n * G + g) * 3 * HW + (long long)hi * W2 + wi) * 16;
    *reinterpret_cast<vr_i32x4*>(u) = ph;
    *reinterpret_cast<vr_i32x4*>(u + HW * 16) = pm;
    *reinterpret_cast<vr_i32x4*>(u + 2 * HW * 16) = pl;
}

void launch_upsample2x_planes(const Tensor& x, void* out, hipStream_t st) {
    const float rh = (x.H > 0) ? (float)(x.H - 1) / (float)(2 * x.H - 1) : 0.f;
    const float rw = (x.W > 0) ? (float)(x.W - 1) / (float)(2 * x.W - 1) : 0.f;
    const long long total = (long long)x.N * ((x.C + 7) / 8) * 4 * x.H * x.W;
    prof_note(0.0, (double)x.N * x.H * x.W * (4.0 * x.C + 4.0 * 48.0 * ((x.C + 7) / 8)));
    VR_LAUNCH(upsample2x_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, static_cast<char*>(out), rh, rw, total);
    VR_HIP(hipGetLastError());
}

// planes -> fp32 [N][C][H][W] (x = p1 + p2 + p3, exact): tests and debug taps
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const char* __restrict__ pl, float* __restrict__ out, int C, int H, int W, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int w = (int)(gid % W);
    long long t = gid / W;
    const int h = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const int G = (C + 7) >> 3;
    const long long HW = (long long)H * W;
    const unsigned short* u = reinterpret_cast<const unsigned short*>(pl + (((long long)n * G + (c >> 3)) * 3 * HW + (long long)h * W + w) * 16) + (c & 7);
    const float a = __uint_as_float((unsigned)u[0] << 16), b = __uint_as_float((unsigned)u[HW * 8] << 16), d = __uint_as_float((unsigned)u[2 * HW * 8] << 16);
    out[gid] = (a + b) + d;
}

void launch_planes_to_f32(const char* pl, float* out, int N, int C, int H, int W, hipStream_t st) {
    const long long total = (long long)N * C * H * W;
    VR_LAUNCH(planes_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pl, out, C, H, W, total);
    VR_HIP(hipGetLastError());
}

// ---- host side --------------------------------------------------------------------------------------------------------------
bool x3p_enabled() {
    static const bool on = [] { const char* e = getenv("VR_CONV_X3P"); return e && atoi(e) != 0; }();      // default OFF (see the header)
    return on;
}

template <int MT, int TH>
static void x3p_launch_t(X3pArgs a, hipStream_t st) {
    using Cfg = X3pCfg<MT, TH>;
    auto kern = conv_x3p_kernel<MT, TH>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    a.tiles_w = (a.W + 31) / 32;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.npt = a.N * a.tiles_h * a.tiles_w;
    a.nct = a.CoutPad / MT;
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

void x3p_launch(const X3pArgs& a, hipStream_t st) {
    static const int env_th = getenv("VR_X3P_TH") ? atoi(getenv("VR_X3P_TH")) : 0;
    const char* dbg_th = getenv("VR_X3P_TH_DEBUG");                 // (tests: vr_debug_kernel("conv_planes") sets it around its launch)
    const int force_th = dbg_th ? atoi(dbg_th) : env_th;
    const bool m64 = a.CoutPad % 64 == 0;
    // 16-row tiles while they still give every CU a few workgroups; 8-row tiles for the small layers
    const long long tiles16 = (long long)a.N * ((a.H + 15) / 16) * ((a.W + 31) / 32) * (a.CoutPad / (m64 ? 64 : 32));
    int TH = tiles16 >= 512 ? 16 : 8;
    if (force_th == 8 || force_th == 16) TH = force_th;
    if (m64) { if (TH == 16) x3p_launch_t<64, 16>(a, st); else x3p_launch_t<64, 8>(a, st); }
    else { if (TH == 16) x3p_launch_t<32, 16>(a, st); else x3p_launch_t<32, 8>(a, st); }
}

}  // namespace vr
