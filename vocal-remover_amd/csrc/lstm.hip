// Bidirectional LSTM recurrence of layers.LSTMModule (lib/layers.py:113-117,128).
//
// The input projection W_ih x_t + b_ih + b_hh for all time steps and both directions is hoisted
// into one 1x1-conv launch of the MFMA kernel (gx below), so what remains here is the strictly
// sequential part: 128 steps of  gates = gx_t + W_hh h_{t-1}  followed by the cell update.
// One workgroup per (sample, direction); W_hh^T lives in LDS for the whole sequence (4H*H floats,
// 64 KB at H=64), thread g owns gate row g, h is broadcast from LDS.  Latency-bound by design:
// the work per step is 4H*H FMAs; all (sample, direction) pairs run concurrently.
#include <cstdlib>

#include "kernels.h"

namespace vr {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void bilstm_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                              const float* __restrict__ whh_r, float* __restrict__ out, float* __restrict__ save,
                              int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = 4 * H;
    float* WT = lds;               // [H][G]
    float* hbuf = WT + H * G;      // [H]
    float* abuf = hbuf + H;        // [G]
    const int n = blockIdx.x, dir = blockIdx.y;
    const int g = threadIdx.x;
    const float* whh = dir ? whh_r : whh_f;
    for (int i = threadIdx.x; i < G * H; i += blockDim.x) {
        const int row = i / H, k = i % H;          // whh[row][k]
        WT[k * G + row] = whh[i];
    }
    if (g < H) hbuf[g] = 0.f;
    float c = 0.f;
    const float* gxp = gx + ((long long)n * 2 * G + (long long)dir * G + g) * T;
    float* outp = out + ((long long)n * 2 * H + (long long)dir * H + g) * T;
    __syncthreads();
    float pre = (g < G) ? gxp[dir ? T - 1 : 0] : 0.f;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        float nxt = 0.f;
        if (g < G && step + 1 < T) nxt = gxp[dir ? t - 1 : t + 1];   // prefetch next step's projection
        if (g < G) {
            float a = pre;
            for (int k = 0; k < H; ++k) a = fmaf(WT[k * G + g], hbuf[k], a);
            abuf[g] = a;
        }
        __syncthreads();
        if (g < H) {
            const float ig = sigmoidf_(abuf[g]);
            const float fg = sigmoidf_(abuf[H + g]);
            const float gg = tanhf(abuf[2 * H + g]);
            const float og = sigmoidf_(abuf[3 * H + g]);
            c = fg * c + ig * gg;
            const float h = og * tanhf(c);
            hbuf[g] = h;
            outp[t] = h;
            if (save) {
                float* sv = save + (((long long)n * 2 + dir) * T + t) * 5 * H + g;
                sv[0] = ig; sv[H] = fg; sv[2 * H] = gg; sv[3 * H] = og; sv[4 * H] = c;
            }
        }
        __syncthreads();
        pre = nxt;
    }
}

// Register-resident quad form for H in {16, 32, 64} (the shipped nets: nout_lstm = 128 -> H = 64), round 6: thread 4 u + q keeps row
// q * H + u of W_hh (H floats) in registers, h_{t-1} is broadcast from LDS with 16-byte reads, four independent FMA chains.  The four gates
// of a hidden unit sit in four ADJACENT LANES of one wave, so the cell update needs no LDS round trip and no barrier between the
// matrix-vector product and the gates -- four DPP quad broadcasts instead.  One barrier per step (h is double-buffered in LDS), every lane
// evaluates ONE transcendental for its gate (sigmoid(x) = 0.5 + 0.5 tanh(x / 2): one instruction stream for all four gates, per-lane
// constants), and h / the projections move to and from HBM four time steps at a time.  0.5 us per step; its predecessor (rounds 2-5: thread
// g = gate row g, two barriers, gate pre-activations through LDS, five libm transcendentals on one wave while three idled) took 0.9 us,
// the LDS form above 1.4 us.  S30 inference: 0.54 -> 0.28 ms of kernel time per step (gpurun_out r6call6; wall time unchanged -- the
// recurrence runs beside the x2 upsample on the side stream).
__device__ __forceinline__ float tanh_fast(float x) {
    // 1 - 2 / (1 + e^(2x)): v_exp_f32 + v_rcp_f32; saturates to +-1 through inf / 0, absolute error ~1e-7
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}
template <int H>
__global__ __launch_bounds__(4 * H) void bilstm_quad_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r, float* __restrict__ out,
                                                            float* __restrict__ save, int T) {
    constexpr int G = 4 * H;
    __shared__ __attribute__((aligned(16))) float hbuf[2][H];
    const int n = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x;
    const int u = tid >> 2, q = tid & 3;                         // hidden unit, gate (0 i, 1 f, 2 g, 3 o): row q * H + u of W_hh
    const int row = q * H + u;
    const float* whh = (dir ? whh_r : whh_f) + (long long)row * H;
    float w[H];
#pragma unroll
    for (int k = 0; k < H; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(whh + k);
        w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
    const float sc = q == 2 ? 1.f : 0.5f, mul = q == 2 ? 1.f : 0.5f, add = q == 2 ? 0.f : 0.5f;
    if (tid < H) { hbuf[0][tid] = 0.f; hbuf[1][tid] = 0.f; }
    float c = 0.f;
    const float* gxp = gx + ((long long)n * 2 * G + (long long)dir * G + row) * T;
    float* outp = out + ((long long)n * 2 * H + (long long)dir * H + u) * T;
    __syncthreads();
    // time runs in blocks of four steps: block b covers t = 4 b .. 4 b + 3 (forward) resp. the mirrored block in descending order
    const int nb = T >> 2;                                       // (launch_bilstm_train takes this kernel only when T % 4 == 0)
    float4 pre = *reinterpret_cast<const float4*>(gxp + (dir ? T - 4 : 0));
    int cur = 0;
    for (int b = 0; b < nb; ++b) {
        const int t0 = dir ? T - 4 - 4 * b : 4 * b;
        float4 nxt = pre;
        if (b + 1 < nb) nxt = *reinterpret_cast<const float4*>(gxp + (dir ? t0 - 4 : t0 + 4));      // the next block's projections
        float hq[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int j = dir ? 3 - s : s;                       // position inside the block, in processing order
            float a0 = j == 0 ? pre.x : (j == 1 ? pre.y : (j == 2 ? pre.z : pre.w)), a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int k = 0; k < H; k += 4) {
                const float4 h4 = *reinterpret_cast<const float4*>(&hbuf[cur][k]);       // same address in every lane: broadcast
                a0 = fmaf(w[k], h4.x, a0);
                a1 = fmaf(w[k + 1], h4.y, a1);
                a2 = fmaf(w[k + 2], h4.z, a2);
                a3 = fmaf(w[k + 3], h4.w, a3);
            }
            const float act = fmaf(tanh_fast(((a0 + a1) + (a2 + a3)) * sc), mul, add);     // sigmoid for i, f, o; tanh for g
            const int ai = __float_as_int(act);
            const float ig = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, true));     // quad_perm [0,0,0,0]
            const float fg = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, true));     // [1,1,1,1]
            const float gg = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, true));     // [2,2,2,2]
            const float og = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, true));     // [3,3,3,3]
            c = fmaf(fg, c, ig * gg);                            // (all four lanes of the quad carry the unit's cell state)
            const float h = og * tanh_fast(c);
            hq[j] = h;
            if (q == 0) hbuf[cur ^ 1][u] = h;
            if (save) {                                          // [n][dir][t][5H]: (i, f, g, o, c) per step, for the backward pass
                float* sv = save + (((long long)n * 2 + dir) * T + (t0 + j)) * 5 * H;
                sv[row] = act;
                if (q == 0) sv[4 * H + u] = c;
            }
            __syncthreads();
            cur ^= 1;
        }
        if (q == 0) *reinterpret_cast<float4*>(outp + t0) = make_float4(hq[0], hq[1], hq[2], hq[3]);
        pre = nxt;
    }
}

void launch_bilstm_train(const float* gx, const float* whh_f, const float* whh_r, float* out, float* save,
                         int N, int T, int H, hipStream_t st) {
    static const bool reg_form = !getenv("VR_LSTM_LDS");
    // (T = frames / 2 with frames % 16 == 0, workspace buffers are 256-byte aligned: the shipped nets always take this branch)
    if (reg_form && (H == 64 || H == 32 || H == 16) && (T & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        if (H == 64) VR_LAUNCH(bilstm_quad_kernel<64>, dim3(N, 2), dim3(256), 0, st, gx, whh_f, whh_r, out, save, T);
        else if (H == 32) VR_LAUNCH(bilstm_quad_kernel<32>, dim3(N, 2), dim3(128), 0, st, gx, whh_f, whh_r, out, save, T);
        else VR_LAUNCH(bilstm_quad_kernel<16>, dim3(N, 2), dim3(64), 0, st, gx, whh_f, whh_r, out, save, T);
        VR_HIP(hipGetLastError());
        return;
    }
    const int G = 4 * H;
    VR_CHECK(G <= 1024, -2, "LSTM hidden size per direction must be <= 256");
    const int threads = ((G + 63) / 64) * 64;
    const size_t lds = (size_t)(H * G + H + G) * sizeof(float);
    VR_CHECK(lds <= 160 * 1024, -2, "LSTM W_hh does not fit LDS");
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(bilstm_kernel), 160 * 1024);
    VR_LAUNCH(bilstm_kernel, dim3(N, 2), dim3(threads), lds, st, gx, whh_f, whh_r, out, save, T, H);
    VR_HIP(hipGetLastError());
}

void launch_bilstm(const float* gx, const float* whh_f, const float* whh_r, float* out,
                   int N, int T, int H, hipStream_t st) {
    launch_bilstm_train(gx, whh_f, whh_r, out, nullptr, N, T, H, st);
}

// ---------------------------------------------------------------------------------------------------
// Backward through time.  One workgroup per (sample, direction), walking the forward order in
// reverse; W_hh ([4H][H], gate-major) stays in LDS, dh_{t-1} = W_hh^T da_t is a 4-way split
// reduction over the gates.
// ---------------------------------------------------------------------------------------------------
__global__ void bilstm_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ save,
                                  const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                  float* __restrict__ dgx, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = 4 * H;
    float* W = lds;              // [G][H]
    float* da = W + G * H;       // [G]
    float* dhn = da + G;         // [H]
    float* part = dhn + H;       // [4][H]
    const int n = blockIdx.x, dir = blockIdx.y;
    const int j = threadIdx.x;
    const float* whh = dir ? whh_r : whh_f;
    for (int i = threadIdx.x; i < G * H; i += blockDim.x) W[i] = whh[i];
    if (j < H) dhn[j] = 0.f;
    float dc_next = 0.f;
    const float* svb = save + ((long long)n * 2 + dir) * T * 5 * H;
    const float* dhp = dh + ((long long)n * 2 * H + (long long)dir * H + j) * T;
    float* dgp = dgx + ((long long)n * 2 * G + (long long)dir * G) * T;
    __syncthreads();
    for (int step = T - 1; step >= 0; --step) {
        const int t = dir ? T - 1 - step : step;
        if (j < H) {
            const float* sv = svb + (long long)t * 5 * H + j;
            const float ig = sv[0], fg = sv[H], gg = sv[2 * H], og = sv[3 * H], c = sv[4 * H];
            float c_prev = 0.f;
            if (step > 0) c_prev = svb[(long long)(dir ? t + 1 : t - 1) * 5 * H + 4 * H + j];
            const float dht = dhp[t] + dhn[j];
            const float tc = tanhf(c);
            const float d_o = dht * tc;
            const float dc = dht * og * (1.f - tc * tc) + dc_next;
            const float d_i = dc * gg, d_g = dc * ig, d_f = dc * c_prev;
            dc_next = dc * fg;
            const float a_i = d_i * ig * (1.f - ig), a_f = d_f * fg * (1.f - fg);
            const float a_g = d_g * (1.f - gg * gg), a_o = d_o * og * (1.f - og);
            da[j] = a_i; da[H + j] = a_f; da[2 * H + j] = a_g; da[3 * H + j] = a_o;
            dgp[(long long)j * T + t] = a_i;
            dgp[(long long)(H + j) * T + t] = a_f;
            dgp[(long long)(2 * H + j) * T + t] = a_g;
            dgp[(long long)(3 * H + j) * T + t] = a_o;
        }
        __syncthreads();
        if (j < G) {
            const int q = j / H, k = j % H;
            float s = 0.f;
            for (int g = q * H; g < (q + 1) * H; ++g) s = fmaf(W[g * H + k], da[g], s);
            part[q * H + k] = s;
        }
        __syncthreads();
        if (j < H) dhn[j] = part[j] + part[H + j] + part[2 * H + j] + part[3 * H + j];
        __syncthreads();
    }
}

// Register-resident form (H <= 64): thread (q, k) keeps W_hh[qH..(q+1)H-1][k] (its quarter of column k) in
// registers, da_t is broadcast from LDS with 16-byte reads; the saved gates of step t-1 are prefetched.
template <int H>
__global__ __launch_bounds__(4 * H) void bilstm_bwd_reg_kernel(const float* __restrict__ dh, const float* __restrict__ save,
                                                               const float* __restrict__ whh_f,
                                                               const float* __restrict__ whh_r, float* __restrict__ dgx,
                                                               int T) {
    constexpr int G = 4 * H;
    __shared__ __attribute__((aligned(16))) float da[G];
    __shared__ float dhn[H];
    __shared__ float part[4 * H];
    const int n = blockIdx.x, dir = blockIdx.y;
    const int j = threadIdx.x;
    const int q = j / H, k = j % H;
    const float* whh = dir ? whh_r : whh_f;
    float w[H];
#pragma unroll
    for (int g = 0; g < H; ++g) w[g] = whh[(long long)(q * H + g) * H + k];
    if (j < H) dhn[j] = 0.f;
    float dc_next = 0.f;
    const float* svb = save + ((long long)n * 2 + dir) * T * 5 * H;
    const float* dhp = dh + ((long long)n * 2 * H + (long long)dir * H + j) * T;
    float* dgp = dgx + ((long long)n * 2 * G + (long long)dir * G) * T;
    __syncthreads();
    for (int step = T - 1; step >= 0; --step) {
        const int t = dir ? T - 1 - step : step;
        if (j < H) {
            const float* sv = svb + (long long)t * 5 * H + j;
            const float ig = sv[0], fg = sv[H], gg = sv[2 * H], og = sv[3 * H], c = sv[4 * H];
            float c_prev = 0.f;
            if (step > 0) c_prev = svb[(long long)(dir ? t + 1 : t - 1) * 5 * H + 4 * H + j];
            const float dht = dhp[t] + dhn[j];
            const float tc = tanhf(c);
            const float d_o = dht * tc;
            const float dc = dht * og * (1.f - tc * tc) + dc_next;
            const float d_i = dc * gg, d_g = dc * ig, d_f = dc * c_prev;
            dc_next = dc * fg;
            const float a_i = d_i * ig * (1.f - ig), a_f = d_f * fg * (1.f - fg);
            const float a_g = d_g * (1.f - gg * gg), a_o = d_o * og * (1.f - og);
            da[j] = a_i; da[H + j] = a_f; da[2 * H + j] = a_g; da[3 * H + j] = a_o;
            dgp[(long long)j * T + t] = a_i;
            dgp[(long long)(H + j) * T + t] = a_f;
            dgp[(long long)(2 * H + j) * T + t] = a_g;
            dgp[(long long)(3 * H + j) * T + t] = a_o;
        }
        __syncthreads();
        {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int g = 0; g < H; g += 4) {
                const float4 d4 = *reinterpret_cast<const float4*>(da + q * H + g);
                s0 = fmaf(w[g], d4.x, s0);
                s1 = fmaf(w[g + 1], d4.y, s1);
                s2 = fmaf(w[g + 2], d4.z, s2);
                s3 = fmaf(w[g + 3], d4.w, s3);
            }
            part[q * H + k] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (j < H) dhn[j] = part[j] + part[H + j] + part[2 * H + j] + part[3 * H + j];
        __syncthreads();
    }
}

void launch_bilstm_bwd(const float* dh, const float* save, const float* whh_f, const float* whh_r, float* dgx,
                       int N, int T, int H, hipStream_t st) {
    static const bool reg_form = !getenv("VR_LSTM_LDS");
    if (reg_form && (H == 64 || H == 32 || H == 16)) {
        if (H == 64) VR_LAUNCH(bilstm_bwd_reg_kernel<64>, dim3(N, 2), dim3(256), 0, st, dh, save, whh_f, whh_r, dgx, T);
        else if (H == 32) VR_LAUNCH(bilstm_bwd_reg_kernel<32>, dim3(N, 2), dim3(128), 0, st, dh, save, whh_f, whh_r, dgx, T);
        else VR_LAUNCH(bilstm_bwd_reg_kernel<16>, dim3(N, 2), dim3(64), 0, st, dh, save, whh_f, whh_r, dgx, T);
        VR_HIP(hipGetLastError());
        return;
    }
    const int G = 4 * H;
    const int threads = ((G + 63) / 64) * 64;
    const size_t lds = (size_t)(G * H + G + H + 4 * H) * sizeof(float);
    VR_CHECK(G <= 1024 && lds <= 160 * 1024, -2, "LSTM hidden size too large for the LDS-resident backward");
    static std::atomic<unsigned long long> attr_done{0};          // per device (bit = device index)
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(bilstm_bwd_kernel), 160 * 1024);
    VR_LAUNCH(bilstm_bwd_kernel, dim3(N, 2), dim3(threads), lds, st, dh, save, whh_f, whh_r, dgx, T, H);
    VR_HIP(hipGetLastError());
}

// dW_hh[g][k] = sum_{n,t} dgx[n][g][t] * h[n][k][t -/+ 1]   (forward / reverse direction): a small GEMM over (n,t).
// Workgroup = 16 gate rows x 64 hidden columns x one slice of the samples (blockIdx.y = slice * 2 + direction); per (sample,
// 64-frame block) the dgx rows and the shifted h rows are staged in LDS with coalesced reads -- the loads of the next block are in
// flight (registers) while thread (k, 4 gate rows) accumulates the 64 frames of this one.  Each slice writes its own partial slab
// and lstm_whh_reduce_kernel adds the slabs in a fixed order: deterministic, no atomics.
//
// The operand reads of a step are waited for with an explicit lgkmcnt(0) before the first FMA.  The compiler's own schedule
// (counted lgkmcnt waits, v_pk_fma_f32 straight behind them) is NOT safe on gfx950 beside conv_x3's TH = 8 kernels: with one of
// those resident on the same CU the last 16 lanes of the first dword of a broadcast ds_read_b128 were consumed before they had
// landed (tools/vgpr_canary.hip reproduces it in isolation: every run wrong in rows g%4 in {0,2}, columns 48..63; 0 wrong with
// the full wait) -- the "race" the batch-16 three-stream test caught in round 3 (hardware fact 5).
constexpr int WHH_MAX_SLICES = 8;

__global__ __launch_bounds__(256) void lstm_whh_grad_kernel(const float* __restrict__ dgx, const float* __restrict__ hout,
                                                            float* __restrict__ part, int N, int T, int H, int nslice) {
    __shared__ float hs[64][65];
    __shared__ float dgs[16][64];
    const int g0 = blockIdx.x * 16, dir = blockIdx.y & 1, slice = blockIdx.y >> 1, k0 = blockIdx.z * 64;
    const int G = 4 * H;
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;      // wq: which 4 of the 16 gate rows / staging row phase
    const int sh = dir ? 1 : -1;
    const int n_lo = (int)((long long)N * slice / nslice), n_hi = (int)((long long)N * (slice + 1) / nslice);
    const int tblocks = (T + 63) / 64, ntile = (n_hi - n_lo) * tblocks;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float ph[16], pd[4];
    auto fetch = [&](int tile) {
        const int n = n_lo + tile / tblocks, t = (tile % tblocks) * 64 + lane, th = t + sh;
        const bool ok = tile < ntile && t < T && th >= 0 && th < T;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + wq + 4 * j;
            ph[j] = (ok && k < H) ? hout[((long long)n * 2 * H + (long long)dir * H + k) * T + th] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = g0 + wq + 4 * j;
            pd[j] = (ok && g < G) ? dgx[((long long)n * 2 * G + (long long)dir * G + g) * T + t] : 0.f;
        }
    };
    fetch(0);
    for (int tile = 0; tile < ntile; ++tile) {
#pragma unroll
        for (int j = 0; j < 16; ++j) hs[wq + 4 * j][lane] = ph[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) dgs[wq + 4 * j][lane] = pd[j];
        __syncthreads();
        fetch(tile + 1);
#pragma unroll 1
        for (int tb = 0; tb < 64; tb += 8) {
            float hv[8], dg[4][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = hs[lane][tb + j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) dg[q][j] = dgs[wq * 4 + q][tb + j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(hv[j]));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(dg[q][j]));
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(dg[q][j], hv[j], acc[q]);
        }
        __syncthreads();
    }
    const int k = k0 + lane;
    float* slab = part + (size_t)(slice * 2 + dir) * G * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = g0 + wq * 4 + q;
        if (g < G && k < H) slab[g * H + k] = acc[q];
    }
}

__global__ __launch_bounds__(256) void lstm_whh_reduce_kernel(const float* __restrict__ part, float* dwf, float* dwr, int GH, int nslice,
                                                              int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x, dir = blockIdx.y;
    if (i >= GH) return;
    float s = 0.f;
    for (int sl = 0; sl < nslice; ++sl) s += part[(size_t)(sl * 2 + dir) * GH + i];
    float* dw = dir ? dwr : dwf;
    dw[i] = accumulate ? dw[i] + s : s;
}

static int whh_slices(int N) { return N < WHH_MAX_SLICES ? (N < 1 ? 1 : N) : WHH_MAX_SLICES; }

size_t lstm_whh_grad_scratch_floats(int N, int H) { return (size_t)whh_slices(N) * 2 * 4 * H * H; }

void launch_lstm_whh_grad(const float* dgx, const float* hout, float* dwhh_f, float* dwhh_r, int N, int T, int H,
                          int accumulate, float* part, hipStream_t st) {
    const int ns = whh_slices(N), GH = 4 * H * H;
    VR_LAUNCH(lstm_whh_grad_kernel, dim3((4 * H + 15) / 16, 2 * ns, (H + 63) / 64), dim3(256), 0, st, dgx, hout, part, N, T,
                       H, ns);
    VR_HIP(hipGetLastError());
    VR_LAUNCH(lstm_whh_reduce_kernel, dim3((GH + 255) / 256, 2), dim3(256), 0, st, part, dwhh_f, dwhh_r, GH, ns, accumulate);
    VR_HIP(hipGetLastError());
}

}  // namespace vr
