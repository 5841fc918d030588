// Shared declarations for libvr_mi355.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace vr {

// ---------------------------------------------------------------------------------------------
// Error plumbing: every HIP call is checked; failures become C++ exceptions that the C ABI turns
// into negative status codes + a thread-local message (include/vr_mi355.h).
// ---------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define VR_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            throw ::vr::Error(-3, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

#define VR_CHECK(cond, code, msg)                                                                 \
    do {                                                                                          \
        if (!(cond)) throw ::vr::Error((code), (msg));                                            \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: remember it per
// (kernel instance, device) so that a process holding handles on several GPUs sets it on each of them.
inline void ensure_lds_attr(std::atomic<unsigned long long>& done_mask, const void* kern, int bytes) {
    int dev = 0;
    VR_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return;
    VR_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done_mask.fetch_or(bit, std::memory_order_release);
}

// Every API call runs on the handle's GPU and puts the caller's current device back afterwards
// (torch keeps its own notion of the current device per thread).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) VR_HIP(hipSetDevice(device));
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---------------------------------------------------------------------------------------------
// Launch profiler (profile.hip; bench.py's per-kernel-class roofline).  While a handle is profiling (vr_profile_begin), every
// kernel launched through VR_LAUNCH is bracketed by HIP events ON THE STREAM IT IS LAUNCHED ON and recorded under its own
// (demangled) kernel name together with the algorithmic FLOPs / HBM bytes noted for it: the executor notes a convolution's
// figures right before launch_conv (strong note), the launch wrappers of the element-wise kernels note theirs (weak: kept only
// if nothing is pending).  A note is consumed by the first launch after it.  Off (g_launch_prof == nullptr) the macro is a
// plain hipLaunchKernelGGL.
// ---------------------------------------------------------------------------------------------
struct LaunchProfiler;
extern thread_local LaunchProfiler* g_launch_prof;
void prof_note(double flops, double bytes, bool strong = false, const char* tag = nullptr);
void prof_note_update(double bytes, const char* tag);           // add bytes / tag to the pending strong note
void prof_note_clear();
LaunchProfiler* prof_create();
void prof_destroy(LaunchProfiler* p);
bool prof_owned_by_this_thread(const LaunchProfiler* p);      // g_launch_prof is thread-local: begin / end / close belong to one thread
void prof_abandon(LaunchProfiler* p);                         // closed from another thread: mark dead (leaked on purpose), never freed under a live VR_LAUNCH
void prof_collect(LaunchProfiler* p, double* conv_ms, double* conv_flops, double* conv_bytes, int* conv_launches, std::string* report,
                  bool dump);
void prof_before(const void* fn, const char* label, hipStream_t st);
void prof_after(hipStream_t st);
void prof_memset_async(void* p, int value, size_t bytes, hipStream_t st);      // hipMemsetAsync, recorded as "memset"

#define VR_LAUNCH(kern, grid, block, lds, st, ...)                                                                     \
    do {                                                                                                               \
        if (::vr::g_launch_prof) ::vr::prof_before(reinterpret_cast<const void*>(kern), #kern, st);                    \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                   \
        if (::vr::g_launch_prof) ::vr::prof_after(st);                                                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Activation tensors in HBM.
//
// Layout: [N, C, H, W] fp32, W (time frames) contiguous, arbitrary N/C/H strides so that band
// splits (lib/nets.py:88-90), frequency concatenation (nets.py:93,99) and the overlapping
// sliding-window crops (inference.py:44-48) are all *views* -- nothing is copied.
//
// Every conv output is stored RAW (pre-BatchNorm).  The BatchNorm affine (scale, shift per
// channel) and the activation slope travel with the tensor and are applied by the CONSUMER when
// it stages the tile into LDS:   value = act(raw * scale[c] + shift[c]),
//                                act(v) = v > 0 ? v : slope * v   (0 ReLU, 0.01 LeakyReLU, 1 identity).
// Eval mode fills (scale, shift) from running stats once; train mode fills them from batch stats
// after the producing conv -- the conv kernels are identical in both modes, and the raw tensor is
// exactly what BatchNorm backward needs.
// `hsplit`: rows h < hsplit use aff0, rows >= hsplit use aff1 (aux1/aux2 hold the low-band and
// high-band nets' outputs stacked along frequency, each with its own BatchNorm).
// ---------------------------------------------------------------------------------------------
struct Tensor {
    float* p = nullptr;
    int N = 0, C = 0, H = 0, W = 0;
    long long sN = 0, sC = 0, sH = 0;
    const float* aff0 = nullptr;   // [C][2] (scale, shift) on device, or null = identity
    const float* aff1 = nullptr;
    int hsplit = 1 << 30;
    float slope = 1.f;
    const float* post = nullptr;   // [N][C] post-activation multiplier (Dropout2d), or null
    float* g = nullptr;            // training: gradient w.r.t. the value consumers see (same strides as p)
};

// One input of a (virtually concatenated) convolution.
struct ConvSrc {
    const float* p;
    const float* aff0;
    const float* aff1;
    long long sN, sC, sH;
    int C, H, W;       // source dims (BEFORE the optional x2 upsample)
    int hsplit;
    float slope;
    int up;            // 1: bilinear x2, align_corners=True (lib/layers.py:52), fused into the load
    float rh, rw;      // (H-1)/(2H-1), (W-1)/(2W-1) as torch's area_pixel_compute_scale<float>
    const float* post; // [N][C] multiplier applied after the activation (Dropout2d keep-mask / 0.9), or null
    int zins;          // 1: the source occupies the even virtual coordinates, zeros in between
                       //    (data-gradient of a stride-2 conv = stride-1 conv over the zero-inserted gradient)
};

// One output segment of a conv launch (the data-gradient of a virtually concatenated input is
// split back to the tensors it came from).
struct ConvDst {
    float* p;                  // null: channels of this segment are not stored (e.g. grad of the network input)
    long long sN, sC, sH;
    int accumulate;            // 1: += (gradient accumulation over several consumers)
    int wshift;                // 1: output column w lands at 2*w (the rows use sH); see ConvArgs::tapmask
};

struct ConvArgs {
    ConvSrc src[3];
    int nsrc, c1, c2;          // src0 = channels [0,c1), src1 = [c1,c2), src2 = [c2,Cin)
    int Cin;
    const float* w;            // [Cin][KS*KS][CoutPad]  (K-major, cout contiguous)
    const float* wino;         // Winograd F(2x2,3x3) weights [Cin][16][CoutPad] (G g G^T), or null
    const void* wino6;         // the same weights as three bf16 planes [ceil(Cin/8)][16][3][CoutPad][8 channels] (bf16 == 2), or null
    const void* x3w;           // the direct weights as three bf16 planes [ceil(Cin/8)][KS*KS][3][CoutPad][8 channels] (conv_x3.hip), or null
    const float* bias;         // [Cout] or null
    int Cout, CoutPad;
    ConvDst dst[3];
    int d1, d2;                // dst0 = output channels [0,d1), dst1 = [d1,d2), dst2 = [d2,Cout)
    float* part;               // per-block BatchNorm partials [npt][Cout][2] (sum, sumsq) or null
    const float* epi;          // eval mode: folded BatchNorm [Cout][2] (scale, shift) applied in the epilogue
    float epi_slope;           //            together with the activation, so the stored tensor is final
    int N, Hout, Wout, Hin, Win;
    int pad_h, pad_w;
    int tiles_w, tiles_h, npt, nct;
    int dbg;                   // perf experiments only (VR_CONV_DBG): 1 = skip staging, 2 = skip MFMAs
    int bf16;                  // Winograd kernels only.  1: MFMA operands rounded to bf16 in registers (fp32 storage and accumulation);
                               // 2: fp32 products as six bf16 products of three-way split operands (conv_stage.h), needs wino6
    long long s2_cls_stride;   // conv_dma_s2d_kernel (fused form of the four parity classes below): floats between the class weight
    int s2_H, s2_W;            // arrays in `w`; full-resolution output dims (the four classes interleave into them)
    int tapmask;               // 0 = all taps; else bit t set = tap t of the 3x3 is used.  The data gradient of a
                               // stride-2 conv is four stride-1 convs over dz, one per output parity (ph, pw), with
                               // 1 / 2 / 2 / 4 live taps and outputs interleaved (dst.wshift, doubled row stride):
                               // 9 tap evaluations instead of the 36 of the zero-insertion form (conv_dma.hip only)
};

struct ConvShape {             // static description used by the launcher
    int KS, stride, dil_h, dil_w;
};

// Weight-gradient launch (wgrad_mfma.hip): `in` describes the conv's virtual input exactly like the
// forward launch (its Hout/Wout are the dims of dz); dz is the gradient at the raw conv output.
struct WgradArgs {
    ConvArgs in;
    const float* dz;
    long long zN, zC, zH;
    int Cout, CoutPad;
    float* part;               // scratch [P][Cin][KS*KS][CoutPad]
    long long part_stride;
    int P, tiles_w, tiles_h, npt, nchunks, nct;
    int dma;                   // 1: every source is a plain tensor -> the loader waves use LDS-DMA (no arithmetic)
    int bf16;                  // 1: bf16 MFMA operands (wgrad_wino.hip, wgrad_gemm.hip)
    int allow_wino;            // 1: 3x3 stride-1 layers with plain inputs may take the Winograd F(3x3,2x2) kernel (wgrad_wino.hip)
};
double launch_wgrad(const WgradArgs& a, const ConvShape& s, float* grad_out, int accumulate, hipStream_t st);
// Deferred slab sums (round 6): while a sink is set (thread-local), launch_wgrad records WHAT its wgrad_reduce launch would have summed
// instead of launching it -- 107 launches of ~10 us per train step -- and Model::backward() sums all of them in ONE launch at the end
// (the slabs live in the step's bump-allocated workspace until the next step).
struct WgReduceDesc { const float* part; long long stride; float* out; long long n; int P; int accumulate; long long blk0; };
void wgrad_defer_to(std::vector<WgReduceDesc>* sink);          // nullptr: launch_wgrad sums immediately again
int wgrad_reduce_vec(const WgReduceDesc* host_descs, int n);    // 4: every slab / gradient is 16-byte aligned and sized (a block sums 256 elements), else 1 (64)
void launch_wgrad_reduce_batched(const WgReduceDesc* d_descs, int n, long long total_blocks, int vec, hipStream_t st);
size_t wgrad_scratch_floats(const WgradArgs& a, const ConvShape& s);

// Returns the algorithmic FLOPs of the launch (2*MACs) for roofline accounting.
double launch_conv(const ConvArgs& a, const ConvShape& s, hipStream_t st);
size_t conv_part_count(const ConvArgs& a, const ConvShape& s);   // #partials rows (npt)
void conv_fill_tiling(ConvArgs& a, const ConvShape& s);
bool conv_dma_eligible(const ConvArgs& a, const ConvShape& s);    // the LDS-DMA kernel covers this launch

}  // namespace vr
