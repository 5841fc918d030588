// Direct 3x3 stride-1 convolution, fp32-grade products from three fp16 products (conv_x3h.hip's arithmetic, "mfma_mode" 3;
// lib/layers.py:12-20), scheduled as an enforced PING-PONG of two wave groups (round 6).
//
// conv_x3h.hip runs four waves that all march through  multiply -> barrier -> split -> barrier  in lockstep; whether the matrix pipe
// is fed while one workgroup loads / splits depends on what the co-resident workgroup happens to be doing, and the vector-memory
// bookkeeping of the prefetch (~100 instructions per pair of channels) sits between the matrix instructions of the multiply phase
// (profiles/r05_x3h_phase_trace.txt: 1.8 k cycles of matrix instructions in a 7.5 k-cycle chunk, MFMA busy 0.22-0.42).
//
// Here a workgroup is 512 threads = two groups of four waves; wave w and wave w + 4 share a SIMD.  In every phase ONE group multiplies
// (ds_read_b128 + v_mfma only: no vector memory, no address arithmetic) while the OTHER group does everything else for the chunks to
// come, and the roles swap at one barrier per phase:
//
//     phase       2k                         2k+1                       2k+2                  ...
//     group A     multiply chunk k           load phase for chunk k+1   multiply chunk k+1
//     group B     load phase for chunk k+1   multiply chunk k           load phase for k+2
//
// The two groups own different OUTPUT tiles over shared pixels or shared weights:
//     SPLITM   two cout tiles (MT each) over the SAME TH x 32 pixels: the split pixel planes P are shared, so the load / split work per
//              matrix instruction halves (Cout >= 64);
//     !SPLITM  two row tiles (TH rows each, one 2 TH x 32 halo tile) for the SAME MT couts (Cout = 32 layers).
// Load phase for chunk c (each wave handles the halo-tile slots of its own threads, p * 512 + tid):
//     a. weight DMA for the next multiply phase;  chunk c's maximum (posted one load phase earlier by BOTH groups) -> running shift
//        (conv_x3h.hip's hysteresis, same code);
//     b. split its pixels of chunk c: raw fp32 staging R (LDS) -> two fp16 planes P[c & 1];  fused bilinear x2: interpolate from the
//        low-resolution tile L[c & 1];
//     c. LDS-DMA the pixels of chunk c + 2 into R[c & 1] (free again) resp. L[(c+2) % 3]: two phases of flight time;
//     d. wait for everything but those (s_waitcnt vmcnt(NXL): every wave issues the same number of DMAs per chunk, void slot blocks land
//        in a sink), i.e. chunk c + 1's pixels and the weights; post chunk c + 1's maximum -> M[(c+1) & 1].
// Pixels travel global -> LDS by DMA (`buffer_load_dword ... lds`): no load is ever in flight in a REGISTER (the first form of this kernel
// prefetched into registers as conv_x3h.hip does, and hipcc moved those registers while the data was still on its way).  P, M and R are
// double-buffered by chunk parity (R is wave-private: a wave DMAs exactly the slots its threads read back), L is a ring of three (the
// other group still interpolates from chunk c's tile while this one stages chunk c + 2).
//
// Numerics are conv_x3h.hip's: per-workgroup, per-chunk power-of-two scaling, 14 matrix instructions per 8-channel chunk and tap set.
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "conv_stage.h"
#include "kernels.h"
#include "lds_dma.h"
#include "x3h_common.h"

namespace vr {

// (parked experiment: the tile descriptor of the library has no `pp` field)
struct X3ppTile { int pp; };

template <int MT, int TH, bool SPLITM, bool UP>
struct X3ppCfg {
    static constexpr int TW = 32, CK = 8, KK = 9;
    static constexpr int THT = SPLITM ? TH : 2 * TH;             // rows of the workgroup's pixel tile
    static constexpr int MTT = SPLITM ? 2 * MT : MT;             // couts of the workgroup's tile
    static constexpr int TH_in = THT + 2, PW = TW + 2;           // halo tile, pixels
    static constexpr int NSLOT = TH_in * PW;
    static constexpr int NPASS = (NSLOT + 511) / 512;            // thread tid owns slots p * 512 + tid
    static constexpr int WM = MT / 32, WN = TH / 4;
    static constexpr int PLANE = NSLOT * 16;                     // bytes of one fp16 plane (8 channels per pixel)
    static constexpr int P_BYTES = 2 * PLANE;
    static constexpr int NWP = KK * 2 * MT;                      // 16-byte weight operands per chunk and cout tile
    static constexpr int W_BYTES = NWP * 16;
    static constexpr int NWINS = NWP / 64;                       // 1 KB DMA wave-instructions per chunk and cout tile
    static constexpr int NWI = (NWINS + 3) / 4;                  // ... per wave of the four that issue them
    static constexpr int RS = (NSLOT + 63) / 64 * 64;            // raw staging: [8 ch][RS] fp32, slot-major inside a channel (LDS-DMA lands lane by lane)
    static constexpr int R_BYTES = 8 * RS * 4;
    static constexpr int LROWS = THT / 2 + 3, LW = 20, LSLOT = LROWS * LW;
    static constexpr int LP = (LSLOT + 63) / 64 * 64;            // low-resolution staging tile of the fused bilinear x2: [8 ch][LP] fp32
    static constexpr int L_BYTES = UP ? 8 * LP * 4 : 0;
    static constexpr int W_OFF = 2 * P_BYTES;                    // SPLITM: one private buffer per group; else two shared ones by chunk parity
    static constexpr int R_OFF = W_OFF + 2 * W_BYTES;
    static constexpr int L_OFF = R_OFF + 2 * R_BYTES;            // R: [parity][R_BYTES]; L: [chunk % 3][L_BYTES]
    static constexpr int E_OFF = L_OFF + 3 * L_BYTES;            // epilogue constants [group][4][MT]: bias, scale, shift, 1 / weight scale
    static constexpr int M_OFF = E_OFF + 4 * MTT * 4;            // [parity][8 waves] chunk maxima (uint bits of |x|)
    static constexpr int SINK_OFF = M_OFF + 64;                  // 256 bytes nobody reads: where the DMAs of void slot blocks land
    static constexpr int LDS_BYTES = SINK_OFF + 256;
    static constexpr int NXL = 8 * NPASS;                        // pixel DMAs a wave issues per chunk (void slot blocks land in the sink)
    static constexpr int NG = 14;                                // matrix-instruction groups per chunk: X0 X1 Y01 X2 X3 Y23 ... X8 Y8
    // waves per SIMD the register budget must allow (two or one workgroups per CU; 64-cout groups hold 64 accumulator registers per lane:
    // 224 bytes of scratch under a 128-register cap)
    static constexpr int WPS = (2 * LDS_BYTES <= 160 * 1024 && MT == 32) ? 4 : 2;
    static_assert(TH % 4 == 0 && MT % 32 == 0 && NWP % 64 == 0 && LDS_BYTES <= 160 * 1024 && LSLOT <= 512 && NPASS <= 3 && NXL + NWI < 64, "tile");
};

// ---- phase trace (diagnostics, VR_CONV_DBG bit 64): one workgroup in the middle of the grid stamps the cycle counter wave by wave at
// the start of each of its first 48 phases, at the end of the phase body and behind the barrier: [wave 8][phase 48][point 3] ----
__device__ long long g_x3pp_trace[8 * 48 * 3];
void x3pp_trace_read(long long* host) { VR_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x3pp_trace), sizeof(long long) * 8 * 48 * 3)); }

template <int MT, int TH, bool SPLITM, bool UP>
__global__ __launch_bounds__(512, (X3ppCfg<MT, TH, SPLITM, UP>::WPS)) void conv_x3pp_kernel(const ConvArgs a) {
    using Cfg = X3ppCfg<MT, TH, SPLITM, UP>;
    constexpr int TW = Cfg::TW, KK = Cfg::KK, PW = Cfg::PW, NSLOT = Cfg::NSLOT, NPASS = Cfg::NPASS, WM = Cfg::WM, WN = Cfg::WN,
                  PLANE = Cfg::PLANE, THT = Cfg::THT, MTT = Cfg::MTT, NWI = Cfg::NWI, RS = Cfg::RS, LP = Cfg::LP;
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int rr = id >> 3;
    const int ct = rr % a.nct;
    // every XCD walks its own contiguous, row-major range of pixel tiles (conv_x3h.hip: neighbours share halo lines in the XCD's L2)
    const int per_xcd = (a.npt + 7) >> 3;
    const int pt = (a.dbg & 16) ? (rr / a.nct) * 8 + xcd : xcd * per_xcd + rr / a.nct;
    if (pt >= a.npt) return;
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int n = pt / tiles_per_img;
    const int trem = pt - n * tiles_per_img;
    const int h0 = (trem / a.tiles_w) * THT;
    const int w0 = (trem % a.tiles_w) * TW;
    const int co0 = ct * MTT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;                     // group (0 = A multiplies in even phases), wave inside the group
    const int co0g = co0 + (SPLITM ? grp * MT : 0);               // this group's first cout
    const int rowg = SPLITM ? 0 : grp * TH;                        // ... first row inside the workgroup's pixel tile
    const int nchunk = (a.Cin + 7) >> 3;
    const unsigned lds0 = (unsigned)(size_t)smem_pp;
    const bool prio = (a.dbg & 32) != 0;                          // experiment: s_setprio 1 over the multiply phase
    const bool tracing = (a.dbg & 64) != 0 && blockIdx.x == gridDim.x / 2;
    int tphase = 0;
    auto stamp = [&](int point) __attribute__((always_inline)) {
        if (tracing && tphase < 48) {
            const long long t = __builtin_readcyclecounter();
            if (lane == 0) g_x3pp_trace[(wave * 48 + tphase) * 3 + point] = t;
        }
        if (point == 2) ++tphase;
    };

    // ---- this thread's pixels of the halo tile: byte offset in a channel plane = row * (4 * sH) + 4 * column (2^31: padding) ----
    auto pixel_offset = [&](int p, unsigned sH4) __attribute__((always_inline)) -> int {
        const int s = p * 512 + tid;
        const int r = s / PW, c = s - r * PW;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
        return ok ? (int)((unsigned)hi * sH4 + (unsigned)(wi * 4)) : (int)0x80000000u;
    };
    // ---- sources that arrive through the decoder's bilinear x2 (align_corners=True; eval: the upsample is not materialised):
    // this thread's low-resolution pixel of the staging tile; for each of its halo pixels the byte offset of the top-left neighbour
    // inside a channel plane of that tile, the (clamped) steps to the +1 row / +1 column neighbours and the two interpolation weights ----
    // (field by field: a reference selected among a.src[] makes hipcc keep the whole argument struct in scratch)
    const int usi = a.src[0].up ? 0 : (a.src[1].up ? 1 : 2);
    const struct { float rh, rw; int H, W; } us = {VR_SEL_F(a, usi, rh), VR_SEL_F(a, usi, rw), VR_SEL_F(a, usi, H), VR_SEL_F(a, usi, W)};
    int lrow = 0, lcol4 = 0;
    int lidx[NPASS], lstep[NPASS];                                 // (lstep: row step << 16 | column step, bytes)
    float lh[NPASS], lw_[NPASS];
    if (UP) {
        const int lr0 = (int)(us.rh * (float)(h0 > 0 ? h0 - 1 : 0)), lc0 = (int)(us.rw * (float)(w0 > 0 ? w0 - 1 : 0));
        const int lr = lr0 + tid / Cfg::LW, lc = lc0 + tid % Cfg::LW;
        const bool lok = tid < Cfg::LSLOT && lr < us.H && lc < us.W;
        lrow = lok ? lr : 0;
        lcol4 = lok ? lc * 4 : (int)0x80000000u;
        static_for<NPASS>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const int s = p * 512 + tid;
            const int r = s / PW, c = s - r * PW;
            const int hi = h0 - 1 + r, wi = w0 - 1 + c;
            const bool ok = s < NSLOT && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
            const float h1r = us.rh * (float)(ok ? hi : 0), w1r = us.rw * (float)(ok ? wi : 0);
            const int h1 = (int)h1r, w1 = (int)w1r;
            lidx[p] = ok ? ((h1 - lr0) * Cfg::LW + (w1 - lc0)) * 4 : -1;
            // the +1 neighbours exist inside the staging tile unless the pixel sits on the last low-resolution row / column, where
            // their weight is exactly 0: step 0 there, so that nothing outside the tile is ever read (0 * inf would poison the sum)
            const int rs = (h1 + 1 < us.H && h1 + 1 - lr0 < Cfg::LROWS) ? Cfg::LW * 4 : 0;
            const int cs = (w1 + 1 < us.W && w1 + 1 - lc0 < Cfg::LW) ? 4 : 0;
            lstep[p] = (rs << 16) | cs;
            lh[p] = h1r - (float)h1;
            lw_[p] = w1r - (float)w1;
        });
    }
    // ---- weight operands of a cout tile: LDS order [tap][plane][m], source x3w[chunk][(tap * 2 + plane) * CoutPad + co0g + m];
    // DMA wave-instruction j covers operands j * 64 + lane: the lane part is one vector offset, the j part rides on the scalar offset.
    //   SPLITM: each group owns ONE buffer and fills it in the load phase right before the multiply phase that reads it;
    //   else:   the groups share two buffers by chunk parity, filled by group B alone (W(c) during phase 2c-2: read by A in phase 2c and
    //           by B in 2c+1; the buffer's previous content, W(c-2), was last read by B in phase 2c-3) ----
    const unsigned wv0 = (unsigned)(((lane / MT) * a.CoutPad + (lane % MT)) * 16);
    const unsigned wjs = (unsigned)((64 / MT) * a.CoutPad * 16);
    const long long wchunk_bytes = (long long)KK * 2 * a.CoutPad * 16;
    auto issue_w = [&](int m, int jw) __attribute__((always_inline)) {     // chunk m; jw: this wave's place among the four (eight: prologue) issuing waves
        const bool livew = m < nchunk && co0g < a.CoutPad;
        const char* wb = static_cast<const char*>(a.x3w) + (livew ? m : 0) * wchunk_bytes + (long long)(livew ? co0g : 0) * 16;
        const i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), livew ? (unsigned)(wchunk_bytes - (long long)co0g * 16) : 0u);
        const unsigned ws_b = lds0 + (unsigned)(Cfg::W_OFF + (SPLITM ? grp : (m & 1)) * Cfg::W_BYTES);
#pragma unroll
        for (int i = 0; i < NWI; ++i) {                            // exactly NWI wave-instructions (the hand-placed waits count them)
            const int j0 = jw + 4 * i;
            const int j = j0 < Cfg::NWINS ? j0 : jw;               // (wave-uniform; a surplus one repeats the wave's first: same bytes, same place)
            dma16s((unsigned)__builtin_amdgcn_readfirstlane((int)(ws_b + (unsigned)j * 1024u)), wv0, wr,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)j * wjs)));
        }
    };
    // ---- pixels: LDS-DMA, one dword per lane and wave-instruction, straight into the raw staging R (plain sources) or the low-resolution
    // tile L (upsampled sources).  NOTHING is in flight in a register: round 6's first form prefetched into registers across the phases as
    // conv_x3h.hip does, and hipcc copied those registers around while their loads were still on the way (live-range splits across the
    // multiply phase, shuffles where the two groups' paths part) -- silently wrong on every large layer.  Every wave DMAs exactly the slots
    // its own threads read back (slot p * 512 + tid = block p * 8 + wave), so its own s_waitcnt vmcnt(0) is all the ordering needed.
    // The channels are visited strictly in order (chunk by chunk): the source of the virtual concat is a running scalar state. ----
    const float* xp = a.src[0].p + (long long)n * a.src[0].sN;
    long long xsC = a.src[0].sC;
    unsigned xsH4 = (unsigned)a.src[0].sH * 4u;
    int xend = a.c1, xsi = 0;
    bool xup = a.src[0].up != 0;
    unsigned upm[2] = {0u, 0u};                                    // per chunk parity: which of the 8 channels are upsampled sources
    int xvo[NPASS];
    static_for<NPASS>([&](auto pc) __attribute__((always_inline)) { xvo[decltype(pc)::value] = pixel_offset(decltype(pc)::value, xsH4); });
    auto next_source = [&]() __attribute__((always_inline)) {
        ++xsi;
        if (xsi == 1) { xp = a.src[1].p + (long long)n * a.src[1].sN; xsC = a.src[1].sC; xsH4 = (unsigned)a.src[1].sH * 4u; xend = a.c2; xup = a.src[1].up != 0; }
        else { xp = a.src[2].p + (long long)n * a.src[2].sN; xsC = a.src[2].sC; xsH4 = (unsigned)a.src[2].sH * 4u; xend = 1 << 30; xup = a.src[2].up != 0; }
        static_for<NPASS>([&](auto pc) __attribute__((always_inline)) { xvo[decltype(pc)::value] = pixel_offset(decltype(pc)::value, xsH4); });
    };
    const unsigned sink = lds0 + (unsigned)Cfg::SINK_OFF;
    auto dma_channel = [&](int k, int cl, auto par, unsigned lofs) __attribute__((always_inline)) {     // lofs = (k % 3) * L_BYTES
        constexpr int PAR = decltype(par)::value;
        const int ci = k * 8 + cl;                                // wave-uniform
        const bool live = ci < a.Cin;
        if (live && ci >= xend) next_source();                    // (a source may be a single channel: two steps at most)
        if (live && ci >= xend) next_source();
        const bool up = UP && live && xup;
        if (cl == 0) upm[PAR] = 0u;
        upm[PAR] |= (up ? 1u : 0u) << cl;
        // channels beyond Cin read zeros through an empty descriptor (R must hold zeros for them)
        const i32x4 xs = make_rsrc(xp, live ? 0x7FFFFFF0u : 0u);
        {
            // block `wave` of pass 0: this thread's pixel, or -- an upsampled source -- its ONE low-resolution pixel of the staging tile
            const unsigned dR = lds0 + (unsigned)(Cfg::R_OFF + PAR * Cfg::R_BYTES + cl * RS * 4) + (unsigned)wave * 256u;
            const unsigned dL = lds0 + (unsigned)(Cfg::L_OFF + cl * LP * 4) + lofs + (unsigned)wave * 256u;
            const unsigned d = up ? (wave * 64 < LP ? dL : sink) : (wave * 64 < RS ? dR : sink);
            dma4((unsigned)__builtin_amdgcn_readfirstlane((int)d), (unsigned)(up ? (int)((unsigned)lrow * xsH4) + lcol4 : xvo[0]), xs);
        }
        const i32x4 xs1 = make_rsrc(xp, (live && !up) ? 0x7FFFFFF0u : 0u);
        static_for<NPASS - 1>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value + 1;
            const bool ok = !up && p * 512 + wave * 64 < RS;       // (wave-uniform; a void block still issues its DMA: the counts stay equal)
            const unsigned dR = lds0 + (unsigned)(Cfg::R_OFF + PAR * Cfg::R_BYTES + cl * RS * 4 + p * 2048) + (unsigned)wave * 256u;
            dma4((unsigned)__builtin_amdgcn_readfirstlane((int)(ok ? dR : sink)), (unsigned)xvo[p], xs1);
        });
        if (live) xp += xsC;
    };
    auto dma_chunk = [&](int k, auto par) __attribute__((always_inline)) {                       // exactly NXL wave-instructions
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of the staging it is about to overwrite have returned
        const unsigned lofs = (unsigned)((k % 3) * Cfg::L_BYTES);
        static_for<8>([&](auto cc) __attribute__((always_inline)) { dma_channel(k, decltype(cc)::value, par, lofs); });
    };
    // this thread's raw value of channel cl, pass p (valid blocks only); upsampled channels: its low-resolution pixel
    auto raw_at = [&](int par, int cl, int p, int tq) __attribute__((always_inline)) -> float {
        return *reinterpret_cast<const float*>(smem_pp + Cfg::R_OFF + par * Cfg::R_BYTES + cl * RS * 4 + (p * 512 + tq) * 4);
    };
    // max |x| over this thread's staged pixels of the chunk of parity PAR (landed: behind vmcnt(0)) -> wave maximum -> M[PAR][wave]
    auto post_max = [&](auto par, int k) __attribute__((always_inline)) {                         // chunk k, of parity PAR
        constexpr int PAR = decltype(par)::value;
        const int lofs = (k % 3) * Cfg::L_BYTES;
        int tq = tid;
        asm volatile("" : "+v"(tq));
        float m = 0.f;
        static_for<8>([&](auto cc) __attribute__((always_inline)) {
            constexpr int cl = decltype(cc)::value;
            if (UP && ((upm[PAR] >> cl) & 1u)) {
                if (wave * 64 < LP)
                    m = fmaxf(m, fabsf(*reinterpret_cast<const float*>(smem_pp + Cfg::L_OFF + lofs + cl * LP * 4 + tq * 4)));
            } else {
                static_for<NPASS>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int p = decltype(pc)::value;
                    if (p * 512 + wave * 64 < RS) m = fmaxf(m, fabsf(raw_at(PAR, cl, p, tq)));
                });
            }
        });
        int b = __float_as_int(m);                                 // non-negative floats order like their bit patterns
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true));   // row_half_mirror
        b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true));   // row_mirror
        const int w = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
                          max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
        if (lane == 0) reinterpret_cast<int*>(smem_pp + Cfg::M_OFF)[PAR * 8 + wave] = w;
    };
    auto read_max_exp = [&](auto par) __attribute__((always_inline)) -> int {                    // biased exponent of the largest |x| of the chunk (255: inf / nan)
        constexpr int PAR = decltype(par)::value;
        const vr_i32x4 m0 = *reinterpret_cast<const vr_i32x4*>(smem_pp + Cfg::M_OFF + PAR * 32);
        const vr_i32x4 m1 = *reinterpret_cast<const vr_i32x4*>(smem_pp + Cfg::M_OFF + PAR * 32 + 16);
        const int w = max(max(max(m0[0], m0[1]), max(m0[2], m0[3])), max(max(m1[0], m1[1]), max(m1[2], m1[3])));
        return __builtin_amdgcn_readfirstlane(w) >> 23;
    };
    // ---- the running power-of-two shift of the pixels (conv_x3h.hip header): x' = x * 2^sh.  Every wave runs the same state machine
    // over the same chunk maxima; the accumulators follow at the start of the multiply phase of the chunk the shift belongs to ----
    int sh = 0, shlo = 0;
    int shc[2] = {0, 0};                                           // shift the chunk of each parity was split with
    float psc = 1.f;
    auto follow = [&](int e, bool first) __attribute__((always_inline)) {
        const int need = 140 - (e < 14 ? 14 : e);                  // chunk maximum -> [2^13, 2^14)
        shlo = (first || need < shlo) ? need : shlo;               // never more than 2^64 ahead of the largest chunk seen so far
        int nsh = sh;
        if (first || need < sh - 1) nsh = need;
        else if (need > sh + 12) {
            nsh = need < sh + 64 ? need : sh + 64;
            nsh = nsh < shlo + 64 ? nsh : shlo + 64;
            nsh = nsh > sh ? nsh : sh;
        }
        sh = nsh;
        psc = x3h_pow2(sh);                                        // sh in [-115, 126]
    };
    // split this thread's pixel of pass p of the chunk staged in R / L[PAR] -> P[PAR]
    auto convert_pass = [&](auto par, auto pc, int lofs) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value, p = decltype(pc)::value;
        if (p * 512 + wave * 64 >= RS) return;                     // (wave-uniform: a void slot block)
        char* const Pw = smem_pp + PAR * Cfg::P_BYTES;
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int s = p * 512 + tq;
        float x[8];
        if (!UP || upm[PAR] != 0xFFu)
            static_for<8>([&](auto cc) __attribute__((always_inline)) { x[decltype(cc)::value] = raw_at(PAR, decltype(cc)::value, p, tq); });
        if (UP && upm[PAR] != 0u) {
            // torch's bilinear, align_corners=True (pointwise.hip: upsample2x_kernel)
            // (opaque copies: hipcc otherwise hoists the neighbour addresses and the complementary weights of every pass and both parities
            // out of the chunk loop -- ~40 registers held across the multiply phases for a handful of VALU instructions)
            int li = lidx[p], ls = lstep[p];
            float h1l = lh[p], w1l = lw_[p];
            asm volatile("" : "+v"(li), "+v"(ls), "+v"(h1l), "+v"(w1l));
            const char* lq = smem_pp + Cfg::L_OFF + lofs + (li >= 0 ? li : 0);
            const int rs = ls >> 16, cs = ls & 0xffff;
            const float h0l = 1.f - h1l, w0l = 1.f - w1l;
            static_for<8>([&](auto cc) __attribute__((always_inline)) {
                constexpr int cl = decltype(cc)::value;
                if ((upm[PAR] >> cl) & 1u) {                       // (wave-uniform)
                    const char* q = lq + cl * LP * 4;
                    const float v00 = *reinterpret_cast<const float*>(q), v01 = *reinterpret_cast<const float*>(q + cs),
                                v10 = *reinterpret_cast<const float*>(q + rs), v11 = *reinterpret_cast<const float*>(q + rs + cs);
                    const float v = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
                    x[cl] = li >= 0 ? v : 0.f;
                }
            });
        }
        if (s < NSLOT) {
            vr_i32x4 ph, pl;
            static_for<4>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                int h, l;
                split2h_pair(x[2 * j], x[2 * j + 1], psc, h, l);
                ph[j] = h; pl[j] = l;
            });
            char* q = Pw + s * 16;
            *reinterpret_cast<vr_i32x4*>(q) = ph;
            *reinterpret_cast<vr_i32x4*>(q + PLANE) = pl;
        }
    };
    auto convert = [&](auto par, int k) __attribute__((always_inline)) {                          // chunk k, of parity PAR
        const int lofs = (k % 3) * Cfg::L_BYTES;
        // (scheduling fences between the passes keep the staging reads of one pass from being hoisted over the previous pass)
        convert_pass(par, std::integral_constant<int, 0>{}, lofs);
        if constexpr (NPASS > 1) { __builtin_amdgcn_sched_barrier(0); convert_pass(par, std::integral_constant<int, 1>{}, lofs); }
        if constexpr (NPASS > 2) { __builtin_amdgcn_sched_barrier(0); convert_pass(par, std::integral_constant<int, 2>{}, lofs); }
    };

    const int khalf = lane >> 5, l31 = lane & 31;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    int acc_sh = 0;                                               // the shift the accumulators are in

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // ---------------- prologue: both groups stage chunk 0, build P(0) together, stage chunk 1; W(0) arrives -------------------------
    if (SPLITM) issue_w(0, gw);
    else {                                                         // shared weights: all eight waves fetch W(0)
#pragma unroll
        for (int i = 0; i < (Cfg::NWINS + 7) / 8; ++i) {
            const int j = wave + 8 * i;
            const char* wb = static_cast<const char*>(a.x3w) + (long long)co0g * 16;
            const i32x4 wr = make_rsrc(reinterpret_cast<const float*>(wb), co0g < a.CoutPad ? (unsigned)(wchunk_bytes - (long long)co0g * 16) : 0u);
            if (j < Cfg::NWINS)
                dma16s((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)Cfg::W_OFF + (unsigned)j * 1024u)), wv0, wr,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)j * wjs)));
        }
    }
    dma_chunk(0, P0{});
    dma_chunk(1, P1{});
    // epilogue constants of the workgroup's cout tile (bias; eval: folded BatchNorm scale / shift; 1 / weight scale): register loads in
    // straight-line code right in front of the wait that names their registers, parked in LDS
    float ecv[4];
    {
        const int ec = co0 + (tid & (MTT - 1));
        const int ecc = ec < a.Cout ? ec : a.Cout - 1;
        const int ecp = ec < a.CoutPad ? ec : a.CoutPad - 1;
        const i32x4 rb = make_rsrc(a.bias, a.bias ? 0x7FFFFFF0u : 0u);
        const i32x4 re = make_rsrc(a.epi, a.epi ? 0x7FFFFFF0u : 0u);
        const i32x4 rw = make_rsrc(reinterpret_cast<const float*>(static_cast<const char*>(a.x3w) + nchunk * wchunk_bytes), 0x7FFFFFF0u);
        ecv[0] = x3h_load(rb, ecc * 4);
        ecv[1] = x3h_load(re, ecc * 8);
        ecv[2] = x3h_load(re, ecc * 8 + 4);
        ecv[3] = x3h_load(rw, ecp * 4);
        asm volatile("s_waitcnt vmcnt(0) ; landed %0 %1 %2 %3" : "+v"(ecv[0]), "+v"(ecv[1]), "+v"(ecv[2]), "+v"(ecv[3]) :: "memory");
    }
    if (tid < MTT) {
        float* E = reinterpret_cast<float*>(smem_pp + Cfg::E_OFF) + (tid / MT) * 4 * MT + (tid % MT);
        E[0] = ecv[0];
        E[MT] = a.epi ? ecv[1] : 1.f;
        E[2 * MT] = a.epi ? ecv[2] : 0.f;
        E[3 * MT] = ecv[3];
    }

    post_max(P0{}, 0);
    lds_barrier();
    follow(read_max_exp(P0{}), true);
    shc[0] = sh;
    convert(P0{}, 0);
    dma_chunk(2, P0{});
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::NXL) : "memory");      // chunk 1 has landed; chunk 2 stays in flight
    post_max(P1{}, 1);
    lds_barrier();

    // ---------------- the two phase bodies ---------------------------------------------------------------------------------------
    // multiply P[PAR] x W: ds_read_b128 + v_mfma only
    auto multiply = [&](auto par) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        if (acc_sh != shc[PAR]) {                                  // the shift moved (rare on real activations): the sums follow
            const int d = shc[PAR] - acc_sh;                       // <= 64 (first chunk: <= 126, on zeros); a large negative d flushes the old sums
            const float f = d < -126 ? 0.f : x3h_pow2(d);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= f;
            acc_sh = shc[PAR];
        }
        // (the lane's operand offsets are rebuilt from an opaque copy of the thread index in every multiply phase: left to itself hipcc keeps
        // every derived address of both parities in registers across the whole kernel)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int khalf = (tq >> 5) & 1, l31 = tq & 31;
        // B operands: pixel (row rowg + gw*WN + ni + ty, col l31 + tx) of plane 0 is at bq + ((ni + ty) * PW + tx) * 16.
        //   X(t): lanes 0-31 plane 0 (b1), lanes 32-63 plane 1 (b2) of tap t's pixel;   Y(t,t+1): plane 0, lanes 32-63 at tap t+1's pixel,
        //   which lies one pixel to the right (t = 0, 4, 6) or PW - 2 pixels on (t = 2: from (0,2) to (1,0))
        const int bq = ((rowg + gw * WN) * PW + l31) * 16;
        const int bX = bq + khalf * PLANE, bY1 = bq + khalf * 16, bY2 = bq + khalf * (PW - 2) * 16;
        // A operands, LDS order [tap][plane][m]:  X(t): a1(t) in both halves;  Y(t,t+1): a2(t) | a2(t+1);  Y(8): a2(8) | 0
        const int aX = l31 * 16, aY = (MT + l31 + khalf * 2 * MT) * 16;
        const char* Pb = smem_pp + PAR * Cfg::P_BYTES;
        const char* Wb = smem_pp + Cfg::W_OFF + (SPLITM ? grp : PAR) * Cfg::W_BYTES;
        vr_f16x8 A[2][WM], B[2][WN];
        // group g of the 14: g = 3q + {0, 1} -> X(2q), X(2q + 1); g = 3q + 2 -> Y(2q, 2q + 1); g = 12 -> X(8); g = 13 -> Y(8)
        auto read_group = [&](int g, int buf) __attribute__((always_inline)) {
            const bool isY = g == 13 || (g < 12 && g % 3 == 2);
            const int t = g >= 12 ? 8 : 2 * (g / 3) + (g % 3 == 1 ? 1 : 0);
            const int ty = t / 3, tx = t % 3;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const char* q = Wb + (t * 2 * MT + mi * 32) * 16;
                if (!isY) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aX);
                else if (t < 8) A[buf][mi] = *reinterpret_cast<const vr_f16x8*>(q + aY);
                else {
                    const vr_i32x4 v = *reinterpret_cast<const vr_i32x4*>(q + aX + MT * 16);
                    vr_i32x4 z;
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = khalf ? 0 : v[j];
                    A[buf][mi] = __builtin_bit_cast(vr_f16x8, z);
                }
            }
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int o = ((ni + ty) * PW + tx) * 16;
                if (!isY) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bX + o);
                else if (t == 8) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bq + o);     // (upper half meets zeros)
                else if (t == 2) B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bY2 + o);
                else B[buf][ni] = *reinterpret_cast<const vr_f16x8*>(Pb + bY1 + o);
            }
        };
        read_group(0, 0);
        if (prio) __builtin_amdgcn_s_setprio(1);
        static_for<Cfg::NG>([&](auto gc) __attribute__((always_inline)) {                         // (a type-indexed loop: left rolled, A[cur] / B[cur] would live in scratch)
            constexpr int g = decltype(gc)::value, cur = g & 1;
            if constexpr (g + 1 < Cfg::NG) read_group(g + 1, cur ^ 1);   // the operand reads of group g+1 go out in front of the matrix instructions of group g
            __builtin_amdgcn_sched_barrier(0);
            static_for<WM * WN>([&](auto ic) __attribute__((always_inline)) {
                constexpr int mi = decltype(ic)::value / WN, ni = decltype(ic)::value % WN;
                acc[mi][ni] = mfma_f16x16(A[cur][mi], B[cur][ni], acc[mi][ni]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // the phase of the group that does NOT multiply: weights for its next multiply phase (chunk m; A: c, B: c - 1), then -- while chunks
    // remain -- the load phase for chunk c (parity PAR), whose raw pixels this group staged in its previous load phase
    auto load_phase = [&](int c, int m, auto par) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        using Q = std::integral_constant<int, PAR ^ 1>;
        if (SPLITM) issue_w(m, gw);                                // private buffer: free since this group's last multiply phase
        else if (grp != 0) issue_w(c, gw);                         // shared buffers: group B fetches W(c) two phases ahead of its first reader
        if (c < nchunk) {
            follow(read_max_exp(par), false);
            shc[PAR] = sh;
            convert(par, c);
            dma_chunk(c + 2, par);                                 // R[PAR] is free again: chunk c+2 (two phases of flight time)
            // outstanding, oldest first: chunk c+1's pixels | the weights issued above | chunk c+2's pixels
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::NXL) : "memory");      // everything but chunk c+2 has landed
            post_max(Q{}, c + 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto bar = [&]() __attribute__((always_inline)) { lds_barrier(); };

    // ONE loop for both groups, the roles chosen by wave-uniform branches around the phase bodies.  No barrier behind the last phase:
    // group A walks into its epilogue beside group B's last multiply.
    for (int k = 0;; k += 2) {
        stamp(0);
        if (grp == 0) multiply(P0{});
        else load_phase(k + 1, k, P1{});
        stamp(1);
        bar();
        stamp(2);
        stamp(0);
        if (grp != 0) multiply(P0{});
        else if (k + 1 < nchunk) load_phase(k + 1, k + 1, P1{});
        stamp(1);
        if (k + 1 >= nchunk) break;
        bar();
        stamp(2);
        stamp(0);
        if (grp == 0) multiply(P1{});
        else load_phase(k + 2, k + 1, P0{});
        stamp(1);
        bar();
        stamp(2);
        stamp(0);
        if (grp != 0) multiply(P1{});
        else if (k + 2 < nchunk) load_phase(k + 2, k + 2, P0{});
        stamp(1);
        if (k + 2 >= nchunk) break;
        bar();
        stamp(2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---------------- epilogue (conv_epilogue.h): bias, (eval) BatchNorm + activation, up to three destination segments ---------
    // (group A gets here one phase early: its stores run beside group B's last multiply phase)
    const float* Eg = reinterpret_cast<const float*>(smem_pp + Cfg::E_OFF) + (SPLITM ? grp * 4 * MT : 0);
    {
        // undo the two scalings: 2^-sh (pixels, this workgroup) and 1 / weight scale (per cout: E[3][m], row m = mi*32 + (r&3) + 8*(r>>2) + 4*khalf)
        const float fo = x3h_pow2(-acc_sh);
        const float* Wi = Eg + 3 * MT;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const vr_f32x4h wi = *reinterpret_cast<const vr_f32x4h*>(Wi + mi * 32 + 8 * rq + 4 * khalf);
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mi][ni][rq * 4 + j] = (acc[mi][ni][rq * 4 + j] * fo) * wi[j];
            }
    }
    {
        int hon[WN], won[WN];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) { hon[ni] = h0 + rowg + gw * WN + ni; won[ni] = w0 + l31; }
        epi_store<MT, WM, WN>(VR_EPI_ARGS(a), acc, Eg, n, co0g, khalf, h0 + rowg + TH <= a.Hout && w0 + TW <= a.Wout, hon, won);
    }
    // ---------------- BatchNorm partial statistics (training): one row per WORKGROUP pixel tile ---------------------------------
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_pp);                      // [8 waves][MT][2]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const int ho = h0 + rowg + gw * WN + ni, wo = w0 + l31;
                    if (ho < a.Hout && wo < a.Wout) {
                        const float v = acc[mi][ni][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
                s1 = half_wave_sum_dpp(s1);
                s2 = half_wave_sum_dpp(s2);
                if (l31 == 16) {                                   // (the sums are complete in lanes 16-31 / 48-63)
                    const int m = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    red[(wave * MT + m) * 2 + 0] = s1;
                    red[(wave * MT + m) * 2 + 1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < MTT) {
            const int m = SPLITM ? tid % MT : tid, w0r = SPLITM ? (tid / MT) * 4 : 0;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < (SPLITM ? 4 : 8); ++w) {
                s1 += red[((w0r + w) * MT + m) * 2 + 0];
                s2 += red[((w0r + w) * MT + m) * 2 + 1];
            }
            const int co = co0 + tid;
            if (co < a.Cout) {
                a.part[((long long)pt * a.Cout + co) * 2 + 0] = s1;
                a.part[((long long)pt * a.Cout + co) * 2 + 1] = s2;
            }
        }
    }
}

template <int MT, int TH, bool SPLITM, bool UP>
static void x3pp_launch_up(const ConvArgs& a, hipStream_t st) {
    using Cfg = X3ppCfg<MT, TH, SPLITM, UP>;
    auto kern = conv_x3pp_kernel<MT, TH, SPLITM, UP>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    const int groups = (a.npt + 7) / 8;
    VR_LAUNCH(kern, dim3(groups * 8 * a.nct), dim3(512), Cfg::LDS_BYTES, st, a);
    VR_HIP(hipGetLastError());
}

template <int MT, int TH, bool SPLITM>
static void x3pp_launch(const ConvArgs& a, hipStream_t st) {
    if (a.src[0].up | a.src[1].up | a.src[2].up) x3pp_launch_up<MT, TH, SPLITM, true>(a, st);
    else x3pp_launch_up<MT, TH, SPLITM, false>(a, st);
}

// tile shapes of the ping-pong kernel (X3Tile::pp): 1 = 128 couts x 8 rows, 2 = 64 couts x 16 rows, 3 = 64 couts x 8 rows,
// 5 = 32 couts x 16 rows (all x 32 columns; 4 = 32 couts x 32 rows does not fit the LDS with the raw staging double-buffered)
void x3pp_tile_dims(int pp, int* couts, int* rows) {
    static const int c[6] = {0, 128, 64, 64, 32, 32}, r[6] = {0, 8, 16, 8, 32, 16};
    *couts = c[pp]; *rows = r[pp];
}
void x3pp_launch_conv(const ConvArgs& a, const X3ppTile& t, hipStream_t st) {
    switch (t.pp) {
        case 1: x3pp_launch<64, 8, true>(a, st); break;
        case 2: x3pp_launch<64, 8, false>(a, st); break;
        case 3: x3pp_launch<32, 8, true>(a, st); break;
        case 5: x3pp_launch<32, 8, false>(a, st); break;
        default: throw Error(-2, "conv_x3pp: unknown tile shape");
    }
}

}  // namespace vr
