// fp16-path convolution as a one-workgroup-per-CU LDS-DMA pipeline (VC_CFG_DMA; stride 1, half-precision input).
//
// Why a different kernel: on the fp16 path a k-step is 16x shorter than on the fp32 path, and in conv_mfma_kernel the
// phases of a workgroup (stage a chunk through registers / contract it with weight fragments fetched from L2 by every
// wave / store) ADD UP instead of overlapping (0.24 + 0.56 + 0.27 ms for 3x3 128->128 x4).  Here
//   * ONE persistent 512-thread workgroup per CU walks a list of 16 x 32-pixel output tiles (2 waves per SIMD);
//   * BOTH operands reach LDS by global_load_lds_dwordx4 (no VGPR staging, no per-wave weight loads from L2):
//       A: the tile's input footprint, one 32-channel chunk at a time, double-buffered -- the next chunk streams in
//          while the current one is contracted; image borders read a zero page, so the image needs no select pass;
//       B: the weights of one phase (8 fragments of 1 KiB = 16 MFMAs per wave) per ring slot, RING slots, fetched
//          RING-2 phases ahead; every byte of B enters the CU once per tile instead of once per wave;
//   * the loads are retired with COUNTED s_waitcnt vmcnt(N) across raw s_barriers (cdna guide, "Pipelining across
//     barriers"); N is a compile-time function of the phase (schedule tables in DmaCfg);
//   * the two waves of a SIMD run the phases half a phase apart (waves 0-3 contract while waves 4-7 read their
//     fragments from LDS and vice versa): the matrix pipe always has a wave whose operands are already in registers;
//   * LDS images are lane-linear (what the DMA writes); the bank-conflict-free layout of the A image comes from
//     permuting the per-lane SOURCE addresses (16-byte channel groups XOR-ed with bits of the column).
// Accumulation order per output = (32-channel chunk, tap, k-step), the order of the classic fp16 instances, and the
// epilogue arithmetic is theirs too: results are bit-identical (tests/test_ops_gpu.py).
#pragma once
#include "conv_mfma.h"

// zero page: source of halo pixels outside the image and of padding fragments; dump: where the stores of pixels outside
// the output go, so that every wave issues the same number of stores per tile (they are counted by the vmcnt waits)
static __device__ __attribute__((aligned(1024))) unsigned char g_vc_dma_zero[2048];
static __device__ __attribute__((aligned(1024))) unsigned char g_vc_dma_dump[1024];

// KO_: diagnostic knock-out mask (tools/dma_check.py --ko; results are garbage by design, the time that remains is what the
// other parts cost): 1 no epilogue, 2 no vmcnt waits, 4 no MFMAs, 8 no fragment reads, 16 no DMA, 32 no stagger, 64 cycle stamps, 128 no s_setprio around the MFMAs, 256 DMA issue BEFORE the fragment reads,
// 512 M0 not restored, 1024 no A DMA, 2048 no B DMA, 4096 two A pieces per phase, 8192 weights by 64-bit lane addresses,
// 16384 no fast epilogue.  Shipped instances use 0.
// TAIL_: the bottleneck-block instance (ICIP2024/src/model/elic.py:69-83: ... -> conv3x3 -> ReLU -> conv1x1, + identity): the
// workgroup also holds the trailing 1x1 layer's weights in LDS and applies it to its accumulators before anything is stored
// (dma_epilogue_tail).  Every wave then owns ALL output channels of its rows (WAVES_N = 1): the 1x1 contracts over them.
// F32_: the EXACT fp32 path on the same pipeline (fp32 tensors, v_mfma_f32_32x32x2_f32): a chunk is 16 channels -- the same
// 64 bytes of a pixel, the same 1 KiB weight fragments ([lane][4 floats] = four k-sub-steps), the same addressing -- but every
// fragment pair feeds four 64-cycle MFMAs instead of one 32-cycle one: the operand streams that bound the fp16 instances
// disappear behind 8x more matrix time.  Accumulation order (16-channel chunk, tap, k-step, sub-step) = the classic fp32
// 7x7 / 5x5 instances' (CK = 16): bit-identical to them.
template <int KH_, int KW_, int NCHUNK_, int NT_, int RING_, int KO_ = 0, bool TAIL_ = false, bool F32_ = false> struct DmaCfg {
    static constexpr int KO = KO_;
    static constexpr bool TAIL = TAIL_, F32 = F32_;
    static constexpr int ESZ = F32_ ? 4 : 2;                   // bytes per input element; a chunk = 64 / ESZ channels
    static constexpr int KH = KH_, KW = KW_, TAPS = KH_ * KW_, NCHUNK = NCHUNK_, NT = NT_, RING = RING_;
    static constexpr int MT = 32, TH = 16, XT = 1, TW = 32;
    static constexpr int WAVES = 8, WAVES_N = (NT_ == 4 && !TAIL_) ? 2 : 1, WAVES_M = WAVES / WAVES_N;
    static constexpr int WM = TH / WAVES_M, WN = NT_ / WAVES_N;
    static constexpr int BN = NT_ * 32;
    // a phase = 8 weight fragments (8 KiB, one DMA instruction per wave) = UPP (chunk, tap) units of 2 k-steps
    static constexpr int UPP = 4 / NT_;
    // (phases per tile: rounded up to a multiple of the ring so that a phase's slot is a compile-time constant; the units
    //  past UT read zero fragments and are skipped by the contraction)
    static constexpr int UT = NCHUNK_ * TAPS, PT = ((UT + UPP - 1) / UPP + RING_ - 1) / RING_ * RING_;
    static constexpr int ROWS_IN = TH + KH_ - 1, COLS = TW + KW_ - 1, PIX = ROWS_IN * COLS;
    static constexpr int NA = (PIX * 4 + 511) / 512;          // DMA instructions per wave and chunk image (512 lanes x 16 B)
    static constexpr int A_BYTES = NA * 8192;
    static constexpr int B_OFF = 2 * A_BYTES, B_BYTES = RING_ * 8192;
    // TAIL: the 1x1 layer's NT x (2 NT) fragments of 1 KiB ([n-tile][k-step], vc_conv_pack_tail_f16) and its bias
    static constexpr int TAIL_OFF = B_OFF + B_BYTES, TAIL_BYTES = TAIL_ ? NT_ * 2 * NT_ * 1024 + BN * 4 : 0;
    static constexpr int BIAS_OFF = TAIL_OFF + TAIL_BYTES;
    static constexpr int LDS_FIXED = BIAS_OFF;                // + 4 * Cout (padded) at launch
    static_assert(!TAIL_ || ((NT_ == 4 || NT_ == 2) && NCHUNK_ == NT_ && KH_ == 3), "the fused tail: 3x3 C -> C followed by 1x1 C -> C, C = 64 or 128");
    // stores per wave and tile the counted waits may rely on: the fast epilogue issues WM * WN * 2 (16 bytes per lane),
    // the general one twice as many (a wait that counts too few younger operations only waits a little longer)
    static constexpr int NST = WM * WN * 2;
    static_assert(NT_ == 1 || NT_ == 2 || NT_ == 4, "1, 2 or 4 N-tiles of 32 channels per workgroup");
    static_assert(PT % RING_ == 0, "ring slot of a phase must be a compile-time constant");
    static_assert(8 * VC_EPI_SCRATCH_FLOATS * 4 <= A_BYTES, "epilogue scratch lives in the chunk buffer that has just been finished");

    static constexpr int pf(int c) { return c * TAPS / UPP; }                  // first / last phase reading chunk c
    static constexpr int pl(int c) { return ((c + 1) * TAPS - 1) / UPP; }
    // While chunk c is contracted, chunk c+1 (or chunk 0 of the next tile) streams into the other buffer.  Its pieces
    // may start two phases after the last read of that buffer and must be OLDER than the weight load whose wait
    // publishes the chunk's first phase.
    static constexpr int wlo(int c) { return c == 0 ? 1 : pl(c - 1) + 2; }
    static constexpr int whi(int c) { return (c + 1 == NCHUNK ? PT : pf(c + 1)) - RING + 2; }
    static constexpr int ppp(int c)          // pieces per phase: as few as the window allows (they come from HBM; a burst blocks the issuing waves)
    {
        const int room = whi(c) - wlo(c) + 1;
        const int need = room > 0 ? (NA + room - 1) / room : NA;
        return (KO & 4096) ? (need > 2 ? need : 2) : need;
    }
    static constexpr int piece_phase(int c, int k) { return wlo(c) + k / ppp(c); }
    static constexpr bool windows_ok()
    {
        for (int c = 0; c < NCHUNK; ++c)
            if (piece_phase(c, NA - 1) > whi(c) || wlo(c) < 0) return false;
        return true;
    }
    static_assert(windows_ok(), "the chunk image does not fit its issue window");
    static constexpr int nA(int p)          // A pieces issued in phase p (before the phase's weight load)
    {
        int n = 0;
        for (int c = 0; c < NCHUNK; ++c)
            for (int k = 0; k < NA; ++k) n += piece_phase(c, k) == p;
        return n;
    }
    // vmcnt for the wait of phase p: the weights of phase p+1 were issued in phase p-RING+3; everything issued after
    // them may still be in flight (A pieces are issued before the weight load of their phase; the NST stores of the
    // previous tile's epilogue lie between its last phase and phase 0).
    static constexpr int nwait(int p)
    {
        int n = 0;
        for (int j = p - RING + 4; j <= p; ++j) n += nA((j + PT) % PT) + 1;
        if (p - RING + 3 < 0) n += NST;
        return n;
    }
    static constexpr bool waits_ok()
    {
        for (int p = 0; p < PT; ++p)
            if (nwait(p) > 63) return false;
        return true;
    }
    static_assert(waits_ok(), "vmcnt is a 6-bit counter");
};

#define VC_DMA_FENCE() asm volatile("" ::: "memory")
#if defined(__HIP_DEVICE_COMPILE__)
#define VC_DMA_KEEP(x) asm volatile("" ::"v"(x))      // diagnostic builds: keeps a value alive whose consumer was knocked out
#else
#define VC_DMA_KEEP(x) ((void)(x))
#endif
#define VC_DMA_BARRIER()                   \
    do {                                   \
        VC_DMA_FENCE();                    \
        __builtin_amdgcn_s_barrier();      \
        VC_DMA_FENCE();                    \
    } while (0)

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to lds_byte_addr + 16 * lane
// (M0 is compiler-managed: the restoring form saves and restores it inside one statement -- cdna guide 5.7 -- the
// non-restoring forms declare it clobbered)
template <bool RESTORE_M0 = true> __device__ __forceinline__ void vc_glds16(const void *src, unsigned lds_byte_addr)
{
    if constexpr (RESTORE_M0) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_byte_addr)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
    }
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the address operand the issuing
// wave has to move is half as wide, which halves the issue cost next to a SIMD partner that is issuing MFMAs
// (tools/micro/glds_rate.hip: 31 vs 77 cycles)
__device__ __forceinline__ void vc_glds16_sbase(const void *uniform_base, unsigned lane_off, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(uniform_base), "s"(lds_byte_addr)
                 : "memory", "m0");
}
template <int N> __device__ __forceinline__ void vc_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef VC_DMA_DIAG
// diagnostic build only: shader-clock totals per segment, summed over all waves (KO bit 64; vc_debug_dma_stamps reads them)
__device__ unsigned long long g_vc_dma_stamps[8];
#define VC_DMA_STAMP(var)                                                                             \
    unsigned var = 0;                                                                                 \
    if constexpr (C::KO & 64) {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t__;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                   \
        var = (unsigned)t__;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define VC_DMA_ACC(slot, t1, t0) \
    if constexpr (C::KO & 64) st_sum[slot] += (t1) - (t0)
#else
#define VC_DMA_STAMP(var)
#define VC_DMA_ACC(slot, t1, t0)
#endif

struct DmaTile {
    int img, oy0, ox0, nblk;
    bool valid;
};

// Epilogue of conv_epilogue_coalesced (same arithmetic per value, same LDS exchange) with UNCONDITIONAL stores: a pixel
// or channel outside the output is written to the dump page instead of being skipped, so that the number of vector-memory
// operations a wave issues per tile is a constant the counted waits can rely on.
template <class C, int MODE>     // MODE 0 plain / ReLU / LeakyReLU, 3 sigmoid, 4 clamp01
__device__ __forceinline__ void dma_epilogue_mode(const ConvArgs &p, f32x16 (&acc)[C::WM][C::WN], int nblk, int wm, int wn, int lane,
                                                  int oy0, int ox0, int img, float *scratch)
{
    constexpr int WM = C::WM, WN = C::WN;
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int wpx = lane & 31, whalf = lane >> 5;          // accumulator layout: pixel, channel half-group
    const int rq = lane & 7, rpx = lane >> 3;              // read-back layout: channel quad, pixel within a group of 8
    const int cps = p.Cout >> 2;
    const bool ps = p.out_mode != VC_OUT_PLAIN;
    const int sc = ps ? 2 : 1;
    const bool res_first = MODE == 0 && p.res_first;
    float *const dump = reinterpret_cast<float *>(g_vc_dma_dump) + 4 * lane;
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int oy = oy0 + wm * WM + t;
        static_for<0, WN>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[t][n][4 * g], acc[t][n][4 * g + 1], acc[t][n][4 * g + 2], acc[t][n][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(&scratch[wpx * VC_EPI_ROWF + 8 * g + 4 * whalf]) = v;
            }
            const int co = nblk * C::BN + (wn * WN + n) * 32 + 4 * rq;   // first of this lane's 4 consecutive channels
            const int pos = ps ? co / cps : 0;
            const int cch = ps ? co - pos * cps : co;
            f32x4 gain = {1.f, 1.f, 1.f, 1.f};
            if (p.chscale) gain = *reinterpret_cast<const f32x4 *>(p.chscale + min(co, p.Cout - 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = 8 * j + rpx;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&scratch[pix * VC_EPI_ROWF + 4 * rq]);
                const int ox = ox0 + pix;
                const bool ok = oy < p.Ho && ox < p.Wo && co < p.Cout;
                const int yy = sc * oy + (pos >> 1), xx = sc * ox + (pos & 1);
                const long long o_off = (long long)img * p.out_sn + (long long)yy * p.out_sh + (long long)xx * p.out_sw + cch;
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    const long long r_off = (long long)img * p.res_sn + (long long)yy * p.res_sh + (long long)xx * p.res_sw + cch;
                    if (p.res_f16) {          // (VC_CFG_RES_F16: the identity of a residual block kept as half)
                        f16x4 rh = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                        if (ok) rh = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(p.res) + r_off);
                        r = f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                    } else if (ok) {
                        r = *reinterpret_cast<const f32x4 *>(p.res + r_off);
                    }
                }
                if (res_first) v += r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (MODE == 3) v[e] = 1.0f / (1.0f + expf(-v[e]));
                    else if constexpr (MODE == 4) v[e] = fminf(fmaxf(v[e], 0.0f), 1.0f);
                    else v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                }
                if (p.chscale) v *= gain;
                if (p.res && !res_first) v += r;
                if (p.out_f16) {
                    const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    _Float16 *dst = ok ? reinterpret_cast<_Float16 *>(p.out) + o_off : reinterpret_cast<_Float16 *>(dump);
                    *reinterpret_cast<f16x4 *>(dst) = hv;
                } else {
                    float *dst = ok ? p.out + o_off : dump;
                    *reinterpret_cast<f32x4 *>(dst) = v;
                }
            }
        });
    });
}

// v_max_f32 as ONE instruction: fmaxf() on an MFMA result makes hipcc put a canonicalising v_max(v, v) in front of it
// (cdna guide, attention notes) -- a third of the fast epilogue's vector instructions
__device__ __forceinline__ float vc_max_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Fast epilogue for the common case inside a chain of fp16-path layers: half-precision output, plain / ReLU / LeakyReLU
// (slope in [0, 1]), no residual / gain / pixel shuffle, tile entirely inside the output.  Per M-tile the wave's
// 32 px x (32 WN) channels are activated and rounded in the accumulator layout, pass through a private 32 x (64 WN)-byte
// LDS image (16-byte chunks XOR-ed with the pixel index: conflict-free ds_read_b128) and leave as 16 bytes per lane =
// whole 128-byte lines per 8 lanes: WM * WN * 2 store instructions per wave and tile instead of WM * WN * 4 of 8 bytes.
// Same value per output as dma_epilogue_mode<0>: max(v, v * neg) == (v >= 0 ? v : v * neg) for 0 <= neg <= 1, one
// round-to-nearest conversion.
template <class C>
__device__ __forceinline__ void dma_epilogue_fast(const ConvArgs &p, f32x16 (&acc)[C::WM][C::WN], int nblk, int wm, int wn, int lane,
                                                  int oy0, int ox0, int img, unsigned char *scratch)
{
    constexpr int WM = C::WM, WN = C::WN;
    constexpr int CH16 = 4 * WN;                 // 16-byte chunks (8 channels) per pixel of this wave's channel range
    constexpr int ROWB = 16 * CH16;              // bytes per pixel in the exchange image
    constexpr int PPI = 64 / CH16;               // pixels per store instruction
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int wpx = lane & 31, wh = lane >> 5;
    const int rc = lane % CH16, rp = lane / CH16;
    // (wave-uniform) first output element of the wave's channel range in the tile's first pixel
    _Float16 *const obase = reinterpret_cast<_Float16 *>(p.out) + (long long)img * p.out_sn + (long long)oy0 * p.out_sh +
                            (long long)ox0 * p.out_sw + nblk * C::BN + wn * WN * 32;
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        static_for<0, WN>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[t][n][4 * g + e];
                    h[e] = (_Float16)vc_max_f32(v, v * neg);
                }
                const int c = 4 * n + g;
                *reinterpret_cast<f16x4 *>(scratch + wpx * ROWB + ((c ^ (wpx & (CH16 - 1))) << 4) + 8 * wh) = h;
            }
        });
        const long long row = (long long)(wm * WM + t) * p.out_sh;
#pragma unroll
        for (int j = 0; j < 32 / PPI; ++j) {
            const int px = PPI * j + rp;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(scratch + px * ROWB + ((rc ^ (px & (CH16 - 1))) << 4));
            *reinterpret_cast<f32x4 *>(obase + row + (long long)px * p.out_sw + 8 * rc) = d;
        }
    });
}

// Residual-block epilogue of the fp16 path (VC_CFG_RES_F16 on a plain 3x3 instance; LHBDC/model/layers.py:48-56,82-91 through
// compressai's ResidualBlock: out = act(conv2(t)) + identity): half-precision output AND a half-precision identity, tile inside the
// output.  The exchange of the fused-tail epilogue: 16 pixels x 64 channels (a pair of N-tiles) at a time through an fp32 scratch,
// read back as 8 consecutive channels per lane -- one 16-byte identity load (requested ONE step ahead) and one 16-byte store per
// lane and pass; the sum is formed in fp32 and rounded once, exactly like dma_epilogue_mode<0> does value by value.
template <class C>
__device__ __forceinline__ void dma_epilogue_resh(const ConvArgs &p, f32x16 (&acc)[C::WM][C::WN], int nblk, int wm, int wn, int lane,
                                                  int oy0, int ox0, int img, float *scratch)
{
    constexpr int WM = C::WM, WN = C::WN, RF = 68, STEPS = WN;          // steps per M-tile: (pair of N-tiles, 16-pixel part)
    static_assert(WN % 2 == 0, "pairs of N-tiles (64 channels = one 128-byte line of a half-precision pixel)");
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int wpx = lane & 31, whalf = lane >> 5;
    const int rp8 = lane >> 3, ro = lane & 7;
    const int co0 = nblk * C::BN + wn * WN * 32;                        // the wave's first output channel
    const _Float16 *const rbase = reinterpret_cast<const _Float16 *>(p.res) + (long long)img * p.res_sn + co0 + 8 * ro;
    _Float16 *const obase = reinterpret_cast<_Float16 *>(p.out) + (long long)img * p.out_sn + co0 + 8 * ro;
    f32x4 rnext[2] = {};
    auto request = [&](int t, int step) {
        const int pr = step >> 1, part = step & 1;
        const int oy = oy0 + wm * WM + t;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = ox0 + 16 * part + 8 * j + rp8;
            rnext[j] = *reinterpret_cast<const f32x4 *>(rbase + (long long)oy * p.res_sh + (long long)ox * p.res_sw + pr * 64);
        }
    };
    request(0, 0);
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int oy = oy0 + wm * WM + t;
        static_for<0, STEPS>([&](auto sc) {
            constexpr int step = decltype(sc)::value, pr = step >> 1, part = step & 1;
            const f32x4 rcur[2] = {rnext[0], rnext[1]};
            if constexpr (step + 1 < STEPS) request(t, step + 1);
            else if constexpr (t + 1 < WM) request(t + 1, 0);
            const int row = ((wpx >> 4) == part) ? (wpx & 15) * RF : 16 * RF;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = acc[t][2 * pr + hh][4 * g + e];
                        v[e] = vc_max_f32(a, a * neg);
                    }
                    *reinterpret_cast<f32x4 *>(&scratch[row + 32 * hh + 8 * g + 4 * whalf]) = v;
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pix = 8 * j + rp8;
                f32x4 v0 = *reinterpret_cast<const f32x4 *>(&scratch[pix * RF + 8 * ro]);
                f32x4 v1 = *reinterpret_cast<const f32x4 *>(&scratch[pix * RF + 8 * ro + 4]);
                const f16x8 rh = __builtin_bit_cast(f16x8, rcur[j]);
                v0 += f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                v1 += f32x4{(float)rh[4], (float)rh[5], (float)rh[6], (float)rh[7]};
                const f16x8 hv = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                                  (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                const int ox = ox0 + 16 * part + pix;
                *reinterpret_cast<f16x8 *>(obase + (long long)oy * p.out_sh + (long long)ox * p.out_sw + pr * 64) = hv;
            }
        });
    });
}

// Bottleneck-block epilogue (DmaCfg::TAIL): out = W2 . act(conv3x3) + b2 (+ residual), the block's trailing 1x1 layer applied
// to the accumulators M-tile by M-tile.  The accumulator layout IS a B operand of v_mfma_f32_32x32x16_f16 up to a permutation of k
// (lane (pixel, h), registers 8 (j % 2) .. + 7 of N-tile j / 2 = channels 32 (j / 2) + 16 (j % 2) + 4 h + {0..3, 8..11}); the 1x1 weights
// are packed to that k order (vc_conv_pack_tail_f16), so no value moves between lanes: activation + rounding to half (exactly
// what the unfused 3x3 layer stores), 32 MFMAs per M-tile with the weight fragments read from LDS one k-step ahead, then the
// coalesced exchange / residual / store of dma_epilogue_mode.  Every store is issued (dump page), like there.
template <class C>
__device__ __forceinline__ void dma_epilogue_tail(const ConvArgs &p, f32x16 (&acc)[C::WM][C::WN], int wm, int lane, int oy0, int ox0, int img,
                                                  float *scratch, const unsigned char *lds8)
{
    constexpr int WM = C::WM, WN = C::WN, NT = C::NT, KQ = 2 * NT;
    static_assert(WN == NT, "every wave owns all channels of its pixels");
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int wpx = lane & 31, whalf = lane >> 5;
    const int rq = lane & 7, rpx = lane >> 3;
    const unsigned char *const wl = lds8 + C::TAIL_OFF + 16 * lane;
    const float *const b2 = reinterpret_cast<const float *>(lds8 + C::TAIL_OFF + NT * KQ * 1024);
    float *const dump = reinterpret_cast<float *>(g_vc_dma_dump) + 4 * lane;
    // Half-precision output path: the residual (the block's identity, half or fp32) is requested ONE exchange step ahead of its
    // use -- the first step of an M-tile before the M-tile's MFMAs.  Requested where it is added, each of the 2 * NT steps of a
    // tile exposed a full HBM round trip (~1.5 us of a 44 us tile at 128 channels, eight times; of an 11 us tile at 64).
    const bool half_path = p.out_f16 && (p.out_sw & 7) == 0 && (p.out_sh & 7) == 0 && (p.out_sn & 7) == 0 &&
                           (!p.res || (((p.res_sw | p.res_sh | p.res_sn) & (p.res_f16 ? 7 : 3)) == 0));
    constexpr int STEPS = 2 * (NT / 2);                  // exchange steps per M-tile: (channel pair, 16-pixel part)
    const int rp8h = lane >> 3, roh = lane & 7;
    f32x4 rnext[2] = {};                                 // [pass j]: 8 halves per lane (an fp32 residual is read where it is added)
    const bool res_ahead = half_path && p.res && p.res_f16;
    auto request = [&](int t, int step) {                // residual of exchange step `step` of M-tile t -> rnext
        const int pr = step >> 1, part = step & 1;
        const int oy = oy0 + wm * WM + t;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = ox0 + 16 * part + 8 * j + rp8h;
            const bool ok = oy < p.Ho && ox < p.Wo;
            const long long r_off = (long long)img * p.res_sn + (long long)oy * p.res_sh + (long long)ox * p.res_sw + pr * 64 + 8 * roh;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rnext[j] = ok ? *reinterpret_cast<const f32x4 *>(reinterpret_cast<const _Float16 *>(p.res) + r_off) : z;
        }
    };
    if (res_ahead) request(0, 0);
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int oy = oy0 + wm * WM + t;
        f32x16 acc2[NT];
#pragma unroll
        for (int o = 0; o < NT; ++o)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(&b2[o * 32 + 8 * g + 4 * whalf]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[o][4 * g + e] = b[e];
            }
        f32x4 wf[2][NT];
#pragma unroll
        for (int o = 0; o < NT; ++o) wf[0][o] = *reinterpret_cast<const f32x4 *>(wl + (o * KQ) * 1024);
        static_for<0, KQ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < KQ) {
#pragma unroll
                for (int o = 0; o < NT; ++o) wf[(j + 1) & 1][o] = *reinterpret_cast<const f32x4 *>(wl + (o * KQ + j + 1) * 1024);
            }
            f16x8 bop;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = acc[t][j / 2][8 * (j % 2) + i];
                bop[i] = (_Float16)vc_max_f32(v, v * neg);
            }
#pragma unroll
            for (int o = 0; o < NT; ++o)
                acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[j & 1][o]), bop, acc2[o], 0, 0, 0);
        });
        // ---- exchange, residual, store ----
        if (half_path) {
            // half-precision output: 16 pixels x 64 channels at a time through the scratch ([16][64 + 4] floats + a dump row the
            // lanes of the other 16 pixels write to), read back as 8 consecutive channels per lane: one 16-byte residual load
            // and one 16-byte store per lane and pass, every store instruction 8 whole 128-byte lines (a 32-channel tile alone
            // would write every line in two halves, 8 bytes per lane: twice the vector-memory instructions)
            constexpr int RF = 68;
            const int rp8 = rp8h, ro = roh;
            static_for<0, STEPS>([&](auto sc) {
                constexpr int step = decltype(sc)::value, pr = step >> 1, part = step & 1;
                const f32x4 rcur[2] = {rnext[0], rnext[1]};
                if (res_ahead) {                           // the next step's residual (of the next M-tile behind the last step)
                    if constexpr (step + 1 < STEPS) request(t, step + 1);
                    else if constexpr (t + 1 < WM) request(t + 1, 0);
                }
                const int row = ((wpx >> 4) == part) ? (wpx & 15) * RF : 16 * RF;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc2[2 * pr + hh][4 * g], acc2[2 * pr + hh][4 * g + 1], acc2[2 * pr + hh][4 * g + 2],
                                         acc2[2 * pr + hh][4 * g + 3]};
                        *reinterpret_cast<f32x4 *>(&scratch[row + 32 * hh + 8 * g + 4 * whalf]) = v;
                    }
                const int co = pr * 64 + 8 * ro;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pix = 8 * j + rp8;
                    f32x4 v0 = *reinterpret_cast<const f32x4 *>(&scratch[pix * RF + 8 * ro]);
                    f32x4 v1 = *reinterpret_cast<const f32x4 *>(&scratch[pix * RF + 8 * ro + 4]);
                    const int ox = ox0 + 16 * part + pix;
                    const bool ok = oy < p.Ho && ox < p.Wo;
                    const long long o_off = (long long)img * p.out_sn + (long long)oy * p.out_sh + (long long)ox * p.out_sw + co;
                    if (res_ahead) {
                        const f16x8 rh = __builtin_bit_cast(f16x8, rcur[j]);
                        v0 += f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                        v1 += f32x4{(float)rh[4], (float)rh[5], (float)rh[6], (float)rh[7]};
                    } else if (p.res && ok) {              // (fp32 identity: the first block of a chain)
                        const long long r_off = (long long)img * p.res_sn + (long long)oy * p.res_sh + (long long)ox * p.res_sw + co;
                        v0 += *reinterpret_cast<const f32x4 *>(p.res + r_off);
                        v1 += *reinterpret_cast<const f32x4 *>(p.res + r_off + 4);
                    }
                    const f16x8 hv = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                                      (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                    _Float16 *dst = ok ? reinterpret_cast<_Float16 *>(p.out) + o_off : reinterpret_cast<_Float16 *>(dump);
                    *reinterpret_cast<f16x8 *>(dst) = hv;
                }
            });
        } else {
        // (fp32 output -- the end of a chain -- or unaligned rows: one N-tile at a time, 4 channels per lane)
        static_for<0, NT>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc2[n][4 * g], acc2[n][4 * g + 1], acc2[n][4 * g + 2], acc2[n][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(&scratch[wpx * VC_EPI_ROWF + 8 * g + 4 * whalf]) = v;
            }
            const int co = n * 32 + 4 * rq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = 8 * j + rpx;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&scratch[pix * VC_EPI_ROWF + 4 * rq]);
                const int ox = ox0 + pix;
                const bool ok = oy < p.Ho && ox < p.Wo;
                const long long o_off = (long long)img * p.out_sn + (long long)oy * p.out_sh + (long long)ox * p.out_sw + co;
                if (p.res) {
                    const long long r_off = (long long)img * p.res_sn + (long long)oy * p.res_sh + (long long)ox * p.res_sw + co;
                    if (p.res_f16) {
                        f16x4 rh = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                        if (ok) rh = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(p.res) + r_off);
                        v += f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                    } else {
                        f32x4 r = {0.f, 0.f, 0.f, 0.f};
                        if (ok) r = *reinterpret_cast<const f32x4 *>(p.res + r_off);
                        v += r;
                    }
                }
                if (p.out_f16) {
                    const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    _Float16 *dst = ok ? reinterpret_cast<_Float16 *>(p.out) + o_off : reinterpret_cast<_Float16 *>(dump);
                    *reinterpret_cast<f16x4 *>(dst) = hv;
                } else {
                    float *dst = ok ? p.out + o_off : dump;
                    *reinterpret_cast<f32x4 *>(dst) = v;
                }
            }
        });
        }
    });
}

template <class C>
__device__ __forceinline__ void dma_epilogue(const ConvArgs &p, f32x16 (&acc)[C::WM][C::WN], int nblk, int wm, int wn, int lane,
                                             int oy0, int ox0, int img, float *scratch)
{
    if (p.act == VC_ACT_SIGMOID) dma_epilogue_mode<C, 3>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, scratch);
    else if (p.act == VC_ACT_CLAMP01) dma_epilogue_mode<C, 4>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, scratch);
    else dma_epilogue_mode<C, 0>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, scratch);
}

template <class C> __global__ void __launch_bounds__(512, 2) conv_dma_kernel(const ConvArgs p)
{
    constexpr int KW = C::KW, TAPS = C::TAPS, NCHUNK = C::NCHUNK, NT = C::NT, RING = C::RING;
    constexpr int WM = C::WM, WN = C::WN, UPP = C::UPP, UT = C::UT, PT = C::PT, NA = C::NA;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];   // the kernel's only LDS object: starts at LDS address 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                       // the two waves of a SIMD belong to different groups
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    // ---- persistent tile list: the 32 workgroups that share an XCD (ids equal mod 8) walk one contiguous range of the
    // banded tile order together, so halos and weights are shared in that XCD's L2 ----
    const int per_img = p.tiles_x * p.tiles_y;
    const int total = per_img * p.N * p.nblks;
    const int xcd = blockIdx.x & 7, xl = blockIdx.x >> 3;
    const int tq = total >> 3, tr = total & 7;
    const int x_start = xcd * tq + min(xcd, tr), x_count = tq + (xcd < tr ? 1 : 0);
    auto tile_at = [&](int it) {
        DmaTile t;
        const int k = it * 32 + xl;
        t.valid = k < x_count;
        const int idx = x_start + (t.valid ? k : 0);
        t.nblk = idx % p.nblks;
        const int t1 = idx / p.nblks;
        t.img = t1 / per_img;
        int tx, ty;
        vc_tile_xy(t1 - t.img * per_img, p.tiles_x, p.tiles_y, p.tile_band, tx, ty);
        t.oy0 = ty * C::TH;
        t.ox0 = tx * C::TW;
        return t;
    };
    DmaTile cur = tile_at(0);
    if (!cur.valid) return;

    const unsigned char *const in_b = reinterpret_cast<const unsigned char *>(p.in);
    const unsigned char *const wpk = reinterpret_cast<const unsigned char *>(p.wpk);
    const int kst = p.cin_pad >> (C::F32 ? 3 : 4);   // packed k-steps (16 halves / 8 floats) per tap
    const unsigned char *const zero_lane = g_vc_dma_zero + 16 * lane;

    // ---- issue helpers ----
    // Piece k of a chunk image: lane (k, tid) fills LDS slot s = 512 k + tid = 4 * pixel + quarter; the quarter holds channel
    // group (quarter ^ swizzle(column)) of the pixel -- the permutation is applied to the SOURCE address.  What does not
    // depend on the tile is kept per lane: the byte offset of the lane's 16 bytes from the footprint's first pixel, and
    // its (row, column) for the border test; a tile contributes a scalar base.
    int a_off[NA], a_rc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int s = k * 512 + tid;
        const int q = s >> 2;
        const int row = q / C::COLS, col = q - row * C::COLS;
        const int g = (s & 3) ^ ((col >> 2) & 3);
        a_off[k] = C::ESZ * (row * (int)p.in_sh + col * (int)p.in_sw + g * (16 / C::ESZ));
        a_rc[k] = q < C::PIX ? (row | (col << 8)) : 0x7f7f7f;       // (a row / column no image reaches)
    }
    auto tile_base = [&](const DmaTile &t) {           // address of the footprint's first pixel (may lie outside the tensor)
        return in_b + C::ESZ * ((long long)t.img * p.in_sn + (long long)(t.oy0 - C::KH / 2) * p.in_sh + (long long)(t.ox0 - KW / 2) * p.in_sw);
    };
    auto issue_a = [&](const DmaTile &t, const unsigned char *tbase, int c, int k, int buf) {
        // (the two per-lane constants pass through an empty asm: the address is then computed HERE, not hoisted to the top
        //  of the tile loop for all pieces of both tiles at once -- 20 registers the accumulators need)
        int rc = a_rc[k], off = a_off[k];
        asm volatile("" : "+v"(rc), "+v"(off));
        const unsigned iy = (unsigned)(t.oy0 - C::KH / 2 + (rc & 0xff)), ix = (unsigned)(t.ox0 - KW / 2 + (rc >> 8));
        const bool ok = t.valid && iy < (unsigned)p.H && ix < (unsigned)p.W;
        const unsigned char *sp = ok ? tbase + (off + c * 64) : zero_lane;
        if constexpr (!(C::KO & (16 | 1024))) vc_glds16<!(C::KO & 512)>(sp, (unsigned)(buf * C::A_BYTES + k * 8192 + wave * 1024));
    };
    // Weights of tile-phase pb (>= PT: of the next tile, same layer): wave w fetches fragment w of the phase =
    // (unit w / (2 NT), k-step (w / NT) % 2, N-tile w % NT) from the packed [n-tile][tap][k-step] array.
    const int b_uu = wave / (2 * NT), b_ks = (wave / NT) & 1, b_nt = wave % NT;
    const long long b_wave_frag = (long long)b_nt * TAPS * kst + b_ks;       // wave-uniform
    const unsigned lane16 = 16 * lane;
    auto issue_b = [&](int pb, int nblk_cur, int nblk_next) {
        const int pp = pb >= PT ? pb - PT : pb;
        const int nblk = pb >= PT ? nblk_next : nblk_cur;
        const int u = pp * UPP + (UPP > 1 ? b_uu : 0);
        const int c = u / TAPS, tap = u - c * TAPS;
        const long long frag = b_wave_frag + (long long)nblk * NT * TAPS * kst + tap * kst + 2 * c;
        const unsigned char *sbase = (UPP * PT == UT || u < UT) ? wpk + frag * 1024 : g_vc_dma_zero;
        if constexpr (!(C::KO & (16 | 2048))) {
            if constexpr (C::KO & 8192) vc_glds16<false>(sbase + lane16, (unsigned)(C::B_OFF + (pp % RING) * 8192 + wave * 1024));
            else vc_glds16_sbase(sbase, lane16, (unsigned)(C::B_OFF + (pp % RING) * 8192 + wave * 1024));
        }
    };

    // ---- prologue: bias of every channel block, first chunk image, weights of the first RING-2 phases ----
    float *const ldsf = reinterpret_cast<float *>(lds8);
    for (int i = tid; i < p.nblks * C::BN; i += 512) ldsf[C::BIAS_OFF / 4 + i] = p.bias[i];
    if constexpr (C::TAIL) {     // the trailing 1x1 layer: its fragments by DMA (lane-linear 1 KiB blocks), its bias behind them
        const unsigned char *const tw = reinterpret_cast<const unsigned char *>(p.tail_wpk);
        for (int f = wave; f < NT * 2 * NT; f += 8) vc_glds16_sbase(tw + f * 1024, (unsigned)(16 * lane), (unsigned)(C::TAIL_OFF + f * 1024));
        for (int i = tid; i < C::BN; i += 512) ldsf[(C::TAIL_OFF + NT * 2 * NT * 1024) / 4 + i] = p.tail_bias[i];
    }
    DmaTile nxt = tile_at(1);
    const unsigned char *cur_base = tile_base(cur), *nxt_base = tile_base(nxt);
#pragma unroll
    for (int k = 0; k < NA; ++k) issue_a(cur, cur_base, 0, k, 0);
#pragma unroll
    for (int pb = 0; pb < RING - 2; ++pb) issue_b(pb, cur.nblk, cur.nblk);
    vc_wait_vmcnt<0>();
    __syncthreads();

    // ---- per-lane LDS read offsets ----
    int a_lane[KW][2];
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int col = li + kx;
            a_lane[kx][ks] = (wm * WM * C::COLS + col) * 64 + (((2 * ks + lh) ^ ((col >> 2) & 3)) << 4);
        }
    const int b_lane = C::B_OFF + wn * WN * 1024 + lane * 16;

#ifdef VC_DMA_DIAG
    unsigned st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    VC_DMA_STAMP(t_begin);
    f32x16 acc[WM][WN];
    f32x4 af[UPP][2][WM], bf[UPP][2][WN];             // the fragments of one phase (waves 4-7 carry them across a barrier)
    int gchunk = 0;                                  // chunks contracted so far: parity = buffer of the current chunk
    for (int it = 0;; ++it) {
        // ---- accumulators start at the bias ----
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(&ldsf[C::BIAS_OFF / 4 + cur.nblk * C::BN + (wn * WN + n) * 32 + 8 * g + 4 * lh]);
#pragma unroll
                for (int t = 0; t < WM; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][n][4 * g + e] = b[e];
            }
        // ---- the three parts of a phase ----
        auto issue_loads = [&](auto pc) {            // DMA of later phases: pieces of the next chunk image, then this phase's weight slot
            constexpr int ph = decltype(pc)::value;
            static_for<0, NCHUNK>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                static_for<0, NA>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (C::piece_phase(c, k) == ph) {
                        // chunk c+1 of this tile, or chunk 0 of the next one, into the buffer chunk c does not use
                        if constexpr (c + 1 < NCHUNK) issue_a(cur, cur_base, c + 1, k, (gchunk + c + 1) & 1);
                        else issue_a(nxt, nxt_base, 0, k, (gchunk + c + 1) & 1);
                    }
                });
            });
            issue_b(ph + RING - 2, cur.nblk, nxt.nblk);
        };
        auto read_frags = [&](auto pc) {             // this phase's fragments: LDS -> registers
            constexpr int ph = decltype(pc)::value;
            static_for<0, UPP>([&](auto uc) {
                constexpr int uu = decltype(uc)::value;
                constexpr int u = ph * UPP + uu;
                if constexpr (u < UT && (C::KO & 8)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int n = 0; n < WN; ++n) bf[uu][ks][n] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)lane;
#pragma unroll
                        for (int t = 0; t < WM; ++t) af[uu][ks][t] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + t);
                    }
                } else if constexpr (u < UT) {
                    constexpr int c = u / TAPS, tap = u % TAPS, ky = tap / KW, kx = tap % KW;
                    const int a_off = ((gchunk + c) & 1) * C::A_BYTES;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            bf[uu][ks][n] = *reinterpret_cast<const f32x4 *>(lds8 + b_lane + (ph % RING) * 8192 + ((uu * 2 + ks) * NT + n) * 1024);
#pragma unroll
                        for (int t = 0; t < WM; ++t)
                            af[uu][ks][t] = *reinterpret_cast<const f32x4 *>(lds8 + a_lane[kx][ks] + a_off + (t + ky) * C::COLS * 64);
                    }
                }
            });
        };
        auto contract = [&](auto pc) {               // 16 MFMAs per wave on the fragments read_frags left in registers
            constexpr int ph = decltype(pc)::value;
            if constexpr (!(C::KO & 128)) __builtin_amdgcn_s_setprio(1);
            static_for<0, UPP>([&](auto uc) {
                constexpr int uu = decltype(uc)::value;
                if constexpr (ph * UPP + uu < UT && (C::KO & 4)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)       // keep the fragments alive without the matrix work
#pragma unroll
                        for (int t = 0; t < WM; ++t) {
                            VC_DMA_KEEP(af[uu][ks][t]);
                            VC_DMA_KEEP(bf[uu][ks][t % WN]);
                        }
                } else if constexpr (ph * UPP + uu < UT && C::F32) {
                    // (sub-steps outermost: consecutive MFMAs go to different accumulators; per accumulator the order
                    //  k-step, sub-step is the classic kernel's)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int t = 0; t < WM; ++t)
#pragma unroll
                                for (int n = 0; n < WN; ++n)
                                    acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[uu][ks][n][e], af[uu][ks][t][e], acc[t][n], 0, 0, 0);
                } else if constexpr (ph * UPP + uu < UT) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int t = 0; t < WM; ++t)
#pragma unroll
                            for (int n = 0; n < WN; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[uu][ks][n]),
                                                                                   __builtin_bit_cast(f16x8, af[uu][ks][t]), acc[t][n], 0, 0, 0);
                }
            });
            if constexpr (!(C::KO & 128)) __builtin_amdgcn_s_setprio(0);
        };

        if (grp == 1 && !(C::KO & 32)) VC_DMA_BARRIER();              // waves 4-7 run half a phase behind waves 0-3
        static_for<0, PT>([&](auto pc) {
            constexpr int ph = decltype(pc)::value;
            VC_DMA_STAMP(t0);
            if constexpr (C::KO & 256) issue_loads(pc);
            VC_DMA_STAMP(t1);
            read_frags(pc);
            VC_DMA_STAMP(t1b);
            if constexpr (!(C::KO & 256)) issue_loads(pc);
            VC_DMA_STAMP(t2);                        // (a stamp waits for the fragment reads: t2 - t1 = LDS reads issued and returned)
            if constexpr (!(C::KO & 2)) vc_wait_vmcnt<C::nwait(ph)>();           // this wave's part of the next phase's weights (and everything older) has landed
            VC_DMA_STAMP(t3);
            VC_DMA_BARRIER();
            VC_DMA_STAMP(t4);
            contract(pc);                            // -- M: 16 MFMAs per wave while the other group reads --
            VC_DMA_STAMP(t5);
            VC_DMA_BARRIER();
            VC_DMA_STAMP(t6);
            VC_DMA_ACC(0, t1, t0);   // DMA issue (before the fragment reads, or after them: KO bit 256)
            VC_DMA_ACC(0, t2, t1b);
            VC_DMA_ACC(1, t1b, t1);  // fragment reads
            VC_DMA_ACC(2, t3, t2);   // vmcnt wait
            VC_DMA_ACC(3, t4, t3);   // barrier after R
            VC_DMA_ACC(4, t5, t4);   // MFMA issue
            VC_DMA_ACC(5, t6, t5);   // barrier after M
        });
        if (grp == 0 && !(C::KO & 32)) VC_DMA_BARRIER();              // (waves 4-7 finish their last phase)

        // ---- epilogue through the buffer of the chunk that has just been finished ----
        gchunk += NCHUNK;
        float *scratch = reinterpret_cast<float *>(lds8 + ((gchunk - 1) & 1) * C::A_BYTES) + wave * VC_EPI_SCRATCH_FLOATS;
        VC_DMA_STAMP(t_e0);
        if constexpr (C::KO & 1) {
#pragma unroll
            for (int t = 0; t < WM; ++t)
#pragma unroll
                for (int n = 0; n < WN; ++n) VC_DMA_KEEP(acc[t][n]);
        } else if constexpr (C::TAIL) {
            dma_epilogue_tail<C>(p, acc, wm, lane, cur.oy0, cur.ox0, cur.img, scratch, lds8);
        } else {
            const bool fast = !(C::KO & 16384) && p.out_f16 && !p.res && !p.chscale && p.out_mode == VC_OUT_PLAIN &&
                              (p.act == VC_ACT_NONE || p.act == VC_ACT_RELU || (p.act == VC_ACT_LRELU && p.slope >= 0.0f && p.slope <= 1.0f)) &&
                              cur.oy0 + C::TH <= p.Ho && cur.ox0 + C::TW <= p.Wo && (cur.nblk + 1) * C::BN <= p.Cout &&
                              (p.out_sw & 7) == 0 && (p.out_sh & 7) == 0 &&
                              (p.out_sn & 7) == 0;
            bool resh = false;
            if constexpr (!C::F32 && C::WN % 2 == 0)        // half output + half identity, whole tile inside the output (VC_CFG_RES_F16)
                resh = !(C::KO & 16384) && p.out_f16 && p.res && p.res_f16 && !p.res_first && !p.chscale && p.out_mode == VC_OUT_PLAIN &&
                       (p.act == VC_ACT_NONE || p.act == VC_ACT_RELU || (p.act == VC_ACT_LRELU && p.slope >= 0.0f && p.slope <= 1.0f)) &&
                       cur.oy0 + C::TH <= p.Ho && cur.ox0 + C::TW <= p.Wo && (cur.nblk + 1) * C::BN <= p.Cout &&
                       ((p.out_sw | p.out_sh | p.out_sn | p.res_sw | p.res_sh | p.res_sn) & 7) == 0;
            if (fast)
                dma_epilogue_fast<C>(p, acc, cur.nblk, wm, wn, lane, cur.oy0, cur.ox0, cur.img,
                                     reinterpret_cast<unsigned char *>(scratch - wave * VC_EPI_SCRATCH_FLOATS) + wave * 4096);
            else if (resh) {
                if constexpr (!C::F32 && C::WN % 2 == 0) dma_epilogue_resh<C>(p, acc, cur.nblk, wm, wn, lane, cur.oy0, cur.ox0, cur.img, scratch);
            } else
                dma_epilogue<C>(p, acc, cur.nblk, wm, wn, lane, cur.oy0, cur.ox0, cur.img, scratch);
        }
        VC_DMA_STAMP(t_e1);
        VC_DMA_ACC(6, t_e1, t_e0);   // epilogue
        if (!nxt.valid) break;
        cur = nxt;
        cur_base = nxt_base;
        nxt = tile_at(it + 2);
        nxt_base = tile_base(nxt);
    }
    vc_wait_vmcnt<0>();                              // no DMA may land in LDS that already belongs to another workgroup
#ifdef VC_DMA_DIAG
    if constexpr (C::KO & 64) {
        VC_DMA_STAMP(t_end);
        st_sum[7] = t_end - t_begin;                 // wave lifetime
        if (lane == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_vc_dma_stamps[i], (unsigned long long)st_sum[i]);
    }
#endif
}

template <class C> int launch_conv_dma(hipStream_t st, const ConvArgs &a)
{
    const size_t lds_bytes = C::LDS_FIXED + (size_t)a.nblks * C::BN * sizeof(float);
    if (lds_bytes > 160 * 1024) return VC_EINVAL;
    auto kern = conv_dma_kernel<C>;
    static vc_lds_raised raised;
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}
