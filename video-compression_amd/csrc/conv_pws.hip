// Pointwise (1x1, stride 1) convolution as an LDS-DMA streaming GEMM -- tile configuration VC_CFG_PWS.
//
// A 1x1 layer is a stream: 32 FLOP/B at 128 channels in fp32, 16 and less on the fp16 path, against a machine balance of
// 157 TFLOP/s : ~5.6 TB/s (tools/micro/stream_rate.hip: what a copy kernel gets) = 28 FLOP/B.  The first streaming kernel
// (conv_pw.hip) kept one tile's activations plus the next tile's prefetch in registers (128 of them at 128 channels): the
// compiler had none left to fetch weight fragments ahead (lgkmcnt(0) in front of every MFMA pair), its loads touched 32
// lines of 32 bytes each, and a residual was requested one 16-pixel group before its use.  Here
//   * ONE persistent 512-thread workgroup per CU; the weights (<= 64 KiB, the packed fragments of the 32-wide
//     configurations) and the bias sit in LDS once per CU;
//   * every wave owns 32-pixel tiles and a private LDS ring of D "sub-rows" (32 pixels x 128 bytes): activations arrive by
//     global_load_lds_dwordx4 in whole 128-byte lines, D-1 sub-rows ahead of the one being contracted, and are retired
//     with counted s_waitcnt vmcnt(N) -- no barrier after the weights are in place, the waves drift apart freely;
//   * the ring image is lane-linear (what the DMA writes); the conflict-free ds_read_b128 layout of the B operand comes
//     from permuting the per-lane SOURCE addresses (16-byte groups XOR-ed with bits of the pixel index);
//   * a step first moves the NEXT sub-row's B operand from the ring into registers and refills that slot, then contracts
//     the current sub-row from registers: D sub-rows are in flight or landed beside the two in registers;
//   * the residual of a tile is requested behind the tile's last DMA wait (inline-asm loads into registers, in the
//     coalesced read-back layout of the epilogue), a whole tile at once instead of 16 pixels ahead of its use;
//   * every store is issued (a pixel past the end of a row repeats the row's last pixel, input and output), so the number
//     of vector-memory operations per tile is a compile-time constant the counted waits rely on.
// Accumulation order (bias first, k-steps in order, (e, 4+e) pairs inside one MFMA) and the epilogue arithmetic are
// those of every other configuration of the layer: results are bit-identical (tests/test_ops_gpu.py).
#include "conv_dma.h"

namespace {

template <int KQ_, int NT_, int PREC_> struct PwsCfg {      // PREC: 0 fp32, 1 fp16 MFMA on an fp32 tensor, 2 fp16 MFMA on a half tensor
    static constexpr int KQ = KQ_, NT = NT_, PREC = PREC_;
    static constexpr bool F16 = PREC_ != 0, INH = PREC_ == 2;
    static constexpr int ESZ = INH ? 2 : 4, KCH = F16 ? 16 : 8;
    static constexpr int ROWB = KQ_ * KCH * ESZ;             // bytes of one input pixel
    static constexpr int S = (ROWB % 128 == 0) ? 128 : 64;   // sub-row: bytes of a pixel one DMA piece covers
    static constexpr int GPS = S / 16, NSUB = ROWB / S, SUBB = 32 * S, NI = SUBB / 1024, PPI = 1024 / S;
    static constexpr int KB = KCH * ESZ;                     // bytes of one k-step of one pixel (32, or 64 for fp32 data on the fp16 path)
    static constexpr int QPS = S / KB, RPQ = KB / 32;        // k-steps per sub-row; 16-byte reads per k-step and lane
    static constexpr int NCUR = QPS * RPQ;
    static constexpr int W_BYTES = NT_ * KQ_ * 1024, BIAS_OFF = W_BYTES, GAIN_OFF = BIAS_OFF + NT_ * 128, WAVE_OFF = GAIN_OFF + NT_ * 128;
    static constexpr int SCR = 17 * VC_EPI_ROWF * 4;         // 16 pixels x (128 B + 16 B pad) + a dump row
    static constexpr int D_FIT = (160 * 1024 - WAVE_OFF - 8 * SCR) / (8 * SUBB);
    static constexpr int D = D_FIT > 4 ? 4 : D_FIT;          // ring depth in sub-rows
    static constexpr int WAVE_BYTES = D * SUBB + SCR, LDS_BYTES = WAVE_OFF + 8 * WAVE_BYTES;
    static_assert(D >= 2 && QPS >= 1 && NSUB * QPS == KQ_, "ring / sub-row geometry");
    // Counted waits (vmcnt retires in issue order).  Step u = (tile i, sub-row s) waits for sub-row u+1, reads it into
    // registers, refills the slot it has just emptied with sub-row u+1+D and contracts sub-row u.  The piece waited for was
    // issued in step u-D; younger than it are the D-1 sub-rows of steps u-D+1 .. u-1 and the stores of every epilogue in
    // between (the tile boundaries after steps u-D .. u-1).  The residual loads of a tile are issued BEHIND the tile's last
    // refill and consumed before the next tile's first wait, so that no wait ever mixes the two kinds of load: a DMA wait
    // counts DMA pieces and stores, a residual wait residual loads and stores (measured next to the alternative --
    // residual at the start of its tile, every wait counting both kinds -- it is also the faster order:
    // profiles/r03/o_pw_check_residual_at_tile_start_variant.log).
    // Stores per tile: units of 16 pixels x 128 bytes, 2 per unit.
    static constexpr int NG32 = 2 * NT_, NG16 = 2 * ((NT_ + 1) / 2);
    // (a SPLIT output -- VC_CFG_OUT_SP3, fp32 instances -- stores three 8-byte pieces instead of one 16-byte group: 6 per unit)
    static constexpr int eops(bool half_plain, bool osp = false) { return half_plain ? NG16 * 2 : NG32 * (osp ? 6 : 2); }
    static constexpr int nb(int s) { return (D - 1 - s >= 0) ? (D - 1 - s) / NSUB + 1 : 0; }
    static constexpr int clamp63(int n) { return n > 63 ? 63 : n; }
    // `young`: the wave's first steps, where the oldest of those epilogues is the one before the first tile -- there is none
    static constexpr int wait_a(int s, bool half_plain, bool young, bool osp = false)
    {
        return clamp63((D - 1) * NI + (nb(s) - (young ? 1 : 0)) * eops(half_plain, osp));
    }
    static constexpr int WAIT_RES = clamp63(2 * (NG32 - 1));
    // split output: behind unit g's residual loads lie the loads of the units after it and the 6 g stores of the units before it
    static constexpr int wait_res_sp(int g) { return clamp63(2 * (NG32 - 1 - g) + 6 * g); }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pws_load8(const void *uniform_base, unsigned lane_off)
{
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(v) : "v"(lane_off), "s"(uniform_base) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 pws_load16(const void *uniform_base, unsigned lane_off)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(lane_off), "s"(uniform_base) : "memory");
    return v;
}

// GDN (fp32 instances, 128 -> 128 only): 0 none; 1 GDN / 2 IGDN of compressai.layers -- the layer is a 1x1 contraction of x^2 and its
// epilogue multiplies the un-squared x by rsqrt / sqrt of the result (conv_mfma.h: VC_IN_SQUARE + VC_EPI_GDN / IGDN with mul == in).
// The B operand registers of lane (pixel, half) ARE the elements its accumulators belong to (k-step 4 t + g, element e <->
// accumulator 4 g + e of N-tile t), so the tile's x stays in 64 registers (squared on the way into the MFMA) and no second read
// of the input exists.  The residual (the block's skip path) is then requested 16 pixels ahead of its use instead of a tile at
// once: 64 registers fewer.
template <class C, int RES, int GDN = 0, bool OSP = false> __global__ void __launch_bounds__(512, 2) conv_pws_kernel(const ConvArgs p)      // RES: 0 none, 1 fp32, 2 half
{
    static_assert(GDN == 0 || (!C::F16 && C::KQ == 16 && C::NT == 4 && RES != 2), "GDN / IGDN: the fp32 128 -> 128 instance");
    constexpr int KQ = C::KQ, NT = C::NT, NSUB = C::NSUB, D = C::D, NI = C::NI, QPS = C::QPS, RPQ = C::RPQ, S = C::S;
    constexpr bool F16 = C::F16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights and bias: once per CU.  The fragments are lane-linear 1 KiB blocks: they go straight into LDS by DMA, all
    //      of a wave's pieces in flight at once (a load -> wait -> ds_write loop per thread cost eight serial L2 round trips,
    //      a third of the whole launch on a 136 x 240 map) ----
    {
        const int kst_total = p.cin_pad / C::KCH;
        const unsigned char *const wpk = reinterpret_cast<const unsigned char *>(p.wpk);
        for (int f = wave; f < NT * KQ; f += 8) {
            const int q = f % KQ, t = f / KQ;
            vc_glds16_sbase(wpk + ((long long)t * kst_total + q) * 1024, (unsigned)(lane * 16), (unsigned)(f * 1024));
        }
        for (int i = tid; i < NT * 32; i += 512) {
            reinterpret_cast<float *>(lds8 + C::BIAS_OFF)[i] = p.bias[i];
            // channel gains too: no compiler-counted global load may sit in the tile loop (its wait would not know of the DMA)
            reinterpret_cast<float *>(lds8 + C::GAIN_OFF)[i] = (p.chscale && i < p.Cout) ? p.chscale[i] : 1.0f;
        }
    }
    const int n = lane & 31, h = lane >> 5, rq = lane & 7, rpx = lane >> 3;
    const unsigned tiles_x = (unsigned)(p.W + 31) >> 5;
    const unsigned ntiles = (unsigned)p.N * (unsigned)p.H * tiles_x;
    // tile -> (CU, wave): consecutive tiles go to consecutive CUs, so that a small map (fewer tiles than waves) occupies one
    // wave per SIMD on every CU before it doubles any up
    const unsigned gw = wave * gridDim.x + blockIdx.x, gstride = gridDim.x * 8;      // (waves w and w + 4 share a SIMD)
    if (gw >= ntiles) {                               // (nothing to do for this wave: it still owes the workgroup its weight pieces)
        vc_wait_vmcnt<0>();
        __syncthreads();
        return;
    }
    const unsigned last_mine = gw + (ntiles - 1 - gw) / gstride * gstride;
    struct Loc { int img, y, x0; };
    auto locate = [&](unsigned tile) {
        tile = min(tile, last_mine);                  // past the end: the last tile again (the counted waits want every request issued)
        const unsigned r = tile / tiles_x;
        Loc l;
        l.x0 = (int)(tile - r * tiles_x) * 32;
        l.img = (int)(r / (unsigned)p.H);
        l.y = (int)(r - (unsigned)l.img * (unsigned)p.H);
        return l;
    };

    // ---- DMA side: lane i of piece k fills ring slot 64 k + i = pixel (PPI k + i / GPS), 16-byte group i % GPS, which holds
    //      source group (i % GPS) ^ swizzle(pixel) ----
    const unsigned char *const in_b = reinterpret_cast<const unsigned char *>(p.in);
    const int in_pix_bytes = (int)p.in_sw * C::ESZ;
    const int dn0 = S == 128 ? (lane >> 3) : (lane >> 2);
    const int dg16 = S == 128 ? (((lane & 7) ^ (lane >> 4)) << 4) : (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned ring = C::WAVE_OFF + wave * C::WAVE_BYTES;
    unsigned d_tile = gw;                              // DMA cursor: tile, sub-row, ring slot
    int d_sub = 0, d_slot = 0;
    Loc dl = locate(d_tile);
    auto d_base = [&]() { return in_b + ((long long)dl.img * p.in_sn + (long long)dl.y * p.in_sh + (long long)dl.x0 * p.in_sw) * C::ESZ; };
    const unsigned char *dbase = d_base();
    int d_wlim = p.W - 1 - dl.x0;
    auto issue_dma = [&]() {
        const unsigned char *sbase = dbase + d_sub * S;
        const unsigned dst = ring + d_slot * C::SUBB;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int nn = min(dn0 + C::PPI * k, d_wlim);
            const unsigned off = (unsigned)(nn * in_pix_bytes + (dg16 ^ ((S == 128 && (k & 1)) ? 64 : 0)));
            vc_glds16_sbase(sbase, off, dst + k * 1024);
        }
        d_slot = d_slot + 1 == D ? 0 : d_slot + 1;
        if (++d_sub == NSUB) {
            d_sub = 0;
            d_tile += gstride;
            dl = locate(d_tile);
            dbase = d_base();
            d_wlim = p.W - 1 - dl.x0;
        }
    };

    // ---- B operand: lane (pixel n, half h) reads 16-byte group (2 q + h) (fp32 data on the fp16 path: 4 q + 2 h + r) of its pixel ----
    const int swz = S == 128 ? ((n >> 1) & 7) : ((n >> 2) & 3);
    const int b_lane = n * S + ((((C::PREC == 1) ? 2 * h : h) ^ swz) << 4);
    f32x4 nxt[C::NCUR], cur[C::NCUR];
    auto read_b = [&](int slot) {
        const unsigned char *base = lds8 + ring + slot * C::SUBB;
#pragma unroll
        for (int qq = 0; qq < QPS; ++qq)
#pragma unroll
            for (int r = 0; r < RPQ; ++r) {
                const int gconst = (C::PREC == 1) ? 4 * qq + r : 2 * qq;
                nxt[qq * RPQ + r] = *reinterpret_cast<const f32x4 *>(base + (b_lane ^ (gconst << 4)));
            }
    };

    // ---- prologue ----
#pragma unroll
    for (int j = 0; j < D; ++j) issue_dma();
    vc_wait_vmcnt<0>();                               // this wave's weight pieces and its first D sub-rows
    __syncthreads();                                  // every wave's weight pieces (the only barrier of the kernel)
    read_b(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_dma();                                      // sub-row D into the slot sub-row 0 has left
    int slot = 0;                                     // ring slot of the sub-row being contracted
    int steps = 0;                                    // steps this wave has taken (saturates at D)

    constexpr bool has_res = RES != 0;                // a template parameter: the residual registers must not pass through a phi
    constexpr bool res_half = RES == 2;
    const int osz = p.out_f16 ? 2 : 4;           // (fp32 instances: the GDN / IGDN instance may store half, conv_api.hip)
    // split output ([n][c/8][h][w][3][8] bf16, image stride out_sn in BYTES; fp32 instances): a TEMPLATE parameter -- the counted waits
    // differ with it, and a run-time branch around a wait is what tools/check_inflight_regs.py rightly refuses
    constexpr bool osp = OSP;
    static_assert(!OSP || !F16, "split output: fp32 instances");
    const bool half_plain = osz == 2 && !has_res && !p.chscale && (p.out_sw % 8) == 0 && (p.out_sh % 8) == 0 && (p.out_sn % 8) == 0;
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const float *const bias_l = reinterpret_cast<const float *>(lds8 + C::BIAS_OFF);
    unsigned char *const scr = lds8 + ring + D * C::SUBB;
    const int lane16 = lane * 16;
    const unsigned char *const res_b = reinterpret_cast<const unsigned char *>(p.res);
    unsigned char *const out_b = reinterpret_cast<unsigned char *>(p.out);
    const int out_pix_bytes = (int)p.out_sw * osz, res_pix_bytes = (int)p.res_sw * (res_half ? 2 : 4);

    for (unsigned tile = gw; tile < ntiles; tile += gstride) {
        const Loc tl = locate(tile);
        const int wlim = p.W - 1 - tl.x0;
        // Read-back layout of the epilogue: lane = 16-byte group rq of pixel 8 j + rpx of a 16-pixel part.  A pixel past the
        // end of the row takes the place of the row's last pixel: its input was that pixel's (the DMA clamps the same way),
        // so it computes, and stores, the same bytes -- every store is issued, nothing needs a mask or a dump page.
        int pxc[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int j = 0; j < 2; ++j) pxc[part][j] = min(part * 16 + 8 * j + rpx, wlim);
        constexpr bool res_jit = GDN != 0 && RES == 1;     // residual requested one 16-pixel unit ahead (GDN instances)
        f32x4 rv[res_jit ? 2 : C::NG32][2];
        f32x2 rvh[C::NG32][2];                    // a half-precision residual: 4 halves per lane and load
        f32x4 xk[GDN ? NSUB : 1][GDN ? C::NCUR : 1];       // GDN: the tile's own input, B-operand layout == accumulator layout
        // ---- accumulators start at the bias ----
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(&bias_l[t * 32 + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
            }
        // ---- contraction, one sub-row per step ----
        static_for<0, NSUB>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int i = 0; i < C::NCUR; ++i) cur[i] = nxt[i];
            if constexpr (GDN != 0) {
#pragma unroll
                for (int i = 0; i < C::NCUR; ++i) {
                    xk[s][i] = cur[i];
                    cur[i] = cur[i] * cur[i];                    // VC_IN_SQUARE (stage_chunk of conv_mfma.h squares the same way)
                }
            }
            if constexpr (C::nb(s) == 0) {
                vc_wait_vmcnt<C::wait_a(s, false, false)>();
            } else {
                const bool young = steps < D;
                if (C::NG16 != C::NG32 && half_plain) {
                    if (young) vc_wait_vmcnt<C::wait_a(s, true, true)>();
                    else vc_wait_vmcnt<C::wait_a(s, true, false)>();
                } else if constexpr (osp) {
                    if (young) vc_wait_vmcnt<C::wait_a(s, false, true, true)>();
                    else vc_wait_vmcnt<C::wait_a(s, false, false, true)>();
                } else {
                    if (young) vc_wait_vmcnt<C::wait_a(s, false, true)>();
                    else vc_wait_vmcnt<C::wait_a(s, false, false)>();
                }
            }
            steps = min(steps + 1, D);
            slot = slot + 1 == D ? 0 : slot + 1;
            read_b(slot);                                            // the next sub-row's operands ...
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... are in registers: its slot takes sub-row u+1+D
            issue_dma();
            if constexpr (s == NSUB - 1 && has_res && !res_jit) {
                // residual of this tile: requested behind the tile's last refill, used in the epilogue (see PwsCfg)
                const unsigned char *rbase =
                    res_b + ((long long)tl.img * p.res_sn + (long long)tl.y * p.res_sh + (long long)tl.x0 * p.res_sw) * (res_half ? 2 : 4);
#pragma unroll
                for (int g = 0; g < C::NG32; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned off = (unsigned)(pxc[g & 1][j] * res_pix_bytes + ((g >> 1) * 32 + 4 * rq) * (res_half ? 2 : 4));
                        if constexpr (res_half) rvh[g][j] = pws_load8(rbase, off);
                        else rv[g][j] = pws_load16(rbase, off);
                    }
            }
            if constexpr (!F16) {
                // fp32: a weight fragment feeds four 64-cycle MFMAs; fetch fragment f+1 while fragment f is consumed (left to
                // itself the scheduler issues the read right in front of its first use and every fourth MFMA waits for LDS)
                constexpr int NF = QPS * NT;
                auto wfrag = [&](int f) {
                    return *reinterpret_cast<const f32x4 *>(
                        __builtin_assume_aligned(lds8 + lane16 + ((f % NT) * KQ + s * QPS + f / NT) * 1024, 16));
                };
                f32x4 wf[2];
                wf[0] = wfrag(0);
                if constexpr (NF > 1) wf[1] = wfrag(1);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, NF>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[f % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[f & 1][e], cur[f / NT][e], acc[f % NT], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);             // (sched_group_barrier patterns were ignored here: pin the order)
                    if constexpr (f + 2 < NF) {
                        wf[f & 1] = wfrag(f + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            } else {
#pragma unroll
                for (int qq = 0; qq < QPS; ++qq) {
                    constexpr int q0 = s * QPS;
                    f32x4 bop = cur[qq * RPQ];
                    if constexpr (C::PREC == 1) {
                        f16x8 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hv[e] = (_Float16)cur[qq * RPQ][e];
                            hv[4 + e] = (_Float16)cur[qq * RPQ + 1][e];
                        }
                        bop = __builtin_bit_cast(f32x4, hv);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x4 wf = *reinterpret_cast<const f32x4 *>(
                            __builtin_assume_aligned(lds8 + lane16 + (t * KQ + q0 + qq) * 1024, 16));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, bop), acc[t],
                                                                        0, 0, 0);
                    }
                }
            }
        });

        // ---- epilogue: activation -> channel gain -> residual, 16 pixels x 128 bytes at a time through the wave's scratch ----
        unsigned char *const obase = out_b + ((long long)tl.img * p.out_sn + (long long)tl.y * p.out_sh + (long long)tl.x0 * p.out_sw) * osz;
        if (half_plain) {
            // half-precision output without residual / gain: the values are final in the accumulator layout, so the exchange
            // carries halves and a read-back lane stores 8 consecutive channels (16 bytes)
            static_for<0, C::NG16>([&](auto gc) {
                constexpr int g = decltype(gc)::value, tp = g >> 1, part = g & 1;
                constexpr bool pair = 2 * tp + 1 < NT;               // an odd last N-tile fills half a line: lanes 4-7 repeat lanes 0-3
                const int row = ((n >> 4) == part) ? (n & 15) : 16;
#pragma unroll
                for (int tt = 0; tt < (pair ? 2 : 1); ++tt)
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        f16x4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[2 * tp + tt][4 * gg + e];
                            v = v >= 0.0f ? v : v * neg;
                            hv[e] = (_Float16)v;
                        }
                        *reinterpret_cast<f16x4 *>(scr + row * (VC_EPI_ROWF * 4) + (tt * 32 + 8 * gg + 4 * h) * 2) = hv;
                    }
                const int rqe = pair ? rq : (rq & 3);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(scr + (8 * j + rpx) * (VC_EPI_ROWF * 4) + rqe * 16);
                    *reinterpret_cast<f32x4 *>(obase + (pxc[part][j] * out_pix_bytes + (tp * 64 + 8 * rqe) * 2)) = v;
                }
            });
        } else {
            const unsigned char *const rbase_jit =
                res_b + ((long long)tl.img * p.res_sn + (long long)tl.y * p.res_sh + (long long)tl.x0 * p.res_sw) * 4;
            auto issue_res = [&](auto gc) {          // (GDN instances) the two residual loads of 16-pixel unit g
                constexpr int g = decltype(gc)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rv[g & 1][j] = pws_load16(rbase_jit, (unsigned)(pxc[g & 1][j] * res_pix_bytes + ((g >> 1) * 32 + 4 * rq) * 4));
            };
            if constexpr (res_jit) issue_res(std::integral_constant<int, 0>{});
            static_for<0, C::NG32>([&](auto gc) {
                constexpr int g = decltype(gc)::value, t = g >> 1, part = g & 1;
                // (no register operand on the wait itself: a tied operand invites a copy of the in-flight registers in FRONT of
                //  it.  The empty statement BEHIND it takes them instead: asm volatile statements keep their order, a copy made
                //  for its operands sits between the two, and every use below depends on its result.)
                if constexpr (res_jit) {
                    // unit g + 1 is requested before unit g is waited for: younger than unit g's loads are then the two stores
                    // of unit g - 1 and the two loads of unit g + 1
                    if constexpr (g + 1 < C::NG32) issue_res(std::integral_constant<int, g + 1>{});
                    vc_wait_vmcnt<(g + 1 < C::NG32 ? 2 : 0) + (g > 0 ? (osp ? 6 : 2) : 0)>();
                    asm volatile("" : "+v"(rv[g & 1][0]), "+v"(rv[g & 1][1]));
                } else if constexpr (has_res) {
                    vc_wait_vmcnt<osp ? C::wait_res_sp(g) : C::WAIT_RES>();
                    if constexpr (res_half) asm volatile("" : "+v"(rvh[g][0]), "+v"(rvh[g][1]));
                    else asm volatile("" : "+v"(rv[g][0]), "+v"(rv[g][1]));
                }
                const int row = ((n >> 4) == part) ? (n & 15) : 16;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    f32x4 v = {acc[t][4 * gg], acc[t][4 * gg + 1], acc[t][4 * gg + 2], acc[t][4 * gg + 3]};
                    if constexpr (GDN != 0) {        // conv_epilogue MODE 1 / 2: IEEE sqrt and divide, like the CPU path's x * rsqrt(norm)
                        const f32x4 x = xk[t][gg];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (GDN == 1) ? x[e] * (1.0f / sqrtf(v[e])) : x[e] * sqrtf(v[e]);
                    }
                    *reinterpret_cast<f32x4 *>(scr + row * (VC_EPI_ROWF * 4) + (8 * gg + 4 * h) * 4) = v;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = *reinterpret_cast<const f32x4 *>(scr + (8 * j + rpx) * (VC_EPI_ROWF * 4) + rq * 16);
                    f32x4 r = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (res_half) {
                        const f16x4 rh = __builtin_bit_cast(f16x4, rvh[g][j]);
                        r = f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                    } else if constexpr (has_res) {
                        r = rv[res_jit ? (g & 1) : g][j];
                    }
                    if constexpr (has_res) { if (p.res_first) v += r; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                    if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(lds8 + C::GAIN_OFF + (t * 32 + 4 * rq) * 4);
                    if constexpr (has_res) { if (!p.res_first) v += r; }
                    unsigned char *dst = obase + (pxc[part][j] * out_pix_bytes + (t * 32 + 4 * rq) * osz);
                    if constexpr (osp) {
                        // the three bf16 pieces of the 4 channels into the pixel's record of group (4 t + rq / 2), half rq & 1
                        vc_store_split4(out_b + (long long)tl.img * p.out_sn +
                                            (((long long)(4 * t + (rq >> 1)) * p.H + tl.y) * p.W + tl.x0 + pxc[part][j]) * 48, rq & 1, v);
                    } else if (osz == 2) {
                        const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *reinterpret_cast<f16x4 *>(dst) = hv;
                    } else {
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
                }
            });
        }
    }
    vc_wait_vmcnt<0>();
}

template <class C, int RES, int GDN = 0, bool OSP = false> int launch_pws_res(hipStream_t st, const ConvArgs &a)
{
    if constexpr (!OSP && !C::F16 && RES != 2) {
        if (a.out_sp3) return launch_pws_res<C, RES, GDN, true>(st, a);
    }
    auto kern = conv_pws_kernel<C, RES, GDN, OSP>;
    static vc_lds_raised raised;
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), C::LDS_BYTES, raised)) return VC_ELAUNCH;
    const long long ntiles = (long long)a.N * a.H * ((a.W + 31) / 32);
    const long long blocks = ntiles < 256 ? ntiles : 256;       // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), C::LDS_BYTES, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}

template <class C> int launch_pws(hipStream_t st, const ConvArgs &a)
{
    if constexpr (!C::F16 && C::KQ == 16 && C::NT == 4) {          // GDN / IGDN: the fp32 128 -> 128 instance
        if (a.epi == VC_EPI_GDN) return a.res ? launch_pws_res<C, 1, 1>(st, a) : launch_pws_res<C, 0, 1>(st, a);
        if (a.epi == VC_EPI_IGDN) return a.res ? launch_pws_res<C, 1, 2>(st, a) : launch_pws_res<C, 0, 2>(st, a);
    }
    if (a.epi != VC_EPI_NONE) return VC_EINVAL;
    if (!a.res) return launch_pws_res<C, 0>(st, a);
    if constexpr (C::F16) {
        if (a.res_f16) return launch_pws_res<C, 2>(st, a);
    }
    return a.res_f16 ? VC_EINVAL : launch_pws_res<C, 1>(st, a);
}

template <int KQ, int PREC> int by_nt(hipStream_t st, const ConvArgs &a)
{
    switch ((a.Cout + 31) / 32) {
    case 1: return launch_pws<PwsCfg<KQ, 1, PREC>>(st, a);
    case 2: return launch_pws<PwsCfg<KQ, 2, PREC>>(st, a);
    case 3: return launch_pws<PwsCfg<KQ, 3, PREC>>(st, a);
    case 4: return launch_pws<PwsCfg<KQ, 4, PREC>>(st, a);
    }
    return VC_EINVAL;
}

template <int PREC> int by_kq(hipStream_t st, const ConvArgs &a)
{
    constexpr int KCH = PREC ? 16 : 8;
    switch (a.Cin / KCH) {                          // 32, 64, 96 or 128 channels
    case 1 * (32 / KCH): return by_nt<1 * (32 / KCH), PREC>(st, a);
    case 2 * (32 / KCH): return by_nt<2 * (32 / KCH), PREC>(st, a);
    case 3 * (32 / KCH): return by_nt<3 * (32 / KCH), PREC>(st, a);
    case 4 * (32 / KCH): return by_nt<4 * (32 / KCH), PREC>(st, a);
    }
    return VC_EINVAL;
}

}  // namespace

bool conv_pws_eligible(const ConvArgs &a, int k, int stride, bool f16)
{
    if (k != 1 || stride != 1 || a.out_mode != VC_OUT_PLAIN) return false;
    if (a.epi != VC_EPI_NONE) {
        // GDN / IGDN: fp32, 128 -> 128, the layer's own input as the multiplicand (what compressai's GDN computes), no gain
        const bool gdn = (a.epi == VC_EPI_GDN || a.epi == VC_EPI_IGDN) && !f16 && a.in_xform == VC_IN_SQUARE && a.Cin == 128 && a.Cout == 128 &&
                         a.mul == a.in && a.mul_sn == a.in_sn && a.mul_sh == a.in_sh && a.mul_sw == a.in_sw && a.act == VC_ACT_NONE &&
                         !a.chscale && !a.res_first && !a.res_f16;
        if (!gdn) return false;
    } else if (a.in_xform != VC_IN_NONE) {
        return false;
    }
    if (a.act != VC_ACT_NONE && a.act != VC_ACT_RELU && a.act != VC_ACT_LRELU) return false;
    if (!a.vec4 || !a.vec_out || (a.Cout % 32) || a.Cout > 128 || a.Cin > 128 || (a.Cin % 32)) return false;
    if ((long long)a.N * a.H * ((a.W + 31) / 32) >= (1ll << 31) || a.in_sw * 4 * 32 >= (1ll << 31) || a.res_sw * 4 * 32 >= (1ll << 31) ||
        a.out_sw * 4 * 32 >= (1ll << 31)) return false;
    if (f16 && a.in_f16 && ((a.in_sw % 8) || (a.in_sh % 8) || (a.in_sn % 8) || ((uintptr_t)a.in % 16))) return false;
    if (!f16 && (a.in_f16 || (a.out_f16 && a.epi == VC_EPI_NONE))) return false;
    if (a.out_sp3 && (f16 || a.out_f16 || (a.Cout % 8))) return false;      // split output: the fp32 instances
    return true;
}

int conv_dispatch_pws(hipStream_t st, const ConvArgs &a, bool f16)
{
    if (!f16) return by_kq<0>(st, a);
    return a.in_f16 ? by_kq<2>(st, a) : by_kq<1>(st, a);
}
