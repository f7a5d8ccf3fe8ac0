// Implicit-GEMM convolution on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
//
//   GEMM view:  M = output pixels (one M-tile = MT consecutive x of one output row)
//               N = output channels (N-tile = MT channels)
//               K = taps x input channels, walked as  channel-chunk(CK) -> tap -> k-step(KS)
//
//   A (pixels x k): a TH x TW output tile's input footprint (with halo) is staged once per channel
//       chunk into LDS as [row][col][CK+4] floats.  The +4 pad makes the lane stride an odd number
//       of 16-byte slots, so the per-lane ds_read_b128 (4 consecutive channels) is conflict-free.
//       Stride-2 convolutions de-interleave even/odd columns while staging so that the lanes of an
//       M-tile still read consecutive LDS columns.
//   B (k x channels): weights are re-packed once at model load into *fragment order*
//       [n-tile][tap][k-step][lane][4] so that each wave fetches a B fragment with one fully
//       coalesced 1 KiB global_load_dwordx4 straight from L2 -- no LDS round trip, no barrier.
//   One ds_read_b128 + one global_load_dwordx4 feed 4 MFMAs (the 4 floats are 4 k-sub-steps).
//
//   The fp32 MFMA is bit-for-bit an fp32 FMA chain (cdna guide section 3), so results differ from the
//   PyTorch-CPU reference only by summation order.
#pragma once
#include "common.h"

#include <type_traits>
#include <utility>

// Tile order inside an image: bands of `band_rows` tile rows (ConvArgs::tile_band, 8) walked column by column, so that the workgroups in flight
// together on an XCD (consecutive ids) form a ~band x (64/band) block of tiles: a tile's vertical halo is wanted by the
// NEXT id and its horizontal halo `band` ids later -- both while the lines are still in that XCD's L2 -- instead of a whole
// tile row (60 tiles at 1080p) later, when they have to come back from the Infinity Cache / HBM.
__device__ __forceinline__ void vc_tile_xy(int t, int tiles_x, int tiles_y, int band_rows, int &tx, int &ty)
{
    const int per_band = band_rows * tiles_x;
    const int band = t / per_band, r = t - band * per_band;
    const int rows = min(band_rows, tiles_y - band * band_rows);     // the last band may be shorter
    tx = r / rows;
    ty = band * band_rows + (r - tx * rows);
}

// compile-time loop: keeps accumulator indices static so the tiles stay in registers
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(std::forward<F>(f));
    }
}

#ifdef VC_STAMPS
// Diagnostic build only (make stamps): per-phase shader-clock totals summed over all waves, read back with
// vc_debug_read_stamps.  Never compiled into libvc_hip.so.
__device__ unsigned long long g_vc_stamps[8];
__device__ int g_vc_skip;
#define VC_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define VC_ACC(slot, t1, t0) do { if ((threadIdx.x & 63) == 0 && !(g_vc_skip & 8)) atomicAdd(&g_vc_stamps[slot], (t1) - (t0)); } while (0)
// (bit 3 of g_vc_skip switches the stamps themselves off: their atomics on eight hot addresses cost several ms per launch)
// phase knock-out (vc_debug_set_skip; tools/stamps.py --knockout): bit 0 = stage only the first channel chunk, bit 1 = no
// contraction, bit 2 = no epilogue.  Results are garbage by design; the kernel time that remains is what the other phases cost.
#define VC_SKIP(bit) (g_vc_skip & (bit))
#else
#define VC_T(var)
#define VC_ACC(slot, t1, t0)
#define VC_SKIP(bit) false
#endif

struct ConvArgs {
    const float *in;
    long long in_sn, in_sh, in_sw;
    int N, H, W, Cin;
    float *out;
    long long out_sn, out_sh, out_sw;
    int Ho, Wo, Cout;
    const float *wpk;
    const float *bias;
    const float *res;
    long long res_sn, res_sh, res_sw;
    const float *mul;
    long long mul_sn, mul_sh, mul_sw;
    const float *chscale;
    int cin_pad;       // Cin rounded up to the channel chunk
    int tiles_x, tiles_y, nblks, total_blocks;
    int act;
    float slope;
    int epi, in_xform, out_mode, vec4;
    int vec_out;   // out / res / mul / chscale allow 16-byte accesses on groups of 4 consecutive output channels
    int in_f16, out_f16;   // fp16 path only: `in` / `out` point at half-precision tensors (strides in elements)
    int res_first;         // add the residual BEFORE the activation (plain/ReLU/LeakyReLU epilogues only)
    int tile_band;         // tile rows per band of the 2-D tile order (vc_tile_xy)
    int res_f16;           // VC_CFG_PWS on the fp16 path only: `res` points at a half-precision tensor
    int pack128;           // VC_CFG_PACK128: weights and bias are padded to whole blocks of 128 output channels
    const void *tail_wpk;  // VC_CFG_DMA only: fused trailing 1x1 layer (vc_conv_pack_tail_f16), NULL = none
    const float *tail_bias;
    // split tensors ([n][c/8][h][w][3 pieces][8] bf16, conv_split.h; image strides in_sn / out_sn / res_sn then count BYTES):
    // in_sp3: VC_CFG_SPLIT only; out_sp3: VC_CFG_SPLIT and the classic fp32 instances; res_sp3: VC_CFG_SPLIT only
    int in_sp3, out_sp3, res_sp3;
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int MT> struct Mfma;
// arow/akk: which pixel of the M-tile and which k-group a lane feeds as the A operand;
// col: which output channel of the N-tile a lane owns in C/D; row(reg, lane): the pixel of accumulator `reg`.
template <> struct Mfma<32> {
    typedef f32x16 acc_t;
    static constexpr int NREG = 16, KS = 8, NT = 32;
    static __device__ __forceinline__ int arow(int lane) { return lane & 31; }
    static __device__ __forceinline__ int akk(int lane) { return lane >> 5; }
    static __device__ __forceinline__ int col(int lane) { return lane & 31; }
    // C/D with the WEIGHT fragment as the first MFMA operand: lane -> pixel, register -> output channel
    static __device__ __forceinline__ int px(int lane) { return lane & 31; }
    static __device__ __forceinline__ int crow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
};
template <> struct Mfma<16> {
    typedef f32x4 acc_t;
    static constexpr int NREG = 4, KS = 16, NT = 16;
    static __device__ __forceinline__ int arow(int lane) { return lane & 15; }
    static __device__ __forceinline__ int akk(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int px(int lane) { return lane & 15; }
    static __device__ __forceinline__ int crow(int reg, int lane) { return (lane >> 4) * 4 + reg; }
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int reg, int lane) { return (lane >> 4) * 4 + reg; }
};

// v_mfma_f32_4x4x1_16B_f32: 16 independent (4 pixel x 4 channel) outer products per instruction, k = 1, at the
// full fp32 matrix rate.  M-tile = 64 consecutive pixels (one per lane), N-tile = 4 output channels: the
// layers that end in 1-4 channels (SPyNet's 16->2 flow head, the mask net's 32->1, U-Net heads) would
// otherwise burn a 16-wide tile on mostly padding.  A (pixels): lane 4b+i; B (weights): lane 4b+j holds
// channel j (replicated over the 16 blocks at pack time); D: lane 4b+j, register i = pixel 4b+i.
template <> struct Mfma<64> {
    typedef f32x4 acc_t;
    static constexpr int NREG = 4, KS = 4, NT = 4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int arow(int lane) { return lane; }
    static __device__ __forceinline__ int akk(int) { return 0; }
    static __device__ __forceinline__ int col(int lane) { return lane & 3; }
    static __device__ __forceinline__ int px(int lane) { return lane; }
    static __device__ __forceinline__ int crow(int reg, int) { return reg; }
    static __device__ __forceinline__ int row(int reg, int lane) { return (lane >> 2) * 4 + reg; }
};

// Tile configuration: 4 waves split the M-tiles of a TH x (XT*MT) output tile; every wave covers all
// WN N-tiles of the block (BN = WN*MT output channels).
template <int MT_, int TH_, int XT_, int WM_, int WN_, int WAVES_N_ = 1> struct TileCfg {
    static constexpr int MT = MT_, TH = TH_, XT = XT_, WM = WM_, WN = WN_;
    // the 4 waves form a (4/WAVES_N) x WAVES_N grid over (M-tiles, N-tiles)
    static constexpr int WAVES_N = WAVES_N_, WAVES_M = 4 / WAVES_N_;
    // waves per SIMD the register allocation must leave room for: the 128-channel block needs ~250
    // registers (2 waves); the narrower ones are held to 168 so that three workgroups share a CU and
    // one of them can always feed the matrix pipe while another stages or stores
    static constexpr int MIN_WAVES = (WM_ * WN_ >= 4) ? 2 : 3;
    static constexpr int NT = Mfma<MT_>::NT;
    static constexpr int TW = XT * MT, BN = WAVES_N * WN * NT;
    static_assert(TH * XT == WAVES_M * WM, "WAVES_M waves x WM M-tiles must cover the tile");
};
typedef TileCfg<32, 8, 1, 2, 4> CfgN128;
typedef TileCfg<32, 8, 1, 2, 2> CfgN64;
typedef TileCfg<32, 8, 1, 2, 1> CfgN32;
typedef TileCfg<16, 8, 2, 4, 1> CfgN16;
typedef TileCfg<64, 8, 1, 2, 1> CfgN4;
// 16 rows x 32 pixels x 32 channels for the 7x7 layers: a 22 x 38 input footprint per 16 x 32 outputs instead of
// 14 x 38 per 8 x 32 (22 % less staged volume per output; 67 KB of LDS, two workgroups per CU).  7x7 64->32
// @4x1088x1920: 12.10 -> 11.86 ms fp32 (89.9 % of peak), 2.25 -> 2.00 ms on the fp16 path.  (The 64-channel variant
// spills; 4-row tiles are slower.)
typedef TileCfg<32, 16, 1, 4, 1> CfgN32T16;
typedef TileCfg<32, 8, 1, 4, 2, 2> CfgN128b;   // 2x2 waves: each wave 4 rows x 64 channels (half the B-fragment loads)

// fp16 path: depth of the register ring that holds prefetched weight fragments.  One k-step is only 32*WM*WN matrix
// cycles there (an fp32 step is 8x longer), far less than an L2 round trip, so the fragment of step g is requested D
// steps ahead.  D divides the steps of a kernel row (ring slots stay compile-time constants across the rolled ky loop)
// and is capped by what the accumulators leave of the register file.
constexpr int vc_ring_depth(int steps_x, int free_regs, int wn, int cap = 8)
{
    int budget = free_regs / (4 * wn);
    if (budget > cap) budget = cap;
    int d = 1;
    for (int c = 2; c <= budget; ++c)
        if (steps_x % c == 0) d = c;
    return d;
}
// registers the classic kernel can spare for the ring next to accumulators, two A fragments and the staging batch
constexpr int vc_ring_regs_classic(int wm, int wn, int nreg, int min_waves)
{
    return (min_waves >= 3 ? 112 : 176) - wm * wn * nreg - 8 * wm;
}

template <int KH, int KW, int S, int CK, class C> struct ConvGeom {
    static constexpr bool POINT = (KH == 1 && KW == 1);
    static constexpr int LS = POINT ? 1 : S;                  // stride as seen by the LDS image
    static constexpr int ROWS_IN = (C::TH - 1) * LS + KH;
    static constexpr int COLS_IN = (C::TW - 1) * LS + KW;
    static constexpr int HALF = (COLS_IN + 1) / 2;
    static constexpr int COLS_L = (LS == 2) ? 2 * HALF : COLS_IN;
    static constexpr int CKP = CK + 4;
    static constexpr int LDS_FLOATS = ROWS_IN * COLS_L * CKP;
    static constexpr int KS = Mfma<C::MT>::KS;
    static constexpr int KSTEPS = CK / KS;
    static_assert(CK % KS == 0, "chunk must be a whole number of k-steps");
    static_assert(LDS_FLOATS * 4 + 16 <= 80 * 1024, "keep two workgroups per CU (160 KiB LDS)");
};

__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    if (act == VC_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == VC_ACT_LRELU) return v >= 0.0f ? v : v * slope;
    if (act == VC_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == VC_ACT_CLAMP01) return fminf(fmaxf(v, 0.0f), 1.0f);
    return v;
}

// INH: the input tensor itself is stored in half precision (VC_CFG_IN_F16): an item is ONE 16-byte load of 8
// channels and needs no conversion -- the producer's epilogue already rounded exactly as this stage would have.
template <int KH, int KW, int S, int CK, class C, bool VEC, bool F16, bool INH = false>
__device__ __forceinline__ void stage_chunk(const ConvArgs &p, float *lds, const float *in_img, int c0, int oy0, int ox0,
                                            int iy0, int ix0, int tid)
{
    typedef ConvGeom<KH, KW, S, CK, C> G;
    constexpr int C4 = CK / 4;                                   // 16-byte LDS items per pixel
    constexpr int CPI = F16 ? 8 : 4;                             // input channels behind one item
    constexpr int ITEMS = G::ROWS_IN * G::COLS_IN * C4;
    constexpr int IPT = (ITEMS + 255) / 256;                    // items per thread
    constexpr int BATCH0 = (IPT + 1) / 2 > 8 ? 8 : (IPT + 1) / 2;   // two rounds per chunk when registers allow
    constexpr int BATCH = (F16 && !INH) ? (BATCH0 > 4 ? 4 : BATCH0) : BATCH0; // (an fp16 item from fp32 data is two 16-byte loads)
#pragma unroll
    for (int b0 = 0; b0 < IPT; b0 += BATCH) {
        f32x4 v[BATCH], v2[(F16 && !INH) ? BATCH : 1];
        int dst[BATCH];
        bool ok[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (b0 + j >= IPT) continue;
            const int idx = tid + (b0 + j) * 256;
            const int c4 = idx % C4;
            const int pc = idx / C4;
            const int col = pc % G::COLS_IN, row = pc / G::COLS_IN;
            const int iy = G::POINT ? (oy0 + row) * S : iy0 + row;
            const int ix = G::POINT ? (ox0 + col) * S : ix0 + col;
            const int ch = c0 + c4 * CPI;
            ok[j] = (idx < ITEMS) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            const int pos = (G::LS == 2) ? ((col & 1) * G::HALF + (col >> 1)) : col;
            // items past the end of the footprint (last round only) go to a 16-byte dump slot behind the tile:
            // no branch anywhere in the staging code, so every round's loads issue back to back
            dst[j] = (idx < ITEMS) ? (row * G::COLS_L + pos) * G::CKP + c4 * 4 : G::LDS_FLOATS;
            const float *src = in_img + (long long)iyc * p.in_sh + (long long)ixc * p.in_sw;
            if (INH) {   // in_img counts halves here (see the caller)
                const _Float16 *q = reinterpret_cast<const _Float16 *>(in_img) + (long long)iyc * p.in_sh +
                                    (long long)ixc * p.in_sw + min(ch, p.Cin - 8);
                v[j] = *reinterpret_cast<const f32x4 *>(q);
            } else if (F16) {   // Cin % 8 == 0 guaranteed by the host
                const float *q = src + min(ch, p.Cin - 8);
                v[j] = *reinterpret_cast<const f32x4 *>(q);
                v2[j] = *reinterpret_cast<const f32x4 *>(q + 4);
            } else if (VEC) {
                v[j] = *reinterpret_cast<const f32x4 *>(src + min(ch, p.Cin - 4));
            } else {
                const int cl = p.Cin - 1;
                v[j].x = src[min(ch, cl)];
                v[j].y = src[min(ch + 1, cl)];
                v[j].z = src[min(ch + 2, cl)];
                v[j].w = src[min(ch + 3, cl)];
                if (ch + 1 >= p.Cin) v[j].y = 0.f;
                if (ch + 2 >= p.Cin) v[j].z = 0.f;
                if (ch + 3 >= p.Cin) v[j].w = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (b0 + j >= IPT) continue;
            f32x4 w = v[j];
            if (INH) {
            } else if (F16) {   // 8 channels -> 8 halves = one 16-byte item
                f16x8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (_Float16)v[j][e];
                    h[4 + e] = (_Float16)v2[j][e];
                }
                w = __builtin_bit_cast(f32x4, h);
            } else if (p.in_xform == VC_IN_SQUARE) {
                w = w * w;
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            w = ok[j] ? w : z;
            *reinterpret_cast<f32x4 *>(&lds[dst[j]]) = w;
        }
    }
}

// ---- epilogue: (GDN) -> activation -> channel gain -> residual -> store (plain / pixel-shuffle) ----
template <class C, bool F16>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &p, typename Mfma<C::MT>::acc_t (&acc)[C::WM][C::WN], int nblk, int wm,
                                              int wn, int lane, int oy0, int ox0, int img)
{
    typedef Mfma<C::MT> M;
    constexpr int MT = C::MT, WM = C::WM, WN = C::WN, NT = C::NT;
    // ---- epilogue: (GDN) -> activation -> channel gain -> residual -> store (plain / pixel-shuffle) ----
    // One lane owns one pixel of the M-tile and, per register quad, 4 consecutive output channels: the residual /
    // GDN-input loads and the store are single 16-byte accesses.  The mode (GDN / IGDN / sigmoid / clamp / plain)
    // is resolved ONCE per wave: a per-element switch unrolled over 128 accumulators cost ~40k instructions.
    const int pxl = M::px(lane);
    auto epilogue = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;   // 0 plain/relu/lrelu, 1 GDN, 2 IGDN, 3 sigmoid, 4 clamp01
        // plain / ReLU / LeakyReLU share one formula: v >= 0 ? v : v * neg
        const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
        static_for<0, WM>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int m = wm * WM + t;
            const int oy = oy0 + m / C::XT;
            const int ox = ox0 + (m % C::XT) * MT + pxl;
            const bool pix_ok = (oy < p.Ho) && (ox < p.Wo);
            const long long mul_pix = (long long)img * p.mul_sn + (long long)oy * p.mul_sh + (long long)ox * p.mul_sw;
            static_for<0, WN>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
#pragma unroll
                for (int g = 0; g < M::NREG / 4; ++g) {
                    const int co = nblk * C::BN + (wn * WN + n) * NT + M::crow(4 * g, lane);   // first of 4 consecutive channels
                    f32x4 v = {acc[t][n][4 * g], acc[t][n][4 * g + 1], acc[t][n][4 * g + 2], acc[t][n][4 * g + 3]};
                    if (pix_ok && co < p.Cout) {
                        const int cps = p.Cout >> 2;
                        const bool ps = p.out_mode != VC_OUT_PLAIN;
                        const int pos = ps ? co / cps : 0;
                        const int cch = ps ? co - pos * cps : co;
                        const int sc = ps ? 2 : 1;
                        const int yy = sc * oy + (pos >> 1), xx = sc * ox + (pos & 1);
                        const long long o_off = (long long)img * p.out_sn + (long long)yy * p.out_sh + (long long)xx * p.out_sw + cch;
                        const long long r_off = (long long)img * p.res_sn + (long long)yy * p.res_sh + (long long)xx * p.res_sw + cch;
                        if (p.vec_out) {
                            if constexpr (MODE == 1 || MODE == 2) {
                                const f32x4 x = *reinterpret_cast<const f32x4 *>(p.mul + mul_pix + co);
#pragma unroll
                                for (int e = 0; e < 4; ++e)   // IEEE sqrt and divide, like the CPU path's x * rsqrt(norm)
                                    v[e] = (MODE == 1) ? x[e] * (1.0f / sqrtf(v[e])) : x[e] * sqrtf(v[e]);
                            }
                            if (MODE == 0 && p.res_first) v += *reinterpret_cast<const f32x4 *>(p.res + r_off);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if constexpr (MODE == 3) v[e] = 1.0f / (1.0f + expf(-v[e]));
                                else if constexpr (MODE == 4) v[e] = fminf(fmaxf(v[e], 0.0f), 1.0f);
                                else if constexpr (MODE == 0) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                            }
                            if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + co);
                            if (p.res && !(MODE == 0 && p.res_first)) v += *reinterpret_cast<const f32x4 *>(p.res + r_off);
                            if (F16 && p.out_f16) {
                                const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                                *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(p.out) + o_off) = hv;
                            } else if (!F16 && p.out_sp3) {
                                // a split-operand consumer behind this layer: the three bf16 pieces instead of the fp32 value
                                const int oh = sc * p.Ho, ow = sc * p.Wo;
                                vc_store_split4(reinterpret_cast<unsigned char *>(p.out) + (long long)img * p.out_sn +
                                                    ((((long long)(cch >> 3)) * oh + yy) * ow + xx) * 48, (cch >> 2) & 1, v);
                            } else {
                                *reinterpret_cast<f32x4 *>(p.out + o_off) = v;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (co + e < p.Cout) {
                                    float w = v[e];
                                    if constexpr (MODE == 1 || MODE == 2) {
                                        const float x = p.mul[mul_pix + co + e];
                                        w = (MODE == 1) ? x * (1.0f / sqrtf(w)) : x * sqrtf(w);
                                    }
                                    // (pixel-shuffle: a quad may straddle two output positions when cout/4 % 4 != 0)
                                    const int pe = ps ? (co + e) / cps : 0;
                                    const int ce = ps ? co + e - pe * cps : co + e;
                                    const int ye = sc * oy + (pe >> 1), xe = sc * ox + (pe & 1);
                                    const long long re = (long long)img * p.res_sn + (long long)ye * p.res_sh + (long long)xe * p.res_sw + ce;
                                    const bool first = MODE == 0 && p.res_first;
                                    if (first) w += p.res[re];
                                    if constexpr (MODE == 3) w = 1.0f / (1.0f + expf(-w));
                                    else if constexpr (MODE == 4) w = fminf(fmaxf(w, 0.0f), 1.0f);
                                    else if constexpr (MODE == 0) w = w >= 0.0f ? w : w * neg;
                                    if (p.chscale) w *= p.chscale[co + e];
                                    if (p.res && !first) w += p.res[re];
                                    const long long oe = (long long)img * p.out_sn + (long long)ye * p.out_sh + (long long)xe * p.out_sw + ce;
                                    if (F16 && p.out_f16) reinterpret_cast<_Float16 *>(p.out)[oe] = (_Float16)w;
                                    else p.out[oe] = w;
                                }
                            }
                        }
                    }
                }
            });
        });
    };
    // (the reference never combines GDN with an activation, nor sigmoid/clamp with GDN)
    if (p.epi == VC_EPI_GDN) epilogue(std::integral_constant<int, 1>{});
    else if (p.epi == VC_EPI_IGDN) epilogue(std::integral_constant<int, 2>{});
    else if (p.act == VC_ACT_SIGMOID) epilogue(std::integral_constant<int, 3>{});
    else if (p.act == VC_ACT_CLAMP01) epilogue(std::integral_constant<int, 4>{});
    else epilogue(std::integral_constant<int, 0>{});
}

// fp16 path, 32-wide tiles: the same epilogue with COALESCED stores.  In the accumulator layout a lane owns one pixel, so
// one store instruction touches 32 different 128-byte lines, 32 bytes each -- unnoticed behind an fp32 contraction, a
// fifth to a third of a workgroup's life behind an fp16 one (tools/stamps.py --fp16).  Each wave passes its 32 px x 32 ch
// accumulator tiles through a private 4.6 KB LDS scratch ([pixel][32 + 4 pad] floats: conflict-free 16-byte writes) and
// reads them back with consecutive lanes along the channels: lane i gets channels 4(i%8).. of pixel 8j + i/8, so a store
// instruction writes whole lines (8 consecutive pixels x 128 bytes for a 32-channel tensor).  The arithmetic per value is
// unchanged -> bit-identical to conv_epilogue.  Plain / ReLU / LeakyReLU / sigmoid / clamp epilogues with 16-byte aligned
// views (p.vec_out); the caller falls back to conv_epilogue otherwise.  `scratch`: this wave's 32 * 36 floats.
constexpr int VC_EPI_ROWF = 36;
constexpr int VC_EPI_SCRATCH_FLOATS = 32 * VC_EPI_ROWF;

template <class C>
__device__ __forceinline__ void conv_epilogue_coalesced(const ConvArgs &p, typename Mfma<C::MT>::acc_t (&acc)[C::WM][C::WN], int nblk,
                                                        int wm, int wn, int lane, int oy0, int ox0, int img, float *scratch)
{
    typedef Mfma<C::MT> M;
    constexpr int MT = C::MT, WM = C::WM, WN = C::WN, NT = C::NT;
    static_assert(MT == 32 && M::NREG == 16, "32 x 32 accumulator tiles");
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int mode = p.act == VC_ACT_SIGMOID ? 3 : (p.act == VC_ACT_CLAMP01 ? 4 : 0);
    const int wpx = lane & 31, whalf = lane >> 5;          // accumulator layout: pixel, channel half-group
    const int rq = lane & 7, rpx = lane >> 3;              // read-back layout: channel quad, pixel within a group of 8
    const int cps = p.Cout >> 2;
    const bool ps = p.out_mode != VC_OUT_PLAIN;
    const int sc = ps ? 2 : 1;
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int m = wm * WM + t;
        const int oy = oy0 + m / C::XT;
        static_for<0, WN>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[t][n][4 * g], acc[t][n][4 * g + 1], acc[t][n][4 * g + 2], acc[t][n][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(&scratch[wpx * VC_EPI_ROWF + 8 * g + 4 * whalf]) = v;
            }
            const int co = nblk * C::BN + (wn * WN + n) * NT + 4 * rq;   // first of this lane's 4 consecutive channels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = 8 * j + rpx;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&scratch[pix * VC_EPI_ROWF + 4 * rq]);
                const int ox = ox0 + (m % C::XT) * MT + pix;
                if (oy < p.Ho && ox < p.Wo && co < p.Cout) {
                    const int pos = ps ? co / cps : 0;
                    const int cch = ps ? co - pos * cps : co;
                    const int yy = sc * oy + (pos >> 1), xx = sc * ox + (pos & 1);
                    const long long o_off = (long long)img * p.out_sn + (long long)yy * p.out_sh + (long long)xx * p.out_sw + cch;
                    const long long r_off = (long long)img * p.res_sn + (long long)yy * p.res_sh + (long long)xx * p.res_sw + cch;
                    if (mode == 0 && p.res_first) v += *reinterpret_cast<const f32x4 *>(p.res + r_off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (mode == 3) v[e] = 1.0f / (1.0f + expf(-v[e]));
                        else if (mode == 4) v[e] = fminf(fmaxf(v[e], 0.0f), 1.0f);
                        else v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                    }
                    if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + co);
                    if (p.res && !(mode == 0 && p.res_first)) v += *reinterpret_cast<const f32x4 *>(p.res + r_off);
                    if (p.out_f16) {
                        const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(p.out) + o_off) = hv;
                    } else {
                        *reinterpret_cast<f32x4 *>(p.out + o_off) = v;
                    }
                }
            }
        });
    });
}

// Half-precision OUTPUT, two N-tiles at a time: a 32-channel tile is only 64 bytes of a half-precision pixel, so the
// function above would write every 128-byte line in two halves, by two instructions of 8 bytes per lane (3x3 128->128
// @4x544x960: the epilogue alone took 0.33 ms for 535 MB -- half the rate of the 7x7 64->32 layer, whose 32-channel
// pixels are whole lines side by side).  Here a PAIR of N-tiles (64 channels = 128 bytes) passes through a
// [pixel][64 + 4] scratch and every lane converts and stores 8 consecutive channels = 16 bytes: an instruction writes 8
// whole lines.  Same arithmetic per value.  Plain output layout only (no pixel shuffle); `scratch`: 32 * 68 floats per wave.
constexpr int VC_EPI2_ROWF = 68;
constexpr int VC_EPI2_SCRATCH_FLOATS = 32 * VC_EPI2_ROWF;

template <class C>
__device__ __forceinline__ void conv_epilogue_coalesced_h2(const ConvArgs &p, typename Mfma<C::MT>::acc_t (&acc)[C::WM][C::WN], int nblk,
                                                           int wm, int wn, int lane, int oy0, int ox0, int img, float *scratch)
{
    typedef Mfma<C::MT> M;
    constexpr int MT = C::MT, WM = C::WM, WN = C::WN, NT = C::NT;
    static_assert(MT == 32 && M::NREG == 16 && WN % 2 == 0, "pairs of 32 x 32 accumulator tiles");
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int mode = p.act == VC_ACT_SIGMOID ? 3 : (p.act == VC_ACT_CLAMP01 ? 4 : 0);
    const int wpx = lane & 31, whalf = lane >> 5;          // accumulator layout: pixel, channel half-group
    const int rq = lane & 7, rpx = lane >> 3;              // read-back layout: channel octet, pixel within a group of 8
    static_for<0, WM>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int m = wm * WM + t;
        const int oy = oy0 + m / C::XT;
        static_for<0, WN / 2>([&](auto nc) {
            constexpr int n2 = decltype(nc)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[t][2 * n2 + h][4 * g], acc[t][2 * n2 + h][4 * g + 1], acc[t][2 * n2 + h][4 * g + 2],
                                     acc[t][2 * n2 + h][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(&scratch[wpx * VC_EPI2_ROWF + 32 * h + 8 * g + 4 * whalf]) = v;
                }
            const int co = nblk * C::BN + (wn * WN + 2 * n2) * NT + 8 * rq;   // first of this lane's 8 consecutive channels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = 8 * j + rpx;
                f32x4 v[2];
                v[0] = *reinterpret_cast<const f32x4 *>(&scratch[pix * VC_EPI2_ROWF + 8 * rq]);
                v[1] = *reinterpret_cast<const f32x4 *>(&scratch[pix * VC_EPI2_ROWF + 8 * rq + 4]);
                const int ox = ox0 + (m % C::XT) * MT + pix;
                if (oy < p.Ho && ox < p.Wo && co < p.Cout) {
                    const long long o_off = (long long)img * p.out_sn + (long long)oy * p.out_sh + (long long)ox * p.out_sw + co;
                    const long long r_off = (long long)img * p.res_sn + (long long)oy * p.res_sh + (long long)ox * p.res_sw + co;
                    f16x8 hv;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 w = v[q];
                        if (mode == 0 && p.res_first) w += *reinterpret_cast<const f32x4 *>(p.res + r_off + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (mode == 3) w[e] = 1.0f / (1.0f + expf(-w[e]));
                            else if (mode == 4) w[e] = fminf(fmaxf(w[e], 0.0f), 1.0f);
                            else w[e] = w[e] >= 0.0f ? w[e] : w[e] * neg;
                        }
                        if (p.chscale) w *= *reinterpret_cast<const f32x4 *>(p.chscale + co + 4 * q);
                        if (p.res && !(mode == 0 && p.res_first)) w += *reinterpret_cast<const f32x4 *>(p.res + r_off + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[4 * q + e] = (_Float16)w[e];
                    }
                    *reinterpret_cast<f16x8 *>(reinterpret_cast<_Float16 *>(p.out) + o_off) = hv;
                }
            }
        });
    });
}

// F16: the "fp16 MFMA conv path" (BASELINE.json configs[4]): activations are converted to half while being
// staged, weights are pre-packed as half fragments, v_mfma_f32_32x32x16_f16 accumulates in fp32.  A 16-byte LDS item /
// weight fragment then carries 8 channels instead of 4 and ONE MFMA consumes it (16x the fp32 matrix rate), so all
// addressing below is unchanged when expressed in 16-byte units: CK stays "LDS floats per pixel", the channel chunk
// doubles.  32-wide tiles only; everything else (tiny channel counts, GDN) stays on the exact fp32 instances.
// BLDS (4-channel configuration only): the weights of the whole layer live in LDS.  A 4x4x1 MFMA consumes a weight
// fragment every 8 cycles; fetched from global memory that is 1 KiB per 64 cycles and SIMD = the CU's whole L1 bandwidth
// (the matrix pipe sat at 45 %).  The fragment holds only 4 x 4 distinct floats (replicated over the 16 blocks), so the
// layer's unique weights are TAPS * cin_pad * 16 bytes; every lane reads its channel's 16 bytes with a broadcast
// ds_read_b128.  Same MFMA sequence, same operands: bit-identical to the global-memory variant.
template <int KH, int KW, int S, int CK, class C, bool F16 = false, bool BLDS = false>
__global__ void __launch_bounds__(256, C::MIN_WAVES) conv_mfma_kernel(const ConvArgs p)
{
    static_assert(!BLDS || (C::MT == 64 && !F16), "LDS-resident weights: 4x4x1 configuration, fp32");
    constexpr int FR = BLDS ? 16 : 256;     // floats between consecutive weight fragments
    typedef ConvGeom<KH, KW, S, CK, C> G;
    typedef Mfma<C::MT> M;
    static_assert(!F16 || C::MT == 32 || C::MT == 16, "the fp16 path uses the 32x32x16 / 16x16x32 MFMAs");
    constexpr int MT = C::MT, WM = C::WM, WN = C::WN, KS = G::KS, KSTEPS = G::KSTEPS;
    constexpr int CKC = F16 ? 2 * CK : CK;   // input channels per chunk
    constexpr int KSC = F16 ? 2 * KS : KS;   // input channels per k-step (per packed fragment)
    constexpr int TAPS = KH * KW;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // ---- block -> (n-block, tile x, tile y, image); consecutive ids share an XCD (and its L2) ----
    int bid = blockIdx.x;
    {
        const int nb = p.total_blocks, xcd = bid & 7, local = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int nblk = bid % p.nblks;
    const int t1 = bid / p.nblks;
    const int img = t1 / (p.tiles_x * p.tiles_y);
    int tx, ty;
    vc_tile_xy(t1 - img * (p.tiles_x * p.tiles_y), p.tiles_x, p.tiles_y, p.tile_band, tx, ty);

    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
    const int li = M::arow(lane), kk = M::akk(lane);
    constexpr int NT = C::NT;

    // ---- accumulators start at the bias of the lane's output channel ----
    typename M::acc_t acc[WM][WN];
    // (the weight fragment is the FIRST MFMA operand: accumulator register -> output channel, lane -> pixel, so a
    //  lane ends up with 4 consecutive channels of one pixel per register quad = one 16-byte store)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) {
            const float b = p.bias[nblk * C::BN + (wn * WN + n) * NT + M::crow(r, lane)];
#pragma unroll
            for (int t = 0; t < WM; ++t) acc[t][n][r] = b;
        }

    // ---- per-lane LDS read bases of the wave's M-tiles ----
    int abase[WM];
#pragma unroll
    for (int t = 0; t < WM; ++t) {
        const int m = wm * WM + t;
        const int row = m / C::XT, xt = m % C::XT;
        abase[t] = ((row * G::LS) * G::COLS_L + (xt * MT + li)) * G::CKP + 4 * kk;
    }

    const int ksteps_total = p.cin_pad / KSC;
    const long long ntile_stride = (long long)TAPS * ksteps_total * FR;
    const float *wlane;
    if constexpr (BLDS) {
        // lanes 0..3 of every packed fragment of this workgroup's N-tile -> [fragment][channel][4 k] behind the tile image
        float *wsm = lds + G::LDS_FLOATS + 4;
        const float *src = p.wpk + (long long)(nblk * (C::BN / NT) + wn * WN) * TAPS * ksteps_total * 256;
        for (int i = threadIdx.x; i < TAPS * ksteps_total * 4; i += 256)
            *reinterpret_cast<f32x4 *>(&wsm[4 * i]) = *reinterpret_cast<const f32x4 *>(src + (long long)(i >> 2) * 256 + (i & 3) * 4);
        wlane = wsm + (lane & 3) * 4;        // (visible after the barrier in front of the first staging)
    } else {
        wlane = p.wpk + (long long)(nblk * (C::BN / NT) + wn * WN) * ntile_stride + lane * 4;
    }

    const int pad_y = KH / 2, pad_x = KW / 2;
    const int iy0 = oy0 * S - pad_y, ix0 = ox0 * S - pad_x;
    // (a half-precision input is addressed in halves: the byte offset of image `img` is 2*img*in_sn)
    const float *in_img = (F16 && p.in_f16)
        ? reinterpret_cast<const float *>(reinterpret_cast<const _Float16 *>(p.in) + (long long)img * p.in_sn)
        : p.in + (long long)img * p.in_sn;

    // fp16 path, 64-channel chunks of a multi-tap kernel (3x3 stride 1): the chunk is contracted as two 32-channel halves,
    // each over all taps, so that every output sums its k-steps in the order (32-channel chunk, tap, k-step) -- the order
    // of the 32-channel-chunk instances (5x5, 7x7) and of the LDS-DMA kernel (conv_dma.h), which is therefore
    // bit-identical to this one.  (1x1: a single tap, the order is the same either way.)
    // fp32, 32-wide tiles, round 4: the same split (two 16-channel halves of a 32-channel chunk, each over all taps), so that the
    // exact fp32 LDS-DMA instances (DmaCfg::F32, 16-channel chunks) are bit-identical to these.
#ifdef VC_NO_F32_SUBS          // A/B builds only (tools/subs_ab.sh): the round-1..3 order of the fp32 3x3 loop
    constexpr int SUBS = (F16 && KSTEPS == 4 && KH * KW > 1) ? 2 : 1;
#else
    constexpr int SUBS = (KSTEPS == 4 && KH * KW > 1 && (F16 || C::MT == 32)) ? 2 : 1;
#endif
    constexpr int KSS = KSTEPS / SUBS;               // k-steps per tap and pass
    constexpr int RING = F16 ? vc_ring_depth(KW * KSS, vc_ring_regs_classic(WM, WN, M::NREG, C::MIN_WAVES), WN) : 1;
    f32x4 ring[RING][WN];
    if constexpr (F16) {          // steps 0 .. RING-1 of the first chunk's first kernel row
#pragma unroll
        for (int d = 0; d < RING; ++d)
#pragma unroll
            for (int n = 0; n < WN; ++n)
                ring[d][n] = *reinterpret_cast<const f32x4 *>(wlane + n * ntile_stride +
                                                              ((long long)(d / KSS) * ksteps_total + d % KSS) * FR);
    }

    VC_T(t_start);
    for (int c0 = 0; c0 < p.cin_pad; c0 += CKC) {
        VC_T(t_a);
        __syncthreads();
        VC_T(t_b);
        // ---- stage the input footprint of this channel chunk ----
        // Two phases per batch: issue all global loads of the batch (addresses clamped into the image so
        // no load sits under a branch), then select-zero / transform and write LDS.  This keeps BATCH
        // independent 16-byte loads in flight per lane instead of one load -> wait -> ds_write at a time.
        if (VC_SKIP(1) && c0 > 0) {
        } else if (F16 && p.in_f16)
            stage_chunk<KH, KW, S, CK, C, true, F16, F16>(p, lds, in_img, c0, oy0, ox0, iy0, ix0, threadIdx.x);
        else if (F16)
            stage_chunk<KH, KW, S, CK, C, true, true>(p, lds, in_img, c0, oy0, ox0, iy0, ix0, threadIdx.x);
        else if (p.vec4)
            stage_chunk<KH, KW, S, CK, C, true, false>(p, lds, in_img, c0, oy0, ox0, iy0, ix0, threadIdx.x);
        else
            stage_chunk<KH, KW, S, CK, C, false, false>(p, lds, in_img, c0, oy0, ox0, iy0, ix0, threadIdx.x);
        VC_T(t_c);
        __syncthreads();
        VC_T(t_d);

        // ---- contraction over taps x k-steps of this chunk, software-pipelined ----
        // The fragments of step s+1 (one 1 KiB global_load_dwordx4 per N-tile, one ds_read_b128 per
        // M-tile) are issued BEFORE the MFMAs of step s, so L2/LDS latency hides behind 8*WM*WN*... cycles
        // of matrix work instead of stalling every step (the compiler alone waits right after issuing).
        const float *wchunk = wlane + (long long)(c0 / KSC) * FR;
        constexpr int STEPS_X = KW * KSTEPS;  // steps per kernel row, fully unrolled
        f32x4 bc[WN], ac[WM], bn[WN], an[WM];
        auto load_b = [&](f32x4(&b)[WN], const float *wrow, int sx) {
            // sx = kx*KSTEPS + ks within the row; consecutive ks are 1 KiB apart, consecutive taps ksteps_total KiB
            const int kx = sx / KSTEPS, ks = sx % KSTEPS;
#pragma unroll
            for (int n = 0; n < WN; ++n)
                b[n] = *reinterpret_cast<const f32x4 *>(wrow + n * ntile_stride + ((long long)kx * ksteps_total + ks) * FR);
        };
        auto load_a = [&](f32x4(&a)[WM], int rowoff, int sx) {
            const int kx = sx / KSTEPS, ks = sx % KSTEPS;
            const int koff = (G::LS == 2) ? ((kx & 1) * G::HALF + (kx >> 1)) * G::CKP : kx * G::CKP;
#pragma unroll
            for (int t = 0; t < WM; ++t) a[t] = *reinterpret_cast<const f32x4 *>(&lds[abase[t] + rowoff + koff + ks * KS]);
        };
        if (VC_SKIP(2)) {
        } else if constexpr (F16) {
            // ---- fp16: weight fragments through a D-deep register ring (filled before the chunk loop and kept full
            // across rows, passes, chunks and the staging barriers), activations one step ahead from LDS ----
            constexpr int STEPS_R = KW * KSS;         // steps per kernel row and pass, fully unrolled
            auto load_bs = [&](f32x4(&b)[WN], const float *wrow, int sx) {
                const int kx = sx / KSS, ks = sx % KSS;
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    b[n] = *reinterpret_cast<const f32x4 *>(wrow + n * ntile_stride + ((long long)kx * ksteps_total + ks) * FR);
            };
            auto load_as = [&](f32x4(&a)[WM], int rowoff, int sx) {
                const int kx = sx / KSS, ks = sx % KSS;
                const int koff = (G::LS == 2) ? ((kx & 1) * G::HALF + (kx >> 1)) * G::CKP : kx * G::CKP;
#pragma unroll
                for (int t = 0; t < WM; ++t) a[t] = *reinterpret_cast<const f32x4 *>(&lds[abase[t] + rowoff + koff + ks * KS]);
            };
            const float *wchunk_n = wlane + (long long)((c0 + CKC < p.cin_pad ? c0 + CKC : c0) / KSC) * FR;
            load_as(ac, 0, 0);
#pragma unroll 1
            for (int r = 0; r < SUBS * KH; ++r) {     // (pass, kernel row); a pass starts KSS k-steps further into the chunk
                const int sub = r / KH, ky = r - sub * KH;
                const int rn = r + 1, subn = rn / KH, kyn = rn - subn * KH;
                const float *wrow = wchunk + ((long long)ky * KW * ksteps_total + sub * KSS) * FR;
                const float *wrow_n = rn < SUBS * KH ? wchunk + ((long long)kyn * KW * ksteps_total + subn * KSS) * FR : wchunk_n;
                const int rowoff = ky * G::COLS_L * G::CKP + sub * KSS * KS;
                const int rowoff_n = rn < SUBS * KH ? kyn * G::COLS_L * G::CKP + subn * KSS * KS : rowoff;
                static_for<0, STEPS_R>([&](auto sc) {
                    constexpr int sx = decltype(sc)::value;
                    constexpr int slot = sx % RING;
                    if constexpr (sx + 1 < STEPS_R) load_as(an, rowoff, sx + 1);
                    else load_as(an, rowoff_n, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < WM; ++t)
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            if constexpr (MT == 32)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ring[slot][n]),
                                                                                   __builtin_bit_cast(f16x8, ac[t]), acc[t][n], 0, 0, 0);
                            else
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[slot][n]),
                                                                                   __builtin_bit_cast(f16x8, ac[t]), acc[t][n], 0, 0, 0);
                        }
                    // refill the slot just consumed with the fragment RING steps ahead (same row, next row / pass or next chunk)
                    if constexpr (sx + RING < STEPS_R) load_bs(ring[slot], wrow, sx + RING);
                    else load_bs(ring[slot], wrow_n, sx + RING - STEPS_R);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < WM; ++t) ac[t] = an[t];
                });
            }
        } else {
        // (pass, kernel row): a pass covers KSS of the chunk's k-steps over all taps (SUBS = 1: the whole chunk)
        constexpr int STEPS_R = KW * KSS;
        auto load_b2 = [&](f32x4(&b)[WN], const float *wrow, int sx) {
            const int kx = sx / KSS, ks = sx % KSS;
#pragma unroll
            for (int n = 0; n < WN; ++n)
                b[n] = *reinterpret_cast<const f32x4 *>(wrow + n * ntile_stride + ((long long)kx * ksteps_total + ks) * FR);
        };
        auto load_a2 = [&](f32x4(&a)[WM], int rowoff, int sx) {
            const int kx = sx / KSS, ks = sx % KSS;
            const int koff = (G::LS == 2) ? ((kx & 1) * G::HALF + (kx >> 1)) * G::CKP : kx * G::CKP;
#pragma unroll
            for (int t = 0; t < WM; ++t) a[t] = *reinterpret_cast<const f32x4 *>(&lds[abase[t] + rowoff + koff + ks * KS]);
        };
        load_b2(bc, wchunk, 0);
        load_a2(ac, 0, 0);
        if constexpr (SUBS == 2) {
            // two passes, each over all taps: the taps of a pass fully unrolled (KH * STEPS_R steps), ONE rolled loop over the
            // passes -- a rolled loop over (pass, kernel row) drains the operand pipeline at every back-edge (6 per chunk
            // instead of 3: 4-9 % on the 3x3 layers, tools/subs_ab.sh)
#pragma unroll 1
            for (int sub = 0; sub < SUBS; ++sub) {
                const float *wpass = wchunk + (long long)sub * KSS * FR;
                const int aoff = sub * KSS * KS;
                const int subn = sub + 1 < SUBS ? sub + 1 : sub;          // (clamped: the last pass re-fetches its first step)
                const float *wpass_n = wchunk + (long long)subn * KSS * FR;
                const int aoff_n = subn * KSS * KS;
                static_for<0, KH * STEPS_R>([&](auto sc) {
                    constexpr int sxx = decltype(sc)::value;
                    constexpr int ky = sxx / STEPS_R, sx = sxx % STEPS_R;
                    constexpr int nxx = sxx + 1, kyn = nxx / STEPS_R, sxn = nxx % STEPS_R;
                    if constexpr (nxx < KH * STEPS_R) {
                        load_b2(bn, wpass + (long long)kyn * KW * ksteps_total * FR, sxn);
                        load_a2(an, kyn * G::COLS_L * G::CKP + aoff, sxn);
                    } else {
                        load_b2(bn, wpass_n, 0);
                        load_a2(an, aoff_n, 0);
                    }
                    (void)ky; (void)sx;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < WM; ++t)
#pragma unroll
                            for (int n = 0; n < WN; ++n) acc[t][n] = M::run(bc[n][e], ac[t][e], acc[t][n]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < WN; ++n) bc[n] = bn[n];
#pragma unroll
                    for (int t = 0; t < WM; ++t) ac[t] = an[t];
                });
            }
        } else {
#pragma unroll 1
        for (int ky = 0; ky < KH; ++ky) {
            const float *wrow = wchunk + (long long)ky * KW * ksteps_total * FR;
            const int rowoff = ky * G::COLS_L * G::CKP;
            // the row after this one (clamped: the last row re-fetches itself, a harmless extra load)
            const int kyn = ky + 1 < KH ? ky + 1 : ky;
            const float *wrow_n = wchunk + (long long)kyn * KW * ksteps_total * FR;
            const int rowoff_n = kyn * G::COLS_L * G::CKP;
            static_for<0, STEPS_R>([&](auto sc) {
                constexpr int sx = decltype(sc)::value;
                if constexpr (sx + 1 < STEPS_R) {
                    load_b2(bn, wrow, sx + 1);
                    load_a2(an, rowoff, sx + 1);
                } else {
                    load_b2(bn, wrow_n, 0);
                    load_a2(an, rowoff_n, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < WM; ++t)
#pragma unroll
                        for (int n = 0; n < WN; ++n) acc[t][n] = M::run(bc[n][e], ac[t][e], acc[t][n]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < WN; ++n) bc[n] = bn[n];
#pragma unroll
                for (int t = 0; t < WM; ++t) ac[t] = an[t];
            });
        }
        }
        }
        VC_T(t_e);
        VC_ACC(0, t_b, t_a);   // barrier before staging (waiting for the slowest wave of the previous chunk)
        VC_ACC(1, t_c, t_b);   // staging: loads + LDS writes
        VC_ACC(2, t_d, t_c);   // barrier after staging
        VC_ACC(3, t_e, t_d);   // contraction loop (MFMA + fragment traffic)
    }

    VC_T(t_loop_end);
    if (VC_SKIP(4)) {
    } else if constexpr (F16 && C::MT == 32 && G::LDS_FLOATS >= 4 * VC_EPI_SCRATCH_FLOATS) {
        if (p.vec_out && p.epi == VC_EPI_NONE) {
            __syncthreads();          // every wave has left the contraction loop: the tile image can serve as scratch
            if constexpr (C::WN % 2 == 0 && G::LDS_FLOATS >= 4 * VC_EPI2_SCRATCH_FLOATS) {
                // (8-channel pieces: Cout % 8 == 0 and 16-byte aligned half-precision rows)
                if (p.out_f16 && p.out_mode == VC_OUT_PLAIN && (p.Cout & 7) == 0 && ((p.out_sw | p.out_sh | p.out_sn) & 7) == 0)
                    conv_epilogue_coalesced_h2<C>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, lds + wave * VC_EPI2_SCRATCH_FLOATS);
                else
                    conv_epilogue_coalesced<C>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, lds + wave * VC_EPI_SCRATCH_FLOATS);
            } else {
                conv_epilogue_coalesced<C>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, lds + wave * VC_EPI_SCRATCH_FLOATS);
            }
        } else {
            conv_epilogue<C, F16>(p, acc, nblk, wm, wn, lane, oy0, ox0, img);
        }
    } else {
        conv_epilogue<C, F16>(p, acc, nblk, wm, wn, lane, oy0, ox0, img);
    }
    VC_T(t_end);
    VC_ACC(4, t_end, t_loop_end);   // epilogue
    VC_ACC(5, t_end, t_start);      // whole wave lifetime after the prologue
    VC_ACC(6, 1ull, 0ull);          // waves
}

template <int KH, int KW, int S, int CK, class C, bool F16 = false, bool BLDS = false> int launch_conv(hipStream_t st, const ConvArgs &a)
{
    typedef ConvGeom<KH, KW, S, CK, C> G;
    const size_t lds_bytes = (G::LDS_FLOATS + 4) * sizeof(float) +       // + the staging dump slot
                             (BLDS ? (size_t)KH * KW * (a.cin_pad / G::KS) * 64 : 0);
    auto kern = conv_mfma_kernel<KH, KW, S, CK, C, F16, BLDS>;
    static vc_lds_raised raised;          // largest size raised per device (common.h)
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.total_blocks), dim3(256), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}

// Every dispatcher source is compiled twice (parallel build): -DVC_TU_F16=0 gives the fp32 instances and the
// conv_dispatch_kN_f32 symbol, -DVC_TU_F16=1 the fp16 ones (conv_dispatch_kN_f16).  The N4 tile exists in fp32 only.
#ifdef VC_TU_F16
template <int KH, int KW, int S, int CK, class C> int launch_conv_p(hipStream_t st, const ConvArgs &a)
{
    return launch_conv<KH, KW, S, CK, C, VC_TU_F16 != 0>(st, a);
}
template <int KH, int KW, int S, int CK> int launch_conv_n4(hipStream_t st, const ConvArgs &a)
{
#if VC_TU_F16
    return VC_EINVAL;
#else
    // weights resident in LDS while they leave two workgroups per CU their tile images (<= 32 KiB: cin <= 40 at 7x7)
    // (VC_N4_GLOBAL_WEIGHTS=1 forces the global-memory variant: A/B runs and the bit-identity test)
    const char *force = getenv("VC_N4_GLOBAL_WEIGHTS");
    if ((size_t)KH * KW * (a.cin_pad / 4) * 64 <= 32 * 1024 && !(force && force[0] == '1'))
        return launch_conv<KH, KW, S, CK, CfgN4, false, true>(st, a);
    return launch_conv<KH, KW, S, CK, CfgN4, false>(st, a);
#endif
}
#if VC_TU_F16
#define VC_DISPATCH(k) conv_dispatch_##k##_f16
#else
#define VC_DISPATCH(k) conv_dispatch_##k##_f32
#endif
#endif

// one dispatcher over the tile configs per kernel size and precision (`ck` = LDS floats per pixel of the instance)
#define VC_DECLARE_DISPATCH(k)                                                             \
    int conv_dispatch_##k##_f32(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck); \
    int conv_dispatch_##k##_f16(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck);
// streaming 1x1 kernel (conv_pw.hip): same packed weights as the 32-wide configurations
bool conv_pw_eligible(const ConvArgs &a, int k, int stride, bool f16);
bool conv_pws_eligible(const ConvArgs &a, int k, int stride, bool f16);
int conv_dispatch_pws(hipStream_t st, const ConvArgs &a, bool f16);
int conv_dispatch_pw(hipStream_t st, const ConvArgs &a, bool f16);
// fp16-path LDS-DMA pipeline (conv_dma.hip): same packed weights again; sets the tile geometry of `a` itself
int conv_dispatch_dma(hipStream_t st, ConvArgs a, int k, int stride, bool f16);
// split-operand fp32 pipeline (conv_split.hip): its own packed weights (vc_conv_pack_weights_split), split input tensor
int conv_dispatch_split(hipStream_t st, ConvArgs a, int k, int stride);
VC_DECLARE_DISPATCH(k1)
VC_DECLARE_DISPATCH(k3)
VC_DECLARE_DISPATCH(k5)
VC_DECLARE_DISPATCH(k7)
