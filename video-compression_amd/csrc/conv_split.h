// fp32 convolution on the bf16 matrix pipe with SPLIT operands (VC_CFG_SPLIT; stride 1, K x K with K = 5 or 7).
//
// Why: the native fp32 MFMA runs at 1/16 of the bf16 rate and the headline path sits at 0.90 of it.  An fp32 value is the
// exact sum of three bf16 pieces (hi + mid + lo, 8 + 8 + 8 significand bits, by truncation: every remainder is exact), the
// product of two pieces is exact in fp32 (16 significand bits), so
//     a * b = sum over the nine piece pairs (exact),  accumulated by the MFMA's own fp32 accumulate
// -- nine v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) instead of sixteen v_mfma_f32_16x16x4_f32-equivalents: 9/16 of the
// matrix time before the clock (tools/micro/mfma_split.hip: 1.39x the native k-step with its LDS operand traffic, error against
// fp64 no worse than the native chain's; the 16 x 16 shape holds a ~15 % higher clock than 32 x 32 x 16 under this load).
//
// Data: the INPUT is a "split" tensor (vc_split3, or the split epilogue of the producing layer): [n][C/8][H][W][3 pieces][8
// channels] bf16 -- per pixel and group of 8 channels ONE 48-byte record hi | mid | lo.  A channel chunk (8 channels) of a tile's
// footprint is then a set of contiguous rows; it reaches LDS by global_load_lds_dwordx4 exactly as it lies (one 16-byte slot
// per (pixel, piece)), borders from a zero page.  The WEIGHTS are split at pack time (vc_conv_pack_weights_split).
//
// Contraction: K of an MFMA = 32 = 4 taps x 8 channels.  The K x K taps of a chunk are walked as UNITS of four taps: 2 x 2
// blocks over the taps with ky, kx < K - 1, vertical runs down the last column, horizontal runs along the last row (7 x 7: 9 +
// 2 + 2 = 13 units for 49 taps, 6 % padding; padded k-groups carry zero weights and re-read the unit's last valid tap, never
// stale LDS).  Lane (pixel l & 15, k-group l >> 4) reads its 16 bytes at (pixel + tap offset of its k-group, piece): the tap
// offset is a per-lane constant per unit class, everything else an immediate.
//
// Pipeline: the LDS-DMA pipeline of conv_dma.h -- one persistent 512-thread workgroup per CU over a tile list in the banded
// order, a phase = one unit {fragment reads, DMA issue, counted vmcnt wait, barrier, 9 x WM x WN MFMAs, barrier}, the two waves
// of a SIMD half a phase apart, chunk images double-buffered, weights through a ring of RING slots fetched RING - 2 units
// ahead -- with the CHUNK loop rolled (13 phases of code; the channel count is a launch parameter).
// Accumulation order per output: (chunk, unit, product pair from lo x lo up to hi x hi); NOT the order of the native instances:
// results differ from them by fp32 summation noise (like the native instances differ from the CPU reference), tested against
// an fp64 reference beside them.
#pragma once
#include "conv_dma.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ---- the tap units of a K x K kernel (shared by the host packer and the kernel) ----
template <int K> struct SplitUnits {
    static constexpr int TPU = 4;                        // taps per unit (one 8-channel plane per k-group)
    static constexpr int Q = (K / 2) * (K / 2);          // 2 x 2 blocks over ky, kx < K - 1
    static constexpr int V = (K + 3) / 4;                // runs of <= 4 down the column kx = K - 1 (K taps)
    static constexpr int H = (K - 1 + 3) / 4;            // runs of <= 4 along the row ky = K - 1, kx < K - 1 (K - 1 taps)
    static constexpr int U = Q + V + H;
    // type: 0 block, 1 vertical run, 2 horizontal run
    static constexpr int type(int u) { return u < Q ? 0 : (u < Q + V ? 1 : 2); }
    static constexpr int ky0(int u) { return u < Q ? 2 * (u / (K / 2)) : (u < Q + V ? 4 * (u - Q) : K - 1); }
    static constexpr int kx0(int u) { return u < Q ? 2 * (u % (K / 2)) : (u < Q + V ? K - 1 : 4 * (u - Q - V)); }
    static constexpr int nvalid(int u)
    {
        return u < Q ? 4 : (u < Q + V ? (K - 4 * (u - Q) < 4 ? K - 4 * (u - Q) : 4) : (K - 1 - 4 * (u - Q - V) < 4 ? K - 1 - 4 * (u - Q - V) : 4));
    }
    // tap of k-group q of unit u (q >= nvalid: padding, zero weight; the ADDRESS of a padded group is its unit's last valid tap)
    static constexpr int ky(int u, int q)
    {
        const int qq = q < nvalid(u) ? q : nvalid(u) - 1;
        return type(u) == 0 ? ky0(u) + (qq >> 1) : (type(u) == 1 ? ky0(u) + qq : ky0(u));
    }
    static constexpr int kx(int u, int q)
    {
        const int qq = q < nvalid(u) ? q : nvalid(u) - 1;
        return type(u) == 0 ? kx0(u) + (qq & 1) : (type(u) == 1 ? kx0(u) : kx0(u) + qq);
    }
    static constexpr bool valid(int u, int q) { return q < nvalid(u); }
    // per-lane offset class of a unit: lanes of k-group q add (dy(cls, q) * COLS + dx(cls, q)) pixels
    static constexpr int cls(int u) { return type(u) * 4 + nvalid(u) - 1; }
    static constexpr int NCLS = 12;
    static constexpr int dy(int c, int q)
    {
        const int t = c / 4, nv = c % 4 + 1, qq = q < nv ? q : nv - 1;
        return t == 0 ? (qq >> 1) : (t == 1 ? qq : 0);
    }
    static constexpr int dx(int c, int q)
    {
        const int t = c / 4, nv = c % 4 + 1, qq = q < nv ? q : nv - 1;
        return t == 0 ? (qq & 1) : (t == 1 ? 0 : qq);
    }
    static constexpr bool covers_all()
    {
        int seen[K * K] = {};
        for (int u = 0; u < U; ++u)
            for (int q = 0; q < 4; ++q)
                if (valid(u, q)) ++seen[ky(u, q) * K + kx(u, q)];
        for (int i = 0; i < K * K; ++i)
            if (seen[i] != 1) return false;
        return true;
    }
    static_assert(covers_all(), "every tap in exactly one unit");
};

// The same for chunks of TWO 8-channel planes (3 x 3 layers): K of an MFMA = 2 taps x 16 channels, k-group q = (tap q >> 1, plane
// q & 1).  Units = horizontal pairs over kx < K - 1 row by row, then vertical pairs down the column kx = K - 1 (3 x 3: 3 + 2 = 5
// units for 9 taps, 10 % padding).  Interface as SplitUnits with q = the TAP index of the k-group.
template <int K> struct SplitPairs {
    static constexpr int TPU = 2;
    static constexpr int HP = (K - 1) / 2;               // horizontal pairs per kernel row
    static constexpr int Q = HP * K;
    static constexpr int V = (K + 1) / 2;                // vertical pairs (the last one may be a single tap)
    static constexpr int U = Q + V;
    static constexpr int type(int u) { return u < Q ? 2 : 1; }
    static constexpr int ky0(int u) { return u < Q ? u / HP : 2 * (u - Q); }
    static constexpr int kx0(int u) { return u < Q ? 2 * (u % HP) : K - 1; }
    static constexpr int nvalid(int u) { return u < Q ? 2 : (K - 2 * (u - Q) < 2 ? 1 : 2); }
    static constexpr int ky(int u, int q) { return type(u) == 1 ? ky0(u) + (q < nvalid(u) ? q : nvalid(u) - 1) : ky0(u); }
    static constexpr int kx(int u, int q) { return type(u) == 2 ? kx0(u) + (q < nvalid(u) ? q : nvalid(u) - 1) : kx0(u); }
    static constexpr bool valid(int u, int q) { return q < nvalid(u); }
    static constexpr int cls(int u) { return type(u) * 4 + nvalid(u) - 1; }
    static constexpr int NCLS = 12;
    static constexpr int dy(int c, int q)
    {
        const int t = c / 4, nv = c % 4 + 1, qq = q < nv ? q : nv - 1;
        return t == 1 ? qq : 0;
    }
    static constexpr int dx(int c, int q)
    {
        const int t = c / 4, nv = c % 4 + 1, qq = q < nv ? q : nv - 1;
        return t == 2 ? qq : 0;
    }
    static constexpr bool covers_all()
    {
        int seen[K * K] = {};
        for (int u = 0; u < U; ++u)
            for (int q = 0; q < 2; ++q)
                if (valid(u, q)) ++seen[ky(u, q) * K + kx(u, q)];
        for (int i = 0; i < K * K; ++i)
            if (seen[i] != 1) return false;
        return true;
    }
    static_assert(covers_all(), "every tap in exactly one unit");
};

template <int K, int CPL> struct SplitUnitsOf { typedef SplitUnits<K> type; };
template <int K> struct SplitUnitsOf<K, 2> { typedef SplitPairs<K> type; };

// NTW: N-tiles of 16 output channels per workgroup (2 or 4: blocks of 32 / 64 channels); CPL: 8-channel planes per chunk (1: K of an
// MFMA = 4 taps x 8 channels, 5 x 5 / 7 x 7; 2: 2 taps x 16 channels, 3 x 3); TH: rows of the 32-pixel-wide output tile (16, or 12
// where two chunk images of 16 rows do not fit beside the weight ring)
template <int K_, int NTW_, int CPL_ = 1, int TH_ = 16, int RING_ = 4, int KO_ = 0> struct SplitCfg {
    typedef typename SplitUnitsOf<K_, CPL_>::type UN;
    static_assert(UN::TPU * CPL_ == 4, "four k-groups of 8 channels per MFMA");
    static constexpr int K = K_, NTW = NTW_, CPL = CPL_, RING = RING_, KO = KO_, U = UN::U, PAD = K_ / 2;
    static constexpr int TH = TH_, TW = 32, BN = 16 * NTW_;
    // wave w owns the M-tiles (16 pixels of a row) m = WM w .. WM w + WM - 1: row m >> 1, columns 16 (m & 1) ..
    static constexpr int WAVES = 8, WM = TH_ * 2 / 8, WN = NTW_;
    static_assert(WM * 8 == TH_ * 2, "the tile's 2 TH M-tiles split evenly over 8 waves");
    static constexpr int ROWS_IN = TH + K_ - 1, COLS = TW + K_ - 1, PIX = ROWS_IN * COLS;
    static constexpr int PXB = 48;                               // bytes per pixel of a plane image: 3 pieces x 8 channels x 2
    // the chunk image: [plane][pixel][piece] in 16-byte slots.  With two planes the second one starts on a 256-byte bank row: the two
    // k-groups that share a ds_read_b128 lane group (same tap, planes 0 / 1) then read 16 disjoint bank quads (pixel stride 48 bytes =
    // 12 banks; a plane offset of 16 banks mod 64, as the unpadded 14 x 34 image has, makes 4 of the 8 pairs collide)
    static constexpr int PLANE_SLOTS = CPL_ == 1 ? 3 * PIX : (3 * PIX + 15) / 16 * 16;
    static constexpr int PLANE_B = PLANE_SLOTS * 16;
    static constexpr int NA = (CPL_ * PLANE_SLOTS + 511) / 512;  // DMA instructions per wave and chunk image
    static constexpr int A_BYTES = NA * 8192;
    static constexpr int FRAGS = 3 * NTW_;                       // weight fragments (1 KiB) per unit: [piece][n-tile]
    // DMA rounds per unit: every wave fetches one KiB per round.  A last round of exactly 4 fragments is fetched by waves 0-3 (= wave
    // group 0) alone -- the counted waits then differ per group --; any other remainder is fetched by all waves (the tail over-reads
    // into the next unit's fragments and lands in slot padding)
    static constexpr int ROUNDS = (FRAGS + 7) / 8, LASTW = FRAGS - 8 * (ROUNDS - 1);
    static constexpr bool EXACT = LASTW == 4;
    static constexpr int SLOTB = EXACT ? FRAGS * 1024 : ROUNDS * 8192;
    static constexpr int nb(int grp) { return ROUNDS - ((EXACT && grp == 1) ? 1 : 0); }
    static constexpr int B_OFF = 2 * A_BYTES, BIAS_OFF = B_OFF + RING_ * SLOTB;
    static constexpr int LDS_FIXED = BIAS_OFF;                   // + 4 * Cout at launch
    static constexpr int D = RING_ - 2;                          // weights are requested D units ahead
    // pieces of the NEXT chunk image: PPP per phase from phase 1 on; they must all be older than the weight request of phase U - D
    static constexpr int WIN = U - D - 1, PPP = (NA + WIN - 1) / WIN;
    static_assert(WIN >= 1, "no issue window for the chunk image");
    static constexpr int piece_phase(int k) { return 1 + k / PPP; }
    static constexpr int nA(int u)
    {
        int n = 0;
        for (int k = 0; k < NA; ++k) n += piece_phase(k) == u;
        return n;
    }
    // vmcnt of phase u: the weights of unit u + 1 were the LAST requests of phase u + 1 - D; younger: everything of the phases after it
    static constexpr int nwait(int u, int grp)
    {
        int n = 0;
        for (int j = u + 2 - D; j <= u; ++j) n += nA(((j % U) + U) % U) + nb(grp);
        return n;
    }
    static_assert(D >= 1 && nwait(U - 1, 0) <= 63, "vmcnt is a 6-bit counter");
};

// Epilogue of the split kernels: lane = pixel (px of M-tile m = WM wave + t: row m >> 1, columns 16 (m & 1) ..), k-group q = channel quad:
// 4 consecutive channels per accumulator -> one 16-byte access (or the 3 x 8 bytes of a split record) per (M-tile, N-tile).  Stores of
// pixels / channels outside the output are simply masked: the counted waits of these kernels count DMA only -- an uncounted younger
// store makes a wait longer, never shorter.
// Round 6: a SPLIT output leaves through LDS as whole records.  The direct form below stores 3 x 8 bytes per lane and accumulator tile
// at a 48-byte stride -- 24 partial 64-byte lines per wave-instruction, 36 instructions per 12-tile wave: the epilogue was 10.6 % of a
// 3x3 wave's lifetime (tools/r06.sh stamps) and, both wave groups passing through it one after the other, about twice that of the
// matrix pipe's idle time.  Here the lanes of a wave stage the pieces of NG N-tiles of one M-tile in a private LDS slice
// ([plane][pixel][48 bytes]: exactly the record image of 2 NG planes x 16 pixels) and write it back out 16 bytes per lane, contiguous:
// every global store instruction covers 1 KiB of whole records (the 768 contiguous bytes of a plane's 16 pixels, then the next plane's).
// `sc`: this wave's slice of a chunk-image buffer nobody reads or fills during the epilogue (the period kernels' Y buffer), >= NG * 1536 bytes.
// (What is left of the epilogue -- 7 % of a 3x3 wave's lifetime, 4 % of it the CU's vector-memory path taking the tile's 147 KB -- could
//  only hide behind the NEXT tile's phases; that needs a second accumulator set, and two waves per SIMD leave 256 registers per wave: the
//  64-channel instances spill 100-190 of them to scratch with it.  Tried and removed)
template <class C, int NG, int T0 = 0, int T1 = C::WM>
__device__ __forceinline__ void split_epilogue_records(const ConvArgs &p, f32x4 (&acc)[C::WM][C::WN], const DmaTile &cur, int wave, int px, int q,
                                                       unsigned char *sc)
{
    constexpr int WM = C::WM, WN = C::WN, UNITS = 96 * NG, NJ = (UNITS + 63) / 64, NGRP = (T1 - T0) * (WN / NG);
    static_assert(WN % NG == 0, "N-tiles are staged in whole groups");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
    const int lane = px + 16 * q;
    int u_plane[NJ], u_w16[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int u = 64 * j + lane;
        u_plane[j] = u / 48;
        u_w16[j] = u - 48 * u_plane[j];
    }
    unsigned char *const outb = reinterpret_cast<unsigned char *>(p.out) + (long long)cur.img * p.out_sn;
    const unsigned char *const resb = reinterpret_cast<const unsigned char *>(p.res) + (long long)cur.img * p.res_sn;
    // group g = (M-tile T0 + g / (WN / NG), N-tiles (g % (WN / NG)) * NG ..): its 2 NG planes x 16 pixels of records, unit u of lane
    auto unit_ok = [&](int g, int j, int &oy, int &oxb, int &plane0) {
        const int t = T0 + g / (WN / NG), n0 = (g % (WN / NG)) * NG, m = WM * wave + t;
        oy = cur.oy0 + (m >> 1);
        oxb = cur.ox0 + 16 * (m & 1);
        plane0 = (cur.nblk * C::BN + 16 * n0) >> 3;
        return (64 * j + lane < UNITS) && oy < p.Ho && (oxb + u_w16[j] / 3 < p.Wo);
    };
    // A SPLIT residual (the identity of a residual block inside a split chain) is read the same way: whole records, 16 bytes per lane,
    // one group ahead of its use (the direct form: 3 x 8 bytes per lane and accumulator tile, as scattered as the stores were -- with it
    // the epilogue was 16 % of a 3x3 wave's lifetime, 7 % without a residual)
    vc_u32x4 rnext[NJ];
    auto fetch_res = [&](int g) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int oy, oxb, plane0;
            const bool ok = unit_ok(g, j, oy, oxb, plane0);
            rnext[j] = vc_u32x4{0u, 0u, 0u, 0u};
            if (ok) rnext[j] = *reinterpret_cast<const vc_u32x4 *>(resb + ((((long long)(plane0 + u_plane[j])) * p.Ho + oy) * p.Wo + oxb) * 48 + u_w16[j] * 16);
        }
    };
    if (p.res_sp3) fetch_res(0);
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        const int t = T0 + g / (WN / NG), n0 = (g % (WN / NG)) * NG;
        const int m = WM * wave + t;
        const int oy = cur.oy0 + (m >> 1), oxb = cur.ox0 + 16 * (m & 1), ox = oxb + px;
        const bool row_ok = oy < p.Ho, pix_ok = row_ok && ox < p.Wo;
        if (p.res_sp3) {
            // this group's residual records into the staging area, the next group's on their way
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (64 * j + lane < UNITS) *reinterpret_cast<vc_u32x4 *>(sc + u_plane[j] * 768 + u_w16[j] * 16) = rnext[j];
            if (g + 1 < NGRP) fetch_res(g + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int nn = 0; nn < NG; ++nn) {
            const int n = n0 + nn;
            const int co = cur.nblk * C::BN + 16 * n + 4 * q;
            unsigned char *rec = sc + ((2 * nn + (q >> 1)) * 16 + px) * 48 + 8 * (q & 1);
            f32x4 v = acc[t][n];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (p.res_sp3) {
                r = vc_load_split4(rec - 8 * (q & 1), q & 1);         // (pixels outside the output: zero records were staged)
            } else if (p.res) {
                if (pix_ok) r = *reinterpret_cast<const f32x4 *>(p.res + (long long)cur.img * p.res_sn + (long long)oy * p.res_sh + (long long)ox * p.res_sw + co);
            }
            if (p.res_first) v += r;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
            if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + co);
            if (p.res && !p.res_first) v += r;
            unsigned h[4], mm[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vc_split3(v[e], h[e], mm[e], l[e]);
            const u32x2 ph = {(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
            const u32x2 pm = {(mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3]};
            const u32x2 pl = {(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
            if constexpr (C::KO & 256) {          // (diagnostic: arithmetic only)
                VC_DMA_KEEP(ph); VC_DMA_KEEP(pm); VC_DMA_KEEP(pl);
            } else {
                // (each lane overwrites exactly the 3 x 8 bytes it has just read its residual from)
                *reinterpret_cast<u32x2 *>(rec) = ph;
                *reinterpret_cast<u32x2 *>(rec + 16) = pm;
                *reinterpret_cast<u32x2 *>(rec + 32) = pl;
            }
        }
        if constexpr (C::KO & 256) continue;
        // (LDS operations of one wave execute in order: the reads below see every lane's writes above; the compiler must keep the order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int plane0 = (cur.nblk * C::BN + 16 * n0) >> 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool ok = (64 * j + lane < UNITS) && row_ok && (oxb + u_w16[j] / 3 < p.Wo);
            if (ok) {
                const vc_u32x4 val = *reinterpret_cast<const vc_u32x4 *>(sc + u_plane[j] * 768 + u_w16[j] * 16);
                if constexpr (C::KO & 128) VC_DMA_KEEP(val);      // (diagnostic: no global stores)
                else *reinterpret_cast<vc_u32x4 *>(outb + ((((long long)(plane0 + u_plane[j])) * p.Ho + oy) * p.Wo + oxb) * 48 + u_w16[j] * 16) = val;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the staging area is re-used by the next group)
    }
}

// M-tiles T0 .. T1 - 1 of the wave.  SC_NG: N-tiles the record staging area `sc` holds (0: sized from the chunk-image buffer the full
// epilogue borrows; 1: the small per-wave area of the deferred epilogue)
template <class C, int T0 = 0, int T1 = C::WM, int SC_NG = 0>
__device__ __forceinline__ void split_epilogue(const ConvArgs &p, f32x4 (&acc)[C::WM][C::WN], const DmaTile &cur, int wave, int px, int q,
                                               unsigned char *sc = nullptr)
{
    constexpr int WM = C::WM, WN = C::WN;
    if constexpr (SC_NG > 0 || (C::A_BYTES / 8 >= 3072 && WN % 2 == 0)) {
        // whole-record stores of a split output (plain layout; the pixel-shuffle store keeps the direct form)
        if (sc && p.out_sp3 && p.out_mode == VC_OUT_PLAIN) {
            constexpr int NG = SC_NG > 0 ? SC_NG : ((C::A_BYTES / 8 >= WN * 1536) ? WN : 2);
            split_epilogue_records<C, NG, T0, T1>(p, acc, cur, wave, px, q, sc);
            return;
        }
    }
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        const int m = WM * wave + t;
        const int oy = cur.oy0 + (m >> 1), ox = cur.ox0 + 16 * (m & 1) + px;
        const bool pix_ok = oy < p.Ho && ox < p.Wo;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = cur.nblk * C::BN + 16 * n + 4 * q;
            const bool ok = pix_ok && co < p.Cout;
            f32x4 v = acc[t][n];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            // nn.PixelShuffle(2) fused into the store (the packed output channels are permuted so that 4 consecutive ones share a position)
            const bool ps = p.out_mode != VC_OUT_PLAIN;
            const int cps = p.Cout >> 2;
            const int pos = ps ? co / cps : 0;
            const int cch = ps ? co - pos * cps : co;
            const int yy = ps ? 2 * oy + (pos >> 1) : oy, xx = ps ? 2 * ox + (pos & 1) : ox;
            const int oh = ps ? 2 * p.Ho : p.Ho, ow = ps ? 2 * p.Wo : p.Wo;
            if (p.res_sp3) {                 // the identity kept as a split tensor: its three pieces sum to the exact fp32 value
                if (ok) r = vc_load_split4(reinterpret_cast<const unsigned char *>(p.res) + (long long)cur.img * p.res_sn +
                                               ((((long long)(cch >> 3)) * oh + yy) * ow + xx) * 48, (cch >> 2) & 1);
            } else if (p.res) {
                const long long r_off = (long long)cur.img * p.res_sn + (long long)yy * p.res_sh + (long long)xx * p.res_sw + cch;
                if (ok) r = *reinterpret_cast<const f32x4 *>(p.res + r_off);
            }
            if (p.res_first) v += r;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
            if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + min(co, p.Cout - 4));
            if (p.res && !p.res_first) v += r;
            if (p.out_sp3) {
                // split output for a VC_CFG_SPLIT consumer: 3 x 8 bytes (4 channels of one piece) into the pixel's 48-byte record
                unsigned char *o8 = reinterpret_cast<unsigned char *>(p.out) + (long long)cur.img * p.out_sn + ((((long long)(cch >> 3)) * oh + yy) * ow + xx) * 48;
                if (ok) vc_store_split4(o8, (cch >> 2) & 1, v);
            } else {
                const long long o_off = (long long)cur.img * p.out_sn + (long long)yy * p.out_sh + (long long)xx * p.out_sw + cch;
                if (ok) *reinterpret_cast<f32x4 *>(p.out + o_off) = v;
            }
        }
    }
}

template <class C> __global__ void __launch_bounds__(512, 2) conv_split_kernel(const ConvArgs p)
{
    typedef typename C::UN UN;
    constexpr int U = C::U, WM = C::WM, WN = C::WN, NA = C::NA, RING = C::RING, D = C::D, CPL = C::CPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];   // the kernel's only LDS object: starts at LDS address 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int px = lane & 15, q = lane >> 4;

    // ---- persistent tile list (as conv_dma_kernel): the 32 workgroups of an XCD walk one contiguous range of the banded order ----
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

    const int nchunk = p.Cin / (8 * CPL);                         // chunks of CPL 8-channel planes (runtime: the chunk loop is rolled)
    const int upt = nchunk * U;                                   // units per tile
    const unsigned char *const in_b = reinterpret_cast<const unsigned char *>(p.in);
    const unsigned char *const wpk = reinterpret_cast<const unsigned char *>(p.wpk);
    const long long plane = (long long)p.H * p.W * C::PXB;        // bytes of one 8-channel plane of one image
    const unsigned char *const zero_lane = g_vc_dma_zero + 16 * lane;

    // ---- A pieces: lane (k, tid) fills slot s = 512 k + tid = (plane, pixel, piece) of the chunk image, from the same position of that plane ----
    int a_off[NA], a_rc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int s = k * 512 + tid;
        const int pl = s / C::PLANE_SLOTS, s2 = s - pl * C::PLANE_SLOTS;
        const int pix = s2 / 3, piece = s2 - 3 * pix;
        const int row = pix / C::COLS, col = pix - row * C::COLS;
        a_off[k] = pl * (int)plane + (row * p.W + col) * C::PXB + piece * 16;
        a_rc[k] = (pl < CPL && s2 < 3 * C::PIX) ? (row | (col << 8)) : 0x7f7f7f;
    }
    auto tile_base = [&](const DmaTile &t) {       // chunk 0's first plane at the footprint's first pixel (may lie outside the tensor)
        return in_b + (long long)t.img * p.in_sn + ((long long)(t.oy0 - C::PAD) * p.W + (t.ox0 - C::PAD)) * C::PXB;
    };
    auto issue_a = [&](const DmaTile &t, const unsigned char *tbase, int c, int k, int buf) {
        int rc = a_rc[k], off = a_off[k];
        asm volatile("" : "+v"(rc), "+v"(off));
        const unsigned iy = (unsigned)(t.oy0 - C::PAD + (rc & 0xff)), ix = (unsigned)(t.ox0 - C::PAD + (rc >> 8));
        const bool ok = t.valid && iy < (unsigned)p.H && ix < (unsigned)p.W;
        const unsigned char *sp = ok ? tbase + ((long long)c * CPL * plane + off) : zero_lane;
        if constexpr (!(C::KO & 16)) vc_glds16<true>(sp, (unsigned)(buf * C::A_BYTES + k * 8192 + wave * 1024));
    };
    // weights of tile-unit g (>= upt: of the next tile): one KiB per wave and round, [nblk][chunk][unit][piece][n-tile] order;
    // ring slot = (ring position of the tile's first unit + g) mod RING
    const unsigned lane16 = 16 * lane;
    unsigned gbase = 0;
    auto issue_b = [&](int g, int nblk_cur, int nblk_next) {
        const int gg = g >= upt ? g - upt : g;
        const int nblk = g >= upt ? nblk_next : nblk_cur;
        const unsigned char *sbase = wpk + ((long long)nblk * upt + gg) * (C::FRAGS * 1024) + wave * 1024;
        const unsigned dst = C::B_OFF + ((gbase + (unsigned)g) % RING) * C::SLOTB + wave * 1024;
#pragma unroll
        for (int r = 0; r < C::ROUNDS; ++r)
            if constexpr (!(C::KO & 16)) {
                if (r + 1 < C::ROUNDS || !C::EXACT || grp == 0) vc_glds16_sbase(sbase + r * 8192, lane16, dst + r * 8192);
            }
    };

    // ---- prologue: bias, first chunk image, weights of the first D units ----
    float *const ldsf = reinterpret_cast<float *>(lds8);
    for (int i = tid; i < p.nblks * C::BN; i += 512) ldsf[C::BIAS_OFF / 4 + i] = p.bias[i];
    DmaTile nxt = tile_at(1);
    const unsigned char *cur_base = tile_base(cur), *nxt_base = tile_base(nxt);
#pragma unroll
    for (int k = 0; k < NA; ++k) issue_a(cur, cur_base, 0, k, 0);
#pragma unroll
    for (int g = 0; g < D; ++g) issue_b(g, cur.nblk, cur.nblk);
    vc_wait_vmcnt<0>();
    __syncthreads();

    // ---- per-lane LDS read offsets: the M-tile's pixel + the tap offset (and plane) of the lane's k-group, per unit class.  With an
    // even WM the M-tiles of a wave sit at compile-time offsets from its first one (immediates); with WM = 3 each has its own base ----
    constexpr bool TIMM = (WM % 2 == 0);
    const int qtap = q / CPL, qpl = q % CPL;
    int a_lane[UN::NCLS][TIMM ? 1 : WM];
#pragma unroll
    for (int c = 0; c < UN::NCLS; ++c) {
        int dy = 0, dx = 0;
#pragma unroll
        for (int qq = 0; qq < UN::TPU; ++qq)
            if (qtap == qq) {
                dy = UN::dy(c, qq);
                dx = UN::dx(c, qq);
            }
#pragma unroll
        for (int t = 0; t < (TIMM ? 1 : WM); ++t) {
            const int m = WM * wave + t;
            a_lane[c][t] = (((m >> 1) + dy) * C::COLS + 16 * (m & 1) + px + dx) * C::PXB + qpl * C::PLANE_B;
        }
    }
    const int b_lane = C::B_OFF + lane * 16;

    f32x4 acc[WM][WN];
    f32x4 af[3][WM], bf[3][WN];
    int gchunk = 0;                                  // chunks contracted so far: parity = buffer of the current chunk
    for (int it = 0;; ++it) {
        // ---- accumulators start at the bias: lane (pixel, channel quad cq = q) owns channels 16 n + 4 q .. + 3 ----
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(&ldsf[C::BIAS_OFF / 4 + cur.nblk * C::BN + 16 * n + 4 * q]);
#pragma unroll
            for (int t = 0; t < WM; ++t) acc[t][n] = b;
        }
        if (grp == 1) VC_DMA_BARRIER();              // waves 4-7 run half a phase behind waves 0-3
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const int abuf = ((gchunk + c) & 1) * C::A_BYTES, nbuf = ((gchunk + c + 1) & 1);
            const bool last_chunk = c + 1 == nchunk;
            const int g0 = c * U;                    // tile-unit index of this chunk's first unit
            static_for<0, U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int cl = UN::cls(u);
                // -- R: this unit's fragments, LDS -> registers --
                const int bslot = b_lane + (int)((gbase + (unsigned)(g0 + u)) % RING) * C::SLOTB;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                    for (int n = 0; n < WN; ++n) bf[pc][n] = *reinterpret_cast<const f32x4 *>(lds8 + bslot + (pc * WN + n) * 1024);
#pragma unroll
                    for (int t = 0; t < WM; ++t) {
                        const int ab = a_lane[cl][TIMM ? 0 : t] + abuf;
                        constexpr int unit_imm = (UN::ky0(u) * C::COLS + UN::kx0(u)) * C::PXB;
                        const int t_imm = TIMM ? ((t >> 1) * C::COLS + 16 * (t & 1)) * C::PXB : 0;
                        af[pc][t] = *reinterpret_cast<const f32x4 *>(lds8 + ab + unit_imm + t_imm + 16 * pc);
                    }
                }
                // -- DMA of later phases: a piece of the next chunk image, then the weights of unit g + D --
                static_for<0, NA>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (C::piece_phase(k) == u) {
                        if (!last_chunk) issue_a(cur, cur_base, c + 1, k, nbuf);
                        else issue_a(nxt, nxt_base, 0, k, nbuf);
                    }
                });
                issue_b(g0 + u + D, cur.nblk, nxt.nblk);
                // -- this wave's part of the next unit's weights (and everything older) has landed --
                // (first phase of a tile: the previous tile's epilogue stores lie between; waiting for them too costs one store
                //  drain per tile)
                if constexpr (!(C::KO & 2)) {
                    if constexpr (C::nwait(u, 0) == C::nwait(u, 1)) vc_wait_vmcnt<C::nwait(u, 0)>();
                    else if (grp == 0) vc_wait_vmcnt<C::nwait(u, 0)>();
                    else vc_wait_vmcnt<C::nwait(u, 1)>();
                }
                VC_DMA_BARRIER();
                // -- M: 9 x WM x WN MFMAs while the other group reads; smallest products first --
                if constexpr (!(C::KO & 4)) {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                        for (int pa = 2; pa >= 0; --pa) {
                            const int pb = sum - pa;
                            if (pb < 0 || pb > 2) continue;
#pragma unroll
                            for (int t = 0; t < WM; ++t)
#pragma unroll
                                for (int n = 0; n < WN; ++n)
                                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[pb][n]),
                                                                                        __builtin_bit_cast(bf16x8, af[pa][t]), acc[t][n], 0, 0, 0);
                        }
                    __builtin_amdgcn_s_setprio(0);
                } else {
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                        for (int t = 0; t < WM; ++t) VC_DMA_KEEP(af[pc][t]);
#pragma unroll
                        for (int n = 0; n < WN; ++n) VC_DMA_KEEP(bf[pc][n]);
                    }
                }
                VC_DMA_BARRIER();
            });
        }
        if (grp == 0) VC_DMA_BARRIER();              // (waves 4-7 finish their last phase)
        gchunk += nchunk;
        gbase = (gbase + (unsigned)upt) % RING;

        // ---- epilogue ----
        if constexpr (!(C::KO & 1)) {
            split_epilogue<C>(p, acc, cur, wave, px, q);
        } else {
#pragma unroll
            for (int t = 0; t < WM; ++t)
#pragma unroll
                for (int n = 0; n < WN; ++n) VC_DMA_KEEP(acc[t][n]);
        }
        if (!nxt.valid) break;
        cur = nxt;
        cur_base = nxt_base;
        nxt = tile_at(it + 2);
        nxt_base = tile_base(nxt);
    }
    vc_wait_vmcnt<0>();                              // no DMA may land in LDS that already belongs to another workgroup
}

// ------------------------------------------------------------------------------------------------------------------------------
// PERIOD instances: two consecutive chunks walked as one period, so that a chunk's k-groups need not fill whole units.  K of an MFMA =
// four (tap, 8-channel plane) k-groups; a chunk has KG = K^2 x CPL of them: 3 x 3 (two planes) 18 = 4.5 units, which SplitPairs<3> pads
// to 5; 5 x 5 25 = 6.25 (padded: 7); 7 x 7 49 = 12.25 (padded: 13).  The 2 KG k-groups of TWO chunks are U = ceil(2 KG / 4) units: 9
// instead of 10, 13 instead of 14, 25 instead of 26 (3 x 3: none padded; 5 x 5 / 7 x 7: the last two k-groups of the last unit carry
// zero weights and re-read a valid tap).  k-group q of unit u is number g = 4 u + q -> chunk g / KG of the period, tap (g % KG) / CPL,
// plane (g % KG) % CPL; one unit straddles the two chunks.  The first chunk of a period lives in image buffer X, the second in Y,
// always: the per-lane read offset of a unit (buffer, plane, tap offset of the lane's k-group) is a loop-invariant register.  DMA
// schedule of a period: Y (this period's second chunk) from phase 1 on -- Y was last read in the last phase of the period before, is
// first read in phase YFIRST = KG / 4 --, X (the next period's / tile's first chunk) from phase XLAST + 2 on -- last read in phase
// XLAST = (KG - 1) / 4, first read in the next phase 0; a piece requested in phase j is waited for at the end of phase j + D - 1.
// Same results as the padded instances up to the order of the k-groups inside a chunk pair.  Input channels: a multiple of 16 CPL.
// ------------------------------------------------------------------------------------------------------------------------------
template <int K_, int NTW_, int TH_ = (K_ == 3 ? 12 : 16), int RING_ = 4, int KO_ = 0> struct SplitPeriodCfg {
    static constexpr int K = K_, NTW = NTW_, CPL = K_ == 3 ? 2 : 1, RING = RING_, KO = KO_, PAD = K_ / 2;
    static constexpr int KG = K_ * K_ * CPL, U = (2 * KG + 3) / 4;
    static constexpr int TH = TH_, TW = 32, BN = 16 * NTW_;
    static constexpr int WAVES = 8, WM = TH_ * 2 / 8, WN = NTW_;
    static_assert(WM * 8 == TH_ * 2, "the tile's 2 TH M-tiles split evenly over 8 waves");
    static constexpr int ROWS_IN = TH + K_ - 1, COLS = TW + K_ - 1, PIX = ROWS_IN * COLS, PXB = 48;
    static constexpr int PLANE_SLOTS = CPL == 1 ? 3 * PIX : (3 * PIX + 15) / 16 * 16, PLANE_B = PLANE_SLOTS * 16;   // second plane on a bank row: SplitCfg
    static constexpr int NA = (CPL * PLANE_SLOTS + 511) / 512, A_BYTES = NA * 8192;
    static constexpr int FRAGS = 3 * NTW_, ROUNDS = (FRAGS + 7) / 8, LASTW = FRAGS - 8 * (ROUNDS - 1);
    static constexpr bool EXACT = LASTW == 4;
    static constexpr int SLOTB = EXACT ? FRAGS * 1024 : ROUNDS * 8192;
    static constexpr int nb(int grp) { return ROUNDS - ((EXACT && grp == 1) ? 1 : 0); }
    static constexpr int B_OFF = 2 * A_BYTES, BIAS_OFF = B_OFF + RING_ * SLOTB, LDS_FIXED = BIAS_OFF;
    static constexpr int D = RING_ - 2;
    static_assert(D == 2, "the issue windows below are laid out for weights requested two units ahead");
    static constexpr bool kg_valid(int u, int q) { return 4 * u + q < 2 * KG; }
    static constexpr int kg_chunk(int u, int q) { return kg_valid(u, q) ? (4 * u + q) / KG : 1; }
    static constexpr int kg_tap(int u, int q) { return kg_valid(u, q) ? ((4 * u + q) % KG) / CPL : 0; }
    static constexpr int kg_plane(int u, int q) { return kg_valid(u, q) ? ((4 * u + q) % KG) % CPL : 0; }
    static constexpr int XLAST = (KG - 1) / 4, YFIRST = KG / 4;
    static constexpr int YWIN = YFIRST - D, XWIN = U - D - XLAST - 1;               // phases 1 .. YFIRST - D / XLAST + 2 .. U - D
    static_assert(YWIN >= 1 && XWIN >= 1, "no issue window for a chunk image");
    static constexpr int PPP = (NA + (YWIN < XWIN ? YWIN : XWIN) - 1) / (YWIN < XWIN ? YWIN : XWIN);   // pieces per phase
    // phase in which piece k of buffer Y (which = 1) / X (which = 0) is requested
    static constexpr int piece_phase(int which, int k) { return (which ? 1 : XLAST + 2) + k / PPP; }
    static_assert(piece_phase(1, NA - 1) <= YFIRST - D && piece_phase(0, NA - 1) <= U - D, "chunk image pieces outside their window");
    static constexpr int nA(int u)
    {
        int n = 0;
        for (int w = 0; w < 2; ++w)
            for (int k = 0; k < NA; ++k) n += piece_phase(w, k) == u;
        return n;
    }
    static constexpr int nwait(int u, int grp)
    {
        int n = 0;
        for (int j = u + 2 - D; j <= u; ++j) n += nA(((j % U) + U) % U) + nb(grp);
        return n;
    }
    static constexpr int max_wait()
    {
        int m = 0;
        for (int u = 0; u < U; ++u) m = nwait(u, 0) > m ? nwait(u, 0) : m;
        return m;
    }
    static_assert(max_wait() <= 63, "vmcnt is a 6-bit counter");
};

template <class C> __global__ void __launch_bounds__(512, 2) conv_split_period_kernel(const ConvArgs p)
{
    constexpr int U = C::U, WM = C::WM, WN = C::WN, NA = C::NA, RING = C::RING, D = C::D;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];   // the kernel's only LDS object: starts at LDS address 0
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int px = lane & 15, q = lane >> 4;

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

    const int nper = p.Cin / (16 * C::CPL);                       // periods of two chunks of CPL 8-channel planes
    const int upt = nper * U;
    const unsigned char *const in_b = reinterpret_cast<const unsigned char *>(p.in);
    const unsigned char *const wpk = reinterpret_cast<const unsigned char *>(p.wpk);
    const long long plane = (long long)p.H * p.W * C::PXB;
    const unsigned char *const zero_lane = g_vc_dma_zero + 16 * lane;

    int a_off[NA], a_rc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int s = k * 512 + tid;
        const int pl = s / C::PLANE_SLOTS, s2 = s - pl * C::PLANE_SLOTS;
        const int pix = s2 / 3, piece = s2 - 3 * pix;
        const int row = pix / C::COLS, col = pix - row * C::COLS;
        a_off[k] = pl * (int)plane + (row * p.W + col) * C::PXB + piece * 16;
        a_rc[k] = (pl < C::CPL && s2 < 3 * C::PIX) ? (row | (col << 8)) : 0x7f7f7f;
    }
    auto tile_base = [&](const DmaTile &t) {
        return in_b + (long long)t.img * p.in_sn + ((long long)(t.oy0 - C::PAD) * p.W + (t.ox0 - C::PAD)) * C::PXB;
    };
    auto issue_a = [&](const DmaTile &t, const unsigned char *tbase, int c, int k, int buf) {      // c: chunk (CPL planes) of the layer
        int rc = a_rc[k], off = a_off[k];
        asm volatile("" : "+v"(rc), "+v"(off));
        const unsigned iy = (unsigned)(t.oy0 - C::PAD + (rc & 0xff)), ix = (unsigned)(t.ox0 - C::PAD + (rc >> 8));
        const bool ok = t.valid && iy < (unsigned)p.H && ix < (unsigned)p.W;
        const unsigned char *sp = ok ? tbase + ((long long)c * C::CPL * plane + off) : zero_lane;
        if constexpr (!(C::KO & 16)) vc_glds16<true>(sp, (unsigned)(buf * C::A_BYTES + k * 8192 + wave * 1024));
    };
    const unsigned lane16 = 16 * lane;
    unsigned gbase = 0;
    auto issue_b = [&](int g, int nblk_cur, int nblk_next) {
        const int gg = g >= upt ? g - upt : g;
        const int nblk = g >= upt ? nblk_next : nblk_cur;
        const unsigned char *sbase = wpk + ((long long)nblk * upt + gg) * (C::FRAGS * 1024) + wave * 1024;
        const unsigned dst = C::B_OFF + ((gbase + (unsigned)g) % RING) * C::SLOTB + wave * 1024;
#pragma unroll
        for (int r = 0; r < C::ROUNDS; ++r)
            if constexpr (!(C::KO & 16)) {
                if (r + 1 < C::ROUNDS || !C::EXACT || grp == 0) vc_glds16_sbase(sbase + r * 8192, lane16, dst + r * 8192);
            }
    };

    float *const ldsf = reinterpret_cast<float *>(lds8);
    for (int i = tid; i < p.nblks * C::BN; i += 512) ldsf[C::BIAS_OFF / 4 + i] = p.bias[i];
    DmaTile nxt = tile_at(1);
    const unsigned char *cur_base = tile_base(cur), *nxt_base = tile_base(nxt);
#pragma unroll
    for (int k = 0; k < NA; ++k) issue_a(cur, cur_base, 0, k, 0);
#pragma unroll
    for (int g = 0; g < D; ++g) issue_b(g, cur.nblk, cur.nblk);
    vc_wait_vmcnt<0>();
    __syncthreads();

    // per-lane read offsets: M-tile base + (buffer, plane, tap) of the lane's k-group in each of the U units
    int a_t[WM], a_unit[U];
#pragma unroll
    for (int t = 0; t < WM; ++t) {
        const int m = WM * wave + t;
        a_t[t] = ((m >> 1) * C::COLS + 16 * (m & 1) + px) * C::PXB;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int off = 0;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            if (q == qq)
                off = C::kg_chunk(u, qq) * C::A_BYTES + C::kg_plane(u, qq) * C::PLANE_B + ((C::kg_tap(u, qq) / C::K) * C::COLS + C::kg_tap(u, qq) % C::K) * C::PXB;
        a_unit[u] = off;
    }
    const int b_lane = C::B_OFF + lane * 16;

#ifdef VC_DMA_DIAG
    unsigned st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    VC_DMA_STAMP(t_begin);
    f32x4 acc[WM][WN];
    f32x4 af[3][WM], bf[3][WN];
    for (int it = 0;; ++it) {
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(&ldsf[C::BIAS_OFF / 4 + cur.nblk * C::BN + 16 * n + 4 * q]);
#pragma unroll
            for (int t = 0; t < WM; ++t) acc[t][n] = b;
        }
        if (grp == 1) VC_DMA_BARRIER();
#pragma unroll 1
        for (int pr = 0; pr < nper; ++pr) {
            const bool last_per = pr + 1 == nper;
            const int g0 = pr * U;
            static_for<0, U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                VC_DMA_STAMP(t0);
                const int bslot = b_lane + (int)((gbase + (unsigned)(g0 + u)) % RING) * C::SLOTB;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                    for (int n = 0; n < WN; ++n) bf[pc][n] = *reinterpret_cast<const f32x4 *>(lds8 + bslot + (pc * WN + n) * 1024);
#pragma unroll
                    for (int t = 0; t < WM; ++t) af[pc][t] = *reinterpret_cast<const f32x4 *>(lds8 + (a_t[t] + a_unit[u]) + 16 * pc);
                }
                VC_DMA_STAMP(t1);            // (a stamp waits for lgkmcnt(0): the fragment reads issued AND returned)
                static_for<0, NA>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (C::piece_phase(1, k) == u) issue_a(cur, cur_base, 2 * pr + 1, k, 1);          // this period's second chunk -> Y
                    if constexpr (C::piece_phase(0, k) == u) {                                                   // the next first chunk -> X
                        if (!last_per) issue_a(cur, cur_base, 2 * pr + 2, k, 0);
                        else issue_a(nxt, nxt_base, 0, k, 0);
                    }
                });
                issue_b(g0 + u + D, cur.nblk, nxt.nblk);
                VC_DMA_STAMP(t2);
                if constexpr (!(C::KO & 2)) {
                    if constexpr (C::nwait(u, 0) == C::nwait(u, 1)) vc_wait_vmcnt<C::nwait(u, 0)>();
                    else if (grp == 0) vc_wait_vmcnt<C::nwait(u, 0)>();
                    else vc_wait_vmcnt<C::nwait(u, 1)>();
                }
                VC_DMA_STAMP(t3);
                VC_DMA_BARRIER();
                VC_DMA_STAMP(t4);
                if constexpr (!(C::KO & 4)) {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                        for (int pa = 2; pa >= 0; --pa) {
                            const int pb = sum - pa;
                            if (pb < 0 || pb > 2) continue;
#pragma unroll
                            for (int t = 0; t < WM; ++t)
#pragma unroll
                                for (int n = 0; n < WN; ++n)
                                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[pb][n]),
                                                                                        __builtin_bit_cast(bf16x8, af[pa][t]), acc[t][n], 0, 0, 0);
                        }
                    __builtin_amdgcn_s_setprio(0);
                } else {
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                        for (int t = 0; t < WM; ++t) VC_DMA_KEEP(af[pc][t]);
#pragma unroll
                        for (int n = 0; n < WN; ++n) VC_DMA_KEEP(bf[pc][n]);
                    }
                }
                VC_DMA_STAMP(t5);
                VC_DMA_BARRIER();
                VC_DMA_STAMP(t6);
                VC_DMA_ACC(0, t2, t1);   // DMA issue
                VC_DMA_ACC(1, t1, t0);   // fragment reads
                VC_DMA_ACC(2, t3, t2);   // vmcnt wait
                VC_DMA_ACC(3, t4, t3);   // barrier after R
                VC_DMA_ACC(4, t5, t4);   // MFMA issue
                VC_DMA_ACC(5, t6, t5);   // barrier after M
            });
        }
        if (grp == 0) VC_DMA_BARRIER();
        gbase = (gbase + (unsigned)upt) % RING;
        VC_DMA_STAMP(t_e0);
        if constexpr (!(C::KO & 1)) {
            // (scratch of the record stores: this wave's eighth of buffer Y -- last read in the tile's last phase, next filled from
            //  phase 1 of the next tile on, which no wave reaches before BOTH wave groups have left their epilogues: the group that is
            //  ahead waits at the barrier of its phase 0 for the other one's start-of-tile barrier)
            split_epilogue<C>(p, acc, cur, wave, px, q, (C::KO & 32) ? nullptr : lds8 + C::A_BYTES + wave * (C::A_BYTES / 8));
        } else {
#pragma unroll
            for (int t = 0; t < WM; ++t)
#pragma unroll
                for (int n = 0; n < WN; ++n) VC_DMA_KEEP(acc[t][n]);
        }
        VC_DMA_STAMP(t_e1);
        VC_DMA_ACC(6, t_e1, t_e0);       // epilogue
        if (!nxt.valid) break;
        cur = nxt;
        cur_base = nxt_base;
        nxt = tile_at(it + 2);
        nxt_base = tile_base(nxt);
    }
    vc_wait_vmcnt<0>();
#ifdef VC_DMA_DIAG
    if constexpr (C::KO & 64) {
        VC_DMA_STAMP(t_end);
        st_sum[7] = t_end - t_begin;                 // wave lifetime
        if (lane == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_vc_dma_stamps[i], (unsigned long long)st_sum[i]);
    }
#endif
}

template <class C> int launch_conv_split_period(hipStream_t st, const ConvArgs &a)
{
    const size_t lds_bytes = C::LDS_FIXED + (size_t)a.nblks * C::BN * sizeof(float);
    if (lds_bytes > 160 * 1024) return VC_EINVAL;
    auto kern = conv_split_period_kernel<C>;
    static vc_lds_raised raised;
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}

template <class C> int launch_conv_split(hipStream_t st, const ConvArgs &a)
{
    const size_t lds_bytes = C::LDS_FIXED + (size_t)a.nblks * C::BN * sizeof(float);
    if (lds_bytes > 160 * 1024) return VC_EINVAL;
    auto kern = conv_split_kernel<C>;
    static vc_lds_raised raised;
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}
