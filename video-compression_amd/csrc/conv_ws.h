// Convolution with PRODUCER / CONSUMER waves and a double-buffered input tile (VC_CFG_WS), fp16 path and fp32 path.
//
// Why a second kernel shape for the fp16 path: v_mfma_f32_32x32x16_f16 retires a channel chunk in 1/8 of the fp32 time,
// so in conv_mfma_kernel's  stage -> barrier -> contract  sequence every staging round (a full memory round trip,
// 2-3 us under load) is as long as the contraction it feeds -- a workgroup of the classic kernel lives ~130k cycles for
// 25k cycles of matrix work (7x7 64->32).  Prefetching the next chunk from the SAME wave does not help: vector-memory
// results retire in order, so the loop's own weight-fragment waits would wait for the prefetch too.  Different waves
// have different counters:
//
//   waves 4..7 (producers): global -> (half conversion) -> LDS buffer (i+1)&1   for item i+1
//   waves 0..3 (consumers): weight-fragment ring + MFMA over LDS buffer i&1      for item i
//   one s_barrier per item; an item = (output tile, channel chunk), tiles walked by a persistent workgroup per CU
//
// so the memory round trips of item i+1 -- including the first chunk of the NEXT tile, behind the epilogue -- hide
// behind the contraction of item i, and the consumer waves hold no staging registers (the 128-channel tile does not
// spill here).  Same tile geometry, LDS image, packed weights, accumulation order and epilogue as conv_mfma_kernel:
// results are bit-identical to the classic fp16 kernel (tests/test_ops_gpu.py).
#pragma once
#include "conv_mfma.h"

// LDS-only synchronisation: __syncthreads() would also drain vmcnt, i.e. wait for the consumer's prefetched weight
// fragments (and the producer's in-flight loads) at every item.
#ifndef VC_WS_RING32
#define VC_WS_RING32 2          // fp32 path: weight fragments requested up to this many k-steps ahead
#endif
#define VC_WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One (tile, chunk) item "in flight" in a producer thread's registers: issue() starts the global loads of the whole
// footprint in ONE round (the producers hold no accumulators), commit() converts and writes the LDS image once the
// target buffer is free.  Between the two the thread waits at the item barrier, so the memory round trip of item i+2
// overlaps the consumers' work on item i -- same LDS image as stage_chunk (conv_mfma.h), item for item.
template <int KH, int KW, int S, int CK, class C, bool F16> struct WsStage {
    typedef ConvGeom<KH, KW, S, CK, C> G;
    static constexpr int C4 = CK / 4, ITEMS = G::ROWS_IN * G::COLS_IN * C4, IPT = (ITEMS + 255) / 256;
    static constexpr int CPI = F16 ? 8 : 4;                   // input channels behind one 16-byte LDS item
    f32x4 v[IPT], v2[F16 ? IPT : 1];
    unsigned ok;

    template <bool INH>
    __device__ __forceinline__ void issue(const ConvArgs &p, const float *in_img, int c0, int oy0, int ox0, int iy0, int ix0, int tid)
    {
        ok = 0u;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int idx = tid + j * 256;
            const int c4 = idx % C4, pc = idx / C4;
            const int col = pc % G::COLS_IN, row = pc / G::COLS_IN;
            const int iy = G::POINT ? (oy0 + row) * S : iy0 + row;
            const int ix = G::POINT ? (ox0 + col) * S : ix0 + col;
            const int ch = c0 + c4 * CPI;
            if ((idx < ITEMS) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin) ok |= 1u << j;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            if constexpr (!F16) {      // fp32 path: one 16-byte load = the 4 channels of the item (Cin % 4 == 0: p.vec4)
                v[j] = *reinterpret_cast<const f32x4 *>(in_img + (long long)iyc * p.in_sh + (long long)ixc * p.in_sw + min(ch, p.Cin - 4));
            } else if (INH) {      // in_img counts halves
                const _Float16 *q = reinterpret_cast<const _Float16 *>(in_img) + (long long)iyc * p.in_sh + (long long)ixc * p.in_sw +
                                    min(ch, p.Cin - 8);
                v[j] = *reinterpret_cast<const f32x4 *>(q);
            } else {
                const float *q = in_img + (long long)iyc * p.in_sh + (long long)ixc * p.in_sw + min(ch, p.Cin - 8);
                v[j] = *reinterpret_cast<const f32x4 *>(q);
                v2[j] = *reinterpret_cast<const f32x4 *>(q + 4);
            }
        }
    }

    template <bool INH> __device__ __forceinline__ void commit(float *buf, int tid) const
    {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int idx = tid + j * 256;
            const int c4 = idx % C4, pc = idx / C4;
            const int col = pc % G::COLS_IN, row = pc / G::COLS_IN;
            const int pos = (G::LS == 2) ? ((col & 1) * G::HALF + (col >> 1)) : col;
            const int dst = (idx < ITEMS) ? (row * G::COLS_L + pos) * G::CKP + c4 * 4 : G::LDS_FLOATS;   // (dump slot)
            f32x4 w = v[j];
            if (F16 && !INH) {
                f16x8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (_Float16)v[j][e];
                    h[4 + e] = (_Float16)v2[j][e];
                }
                w = __builtin_bit_cast(f32x4, h);
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(&buf[dst]) = ((ok >> j) & 1u) ? w : z;
        }
    }
};

template <int KH, int KW, int S, int CK, class C, bool F16>
__global__ void __launch_bounds__(512, 2) conv_ws_kernel(const ConvArgs p)
{
    typedef ConvGeom<KH, KW, S, CK, C> G;
    typedef Mfma<C::MT> M;
    static_assert(C::MT == 32 || C::MT == 16, "32- and 16-wide tiles");
    constexpr int MT = C::MT, WM = C::WM, WN = C::WN, KS = G::KS, KSTEPS = G::KSTEPS, NT = C::NT;
    constexpr int CKC = F16 ? 2 * CK : CK, KSC = F16 ? 2 * KS : KS, TAPS = KH * KW, STEPS_X = KW * KSTEPS;
    constexpr int BUF = G::LDS_FLOATS + 4;                 // one input-tile image + the staging dump slot
    constexpr bool COALESCED = C::MT == 32;                // coalesced epilogue through a per-wave LDS scratch behind the buffers
    // HAND-OFF epilogue (32-channel tiles whose fp32 result fits the tile buffer that has just been consumed): the
    // consumers only dump their accumulators into that buffer; the PRODUCER waves apply the epilogue and store.  A store
    // issued by a consumer wave would sit in the same vmcnt as its weight-fragment ring -- every ring wait of the next
    // item would wait for the writes to be acknowledged (stamps: the "epilogue" was 20-40 % of a consumer's life).
    constexpr bool HANDOFF = C::MT == 32 && WN == 1 && C::WAVES_N == 1 && C::TH * C::TW * 32 <= G::LDS_FLOATS;
    static_assert((2 * BUF + 4 * VC_EPI_SCRATCH_FLOATS + 4) * 4 <= 160 * 1024, "two tile images + scratch + flag must fit the CU's LDS");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // ---- persistent workgroup: block b serves XCD b&7; the XCD's tile range is contiguous (neighbouring tiles share
    // halos in that XCD's L2), its blocks take the range round-robin so that tiles in flight together are adjacent ----
    const int nb = p.total_blocks, grid = gridDim.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = q + (xcd < r ? 1 : 0);
    const int peers = (grid + 7 - xcd) >> 3;               // blocks of this launch on the same XCD
    const int n_tiles = local < count ? (count - local + peers - 1) / peers : 0;
    const int nch = p.cin_pad / CKC;
    const int n_items = n_tiles * nch;
    if (n_items == 0) return;

    auto decode = [&](int item, int &nblk, int &oy0, int &ox0, int &img, int &c0) {
        int bid = first + local + (item / nch) * peers;
        c0 = (item % nch) * CKC;
        nblk = bid % p.nblks;
        const int t1 = bid / p.nblks;
        img = t1 / (p.tiles_x * p.tiles_y);
        int tx, ty;
        vc_tile_xy(t1 - img * (p.tiles_x * p.tiles_y), p.tiles_x, p.tiles_y, p.tile_band, tx, ty);
        oy0 = ty * C::TH;
        ox0 = tx * C::TW;
    };

    if (threadIdx.x >= 256) {
        // =============================== producers ===============================
        const int tid = threadIdx.x - 256;
        WsStage<KH, KW, S, CK, C, F16> st;
        const bool inh = F16 && p.in_f16 != 0;
        auto issue_item = [&](int item) {
            int nblk, oy0, ox0, img, c0;
            decode(item, nblk, oy0, ox0, img, c0);
            const int iy0 = oy0 * S - KH / 2, ix0 = ox0 * S - KW / 2;
            if (inh)
                st.template issue<true>(p, reinterpret_cast<const float *>(reinterpret_cast<const _Float16 *>(p.in) + (long long)img * p.in_sn),
                                        c0, oy0, ox0, iy0, ix0, tid);
            else
                st.template issue<false>(p, p.in + (long long)img * p.in_sn, c0, oy0, ox0, iy0, ix0, tid);
        };
        auto commit_item = [&](int item) {
            float *dst = lds + (item & 1) * BUF;
            if (inh) st.template commit<true>(dst, tid);
            else st.template commit<false>(dst, tid);
        };
        // hand-off epilogue of the tile whose last chunk was `item`: rows of [pixel][32 channels] fp32 accumulators
        // (16-byte slots XOR-swizzled by the pixel: conflict-free for the consumers' writes and these reads) ->
        // activation / gain / residual -> coalesced stores, 8 consecutive pixels x 128 bytes per wave instruction
        auto store_tile = [&](int item) {
            int nblk, oy0, ox0, img, c0;
            decode(item, nblk, oy0, ox0, img, c0);
            const float *src = lds + (item & 1) * BUF;
            const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);
            const int mode = p.act == VC_ACT_SIGMOID ? 3 : (p.act == VC_ACT_CLAMP01 ? 4 : 0);
            const int cps = p.Cout >> 2;
            const bool ps = p.out_mode != VC_OUT_PLAIN;
            const int sc = ps ? 2 : 1;
            const int q = tid & 7;
            const int co = nblk * C::BN + 4 * q;
            constexpr int QUADS = C::TH * C::TW * 8;
#pragma unroll 4
            for (int k = tid; k < QUADS; k += 256) {
                const int pixel = k >> 3;                       // m * 32 + px, m = row * XT + xt
                const int m = pixel >> 5, px = pixel & 31;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&src[pixel * 32 + 4 * (q ^ ((px >> 1) & 7))]);
                const int oy = oy0 + m / C::XT, ox = ox0 + (m % C::XT) * MT + px;
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
        };
        const bool handoff = HANDOFF && p.vec_out && p.epi == VC_EPI_NONE;
        // producer-only rendezvous (an s_barrier would involve the consumers): a counter in LDS behind the scratch
        int *flag = reinterpret_cast<int *>(lds + 2 * BUF + 4 * VC_EPI_SCRATCH_FLOATS);
        int epoch = 0;
        if (tid == 0) *flag = 0;
        issue_item(0);
        commit_item(0);
        if (n_items > 1) issue_item(1);
        VC_WS_BARRIER();                      // item 0 is staged (and the counter initialised)
        for (int item = 0; item < n_items; ++item) {
            // the consumers contract `item`; its successor's loads were issued one iteration ago
            VC_T(p_a);
            if (item + 1 < n_items) commit_item(item + 1);
            VC_T(p_b);
            if (item + 2 < n_items) issue_item(item + 2);
            VC_T(p_c);
            VC_WS_BARRIER();
            VC_T(p_d);
            VC_ACC(0, p_b, p_a);              // producers: commit (wait for the loads + convert + LDS writes)
            VC_ACC(1, p_c, p_b);              // producers: issuing the next item's loads
            VC_ACC(7, p_d, p_c);              // producers: waiting for the consumers at the item barrier
            if (handoff && (item % nch) == nch - 1) {
                VC_WS_BARRIER();              // the consumers have dumped the tile into buffer item&1
                store_tile(item);
                // every producer wave must have READ its part of the dump before any of them stages item+2 over it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if ((tid & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                epoch += 4;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
        }
        return;
    }

    // =============================== consumers ===============================
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
    const int li = M::arow(lane), kk = M::akk(lane);
    int abase[WM];
#pragma unroll
    for (int t = 0; t < WM; ++t) {
        const int m = wm * WM + t;
        const int row = m / C::XT, xt = m % C::XT;
        abase[t] = ((row * G::LS) * G::COLS_L + (xt * MT + li)) * G::CKP + 4 * kk;
    }
    const int ksteps_total = p.cin_pad / KSC;
    const long long ntile_stride = (long long)TAPS * ksteps_total * 256;
    auto wbase = [&](int nblk, int c0) {     // this lane's slice of the fragments of (N-block, chunk)
        return p.wpk + (long long)(nblk * (C::BN / NT) + wn * WN) * ntile_stride + lane * 4 + (long long)(c0 / KSC) * 256;
    };
    auto load_b = [&](f32x4(&b)[WN], const float *wrow, int sx) {
        const int kx = sx / KSTEPS, ks = sx % KSTEPS;
#pragma unroll
        for (int n = 0; n < WN; ++n)
            b[n] = *reinterpret_cast<const f32x4 *>(wrow + n * ntile_stride + ((long long)kx * ksteps_total + ks) * 256);
    };

    // consumers keep accumulators, three A fragments and the ring -- no staging registers
    // (one consumer wave per SIMD: nobody else covers an L2 round trip of ~1500 cycles -> fp16: up to a whole kernel row
    //  ahead; an fp32 step is 8x longer: two steps suffice)
    constexpr int RING = vc_ring_depth(STEPS_X, 224 - WM * WN * M::NREG - 12 * WM, WN, F16 ? STEPS_X : VC_WS_RING32);
    f32x4 ring[RING][WN];
    typename M::acc_t acc[WM][WN];

    const bool handoff = HANDOFF && p.vec_out && p.epi == VC_EPI_NONE;
    int nblk, oy0, ox0, img, c0;
    decode(0, nblk, oy0, ox0, img, c0);
    const float *wchunk = wbase(nblk, c0);
#pragma unroll
    for (int d = 0; d < RING; ++d) load_b(ring[d], wchunk, d);     // steps 0 .. RING-1 of the first item (RING <= STEPS_X)
    VC_WS_BARRIER();                                               // item 0 is staged

    VC_T(t_start);
    for (int item = 0; item < n_items; ++item) {
        VC_T(t_a);
        const float *buf = lds + (item & 1) * BUF;
        if (c0 == 0) {                                             // first chunk of a tile: accumulators = bias
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int rg = 0; rg < M::NREG; ++rg) {
                    const float b = p.bias[nblk * C::BN + (wn * WN + n) * NT + M::crow(rg, lane)];
#pragma unroll
                    for (int t = 0; t < WM; ++t) acc[t][n][rg] = b;
                }
        }
        // where the ring continues after this item's last kernel row: the next item's fragments (or, at the very end,
        // this item's again -- a harmless extra load)
        int nblk_n = nblk, oy0_n = oy0, ox0_n = ox0, img_n = img, c0_n = c0;
        if (item + 1 < n_items) decode(item + 1, nblk_n, oy0_n, ox0_n, img_n, c0_n);
        const float *wchunk_n = wbase(nblk_n, c0_n);

        auto load_a = [&](f32x4(&a)[WM], int rowoff, int sx) {
            const int kx = sx / KSTEPS, ks = sx % KSTEPS;
            const int koff = (G::LS == 2) ? ((kx & 1) * G::HALF + (kx >> 1)) * G::CKP : kx * G::CKP;
#pragma unroll
            for (int t = 0; t < WM; ++t) a[t] = *reinterpret_cast<const f32x4 *>(&buf[abase[t] + rowoff + koff + ks * KS]);
        };
        // activations two steps ahead (one consumer wave per SIMD: nobody else covers an LDS round trip)
        f32x4 a0[WM], a1[WM], a2[WM];
        load_a(a0, 0, 0);
        load_a(a1, 0, STEPS_X > 1 ? 1 : 0);
#pragma unroll 1
        for (int ky = 0; ky < KH; ++ky) {
            const float *wrow = wchunk + (long long)ky * KW * ksteps_total * 256;
            const float *wrow_n = ky + 1 < KH ? wrow + (long long)KW * ksteps_total * 256 : wchunk_n;
            const int rowoff = ky * G::COLS_L * G::CKP;
            const int rowoff_n = (ky + 1 < KH ? ky + 1 : ky) * G::COLS_L * G::CKP;   // (last row: re-reads itself, unused)
            static_for<0, STEPS_X>([&](auto sc) {
                constexpr int sx = decltype(sc)::value;
                constexpr int slot = sx % RING;
                if constexpr (sx + 2 < STEPS_X) load_a(a2, rowoff, sx + 2);
                else load_a(a2, rowoff_n, (sx + 2 - STEPS_X) % STEPS_X);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) {
#pragma unroll
                    for (int t = 0; t < WM; ++t)
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            if constexpr (MT == 32)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ring[slot][n]),
                                                                                   __builtin_bit_cast(f16x8, a0[t]), acc[t][n], 0, 0, 0);
                            else
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[slot][n]),
                                                                                   __builtin_bit_cast(f16x8, a0[t]), acc[t][n], 0, 0, 0);
                        }
                } else {        // exact fp32 FMA chains, in conv_mfma_kernel's order: e -> M-tile -> N-tile
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < WM; ++t)
#pragma unroll
                            for (int n = 0; n < WN; ++n) acc[t][n] = M::run(ring[slot][n][e], a0[t][e], acc[t][n]);
                }
                if constexpr (sx + RING < STEPS_X) load_b(ring[slot], wrow, sx + RING);
                else load_b(ring[slot], wrow_n, sx + RING - STEPS_X);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < WM; ++t) {
                    a0[t] = a1[t];
                    a1[t] = a2[t];
                }
            });
        }
        VC_T(t_b);
        const bool last_chunk = c0 + CKC >= p.cin_pad;
        if (last_chunk && !handoff) {
            if constexpr (COALESCED) {
                if (p.vec_out && p.epi == VC_EPI_NONE)
                    conv_epilogue_coalesced<C>(p, acc, nblk, wm, wn, lane, oy0, ox0, img, lds + 2 * BUF + wave * VC_EPI_SCRATCH_FLOATS);
                else
                    conv_epilogue<C, F16>(p, acc, nblk, wm, wn, lane, oy0, ox0, img);
            } else {
                conv_epilogue<C, F16>(p, acc, nblk, wm, wn, lane, oy0, ox0, img);
            }
        }
        nblk = nblk_n; oy0 = oy0_n; ox0 = ox0_n; img = img_n; c0 = c0_n;
        wchunk = wchunk_n;
        VC_T(t_c);
        VC_WS_BARRIER();            // buffer item&1 may be overwritten, buffer (item+1)&1 is staged
        if constexpr (HANDOFF) {
            if (last_chunk && handoff) {       // dump the accumulators into the buffer just consumed, for the producers
                float *dst = lds + (item & 1) * BUF;
                const int wpx = lane & 31, whalf = lane >> 5, sw = (wpx >> 1) & 7;
#pragma unroll
                for (int t = 0; t < WM; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[t][0][4 * g], acc[t][0][4 * g + 1], acc[t][0][4 * g + 2], acc[t][0][4 * g + 3]};
                        *reinterpret_cast<f32x4 *>(&dst[((wm * WM + t) * 32 + wpx) * 32 + 4 * ((2 * g + whalf) ^ sw)]) = v;
                    }
                VC_WS_BARRIER();
            }
        }
        VC_T(t_d);
        VC_ACC(3, t_b, t_a);        // contraction of the item
        VC_ACC(4, t_c, t_b);        // epilogue (last chunk of a tile only)
        VC_ACC(2, t_d, t_c);        // waiting for the producers at the item barrier
    }
    VC_T(t_end);
    VC_ACC(5, t_end, t_start);      // consumer wave lifetime
    VC_ACC(6, 1ull, 0ull);          // consumer waves
}

template <int KH, int KW, int S, int CK, class C, bool F16> int launch_conv_ws(hipStream_t st, const ConvArgs &a)
{
    typedef ConvGeom<KH, KW, S, CK, C> G;
    constexpr size_t lds_bytes = (2 * (size_t)(G::LDS_FLOATS + 4) + 4 * VC_EPI_SCRATCH_FLOATS + 4) * sizeof(float);
    auto kern = conv_ws_kernel<KH, KW, S, CK, C, F16>;
    static std::atomic<uint64_t> raised{0};
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    const int grid = a.total_blocks < 256 ? a.total_blocks : 256;      // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}
