// Pointwise (1x1, stride 1) convolution as a streaming GEMM -- tile configuration VC_CFG_PW.
//
// A 1x1 layer has no spatial reuse: 288 FLOP/B for the 3x3 128->128 layer become 32 FLOP/B, i.e. at 128 channels
// the fp32 matrix pipe and HBM saturate at about the same time and on the fp16 path the layer is purely
// bandwidth-bound.  The general kernel (conv_mfma.h) stages every channel chunk through LDS behind two barriers and
// leaves the memory system idle while it computes; here
//   * the weights (<= 64 KiB, the packed fragments of the 32-wide configurations, unchanged) live in LDS for the
//     lifetime of a persistent workgroup,
//   * every wave owns 32-pixel tiles of a row and reads its activations straight from global memory into the MFMA
//     operand layout (lane = pixel, 16 bytes = the 4 (fp32) / 8 (fp16) channels it feeds to one k-step), and
//   * the loads of the NEXT tile are issued before the matrix work of the current one.  The contraction loop touches
//     only LDS and registers, so nothing in it waits on the vector-memory counter and the prefetch really overlaps.
// Accumulation order (k-steps of 8 channels, pairs (e, 4+e) inside one MFMA, bias first) equals the general kernel's:
// results are bit-identical to the other tile configurations of the layer.
#include "conv_mfma.h"

namespace {

template <int KQ, bool F16, bool INH> struct PwRaw {
    // registers holding one tile's activations as loaded: fp32 k-step = 1 x 16 B, fp16 k-step = 2 x 16 B of
    // fp32 data (8 channels each half-wave) or 1 x 16 B when the tensor is stored as half
    static constexpr int PER_STEP = (F16 && !INH) ? 2 : 1;
    f32x4 v[KQ * PER_STEP];
};

// EPI: 0 = plain, 1 = GDN, 2 = IGDN (fp32, cin == cout): the layer contracts x^2 and multiplies x by rsqrt / sqrt of the
// result.  Both the B-operand layout and the accumulator layout give lane half h the channels 8q + 4h + e, so the x a
// lane needs in the epilogue is the very register it squared for k-step q = 4t + g -- no second read of the input.
template <int KQ, int NT, bool F16, bool INH, int EPI = 0>
__global__ void __launch_bounds__(256, 2) conv_pw_kernel(const ConvArgs p)
{
    static_assert(EPI == 0 || (!F16 && KQ == 4 * NT), "GDN needs fp32 and cin == cout");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [NT][KQ][64 lanes][4] weights | bias[NT*32]
    constexpr int KCH = F16 ? 16 : 8;                              // input channels per k-step
    const int kst_total = p.cin_pad / KCH;
    for (int i = threadIdx.x; i < NT * KQ * 64; i += 256) {
        const int ln = i & 63, q = (i >> 6) % KQ, t = (i >> 6) / KQ;
        reinterpret_cast<f32x4 *>(lds)[i] = reinterpret_cast<const f32x4 *>(p.wpk)[((long long)t * kst_total + q) * 64 + ln];
    }
    float *bias_l = lds + NT * KQ * 256;
    for (int i = threadIdx.x; i < NT * 32; i += 256) bias_l[i] = p.bias[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, n = lane & 31;
    const int tiles_x = (p.W + 31) >> 5;
    const long long ntiles = (long long)p.N * p.H * tiles_x;
    const long long stride = (long long)gridDim.x * 4;

    auto locate = [&](long long tile, int &img, int &y, int &x) {
        const int tx = (int)(tile % tiles_x);
        const long long r = tile / tiles_x;
        y = (int)(r % p.H);
        img = (int)(r / p.H);
        x = tx * 32 + n;
    };
    auto load = [&](long long tile, PwRaw<KQ, F16, INH> &raw) {
        int img, y, x;
        locate(tile, img, y, x);
        const long long off = (long long)img * p.in_sn + (long long)y * p.in_sh + (long long)min(x, p.W - 1) * p.in_sw + KCH / 2 * h;
        if (INH) {
            const _Float16 *src = reinterpret_cast<const _Float16 *>(p.in) + off;
#pragma unroll
            for (int q = 0; q < KQ; ++q) raw.v[q] = *reinterpret_cast<const f32x4 *>(src + q * 16);
        } else if (F16) {
            const float *src = p.in + off;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                raw.v[2 * q] = *reinterpret_cast<const f32x4 *>(src + q * 16);
                raw.v[2 * q + 1] = *reinterpret_cast<const f32x4 *>(src + q * 16 + 4);
            }
        } else {
            const float *src = p.in + off;
#pragma unroll
            for (int q = 0; q < KQ; ++q) raw.v[q] = *reinterpret_cast<const f32x4 *>(src + q * 8);
        }
    };

    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    PwRaw<KQ, F16, INH> raw;
    load(tile, raw);
    const float neg = (p.act == VC_ACT_NONE) ? 1.0f : (p.act == VC_ACT_RELU ? 0.0f : p.slope);

    for (; tile < ntiles; tile += stride) {
        // ---- operands of this tile (conversion to half happens here, after the loads have landed) ----
        f32x4 cur[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (F16 && !INH) {
                f16x8 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hv[e] = (_Float16)raw.v[2 * q][e];
                    hv[4 + e] = (_Float16)raw.v[2 * q + 1][e];
                }
                cur[q] = __builtin_bit_cast(f32x4, hv);
            } else {
                cur[q] = raw.v[q];
            }
        }
        // ---- prefetch the next tile; nothing below waits on these loads until the epilogue ----
        if (tile + stride < ntiles) load(tile + stride, raw);

        // The weights are tile-invariant, and the compiler would happily hoist all NT*KQ fragment reads out of the
        // tile loop (256 registers for 128->128); laundering the lane offset once per tile keeps them in LDS.
        int lane4 = lane * 4, h4 = 4 * h;
        asm volatile("" : "+v"(lane4), "+v"(h4));
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(&bias_l[t * 32 + 8 * g + h4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
            }
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 wf = *reinterpret_cast<const f32x4 *>(&lds[(t * KQ + q) * 256 + lane4]);
                if constexpr (F16) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, cur[q]),
                                                                    acc[t], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float b = EPI ? cur[q][e] * cur[q][e] : cur[q][e];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[e], b, acc[t], 0, 0, 0);
                    }
                }
            }

        // ---- epilogue: activation -> channel gain -> residual -> 16-byte (8-byte for half) stores ----
        int img, y, x;
        locate(tile, img, y, x);
        if constexpr (EPI == 0) {
            // COALESCED stores (the accumulator layout gives a lane one pixel: a store instruction would touch 32 lines,
            // 32 bytes each).  Each 32 px x 32 ch accumulator tile passes through this wave's LDS scratch and comes back
            // with consecutive lanes along the channels: lane i = channels 4(i%8).. of pixel 8j + i/8, so an instruction
            // writes whole 128-byte lines of 8 consecutive pixels.  Same value-by-value arithmetic as below.
            // (the 64 KiB of fp32 128x128 weights leave room for half a tile per wave if two workgroups are to share a CU:
            //  PXS pixels go through the scratch at a time)
            constexpr int PXS = (NT * KQ >= 64) ? 16 : 32;
            float *scratch = bias_l + NT * 32 + wave * ((PXS + 1) * VC_EPI_ROWF);
            const int rq = lane & 7, rpx = lane >> 3;
            // The residual pieces of a group (one N-tile x PXS pixels) are requested TOGETHER, one group ahead of their use
            // (addresses clamped, no branch): with one 16-byte load -> wait -> add -> store per piece every wait also sat
            // behind the previous piece's store (loads and stores retire in order on one counter).
            constexpr int PJ = PXS / 8, PARTS = 32 / PXS, NG = NT * PARTS;
            // (the 128-channel fp32 instance only: it runs two waves per SIMD anyway; the narrower, bandwidth-bound instances
            //  keep their registers -- 16 to 32 more would cost them a wave per SIMD -- and load each piece where it is used)
            constexpr bool AHEAD = !F16 && NT * KQ >= 64;
            f32x4 rv[2][PJ];
            auto request = [&](int gidx, f32x4(&dst)[PJ]) {
                const int t = gidx / PARTS, part = gidx % PARTS;
                const int co = min(t * 32 + 4 * rq, p.Cout - 4);
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int xx = min(x - n + part * PXS + 8 * j + rpx, p.W - 1);
                    dst[j] = *reinterpret_cast<const f32x4 *>(p.res + (long long)img * p.res_sn + (long long)y * p.res_sh +
                                                              (long long)xx * p.res_sw + co);
                }
            };
            if (AHEAD && p.res) request(0, rv[0]);
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) {
                const int t = gidx / PARTS, part = gidx % PARTS;
                const int co = t * 32 + 4 * rq;
                if (AHEAD && p.res && gidx + 1 < NG) request(gidx + 1, rv[(gidx + 1) & 1]);
                {
                    // every lane writes, unconditionally: lanes of the other part go to a dump row.  (A divergent `if` around
                    // the writes lets the compiler duplicate the reads below into both paths -- lanes that skip the branch
                    // would then read before the others have written; the exchange needs the wave converged.)
                    const int row = (PXS == 32 || (n / PXS) == part) ? (n % PXS) : PXS;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                        *reinterpret_cast<f32x4 *>(&scratch[row * VC_EPI_ROWF + 8 * g + 4 * h]) = v;
                    }
#pragma unroll
                    for (int j = 0; j < PJ; ++j) {
                        const int pl = 8 * j + rpx;                       // pixel within the part
                        f32x4 v = *reinterpret_cast<const f32x4 *>(&scratch[pl * VC_EPI_ROWF + 4 * rq]);
                        const int xx = x - n + part * PXS + pl;           // x of this lane's read-back pixel
                        if (xx < p.W && co < p.Cout) {
                            const long long o_pix = (long long)img * p.out_sn + (long long)y * p.out_sh + (long long)xx * p.out_sw;
                            f32x4 r = rv[gidx & 1][j];
                            if (!AHEAD && p.res)
                                r = *reinterpret_cast<const f32x4 *>(p.res + (long long)img * p.res_sn + (long long)y * p.res_sh +
                                                                     (long long)xx * p.res_sw + co);
                            if (p.res_first) v += r;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                            if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + co);
                            if (p.res && !p.res_first) v += r;
                            if (F16 && p.out_f16) {
                                const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                                *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(p.out) + o_pix + co) = hv;
                            } else {
                                *reinterpret_cast<f32x4 *>(p.out + o_pix + co) = v;
                            }
                        }
                    }
                }
            }
        } else if (x < p.W) {
            const long long o_pix = (long long)img * p.out_sn + (long long)y * p.out_sh + (long long)x * p.out_sw;
            const long long r_pix = (long long)img * p.res_sn + (long long)y * p.res_sh + (long long)x * p.res_sw;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = t * 32 + 8 * g + 4 * h;
                    if (co < p.Cout) {
                        f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                        if constexpr (EPI != 0) {       // IEEE sqrt and divide, like the general kernel
                            const f32x4 xin = cur[(4 * t + g) < KQ ? 4 * t + g : 0];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (EPI == 1) ? xin[e] * (1.0f / sqrtf(v[e])) : xin[e] * sqrtf(v[e]);
                        }
                        if (p.res_first) v += *reinterpret_cast<const f32x4 *>(p.res + r_pix + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.0f ? v[e] : v[e] * neg;
                        if (p.chscale) v *= *reinterpret_cast<const f32x4 *>(p.chscale + co);
                        if (p.res && !p.res_first) v += *reinterpret_cast<const f32x4 *>(p.res + r_pix + co);
                        if (F16 && p.out_f16) {
                            const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(p.out) + o_pix + co) = hv;
                        } else {
                            *reinterpret_cast<f32x4 *>(p.out + o_pix + co) = v;
                        }
                    }
                }
        }
    }
}

template <int KQ, int NT, bool F16, bool INH, int EPI = 0> int launch_pw(hipStream_t st, const ConvArgs &a)
{
    const size_t lds_bytes = (size_t)(NT * KQ * 256 + NT * 32 + (EPI == 0 ? 4 * (((NT * KQ >= 64) ? 16 : 32) + 1) * VC_EPI_ROWF : 0)) * sizeof(float);
    auto kern = conv_pw_kernel<KQ, NT, F16, INH, EPI>;
    static vc_lds_raised raised;          // largest size raised per device (common.h)
    if (!vc_raise_lds_limit(reinterpret_cast<const void *>(kern), lds_bytes, raised)) return VC_ELAUNCH;
    const long long ntiles = (long long)a.N * a.H * ((a.W + 31) / 32);
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;               // persistent workgroups: up to four per CU (registers/LDS allow 2-4)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
}

template <int KQ, bool F16, bool INH> int by_nt(hipStream_t st, const ConvArgs &a)
{
    switch ((a.Cout + 31) / 32) {
    case 1: return launch_pw<KQ, 1, F16, INH>(st, a);
    case 2: return launch_pw<KQ, 2, F16, INH>(st, a);
    case 3: return launch_pw<KQ, 3, F16, INH>(st, a);
    case 4: return launch_pw<KQ, 4, F16, INH>(st, a);
    }
    return VC_EINVAL;
}

template <bool F16, bool INH> int by_kq(hipStream_t st, const ConvArgs &a)
{
    constexpr int KCH = F16 ? 16 : 8;
    if (a.Cin % KCH) return VC_EINVAL;
    const int steps = a.Cin / KCH;      // 32, 64, 96 or 128 channels
    if constexpr (F16) {
        switch (steps) {
        case 2: return by_nt<2, F16, INH>(st, a);
        case 4: return by_nt<4, F16, INH>(st, a);
        case 6: return by_nt<6, F16, INH>(st, a);
        case 8: return by_nt<8, F16, INH>(st, a);
        }
    } else {
        switch (steps) {
        case 4: return by_nt<4, F16, INH>(st, a);
        case 8: return by_nt<8, F16, INH>(st, a);
        case 12: return by_nt<12, F16, INH>(st, a);
        case 16: return by_nt<16, F16, INH>(st, a);
        }
    }
    return VC_EINVAL;
}

}  // namespace

bool conv_pw_eligible(const ConvArgs &a, int k, int stride, bool f16)
{
    if (k != 1 || stride != 1 || a.out_mode != VC_OUT_PLAIN) return false;
    if (a.epi != VC_EPI_NONE) {   // GDN / IGDN: x^2 contraction whose multiplier is the layer's own input, 128 channels, fp32
        return !f16 && a.in_xform == VC_IN_SQUARE && a.act == VC_ACT_NONE && a.Cin == 128 && a.Cout == 128 && a.vec4 && a.vec_out &&
               a.mul == a.in && a.mul_sn == a.in_sn && a.mul_sh == a.in_sh && a.mul_sw == a.in_sw && !a.res_first;
    }
    if (a.in_xform != VC_IN_NONE) return false;
    if (a.act != VC_ACT_NONE && a.act != VC_ACT_RELU && a.act != VC_ACT_LRELU) return false;
    if (!a.vec4 || !a.vec_out || (a.Cout % 4) || a.Cout > 128 || a.Cin > 128 || a.Cin < 32) return false;
    const int kch = f16 ? 16 : 8;
    if (a.Cin % kch) return false;
    const int steps = a.Cin / kch;
    return f16 ? (steps == 2 || steps == 4 || steps == 6 || steps == 8) : (steps == 4 || steps == 8 || steps == 12 || steps == 16);
}

int conv_dispatch_pw(hipStream_t st, const ConvArgs &a, bool f16)
{
    if (a.epi == VC_EPI_GDN) return launch_pw<16, 4, false, false, 1>(st, a);
    if (a.epi == VC_EPI_IGDN) return launch_pw<16, 4, false, false, 2>(st, a);
    if (!f16) return by_kq<false, false>(st, a);
    return a.in_f16 ? by_kq<true, true>(st, a) : by_kq<true, false>(st, a);
}
