// Entropy-model kernels: fused quantise + likelihood + (-log2) + reduction, and the symboliser /
// de-quantiser that sit either side of the host range coder.
//
//   factorised prior  (CompressAI EntropyBottleneck): p = |sigmoid(s*u) - sigmoid(s*l)| with
//                     u,l = logistic-CDF logits of z_hat +- 1/2 through a per-channel 1-3-3-3-3-1 MLP
//   Gaussian conditional (GaussianConditional): p = Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), Phi via erfc
//
// Both are HBM-bound streaming kernels (one read of each operand, one write of the quantised tensor);
// the bit count is reduced wavefront -> workgroup -> one double per workgroup (deterministic), and
// vc_bits_reduce folds the partials.  Symbols/indexes are written in the (n,c,y,x) order the range
// coder consumes (CompressAI flattens NCHW tensors).
#include "common.h"

#define ENT_BLOCK 256
#define ENT_SLOTS 1024

extern "C" int vc_bits_slots(void) { return ENT_SLOTS; }

__device__ __forceinline__ double block_sum(double v, double *sm)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < ENT_BLOCK / 64; ++w) r += sm[w];
    return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// logits of the cumulative at x for one channel (params: see VC_EB_PARAMS_PER_CHANNEL in vc_hip.h)
__device__ __forceinline__ float eb_logits(const float *__restrict__ q, float x)
{
    float l[3], m[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = q[i] * x + q[3 + i];
        l[i] = t + q[6 + i] * tanhf(t);
    }
    const float *r = q + 9;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float t = r[3 * i] * l[0];
            t += r[3 * i + 1] * l[1];
            t += r[3 * i + 2] * l[2];
            t += r[9 + i];
            m[i] = t + r[12 + i] * tanhf(t);
        }
        l[0] = m[0]; l[1] = m[1]; l[2] = m[2];
        r += 15;
    }
    float t = r[0] * l[0];
    t += r[1] * l[1];
    t += r[2] * l[2];
    return t + r[3];
}

__global__ void __launch_bounds__(ENT_BLOCK) k_eb_forward(vc_view z, const float *__restrict__ params,
                                                          const float *__restrict__ in_gain,
                                                          const float *__restrict__ out_gain, vc_view zh,
                                                          int32_t *__restrict__ symbols, double *__restrict__ partial,
                                                          float *__restrict__ likelihoods)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double bits = 0.0;
    const long long total = (long long)z.n * z.h * z.w * z.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % z.c);
        long long t = i / z.c;
        const int x = (int)(t % z.w); t /= z.w;
        const int y = (int)(t % z.h);
        const int n = (int)(t / z.h);
        const float *q = params + (long long)c * VC_EB_PARAMS_PER_CHANNEL;
        const float med = q[58];
        float v = z.p[view_off(z, n, y, x) + c];
        if (in_gain) v *= in_gain[c];
        const float sym = rintf(v - med);             // torch.round: half to even
        const float zq = sym + med;
        const float lower = eb_logits(q, zq - 0.5f), upper = eb_logits(q, zq + 0.5f);
        const float sum = lower + upper;
        const float sg = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
        float lik = fabsf(sigmoidf_(sg * upper) - sigmoidf_(sg * lower));
        lik = fmaxf(lik, 1e-9f);
        bits -= (double)log2f(lik);
        if (zh.p) zh.p[view_off(zh, n, y, x) + c] = out_gain ? zq * out_gain[c] : zq;
        const long long o = (((long long)n * z.c + c) * z.h + y) * z.w + x;
        if (symbols) symbols[o] = (int32_t)sym;
        if (likelihoods) likelihoods[o] = lik;
    }
    const double r = block_sum(bits, sm);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = r;
}

extern "C" int vc_eb_forward(vc_stream s, vc_view z, const float *params, const float *in_gain, const float *out_gain,
                             vc_view z_hat, int32_t *symbols, double *bits_partial, int bits_slots, float *likelihoods)
{
    if (!z.p || !params) return VC_EINVAL;
    if (bits_partial && bits_slots != ENT_SLOTS) return VC_EINVAL;
    hipLaunchKernelGGL(k_eb_forward, dim3(ENT_SLOTS), dim3(ENT_BLOCK), 0, as_stream(s), z, params, in_gain, out_gain,
                       z_hat, symbols, bits_partial, likelihoods);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_eb_dequant(const int32_t *__restrict__ symbols, const float *__restrict__ params,
                             const float *__restrict__ out_gain, vc_view zh)
{
    const long long total = (long long)zh.n * zh.h * zh.w * zh.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % zh.c);
        long long t = i / zh.c;
        const int x = (int)(t % zh.w); t /= zh.w;
        const int y = (int)(t % zh.h);
        const int n = (int)(t / zh.h);
        const float med = params[(long long)c * VC_EB_PARAMS_PER_CHANNEL + 58];
        const float v = (float)symbols[(((long long)n * zh.c + c) * zh.h + y) * zh.w + x] + med;
        zh.p[view_off(zh, n, y, x) + c] = out_gain ? v * out_gain[c] : v;
    }
}

extern "C" int vc_eb_dequant(vc_stream s, const int32_t *symbols, const float *params, const float *out_gain, vc_view z_hat)
{
    if (!symbols || !params || !z_hat.p) return VC_EINVAL;
    const long long total = (long long)z_hat.n * z_hat.h * z_hat.w * z_hat.c;
    hipLaunchKernelGGL(k_eb_dequant, dim3(ew_grid(total, ENT_BLOCK)), dim3(ENT_BLOCK), 0, as_stream(s), symbols, params, out_gain, z_hat);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__device__ __forceinline__ int scale_index(float s, const float *__restrict__ table, int n_scales)
{
    // build_indexes: (n_scales-1) - #{t in table[:-1] : s <= t}; table is ascending, so count from the top
    int idx = n_scales - 1;
    for (int k = 0; k < n_scales - 1; ++k) idx -= (s <= table[k]) ? 1 : 0;
    return idx;
}

__global__ void __launch_bounds__(ENT_BLOCK) k_gc_forward(vc_view yv, vc_view sc, vc_view mu, const float *__restrict__ in_gain,
                                                          const float *__restrict__ out_gain, vc_view yh,
                                                          double *__restrict__ partial, const float *__restrict__ sym_src,
                                                          int32_t *__restrict__ symbols, int32_t *__restrict__ indexes,
                                                          const float *__restrict__ table, int n_scales,
                                                          float *__restrict__ likelihoods)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double bits = 0.0;
    const float kc = -0.70710678118654752440f;  // float(-(2 ** -0.5))
    const long long total = (long long)yv.n * yv.h * yv.w * yv.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % yv.c);
        long long t = i / yv.c;
        const int x = (int)(t % yv.w); t /= yv.w;
        const int y = (int)(t % yv.h);
        const int n = (int)(t / yv.h);
        const long long oy = view_off(yv, n, y, x) + c;
        float v = yv.p[oy];
        if (in_gain) v *= in_gain[c];
        const float m = mu.p[view_off(mu, n, y, x) + c];
        float s = sc.p[view_off(sc, n, y, x) + c];
        s = fmaxf(s, 0.11f);                          // lower_bound_scale
        const float q = rintf(v - m);
        const float yq = q + m;
        const float a = fabsf(yq - m);                // the reference subtracts the mean again
        const float upper = 0.5f * erfcf(kc * ((0.5f - a) / s));
        const float lower = 0.5f * erfcf(kc * ((-0.5f - a) / s));
        const float lik = fmaxf(upper - lower, 1e-9f);
        bits -= (double)log2f(lik);
        if (yh.p) yh.p[view_off(yh, n, y, x) + c] = out_gain ? yq * out_gain[c] : yq;
        const long long o = (((long long)n * yv.c + c) * yv.h + y) * yv.w + x;
        if (symbols) symbols[o] = sym_src ? (int32_t)rintf(sym_src[oy] - m) : (int32_t)q;
        if (indexes) indexes[o] = scale_index(s, table, n_scales);
        if (likelihoods) likelihoods[o] = lik;
    }
    const double r = block_sum(bits, sm);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = r;
}

extern "C" int vc_gc_forward(vc_stream s, vc_view y, vc_view scales, vc_view means, const float *in_gain,
                             const float *out_gain, vc_view y_hat, double *bits_partial, int bits_slots,
                             const float *sym_src_p, int32_t *symbols, int32_t *indexes, const float *scale_table,
                             int n_scales, float *likelihoods)
{
    if (!y.p || !scales.p || !means.p) return VC_EINVAL;
    if (bits_partial && bits_slots != ENT_SLOTS) return VC_EINVAL;
    if (indexes && (!scale_table || n_scales < 2)) return VC_EINVAL;
    if (sym_src_p && !symbols) return VC_EINVAL;
    hipLaunchKernelGGL(k_gc_forward, dim3(ENT_SLOTS), dim3(ENT_BLOCK), 0, as_stream(s), y, scales, means, in_gain, out_gain,
                       y_hat, bits_partial, sym_src_p, symbols, indexes, scale_table, n_scales, likelihoods);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_gc_indexes(vc_view sc, const float *__restrict__ table, int n_scales, int32_t *__restrict__ indexes)
{
    const long long total = (long long)sc.n * sc.h * sc.w * sc.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % sc.c);
        long long t = i / sc.c;
        const int x = (int)(t % sc.w); t /= sc.w;
        const int y = (int)(t % sc.h);
        const int n = (int)(t / sc.h);
        const float s = fmaxf(sc.p[view_off(sc, n, y, x) + c], 0.11f);
        indexes[(((long long)n * sc.c + c) * sc.h + y) * sc.w + x] = scale_index(s, table, n_scales);
    }
}

extern "C" int vc_gc_indexes(vc_stream s, vc_view scales, const float *scale_table, int n_scales, int32_t *indexes)
{
    if (!scales.p || !scale_table || !indexes || n_scales < 2) return VC_EINVAL;
    const long long total = (long long)scales.n * scales.h * scales.w * scales.c;
    hipLaunchKernelGGL(k_gc_indexes, dim3(ew_grid(total, ENT_BLOCK)), dim3(ENT_BLOCK), 0, as_stream(s), scales, scale_table, n_scales, indexes);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_gc_dequant(const int32_t *__restrict__ symbols, vc_view mu, const float *__restrict__ out_gain, vc_view yh)
{
    const long long total = (long long)yh.n * yh.h * yh.w * yh.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % yh.c);
        long long t = i / yh.c;
        const int x = (int)(t % yh.w); t /= yh.w;
        const int y = (int)(t % yh.h);
        const int n = (int)(t / yh.h);
        const float v = (float)symbols[(((long long)n * yh.c + c) * yh.h + y) * yh.w + x] + mu.p[view_off(mu, n, y, x) + c];
        yh.p[view_off(yh, n, y, x) + c] = out_gain ? v * out_gain[c] : v;
    }
}

extern "C" int vc_gc_dequant(vc_stream s, const int32_t *symbols, vc_view means, const float *out_gain, vc_view y_hat)
{
    if (!symbols || !means.p || !y_hat.p) return VC_EINVAL;
    const long long total = (long long)y_hat.n * y_hat.h * y_hat.w * y_hat.c;
    hipLaunchKernelGGL(k_gc_dequant, dim3(ew_grid(total, ENT_BLOCK)), dim3(ENT_BLOCK), 0, as_stream(s), symbols, means, out_gain, y_hat);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_bits_reduce(const double *__restrict__ partial, int slots, double *__restrict__ out)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double v = 0.0;
    for (int j = threadIdx.x; j < slots; j += blockDim.x) v += partial[(long long)blockIdx.x * slots + j];
    const double r = block_sum(v, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

extern "C" int vc_bits_reduce(vc_stream s, const double *partial, int slots, int count, double *out)
{
    if (!partial || !out || slots < 1 || count < 1) return VC_EINVAL;
    hipLaunchKernelGGL(k_bits_reduce, dim3(count), dim3(ENT_BLOCK), 0, as_stream(s), partial, slots, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// ICIP2024 flow-resolution search on the device (opt_helpers.py:41-51)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ENT_BLOCK) k_sse_clamp01(vc_view pred, vc_view cur, double *__restrict__ partial)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double acc = 0.0;
    const long long total = (long long)pred.n * pred.h * pred.w * pred.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % pred.c);
        long long t = i / pred.c;
        const int x = (int)(t % pred.w); t /= pred.w;
        const int y = (int)(t % pred.h);
        const int n = (int)(t / pred.h);
        const float p = fminf(fmaxf(pred.p[view_off(pred, n, y, x) + c], 0.0f), 1.0f);
        const float d = p - cur.p[view_off(cur, n, y, x) + c];
        acc += (double)(d * d);
    }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

extern "C" int vc_sse_clamp01(vc_stream s, vc_view pred, vc_view cur, double *sse_partial, int slots)
{
    if (!pred.p || !cur.p || !sse_partial || slots != ENT_SLOTS) return VC_EINVAL;
    if (pred.n != cur.n || pred.h != cur.h || pred.w != cur.w || pred.c != cur.c) return VC_EINVAL;
    hipLaunchKernelGGL(k_sse_clamp01, dim3(ENT_SLOTS), dim3(ENT_BLOCK), 0, as_stream(s), pred, cur, sse_partial);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

struct FlowCandidates {
    vc_view v[8];
};

__global__ void k_select_flow(const double *__restrict__ sse, int count, double n_elems, FlowCandidates cands, vc_view out,
                              int32_t *__restrict__ choice)
{
    int pick = 0;
    float best = 0.0f;
    for (int i = 0; i < count; ++i) {
        const float mse = (float)(sse[i] / n_elems);
        const float psnr = 10.0f * log10f(1.0f / mse);
        if (psnr > best) {
            best = psnr;
            pick = i;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && choice) *choice = pick;
    const vc_view src = cands.v[pick];
    const long long total = (long long)out.n * out.h * out.w * out.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % out.c);
        long long t = i / out.c;
        const int x = (int)(t % out.w); t /= out.w;
        const int y = (int)(t % out.h);
        const int n = (int)(t / out.h);
        out.p[view_off(out, n, y, x) + c] = src.p[view_off(src, n, y, x) + c];
    }
}

extern "C" int vc_select_flow(vc_stream s, const double *sse, int count, double n_elems, const vc_view *candidates,
                              vc_view out, int32_t *choice)
{
    if (!sse || !candidates || !out.p || count < 1 || count > 8 || !(n_elems > 0.0)) return VC_EINVAL;
    FlowCandidates c = {};
    for (int i = 0; i < count; ++i) {
        const vc_view &v = candidates[i];
        if (!v.p || v.n != out.n || v.h != out.h || v.w != out.w || v.c != out.c) return VC_EINVAL;
        c.v[i] = v;
    }
    const long long total = (long long)out.n * out.h * out.w * out.c;
    hipLaunchKernelGGL(k_select_flow, dim3(ew_grid(total, ENT_BLOCK)), dim3(ENT_BLOCK), 0, as_stream(s), sse, count, n_elems,
                       c, out, choice);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// PSNR of the evaluation loops (LHBDC/test/testing.py:176-182, test/utils.py:32-51): both frames are clamped to
// [0,1], scaled to 0..255 and rounded (half to even, like torch.round), the squared error is averaged over the
// un-padded [:h, :w] crop of all channels in double precision, PSNR = 10 log10(255^2 / mse).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ENT_BLOCK) k_sse_uint8(const float *__restrict__ a, const float *__restrict__ b, int C, int H, int W,
                                                         int h, int w, double *__restrict__ partial)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double acc = 0.0;
    const long long total = (long long)C * h * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        long long t = i / w;
        const int y = (int)(t % h);
        const int c = (int)(t / h);
        const long long o = ((long long)c * H + y) * W + x;
        const float qa = rintf(fminf(fmaxf(a[o], 0.0f), 1.0f) * 255.0f);
        const float qb = rintf(fminf(fmaxf(b[o], 0.0f), 1.0f) * 255.0f);
        const double d = (double)qa - (double)qb;
        acc += d * d;
    }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void k_psnr_from_sse(const double *__restrict__ partial, int slots, double count, double *__restrict__ out)
{
    __shared__ double sm[ENT_BLOCK / 64];
    double v = 0.0;
    for (int j = threadIdx.x; j < slots; j += blockDim.x) v += partial[j];
    const double r = block_sum(v, sm);
    if (threadIdx.x == 0) *out = 10.0 * log10(255.0 * 255.0 / (r / count));
}

extern "C" int vc_psnr_uint8(vc_stream s, const float *a_chw, const float *b_chw, int channels, int H, int W, int h, int w,
                             double *scratch, int slots, double *psnr_out)
{
    if (!a_chw || !b_chw || !scratch || !psnr_out || slots != ENT_SLOTS) return VC_EINVAL;
    if (channels < 1 || h < 1 || w < 1 || h > H || w > W) return VC_EINVAL;
    hipLaunchKernelGGL(k_sse_uint8, dim3(ENT_SLOTS), dim3(ENT_BLOCK), 0, as_stream(s), a_chw, b_chw, channels, H, W, h, w, scratch);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_psnr_from_sse, dim3(1), dim3(ENT_BLOCK), 0, as_stream(s), scratch, slots, (double)channels * h * w, psnr_out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Scale refinement (round 5; DESIGN section 8 (5) of round 4): the scale-table index of a latent is the fragile integer of the
// CompressAI format -- a scale within fp32 summation noise of one of the 64 table entries lands in the neighbouring bin on another
// platform.  For exactly those elements (relative distance to a table entry <= rel_eps) the last hyper-synthesis layer
// (LHBDC/model/layers.py:82-91: conv3x3(3N/2 -> 2N), first N output channels = the scales) is recomputed in fp64 -- 9 * cin exact
// products, double accumulation, ONE rounding to fp32: this side's value is then the correctly rounded one, independent of any
// summation order.  ~1e-4 of the elements: a few hundred dot products per frame.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_refine_scales(vc_view sc, vc_view in, const float *__restrict__ w, const float *__restrict__ bias,
                                                       const float *__restrict__ table, int n_scales, float rel_eps, int *__restrict__ counter)
{
    const long long total = (long long)sc.n * sc.h * sc.w * sc.c;
    const int lane = threadIdx.x & 63;
    const int cin = in.c, kk = 9 * cin;
    const float lt0 = logf(table[0]), step = (logf(table[n_scales - 1]) - lt0) / (float)(n_scales - 1);
    const long long nwave = (long long)gridDim.x * (blockDim.x >> 6), wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long base = wave0 * 64; base < total; base += nwave * 64) {
        const long long i = base + lane;
        bool near = false;
        if (i < total) {
            const int c = (int)(i % sc.c);
            long long t = i / sc.c;
            const int x = (int)(t % sc.w); t /= sc.w;
            const int y = (int)(t % sc.h);
            const int n = (int)(t / sc.h);
            const float s = sc.p[view_off(sc, n, y, x) + c];
            if (s > table[0] * (1.0f - rel_eps)) {
                int k = (int)floorf((logf(s) - lt0) / step + 0.5f);
                k = min(max(k, 0), n_scales - 1);
                // (the table is log-spaced only up to fp32 rounding of exp(): test the neighbours too)
                for (int j = max(k - 1, 0); j <= min(k + 1, n_scales - 1); ++j) near = near || fabsf(s - table[j]) <= rel_eps * table[j];
            }
        }
        unsigned long long mask = __ballot(near);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const long long e = base + src;                    // wave-uniform
            const int c = (int)(e % sc.c);
            long long t = e / sc.c;
            const int x = (int)(t % sc.w); t /= sc.w;
            const int y = (int)(t % sc.h);
            const int n = (int)(t / sc.h);
            double acc = 0.0;
            for (int idx = lane; idx < kk; idx += 64) {
                const int tap = idx / cin, ci = idx - tap * cin;
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if (iy >= 0 && iy < in.h && ix >= 0 && ix < in.w)
                    acc = fma((double)in.p[view_off(in, n, iy, ix) + ci], (double)w[((long long)c * cin + ci) * 9 + tap], acc);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                sc.p[view_off(sc, n, y, x) + c] = (float)(acc + (double)(bias ? bias[c] : 0.0f));
                if (counter) atomicAdd(counter, 1);
            }
        }
    }
}

extern "C" int vc_refine_scales(vc_stream s, vc_view scales, vc_view in, const float *w_oihw, const float *bias, const float *scale_table,
                                int n_scales, float rel_eps, int *counter)
{
    if (!scales.p || !in.p || !w_oihw || !scale_table || n_scales < 2 || scales.n != in.n || scales.h != in.h || scales.w != in.w ||
        !(rel_eps > 0.0f) || rel_eps > 1e-2f)
        return VC_EINVAL;
    const long long total = (long long)scales.n * scales.h * scales.w * scales.c;
    if (total <= 0) return VC_OK;
    hipLaunchKernelGGL(k_refine_scales, dim3(ew_grid((total + 63) / 64 * 64, 256)), dim3(256), 0, as_stream(s), scales, in, w_oihw, bias, scale_table,
                       n_scales, rel_eps, counter);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Symbol refinement (round 6): the twin of vc_refine_scales for the coded integers themselves.  A symbol is round(y - mu)
// (GaussianConditional.compress, LHBDC/model/layers.py:93-104,168-179) or round(z * gain - median) (EntropyBottleneck.compress);
// a difference within fp32 summation noise of a half-integer rounds the other way on another platform.  For exactly those
// elements (| frac(d) - 1/2 | <= eps; ~2 eps of them) the producing layers -- the analysis transform's last convolution for y / z
// and the hyper-synthesis transform's last convolution for mu -- are recomputed from their inputs in fp64 (k * k * cin exact
// products, double accumulation, one rounding to fp32 each) and the symbol is taken from those values.  y, z and mu in memory are
// NOT changed (the decoder cannot know which elements were refined: its mu must stay the encoder's mu); only `symbols` and, when
// given, the de-quantised tensor `hat` ( = (symbol + mu) * out_gain: the closed-loop encoder's reconstruction) are rewritten.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_conv_f64(const vc_refine_layer &L, int n, int oy, int ox, int oc, int lane)
{
    const int cin = L.in.c, k = L.k, kk = k * k * cin, pad = k >> 1;
    double acc = 0.0;
    for (int idx = lane; idx < kk; idx += 64) {
        const int tap = idx / cin, ci = idx - tap * cin;
        const int iy = oy * L.stride + tap / k - pad, ix = ox * L.stride + tap % k - pad;
        if (iy >= 0 && iy < L.in.h && ix >= 0 && ix < L.in.w)
            acc = fma((double)L.in.p[view_off(L.in, n, iy, ix) + ci], (double)L.w_oihw[((long long)(L.c0 + oc) * cin + ci) * (k * k) + tap], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return (float)(acc + (double)(L.bias ? L.bias[L.c0 + oc] : 0.0f));
}

// MODE 0: Gaussian conditional (centre = mu, a tensor produced by layer `lb`); MODE 1: factorised prior (centre = the channel's median)
template <int MODE>
__global__ void __launch_bounds__(256) k_refine_symbols(vc_view v, vc_refine_layer la, vc_view mu, vc_refine_layer lb,
                                                        const float *__restrict__ eb_params, const float *__restrict__ in_gain, float eps,
                                                        int32_t *__restrict__ symbols, vc_view hat, const float *__restrict__ out_gain,
                                                        int *__restrict__ counter)
{
    const long long total = (long long)v.n * v.h * v.w * v.c;
    const int lane = threadIdx.x & 63;
    const long long nwave = (long long)gridDim.x * (blockDim.x >> 6), wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long long base = wave0 * 64; base < total; base += nwave * 64) {
        const long long i = base + lane;
        bool near = false;
        if (i < total) {
            const int c = (int)(i % v.c);
            long long t = i / v.c;
            const int x = (int)(t % v.w); t /= v.w;
            const int y = (int)(t % v.h);
            const int n = (int)(t / v.h);
            float a = v.p[view_off(v, n, y, x) + c];
            if (in_gain) a *= in_gain[c];
            const float m = MODE == 0 ? mu.p[view_off(mu, n, y, x) + c] : eb_params[(long long)c * VC_EB_PARAMS_PER_CHANNEL + 58];
            const float d = a - m;
            near = fabsf((d - floorf(d)) - 0.5f) <= eps;
        }
        unsigned long long mask = __ballot(near);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const long long e = base + src;                    // wave-uniform
            const int c = (int)(e % v.c);
            long long t = e / v.c;
            const int x = (int)(t % v.w); t /= v.w;
            const int y = (int)(t % v.h);
            const int n = (int)(t / v.h);
            float a = wave_conv_f64(la, n, y, x, c, lane);
            if (in_gain) a *= in_gain[c];
            const float m_mem = MODE == 0 ? mu.p[view_off(mu, n, y, x) + c] : eb_params[(long long)c * VC_EB_PARAMS_PER_CHANNEL + 58];
            const float m = MODE == 0 ? wave_conv_f64(lb, n, y, x, c, lane) : m_mem;
            if (lane == 0) {
                const float q = rintf(a - m);
                symbols[(((long long)n * v.c + c) * v.h + y) * v.w + x] = (int32_t)q;
                if (hat.p) {
                    const float r = q + m_mem;              // the decoder's own centre
                    hat.p[view_off(hat, n, y, x) + c] = out_gain ? r * out_gain[c] : r;
                }
                if (counter) atomicAdd(counter, 1);
            }
        }
    }
}

static bool refine_layer_ok(const vc_refine_layer &L, const vc_view &out, int need_c)
{
    if (!L.in.p || !L.w_oihw || L.k < 1 || !(L.k & 1) || L.k > 7 || L.stride < 1 || L.stride > 2 || L.c0 < 0 || need_c < 1) return false;
    return L.in.n == out.n && (L.in.h - 1) / L.stride + 1 == out.h && (L.in.w - 1) / L.stride + 1 == out.w;
}

extern "C" int vc_refine_y_symbols(vc_stream s, vc_view y, vc_refine_layer y_layer, vc_view means, vc_refine_layer mu_layer, float eps,
                                   int32_t *symbols, vc_view y_hat, const float *out_gain, int *counter)
{
    if (!y.p || !means.p || !symbols || !(eps > 0.0f) || eps > 1e-2f) return VC_EINVAL;
    if (means.n != y.n || means.h != y.h || means.w != y.w || means.c != y.c) return VC_EINVAL;
    if (y_hat.p && (y_hat.n != y.n || y_hat.h != y.h || y_hat.w != y.w || y_hat.c != y.c)) return VC_EINVAL;
    if (!refine_layer_ok(y_layer, y, y.c) || !refine_layer_ok(mu_layer, y, y.c)) return VC_EINVAL;
    const long long total = (long long)y.n * y.h * y.w * y.c;
    if (total <= 0) return VC_OK;
    hipLaunchKernelGGL(k_refine_symbols<0>, dim3(ew_grid((total + 63) / 64 * 64, 256)), dim3(256), 0, as_stream(s), y, y_layer, means, mu_layer,
                       (const float *)nullptr, (const float *)nullptr, eps, symbols, y_hat, out_gain, counter);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_refine_z_symbols(vc_stream s, vc_view z, vc_refine_layer z_layer, const float *eb_params, const float *in_gain, float eps,
                                   int32_t *symbols, vc_view z_hat, const float *out_gain, int *counter)
{
    if (!z.p || !eb_params || !symbols || !(eps > 0.0f) || eps > 1e-2f) return VC_EINVAL;
    if (z_hat.p && (z_hat.n != z.n || z_hat.h != z.h || z_hat.w != z.w || z_hat.c != z.c)) return VC_EINVAL;
    if (!refine_layer_ok(z_layer, z, z.c)) return VC_EINVAL;
    const long long total = (long long)z.n * z.h * z.w * z.c;
    if (total <= 0) return VC_OK;
    vc_view none = {};
    vc_refine_layer nol = {};
    hipLaunchKernelGGL(k_refine_symbols<1>, dim3(ew_grid((total + 63) / 64 * 64, 256)), dim3(256), 0, as_stream(s), z, z_layer, none, nol,
                       eb_params, in_gain, eps, symbols, z_hat, out_gain, counter);
    VC_LAUNCH_CHECK();
    return VC_OK;
}
