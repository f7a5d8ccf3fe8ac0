// Micro-benchmark: the fp32 matrix pipe WITH the operand traffic of the convolution loop.  Per "k-step" one wave issues
// 16 x v_mfma_f32_32x32x2_f32 (4 accumulators x 4 k-sub-steps) fed by LDSR x ds_read_b128 (activations) and GLD x
// global_load_dwordx4 (weights), software-pipelined one step ahead like conv_mfma_kernel -- nothing else.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_feed.hip -o tools/micro/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDSR, int GLD> __global__ void __launch_bounds__(256) k(float *out, const float *w, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    const float *wl = w + lane * 4;
    // two register sets used alternately (ping-pong, the loop is unrolled by two): no register copies in the loop, like the
    // fully unrolled kernel row of conv_mfma_kernel
    f32x4 a[2][4], b[2];
    for (int t = 0; t < 4; ++t) a[0][t] = *reinterpret_cast<const f32x4 *>(&lds[(t * 1024 + lane * 20) & 16383]);
    b[0] = *reinterpret_cast<const f32x4 *>(wl);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cur = half, nxt = half ^ 1;
            if (LDSR)
                for (int t = 0; t < 4; ++t)
                    a[nxt][t] = *reinterpret_cast<const f32x4 *>(&lds[(((it + half) & 7) * 80 + t * 3040 + lane * 20) & 16380]);
            else
                for (int t = 0; t < 4; ++t) a[nxt][t] = a[cur][t];
            if (GLD) b[nxt] = *reinterpret_cast<const f32x4 *>(wl + (((it + half) & 63) << 8));
            else b[nxt] = b[cur];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][e], a[cur][t][e], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDSR, int GLD> void run(int blocks_per_cu, const char *label, const float *w)
{
    const int blocks = 256 * blocks_per_cu, iters = 8000;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<LDSR, GLD>), dim3(blocks), dim3(256), 65536, 0, out, w, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
    printf("%-58s %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)\n", label, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main()
{
    float *w;
    hipMalloc(&w, 64 * 256 * 4 + 4096);
    hipMemset(w, 0, 64 * 256 * 4 + 4096);
    run<0, 0>(1, "1 wave/SIMD, operands in registers", w);
    run<1, 0>(1, "1 wave/SIMD, 4 ds_read_b128 per 16 MFMAs", w);
    run<0, 1>(1, "1 wave/SIMD, 1 global_load_dwordx4 per 16 MFMAs", w);
    run<1, 1>(1, "1 wave/SIMD, 4 ds_read_b128 + 1 global load per 16 MFMAs", w);
    run<1, 1>(2, "2 waves/SIMD, 4 ds_read_b128 + 1 global load per 16 MFMAs", w);
    return 0;
}
