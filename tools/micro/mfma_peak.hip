// Micro-benchmark: what does the fp32 matrix pipe deliver with nothing else in the way?  Each wave runs back-to-back
// v_mfma_f32_32x32x2_f32 on NACC independent accumulators (operands in registers, no memory traffic), W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC> __global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(int blocks_per_cu, const char *label)
{
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f, 1e-3f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4 * NACC * (2.0 * 32 * 32 * 2);
    printf("%-44s %8.3f ms  %7.1f TFLOP/s (%.1f %% of 157.3)\n", label, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main()
{
    run<4>(1, "1 wave/SIMD, 4 independent accumulators");
    run<4>(2, "2 waves/SIMD, 4 independent accumulators");
    run<1>(1, "1 wave/SIMD, 1 accumulator (dependent chain)");
    run<2>(1, "1 wave/SIMD, 2 accumulators");
    run<8>(1, "1 wave/SIMD, 8 accumulators");
    return 0;
}
