// What does a streaming kernel get out of HBM on this part?  The ceiling for the 1x1 layers (32 FLOP/B and below).
//   mode 0: copy            (read X, write X)
//   mode 1: read only       (sum into a register, one store per thread at the end)
//   mode 2: write only
//   mode 3: read 2X write X (a 1x1 layer with a residual)
//   mode 4: copy through LDS-DMA: global_load_lds_dwordx4 into a per-wave ring, ds_read_b128, global_store (what the
//           streaming 1x1 kernel does with its activations)
// U = independent 16-byte accesses in flight per lane and iteration; NT = non-temporal loads/stores.
// Build: hipcc --offload-arch=gfx950 -O3 stream_rate.hip -o stream_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f32x4 ld(const f32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f32x4 *p, f32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int MODE, int U, bool NT> __global__ void __launch_bounds__(256)
stream(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *__restrict__ o, size_t n4, float *sink)
{
    const size_t chunk = 256 * U;
    const size_t nchunks = n4 / chunk;
    f32x4 s = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        f32x4 v[U], w[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + base + 256 * u);
        }
        if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = ld<NT>(b + base + 256 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 1) s += v[u];
            else if (MODE == 2) st<NT>(o + base + 256 * u, s);
            else if (MODE == 3) st<NT>(o + base + 256 * u, v[u] + w[u]);
            else st<NT>(o + base + 256 * u, v[u]);
        }
    }
    if (MODE == 1) sink[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// LDS-DMA copy: every wave owns a ring of D slots of 1 KiB x U; slot i+D-1 is requested before slot i is consumed.
template <int U, int D> __global__ void __launch_bounds__(256)
stream_dma(const f32x4 *__restrict__ a, f32x4 *__restrict__ o, size_t n4)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t chunk = 64 * U;                                  // 16-byte elements per wave and slot
    const size_t nchunks = n4 / chunk;
    const size_t first = (size_t)blockIdx.x * 4 + wave, step = (size_t)gridDim.x * 4;
    const unsigned ring = wave * (D * U * 1024);
    auto request = [&](size_t c, int slot) {
        const f32x4 *src = a + c * chunk + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned dst = ring + (slot * U + u) * 1024;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src + 64 * u) : "memory");
        }
    };
    size_t c = first;
    int issued = 0;
    for (size_t cc = c; issued < D - 1 && cc < nchunks; cc += step, ++issued) request(cc, issued);
    int slot = 0;
    size_t ahead = c + (size_t)(D - 1) * step;
    for (; c < nchunks; c += step, ahead += step) {
        if (ahead < nchunks) {
            request(ahead, (slot + D - 1) % D);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 2 * U) : "memory");   // D-1 younger requests and the stores between them may stay
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + ring + (slot * U + u) * 1024 + lane * 16);
            o[c * chunk + lane + 64 * u] = v;
        }
        slot = (slot + 1) % D;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> float timed(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t sizes[2] = {(size_t)544 * 960 * 128 * 4, (size_t)1088 * 1920 * 128 * 4};   // 267 MB, 1069 MB
    f32x4 *a, *b, *o;
    float *sink;
    CK(hipMalloc(&a, sizes[1]));
    CK(hipMalloc(&b, sizes[1]));
    CK(hipMalloc(&o, sizes[1]));
    CK(hipMalloc(&sink, 8192 * 256 * 4));
    CK(hipMemset(a, 0, sizes[1]));
    CK(hipMemset(b, 0, sizes[1]));
    CK(hipMemset(o, 0, sizes[1]));
    const char *names[5] = {"copy", "read", "write", "read2+write", "dma copy"};
    for (int si = 0; si < 2; ++si) {
        const size_t bytes = sizes[si], n4 = bytes / 16;
        printf("== X = %.0f MB ==\n", bytes / 1e6);
        for (int grid : {256, 512, 1024, 2048, 4096}) {
#define RUN(MODE, U, NT, MULT)                                                                                          \
    {                                                                                                                   \
        const float ms = timed([&] { hipLaunchKernelGGL((stream<MODE, U, NT>), dim3(grid), dim3(256), 0, 0, a, b, o, n4, sink); }, 10); \
        printf("%-12s U=%d nt=%d grid=%4d : %.3f ms  %.0f GB/s\n", names[MODE], U, NT, grid, ms, MULT * bytes / ms / 1e6);  \
    }
            RUN(0, 1, false, 2.0) RUN(0, 4, false, 2.0) RUN(0, 8, false, 2.0) RUN(0, 4, true, 2.0) RUN(0, 8, true, 2.0)
            RUN(1, 4, false, 1.0) RUN(1, 8, false, 1.0) RUN(1, 8, true, 1.0)
            RUN(2, 4, false, 1.0) RUN(2, 4, true, 1.0)
            RUN(3, 4, false, 3.0) RUN(3, 4, true, 3.0)
#define RUND(U, D)                                                                                                      \
    {                                                                                                                   \
        const float ms = timed([&] { hipLaunchKernelGGL((stream_dma<U, D>), dim3(grid), dim3(256), 4 * D * U * 1024, 0, a, o, n4); }, 10); \
        printf("%-12s U=%d D=%d grid=%4d : %.3f ms  %.0f GB/s\n", names[4], U, D, grid, ms, 2.0 * bytes / ms / 1e6);        \
    }
            if (grid <= 1024) { RUND(4, 2) RUND(4, 3) RUND(8, 2) }
        }
        // one chunk per workgroup (no persistence): the hardware dispatcher does the striding
        {
            const unsigned g = (unsigned)(n4 / (256 * 4));
            const float ms = timed([&] { hipLaunchKernelGGL((stream<0, 4, false>), dim3(g), dim3(256), 0, 0, a, b, o, n4, sink); }, 10);
            printf("copy U=4 one chunk per workgroup (grid %u): %.3f ms  %.0f GB/s\n", g, ms, 2.0 * bytes / ms / 1e6);
        }
    }
    return 0;
}
