// Micro-benchmark (round 5, the go / no-go gate of the split-operand fp32 path): an fp32 contraction done on the bf16 matrix
// pipe.  x = hi + mid + lo with three bf16 pieces (8 + 8 + 8 significand bits, obtained by truncation: every piece and every
// remainder is exact), so a * b = the sum of the nine piece products, each of them exact in fp32 (16-bit significands); the
// accumulation is the MFMA's own fp32 accumulate.  Nine v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each, K = 2): 288 against 512 matrix cycles per 32 x 32 x 16 block -- if the clock holds.
//
// Part 1 (rate): the k-step with its LDS operand traffic and nothing else, as tools/micro/mfma_feed.hip does for the native
//   fp32 k-step: a wave owns 2 x 2 tiles of 32 x 32 (or 4 x 4 of 16 x 16); per 16 channels it reads (2 + 2) x 3 piece fragments
//   by ds_read_b128 one step ahead and issues 36 MFMAs.  RANDOM operands (the clock this part holds under bf16 MFMA load
//   depends on the data).  Reported: fp32-equivalent TFLOP/s (2 M N K / time), the in-kernel clock (s_memtime over
//   s_memrealtime), and the native fp32 k-step on the same tile arrangement beside it.
// Part 2 (numerics): C = A . B (32 x 32 x K) by the native fp32 MFMA chain, by the nine-product split in two issue orders
//   and by the six-product variant (no lo x lo, mid x lo, lo x mid), against an fp64 reference.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_split.hip -o tools/micro/mfma_split ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: native fp32, 32x32x2      (per step of 16 channels: 8 reads, 32 MFMAs of 64 cycles)
// MODE 1: split, nine products, 32x32x16 bf16   (12 reads, 36 MFMAs of 32 cycles)
// MODE 2: split, six products                   (12 reads, 24 MFMAs)
// MODE 3: split, nine products, 16x16x32 bf16, 4 x 4 tiles of 16 x 16, a step = 32 channels (24 reads, 144 MFMAs of 16 cycles)
// LDSR 0: operands stay in registers (the pipe alone)
template <int MODE, int LDSR> __global__ void __launch_bounds__(256) k_rate(float *out, unsigned long long *clk, const unsigned *seed_data, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed_data[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    }
    float s = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc[2][2];
        for (int t = 0; t < 2; ++t)
            for (int n = 0; n < 2; ++n)
                for (int j = 0; j < 16; ++j) acc[t][n][j] = 0.f;
        f32x4 a[2][2][2], b[2][2][2];       // [set][ks][tile]
        for (int ks = 0; ks < 2; ++ks)
            for (int t = 0; t < 2; ++t) {
                a[0][ks][t] = *reinterpret_cast<const f32x4 *>(&lds[((ks * 2 + t) * 256 + lane * 4) & 16380]);
                b[0][ks][t] = *reinterpret_cast<const f32x4 *>(&lds[(2048 + (ks * 2 + t) * 256 + lane * 4) & 16380]);
            }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cur = half, nxt = half ^ 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (LDSR) {
                            a[nxt][ks][t] = *reinterpret_cast<const f32x4 *>(&lds[(((it + half) & 7) * 1024 + (ks * 2 + t) * 256 + lane * 4) & 16380]);
                            b[nxt][ks][t] = *reinterpret_cast<const f32x4 *>(&lds[(8192 + ((it + half) & 7) * 1024 + (ks * 2 + t) * 256 + lane * 4) & 16380]);
                        } else {
                            a[nxt][ks][t] = a[cur][ks][t];
                            b[nxt][ks][t] = b[cur][ks][t];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int n = 0; n < 2; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][ks][n][e], a[cur][ks][t][e], acc[t][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int t = 0; t < 2; ++t)
            for (int n = 0; n < 2; ++n)
                for (int j = 0; j < 16; ++j) s += acc[t][n][j];
    } else if constexpr (MODE == 1 || MODE == 2) {
        f32x16 acc[2][2];
        for (int t = 0; t < 2; ++t)
            for (int n = 0; n < 2; ++n)
                for (int j = 0; j < 16; ++j) acc[t][n][j] = 0.f;
        f32x4 a[2][3][2], b[2][3][2];       // [set][piece][tile]
        for (int pc = 0; pc < 3; ++pc)
            for (int t = 0; t < 2; ++t) {
                a[0][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[((pc * 2 + t) * 256 + lane * 4) & 16380]);
                b[0][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(8192 + (pc * 2 + t) * 256 + lane * 4) & 16380]);
            }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cur = half, nxt = half ^ 1;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (LDSR) {
                            // piece planes: hi at 0, mid at 2048, lo at 4096 dwords (A), + 8192 (B); the data generator gives the planes realistic magnitudes
                            a[nxt][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(pc * 2048 + ((it + half) & 3) * 512 + t * 256 + lane * 4) & 16380]);
                            b[nxt][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(8192 + pc * 2048 + ((it + half) & 3) * 512 + t * 256 + lane * 4) & 16380]);
                        } else {
                            a[nxt][pc][t] = a[cur][pc][t];
                            b[nxt][pc][t] = b[cur][pc][t];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
                // smallest products first: lo lo, (mid lo, lo mid), (hi lo, mid mid, lo hi), (hi mid, mid hi), hi hi
#pragma unroll
                for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa) {
                        const int pb = sum - pa;
                        if (pb < 0 || pb > 2) continue;
                        if (MODE == 2 && sum > 2) continue;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int n = 0; n < 2; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[cur][pb][n]),
                                                                                    __builtin_bit_cast(bf16x8, a[cur][pa][t]), acc[t][n], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int t = 0; t < 2; ++t)
            for (int n = 0; n < 2; ++n)
                for (int j = 0; j < 16; ++j) s += acc[t][n][j];
    } else {
        f32x4 acc[4][4];
        for (int t = 0; t < 4; ++t)
            for (int n = 0; n < 4; ++n)
                for (int j = 0; j < 4; ++j) acc[t][n][j] = 0.f;
        f32x4 a[2][3][4], b[2][3][4];
        for (int pc = 0; pc < 3; ++pc)
            for (int t = 0; t < 4; ++t) {
                a[0][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[((pc * 4 + t) * 256 + lane * 4) & 16380]);
                b[0][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(8192 + (pc * 4 + t) * 256 + lane * 4) & 16380]);
            }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cur = half, nxt = half ^ 1;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (LDSR) {
                            a[nxt][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(pc * 2048 + ((it + half) & 1) * 1024 + t * 256 + lane * 4) & 16380]);
                            b[nxt][pc][t] = *reinterpret_cast<const f32x4 *>(&lds[(8192 + pc * 2048 + ((it + half) & 1) * 1024 + t * 256 + lane * 4) & 16380]);
                        } else {
                            a[nxt][pc][t] = a[cur][pc][t];
                            b[nxt][pc][t] = b[cur][pc][t];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa) {
                        const int pb = sum - pa;
                        if (pb < 0 || pb > 2) continue;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[cur][pb][n]),
                                                                                    __builtin_bit_cast(bf16x8, a[cur][pa][t]), acc[t][n], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int t = 0; t < 4; ++t)
            for (int n = 0; n < 4; ++n)
                for (int j = 0; j < 4; ++j) s += acc[t][n][j];
    }
    if (lane == 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[2 * w] = t1 - t0;
        clk[2 * w + 1] = r1 - r0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short bf16_trunc(float x)
{
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h)
{
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void split3(float x, unsigned short p[3])
{
    p[0] = bf16_trunc(x);
    const float r1 = x - bf16_to_f(p[0]);
    p[1] = bf16_trunc(r1);
    const float r2 = r1 - bf16_to_f(p[1]);
    p[2] = bf16_trunc(r2);
    if (bf16_to_f(p[2]) != r2) {
        fprintf(stderr, "split not exact for %a\n", x);
        exit(1);
    }
}
static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0); }

template <int MODE, int LDSR> double run_rate(int blocks_per_cu, const char *label, const unsigned *d_data, double fp32_ref = 0.0)
{
    const int blocks = 256 * blocks_per_cu;
    int iters = MODE == 0 ? 6000 : (MODE == 3 ? 5000 : 10000);
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&clk, blocks * 4 * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    // ~1.5 s of back-to-back launches before the timed ones (the clock settles under load), then the median of 5
    std::vector<float> t;
    for (int rep = 0; rep < 45; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<MODE, LDSR>), dim3(blocks), dim3(256), 65536, 0, out, clk, d_data, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 40) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    ms = t[t.size() / 2];
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int w = 0; w < blocks * 4; ++w) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    // fp32-equivalent work: per step a wave contracts 64 px x 64 ch over 16 channels (32 in MODE 3)
    const double kk = MODE == 3 ? 32.0 : 16.0;
    const double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * kk;
    const double tf = flop / ms / 1e9;
    printf("%-78s %8.3f ms  %7.1f TFLOP/s fp32-equivalent  clock %.2f GHz", label, ms, tf, ghz[ghz.size() / 2]);
    if (fp32_ref > 0) printf("  = %.2fx the native k-step", tf / fp32_ref);
    printf("\n");
    hipFree(out);
    hipFree(clk);
    return tf;
}

// ---- numerics: one wave, C[32][32] = A[32][K] . B[K][32] ----
// VAR 0 native fp32 chain; 1 split9 small products first; 2 split9 large first; 3 split6 (small first)
template <int VAR> __global__ void __launch_bounds__(64) k_num(float *C, const float *A, const float *B, int K)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    if constexpr (VAR == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            bf16x8 ap[3], bp[3];
            for (int j = 0; j < 8; ++j) {
                float x = A[r * K + k0 + 8 * h + j], y = B[(k0 + 8 * h + j) * 32 + r];
                for (int pc = 0; pc < 3; ++pc) {
                    const unsigned ux = __builtin_bit_cast(unsigned, x) & 0xffff0000u, uy = __builtin_bit_cast(unsigned, y) & 0xffff0000u;
                    ap[pc][j] = (short)(ux >> 16);
                    bp[pc][j] = (short)(uy >> 16);
                    x -= __builtin_bit_cast(float, ux);
                    y -= __builtin_bit_cast(float, uy);
                }
            }
            if (VAR == 2) {
                for (int sum = 0; sum <= 4; ++sum)
                    for (int pa = 0; pa <= 2; ++pa) {
                        const int pb = sum - pa;
                        if (pb < 0 || pb > 2) continue;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa], bp[pb], acc, 0, 0, 0);
                    }
            } else {
                for (int sum = 4; sum >= 0; --sum)
                    for (int pa = 2; pa >= 0; --pa) {
                        const int pb = sum - pa;
                        if (pb < 0 || pb > 2) continue;
                        if (VAR == 3 && sum > 2) continue;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa], bp[pb], acc, 0, 0, 0);
                    }
            }
        }
    }
    for (int j = 0; j < 16; ++j) C[((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = acc[j];
}

template <int VAR> void run_num(const char *label, int K, const std::vector<float> &A, const std::vector<float> &B, const float *dA, const float *dB)
{
    float *dC;
    hipMalloc(&dC, 32 * 32 * 4);
    hipLaunchKernelGGL(k_num<VAR>, dim3(1), dim3(64), 0, 0, dC, dA, dB, K);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double worst = 0, sq = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
                ref += (double)A[i * K + k] * (double)B[k * 32 + j];
                mag += fabs((double)A[i * K + k] * (double)B[k * 32 + j]);
            }
            const double e = fabs((double)C[i * 32 + j] - ref) / mag;
            worst = e > worst ? e : worst;
            sq += e * e;
        }
    printf("  K = %5d  %-46s max |err| / sum|a b| = %.3e   rms = %.3e\n", K, label, worst, sqrt(sq / 1024));
    hipFree(dC);
}

int main()
{
    srand(12345);
    // LDS image for the rate kernels: planes of bf16 pieces of random floats (pairs packed per dword), A at dword 0, B at 8192;
    // within each: hi plane at 0, mid at 2048, lo at 4096, and a fourth plane of raw fp32 for MODE 0's slack (MODE 0 reads the
    // words as floats: random fp32 bit patterns with sane exponents)
    std::vector<unsigned> img(16384);
    for (int side = 0; side < 2; ++side)
        for (int i = 0; i < 2048; ++i) {
            unsigned short p0[3], p1[3];
            split3(frand(), p0);
            split3(frand(), p1);
            for (int pc = 0; pc < 3; ++pc) img[side * 8192 + pc * 2048 + i] = (unsigned)p0[pc] | ((unsigned)p1[pc] << 16);
            float f = frand();
            memcpy(&img[side * 8192 + 6144 + i], &f, 4);
        }
    std::vector<unsigned> img32(16384);
    for (int i = 0; i < 16384; ++i) {
        float f = frand();
        memcpy(&img32[i], &f, 4);
    }
    unsigned *d_img, *d_img32;
    hipMalloc(&d_img, 65536);
    hipMalloc(&d_img32, 65536);
    hipMemcpy(d_img, img.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(d_img32, img32.data(), 65536, hipMemcpyHostToDevice);

    printf("== part 1: rate (random operands; fp32-equivalent FLOP = 2 M N K of the fp32 contraction) ==\n");
    const double n1 = run_rate<0, 1>(1, "native fp32 32x32x2, 1 wave/SIMD, 8 ds_read_b128 per 32 MFMAs", d_img32);
    const double n2 = run_rate<0, 1>(2, "native fp32 32x32x2, 2 waves/SIMD", d_img32);
    run_rate<1, 0>(1, "split 9 x 32x32x16 bf16, 1 wave/SIMD, operands in registers", d_img, n1);
    run_rate<1, 1>(1, "split 9 x 32x32x16 bf16, 1 wave/SIMD, 12 ds_read_b128 per 36 MFMAs", d_img, n1);
    run_rate<1, 1>(2, "split 9 x 32x32x16 bf16, 2 waves/SIMD", d_img, n2);
    run_rate<3, 0>(1, "split 9 x 16x16x32 bf16, 1 wave/SIMD, operands in registers", d_img, n1);
    run_rate<3, 1>(1, "split 9 x 16x16x32 bf16, 1 wave/SIMD, 24 ds_read_b128 per 144 MFMAs", d_img, n1);
    run_rate<2, 1>(1, "split 6 x 32x32x16 bf16 (not exact; reported only), 1 wave/SIMD", d_img, n1);
    run_rate<2, 1>(2, "split 6 x 32x32x16 bf16 (not exact; reported only), 2 waves/SIMD", d_img, n2);

    printf("== part 2: numerics against fp64 (uniform [-1, 1) operands) ==\n");
    for (int K : {16, 288, 1568, 4096}) {
        std::vector<float> A(32 * K), B(K * 32);
        for (auto &v : A) v = frand();
        for (auto &v : B) v = frand();
        float *dA, *dB;
        hipMalloc(&dA, A.size() * 4);
        hipMalloc(&dB, B.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        run_num<0>("native v_mfma_f32_32x32x2_f32 chain", K, A, B, dA, dB);
        run_num<1>("split, nine products, small first", K, A, B, dA, dB);
        run_num<2>("split, nine products, large first", K, A, B, dA, dB);
        run_num<3>("split, six products", K, A, B, dA, dB);
        hipFree(dA);
        hipFree(dB);
    }
    return 0;
}
