// What does one LDS-DMA instruction cost the wave that issues it, and the CU?  (gfx950; L2-resident sources)
// One 512-thread workgroup per CU; every wave issues NI instructions back to back (cycles per instruction from s_memtime),
// for different instruction forms / source patterns / numbers of issuing waves, with and without MFMAs on the partner waves.
//   form 0: global_load_lds_dwordx4, 64-bit per-lane address, 1 KiB contiguous per wave-instruction (a weight fragment)
//   form 1: same, 16 pixels x 64 B (a 32-channel chunk image piece: 16 half lines)
//   form 2: same, 8 pixels x 128 B (full lines, 256-B pixel stride)
//   form 3: global_load_lds_dwordx4 with SGPR base + 32-bit lane offset, contiguous
//   form 4: plain global_load_dwordx4 to VGPRs, contiguous (for comparison)
//   form 5: global_load_lds_dword (4 B per lane), contiguous 256 B
//   form 6: buffer_load_dwordx4 ... offen lds (SGPR resource descriptor + 32-bit lane offset), 16 px x 64 B
//   form 7: global_load_lds_dwordx4 with SGPR base + 32-bit lane offset, 16 px x 64 B (form 1 with the cheaper address)
// Build: hipcc --offload-arch=gfx950 -O3 glds_rate.hip -o glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int NI> __global__ void __launch_bounds__(512, 2)
rate(const unsigned char *src, unsigned long long *out, int issuing_waves, int mfma_waves, int reps, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char *base = src + (size_t)blockIdx.x * (1 << 20);
    f32x16 acc = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
    unsigned long long total = 0;
    f32x4 vsum = {0, 0, 0, 0};
    i32x4 rsrc;          // raw buffer over this CU's 1 MiB window (stride 0: out-of-range offsets read zero)
    {
        const unsigned long long a = (unsigned long long)base;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff));
        rsrc[2] = __builtin_amdgcn_readfirstlane(1 << 20);
        rsrc[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    }
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        if (wave < issuing_waves) {
            unsigned long long t0, t1;
            f32x4 vv[NI];
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const unsigned ldsaddr = wave * 16384 + (i % 16) * 1024;
                const size_t blk = (size_t)((r * NI + i) * 8 + wave) & 1023;      // 1 KiB block index inside the CU's 1 MiB window
                if constexpr (FORM == 0) {
                    const unsigned char *p = base + blk * 1024 + lane * 16;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 1) {
                    const unsigned char *p = base + (blk & 511) * 2048 + (lane >> 2) * 256 + (lane & 3) * 16;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 2) {
                    const unsigned char *p = base + (blk & 511) * 2048 + (lane >> 3) * 256 + (lane & 7) * 16;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 3) {
                    const unsigned voff = (unsigned)(blk * 1024 + lane * 16);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 6) {
                    const unsigned voff = (unsigned)((blk & 511) * 2048 + (lane >> 2) * 256 + (lane & 3) * 16);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 7) {
                    const unsigned voff = (unsigned)((blk & 511) * 2048 + (lane >> 2) * 256 + (lane & 3) * 16);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory");
                } else if constexpr (FORM == 4) {
                    vv[i] = *reinterpret_cast<const f32x4 *>(base + blk * 1024 + lane * 16);
                } else {
                    const unsigned char *p = base + blk * 1024 + lane * 4;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p), "s"(ldsaddr) : "memory");
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            total += t1 - t0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (FORM == 4)
                for (int i = 0; i < NI; ++i) vsum += vv[i];
        } else if (wave >= 8 - mfma_waves) {
#pragma unroll
            for (int i = 0; i < 24; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    if (lane == 0 && wave < issuing_waves) atomicAdd(out, total);
    if (acc[0] == 12345.f || vsum[0] == 1.f) sink[0] = acc[1];
}

template <int FORM> void run(const unsigned char *src, unsigned long long *out, float *sink, const char *name)
{
    constexpr int NI = 16;
    const int reps = 50;
    hipFuncSetAttribute((const void *)rate<FORM, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    for (int mf : {0, 4})
        for (int iw : {1, 4, 8}) {
            if (iw + mf > 8) continue;
            hipMemset(out, 0, 8);
            hipLaunchKernelGGL((rate<FORM, NI>), dim3(256), dim3(512), 144 * 1024, 0, src, out, iw, mf, reps, sink);
            hipDeviceSynchronize();
            hipLaunchKernelGGL((rate<FORM, NI>), dim3(256), dim3(512), 144 * 1024, 0, src, out, iw, mf, reps, sink);
            hipDeviceSynchronize();
            unsigned long long t = 0;
            hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            const double per = (double)t / 2 / (256.0 * iw * reps * NI);
            printf("%-52s issuing waves %d, MFMA waves %d: %7.1f cycles per instruction per wave (= %6.1f cycles per instruction per CU)\n", name, iw,
                   mf, per, per / iw);
        }
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned char *src;
    unsigned long long *out;
    float *sink;
    hipMalloc(&src, 264u << 20);
    hipMemset(src, 1, 264u << 20);
    hipMalloc(&out, 8);
    hipMalloc(&sink, 64);
    run<0>(src, out, sink, "glds x4, 64-bit addr, 1 KiB contiguous");
    run<1>(src, out, sink, "glds x4, 64-bit addr, 16 px x 64 B");
    run<2>(src, out, sink, "glds x4, 64-bit addr, 8 px x 128 B");
    run<3>(src, out, sink, "glds x4, SGPR base + 32-bit offset, contiguous");
    run<4>(src, out, sink, "global_load_dwordx4 -> VGPR, contiguous");
    run<5>(src, out, sink, "glds x1 (4 B/lane), contiguous");
    run<6>(src, out, sink, "buffer_load x4 lds, descriptor + 32-bit offset, 16 px x 64 B");
    run<7>(src, out, sink, "glds x4, SGPR base + 32-bit offset, 16 px x 64 B");
    return 0;
}
