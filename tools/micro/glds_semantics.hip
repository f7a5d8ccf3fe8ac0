// Probe of the LDS-DMA instruction forms the fp16 convolution pipeline relies on (gfx950):
//   (1) where global_load_lds_dwordx4 lands: M0 base + lane*16, and whether the instruction's immediate offset moves
//       the LDS destination as well as the global source;
//   (2) LDS destinations beyond 64 KiB;
//   (3) the SGPR-base + 32-bit VGPR-offset addressing form;
//   (4) counted vmcnt: data of an older DMA is visible after vmcnt(1) while a younger one is still in flight.
// Build: hipcc --offload-arch=gfx950 -O3 glds_semantics.hip -o glds_semantics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GLDS(src, ldsaddr, IMM)                                                                                       \
    do {                                                                                                              \
        unsigned keep__;                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:" #IMM \
                     "\n\ts_mov_b32 m0, %0"                                                                           \
                     : "=&s"(keep__) : "v"(src), "s"(ldsaddr) : "memory");                                            \
    } while (0)

__global__ void probe(const unsigned *in, unsigned *out, int mode, unsigned lds_base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x;
    // poison
    for (int i = tid; i < 40 * 1024; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned dst = lds_base + wave * 2048;     // bytes, wave-uniform; 2 KiB apart so shifts are visible
    const unsigned *src = in + tid * 4;              // lane-linear 16-byte items
    if (mode == 0) {
        GLDS(src, dst, 0);
    } else if (mode == 1) {
        GLDS(src, dst, 256);                         // immediate offset: source +256 B; LDS +256 B too?
    } else if (mode == 2) {
        const unsigned voff = tid * 16;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(in), "s"(dst) : "memory");
    } else if (mode == 3) {                          // per-lane source permutation: lane l reads item (l ^ 5)
        const unsigned *s2 = in + ((tid & ~63) + ((tid & 63) ^ 5)) * 4;
        GLDS(s2, dst, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 40 * 1024; i += blockDim.x) out[i] = lds[i];
}

// counted vmcnt: each wave issues DMA a (into slot A) then DMA b (slot B); waits vmcnt(1); barrier; reads slot A.
__global__ void counted(const unsigned *in, unsigned *out, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned *sa = in + ((size_t)(r * 2) * 256 + tid) * 4 + (size_t)blockIdx.x * 4096;
        const unsigned *sb = in + ((size_t)(r * 2 + 1) * 256 + tid) * 4 + (size_t)blockIdx.x * 4096;
        const unsigned da = wave * 1024, db = 4096 + wave * 1024;
        GLDS(sa, da, 0);
        GLDS(sb, db, 0);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // read another wave's part of slot A
        const int o = ((tid + 64) & 255) * 4;
        const unsigned v = lds[o];
        const unsigned expect = in[((size_t)(r * 2) * 256 + ((tid + 64) & 255)) * 4 + (size_t)blockIdx.x * 4096];
        bad += (v != expect);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * 256 + tid] = bad;
}

int main()
{
    const int N = 1 << 22;
    std::vector<unsigned> h(N);
    for (int i = 0; i < N; ++i) h[i] = i;
    unsigned *din, *dout;
    hipMalloc(&din, N * 4);
    hipMalloc(&dout, N * 4);
    hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned> o(40 * 1024);
    for (int mode = 0; mode < 4; ++mode)
        for (unsigned base : {0u, 100u * 1024u}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, din, dout, mode, base);
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d launch failed\n", mode); return 1; }
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            // report every run of non-poison dwords: [lds byte start, length) and the first value
            printf("mode %d base %6u:", mode, base);
            size_t i = 0; int runs = 0;
            while (i < o.size()) {
                if (o[i] == 0xdeadbeefu) { ++i; continue; }
                size_t j = i;
                while (j < o.size() && o[j] != 0xdeadbeefu) ++j;
                if (runs < 6) printf("  [%zu,+%zu) first=%u(item %u)", i * 4, (j - i) * 4, o[i], o[i] / 4);
                ++runs; i = j;
            }
            printf("  (%d runs)\n", runs);
            if (mode == 3) {
                printf("    mode 3 lane order at wave 0:");
                for (int l = 0; l < 8; ++l) printf(" %u", o[base / 4 + l * 4] / 4);
                printf("\n");
            }
        }
    hipLaunchKernelGGL(counted, dim3(512), dim3(256), 8192, 0, din, dout, 200);
    hipDeviceSynchronize();
    std::vector<unsigned> c(512 * 256);
    hipMemcpy(c.data(), dout, c.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0;
    for (auto v : c) bad += v;
    printf("counted vmcnt(1): %llu stale reads out of %d\n", bad, 512 * 256 * 200);
    return 0;
}
