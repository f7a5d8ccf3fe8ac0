// Probe of buffer_load_dwordx4 ... offen lds on gfx950 (a candidate for the activation stream of the LDS-DMA pipeline: an SGPR
// resource descriptor + a 32-bit per-lane offset instead of a 64-bit per-lane address, and the hardware's range check instead
// of a select against a zero page):
//   (1) where the data lands: M0 base + lane * 16?
//   (2) a lane whose offset is beyond num_records: does it write ZEROS to its LDS slot, or leave the slot untouched?
// Build: hipcc --offload-arch=gfx950 -O3 buffer_lds_semantics.hip -o buffer_lds_semantics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned *in, unsigned *out, unsigned nbytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long a = (unsigned long long)in;
    i32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)nbytes);
    rsrc[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // lane l reads item (l ^ 3) of its wave's block; odd lanes of wave 1 point far outside the buffer
    unsigned voff = (unsigned)((wave * 64 + (lane ^ 3)) * 16);
    if (wave == 1 && (lane & 1)) voff = 0x7ffffff0u;
    const unsigned dst = 1024 + wave * 2048;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(dst) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += blockDim.x) out[i] = lds[i];
}

int main()
{
    std::vector<unsigned> h(1 << 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (unsigned)i;
    unsigned *in, *out;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&out, 4096 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 16384, 0, in, out, 128 * 16);
    std::vector<unsigned> o(4096);
    hipMemcpy(o.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
    int landed = 0, zeros = 0, untouched = 0, other = 0;
    for (int wave = 0; wave < 2; ++wave)
        for (int lane = 0; lane < 64; ++lane) {
            const unsigned *slot = &o[(1024 + wave * 2048) / 4 + lane * 4];
            const unsigned expect = 0x10000000u + (unsigned)((wave * 64 + (lane ^ 3)) * 4);
            const bool oob = wave == 1 && (lane & 1);
            if (!oob) landed += (slot[0] == expect && slot[3] == expect + 3);
            else if (slot[0] == 0 && slot[3] == 0) ++zeros;
            else if (slot[0] == 0xdeadbeefu) ++untouched;
            else ++other;
        }
    printf("in-range lanes landed at M0 + 16 * lane with their (permuted) source: %d of 96\n", landed);
    printf("out-of-range lanes: %d wrote zeros, %d left the slot untouched, %d something else (of 32)\n", zeros, untouched, other);
    return 0;
}
