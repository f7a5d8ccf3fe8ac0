// LDS canary: does a neighbour workgroup on the same CU (another stream / another process) write into LDS it does not own?
// Every workgroup fills 12 KiB of LDS with a pattern, sleeps, checks it.  hipcc --offload-arch=gfx950 -shared -fPIC -o lds_canary.so
#include <hip/hip_runtime.h>
#define WORDS 3104
__global__ void __launch_bounds__(256) k_canary(unsigned *bad, unsigned *sample, int spin)
{
    __shared__ unsigned sm[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) sm[i] = 0xC0DE0000u | (unsigned)i;
    __syncthreads();
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(127);
    __syncthreads();
    for (int i = threadIdx.x; i < WORDS; i += 256) {
        const unsigned v = sm[i];
        if (v != (0xC0DE0000u | (unsigned)i)) {
            const unsigned k = atomicAdd(bad, 1u);
            if (k < 64) { sample[2 * k] = (unsigned)i; sample[2 * k + 1] = v; }
        }
    }
}
extern "C" int canary_launch(void *stream, unsigned *bad, unsigned *sample, int blocks, int spin)
{
    hipLaunchKernelGGL(k_canary, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bad, sample, spin);
    return (int)hipGetLastError();
}

// LDS poison: every CU's whole LDS filled with a pattern (one 160 KiB workgroup per CU, twice over) -- a kernel that reads LDS it has
// not written gives different results for different patterns
__global__ void __launch_bounds__(1024) k_lds_fill(unsigned pattern, int words, unsigned *sink)
{
    extern __shared__ unsigned dyn[];
    for (int i = threadIdx.x; i < words; i += 1024) dyn[i] = pattern;
    __syncthreads();
    if (dyn[(threadIdx.x * 97u) % (unsigned)words] != pattern) *sink = 1;
}
extern "C" int lds_fill(void *stream, unsigned pattern, unsigned *sink)
{
    static bool raised = false;
    const int bytes = 160 * 1024;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_fill), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
        raised = true;
    }
    hipLaunchKernelGGL(k_lds_fill, dim3(512), dim3(1024), bytes, (hipStream_t)stream, pattern, bytes / 4, sink);
    return (int)hipGetLastError();
}
