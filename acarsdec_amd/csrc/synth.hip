// synth.hip -- device-side input generators, so that multi-GB synthetic workloads never cross PCIe.
// Not part of the decode path: only fills input buffers before a timed region.
#include <hip/hip_runtime.h>
#include "acg_internal.h"

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// every 16-byte chunk gets distinct, seed-dependent bytes: row r, chunk c -> hash(seed, r, c)
__global__ void fill_random_u8_kernel(uint8_t* dev, size_t pitch, int nrows, size_t chunks_per_row,
                                      uint64_t seed)
{
    const size_t total = (size_t)nrows * chunks_per_row;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        const size_t r = g / chunks_per_row;
        const size_t c = g - r * chunks_per_row;
        const uint64_t k = seed * 0xD1B54A32D192ED03ull + r * 0x100000001B3ull;
        const uint64_t a = splitmix64(k + 2 * c);
        const uint64_t b = splitmix64(k + 2 * c + 1);
        uint4 v;
        v.x = (unsigned int)a; v.y = (unsigned int)(a >> 32);
        v.z = (unsigned int)b; v.w = (unsigned int)(b >> 32);
        *(uint4*)(dev + r * pitch + (c << 4)) = v;
    }
}

extern "C" int acg_launch_fill_random(uint8_t* dev, size_t pitch, int nrows, size_t row_bytes,
                                      uint64_t seed, void* stream)
{
    const size_t cpr = row_bytes / 16;
    hipLaunchKernelGGL(fill_random_u8_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream,
                       dev, pitch, nrows, cpr, seed);
    return (int)hipGetLastError();
}
