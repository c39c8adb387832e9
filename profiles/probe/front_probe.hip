// front_probe.hip -- how does the read bandwidth of a wave-granular streaming reader depend on the shape of the
// "front" it moves through memory?  Every wave reads runs of RUN KiB (1 KiB = one 16-byte-per-lane wave-load,
// non-temporal buffer loads, DEPTH loads in flight per wave); run r of iteration i is run index i * nwaves + w
// (static interleave), so the grid reads one front of nwaves * RUN KiB that moves through the buffer in address
// order.  Evidence for the down-converter's run size / residency choice.  Not product code.
//   build: hipcc --offload-arch=gfx950 -O3 -o front_probe front_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void reader(const unsigned char* __restrict__ src, size_t nbytes, unsigned int* sink, int run_kib)
{
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6);
    const size_t run_bytes = (size_t)run_kib << 10;
    const size_t nrun = nbytes / run_bytes;
    u4v acc = {0, 0, 0, 0};
    for (size_t r = wave; r < nrun; r += nwaves) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + r * run_bytes), 0, (int)run_bytes, 0x00020000);
        for (int k = 0; k < run_kib; k += DEPTH) {
            u4v v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (k + d) * 1024, 2);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH>
static double run(const unsigned char* d, size_t nbytes, unsigned int* sink, int wg_per_cu, int waves_per_wg, int run_kib)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(256 * wg_per_cu), blk(64 * waves_per_wg);
    hipLaunchKernelGGL(reader<DEPTH>, grid, blk, 0, 0, d, nbytes, sink, run_kib);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(reader<DEPTH>, grid, blk, 0, 0, d, nbytes, sink, run_kib);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)(nbytes / ((size_t)run_kib << 10) * ((size_t)run_kib << 10)) * 3 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv)
{
    const size_t nbytes = (argc > 1 ? (size_t)atof(argv[1]) : 16.0) * (1ull << 30);
    unsigned char* d = nullptr;
    unsigned int* sink = nullptr;
    if (hipMalloc(&d, nbytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0x5a, nbytes);
    hipDeviceSynchronize();
    printf("buffer %.1f GiB\n", nbytes / 1073741824.0);
    const int runs[] = {10, 50, 250, 1000, 4000};
    for (int wpc : {4, 8, 16})
        for (int run_kib : runs) {
            const int wg = wpc / 4;
            printf("waves/CU %2d run %4d KiB front %7.1f MiB:  depth 5 %6.0f   depth 10 %6.0f GB/s\n", wpc, run_kib,
                   256.0 * wpc * run_kib / 1024.0, run<5>(d, nbytes, sink, wg, 4, run_kib), run<10>(d, nbytes, sink, wg, 4, run_kib));
            fflush(stdout);
        }
    return 0;
}
