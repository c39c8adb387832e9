// valu_beside_loads_probe.hip -- does the return path of the loads matter once the SIMDs are half busy?
// Streaming readers with the down-converter's access pattern (8 waves per CU, runs of 50 KiB per wave, one front in
// address order) and its arithmetic per KiB (16 cvt + 8 pk_add + 16 pk_fma + 2 adds on the 16 bytes a lane holds):
//   reg:  buffer_load_dwordx4 ... nt into VGPRs (what fir_u8_direct_kernel does), DEPTH loads in flight
//   dma:  buffer_load_dwordx4 ... lds nt into a per-wave LDS ring, each lane reads its own 16 bytes back (ds_read_b128)
// each with WORK = 0 (pure reader) and WORK = 1 (arithmetic).  Not product code.
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_beside_loads_probe valu_beside_loads_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WORK>
__device__ __forceinline__ void chew(const u4v d, f2& accA, f2& accB, const f2 (&w)[8])
{
    if (!WORK) {
        accA.x += __uint_as_float(d.x ^ d.y ^ d.z ^ d.w);
        return;
    }
    f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned int word = d[j >> 1];
        const unsigned int sh = (j & 1) * 16;
        f2 t;
        t.x = (float)((word >> sh) & 0xffu) - 127.37f;
        t.y = (float)((word >> (sh + 8)) & 0xffu) - 127.37f;
        const f2 ws = {w[j].y, w[j].x};
        a = __builtin_elementwise_fma(t, w[j], a);
        b = __builtin_elementwise_fma(t, ws, b);
    }
    accA.x += a.x - a.y;
    accB.x += b.x + b.y;
}

template <int DEPTH, int WORK>
__global__ __launch_bounds__(256) void reg_reader(const unsigned char* __restrict__ src, size_t nbytes, float* sink, int run_kib)
{
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const size_t run_bytes = (size_t)run_kib << 10, nrun = nbytes / run_bytes;
    f2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
    f2 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (f2){0.001f * (lane + j), 0.002f * (j + 1)};
    for (size_t r = wave; r < nrun; r += nwaves) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + r * run_bytes), 0, (int)run_bytes, 0x00020000);
        u4v st[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) st[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, d * 1024, 2);
        for (int k = 0; k < run_kib; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __builtin_amdgcn_sched_barrier(0);
                const u4v x = st[d];
                if (k + DEPTH + d < run_kib) st[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (k + DEPTH + d) * 1024, 2);
                __builtin_amdgcn_sched_barrier(0);
                chew<WORK>(x, accA, accB, w);
            }
        }
    }
    if (accA.x + accB.x == 12345.678f) sink[0] = 1.f;
}

// LDS-DMA: 1 KiB per wave-instruction straight into the wave's LDS ring (M0 = LDS byte address of the slot)
__device__ __forceinline__ void dma_1k(const unsigned char* g, unsigned int lds_addr)
{
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

template <int DEPTH, int WORK>
__global__ __launch_bounds__(256) void dma_reader(const unsigned char* __restrict__ src, size_t nbytes, float* sink, int run_kib)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const size_t run_bytes = (size_t)run_kib << 10, nrun = nbytes / run_bytes;
    unsigned char* mine = ring + wv * DEPTH * 1024;
    const unsigned int base = (unsigned int)(size_t)mine;
    f2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
    f2 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (f2){0.001f * (lane + j), 0.002f * (j + 1)};
    for (size_t r = wave; r < nrun; r += nwaves) {
        const unsigned char* gp = src + r * run_bytes + lane * 16;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) dma_1k(gp + d * 1024, base + d * 1024);
        for (int k = 0; k < run_kib; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                // slot d holds position k + d: DEPTH - 1 younger fills may stay in flight
                if (DEPTH == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (DEPTH == 10) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                const u4v x = *(const u4v*)(mine + d * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot is read before it is refilled
                if (k + DEPTH + d < run_kib) dma_1k(gp + (size_t)(k + DEPTH + d) * 1024, base + d * 1024);
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // run tail: nothing refills, drain
                chew<WORK>(x, accA, accB, w);
            }
        }
    }
    if (accA.x + accB.x == 12345.678f) sink[0] = 1.f;
}

template <typename K>
static double timeit(K kernel, const unsigned char* d, size_t nbytes, float* sink, int run_kib, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(512), dim3(256), lds, 0, d, nbytes, sink, run_kib);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(512), dim3(256), lds, 0, d, nbytes, sink, run_kib);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)(nbytes / ((size_t)run_kib << 10) * ((size_t)run_kib << 10)) * 3 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv)
{
    const size_t nbytes = (argc > 1 ? (size_t)atof(argv[1]) : 16.0) * (1ull << 30);
    unsigned char* d = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&d, nbytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0x5a, nbytes);
    (void)hipDeviceSynchronize();
    printf("buffer %.1f GiB, 8 waves per CU, runs of 50 KiB\n", nbytes / 1073741824.0);
    printf("reg  depth 5   pure %6.0f   with arithmetic %6.0f GB/s\n", timeit(reg_reader<5, 0>, d, nbytes, sink, 50, 0), timeit(reg_reader<5, 1>, d, nbytes, sink, 50, 0));
    printf("reg  depth 10  pure %6.0f   with arithmetic %6.0f GB/s\n", timeit(reg_reader<10, 0>, d, nbytes, sink, 50, 0), timeit(reg_reader<10, 1>, d, nbytes, sink, 50, 0));
    printf("dma  depth 5   pure %6.0f   with arithmetic %6.0f GB/s\n", timeit(dma_reader<5, 0>, d, nbytes, sink, 50, 4 * 5 * 1024), timeit(dma_reader<5, 1>, d, nbytes, sink, 50, 4 * 5 * 1024));
    printf("dma  depth 10  pure %6.0f   with arithmetic %6.0f GB/s\n", timeit(dma_reader<10, 0>, d, nbytes, sink, 50, 4 * 10 * 1024), timeit(dma_reader<10, 1>, d, nbytes, sink, 50, 4 * 10 * 1024));
    return 0;
}
