// hbm_read_probe.hip -- what read bandwidth does this MI355X give to differently shaped streaming
// readers of one 3.4 GB buffer (the size of one default bench.py down-converter launch)?  Evidence for
// the "practical read ceiling" quoted next to the 8 TB/s spec figure in DESIGN.md.  Not product code.
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_read_probe hbm_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// (1) register loads: WG of 256 threads walks contiguous TILE-byte tiles, U loads of 16 B per thread in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void reg_reader(const u4v* __restrict__ src, size_t nvec, unsigned int* sink, int tile_vec)
{
    const size_t ntile = nvec / tile_vec;
    u4v acc = {0, 0, 0, 0};
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const u4v* p = src + t * tile_vec;
        for (int i = threadIdx.x; i < tile_vec; i += 256 * U) {
            u4v v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = i + u * 256;
                if (k < tile_vec) v[u] = NT ? __builtin_nontemporal_load(p + k) : p[k];
                else v[u] = (u4v){0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// (2) LDS-DMA: every wave of the WG streams its own contiguous chunks into an LDS ring,
// DEPTH KiB-fills in flight per wave (global_load_lds_dwordx4 = 1 KiB per wave instruction)
template <bool NT>
__device__ __forceinline__ void dma_1k(const void* g, unsigned int lds_off)
{
    unsigned int keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}

template <int DEPTH, bool NT>
__global__ void dma_reader(const unsigned char* __restrict__ src, size_t nbytes, unsigned int* sink, int chunk_kib)
{
    extern __shared__ unsigned char ring[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwave = blockDim.x >> 6;
    const size_t chunk = (size_t)chunk_kib << 10;
    const size_t nchunk = nbytes / chunk;
    const unsigned int base = (unsigned int)(size_t)ring + wave * DEPTH * 1024;   // LDS byte address of this wave's ring
    int slot = 0;
    for (size_t c = (size_t)blockIdx.x * nwave + wave; c < nchunk; c += (size_t)gridDim.x * nwave) {
        const unsigned char* p = src + c * chunk + lane * 16;
        for (int k = 0; k < chunk_kib; ++k) {
            dma_1k<NT>(p + ((size_t)k << 10), base + slot * 1024);
            slot = (slot + 1 == DEPTH) ? 0 : slot + 1;
            // keep at most DEPTH-1 fills outstanding before reusing a slot
            if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x] == 0x5a && ring[threadIdx.x + 1] == 0xa5 && src == nullptr) sink[0] = 1;
}


// (3) the down-converter's load structure without its arithmetic, feature by feature:
//   RUN   tiles are taken in runs of RUN consecutive tiles per workgroup (1 = interleaved like reg_reader)
//   LDSW  stage the tile through LDS (ds_write_b128 + barrier) instead of xor-ing registers
//   STORE one 256-byte store per tile (the dm row segment)
//   PRE   issue the loads of tile t+1 before consuming tile t (one tile always in flight)
template <int RUN, bool LDSW, bool STORE, bool PRE>
__global__ __launch_bounds__(256) void fir_like_reader(const u4v* __restrict__ src, size_t nvec, unsigned int* sink, float* out, int tile_vec)
{
    extern __shared__ unsigned char smem[];
    u4v* tileL = (u4v*)smem;
    const size_t ntile = nvec / tile_vec;
    const size_t nrun = ntile / RUN;
    constexpr int U = 7;
    u4v acc = {0, 0, 0, 0};
    u4v v[U];
    auto fetch = [&](size_t t) {
        const u4v* p = src + t * tile_vec;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = threadIdx.x + u * 256;
            if (k < tile_vec) v[u] = __builtin_nontemporal_load(p + k);
        }
    };
    auto consume = [&](size_t t) {
        if (LDSW) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = threadIdx.x + u * 256;
                if (k < tile_vec) tileL[k] = v[u];
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = threadIdx.x + u * 256;
                if (k < tile_vec) acc ^= v[u];
            }
        }
    };
    auto after = [&](size_t t) {
        if (LDSW) {
            acc ^= tileL[(threadIdx.x * 7) % tile_vec];
            __syncthreads();
        }
        if (STORE && threadIdx.x < 64) out[t * 64 + threadIdx.x] = (float)acc.x;
    };
    for (size_t r = blockIdx.x; r < nrun; r += gridDim.x) {
        if (PRE) {
            fetch(r * RUN);
            for (int q = 0; q < RUN; ++q) {
                const size_t t = r * RUN + q;
                consume(t);
                if (q + 1 < RUN) fetch(t + 1);
                after(t);
            }
        } else {
            for (int q = 0; q < RUN; ++q) {
                const size_t t = r * RUN + q;
                fetch(t);
                consume(t);
                after(t);
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// (4) the real kernel's skeleton, feature by feature (bit flags):
//   1  second barrier + 4 KB partial-sum exchange through LDS per tile
//   2  wave 0 finishes the tile: f64 sqrt per lane + one 256-byte store
//   4  wave-uniform branches around every load / LDS write (as fir_u8_persist_kernel)
//   8  dynamic runs of 4 tiles from an atomic dispenser instead of the static contiguous partition
template <int F>
__global__ __launch_bounds__(256) void fir_skel(const unsigned char* __restrict__ src, size_t ntile, unsigned int* sink, float* out,
                                                int cpr, unsigned int* counter)
{
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_chunks = 64 * cpr;
    const size_t tile_bytes = (size_t)tile_chunks * 16;
    float4* red = (float4*)(smem + tile_bytes);
    int* s_next = (int*)(smem + tile_bytes + 8192);
    uint4 stage[10];
    const unsigned int voff = tid * 16;
    auto fetch = [&](size_t t) {
        const unsigned char* tb = src + t * tile_bytes;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if (F & 4) {
                if (wave + 4 * i < cpr) {
                    const u4v x = __builtin_nontemporal_load((const u4v*)(tb + (size_t)i * 4096 + voff));
                    stage[i] = make_uint4(x.x, x.y, x.z, x.w);
                }
            } else {
                const int c = tid + i * 256;
                if (c < tile_chunks) {
                    const u4v x = __builtin_nontemporal_load((const u4v*)(tb + (size_t)c * 16));
                    stage[i] = make_uint4(x.x, x.y, x.z, x.w);
                }
            }
        }
    };
    size_t g0, g1;
    const size_t nrun = (ntile + 3) / 4;
    if (F & 8) { g0 = (size_t)blockIdx.x * 4; g1 = g0 + 4 < ntile ? g0 + 4 : ntile; }
    else { g0 = ntile * blockIdx.x / gridDim.x; g1 = ntile * (blockIdx.x + 1) / gridDim.x; }
    if (g0 >= g1) return;
    float acc = 0.f;
    unsigned int pending = 0;
    bool have_prev = false; size_t prev_g = 0; int par = 0;
    fetch(g0);
    for (size_t g = g0;;) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = tid + i * 256;
            if ((F & 4) ? (wave + 4 * i < cpr) : (c < tile_chunks)) *(uint4*)(smem + (size_t)c * 16) = stage[i];
        }
        if (F & 8) {
            if (g == g0 && tid == 0) pending = atomicAdd(counter, 1u);
            if (g == g0 + 1 && tid == 0) *s_next = (int)(gridDim.x + pending);
        }
        __syncthreads();
        bool more = g + 1 < g1;
        size_t ng = g + 1, ng0 = g0, ng1 = g1;
        if ((F & 8) && !more) {
            const size_t nr = (g1 - g0 >= 2) ? (size_t)*s_next : nrun;
            if (nr < nrun) { ng0 = nr * 4; ng1 = ng0 + 4 < ntile ? ng0 + 4 : ntile; ng = ng0; more = true; }
        }
        if (more) fetch(ng);
        if ((F & 16) && wave == 0 && have_prev) {      // finish the PREVIOUS tile now, off the barrier-to-barrier path
            const float4* rp = red + ((par ^ 1) * 256);
            const float4 r0 = rp[lane], r1 = rp[64 + lane], r2 = rp[128 + lane], r3 = rp[192 + lane];
            const float Dr = ((r0.x + r1.x) + (r2.x + r3.x)) - ((r0.y + r1.y) + (r2.y + r3.y));
            const float Di = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r0.w + r1.w) + (r2.w + r3.w));
            out[prev_g * 64 + lane] = (float)__dsqrt_rn((double)Dr * (double)Dr + (double)Di * (double)Di);
        }
        const float4 q = *(const float4*)(smem + (size_t)lane * cpr * 16 + wave * 16);
        acc += q.x;
        if (F & 16) {
            red[par * 256 + wave * 64 + lane] = q;
            __syncthreads();
            have_prev = true; prev_g = g; par ^= 1;
        } else if (F & 1) {
            red[wave * 64 + lane] = q;
            __syncthreads();
        }
        if ((F & 2) && !(F & 16) && wave == 0) {
            float4 r0 = q, r1 = q, r2 = q, r3 = q;
            if (F & 1) { r0 = red[lane]; r1 = red[64 + lane]; r2 = red[128 + lane]; r3 = red[192 + lane]; }
            const float Dr = ((r0.x + r1.x) + (r2.x + r3.x)) - ((r0.y + r1.y) + (r2.y + r3.y));
            const float Di = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r0.w + r1.w) + (r2.w + r3.w));
            out[g * 64 + lane] = (float)__dsqrt_rn((double)Dr * (double)Dr + (double)Di * (double)Di);
        }
        if (!more) break;
        g = ng; g0 = ng0; g1 = ng1;
    }
    if ((F & 8) && tid == 0) {
        const unsigned int d = atomicAdd(counter + 1, 1u);
        if (d == gridDim.x - 1) { counter[0] = 0; counter[1] = 0; }
    }
    if (acc == 1.2345f) sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static double time_it(F launch, size_t bytes, int reps = 6)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    double best = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double gbs = bytes / (ms * 1e-3) / 1e9;
        if (gbs > best) best = gbs;
    }
    CK(hipGetLastError());
    return best;
}

int main()
{
    const size_t bytes = (size_t)1024 * 8 * 1024 * 400;        // 1024 channels x 8 callbacks x 1024 windows x 400 B
    unsigned char* d; unsigned int* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    const size_t nvec = bytes / 16;
    printf("buffer %.2f GB\n", bytes / 1e9);
    const int tile_vec = 1600;                                   // 25.6 KB = one FIR tile at M = 200
    for (int wg : {1, 2, 4, 5, 8}) {
        const int grid = 256 * wg;
        printf("reg U=4 nt  wg/cu=%d : %.0f GB/s\n", wg, time_it([&] { hipLaunchKernelGGL((reg_reader<4, true>), dim3(grid), dim3(256), 0, 0, (const u4v*)d, nvec, sink, tile_vec); }, bytes));
        printf("reg U=7 nt  wg/cu=%d : %.0f GB/s\n", wg, time_it([&] { hipLaunchKernelGGL((reg_reader<7, true>), dim3(grid), dim3(256), 0, 0, (const u4v*)d, nvec, sink, tile_vec); }, bytes));
        printf("reg U=7 def wg/cu=%d : %.0f GB/s\n", wg, time_it([&] { hipLaunchKernelGGL((reg_reader<7, false>), dim3(grid), dim3(256), 0, 0, (const u4v*)d, nvec, sink, tile_vec); }, bytes));
    }
    CK(hipFuncSetAttribute((const void*)dma_reader<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)dma_reader<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)dma_reader<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)dma_reader<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)dma_reader<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int waves : {1, 2, 4, 8}) {
        for (int wg : {1, 2}) {
            const int grid = 256 * wg;
            const int thr = 64 * waves;
            if ((size_t)waves * 32 * 1024 * wg <= 160 * 1024)
                printf("dma depth32 nt  waves/wg=%d wg/cu=%d chunk=25KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<32, true>), dim3(grid), dim3(thr), waves * 32 * 1024, 0, d, bytes, sink, 25); }, bytes));
            if ((size_t)waves * 16 * 1024 * wg <= 160 * 1024)
                printf("dma depth16 nt  waves/wg=%d wg/cu=%d chunk=25KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<16, true>), dim3(grid), dim3(thr), waves * 16 * 1024, 0, d, bytes, sink, 25); }, bytes));
            if ((size_t)waves * 8 * 1024 * wg <= 160 * 1024) {
                printf("dma depth8  nt  waves/wg=%d wg/cu=%d chunk=25KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<8, true>), dim3(grid), dim3(thr), waves * 8 * 1024, 0, d, bytes, sink, 25); }, bytes));
                printf("dma depth8  def waves/wg=%d wg/cu=%d chunk=25KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<8, false>), dim3(grid), dim3(thr), waves * 8 * 1024, 0, d, bytes, sink, 25); }, bytes));
                printf("dma depth8  nt  waves/wg=%d wg/cu=%d chunk=100KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<8, true>), dim3(grid), dim3(thr), waves * 8 * 1024, 0, d, bytes, sink, 100); }, bytes));
            }
            printf("dma depth4  nt  waves/wg=%d wg/cu=%d chunk=25KiB: %.0f GB/s\n", waves, wg, time_it([&] { hipLaunchKernelGGL((dma_reader<4, true>), dim3(grid), dim3(thr), waves * 4 * 1024, 0, d, bytes, sink, 25); }, bytes));
        }
    }

    {
        float* out; CK(hipMalloc(&out, (nvec / tile_vec) * 64 * sizeof(float)));
        const size_t lds = (size_t)tile_vec * 16;
        for (int wg : {2, 5}) {
            const int grid = 256 * wg;
#define FL(RUN, L, S, P) printf("fir-like run=%d lds=%d store=%d prefetch=%d wg/cu=%d : %.0f GB/s\n", RUN, L, S, P, wg, \
            time_it([&] { hipLaunchKernelGGL((fir_like_reader<RUN, L, S, P>), dim3(grid), dim3(256), lds, 0, (const u4v*)d, nvec, sink, out, tile_vec); }, bytes));
            FL(1, false, false, false)
            FL(4, false, false, false)
            FL(4, true, false, false)
            FL(4, true, true, false)
            FL(4, true, true, true)
            FL(4, false, true, false)
            FL(1, true, true, false)
            FL(1, false, true, false)
            FL(16, true, true, true)
        }
    }

    {
        float* out; CK(hipMalloc(&out, (nvec / tile_vec) * 64 * sizeof(float)));
        unsigned int* counter; CK(hipMalloc(&counter, 8)); CK(hipMemset(counter, 0, 8));
        const size_t ntile = nvec / tile_vec;
        const size_t lds = (size_t)tile_vec * 16 + 8192 + 16;
        const int grid = 256 * 4;
#define SK(F) printf("skeleton flags=%2d (wg/cu=4): %.0f GB/s\n", F, \
            time_it([&] { hipLaunchKernelGGL((fir_skel<F>), dim3(grid), dim3(256), lds, 0, d, ntile, sink, out, 25, counter); }, bytes));
        SK(0) SK(1) SK(3) SK(19) SK(7) SK(23) SK(8) SK(15) SK(31) SK(0)
    }
    return 0;
}
