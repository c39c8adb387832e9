// write_beside_read_probe.hip -- what a 1 % write stream costs a streaming reader, by the shape of the writes.
// The down-converter reads 25.6 KB per 64-window tile and writes 256 bytes of dm for it; with the dm rows folded into one
// (no write stream to HBM) it reads 0.84-0.85 of spec, with the real dm 0.75-0.83 depending on where the decoder's
// buffers lie (profiles/probe/placement_probe.py).  This probe takes the arithmetic away: every wave reads runs of
// RUN tiles (25 wave-loads of 1 KiB per tile, DEPTH in flight, static interleave: one front in address order) and
// writes 64 floats per tile in one of several ways.  Not product code.
//   build: hipcc --offload-arch=gfx950 -O3 -o write_beside_read_probe write_beside_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

enum { W_NONE = 0, W_TILE, W_TILE_NT, W_TILE_SC, W_RUN_END, W_RUN_END_NT, W_FOLDED, W_HALFWAVE_2TILES,
       W_TILE_SC0, W_TILE_SC1, W_TILE_NT_SC, W_RUN_END_SC, W_RUN_END_NT_SC, W_4TILES_NT, NMODE };
static const char* mode_name[NMODE] = {
    "no writes",
    "256 B per tile, plain store (the kernel's way)",
    "256 B per tile, nt store",
    "256 B per tile, sc0 sc1 store",
    "whole run (RUN x 256 B) at the end of the run, plain b128 stores",
    "whole run at the end of the run, nt b128 stores",
    "256 B per tile, all rows folded into the first 16 KiB",
    "512 B per two tiles (lanes 0..31 hold tile a, 32..63 tile b: one store of 2 x 256 B)",
    "256 B per tile, sc0 store",
    "256 B per tile, sc1 store",
    "256 B per tile, nt sc0 sc1 store",
    "whole run at the end of the run, sc0 sc1 b128 stores",
    "whole run at the end of the run, nt sc0 sc1 b128 stores",
    "1 KiB per four tiles, nt b128 store",
};

template <int MODE, int DEPTH, int RUN>
__global__ __launch_bounds__(256) void reader_writer(const unsigned char* __restrict__ src, size_t nbytes, float* __restrict__ out,
                                                      unsigned int* sink)
{
    constexpr int TILE_KIB = 25;
    __shared__ float keep[4][RUN * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const unsigned int wave = blockIdx.x * (blockDim.x >> 6) + wv;
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6);
    const size_t run_bytes = (size_t)RUN * TILE_KIB * 1024;
    const size_t nrun = nbytes / run_bytes;
    u4v acc = {0, 0, 0, 0};
    for (size_t r = wave; r < nrun; r += nwaves) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + r * run_bytes), 0, (int)run_bytes, 0x00020000);
        float* row = out + r * (RUN * 64);                      // the run's 64 RUN floats of output, contiguous
        for (int t = 0; t < RUN; ++t) {
#pragma unroll
            for (int k = 0; k < TILE_KIB; k += DEPTH) {
                u4v v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (t * TILE_KIB + k + d) * 1024, 2);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
            }
            const float val = __uint_as_float((acc.x & 0x007fffffu) | 0x3f800000u);
            if (MODE == W_TILE) row[t * 64 + lane] = val;
            if (MODE == W_TILE_NT) __builtin_nontemporal_store(val, &row[t * 64 + lane]);
            if (MODE == W_TILE_SC) {
                float* p = &row[t * 64 + lane];
                asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(val) : "memory");
            }
            if (MODE == W_TILE_SC0) { float* p = &row[t * 64 + lane]; asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(val) : "memory"); }
            if (MODE == W_TILE_SC1) { float* p = &row[t * 64 + lane]; asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(val) : "memory"); }
            if (MODE == W_TILE_NT_SC) { float* p = &row[t * 64 + lane]; asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(val) : "memory"); }
            if (MODE == W_FOLDED) out[(t & 63) * 64 + lane] = val;
            if (MODE == W_RUN_END || MODE == W_RUN_END_NT || MODE == W_RUN_END_SC || MODE == W_RUN_END_NT_SC || MODE == W_4TILES_NT) keep[wv][t * 64 + lane] = val;
            if (MODE == W_4TILES_NT && (t & 3) == 3) {
                const f4v q = *(const f4v*)&keep[wv][((t >> 2) * 64 + lane) * 4];
                __builtin_nontemporal_store(q, (f4v*)&row[((t >> 2) * 64 + lane) * 4]);
            }
            if (MODE == W_HALFWAVE_2TILES) {
                keep[wv][(t & 1) * 64 + lane] = val;
                if (t & 1) {
                    // lanes 0..63 write floats 0..127 of the tile pair as 2 floats per lane: one 512-byte store
                    const float a = keep[wv][2 * lane], b = keep[wv][2 * lane + 1];
                    float2 ab = make_float2(a, b);
                    *(float2*)&row[(t - 1) * 64 + 2 * lane] = ab;
                }
            }
        }
        if (MODE == W_RUN_END || MODE == W_RUN_END_NT || MODE == W_RUN_END_SC || MODE == W_RUN_END_NT_SC) {
            // RUN x 64 floats = RUN x 16 float4: lane writes float4 number (i * 64 + lane)
#pragma unroll
            for (int i = 0; i < RUN / 4; ++i) {
                const f4v q = *(const f4v*)&keep[wv][(i * 64 + lane) * 4];
                f4v* p = (f4v*)&row[(i * 64 + lane) * 4];
                if (MODE == W_RUN_END) *p = q;
                else if (MODE == W_RUN_END_NT) __builtin_nontemporal_store(q, p);
                else if (MODE == W_RUN_END_SC) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(q) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(q) : "memory");
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH, int RUN>
static double run(const unsigned char* d, size_t nbytes, float* out, unsigned int* sink, int wg_per_cu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(256 * wg_per_cu), blk(256);
    hipLaunchKernelGGL((reader_writer<MODE, DEPTH, RUN>), grid, blk, 0, 0, d, nbytes, out, sink);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((reader_writer<MODE, DEPTH, RUN>), grid, blk, 0, 0, d, nbytes, out, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const size_t run_bytes = (size_t)RUN * 25 * 1024;
    return (double)(nbytes / run_bytes * run_bytes) * 3 / (ms * 1e-3) / 1e9;
}


// all waves hold their results in LDS and flush when the chip-wide 100 MHz clock crosses a multiple of 2^SHIFT ticks (or
// the buffer is full, or the run ends): the writes of the whole chip come in bursts instead of a trickle
template <int SHIFT, int DEPTH, int RUN, int CAP>
__global__ __launch_bounds__(256) void reader_epoch_writer(const unsigned char* __restrict__ src, size_t nbytes, float* __restrict__ out,
                                                            unsigned int* sink)
{
    constexpr int TILE_KIB = 25;
    __shared__ float keep[4][CAP * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const unsigned int wave = blockIdx.x * (blockDim.x >> 6) + wv;
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6);
    const size_t run_bytes = (size_t)RUN * TILE_KIB * 1024;
    const size_t nrun = nbytes / run_bytes;
    u4v acc = {0, 0, 0, 0};
    unsigned long long epoch = __builtin_amdgcn_s_memrealtime() >> SHIFT;
    for (size_t r = wave; r < nrun; r += nwaves) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + r * run_bytes), 0, (int)run_bytes, 0x00020000);
        float* row = out + r * (RUN * 64);
        int first = 0;                                              // tiles [first, t] wait in LDS
        for (int t = 0; t < RUN; ++t) {
#pragma unroll
            for (int k = 0; k < TILE_KIB; k += DEPTH) {
                u4v v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (t * TILE_KIB + k + d) * 1024, 2);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
            }
            const float val = __uint_as_float((acc.x & 0x007fffffu) | 0x3f800000u);
            keep[wv][(t - first) * 64 + lane] = val;
            const unsigned long long e = __builtin_amdgcn_s_memrealtime() >> SHIFT;
            if (e != epoch || t - first + 1 == CAP || t + 1 == RUN) {
                epoch = e;
                const int nq = (t - first + 1) * 16;                // float4s to write
                for (int i = lane; i < nq; i += 64) {
                    const f4v q = *(const f4v*)&keep[wv][i * 4];
                    f4v* p = (f4v*)&row[first * 64 + i * 4];
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(q) : "memory");
                }
                first = t + 1;
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int SHIFT, int DEPTH, int RUN, int CAP>
static double run_epoch(const unsigned char* d, size_t nbytes, float* out, unsigned int* sink, int wg_per_cu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(256 * wg_per_cu), blk(256);
    hipLaunchKernelGGL((reader_epoch_writer<SHIFT, DEPTH, RUN, CAP>), grid, blk, 0, 0, d, nbytes, out, sink);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((reader_epoch_writer<SHIFT, DEPTH, RUN, CAP>), grid, blk, 0, 0, d, nbytes, out, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const size_t run_bytes = (size_t)RUN * 25 * 1024;
    return (double)(nbytes / run_bytes * run_bytes) * 3 / (ms * 1e-3) / 1e9;
}

template <int DEPTH, int RUN>
static void sweep(const unsigned char* d, size_t nbytes, float* out, unsigned int* sink, int wg_per_cu, const char* label)
{
    printf("%s, %d waves per CU, %d loads in flight, runs of %d tiles (%d KiB):\n", label, 4 * wg_per_cu, DEPTH, RUN, RUN * 25);
    printf("   %-90s %6.0f GB/s\n", mode_name[0], run<W_NONE, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[1], run<W_TILE, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[2], run<W_TILE_NT, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[3], run<W_TILE_SC, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[4], run<W_RUN_END, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[5], run<W_RUN_END_NT, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[6], run<W_FOLDED, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[7], run<W_HALFWAVE_2TILES, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[8], run<W_TILE_SC0, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[9], run<W_TILE_SC1, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[10], run<W_TILE_NT_SC, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[11], run<W_RUN_END_SC, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[12], run<W_RUN_END_NT_SC, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    printf("   %-90s %6.0f GB/s\n", mode_name[13], run<W_4TILES_NT, DEPTH, RUN>(d, nbytes, out, sink, wg_per_cu));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    // write_beside_read_probe [GiB of input = 24] [experiment = 0: all]
    //   1 store shapes on three output allocations   2 chip-wide bursts by epoch length   3 offsets inside one 6 GiB allocation
    //   4 allocation sizes   5 two inputs x several outputs (the pair decides)
    // (profiles/r02_experiments/write_beside_read{,_2,_3}.txt: 1 (+ 2 in _3);  _4: 3;  _5: 4;  _6: 5)
    const size_t nbytes = (argc > 1 ? (size_t)atof(argv[1]) : 24.0) * (1ull << 30);
    const int which = argc > 2 ? atoi(argv[2]) : 0;
    unsigned char* d = nullptr;
    unsigned int* sink = nullptr;
    if (hipMalloc(&d, nbytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0x5a, nbytes);
    const size_t out_bytes = nbytes / 100 + (1 << 20);           // 256 B per 25.6 KB
    float* outs[3] = {nullptr, nullptr, nullptr};
    void* spacer[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        if (hipMalloc(&outs[i], out_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMalloc(&spacer[i], (size_t)(i + 1) * 37 * 4096 * 1024 + 12345);       // the next output buffer lands elsewhere
    }
    (void)hipDeviceSynchronize();
    printf("input %.1f GiB at %p; outputs of %.0f MiB at %p %p %p\n", nbytes / 1073741824.0, (void*)d, out_bytes / 1048576.0, (void*)outs[0],
           (void*)outs[1], (void*)outs[2]);
    if (which == 0 || which == 1) {
        sweep<5, 16>(d, nbytes, outs[0], sink, 2, "output buffer 0");
        sweep<5, 16>(d, nbytes, outs[1], sink, 2, "output buffer 1");
        sweep<5, 16>(d, nbytes, outs[2], sink, 2, "output buffer 2");
    }
    if (which == 0 || which == 2) {
        for (int b = 0; b < 3; b += 2) {
            printf("output buffer %d, 8 waves per CU, runs of 64 tiles, results held in LDS (up to 32 tiles) and flushed when the 100 MHz clock crosses 2^k ticks:\n", b);
            printf("   no clock, flushed every 32 tiles (k = 40)  %6.0f GB/s\n", run_epoch<40, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 10 (10 us)   %6.0f GB/s\n", run_epoch<10, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 11 (20 us)   %6.0f GB/s\n", run_epoch<11, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 12 (41 us)   %6.0f GB/s\n", run_epoch<12, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 13 (82 us)   %6.0f GB/s\n", run_epoch<13, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 14 (164 us)  %6.0f GB/s\n", run_epoch<14, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            printf("   k = 15 (328 us)  %6.0f GB/s\n", run_epoch<15, 5, 64, 32>(d, nbytes, outs[b], sink, 2));
            fflush(stdout);
        }
    }
    if (which == 0 || which == 3) {
        // placement inside ONE allocation: is it the address bits or the backing of the allocation?
        unsigned char* slab = nullptr;
        if (hipMalloc(&slab, 6ull << 30) != hipSuccess) { printf("slab alloc failed\n"); return 1; }
        printf("slab of 6 GiB at %p: output at slab + offset;  plain per-tile store | run-end sc0 sc1 b128 | epoch 82 us (runs of 64)\n", (void*)slab);
        const size_t offs[] = {0, 1ull << 20, 2ull << 20, 64ull << 20, 256ull << 20, 512ull << 20, 1ull << 30, (1ull << 30) + (300ull << 20), 2ull << 30,
                               3ull << 30, (3ull << 30) + (64ull << 10), 4ull << 30, 5ull << 30};
        for (size_t o : offs) {
            float* out = (float*)(slab + o);
            printf("   offset %8.2f MiB: %6.0f | %6.0f | %6.0f GB/s\n", o / 1048576.0, run<W_TILE, 5, 16>(d, nbytes, out, sink, 2),
                   run<W_RUN_END_SC, 5, 16>(d, nbytes, out, sink, 2), run_epoch<13, 5, 64, 32>(d, nbytes, out, sink, 2));
            fflush(stdout);
        }
        printf("the three separate allocations again (plain per-tile store): %6.0f %6.0f %6.0f GB/s\n", run<W_TILE, 5, 16>(d, nbytes, outs[0], sink, 2),
               run<W_TILE, 5, 16>(d, nbytes, outs[1], sink, 2), run<W_TILE, 5, 16>(d, nbytes, outs[2], sink, 2));
        (void)hipFree(slab);
    }
    if (which == 0 || which == 4) {
        printf("output = base of a fresh allocation of the given size; plain per-tile store | run-end sc0 sc1 b128 | epoch 82 us\n");
        const size_t sizes_mib[] = {248, 256, 248, 512, 248, 1024, 248, 2048, 248, 4096, 248, 8192, 248};
        for (size_t mib : sizes_mib) {
            void* a = nullptr;                                       // (kept alive: the next one lands elsewhere)
            if (hipMalloc(&a, mib << 20) != hipSuccess) { printf("alloc of %zu MiB failed\n", mib); break; }
            float* out = (float*)a;
            printf("   %5zu MiB at %p: %6.0f | %6.0f | %6.0f GB/s\n", mib, a, run<W_TILE, 5, 16>(d, nbytes, out, sink, 2),
                   run<W_RUN_END_SC, 5, 16>(d, nbytes, out, sink, 2), run_epoch<13, 5, 64, 32>(d, nbytes, out, sink, 2));
            fflush(stdout);
        }
    }
    if (which == 0 || which == 5) {
        // is "good" / "bad" a property of the output allocation alone, or of the pair (input, output)?
        unsigned char* d2 = nullptr;
        if (hipMalloc(&d2, nbytes) != hipSuccess) { printf("second input alloc failed\n"); return 1; }
        (void)hipMemset(d2, 0x33, nbytes);
        (void)hipDeviceSynchronize();
        printf("second input at %p.  output = base of a fresh allocation; plain per-tile store with input 1 | input 2 | input 1, second half of the input only\n", (void*)d2);
        const size_t sizes_mib[] = {248, 256, 512, 248, 1024, 248, 300, 400, 248, 2048, 248};
        for (size_t mib : sizes_mib) {
            void* a = nullptr;
            if (hipMalloc(&a, mib << 20) != hipSuccess) { printf("alloc of %zu MiB failed\n", mib); break; }
            float* out = (float*)a;
            printf("   %5zu MiB at %p: %6.0f | %6.0f | %6.0f GB/s\n", mib, a, run<W_TILE, 5, 16>(d, nbytes, out, sink, 2),
                   run<W_TILE, 5, 16>(d2, nbytes, out, sink, 2), run<W_TILE, 5, 16>(d + nbytes / 2, nbytes / 2, out, sink, 2));
            fflush(stdout);
        }
    }
    return 0;
}
