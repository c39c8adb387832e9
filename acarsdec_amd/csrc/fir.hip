// fir.hip -- the RTL down-converter on gfx950: u8 I/Q -> (x-127.37) -> per channel
// D = sum_k vb[k]*wf[k] over a window of M input samples -> |D|   (rtl.c:332-354).
//
// Shape of the problem: each input byte is used once per channel (taps == decimation), 8 flop per
// 2 bytes -> HBM-read bound, no data reuse, no MFMA (1-D convolution).  Design:
//   * a workgroup (4 waves) owns a contiguous run of 64-window tiles of ONE channel and walks it
//     in time (persistent over its segment);
//   * a tile (64 windows x 2M bytes, contiguous in HBM) is fetched with 16-byte coalesced loads,
//     staged through registers into LDS rows whose stride is an odd number of 16-byte slots, so
//     that "lane = window" ds_read_b128 reads are bank-conflict free;
//   * the loads of tile t+1 are issued before tile t is computed and only waited for at the LDS
//     write (issue-early / write-late), so HBM latency hides under the VALU work;
//   * inside a wave every lane is at the same tap index, so taps are wave-uniform: they come in
//     through scalar loads (SGPRs), not VGPRs or LDS;
//   * the 4 waves split the tap range; partial sums meet in LDS; wave 0 takes |D| and writes 64
//     consecutive floats of dm.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <cstdio>
#include <stdlib.h>
#include <type_traits>
#include "acg_internal.h"

typedef float f2 __attribute__((ext_vector_type(2)));

#define FIR_MAXLD 10   // 64 rows * 640 B (M=320) / 16 B / 256 threads
#define FIR_RUN 4      // tiles per dynamically scheduled run (>= 2)

extern __shared__ __attribute__((aligned(16))) unsigned char fir_smem[];

// The run dispenser is an atomic whose result is needed a whole tile later.  Written as a compiler
// intrinsic it is waited for on the spot (it sits in divergent control flow), which stalls the
// requesting wave -- and at the next barrier the whole workgroup -- for a full device-scope round
// trip per run.  As inline asm the compiler does not track it; take_ticket() is the matching wait
// (vmcnt retires in order, and it is called where no newer load is outstanding).
__device__ __forceinline__ void request_ticket(unsigned int* counter, unsigned int& ticket)
{
    const unsigned int one = 1u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(counter), "v"(one) : "memory");
}
__device__ __forceinline__ unsigned int take_ticket(unsigned int& ticket)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) : : "memory");
    return ticket;
}
// The dispenser is two words, {tickets handed out, workgroups finished}, both zero between launches:
// run ids are gridDim.x + ticket (the first run of a workgroup is its own id), and the last workgroup
// to leave re-arms the pair, so no memset launch precedes the kernel.  A workgroup's own request has
// been performed before it signs off (take_ticket), so no ticket atomic can land after the reset.
__device__ __forceinline__ void dispenser_sign_off(unsigned int* counter, unsigned int& pending)
{
    (void)take_ticket(pending);
    const unsigned int d = atomicAdd(counter + 1, 1u);
    if (d == gridDim.x - 1) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


__device__ __forceinline__ float cabs_like_glibc(float re, float im)
{
    // glibc 2.35 cabsf/hypotf == (float)sqrt((double)x*x + (double)y*y) (checked on 2e8 inputs);
    // products are exact in double, one rounding in the sum, correctly rounded sqrt, one narrowing.
    double d = (double)re * (double)re + (double)im * (double)im;
    return (float)__dsqrt_rn(d);
}

#ifdef ACG_LAB   // round 1's first kernel (ACG_FIR_VARIANT=0): lab build only (libacarsdec_amd_lab.so)
#include "lab/fir_tile.inc"
#endif  // ACG_LAB

// Persistent variant: the grid is exactly the number of workgroups the chip holds at once and the
// (channel, tile) space is cut into equal contiguous runs, one per workgroup, so that all workgroups
// finish together (no partial last wave of workgroups).  A run may cross channel boundaries; the
// stream/tap base pointers follow.  NT selects non-temporal loads for the input, which is read once.
template <bool NT, bool DYN, bool FULL>
__global__ __launch_bounds__(ACG_WG_FIR) void fir_u8_persist_kernel(const FirArgs a,
                                                                     const uint8_t* __restrict__ iq_base,
                                                                     const float* __restrict__ taps_base,
                                                                     const int* __restrict__ stream_of,
                                                                     float* __restrict__ dm_base)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (a.nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a.nch * ntile;
    // static: equal contiguous runs.  dynamic: runs of FIR_RUN tiles handed out by an atomic counter
    // (zeroed by the launcher), so that workgroups slowed by co-running demodulator waves simply
    // take fewer runs instead of stretching the kernel's tail.
    const long long nrun = (G + FIR_RUN - 1) / FIR_RUN;
    long long run = blockIdx.x;                       // first run: own id (the counter starts at gridDim.x)
    long long g0 = DYN ? run * FIR_RUN : G * blockIdx.x / gridDim.x;
    long long g1 = DYN ? (g0 + FIR_RUN < G ? g0 + FIR_RUN : G) : G * (blockIdx.x + 1) / gridDim.x;
    if (g0 >= g1) return;
    int* s_next = (int*)(fir_smem + ACG_TILE_WIN * a.row_stride + 4 * 64 * sizeof(float4));

    const int cpr = a.cpr;
    const int tile_chunks = ACG_TILE_WIN * cpr;
    const int total_chunks = a.nwin * cpr;
    const int pad = a.row_stride - a.row_bytes;
    const unsigned int magic = a.cpr_magic;
    unsigned char* tileL = fir_smem;
    float4* red = (float4*)(fir_smem + ACG_TILE_WIN * a.row_stride);
    const int nck = a.ntaps_pad >> 3;
    const int c0 = nck * wave / 4;
    const int c1 = nck * (wave + 1) / 4;

    int ch = (int)(g0 / ntile);
    int t = (int)(g0 - (long long)ch * ntile);
    uint4 stage[FIR_MAXLD];

    // A tile is 64*cpr chunks and a wave owns 64 consecutive chunks per pass, so "chunk inside the tile"
    // is a whole-wave property (wave + 4*i < cpr): scalar branches, no per-lane predicates, and the
    // address is one uniform base per pass + a per-thread constant offset (global_load ... saddr form:
    // no vector address arithmetic at all).  FULL = every tile lies completely inside the row
    // (nwin % 64 == 0, always true for whole callbacks), which removes the last per-lane bound.
    const unsigned int voff = (unsigned int)tid << 4;
    const size_t tile_bytes = (size_t)tile_chunks << 4;
    // the row base follows the channel; looked up only when the channel changes (a dependent scalar
    // load in front of every tile's loads would leave the workgroup without loads in flight meanwhile)
    int row_ch = -1;
    const uint8_t* __restrict__ row_base = iq_base;
    auto fetch = [&](int fch, int ft) {
        if (fch != row_ch) {
            row_base = iq_base + (size_t)stream_of[fch] * a.pitch;
            row_ch = fch;
        }
        const uint8_t* __restrict__ tb = row_base + (size_t)ft * tile_bytes;
        const int base = ft * tile_chunks;
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < FIR_MAXLD; ++i) {
            if (wave + 4 * i < cpr) {
                if (FULL || base + tid + i * ACG_WG_FIR < total_chunks) {
                    const u4v* p = (const u4v*)(tb + (size_t)i * (ACG_WG_FIR * 16) + (size_t)voff);
                    const u4v x = NT ? __builtin_nontemporal_load(p) : *p;
                    stage[i] = make_uint4(x.x, x.y, x.z, x.w);
                } else {
                    stage[i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    fetch(ch, t);
    unsigned int pending_next = 0;        // thread 0: id of the next run (valid one barrier after the request)
    for (long long g = g0;;) {
#pragma unroll
        for (int i = 0; i < FIR_MAXLD; ++i) {
            if (wave + 4 * i < cpr) {
                const int c = tid + i * ACG_WG_FIR;
                const int r = (int)(((unsigned int)c * magic) >> 20);
                *(uint4*)(tileL + (c << 4) + r * pad) = stage[i];
            }
        }
        if (DYN) {
            // first tile of a run: ask for the next run; the answer is published at this barrier
            // one tile later and read when the last tile of the run prefetches across the run boundary
            if (g == g0 && tid == 0) request_ticket(a.work_counter, pending_next);
            if (g == g0 + 1 && tid == 0) *s_next = (int)(gridDim.x + take_ticket(pending_next));
        }
        __syncthreads();

        // where does the stream go next?
        int nch_ = ch, nt_ = t + 1;
        if (nt_ == ntile) { nt_ = 0; ++nch_; }
        bool more = g + 1 < g1;
        long long ng0 = g0, ng1 = g1, ng = g + 1;
        if (DYN && !more) {
            // (runs are >= 2 tiles except possibly the very last one, which has no successor anyway)
            const long long nr = (g1 - g0 >= 2) ? (long long)*s_next : nrun;
            if (nr < nrun) {
                ng0 = nr * FIR_RUN;
                ng1 = ng0 + FIR_RUN < G ? ng0 + FIR_RUN : G;
                ng = ng0;
                nch_ = (int)((unsigned int)ng0 / (unsigned int)ntile);      // G < 2^31 (launcher)
                nt_ = (int)((unsigned int)ng0 - (unsigned int)nch_ * (unsigned int)ntile);
                more = true;
            }
        }
        if (more) fetch(nch_, nt_);

        const float* __restrict__ taps = taps_base + (size_t)ch * a.ntaps_pad * 2;
        f2 accA = {0.f, 0.f};
        f2 accB = {0.f, 0.f};
        const unsigned char* rowp = tileL + lane * a.row_stride;
        for (int c = c0; c < c1; ++c) {
            const uint4 q = *(const uint4*)(rowp + (c << 4));
            const float* __restrict__ w = taps + (c << 4);
            const unsigned int qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned int word = qq[j >> 1];
                const unsigned int sh = (j & 1) * 16;
                f2 tt;
                tt.x = (float)((word >> sh) & 0xffu) - 127.37f;
                tt.y = (float)((word >> (sh + 8)) & 0xffu) - 127.37f;
                const f2 wv = {w[2 * j], w[2 * j + 1]};
                const f2 ws = {w[2 * j + 1], w[2 * j]};
                accA = __builtin_elementwise_fma(tt, wv, accA);
                accB = __builtin_elementwise_fma(tt, ws, accB);
            }
        }
        red[wave * 64 + lane] = make_float4(accA.x, accA.y, accB.x, accB.y);
        __syncthreads();

        if (wave == 0) {
            const float4 r0 = red[lane], r1 = red[64 + lane], r2 = red[128 + lane], r3 = red[192 + lane];
            const float Dr = ((r0.x + r1.x) + (r2.x + r3.x)) - ((r0.y + r1.y) + (r2.y + r3.y));
            const float Di = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r0.w + r1.w) + (r2.w + r3.w));
            const int m = t * ACG_TILE_WIN + lane;
            if (m < a.nwin) dm_base[(size_t)ch * a.dm_pitch + m] = cabs_like_glibc(Dr, Di);             // rtl.c:353
        }
        if (!more) break;
        ch = nch_;
        t = nt_;
        g = ng;
        g0 = ng0;
        g1 = ng1;
    }
    if (DYN && tid == 0) dispenser_sign_off(a.work_counter, pending_next);
}

// Shared-stream variant -- rtl.c's own shape: one dongle's stream feeds several channels (up to 16,
// acarsdec.h:30).  A work unit is (group of <= FIR_KC channels of ONE stream, tile): the tile is loaded
// and converted u8 -> f32 once and every channel of the group takes its taps to the same registers.
// Per 8 complex samples: 1 ds_read_b128 + 24 conversion ops shared, 16 v_pk_fma_f32 per channel, so the
// kernel is VALU-bound at (2 + 3/K) lane-ops per channel-sample instead of HBM/fabric-bound at 5.
// The arithmetic per channel (tap split over the 4 waves, reduction order) is that of
// fir_u8_persist_kernel: both give bit-identical dm.  Partial sums meet in LDS four channels at a
// time; wave w finishes channel 4r + w.  Dynamic run dispenser as above.
// Taps come from a regrouped copy of the tap table, [group][8-sample step][channel of the group][16 floats]
// (regroup_taps_kernel), so that one scalar base + immediate offsets reaches all channels of a step.
#define FIR_KC 8
__global__ void regroup_taps_kernel(const float* __restrict__ taps, float* __restrict__ gtaps,
                                    const int4* __restrict__ groups, const int* __restrict__ group_ch, int ntaps_pad)
{
    const int4 gi = groups[blockIdx.x];
    const int kc = gi.z;
    const int n = kc * ntaps_pad * 2;
    float* out = gtaps + (size_t)gi.y * ntaps_pad * 2;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const int e = idx & 15;
        const int q = idx >> 4;
        const int c = q / kc;
        const int k = q - c * kc;
        out[idx] = taps[(size_t)group_ch[gi.y + k] * ntaps_pad * 2 + (c << 4) + e];
    }
}

template <bool FULL>
__global__ __launch_bounds__(ACG_WG_FIR) void fir_u8_shared_kernel(const FirArgs a,
                                                                    const uint8_t* __restrict__ iq_base,
                                                                    const float* __restrict__ taps_base,
                                                                    const int4* __restrict__ groups,
                                                                    const int* __restrict__ group_ch,
                                                                    float* __restrict__ dm_base)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (a.nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a.ngroups * ntile;
    const long long nrun = (G + FIR_RUN - 1) / FIR_RUN;
    long long g0 = (long long)blockIdx.x * FIR_RUN;
    long long g1 = g0 + FIR_RUN < G ? g0 + FIR_RUN : G;
    if (g0 >= g1) return;
    int* s_next = (int*)(fir_smem + ACG_TILE_WIN * a.row_stride + 16 * 64 * sizeof(float4));

    const int cpr = a.cpr;
    const int tile_chunks = ACG_TILE_WIN * cpr;
    const int total_chunks = a.nwin * cpr;
    const int pad = a.row_stride - a.row_bytes;
    const unsigned int magic = a.cpr_magic;
    unsigned char* tileL = fir_smem;
    float4* red = (float4*)(fir_smem + ACG_TILE_WIN * a.row_stride);      // [4 channels][4 waves][64]
    const int nck = a.ntaps_pad >> 3;
    const int c0 = nck * wave / 4;
    const int c1 = nck * (wave + 1) / 4;

    int grp = (int)(g0 / ntile);
    int t = (int)(g0 - (long long)grp * ntile);
    uint4 stage[FIR_MAXLD];

    // wave-granular tile loads from one uniform base per pass, as in fir_u8_persist_kernel
    const unsigned int voff = (unsigned int)tid << 4;
    const size_t tile_bytes = (size_t)tile_chunks << 4;
    int row_grp = -1;
    const uint8_t* __restrict__ row_base = iq_base;
    auto fetch = [&](int fgrp, int ft) {
        if (fgrp != row_grp) {
            row_base = iq_base + (size_t)groups[fgrp].x * a.pitch;
            row_grp = fgrp;
        }
        const uint8_t* __restrict__ tb = row_base + (size_t)ft * tile_bytes;
        const int base = ft * tile_chunks;
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < FIR_MAXLD; ++i) {
            if (wave + 4 * i < cpr) {
                if (FULL || base + tid + i * ACG_WG_FIR < total_chunks) {
                    const u4v x = __builtin_nontemporal_load((const u4v*)(tb + (size_t)i * (ACG_WG_FIR * 16) + (size_t)voff));
                    stage[i] = make_uint4(x.x, x.y, x.z, x.w);
                } else {
                    stage[i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    fetch(grp, t);
    unsigned int pending_next = 0;
    for (long long g = g0;;) {
#pragma unroll
        for (int i = 0; i < FIR_MAXLD; ++i) {
            if (wave + 4 * i < cpr) {
                const int c = tid + i * ACG_WG_FIR;
                const int r = (int)(((unsigned int)c * magic) >> 20);
                *(uint4*)(tileL + (c << 4) + r * pad) = stage[i];
            }
        }
        if (g == g0 && tid == 0) request_ticket(a.work_counter, pending_next);
        if (g == g0 + 1 && tid == 0) *s_next = (int)(gridDim.x + take_ticket(pending_next));
        __syncthreads();

        int ngrp = grp, nt_ = t + 1;
        if (nt_ == ntile) { nt_ = 0; ++ngrp; }
        bool more = g + 1 < g1;
        long long ng0 = g0, ng1 = g1, ng = g + 1;
        if (!more) {
            const long long nr = (g1 - g0 >= 2) ? (long long)*s_next : nrun;
            if (nr < nrun) {
                ng0 = nr * FIR_RUN;
                ng1 = ng0 + FIR_RUN < G ? ng0 + FIR_RUN : G;
                ng = ng0;
                ngrp = (int)((unsigned int)ng0 / (unsigned int)ntile);      // G < 2^31 (launcher)
                nt_ = (int)((unsigned int)ng0 - (unsigned int)ngrp * (unsigned int)ntile);
                more = true;
            }
        }
        if (more) fetch(ngrp, nt_);

        const int4 gi = groups[grp];                                        // wave-uniform
        const int kc = gi.z;
        const float* __restrict__ gt = taps_base + (size_t)gi.y * a.ntaps_pad * 2;
        f2 accA[FIR_KC], accB[FIR_KC];
#pragma unroll
        for (int k = 0; k < FIR_KC; ++k) {
            accA[k] = {0.f, 0.f};
            accB[k] = {0.f, 0.f};
        }
        const unsigned char* rowp = tileL + lane * a.row_stride;
        auto taps_pass = [&](auto full) {
            constexpr bool ALLK = decltype(full)::value;
            const int kcc = ALLK ? FIR_KC : kc;
            for (int c = c0; c < c1; ++c) {
                const float* __restrict__ wc = gt + (size_t)(c * kcc) * 16;
                const uint4 q = *(const uint4*)(rowp + (c << 4));
                const unsigned int qq[4] = {q.x, q.y, q.z, q.w};
                f2 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned int word = qq[j >> 1];
                    const unsigned int sh = (j & 1) * 16;
                    x[j].x = (float)((word >> sh) & 0xffu) - 127.37f;
                    x[j].y = (float)((word >> (sh + 8)) & 0xffu) - 127.37f;
                }
#pragma unroll
                for (int k = 0; k < FIR_KC; ++k) {
                    if (ALLK || k < kc) {
                        const float* __restrict__ w = wc + k * 16;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const f2 wv = {w[2 * j], w[2 * j + 1]};
                            const f2 ws = {w[2 * j + 1], w[2 * j]};
                            accA[k] = __builtin_elementwise_fma(x[j], wv, accA[k]);
                            accB[k] = __builtin_elementwise_fma(x[j], ws, accB[k]);
                        }
                    }
                }
            }
        };
        if (kc == FIR_KC) taps_pass(std::true_type{});
        else taps_pass(std::false_type{});

        const int m = t * ACG_TILE_WIN + lane;
#pragma unroll
        for (int r = 0; r < FIR_KC / 4; ++r) {
            if (4 * r < kc) {                                               // uniform
                if (r > 0) __syncthreads();                                 // round r-1's reads are done
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    red[(kk * 4 + wave) * 64 + lane] = make_float4(accA[4 * r + kk].x, accA[4 * r + kk].y,
                                                                   accB[4 * r + kk].x, accB[4 * r + kk].y);
                __syncthreads();
                if (4 * r + wave < kc) {
                    const float4* rr = red + (wave * 4) * 64 + lane;
                    const float4 r0 = rr[0], r1 = rr[64], r2 = rr[128], r3 = rr[192];
                    const float Dr = ((r0.x + r1.x) + (r2.x + r3.x)) - ((r0.y + r1.y) + (r2.y + r3.y));
                    const float Di = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r0.w + r1.w) + (r2.w + r3.w));
                    const int mych = group_ch[gi.y + 4 * r + wave];          // wave-uniform
                    if (m < a.nwin) dm_base[(size_t)mych * a.dm_pitch + m] = cabs_like_glibc(Dr, Di);
                }
            }
        }
        if (!more) break;
        grp = ngrp;
        t = nt_;
        g = ng;
        g0 = ng0;
        g1 = ng1;
    }
    if (tid == 0) dispenser_sign_off(a.work_counter, pending_next);
}

#ifdef ACG_LAB   // LDS-DMA variant (ACG_FIR_VARIANT=4): lab build only
#include "lab/fir_dma.inc"
#endif  // ACG_LAB

// ---------------------------------------------------------------------------------------------------
// Wave-private streaming variant -- the default for whole callbacks at the documented rates
// (rtlMult 160 / 192 / 200, rtl.c:244-262).  What the workgroup-granular kernels above pay for is
// structure, not arithmetic: two barriers per tile, a partial-sum exchange between the four
// waves and a single finishing wave.  Here a wave owns its tiles outright and never talks to another wave:
//   * the input never passes through LDS: a tile (64 windows x 2M bytes = CPR KiB, contiguous in HBM) is
//     read as CPR wave-loads of 1 KiB (16 bytes per lane, non-temporal, one uniform base + lane offset),
//     U loads always in flight per wave, across tile and run boundaries -- the access pattern of a pure
//     streaming reader;
//   * lane L of wave-load q holds 16-byte chunk i = 64 q + L = 8 consecutive complex samples of window
//     i / CPR at column i % CPR.  The taps of that column come from a per-wave LDS copy of the channel's
//     tap table, laid out [4 planes][CPR + 63 columns] with the columns replicated past CPR, so that
//     the lane address is one base + a compile-time offset per q (no per-lane modulo) and the four
//     ds_read_b128 of a step are bank-conflict free;
//   * the 8-sample partial sum (re, im) of every chunk goes to a per-wave LDS array; after the CPR-th load
//     lane = window adds the CPR partials of its window (row stride odd: conflict free), takes |D| and
//     the wave writes 64 consecutive floats of dm.  All 64 lanes of every wave do this (no finishing wave);
//   * work is handed out per WAVE in runs of 2 tiles (one channel) by a sharded ticket dispenser
//     (ACG_DISP_SHARDS counters, each serving a contiguous eighth of the run space in address order;
//     a wave whose shard is exhausted moves on to the next one).  The ticket for the next run is
//     requested a whole tile before it is needed; the next run's taps are fetched during the last tile.
// Summation order: 8 sequential fused multiply-adds per chunk, then CPR sequential adds (the reference
// adds M terms sequentially; its own -Ofast build reassociates, SURVEY 8c: dm within 1e-5 relative).
#define FIRD_R 2

template <int CPR, int U_ = 0, int B_ = 0>
struct FirD {
    // U staging slots per lane: run position p lives in slot p % U, so U divides the loads of a run.  Loads are
    // issued in bursts of B consecutive wave-loads (B KiB contiguous per wave) every B steps, as the slots of the
    // B positions consumed since the last burst are free again: between U - B and U KiB per wave are in flight.
    static constexpr int U = U_ ? U_ : ((FIRD_R * CPR) % 10 == 0) ? 10 : 12;
    static constexpr int B = B_ ? B_ : U / 2;
    static constexpr int TS = CPR + 63;                                        // tap columns incl. replicas
    static constexpr int PW = (CPR & 1) ? CPR : CPR + 1;                       // partial-sum row stride (odd)
    static constexpr int TAB_BYTES = 4 * TS * 16;
    static constexpr int P_BYTES = 64 * PW * 8;
    static constexpr int WAVE_LDS = TAB_BYTES + P_BYTES;
    static_assert((FIRD_R * CPR) % U == 0 && (FIRD_R * CPR) % B == 0 && B <= U, "slots and bursts must divide the loads per run");
};

typedef unsigned int u4v_t __attribute__((ext_vector_type(4)));

// Raw buffer descriptor over [base, base + bytes): loads beyond `bytes` return 0 without touching memory,
// so "no next run" is a descriptor of 0 bytes instead of a branch around every prefetch.  Loads through a
// descriptor are `buffer_load_dwordx4 v, voff, s[rsrc], soffset offen nt`: one VGPR offset (lane * 16), the
// tile / load position as scalar offset, no vector address arithmetic, the non-temporal bit explicit.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fird_rsrc(const void* base, unsigned int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u4v_t fird_load(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2 /* nt */);
}

// The ticket for the run after the current one is requested by lane 0 a whole tile before it is looked at.
// As a compiler builtin the atomic is waited for on the spot (the uniform-address atomic optimiser wants
// its result at once); as inline asm the compiler does not know about it.  Returns are in issue order, so
// once at most `younger` vector-memory operations are outstanding the (older) atomic has returned:
// ticket_take waits for exactly that and costs nothing when U later loads have long been consumed.
// tests/test_host_logic.py disassembles the kernel and checks that nothing touches the ticket register
// between the two statements.
__device__ __forceinline__ void ticket_request(unsigned int* counter, unsigned int& ticket)
{
    const unsigned int one = 1u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(counter), "v"(one) : "memory");
}
template <int YOUNGER>
__device__ __forceinline__ unsigned int ticket_take(unsigned int& ticket)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ticket) : "i"(YOUNGER) : "memory");
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
}

// one tile: CPR steps.  A step is fenced off from its neighbours with scheduling barriers -- left alone, the scheduler
// delays the loads to shorten live ranges (the pipeline then runs dry at the end of a tile) or bunches them up at
// random, and the result changes with every edit.  Inside a step: first the memory instructions whose results are
// needed LATER (every B-th step the burst of B wave-loads U - B positions ahead, the next step's taps from LDS), then
// this step's arithmetic on data that arrived long ago.
// FOLD: sum (x - 127.37)(w) = sum x w - 127.37 (1 + j) sum w.  The second term is a constant of the channel (dc, formed
// when the tap table is loaded), so the eight v_pk_add_f32 per 16 input bytes that subtract 127.37 from every sample
// (a fifth of the loop's vector instructions) become one subtraction per window.  x - 127.37f is exact in f32 either
// way; what changes is the order of the roundings in the sum, at the 1e-7 level of full scale like any re-association
// (rtl.c builds with -Ofast and re-associates itself; the parity bar for dm is 1e-5).
template <int CPR, int UU, int BB, int TILE, bool FOLD, bool WT = false, bool NOMAC = false>
__device__ __forceinline__ void fird_tile(u4v_t* st /* [U] */, __amdgpu_buffer_rsrc_t cur, __amdgpu_buffer_rsrc_t nxt,
                                          unsigned int voff, const float4* Tl, f2* Pw, const f2* Pr, float* __restrict__ dm_out, int lane,
                                          f2 dc)
{
    typedef FirD<CPR, UU, BB> F;
    float4 w0 = Tl[0 * F::TS], w1 = Tl[1 * F::TS], w2 = Tl[2 * F::TS], w3 = Tl[3 * F::TS];          // step 0: column of lane 0 is 0
#pragma unroll
    for (int q = 0; q < CPR; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        const int p = TILE * CPR + q;                         // run position consumed by this step
        const u4v_t d = st[p % F::U];
        if (p % F::B == 0) {
#pragma unroll
            for (int b = 0; b < F::B; ++b) {
                const int pos = p + F::U - F::B + b;          // slots of positions p - B .. p - 1 are free
                if (pos < FIRD_R * CPR) st[pos % F::U] = fird_load(cur, voff, (unsigned int)pos * 1024u);
                else st[pos % F::U] = fird_load(nxt, voff, (unsigned int)(pos - FIRD_R * CPR) * 1024u);
            }
        }
        float4 n0 = w0, n1 = w1, n2 = w2, n3 = w3;
        if (q + 1 < CPR) {
            const int c1 = ((q + 1) * 64) % CPR;              // column of lane 0 in the next load (compile time)
            n0 = Tl[0 * F::TS + c1]; n1 = Tl[1 * F::TS + c1]; n2 = Tl[2 * F::TS + c1]; n3 = Tl[3 * F::TS + c1];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
        f2 accA = {0.f, 0.f};       // (sum tr*wr, sum ti*wi)
        f2 accB = {0.f, 0.f};       // (sum tr*wi, sum ti*wr)
        if (NOMAC) {                // measurement variant 56: the loads are consumed, nothing is converted or multiplied, no tap reads
            accA.x = __uint_as_float((d[0] ^ d[1]) & 0x3fffffffu);
            accB.x = __uint_as_float((d[2] ^ d[3]) & 0x3fffffffu);
        } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int word = d[j >> 1];
            const unsigned int sh = (j & 1) * 16;
            f2 tt;
            tt.x = (float)((word >> sh) & 0xffu);                      // rtl.c:338 (the conversion and the difference are exact in f32)
            tt.y = (float)((word >> (sh + 8)) & 0xffu);                // rtl.c:339
            if (!FOLD) { tt.x -= 127.37f; tt.y -= 127.37f; }
            const f2 wv = {w[2 * j], w[2 * j + 1]};
            const f2 ws = {w[2 * j + 1], w[2 * j]};
            accA = __builtin_elementwise_fma(tt, wv, accA);
            accB = __builtin_elementwise_fma(tt, ws, accB);
        }
        const f2 part = {accA.x - accA.y, accB.x + accB.y};
        if (F::PW == CPR) {
            Pw[q * 64] = part;                                         // chunk i = 64 q + lane, rows of CPR
        } else {
            const unsigned int i = (unsigned int)(q * 64 + lane);
            const unsigned int r = (i * ((65536u + CPR - 1) / CPR)) >> 16;   // i / CPR for i < 64 * CPR <= 2048
            Pw[q * 64 + (int)r] = part;                                // + one pad entry per completed row
        }
        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
    __builtin_amdgcn_sched_barrier(0);
    // lane = window: add the CPR partial sums of the row, |D| (rtl.c:353), 64 consecutive floats
    f2 D = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < CPR; ++j) D = D + Pr[j];
    if (FOLD) D = D - dc;
    if (WT) {                                                      // write-through store (the default; 55 = write-back): see firc_flush
        float* p = dm_out + lane;
        const float v = cabs_like_glibc(D.x, D.y);
        asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    } else {
        dm_out[lane] = cabs_like_glibc(D.x, D.y);
    }
}

template <int CPR, int UU, int BB, bool FOLD, bool WT = false, bool NOMAC = false>
__device__ __forceinline__ void fird_body(const FirArgs& a, const uint8_t* __restrict__ iq_base, const float* __restrict__ taps_base,
                                          const int* __restrict__ stream_of, float* __restrict__ dm_base)
{
    typedef FirD<CPR, UU, BB> F;
    constexpr unsigned int NONE = 0xffffffffu;
    // beside demodulator waves (which have slack whenever this kernel is the longer stage) this stream wins the issue
    // arbitration: the demodulator then fills the gaps instead of stretching the bandwidth-bound stage
    if (a.high_prio) __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* my = fir_smem + wave * F::WAVE_LDS;
    float4* T = (float4*)my;
    f2* P = (f2*)(my + F::TAB_BYTES);
    const float4* Tl = T + lane;                                   // + (plane * TS + c0(q)) compile-time slots
    f2* Pw = P + lane;                                             // + 64 q entries
    const f2* Pr = P + lane * F::PW;

    // A run is `pairs` consecutive two-tile bodies of one channel (launcher: 1 for small launches, up to 8 for large
    // ones, so that every wave still gets ~32 runs): the per-run work (ticket, row lookup, tap table) is paid once per run.
    const unsigned int pairs = (unsigned int)a.run_pairs;
    const unsigned int ntile = (unsigned int)a.nwin / ACG_TILE_WIN;                 // whole tiles only (launcher)
    const unsigned int runs_per_ch = ntile / (FIRD_R * pairs);                      // ntile % (FIRD_R * pairs) == 0 (launcher)
    const unsigned int nrun = (unsigned int)a.nch * runs_per_ch;
    const unsigned int wpg = blockDim.x >> 6;                                       // the waves of a workgroup never talk to each other
    const unsigned int nwaves = gridDim.x * wpg;
    const unsigned int wg = blockIdx.x * wpg + (unsigned int)wave;
    unsigned int* ctr = a.work_counter;
    constexpr unsigned int tile_bytes = (unsigned int)CPR * 1024u;
    constexpr unsigned int run_bytes = FIRD_R * tile_bytes;                         // bytes of one two-tile body
    const unsigned int voff = (unsigned int)lane << 4;
    const int nck = a.ntaps_pad >> 3;                                               // tap columns that carry taps

    // Shard s hands out runs s, s + SHARDS, s + 2 SHARDS, ...: the shards advance together, so the whole grid reads
    // ONE front that moves through the input in address order (HBM likes that better than eight distant fronts),
    // while the ticket atomics are spread over eight L2 channels.  The first run of wave wg is run wg (static);
    // tickets count on from there.
    auto shard_runs = [&](unsigned int s) { return (nrun + ACG_DISP_SHARDS - 1 - s) / ACG_DISP_SHARDS; };
    auto shard_static = [&](unsigned int s) { return (nwaves + ACG_DISP_SHARDS - 1 - s) / ACG_DISP_SHARDS; };
    auto run_of_ticket = [&](unsigned int s, unsigned int t) -> unsigned int {
        const unsigned long long k = (unsigned long long)shard_static(s) + t;
        return k < shard_runs(s) ? (unsigned int)k * ACG_DISP_SHARDS + s : NONE;
    };
    auto probe = [&](unsigned int& s) -> unsigned int {             // synchronous: next run of shard s, else of the following shards
        for (int k = 0; k < ACG_DISP_SHARDS; ++k) {
            unsigned int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ctr + s * ACG_DISP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            const unsigned int r = run_of_ticket(s, t);
            if (r != NONE) return r;
            s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
        }
        return NONE;
    };
    auto sign_off = [&]() {
        if (lane == 0) {
            const unsigned int d = atomicAdd(ctr + ACG_DISP_SHARDS * ACG_DISP_STRIDE, 1u);
            if (d == nwaves - 1) {                                  // every wave's requests have been answered: re-arm
#pragma unroll
                for (int k = 0; k <= ACG_DISP_SHARDS; ++k)
                    __hip_atomic_store(ctr + k * ACG_DISP_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto run_base = [&](unsigned int run, unsigned int& ch, unsigned int& t0) -> const uint8_t* {
        ch = run / runs_per_ch;
        t0 = (run - ch * runs_per_ch) * FIRD_R * pairs;
        const size_t row = (size_t)(a.stream_identity ? (int)ch : stream_of[ch]) * a.pitch;
        return iq_base + row + (size_t)t0 * tile_bytes;
    };
    // taps of channel ch, column = lane (64 bytes); the values are not looked at before write_taps, so the
    // loads stay in flight under the last tile of the run
    auto fetch_taps = [&](unsigned int ch, float4 (&tp)[4]) {
        const float4* src = (const float4*)(taps_base + (size_t)ch * a.ntaps_pad * 2) + (lane < nck ? lane : 0) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) tp[k] = src[k];
    };
    f2 dc = {0.f, 0.f};                                             // 127.37 (1 + j) sum w of the current run's channel
    auto write_taps = [&](const float4 (&tp)[4]) {
        const bool on = lane < nck;                                 // columns beyond the last tap: zero
        if (FOLD) {
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sr += tp[k].x + tp[k].z; si += tp[k].y + tp[k].w; }
            sr = on ? sr : 0.f;
            si = on ? si : 0.f;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { sr += __shfl_xor(sr, m, 64); si += __shfl_xor(si, m, 64); }
            dc.x = 127.37f * (sr - si);
            dc.y = 127.37f * (sr + si);
        }
#pragma unroll
        for (int rep = 0; rep * CPR < F::TS; ++rep) {
            const int u = lane + rep * CPR;
            if (lane < CPR && u < F::TS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) T[k * F::TS + u] = on ? tp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    unsigned int s = wg % ACG_DISP_SHARDS;
    unsigned int run = wg < nrun ? wg : NONE;
    if (run == NONE) {                                              // tiny launches: fewer runs than waves
        s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
        run = probe(s);
    }
    if (run == NONE) { sign_off(); return; }

    unsigned int ch, t0;
    const uint8_t* base = run_base(run, ch, t0);
    {
        float4 tp[4];
        fetch_taps(ch, tp);
        write_taps(tp);
    }
    __amdgpu_buffer_rsrc_t cur = fird_rsrc(base, run_bytes);
    u4v_t st[F::U];
#pragma unroll
    for (int i = 0; i < F::U - F::B; ++i) {                         // step 0 issues the burst U - B .. U - 1 itself
        st[i] = fird_load(cur, voff, (unsigned int)i * 1024u);
        __builtin_amdgcn_sched_barrier(0);                          // keep the loads in issue order (the waits count on it)
    }

    for (;;) {
        // ask for the run after this one now; the answer is looked at in the run's last body, at least a tile later
        unsigned int tk;
        if (lane == 0) ticket_request(ctr + s * ACG_DISP_STRIDE, tk);
        float* __restrict__ dm_out = dm_base + (size_t)ch * a.dm_pitch + (size_t)t0 * ACG_TILE_WIN;
        static_assert(FIRD_R == 2, "a body is unrolled by hand: first tile, second tile");
        bool has_next = true;
        unsigned int nrun_ = NONE, nch_ = ch, nt0 = t0;
        float4 tp[4];
        for (unsigned int j = 0; j < pairs; ++j) {
            fird_tile<CPR, UU, BB, 0, FOLD, WT, NOMAC>(st, cur, cur, voff, Tl, Pw, Pr, dm_out, lane, dc);
            // second tile of the body: where does the stream go next?
            const uint8_t* nbase = base + run_bytes;                // the next body of this run ...
            if (j + 1 == pairs) {                                   // ... or the first body of the next run
                // (U loads and a store are in flight, all younger than the ticket)
                nrun_ = run_of_ticket(s, ticket_take<F::U + 1>(tk));
                if (nrun_ == NONE) {
                    s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
                    nrun_ = probe(s);
                }
                has_next = nrun_ != NONE;
                nbase = base;
                if (has_next) nbase = run_base(nrun_, nch_, nt0);
                fetch_taps(nch_, tp);
            }
            const __amdgpu_buffer_rsrc_t nxt = fird_rsrc(nbase, has_next ? run_bytes : 0u);
            fird_tile<CPR, UU, BB, 1, FOLD, WT, NOMAC>(st, cur, nxt, voff, Tl, Pw, Pr, dm_out + ACG_TILE_WIN, lane, dc);
            dm_out += FIRD_R * ACG_TILE_WIN;
            base = nbase;
            cur = nxt;
        }
        if (!has_next) break;
        write_taps(tp);
        run = nrun_;
        ch = nch_;
        t0 = nt0;
    }
    sign_off();
}

template <int CPR, int UU = 0, int BB = 0, bool FOLD = true, bool WT = false, bool NOMAC = false>
__global__ __launch_bounds__(ACG_WG_FIR) void fir_u8_direct_kernel(const FirArgs a,
                                                                    const uint8_t* __restrict__ iq_base,
                                                                    const float* __restrict__ taps_base,
                                                                    const int* __restrict__ stream_of,
                                                                    float* __restrict__ dm_base)
{
    fird_body<CPR, UU, BB, FOLD, WT, NOMAC>(a, iq_base, taps_base, stream_of, dm_base);
}

#ifdef ACG_LAB   // register-tap and matrix-pipe variants (ACG_FIR_VARIANT=7 / 8 / 6, 70..73): lab build only
#include "lab/fir_coltap_mfma.inc"
#endif  // ACG_LAB

// ---------------------------------------------------------------------------------------------------
// The other front ends' sample formats: same tile scheme, 4 bytes per input sample.
//   FMT_CS16   interleaved int16 I,Q        soapy.c:238-241   (x/32768 folded into the output scale)
//   FMT_SPLIT  int16 I plane + int16 Q plane sdrplay.c:219-223 (cabsf(D)/4 = output scale)
//   FMT_F32R   real float32 samples          air.c:314-324     (complex tap x real sample)
// A row (one window) is 4*M bytes in LDS; for FMT_SPLIT it is [I half | Q half]; windows longer than
// 52 chunks go through LDS in column slices.  Static contiguous partition of the (channel, tile)
// space, non-temporal loads, taps through scalar loads.
#define FMT_CS16 1
#define FMT_SPLIT 2
#define FMT_F32R 3
#define FMT_MAXLD 14   // 64 rows * 52 chunks (one slice) / 256 threads

template <int FMT>
__global__ __launch_bounds__(ACG_WG_FIR) void fir_fmt_kernel(const FirArgs a,
                                                              const uint8_t* __restrict__ in_base,
                                                              const float* __restrict__ taps_base,
                                                              const int* __restrict__ stream_of,
                                                              float* __restrict__ dm_base)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (a.nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a.nch * ntile;
    const long long g0 = G * blockIdx.x / gridDim.x;
    const long long g1 = G * (blockIdx.x + 1) / gridDim.x;
    if (g0 >= g1) return;

    // A window longer than 52 chunks (Airspy: 480 / 800 samples) passes through LDS in K column
    // slices of cpr chunks; the partial sums stay in registers across the slices of one tile.
    const int K = a.kseg;
    const int cpr = a.cpr;                              // 16-byte chunks per LDS row (one slice)
    const int half = cpr >> 1;                          // FMT_SPLIT: chunks per plane row (K == 1)
    const int tile_chunks = ACG_TILE_WIN * cpr;
    const int pad = a.row_stride - (cpr << 4);
    const unsigned int magic = a.cpr_magic;
    unsigned char* tileL = fir_smem;
    float4* red = (float4*)(fir_smem + ACG_TILE_WIN * a.row_stride);
    constexpr int SPC = (FMT == FMT_SPLIT) ? 8 : 4;     // samples (taps) per inner step
    const int nstep_total = a.ntaps_pad / SPC;
    const int nld = (tile_chunks + ACG_WG_FIR - 1) / ACG_WG_FIR;

    int ch = (int)(g0 / ntile);
    int t = (int)(g0 - (long long)ch * ntile);
    int k = 0;
    uint4 stage[FMT_MAXLD];

    auto fetch = [&](int fch, int ft, int fk) {
        const uint8_t* __restrict__ src = in_base + (size_t)stream_of[fch] * a.pitch;
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const int col0 = fk * cpr;
#pragma unroll
        for (int i = 0; i < FMT_MAXLD; ++i) {
            if (i < nld) {
                const int c = tid + i * ACG_WG_FIR;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < tile_chunks) {
                    const int r = (int)(((unsigned int)c * magic) >> 20);
                    const int col = c - r * cpr;
                    const int win = ft * ACG_TILE_WIN + r;
                    if (win < a.nwin && col0 + col < a.cpr_total) {
                        size_t off;
                        if (FMT == FMT_SPLIT)           // I plane, then Q plane a.plane bytes further
                            off = (col < half) ? (size_t)win * (a.row_bytes >> 1) + ((size_t)col << 4)
                                               : a.plane + (size_t)win * (a.row_bytes >> 1) + ((size_t)(col - half) << 4);
                        else
                            off = (size_t)win * a.row_bytes + ((size_t)(col0 + col) << 4);
                        const u4v x = __builtin_nontemporal_load((const u4v*)(src + off));
                        v = make_uint4(x.x, x.y, x.z, x.w);
                    }
                }
                stage[i] = v;
            }
        }
    };

    f2 accA = {0.f, 0.f};       // CS16/SPLIT: (sum r*wr, sum g*wi)   F32R: (sum s*wr, sum s*wi)
    f2 accB = {0.f, 0.f};       // CS16/SPLIT: (sum r*wi, sum g*wr)
    const long long u1 = (g1 - g0) * K;
    fetch(ch, t, 0);
    for (long long u = 0; u < u1; ++u) {
#pragma unroll
        for (int i = 0; i < FMT_MAXLD; ++i) {
            if (i < nld) {
                const int c = tid + i * ACG_WG_FIR;
                if (c < tile_chunks) {
                    const int r = (int)(((unsigned int)c * magic) >> 20);
                    *(uint4*)(tileL + (c << 4) + r * pad) = stage[i];
                }
            }
        }
        __syncthreads();
        int nch_ = ch, nt_ = t, nk_ = k + 1;
        if (nk_ == K) {
            nk_ = 0;
            if (++nt_ == ntile) { nt_ = 0; ++nch_; }
        }
        if (u + 1 < u1) fetch(nch_, nt_, nk_);

        // this slice's inner steps, split over the 4 waves
        const int nstep = (FMT == FMT_SPLIT) ? nstep_total : max(0, min(cpr, nstep_total - k * cpr));
        const int c0 = nstep * wave / 4;
        const int c1 = nstep * (wave + 1) / 4;
        const float* __restrict__ taps = taps_base + (size_t)ch * a.ntaps_pad * 2 + (size_t)k * cpr * SPC * 2;
        const unsigned char* rowp = tileL + lane * a.row_stride;
        for (int c = c0; c < c1; ++c) {
            const float* __restrict__ w = taps + c * SPC * 2;                  // wave-uniform
            if (FMT == FMT_CS16) {
                const uint4 q = *(const uint4*)(rowp + (c << 4));
                const unsigned int qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f2 tt;
                    tt.x = (float)(short)(qq[j] & 0xffffu);                    // soapy.c:238
                    tt.y = (float)((int)qq[j] >> 16);                          // soapy.c:239
                    const f2 wv = {w[2 * j], w[2 * j + 1]};
                    const f2 ws = {w[2 * j + 1], w[2 * j]};
                    accA = __builtin_elementwise_fma(tt, wv, accA);
                    accB = __builtin_elementwise_fma(tt, ws, accB);
                }
            } else if (FMT == FMT_SPLIT) {
                const uint4 qi = *(const uint4*)(rowp + (c << 4));
                const uint4 qq_ = *(const uint4*)(rowp + ((c + half) << 4));
                const unsigned int xi[4] = {qi.x, qi.y, qi.z, qi.w};
                const unsigned int xq[4] = {qq_.x, qq_.y, qq_.z, qq_.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f2 tt;
                    tt.x = (j & 1) ? (float)((int)xi[j >> 1] >> 16) : (float)(short)(xi[j >> 1] & 0xffffu);   // sdrplay.c:219
                    tt.y = (j & 1) ? (float)((int)xq[j >> 1] >> 16) : (float)(short)(xq[j >> 1] & 0xffffu);   // sdrplay.c:220
                    const f2 wv = {w[2 * j], w[2 * j + 1]};
                    const f2 ws = {w[2 * j + 1], w[2 * j]};
                    accA = __builtin_elementwise_fma(tt, wv, accA);
                    accB = __builtin_elementwise_fma(tt, ws, accB);
                }
            } else {
                const float4 q = *(const float4*)(rowp + (c << 4));
                const float sv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f2 tt = {sv[j], sv[j]};                              // air.c:315-316: wf[i] * S
                    const f2 wv = {w[2 * j], w[2 * j + 1]};
                    accA = __builtin_elementwise_fma(tt, wv, accA);
                }
            }
        }
        if (nk_ == 0) {                                 // last slice of this tile: reduce the 4 waves, emit |D|
            red[wave * 64 + lane] = make_float4(accA.x, accA.y, accB.x, accB.y);
            accA = {0.f, 0.f};
            accB = {0.f, 0.f};
            __syncthreads();
            if (wave == 0) {
                const float4 r0 = red[lane], r1 = red[64 + lane], r2 = red[128 + lane], r3 = red[192 + lane];
                float Dr, Di;
                if (FMT == FMT_F32R) {
                    Dr = (r0.x + r1.x) + (r2.x + r3.x);
                    Di = (r0.y + r1.y) + (r2.y + r3.y);
                } else {
                    Dr = ((r0.x + r1.x) + (r2.x + r3.x)) - ((r0.y + r1.y) + (r2.y + r3.y));
                    Di = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r0.w + r1.w) + (r2.w + r3.w));
                }
                const int m = t * ACG_TILE_WIN + lane;
                // power-of-two output scale (1/32768 soapy.c:241, 1/4 sdrplay.c:225): exact, commutes with cabsf
                if (m < a.nwin) dm_base[(size_t)ch * a.dm_pitch + m] = cabs_like_glibc(Dr, Di) * a.out_scale;
            }
        } else {
            __syncthreads();                            // the slice in LDS is overwritten next round
        }
        ch = nch_;
        t = nt_;
        k = nk_;
    }
}

// ---------------------------------------------------------------------------------------------------
// The wave-private streaming kernel for the other front ends' sample formats (4 bytes per sample; split planes: two
// 16-byte chunks -- 8 I and 8 Q samples -- per step).  Same structure as fir_u8_direct_kernel: the input never passes
// through LDS, wave-loads are issued in bursts U - B positions ahead, the taps of a lane's column come from a per-wave
// LDS copy of the channel's tap table with replicated columns, per-chunk partial sums are transposed through a per-wave
// LDS array, work is handed out per wave by the sharded ticket dispenser.  What the format changes:
//   * a window is CPR chunks per plane (M = 200 interleaved int16 or real f32: 50; Airspy at 6 / 10 Msps: 120 / 200),
//     so a tile is W = 32, 16 or 8 windows (W * CPR chunks = a whole number of wave-loads) and S = 64 / W adjacent lanes
//     add up the partial sums of one window -- E = CPR / S consecutive entries each -- and meet through S - 1 lane
//     exchanges;
//   * the arithmetic per chunk: 4 complex samples (CS16), 4 real samples against complex taps (F32R: half the
//     multiply-adds), 8 complex samples from two planes (SPLIT); the power-of-two output scale is applied to |D|.
template <int FMT, int CPR, int W>
struct FirX {
    static constexpr int NPL = (FMT == FMT_SPLIT) ? 4 : 2;             // 16-byte tap planes per column (8 or 4 taps)
    static constexpr int SPC = (FMT == FMT_SPLIT) ? 8 : 4;             // samples (taps) per column
    static constexpr int NLD = (FMT == FMT_SPLIT) ? 2 : 1;             // wave-loads per step
    static constexpr int LT = W * CPR / 64;                            // steps per tile
    static constexpr int S = 64 / W;                                   // lanes per window in the reduction
    static constexpr int E = CPR / S;                                  // partial sums per lane
    static constexpr int EP = (E & 1) ? E : E + 1;                     // odd lane stride: conflict-free reads
    static constexpr int LR = FIRD_R * LT;                             // steps per run
    static constexpr int U = (LR % 10 == 0) ? 10 : 12;
    static constexpr int B = U / 2;
    static constexpr int TS = CPR + 63;
    static constexpr int NPASS = (CPR + 63) / 64;                      // tap-table columns per lane
    static constexpr int TAB_BYTES = NPL * TS * 16;
    static constexpr int P_BYTES = 64 * EP * 8;
    static constexpr int WAVE_LDS = TAB_BYTES + P_BYTES;
    static constexpr unsigned int TILE_BYTES = (unsigned int)W * CPR * 16u;     // per plane
    static_assert((W * CPR) % 64 == 0 && CPR % S == 0 && LR % U == 0 && LR % B == 0 && B <= U, "tile geometry");
};

template <int FMT, int CPR, int W, int TILE>
__device__ __forceinline__ void firx_tile(u4v_t* st /* [NLD][U] */, __amdgpu_buffer_rsrc_t cur0, __amdgpu_buffer_rsrc_t cur1,
                                          __amdgpu_buffer_rsrc_t nxt0, __amdgpu_buffer_rsrc_t nxt1, unsigned int voff, const float4* Tl,
                                          f2* Pw, const f2* Pr, float* __restrict__ dm_out, int lane, float out_scale)
{
    typedef FirX<FMT, CPR, W> F;
    float4 w[F::NPL], nw[F::NPL];
#pragma unroll
    for (int k = 0; k < F::NPL; ++k) w[k] = Tl[k * F::TS];                 // step 0: column of lane 0 is 0
#pragma unroll
    for (int q = 0; q < F::LT; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        const int p = TILE * F::LT + q;                                    // run position consumed by this step
        u4v_t d[F::NLD];
#pragma unroll
        for (int l = 0; l < F::NLD; ++l) d[l] = st[l * F::U + p % F::U];
        if (p % F::B == 0) {
#pragma unroll
            for (int b = 0; b < F::B; ++b) {
                const int pos = p + F::U - F::B + b;
                if (pos < F::LR) {
                    st[pos % F::U] = fird_load(cur0, voff, (unsigned int)pos * 1024u);
                    if (F::NLD == 2) st[F::U + pos % F::U] = fird_load(cur1, voff, (unsigned int)pos * 1024u);
                } else {
                    st[pos % F::U] = fird_load(nxt0, voff, (unsigned int)(pos - F::LR) * 1024u);
                    if (F::NLD == 2) st[F::U + pos % F::U] = fird_load(nxt1, voff, (unsigned int)(pos - F::LR) * 1024u);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < F::NPL; ++k) nw[k] = w[k];
        if (q + 1 < F::LT) {
            const int c1 = ((q + 1) * 64) % CPR;
#pragma unroll
            for (int k = 0; k < F::NPL; ++k) nw[k] = Tl[k * F::TS + c1];
        }
        __builtin_amdgcn_sched_barrier(0);
        float wf[4 * F::NPL];
#pragma unroll
        for (int k = 0; k < F::NPL; ++k) { wf[4 * k] = w[k].x; wf[4 * k + 1] = w[k].y; wf[4 * k + 2] = w[k].z; wf[4 * k + 3] = w[k].w; }
        f2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
        f2 part;
        if (FMT == FMT_CS16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f2 tt;
                tt.x = (float)(short)(d[0][j] & 0xffffu);                  // soapy.c:238
                tt.y = (float)((int)d[0][j] >> 16);                        // soapy.c:239
                const f2 wv = {wf[2 * j], wf[2 * j + 1]};
                const f2 ws = {wf[2 * j + 1], wf[2 * j]};
                accA = __builtin_elementwise_fma(tt, wv, accA);
                accB = __builtin_elementwise_fma(tt, ws, accB);
            }
            part = {accA.x - accA.y, accB.x + accB.y};
        } else if (FMT == FMT_SPLIT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned int wi_ = d[0][j >> 1], wq_ = d[F::NLD - 1][j >> 1];
                f2 tt;
                tt.x = (j & 1) ? (float)((int)wi_ >> 16) : (float)(short)(wi_ & 0xffffu);     // sdrplay.c:219
                tt.y = (j & 1) ? (float)((int)wq_ >> 16) : (float)(short)(wq_ & 0xffffu);     // sdrplay.c:220
                const f2 wv = {wf[2 * j], wf[2 * j + 1]};
                const f2 ws = {wf[2 * j + 1], wf[2 * j]};
                accA = __builtin_elementwise_fma(tt, wv, accA);
                accB = __builtin_elementwise_fma(tt, ws, accB);
            }
            part = {accA.x - accA.y, accB.x + accB.y};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sv = __uint_as_float(d[0][j]);
                const f2 tt = {sv, sv};                                    // air.c:315-316: wf[i] * S
                const f2 wv = {wf[2 * j], wf[2 * j + 1]};
                accA = __builtin_elementwise_fma(tt, wv, accA);
            }
            part = accA;
        }
        if (F::EP == F::E) {
            Pw[q * 64] = part;
        } else {
            const unsigned int i = (unsigned int)(q * 64 + lane);
            const unsigned int r = (i * ((65536u + F::E - 1) / F::E)) >> 16;           // i / E for i < 64 * E <= 4096
            Pw[q * 64 + (int)r] = part;
        }
#pragma unroll
        for (int k = 0; k < F::NPL; ++k) w[k] = nw[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    // lane l adds entries l * E .. l * E + E - 1 (window l / S, its (l % S)-th part); S adjacent lanes meet
    f2 D = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < F::E; ++j) D = D + Pr[j];
#pragma unroll
    for (int m = 1; m < F::S; m <<= 1) {
        D.x += __shfl_xor(D.x, m);
        D.y += __shfl_xor(D.y, m);
    }
    // power-of-two output scale (1/32768 soapy.c:241, 1/4 sdrplay.c:225): exact, commutes with cabsf
    // (write-through like the u8 kernel's: beside the streaming reads a dirty line that waits in L2 costs more than one that
    //  leaves at once)
    if (lane % F::S == 0) {
        float* p = dm_out + lane / F::S;
        const float v = cabs_like_glibc(D.x, D.y) * out_scale;
        asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    }
}

template <int FMT, int CPR, int W>
__global__ __launch_bounds__(ACG_WG_FIR) void fir_fmt_direct_kernel(const FirArgs a, const uint8_t* __restrict__ in_base,
                                                                     const float* __restrict__ taps_base,
                                                                     const int* __restrict__ stream_of, float* __restrict__ dm_base)
{
    typedef FirX<FMT, CPR, W> F;
    constexpr unsigned int NONE = 0xffffffffu;
    if (a.high_prio) __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* my = fir_smem + wave * F::WAVE_LDS;
    float4* T = (float4*)my;
    f2* P = (f2*)(my + F::TAB_BYTES);
    const float4* Tl = T + lane;
    f2* Pw = P + lane;
    const f2* Pr = P + lane * F::EP;

    // a run = `pairs` consecutive two-tile bodies of one channel (launcher: as many as leave every wave ~32 runs), so that the
    // ticket, the row lookup and the tap table are paid once per run -- as in fir_u8_direct_kernel
    const unsigned int pairs = (unsigned int)a.run_pairs;
    const unsigned int ntile = (unsigned int)a.nwin / W;                            // whole tiles only (launcher)
    const unsigned int runs_per_ch = ntile / (FIRD_R * pairs);
    const unsigned int nrun = (unsigned int)a.nch * runs_per_ch;
    const unsigned int wpg = blockDim.x >> 6;                                       // the waves of a workgroup never talk to each other
    const unsigned int nwaves = gridDim.x * wpg;
    const unsigned int wg = blockIdx.x * wpg + (unsigned int)wave;
    unsigned int* ctr = a.work_counter;
    constexpr unsigned int run_bytes = FIRD_R * F::TILE_BYTES;
    const unsigned int voff = (unsigned int)lane << 4;
    const int nck = a.ntaps_pad / F::SPC;                                           // tap columns that carry taps

    auto shard_runs = [&](unsigned int s) { return (nrun + ACG_DISP_SHARDS - 1 - s) / ACG_DISP_SHARDS; };
    auto shard_static = [&](unsigned int s) { return (nwaves + ACG_DISP_SHARDS - 1 - s) / ACG_DISP_SHARDS; };
    auto run_of_ticket = [&](unsigned int s, unsigned int t) -> unsigned int {
        const unsigned long long k = (unsigned long long)shard_static(s) + t;
        return k < shard_runs(s) ? (unsigned int)k * ACG_DISP_SHARDS + s : NONE;
    };
    auto probe = [&](unsigned int& s) -> unsigned int {
        for (int k = 0; k < ACG_DISP_SHARDS; ++k) {
            unsigned int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ctr + s * ACG_DISP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            const unsigned int r = run_of_ticket(s, t);
            if (r != NONE) return r;
            s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
        }
        return NONE;
    };
    auto sign_off = [&]() {
        if (lane == 0) {
            const unsigned int d = atomicAdd(ctr + ACG_DISP_SHARDS * ACG_DISP_STRIDE, 1u);
            if (d == nwaves - 1) {
#pragma unroll
                for (int k = 0; k <= ACG_DISP_SHARDS; ++k)
                    __hip_atomic_store(ctr + k * ACG_DISP_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto run_base = [&](unsigned int run, unsigned int& ch, unsigned int& t0) -> const uint8_t* {
        ch = run / runs_per_ch;
        t0 = (run - ch * runs_per_ch) * FIRD_R * pairs;
        return in_base + (size_t)(a.stream_identity ? (int)ch : stream_of[ch]) * a.pitch + (size_t)t0 * F::TILE_BYTES;
    };
    // taps of channel ch: lane takes columns lane, lane + 64, ... (16 * NPL bytes each)
    auto fetch_taps = [&](unsigned int ch, float4 (&tp)[F::NPASS][F::NPL]) {
#pragma unroll
        for (int ps = 0; ps < F::NPASS; ++ps) {
            const int col = lane + 64 * ps;
            const float4* src = (const float4*)(taps_base + (size_t)ch * a.ntaps_pad * 2) + (col < nck ? col : 0) * F::NPL;
#pragma unroll
            for (int k = 0; k < F::NPL; ++k) tp[ps][k] = src[k];
        }
    };
    auto write_taps = [&](const float4 (&tp)[F::NPASS][F::NPL]) {
#pragma unroll
        for (int ps = 0; ps < F::NPASS; ++ps) {
            const int col = lane + 64 * ps;
            const bool on = col < nck;
#pragma unroll
            for (int rep = 0; rep * CPR < F::TS; ++rep) {
                const int u = col + rep * CPR;
                if (col < CPR && u < F::TS) {
#pragma unroll
                    for (int k = 0; k < F::NPL; ++k) T[k * F::TS + u] = on ? tp[ps][k] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };

    unsigned int s = wg % ACG_DISP_SHARDS;
    unsigned int run = wg < nrun ? wg : NONE;
    if (run == NONE) {
        s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
        run = probe(s);
    }
    if (run == NONE) { sign_off(); return; }

    unsigned int ch, t0;
    const uint8_t* base = run_base(run, ch, t0);
    {
        float4 tp[F::NPASS][F::NPL];
        fetch_taps(ch, tp);
        write_taps(tp);
    }
    __amdgpu_buffer_rsrc_t cur0 = fird_rsrc(base, run_bytes);
    __amdgpu_buffer_rsrc_t cur1 = fird_rsrc(base + (F::NLD == 2 ? a.plane : 0), run_bytes);
    u4v_t st[F::NLD * F::U];
#pragma unroll
    for (int i = 0; i < F::U - F::B; ++i) {
        st[i] = fird_load(cur0, voff, (unsigned int)i * 1024u);
        if (F::NLD == 2) st[F::U + i] = fird_load(cur1, voff, (unsigned int)i * 1024u);
        __builtin_amdgcn_sched_barrier(0);
    }

    for (;;) {
        unsigned int tk;
        if (lane == 0) ticket_request(ctr + s * ACG_DISP_STRIDE, tk);
        float* __restrict__ dm_out = dm_base + (size_t)ch * a.dm_pitch + (size_t)t0 * W;
        static_assert(FIRD_R == 2, "a body is unrolled by hand: first tile, second tile");
        bool has_next = true;
        unsigned int nrun_ = NONE, nch_ = ch, nt0 = t0;
        float4 tp[F::NPASS][F::NPL];
        for (unsigned int j = 0; j < pairs; ++j) {
            firx_tile<FMT, CPR, W, 0>(st, cur0, cur1, cur0, cur1, voff, Tl, Pw, Pr, dm_out, lane, a.out_scale);
            // second tile of the body: where does the stream go next -- the next body of this run, or the next run's first
            const uint8_t* nbase = base + run_bytes;
            if (j + 1 == pairs) {
                nrun_ = run_of_ticket(s, ticket_take<F::NLD * F::U + 1>(tk));
                if (nrun_ == NONE) {
                    s = (s + 1 == ACG_DISP_SHARDS) ? 0 : s + 1;
                    nrun_ = probe(s);
                }
                has_next = nrun_ != NONE;
                nbase = base;
                if (has_next) nbase = run_base(nrun_, nch_, nt0);
                fetch_taps(nch_, tp);
            }
            const __amdgpu_buffer_rsrc_t nxt0 = fird_rsrc(nbase, has_next ? run_bytes : 0u);
            const __amdgpu_buffer_rsrc_t nxt1 = fird_rsrc(nbase + (F::NLD == 2 ? a.plane : 0), has_next ? run_bytes : 0u);
            firx_tile<FMT, CPR, W, 1>(st, cur0, cur1, nxt0, nxt1, voff, Tl, Pw, Pr, dm_out + W, lane, a.out_scale);
            dm_out += FIRD_R * W;
            base = nbase;
            cur0 = nxt0;
            cur1 = nxt1;
        }
        if (!has_next) break;
        write_taps(tp);
        run = nrun_;
        ch = nch_;
        t0 = nt0;
    }
    sign_off();
}

// Launch-side state is per DEVICE (function attributes belong to the device's code object; the CU count is
// the device's): a process may hold contexts on several GPUs.
#define FIR_MAXDEV 64
struct FirDev {
    bool ready;
    int num_cu;
    std::map<const void*, size_t>* lds_optin;      // kernel -> dynamic LDS bytes it has been opted in for on this device
};
static FirDev g_firdev[FIR_MAXDEV];
static std::mutex g_firdev_mx;

static int fir_device(FirDev** out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= FIR_MAXDEV) return (int)hipErrorInvalidDevice;
    FirDev* d = &g_firdev[dev];
    std::lock_guard<std::mutex> lk(g_firdev_mx);
    if (!d->ready) {
        d->num_cu = 256;
        (void)hipDeviceGetAttribute(&d->num_cu, hipDeviceAttributeMultiprocessorCount, dev);
        d->lds_optin = new std::map<const void*, size_t>();
        d->ready = true;
    }
    *out = d;
    return 0;
}

// Dynamic LDS above the default limit needs an opt-in per kernel (hipFuncAttributeMaxDynamicSharedMemorySize).  It is
// made HERE, at the launch site, with the bytes the launch is about to ask for and for exactly the instantiation being
// launched -- a table of instantiations kept elsewhere drifts from the launch switch (round 2's did: the write-through
// kernels that are the default were not in it).
static int fir_lds_optin(const void* kernel, size_t bytes)
{
    FirDev* d = nullptr;
    if (int e = fir_device(&d)) return e;
    std::lock_guard<std::mutex> lk(g_firdev_mx);
    size_t& have = (*d->lds_optin)[kernel];
    if (bytes > have) {
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        have = bytes;
    }
    return 0;
}
#define FIR_LAUNCH(K_, grid_, blk_, lds_, stream_, ...)                                            \
    do {                                                                                           \
        if (int oe_ = fir_lds_optin((const void*)(K_), (size_t)(lds_))) return oe_;                \
        hipLaunchKernelGGL((K_), grid_, blk_, lds_, stream_, __VA_ARGS__);                         \
    } while (0)

// measurement / layout switches: one table, filled from the environment once per process (acg_api.cpp)
extern "C" int acg_tune_get(const char* name, int dflt);
extern "C" int acg_tune_has(const char* name);
static int env_int(const char* name, int dflt) { return acg_tune_get(name, dflt); }

template <int FMT, int CPR, int W>
static int launch_fmt_direct(const FirArgs* a, int num_cu, hipStream_t stream)
{
    typedef FirX<FMT, CPR, W> F;
    // workgroup shape as for the u8 kernel (launch_direct): 4-wave workgroups alone, single-wave workgroups (seven per CU)
    // beside demodulator workgroups on the same CUs, so that the demodulator's LDS fits next to them; 8 waves per CU at most
    int wpg = env_int("ACG_FIR_WAVES_PER_WG", a->shares_cus ? 1 : 4);
    if (wpg != 1 && wpg != 2 && wpg != 4) wpg = 4;
    const size_t lds = (size_t)wpg * F::WAVE_LDS;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8 / wpg) per_cu = 8 / wpg;
    if (a->shares_cus && wpg == 1 && per_cu > 7) per_cu = 7;
    if (per_cu < 1) return (int)hipErrorInvalidValue;
    per_cu = env_int("ACG_FIR_WG_PER_CU", per_cu);
    long long grid = (long long)(a->ncu > 0 ? a->ncu : num_cu) * per_cu;
    const long long bodies_per_ch = a->nwin / W / FIRD_R;
    const long long bodies = (long long)a->nch * bodies_per_ch;
    int pairs = 1;
    while (pairs < 8 && bodies_per_ch % (2 * pairs) == 0 && bodies / (2 * pairs) >= 32 * grid * wpg) pairs *= 2;
    pairs = env_int("ACG_FIR_RUN_PAIRS", pairs);
    if (pairs < 1 || pairs > 8 || (pairs & (pairs - 1)) || bodies_per_ch % pairs) pairs = 1;      // 1, 2, 4 or 8
    const long long nrun = bodies / pairs;
    const long long need = (nrun + wpg - 1) / wpg;
    if (grid > need) grid = need;
    FirArgs b = *a;
    b.run_pairs = pairs;
    FIR_LAUNCH((fir_fmt_direct_kernel<FMT, CPR, W>), dim3((unsigned int)grid), dim3(64 * wpg), lds, stream, b, a->iq, a->taps,
                       a->stream_of, a->dm);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_fir_fmt(const FirArgs* a, int fmt, void* stream)
{
    const size_t lds = (size_t)ACG_TILE_WIN * a->row_stride + 4 * 64 * sizeof(float4);
    FirDev* fd = nullptr;
    if (int e = fir_device(&fd)) return e;
    const int num_cu = fd->num_cu;
    // the wave-private streaming kernel where it is instantiated for the window length and the launch is whole runs
    // (ACG_FIR_VARIANT=3 forces the workgroup-granular kernel below)
    if (env_int("ACG_FIR_VARIANT", 5) >= 5 && a->nwin > 0 && (long long)a->nch * a->nwin < (1ll << 31)) {
        const int cprw = a->cpr_total;                              // 16-byte chunks per window (per plane for split int16)
#define FIRX_TRY(F_, C_, W_) \
        if (fmt == F_ && (F_ == FMT_SPLIT ? cprw / 2 : cprw) == C_ && a->nwin % (W_ * FIRD_R) == 0) \
            return launch_fmt_direct<F_, C_, W_>(a, num_cu, (hipStream_t)stream);
        FIRX_TRY(FMT_CS16, 40, 32) FIRX_TRY(FMT_CS16, 48, 32) FIRX_TRY(FMT_CS16, 50, 32)
        FIRX_TRY(FMT_F32R, 50, 32) FIRX_TRY(FMT_F32R, 60, 16) FIRX_TRY(FMT_F32R, 120, 8) FIRX_TRY(FMT_F32R, 200, 8)
        FIRX_TRY(FMT_SPLIT, 20, 64)
#undef FIRX_TRY
    }
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    const long long ntile = (a->nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a->nch * ntile;
    long long grid = (long long)(a->ncu > 0 ? a->ncu : num_cu) * per_cu;
    if (grid > G) grid = G;
    switch (fmt) {
    case FMT_CS16:
        FIR_LAUNCH(fir_fmt_kernel<FMT_CS16>, dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a, a->iq, a->taps, a->stream_of, a->dm);
        break;
    case FMT_SPLIT:
        FIR_LAUNCH(fir_fmt_kernel<FMT_SPLIT>, dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a, a->iq, a->taps, a->stream_of, a->dm);
        break;
    case FMT_F32R:
        FIR_LAUNCH(fir_fmt_kernel<FMT_F32R>, dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a, a->iq, a->taps, a->stream_of, a->dm);
        break;
    default:
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

// Exact-order kernel: one thread per output, the M terms accumulated sequentially with separately rounded products,
// differences and sums and the 127.37 subtracted per sample -- rtl.c:335-353 operation for operation as an IEEE (-O2, no
// contraction) build of the reference executes it, so dm is BIT-IDENTICAL to such a build (and to oracle/orc_fir_u8).
// Two uses: the fallback for decimations whose window is not a whole number of 16-byte chunks (M % 8 != 0), and the
// verification mode (ACG_F_EXACT_FIR): the streaming kernels re-associate the sum (1e-7 relative), which can flip a
// razor-edge soft decision of the demodulator; with this kernel in front the whole GPU path is bit-identical end to end.
__global__ void fir_u8_generic_kernel(const FirArgs a)
{
    // no fused multiply-add anywhere in this kernel: HIP's __fmul_rn / __fadd_rn are plain operators and contract like any
    // other under the default -ffp-contract=fast (round 2's fallback kernel did: dm was 1e-8 off the oracle, inside the
    // tolerance nobody looked under).  tests/test_host_logic.py disassembles the kernel and checks.
#pragma clang fp contract(off)
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.nch * a.nwin;
    if (gid >= total) return;
    const int ch = (int)(gid / a.nwin);
    const int m = (int)(gid - (long long)ch * a.nwin);
    const uint8_t* p = a.iq + (size_t)a.stream_of[ch] * a.pitch + (size_t)m * a.row_bytes;
    const float* w = a.taps + (size_t)ch * a.ntaps_pad * 2;
    float Dr = 0.f, Di = 0.f;
    for (int k = 0; k < a.ntaps; ++k) {
        // plain operators on purpose: the pragma governs the operations written HERE, not the bodies of header inlines
        const float r = (float)p[2 * k] - 127.37f;                         // rtl.c:338
        const float g = (float)p[2 * k + 1] - 127.37f;                     // rtl.c:339
        const float wr = w[2 * k], wi = w[2 * k + 1];
        const float pr = r * wr - g * wi;                                  // rtl.c:351 vb[ind] * wf[ind]
        const float pi = r * wi + g * wr;
        Dr = Dr + pr;                                                      // D +=
        Di = Di + pi;
    }
    a.dm[(size_t)ch * a.dm_pitch + m] = cabs_like_glibc(Dr, Di);
}

extern "C" size_t acg_fir_lds_bytes(const FirArgs* a)
{
    return (size_t)ACG_TILE_WIN * a->row_stride + 4 * 64 * sizeof(float4) + 16;
}

// wave-private streaming kernel: whole tiles, runs inside one channel, a rate it is instantiated for
template <int CPR, int UU = 0, int BB = 0, bool FOLD = true, bool WT = false, bool NOMAC = false>
static int launch_direct(const FirArgs* a, int num_cu, hipStream_t stream)
{
    // The waves of a workgroup are independent, so the workgroup size only decides in what pieces LDS is handed out
    // (ACG_FIR_WAVES_PER_WG = 1, 2, 4).  Four measured best alone (+5 % over single-wave workgroups at 16 384
    // channels) and within noise of the others beside the demodulator.  At most 8 waves per CU (2 per SIMD).
    // Beside demodulator workgroups on the same CUs (no CU partition: > 2048 channels) single-wave workgroups, seven per
    // CU: the demodulator's LDS (28 KiB per CU at 16 384 channels) then fits next to them instead of keeping a whole
    // 4-wave workgroup out (+13 % whole job at 4096 channels, +2 % at 16 384).
    int wpg = env_int("ACG_FIR_WAVES_PER_WG", a->shares_cus ? 1 : 4);
    if (wpg != 1 && wpg != 2 && wpg != 4) wpg = 4;
    const size_t lds = (size_t)wpg * FirD<CPR>::WAVE_LDS;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8 / wpg) per_cu = 8 / wpg;
    if (a->shares_cus && wpg == 1 && per_cu > 7) per_cu = 7;
    per_cu = env_int("ACG_FIR_WG_PER_CU", per_cu);
    long long grid = (long long)(a->ncu > 0 ? a->ncu : num_cu) * per_cu;
    // two-tile bodies per run: as many (1, 2, 4, 8) as leave every resident wave ~32 runs (the tail of the launch is
    // one run long) and divide the bodies of a channel
    const long long bodies_per_ch = a->nwin / ACG_TILE_WIN / FIRD_R;
    const long long bodies = (long long)a->nch * bodies_per_ch;
    int pairs = 1;
    while (pairs < 8 && bodies_per_ch % (2 * pairs) == 0 && bodies / (2 * pairs) >= 32 * grid * wpg) pairs *= 2;
    pairs = env_int("ACG_FIR_RUN_PAIRS", pairs);
    if (pairs < 1 || pairs > 8 || (pairs & (pairs - 1)) || bodies_per_ch % pairs) pairs = 1;      // 1, 2, 4 or 8
    const long long nrun = bodies / pairs;
    const long long need = (nrun + wpg - 1) / wpg;
    if (grid > need) grid = need;
    FirArgs b = *a;
    b.run_pairs = pairs;
#ifdef ACG_LAB
    if (acg_tune_get("ACG_FIR_DEBUG_DMPITCH0", 0)) b.dm_pitch = 0;          // measurement aid: every channel's dm lands in the first row (no write stream to HBM)
    if (acg_tune_has("ACG_FIR_DEBUG_SHAPE"))
        fprintf(stderr, "fir_u8_direct<%d>: nch %d nwin %d  wpg %d per_cu %d grid %lld  bodies/ch %lld pairs %d runs %lld  shares_cus %d prio %d\n",
                CPR, a->nch, a->nwin, wpg, per_cu, grid, bodies_per_ch, pairs, nrun, a->shares_cus, a->high_prio);
#endif
    FIR_LAUNCH((fir_u8_direct_kernel<CPR, UU, BB, FOLD, WT, NOMAC>), dim3((unsigned int)grid), dim3(64 * wpg), lds, stream, b, a->iq, a->taps,
                       a->stream_of, a->dm);
    return (int)hipGetLastError();
}

#ifdef ACG_LAB   // launchers of the lab variants
#include "lab/fir_lab_launch.inc"
#endif  // ACG_LAB

extern "C" int acg_launch_fir(const FirArgs* a, void* stream)
{
    const size_t lds = acg_fir_lds_bytes(a);
    FirDev* fd = nullptr;
    if (int e = fir_device(&fd)) return e;
    const int num_cu = fd->num_cu;
    // ACG_FIR_VARIANT.  The PRODUCT library has two kernels behind this entry point: 5 (default) the wave-private streaming
    // kernel where it applies (whole two-tile bodies, rtlMult 160 / 192 / 200; dm stored write-through), and 3, the
    // workgroup-granular kernel with the dynamic run dispenser, for everything else (ragged launches, other window lengths).
    // The LAB build (libacarsdec_amd_lab.so, -DACG_LAB; tests and probes only) adds: 0 one workgroup per segment; 1 / 2 static
    // persistent partition without / with non-temporal loads; 4 LDS-DMA double buffering; 6 matrix pipe; 7 taps in registers;
    // 8 = 7 with the results parked in LDS and written in chip-wide bursts; 50..56, 70..73 measurement knobs of 5 and 7
    // (55: write-back stores; 56: no arithmetic, dm is garbage).
    int variant = env_int("ACG_FIR_VARIANT", 5);
#ifndef ACG_LAB
    if (variant != 3) variant = 5;
#else
    if ((variant == 7 || variant == 8 || (variant >= 70 && variant <= 73)) && a->cpr == 25 && a->nwin > 0 && a->nwin % ACG_TILE_WIN == 0 && (a->nwin / ACG_TILE_WIN) % FIRD_R == 0 &&
        (long long)a->nch * (a->nwin / ACG_TILE_WIN) < (1ll << 31) && a->ntaps_pad <= a->decim)
    {            // 70..73: measurement knobs (staging slots, burst length)
        if (variant == 8) return launch_coltap<true, 13, 1, true>(a, num_cu, (hipStream_t)stream);      // results parked in LDS, chip-wide bursts
        if (variant == 70) return launch_coltap<true, 26, 1>(a, num_cu, (hipStream_t)stream);
        if (variant == 71) return launch_coltap<true, 26, 2>(a, num_cu, (hipStream_t)stream);
        if (variant == 72) return launch_coltap<true, 26, 13>(a, num_cu, (hipStream_t)stream);
        if (variant == 73) return launch_coltap<true, 4, 2>(a, num_cu, (hipStream_t)stream);
        return launch_coltap<>(a, num_cu, (hipStream_t)stream);
    }
    if (variant == 6 && a->cpr == 25 && a->nwin > 0 && a->nwin % ACG_TILE_WIN == 0 && (a->nwin / ACG_TILE_WIN) % FIRD_R == 0 &&
        (long long)a->nch * (a->nwin / ACG_TILE_WIN) < (1ll << 31) && a->ntaps_pad <= a->decim)
        return launch_mfma<25>(a, num_cu, (hipStream_t)stream);
#endif
    if ((variant == 5 || (variant >= 50 && variant <= 56)) && a->nwin > 0 && a->nwin % ACG_TILE_WIN == 0 && (a->nwin / ACG_TILE_WIN) % FIRD_R == 0 &&
        (long long)a->nch * (a->nwin / ACG_TILE_WIN) < (1ll << 31) && a->ntaps_pad <= a->decim) {
        switch (a->cpr) {
        // (dm is stored write-through: beside the streaming reads a write-back line costs more, see firc_flush)
        case 20: return launch_direct<20, 0, 0, true, true>(a, num_cu, (hipStream_t)stream);
        case 24: return launch_direct<24, 0, 0, true, true>(a, num_cu, (hipStream_t)stream);
        case 25:
#ifdef ACG_LAB    // 50..54: measurement knobs (staging slots, burst length, 127.37 per sample)
            if (variant == 50) return launch_direct<25, 10, 1>(a, num_cu, (hipStream_t)stream);
            if (variant == 51) return launch_direct<25, 25, 5>(a, num_cu, (hipStream_t)stream);
            if (variant == 52) return launch_direct<25, 25, 10>(a, num_cu, (hipStream_t)stream);
            if (variant == 53) return launch_direct<25, 10, 10>(a, num_cu, (hipStream_t)stream);
            if (variant == 54) return launch_direct<25, 0, 0, false>(a, num_cu, (hipStream_t)stream);      // 127.37 subtracted per sample
            if (variant == 56) return launch_direct<25, 0, 0, true, true, true>(a, num_cu, (hipStream_t)stream);   // measurement: no arithmetic (dm is garbage)
            if (variant == 55) return launch_direct<25>(a, num_cu, (hipStream_t)stream);                    // dm stored write-back (round 2a)
#endif
            return launch_direct<25, 0, 0, true, true>(a, num_cu, (hipStream_t)stream);
        default: break;
        }
    }
#ifdef ACG_LAB
    if (variant == 0) {
        const unsigned int grid = (unsigned int)a->nch * (unsigned int)a->nseg;
        FIR_LAUNCH(fir_u8_tile_kernel, dim3(grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a,
                           a->iq, a->taps, a->stream_of, a->dm);
        return (int)hipGetLastError();
    }
    if (variant == 4 && (a->cpr & 1) && a->row_stride == a->row_bytes) {
        const size_t tile = (size_t)ACG_TILE_WIN * a->row_bytes;
        const size_t lds2 = 2 * tile + 2 * 4 * 64 * sizeof(float4) + 16;
        if (lds2 <= 96 * 1024) {
            int per = (int)((160 * 1024) / lds2);
            if (per > 4) per = 4;
            per = env_int("ACG_FIR_WG_PER_CU", per);
            const long long ntile4 = (a->nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
            const long long G4 = (long long)a->nch * ntile4;
            const long long nrun4 = (G4 + FIR_RUN - 1) / FIR_RUN;
            long long grid4 = (long long)num_cu * per;
            if (grid4 > nrun4) grid4 = nrun4;
            FIR_LAUNCH(fir_u8_dma_kernel, dim3((unsigned int)grid4), dim3(ACG_WG_FIR), lds2, (hipStream_t)stream,
                               *a, a->iq, a->taps, a->stream_of, a->dm);
            return (int)hipGetLastError();
        }
    }
#endif
    // resident workgroups: LDS-limited (160 KiB per CU), at most 5 (VGPR budget of 4-wave workgroups)
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 5) per_cu = 5;
    if (a->wg_per_cu > 0 && per_cu > a->wg_per_cu) per_cu = a->wg_per_cu;
    if (per_cu < 1) per_cu = 1;
    per_cu = env_int("ACG_FIR_WG_PER_CU", per_cu);
    const long long ntile = (a->nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a->nch * ntile;
    long long grid = (long long)(a->ncu > 0 ? a->ncu : num_cu) * per_cu;
    if (grid > G) grid = G;
    FirArgs b = *a;
#ifdef ACG_LAB
    const bool nocompute = acg_tune_has("ACG_FIR_DEBUG_NOCOMPUTE") != 0;   // measurement aid: loads + LDS staging only
    if (nocompute) b.ntaps_pad = 0;
    if (variant == 1) {
        FIR_LAUNCH((fir_u8_persist_kernel<false, false, false>), dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds,
                           (hipStream_t)stream, b, a->iq, a->taps, a->stream_of, a->dm);
        return (int)hipGetLastError();
    }
    if (variant == 2) {
        FIR_LAUNCH((fir_u8_persist_kernel<true, false, false>), dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds,
                           (hipStream_t)stream, b, a->iq, a->taps, a->stream_of, a->dm);
        return (int)hipGetLastError();
    }
#endif
    const long long nrun = (G + FIR_RUN - 1) / FIR_RUN;
    if (grid > nrun) grid = nrun;
    if (G >= (1ll << 31)) return (int)hipErrorInvalidValue;
    if (a->nwin % ACG_TILE_WIN == 0)      // whole callbacks: every tile is complete
        FIR_LAUNCH((fir_u8_persist_kernel<true, true, true>), dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds,
                           (hipStream_t)stream, b, a->iq, a->taps, a->stream_of, a->dm);
    else
        FIR_LAUNCH((fir_u8_persist_kernel<true, true, false>), dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds,
                           (hipStream_t)stream, b, a->iq, a->taps, a->stream_of, a->dm);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_fir_shared(const FirArgs* a, void* stream)
{
    const size_t lds = (size_t)ACG_TILE_WIN * a->row_stride + 16 * 64 * sizeof(float4) + 16;
    FirDev* fd = nullptr;
    if (int e = fir_device(&fd)) return e;
    const int num_cu = fd->num_cu;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    per_cu = env_int("ACG_FIR_WG_PER_CU", per_cu);
    const long long ntile = (a->nwin + ACG_TILE_WIN - 1) / ACG_TILE_WIN;
    const long long G = (long long)a->ngroups * ntile;
    const long long nrun = (G + FIR_RUN - 1) / FIR_RUN;
    long long grid = (long long)(a->ncu > 0 ? a->ncu : num_cu) * per_cu;
    if (grid > nrun) grid = nrun;
    if (G >= (1ll << 31)) return (int)hipErrorInvalidValue;
    if (a->nwin % ACG_TILE_WIN == 0)
        FIR_LAUNCH(fir_u8_shared_kernel<true>, dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a,
                           a->iq, a->gtaps, a->groups, a->group_ch, a->dm);
    else
        FIR_LAUNCH(fir_u8_shared_kernel<false>, dim3((unsigned int)grid), dim3(ACG_WG_FIR), lds, (hipStream_t)stream, *a,
                           a->iq, a->gtaps, a->groups, a->group_ch, a->dm);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_regroup_taps(const FirArgs* a, void* stream)
{
    FIR_LAUNCH(regroup_taps_kernel, dim3((unsigned int)a->ngroups), dim3(256), 0, (hipStream_t)stream, a->taps,
                       (float*)a->gtaps, a->groups, a->group_ch, a->ntaps_pad);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_fir_generic(const FirArgs* a, void* stream)
{
    const long long total = (long long)a->nch * a->nwin;
    const unsigned int grid = (unsigned int)((total + 255) / 256);
    FIR_LAUNCH(fir_u8_generic_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
