// fir_mm.hip -- the shared-stream down-converter on the matrix pipe (gfx950).
//
// rtl.c's own shape (rtl.c:344-354): ONE dongle stream feeds K channels (up to 16, acarsdec.h:30).  With one stream per channel
// the down-converter is a 1-D convolution that uses every input byte once -- HBM-bound, no matrix work (fir.hip).  With K
// channels per stream the same bytes meet K tap tables:
//       D[window][channel] = sum_k  x[window][k] * w[channel][k]            (32 windows x 2M bytes) x (2M bytes x 2K columns)
// is a dense contraction with 4K flop per input byte.  On the vector pipe that is VALU-bound from K ~ 5 on (fir_u8_shared_kernel:
// 0.3-0.4 of the packed-f32 peak at K = 8); on the matrix pipe it is HBM-bound again.  And it can be done in EXACT arithmetic:
//   * the samples are bytes: s = u8 - 128 is an int8 (one v_xor per four samples, no conversion);
//   * a channel's taps are quantised ONCE per tap table to 31-bit fixed point against the channel's largest tap
//     (q = rint(w * 2^(30 - e)), 2^(e-1) <= max |w| < 2^e: every tap within 2^7 of the largest keeps all 24 bits of its f32
//     mantissa, smaller ones are cut at 2^-31 of the largest -- 1e-10 of full scale over a window) and split into four balanced
//     base-256 digits, q = d0 + 2^8 d1 + 2^16 d2 + 2^24 d3, d in [-128, 127];
//   * v_mfma_i32_32x32x32_i8 accumulates sum s * d_p in int32 WITHOUT ROUNDING (|sum| < 400 * 128 * 128 < 2^23);
//   * the four digit sums are recombined exactly in f64, scaled by 2^(e-30), the constant of the channel
//     (128 - 127.37f) (1 + j) sum w is added, and the result is rounded to f32 ONCE: D is the correctly rounded value of
//     rtl.c:349-351's sum (up to the 2^-31 tap cut) instead of a chain of 2M rounded f32 operations; |D| as glibc's cabsf.
// The reference's own -Ofast build re-associates that sum (dm within 1e-5 relative); this kernel sits inside
// 2e-7 of the exact value (tests/test_gpu_round6.py).
//
// Matrix layout (one MFMA = 32 rows x 32 columns x 32 bytes of k):
//   columns  = 32 consecutive windows of the stream (B operand: lane l holds window l & 31, bytes 32 k + 16 (l >> 5) .. + 16
//              of its row: one ds_read_b128 from the wave's LDS copy of the tile);
//   rows     = 4 channels x (re, im) x 4 digits (A operand, resident in VGPRs for a whole run: a group of <= 8 channels is
//              two MFMAs per k-step).  Row r = digit + 4 (channel & 1) + 8 (re/im) + 16 (channel >> 1 & 1), chosen so that
//              the C/D layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) leaves all four digits AND
//              re and im of a channel in ONE lane: the recombination and |D| need no cross-lane traffic;
//   k        = the 2M interleaved I/Q bytes of a window; the coefficient of byte b of column "re" is (b odd ? -wi : wr)[b / 2],
//              of column "im" (b odd ? wr : wi)[b / 2]; bytes past 2M (the last k-step of an odd row length) meet zero digits.
// Memory path: the wave-private streaming pattern of fir.hip -- 1 KiB coalesced non-temporal wave-loads of a tile (32 windows,
// contiguous in HBM) into registers while the previous tile is multiplied, then one pass of ds_write_b128 into the wave's LDS
// tile (row stride an odd number of 16-byte slots: the window-per-lane reads are bank-conflict free) -- no barrier, no other
// wave involved.  8 waves per CU with 13.2 KiB of LDS and 10-12.5 KiB of loads in flight each, or (beside the demodulator) 4 waves
// per CU with two tiles = 20-25 KiB in flight each.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include "acg_internal.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef unsigned int u4m_t __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) unsigned char mm_smem[];
extern "C" int acg_tune_get(const char* name, int dflt);          // measurement / layout switches (acg_api.cpp)

template <int CPR>
struct FirMM {
    static constexpr int WIN = 32;                                   // windows per tile = columns of one MFMA
    static constexpr int RB = CPR * 16;                              // bytes per window (2 M)
    static constexpr int KS = (CPR + 1) / 2;                         // 32-byte k-steps per window
    static constexpr int SLOTS = (CPR & 1) ? CPR : CPR + 1;          // LDS row stride in 16-byte slots: odd
    static constexpr int S = SLOTS * 16;
    static constexpr int NLD = (WIN * CPR + 63) / 64;                // wave-loads per tile (the last one may be half empty)
    static constexpr int TILE_BYTES = WIN * RB;
    static constexpr int TILE_LDS = WIN * S + 16;                    // + the slot the last k-step of the last row reads
    static constexpr int CONST_LDS = 8 * 32;                         // MmChan of the group's channels
    static constexpr int WAVE_LDS = ((TILE_LDS + CONST_LDS + 127) / 128) * 128;
    static constexpr int IMG_V4 = KS * 2 * 64;                       // 16-byte entries of a group's A-operand image
};

// ---- per tap table: the channel's scale and constant, then the groups' A-operand images ---------------------------------
__global__ void mm_chan_kernel(const float* __restrict__ taps, int ntaps_pad, int M, MmChan* __restrict__ out)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    const float2* tp = (const float2*)(taps + (size_t)ch * ntaps_pad * 2);
    const int n1 = ntaps_pad < M ? ntaps_pad : M;
    float mx = 0.f;
    for (int n = lane; n < n1; n += 64) mx = fmaxf(mx, fmaxf(fabsf(tp[n].x), fabsf(tp[n].y)));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    int e = 0;
    const bool any = mx > 0.f && mx < 3.0e38f;
    if (any) (void)frexpf(mx, &e);                                   // mx = f 2^e, 0.5 <= f < 1
    const double up = any ? ldexp(1.0, 30 - e) : 0.0;
    long long sr = 0, si = 0;
    for (int n = lane; n < n1; n += 64) {
        sr += (long long)__double2int_rn((double)tp[n].x * up);
        si += (long long)__double2int_rn((double)tp[n].y * up);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sr += __shfl_xor(sr, m, 64); si += __shfl_xor(si, m, 64); }
    if (lane == 0) {
        const double scale = any ? ldexp(1.0, e - 30) : 0.0;
        const double c = 128.0 - (double)127.37f;                    // u8 - 127.37f = (u8 - 128) + c, exactly (rtl.c:338-339)
        MmChan o;
        o.scale = scale;
        o.dc_re = c * (double)(sr - si) * scale;                     // c (1 + j) sum w
        o.dc_im = c * (double)(sr + si) * scale;
        o.up = up;
        out[ch] = o;
    }
}

template <int CPR>
__global__ void mm_image_kernel(const float* __restrict__ taps, int ntaps_pad, int M, const int4* __restrict__ groups,
                                const int* __restrict__ group_ch, const MmChan* __restrict__ mmch, u4m_t* __restrict__ img)
{
    typedef FirMM<CPR> F;
    const int4 gi = groups[blockIdx.x];
    const int kc = gi.z;
    const int n1 = ntaps_pad < M ? ntaps_pad : M;
    for (int item = threadIdx.x; item < F::IMG_V4; item += blockDim.x) {
        const int l = item & 63, m = (item >> 6) & 1, k = item >> 7;
        const int rho = l & 31, h = l >> 5;
        const int piece = rho & 3, chl = 4 * m + ((rho >> 2) & 1) + 2 * (rho >> 4), reim = (rho >> 3) & 1;
        u4m_t w = {0u, 0u, 0u, 0u};
        if (chl < kc) {
            const int ch = group_ch[gi.y + chl];
            const float2* tp = (const float2*)(taps + (size_t)ch * ntaps_pad * 2);
            const double up = mmch[ch].up;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int b = 32 * k + 16 * h + j, n = b >> 1;
                int q = 0;
                if (n < n1) {
                    const float2 t = tp[n];
                    const float coef = reim ? ((b & 1) ? t.x : t.y) : ((b & 1) ? -t.y : t.x);
                    q = __double2int_rn((double)coef * up);
                }
                for (int p = 0; p < piece; ++p) q = (q - (int)(signed char)(q & 255)) >> 8;       // exact: the difference is a multiple of 256
                const unsigned int d = (unsigned int)(piece == 3 ? q : (int)(signed char)(q & 255)) & 255u;
                w[j >> 2] |= d << (8 * (j & 3));
            }
        }
        img[(size_t)blockIdx.x * F::IMG_V4 + item] = w;
    }
}

__device__ __forceinline__ float mm_cabs(float re, float im)
{
    // glibc 2.35 cabsf == (float)sqrt((double)x*x + (double)y*y) (fir.hip: cabs_like_glibc)
    const double d = (double)re * (double)re + (double)im * (double)im;
    return (float)__dsqrt_rn(d);
}

// One run = `tpr` consecutive tiles of one group.  Runs are numbered group-major (a stream is read front to back); the first
// run of wave wg is run wg, further ones come from an atomic ticket (words [0] = tickets, [1] = waves finished: the last wave
// out re-arms both, no memset between launches).
// STAGES = tiles in flight per wave.  1: two waves per SIMD (<= 256 VGPRs each), the next tile's loads wait in 52 registers while
// this one is multiplied.  2: ONE wave per SIMD with two tiles (25 KiB) in flight in 104 registers -- the same bytes in flight per
// CU, and a SIMD's register file then still holds a demodulator wave (136 VGPRs) beside it (two 232-register waves do not leave room
// for one).  Measured in one process at 16 384 channels on 2 048 streams: 12.24 M channel*Msps against 12.12 M (+1 %): taken beside the
// demodulator, but register co-residency is not what bounds this shape -- the vector pipe is: the demodulator needs ~75 % of a SIMD's
// issue slots and this kernel's epilogue (the exact recombination and |D| of 256 outputs per tile: ~200 instructions, 120 of them
// f64) ~20 % at this rate (profiles/LEDGER.md, round 6).
template <int CPR, int STAGES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STAGES == 1 ? 2 : 1, STAGES == 1 ? 2 : 1)))
void fir_u8_mm_kernel(const FirArgs a, const uint8_t* __restrict__ iq_base, const u4m_t* __restrict__ img,
                      const MmChan* __restrict__ mmch, const int4* __restrict__ groups, const int* __restrict__ group_ch,
                      float* __restrict__ dm_base)
{
    typedef FirMM<CPR> F;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* tile = mm_smem + wave * F::WAVE_LDS;
    MmChan* constL = (MmChan*)(tile + F::TILE_LDS);
    const unsigned int wpg = blockDim.x >> 6;
    const unsigned int nwaves = gridDim.x * wpg;
    const unsigned int wg = blockIdx.x * wpg + (unsigned int)wave;
    const unsigned int tpr = (unsigned int)a.run_pairs;                              // tiles per run (launcher)
    const unsigned int rpg = ((unsigned int)a.nwin / F::WIN) / tpr;                  // runs per group
    const unsigned int nrun = (unsigned int)a.ngroups * rpg;
    unsigned int* ctr = a.work_counter;
    const int w = lane & 31, h = lane >> 5;
    const unsigned char* rowp = tile + w * F::S + 16 * h;

    // where this lane's chunk of wave-load q goes in the LDS tile, and whether the chunk exists (the last load of an odd tile)
    unsigned int voff[F::NLD];
    unsigned int ldsoff[F::NLD];
#pragma unroll
    for (int q = 0; q < F::NLD; ++q) {
        const unsigned int c = (unsigned int)(q * 64 + lane);
        const unsigned int r = (c * ((65536u + CPR - 1) / CPR)) >> 16;                // c / CPR for c < 2048
        ldsoff[q] = (c << 4) + r * (unsigned int)(F::S - F::RB);
        voff[q] = c < (unsigned int)(F::WIN * CPR) ? ((unsigned int)lane << 4) : 0x80000000u;   // past the descriptor: returns 0, touches nothing
    }

    unsigned int run = wg;
    if (run >= nrun) {
        unsigned int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run = nwaves + (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    }
    while (run < nrun) {
        const unsigned int g = run / rpg;
        const unsigned int t0 = (run - g * rpg) * tpr;
        const int4 gi = groups[g];
        const int kc = gi.z;
        const uint8_t* base = iq_base + (size_t)gi.x * a.pitch + (size_t)t0 * F::TILE_BYTES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(tpr * (unsigned int)F::TILE_BYTES), 0x00020000);
        u4m_t stA[F::NLD], stB[F::NLD];
#pragma unroll
        for (int q = 0; q < F::NLD; ++q) stA[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[q], q * 1024, 2 /* nt */);
        if (STAGES == 2 && tpr > 1) {
#pragma unroll
            for (int q = 0; q < F::NLD; ++q) stB[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[q], F::TILE_BYTES + q * 1024, 2);
        }

        // the group's A operand (digits of its <= 8 tap tables) into registers, its channels' constants into LDS
        v4i_t tap[F::KS][2];
        {
            const v4i_t* ip = (const v4i_t*)img + (size_t)g * F::IMG_V4 + lane;
#pragma unroll
            for (int k = 0; k < F::KS; ++k) {
                tap[k][0] = ip[(k * 2 + 0) * 64];
                tap[k][1] = ip[(k * 2 + 1) * 64];
            }
            if (lane < 8) {
                const double4* src = (const double4*)(mmch + group_ch[gi.y + (lane < kc ? lane : 0)]);
                double4 v = *src;
                if (lane >= kc) v = make_double4(0.0, 0.0, 0.0, 0.0);
                *(double4*)(constL + lane) = v;
            }
        }
        // the four channels this lane finishes: local index 4 m + (lane >> 5) + 2 chhi
        float* dmrow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chl = 4 * (i >> 1) + h + 2 * (i & 1);
            const int ch = chl < kc ? group_ch[gi.y + chl] : -1;
            dmrow[i] = ch >= 0 ? dm_base + (size_t)ch * a.dm_pitch + (size_t)t0 * F::WIN + w : nullptr;
        }

        auto tile_step = [&](u4m_t (&st)[F::NLD], unsigned int t) {
            // ---- tile t: registers -> LDS (the previous tile's reads are older LDS operations of this wave: they are done)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < F::NLD; ++q)
                if (q + 1 < F::NLD || (F::WIN * CPR) % 64 == 0 || lane < (F::WIN * CPR) % 64) *(u4m_t*)(tile + ldsoff[q]) = st[q];
            // ---- ask for tile t + STAGES (lands in these registers while this one and the next are multiplied)
            if (t + STAGES < tpr) {
#pragma unroll
                for (int q = 0; q < F::NLD; ++q)
                    st[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[q], (int)((t + STAGES) * (unsigned int)F::TILE_BYTES) + q * 1024, 2);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- KS k-steps, two MFMAs each (channels 0-3 / 4-7 of the group)
            v16i_t acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            v16i_t acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < F::KS; ++k) {
                u4m_t d = *(const u4m_t*)(rowp + 32 * k);
                d ^= 0x80808080u;                                                      // u8 -> u8 - 128 as int8
                v4i_t b;
                b[0] = (int)d[0]; b[1] = (int)d[1]; b[2] = (int)d[2]; b[3] = (int)d[3];
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tap[k][0], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tap[k][1], b, acc1, 0, 0, 0);
            }
            // ---- lane = (window, channel parity): four channels x (re, im) x four digits, all in this lane
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chl = 4 * (i >> 1) + h + 2 * (i & 1);
                const MmChan cc = constL[chl];
                float v[2];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) {
                    const int r0 = 4 * (2 * (i & 1) + ri);                            // register group reg >> 2 = re/im + 2 chhi
                    const int a0 = (i >> 1) ? acc1[r0 + 0] : acc0[r0 + 0];
                    const int a1 = (i >> 1) ? acc1[r0 + 1] : acc0[r0 + 1];
                    const int a2 = (i >> 1) ? acc1[r0 + 2] : acc0[r0 + 2];
                    const int a3 = (i >> 1) ? acc1[r0 + 3] : acc0[r0 + 3];
                    const int lo = a1 * 256 + a0;                                      // |a1| < 2^23 - 2^15: fits
                    const int hi = a3 * 256 + a2;                                      // |a3| <= 400 * 128 * 65
                    const double D = __fma_rn((double)hi, 65536.0, (double)lo);        // exact (47 bits)
                    v[ri] = (float)__fma_rn(D, cc.scale, ri ? cc.dc_im : cc.dc_re);   // the one rounding of the sum
                }
                if (dmrow[i]) {
                    dmrow[i][0] = mm_cabs(v[0], v[1]);
                    dmrow[i] += F::WIN;
                }
            }
        };
        if (STAGES == 1) {
            for (unsigned int t = 0; t < tpr; ++t) tile_step(stA, t);
        } else {
            for (unsigned int t = 0; t < tpr; t += 2) {
                tile_step(stA, t);
                if (t + 1 < tpr) tile_step(stB, t + 1);
            }
        }
        unsigned int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run = nwaves + (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
    }
    if (lane == 0) {
        const unsigned int d = atomicAdd(ctr + 1, 1u);
        if (d == nwaves - 1) {                                      // every wave's draws have been answered: re-arm
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- one stream per channel on the matrix pipe ---------------------------------------------------------------------------
// fir_u8_mm1_kernel: the K = 1 case of the contraction above.  It does not need the matrix pipe for its arithmetic rate (8 flop per
// 2 bytes) -- it uses it to get the arithmetic OFF the vector pipe: the wave-private streaming kernel of fir.hip spends 16 v_cvt +
// 16 v_pk_fma + 4 ds_read_b128 of taps per 16 input bytes and lane (~40 VALU / LDS issues per KiB and wave: 40 % of every SIMD's
// issue slots at 6 TB/s), and beside it runs the demodulator, a serial chain that needs ~75 % of a SIMD by itself -- together they
// are VALU-bound before they are HBM-bound, and the arithmetic's power sets the shader clock the demodulator runs at.  Here a KiB
// costs 4 v_xor + one ds_write_b128 / ds_read_b128 pair + one MFMA issue, and the sum is exact (see above).
//   rows of the MFMA: digit + 8 (re / im) -- 8 of the 32 rows carry the channel's digits, the others are zero (the matrix pipe is
//   at ~10 % either way); C/D layout: lanes 0..31 hold re (regs 0..3) and im (regs 4..7) of their window, lanes 32..63 hold zeros.
//   A operand: 13 x 4 VGPRs, loaded once per run from the channel's compact digit image ([k-step][half][8 rows][16 B] = 3.3 KiB).
// Workgroup = one wave (LDS is handed out in 13 KiB pieces: seven fit beside the demodulator's 62 KiB per CU).
template <int CPR>
__global__ void mm1_image_kernel(const float* __restrict__ taps, int ntaps_pad, int M, const MmChan* __restrict__ mmch,
                                 u4m_t* __restrict__ img)
{
    typedef FirMM<CPR> F;
    const int ch = blockIdx.x;
    const int n1 = ntaps_pad < M ? ntaps_pad : M;
    const float2* tp = (const float2*)(taps + (size_t)ch * ntaps_pad * 2);
    const double up = mmch[ch].up;
    for (int item = threadIdx.x; item < F::KS * 16; item += blockDim.x) {            // (k, h, compact row): 16 bytes each
        const int ci = item & 7, h = (item >> 3) & 1, k = item >> 4;
        const int piece = ci & 3, reim = ci >> 2;
        u4m_t w = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = 32 * k + 16 * h + j, n = b >> 1;
            int q = 0;
            if (n < n1) {
                const float2 t = tp[n];
                const float coef = reim ? ((b & 1) ? t.x : t.y) : ((b & 1) ? -t.y : t.x);
                q = __double2int_rn((double)coef * up);
            }
            for (int p = 0; p < piece; ++p) q = (q - (int)(signed char)(q & 255)) >> 8;
            const unsigned int d = (unsigned int)(int)(signed char)(q & 255) & 255u;
            w[j >> 2] |= d << (8 * (j & 3));
        }
        img[(size_t)ch * (F::KS * 16) + item] = w;
    }
}

template <int CPR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 3)))
void fir_u8_mm1_kernel(const FirArgs a, const uint8_t* __restrict__ iq_base, const u4m_t* __restrict__ img,
                       const MmChan* __restrict__ mmch, const int* __restrict__ stream_of, float* __restrict__ dm_base)
{
    typedef FirMM<CPR> F;
    const int lane = threadIdx.x & 63;
    unsigned char* tile = mm_smem;
    const unsigned int nwaves = gridDim.x;
    const unsigned int wg = blockIdx.x;
    const unsigned int tpr = (unsigned int)a.run_pairs;                              // tiles per run (launcher)
    const unsigned int rpc = ((unsigned int)a.nwin / F::WIN) / tpr;                  // runs per channel
    const unsigned int nrun = (unsigned int)a.nch * rpc;
    unsigned int* ctr = a.work_counter;
    const int w = lane & 31, h = lane >> 5;
    const unsigned char* rowp = tile + w * F::S + 16 * h;
    if (a.high_prio) __builtin_amdgcn_s_setprio(2);

    unsigned int voff[F::NLD];
    unsigned int ldsoff[F::NLD];
#pragma unroll
    for (int q = 0; q < F::NLD; ++q) {
        const unsigned int c = (unsigned int)(q * 64 + lane);
        const unsigned int r = (c * ((65536u + CPR - 1) / CPR)) >> 16;
        ldsoff[q] = (c << 4) + r * (unsigned int)(F::S - F::RB);
        voff[q] = c < (unsigned int)(F::WIN * CPR) ? ((unsigned int)lane << 4) : 0x80000000u;
    }
    // this lane's piece of the A operand: row = lane & 31 -- rows 0..3 (re digits) and 8..11 (im digits) exist
    const int arow = lane & 31;
    const bool ahas = (arow & 4) == 0 && arow < 12;
    const int aci = (arow & 3) + 4 * (arow >> 3);

    unsigned int run = wg;
    if (run >= nrun) {
        unsigned int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run = nwaves + (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    }
    while (run < nrun) {
        const unsigned int ch = run / rpc;
        const unsigned int t0 = (run - ch * rpc) * tpr;
        const size_t srow = (size_t)(a.stream_identity ? (int)ch : stream_of[ch]) * a.pitch;
        const uint8_t* base = iq_base + srow + (size_t)t0 * F::TILE_BYTES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(tpr * (unsigned int)F::TILE_BYTES), 0x00020000);
        u4m_t st[F::NLD];
#pragma unroll
        for (int q = 0; q < F::NLD; ++q) st[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[q], q * 1024, 2 /* nt */);
        v4i_t tap[F::KS];
        {
            const v4i_t* ip = (const v4i_t*)img + (size_t)ch * (F::KS * 16) + (h * 8 + aci);
            const v4i_t z = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < F::KS; ++k) tap[k] = ahas ? ip[k * 16] : z;
        }
        const MmChan cc = mmch[ch];                                                   // (wave-uniform: scalar loads)
        float* dmp = dm_base + (size_t)ch * a.dm_pitch + (size_t)t0 * F::WIN + w;

        for (unsigned int t = 0; t < tpr; ++t) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < F::NLD; ++q)
                if (q + 1 < F::NLD || (F::WIN * CPR) % 64 == 0 || lane < (F::WIN * CPR) % 64) *(u4m_t*)(tile + ldsoff[q]) = st[q];
            if (t + 1 < tpr) {
#pragma unroll
                for (int q = 0; q < F::NLD; ++q)
                    st[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[q], (int)((t + 1) * (unsigned int)F::TILE_BYTES) + q * 1024, 2);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // two accumulator chains (even / odd k-steps): no MFMA waits for the one before it; integer sums add up exactly
            v16i_t acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            v16i_t acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < F::KS; ++k) {
                u4m_t d = *(const u4m_t*)(rowp + 32 * k);
                d ^= 0x80808080u;
                v4i_t b;
                b[0] = (int)d[0]; b[1] = (int)d[1]; b[2] = (int)d[2]; b[3] = (int)d[3];
                if (k & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tap[k], b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tap[k], b, acc0, 0, 0, 0);
            }
            if (h == 0) {                                                             // lanes 0..31: re in regs 0..3, im in regs 4..7
                float v[2];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) {
                    const int a0 = acc0[4 * ri + 0] + acc1[4 * ri + 0], a1 = acc0[4 * ri + 1] + acc1[4 * ri + 1];
                    const int a2 = acc0[4 * ri + 2] + acc1[4 * ri + 2], a3 = acc0[4 * ri + 3] + acc1[4 * ri + 3];
                    const int lo = a1 * 256 + a0;
                    const int hi = a3 * 256 + a2;
                    const double D = __fma_rn((double)hi, 65536.0, (double)lo);
                    v[ri] = (float)__fma_rn(D, cc.scale, ri ? cc.dc_im : cc.dc_re);
                }
                const float mag = mm_cabs(v[0], v[1]);
                asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(dmp), "v"(mag) : "memory");      // write-through, as fir.hip's kernel
            }
            dmp += F::WIN;
        }
        unsigned int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run = nwaves + (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
    }
    if (lane == 0) {
        const unsigned int d = atomicAdd(ctr + 1, 1u);
        if (d == nwaves - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- launch side ------------------------------------------------------------------------------------------------------
namespace {
struct MmDev {
    bool ready = false;
    int num_cu = 256;
    std::map<const void*, size_t> optin;
};
MmDev g_mmdev[64];
std::mutex g_mm_mx;

int mm_device(MmDev** out)
{
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_mm_mx);
    MmDev* d = &g_mmdev[dev];
    if (!d->ready) {
        (void)hipDeviceGetAttribute(&d->num_cu, hipDeviceAttributeMultiprocessorCount, dev);
        d->ready = true;
    }
    *out = d;
    return 0;
}

int mm_optin(MmDev* d, const void* kernel, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_mm_mx);
    size_t& have = d->optin[kernel];
    if (bytes > have) {
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        have = bytes;
    }
    return 0;
}

template <int CPR, int STAGES>
int launch_mm(const FirArgs* a, hipStream_t stream)
{
    typedef FirMM<CPR> F;
    MmDev* d = nullptr;
    if (int e = mm_device(&d)) return e;
    const int ncu = a->ncu > 0 ? a->ncu : d->num_cu;
    const unsigned int nwaves = (unsigned int)ncu * (STAGES == 1 ? 8u : 4u);
    const unsigned int ntile = (unsigned int)a->nwin / F::WIN;
    // ~4 runs per wave where the launch is large enough, at least two tiles per run (a run pays one tile of load latency
    // and 26 KiB of digits from L2)
    unsigned int rpg = 1;
    while (rpg * 2 * (unsigned int)a->ngroups <= 4 * nwaves && ntile % (rpg * 2) == 0 && ntile / (rpg * 2) >= 2) rpg *= 2;
    FirArgs b = *a;
    b.run_pairs = (int)(ntile / rpg);
    const unsigned long long nrun = (unsigned long long)a->ngroups * rpg;
    const unsigned int need = (unsigned int)((nrun + 3) / 4);
    const unsigned int blocks = (unsigned int)ncu * (STAGES == 1 ? 2u : 1u);
    const unsigned int grid = need < blocks ? need : blocks;
    const size_t lds = (size_t)4 * F::WAVE_LDS;
    if (int e = mm_optin(d, (const void*)fir_u8_mm_kernel<CPR, STAGES>, lds)) return e;
    hipLaunchKernelGGL((fir_u8_mm_kernel<CPR, STAGES>), dim3(grid), dim3(256), lds, stream, b, a->iq, (const u4m_t*)a->mm_img,
                       (const MmChan*)a->mm_chan, a->groups, a->group_ch, a->dm);
    return (int)hipGetLastError();
}
}  // namespace

// bytes of the A-operand image of `ngroups` groups at this decimation, 0 if the matrix kernel does not take the shape
extern "C" size_t acg_fir_mm_image_bytes(int decim, int ngroups)
{
    const int cpr = decim / 8;
    if (decim % 8 != 0 || !(cpr == 20 || cpr == 24 || cpr == 25)) return 0;
    return (size_t)ngroups * (size_t)((cpr + 1) / 2) * 2 * 1024;
}

// 1 if acg_launch_fir_mm takes this launch (rtlMult 160 / 192 / 200, whole 32-window tiles, < 2^31 bytes per run)
extern "C" int acg_fir_mm_takes(const FirArgs* a)
{
    return a->ngroups > 0 && a->mm_img && a->mm_chan && acg_fir_mm_image_bytes(a->decim, 1) != 0 && a->nwin > 0 && a->nwin % 64 == 0 &&
           (unsigned long long)a->nwin * 2ull * (unsigned long long)a->decim < (1ull << 31);
}

extern "C" int acg_launch_fir_mm_prep(const FirArgs* a, void* stream)
{
    hipLaunchKernelGGL(mm_chan_kernel, dim3((unsigned int)a->nch), dim3(64), 0, (hipStream_t)stream, a->taps, a->ntaps_pad, a->decim,
                       (MmChan*)a->mm_chan);
    switch (a->decim / 8) {
    case 20: hipLaunchKernelGGL(mm_image_kernel<20>, dim3((unsigned int)a->ngroups), dim3(256), 0, (hipStream_t)stream, a->taps, a->ntaps_pad,
                                a->decim, a->groups, a->group_ch, (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    case 24: hipLaunchKernelGGL(mm_image_kernel<24>, dim3((unsigned int)a->ngroups), dim3(256), 0, (hipStream_t)stream, a->taps, a->ntaps_pad,
                                a->decim, a->groups, a->group_ch, (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    case 25: hipLaunchKernelGGL(mm_image_kernel<25>, dim3((unsigned int)a->ngroups), dim3(256), 0, (hipStream_t)stream, a->taps, a->ntaps_pad,
                                a->decim, a->groups, a->group_ch, (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

extern "C" int acg_launch_fir_mm(const FirArgs* a, void* stream)
{
    if (!acg_fir_mm_takes(a)) return (int)hipErrorInvalidValue;
    // beside the demodulator on the same CUs: one wave per SIMD with two tiles in flight (see the kernel's comment)
    const int stages = acg_tune_get("ACG_FIR_MM_STAGES", a->shares_cus ? 2 : 1) == 2 ? 2 : 1;
    switch (a->decim / 8) {
    case 20: return stages == 2 ? launch_mm<20, 2>(a, (hipStream_t)stream) : launch_mm<20, 1>(a, (hipStream_t)stream);
    case 24: return stages == 2 ? launch_mm<24, 2>(a, (hipStream_t)stream) : launch_mm<24, 1>(a, (hipStream_t)stream);
    case 25: return stages == 2 ? launch_mm<25, 2>(a, (hipStream_t)stream) : launch_mm<25, 1>(a, (hipStream_t)stream);
    }
    return (int)hipErrorInvalidValue;
}

// ---- one stream per channel (fir_u8_mm1_kernel) ---------------------------------------------------------------------------
namespace {
template <int CPR>
int launch_mm1(const FirArgs* a, hipStream_t stream)
{
    typedef FirMM<CPR> F;
    MmDev* d = nullptr;
    if (int e = mm_device(&d)) return e;
    const int ncu = a->ncu > 0 ? a->ncu : d->num_cu;
    // 13.2 KiB of LDS and 160 VGPRs per wave: twelve fit a CU that the demodulator does not share (<= 2048 channels: measured
    // 8 / 10 / 12 waves 2.30 / 2.33 / 2.46 M channel*Msps at 2048 channels); beside the demodulator's workgroups (15.4 KiB each: two
    // per CU up to 4096 channels, four from 8192) nine or seven (profiles/r06_mm1_waves_ab_*.json)
    int per_cu = !a->shares_cus ? 12 : a->nch >= 8192 ? 7 : 9;
    per_cu = acg_tune_get("ACG_FIR_MM1_WAVES", per_cu);
    if (per_cu < 1 || per_cu > 12) per_cu = 8;
    const unsigned int nwaves = (unsigned int)ncu * (unsigned int)per_cu;
    const unsigned int ntile = (unsigned int)a->nwin / F::WIN;
    unsigned int rpc = 1;                                            // ~4 runs per wave, at least 8 tiles per run
    while (rpc * 2 * (unsigned int)a->nch <= 4 * nwaves && ntile % (rpc * 2) == 0 && ntile / (rpc * 2) >= 8) rpc *= 2;
    FirArgs b = *a;
    b.run_pairs = (int)(ntile / rpc);
    const unsigned long long nrun = (unsigned long long)a->nch * rpc;
    const unsigned int grid = nrun < nwaves ? (unsigned int)nrun : nwaves;
    const size_t lds = (size_t)F::WAVE_LDS;
    if (int e = mm_optin(d, (const void*)fir_u8_mm1_kernel<CPR>, lds)) return e;
    hipLaunchKernelGGL(fir_u8_mm1_kernel<CPR>, dim3(grid), dim3(64), lds, stream, b, a->iq, (const u4m_t*)a->mm_img,
                       (const MmChan*)a->mm_chan, a->stream_of, a->dm);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" size_t acg_fir_mm1_image_bytes(int decim, int nch)
{
    const int cpr = decim / 8;
    if (decim % 8 != 0 || !(cpr == 20 || cpr == 24 || cpr == 25)) return 0;
    return (size_t)nch * (size_t)((cpr + 1) / 2) * 256;
}

extern "C" int acg_fir_mm1_takes(const FirArgs* a)
{
    return a->ngroups == 0 && a->mm_img && a->mm_chan && acg_fir_mm1_image_bytes(a->decim, 1) != 0 && a->nwin > 0 && a->nwin % 256 == 0 &&
           a->ntaps_pad <= a->decim && (unsigned long long)a->nwin * 2ull * (unsigned long long)a->decim < (1ull << 31) &&
           (unsigned long long)a->nch * ((unsigned long long)a->nwin / 32) < (1ull << 31);
}

extern "C" int acg_launch_fir_mm1_prep(const FirArgs* a, void* stream)
{
    hipLaunchKernelGGL(mm_chan_kernel, dim3((unsigned int)a->nch), dim3(64), 0, (hipStream_t)stream, a->taps, a->ntaps_pad, a->decim,
                       (MmChan*)a->mm_chan);
    switch (a->decim / 8) {
    case 20: hipLaunchKernelGGL(mm1_image_kernel<20>, dim3((unsigned int)a->nch), dim3(64), 0, (hipStream_t)stream, a->taps, a->ntaps_pad, a->decim,
                                (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    case 24: hipLaunchKernelGGL(mm1_image_kernel<24>, dim3((unsigned int)a->nch), dim3(64), 0, (hipStream_t)stream, a->taps, a->ntaps_pad, a->decim,
                                (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    case 25: hipLaunchKernelGGL(mm1_image_kernel<25>, dim3((unsigned int)a->nch), dim3(64), 0, (hipStream_t)stream, a->taps, a->ntaps_pad, a->decim,
                                (const MmChan*)a->mm_chan, (u4m_t*)a->mm_img); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

extern "C" int acg_launch_fir_mm1(const FirArgs* a, void* stream)
{
    if (!acg_fir_mm1_takes(a)) return (int)hipErrorInvalidValue;
    switch (a->decim / 8) {
    case 20: return launch_mm1<20>(a, (hipStream_t)stream);
    case 24: return launch_mm1<24>(a, (hipStream_t)stream);
    case 25: return launch_mm1<25>(a, (hipStream_t)stream);
    }
    return (int)hipErrorInvalidValue;
}
