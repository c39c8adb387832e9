// msk.hip -- demodMSK() (msk.c:67-137) + putbit() (msk.c:53-63) + the decodeAcars() framing FSM
// (acars.c:239-375) for thousands of channels on gfx950.
//
// The loop is a strict serial recurrence per channel (the PLL feeds the VCO that produced the
// sample it is fed from; the framing FSM writes MskDf / MskS back into it, acars.c:242,259,274),
// so the only parallelism is across channels: one lane per channel, one wave per workgroup,
// state resident in HBM between launches.  Arithmetic follows the reference's C promotions
// operation for operation (f64 VCO/PLL, f32 clock and filter, f64 divide): this translation unit
// is compiled with -ffp-contract=off so no multiply-add is fused that the reference's IEEE build
// keeps separate.
//
// SIMT scheduling: a bit is decided every 5 or 6 input samples and the per-bit work (matched
// filter, normalise, PLL, FSM) costs about as much as one sample's VCO + sincos.  Testing for a
// bit after every sample would make the wave execute the bit branch on nearly every sample (some
// lane always fires).  Instead each lane runs up to 6 samples until ITS bit fires, then all lanes
// run the bit branch together: same per-channel operation sequence, ~1/3 fewer issued
// instructions.  Lanes drift apart by a few samples; each stops at its own len.
#include <hip/hip_runtime.h>
#include "acg_internal.h"

#define FLEN 11
#define MFLTOVER 12

// acars.c:22-27
#define SYN 0x16
#define SOH 0x01
#define ETX 0x83
#define ETB 0x97
#define DLE 0x7f
#define MAXPERR 3

enum { WSYN = 0, SYN2, SOH1, TXT, CRC1, CRC2, END };   // acarsdec.h:88

#define K_TWOPI   (2.0 * 3.14159265358979323846)
#define K_3PI2    (3 * 3.14159265358979323846 / 2.0)
#define K_VCO     (1800.0 / 12500 * 2.0 * 3.14159265358979323846)      // msk.c:81

struct Lane {
    double phi, df, lvlsum;
    float clk;
    int bitcount;
    unsigned int S, idx;
    int nbits, astate, blen, berr;
    unsigned int outbits, crc0;
    long long nbit_total;
};

__device__ __forceinline__ void reset_acars(Lane& L)          // acars.c:239-244
{
    L.astate = WSYN;
    L.df = 0;
    L.nbits = 1;
}

__device__ __forceinline__ void put_frame(Lane& L, const MskArgs& a, int ch, unsigned char crc1,
                                          const unsigned char* txt, long long sample_index)
{
    // acars.c:350-369: queue the block.  lvl = 10*log10(MskLvlSum/MskBitCount) is taken on the host
    // from the two operands (same libm call as the reference).
    const unsigned int slot = atomicAdd(a.frame_count, 1u);
    if (slot < a.frame_cap) {
        AcgFrameRec* f = a.frames + slot;
        f->chn = ch;
        f->len = L.blen;
        f->err = L.berr;
        f->bitcount = L.bitcount;
        f->lvlsum = L.lvlsum;
        f->end_bit = L.nbit_total;
        f->end_sample = sample_index;
        f->crc[0] = (unsigned char)L.crc0;
        f->crc[1] = crc1;
        const uint4* s = (const uint4*)txt;
        uint4* d = (uint4*)f->txt;
        const int nv = (L.blen + 15) >> 4;
        for (int i = 0; i < nv; ++i) d[i] = s[i];
    }
    L.astate = END;
    L.nbits = 8;
}

__device__ __forceinline__ void decode_acars(Lane& L, const MskArgs& a, int ch, unsigned char* txt,
                                             long long sample_index)
{
    const unsigned int r = L.outbits & 0xffu;
    switch (L.astate) {
    case WSYN:                                                 // acars.c:252-265
        if (r == SYN) { L.astate = SYN2; L.nbits = 8; return; }
        if (r == (0xffu & ~SYN)) { L.S ^= 2; L.astate = SYN2; L.nbits = 8; return; }
        L.nbits = 1;
        return;
    case SYN2:                                                 // acars.c:267-279
        if (r == SYN) { L.astate = SOH1; L.nbits = 8; return; }
        if (r == (0xffu & ~SYN)) { L.S ^= 2; L.nbits = 8; return; }
        reset_acars(L);
        return;
    case SOH1:                                                 // acars.c:281-301
        if (r == SOH) {
            L.astate = TXT;
            L.blen = 0;
            L.berr = 0;
            L.nbits = 8;
            L.lvlsum = 0;
            L.bitcount = 0;
            return;
        }
        reset_acars(L);
        return;
    case TXT:                                                  // acars.c:303-341
        txt[L.blen] = (unsigned char)r;
        L.blen++;
        if ((__popc(r) & 1) == 0) {
            L.berr++;
            if (L.berr > MAXPERR + 1) { reset_acars(L); return; }
        }
        if (r == ETX || r == ETB) { L.astate = CRC1; L.nbits = 8; return; }
        if (L.blen > 20 && r == DLE) {
            L.blen -= 3;
            L.crc0 = txt[L.blen];
            const unsigned char c1 = txt[L.blen + 1];
            L.astate = CRC2;
            put_frame(L, a, ch, c1, txt, sample_index);
            return;
        }
        if (L.blen > 240) { reset_acars(L); return; }
        L.nbits = 8;
        return;
    case CRC1:                                                 // acars.c:343-347
        L.crc0 = r;
        L.astate = CRC2;
        L.nbits = 8;
        return;
    case CRC2:                                                 // acars.c:348-369
        put_frame(L, a, ch, (unsigned char)r, txt, sample_index);
        return;
    default:                                                   // END, acars.c:370-373
        reset_acars(L);
        L.nbits = 8;
        return;
    }
}

__global__ __launch_bounds__(ACG_WG_MSK) void msk_demod_kernel(const MskArgs a)
{
    __shared__ float2 ring[FLEN][ACG_WG_MSK];      // inb[] of 64 channels, one column per lane
    __shared__ float hs[FLEN * MFLTOVER + 1];

    const int tid = threadIdx.x;
    for (int i = tid; i < FLEN * MFLTOVER + 1; i += ACG_WG_MSK) hs[i] = a.h[i];

    const int ch = blockIdx.x * ACG_WG_MSK + tid;
    const bool active = ch < a.nch;
    const int chc = active ? ch : a.nch - 1;
    AcgChan* st = a.st + chc;

    Lane L;
    L.phi = st->phi; L.df = st->df; L.lvlsum = st->lvlsum;
    L.clk = st->clk; L.bitcount = st->bitcount; L.S = st->S; L.idx = st->idx;
    L.nbits = st->nbits; L.astate = st->astate; L.blen = st->blen; L.berr = st->berr;
    L.outbits = st->outbits; L.crc0 = st->crc0; L.nbit_total = st->nbit_total;
    const long long samp0 = st->nsamp_total;
#pragma unroll
    for (int j = 0; j < FLEN; ++j) ring[j][tid] = make_float2(st->inb[2 * j], st->inb[2 * j + 1]);
    __syncthreads();

    const float* __restrict__ dm = a.dm + (size_t)chc * a.dm_pitch;
    unsigned char* txt = a.txt + (size_t)chc * 256;
    float2* bits = a.bits ? a.bits + (size_t)chc * a.bit_cap : nullptr;
    const int len = active ? a.len : 0;
    int nb = 0;
    int n = 0;
    unsigned int idx = L.idx;
    double p = L.phi;

    while (n < len) {
        bool fired = false;
        double s = 0;
#pragma unroll 1
        for (int u = 0; u < 6 && !fired && n < len; ++u) {
            // VCO, msk.c:81-83
            s = K_VCO + L.df;
            p += s;
            if (p >= K_TWOPI) p -= K_TWOPI;
            // mixer, msk.c:86-91: in * cexp(-j p) in double, narrowed to float complex
            double sn, cs;
            sincos(p, &sn, &cs);
            const double in = (double)dm[n];
            ring[idx][tid] = make_float2((float)(in * cs), (float)(in * (-sn)));
            idx = (idx + 1 == FLEN) ? 0 : idx + 1;
            // bit clock, msk.c:95-96
            L.clk = (float)((double)L.clk + s);
            ++n;
            fired = (double)L.clk >= K_3PI2 - s / 2;
        }
        if (fired) {
            L.clk = (float)((double)L.clk - K_3PI2);                      // msk.c:100
            // matched filter, msk.c:103-107
            int o = (int)(MFLTOVER * ((double)L.clk / s + 0.5));
            if (o > MFLTOVER) o = MFLTOVER;
            if (o < 0) o = 0;      // memory safety only: the reference indexes h[] out of bounds here
            float vr = 0.f, vi = 0.f;
            unsigned int k = idx;
#pragma unroll
            for (int j = 0; j < FLEN; ++j, o += MFLTOVER) {
                const float hh = hs[o];
                const float2 x = ring[k][tid];
                vr = vr + hh * x.x;
                vi = vi + hh * x.y;
                k = (k + 1 == FLEN) ? 0 : k + 1;
            }
            // normalise, msk.c:110-113
            const float lvl = (float)__dsqrt_rn((double)vr * (double)vr + (double)vi * (double)vi);
            const double d = (double)lvl + 1e-8;
            vr = (float)((double)vr / d);
            vi = (float)((double)vi / d);
            L.lvlsum += (double)(lvl * lvl / 4);
            L.bitcount++;
            // decision + phase detector, msk.c:115-121
            float vo;
            double dphi;
            if (L.S & 1) {
                vo = vi;
                dphi = (vo >= 0) ? -(double)vr : (double)vr;
            } else {
                vo = vr;
                dphi = (vo >= 0) ? (double)vi : -(double)vi;
            }
            const float sv = (L.S & 2) ? -vo : vo;                         // msk.c:122-126
            if (bits && nb < a.bit_cap) bits[nb] = make_float2(sv, lvl);
            ++nb;
            // putbit, msk.c:53-63
            L.outbits = (L.outbits >> 1) & 0x7fu;
            if (sv > 0) L.outbits |= 0x80u;
            L.nbits--;
            if (L.nbits <= 0) decode_acars(L, a, ch, txt, samp0 + n - 1);
            L.nbit_total++;
            L.S++;
            // PLL filter, msk.c:130 (float constants promoted to double)
            L.df = (double)0.52f * L.df + (1.0 - (double)0.52f) * (double)38e-4f * dphi;
        }
    }

    if (active) {
        st->phi = p; st->df = L.df; st->lvlsum = L.lvlsum;
        st->clk = L.clk; st->bitcount = L.bitcount; st->S = L.S; st->idx = idx;
        st->nbits = L.nbits; st->astate = L.astate; st->blen = L.blen; st->berr = L.berr;
        st->outbits = L.outbits; st->crc0 = L.crc0; st->nbit_total = L.nbit_total;
        st->nsamp_total = samp0 + len;
#pragma unroll
        for (int j = 0; j < FLEN; ++j) {
            const float2 x = ring[j][tid];
            st->inb[2 * j] = x.x;
            st->inb[2 * j + 1] = x.y;
        }
        a.nbits_out[ch] = nb;
    }
}

extern "C" int acg_launch_msk(const MskArgs* a, void* stream)
{
    const unsigned int grid = (unsigned int)((a->nch + ACG_WG_MSK - 1) / ACG_WG_MSK);
    hipLaunchKernelGGL(msk_demod_kernel, dim3(grid), dim3(ACG_WG_MSK), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
