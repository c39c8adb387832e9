// msk_common.h -- device helpers shared by the demodulator kernels (msk.hip: one wave per channel group; msk2.hip: the
// same stream split over two waves): constants of msk.c / acars.c, the framing state machine (acars.c:239-375), the mixer's
// sin/cos, the loop's f64 quotients and square root.  Included by translation units built with -ffp-contract=off.
#pragma once
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

// soh_slot = &AcgChan::soh32 of the channel: the SOH stamp lives in the channel's state record in HBM, written and read by the
// group leader on these rare paths only.  (As a field of Lane it was one more value alive across the per-bit loop: the
// register allocator of this kernel's recipe started spilling -- ten scratch accesses in the loop, the demodulator alone 0.750
// -> 0.779 us per bit and 38 % slower beside the down-converter at 4096 channels; round 5, GPU calls 2-4.)
__device__ __forceinline__ void put_frame(Lane& L, const MskArgs& a, int ch, unsigned char crc1,
                                          const unsigned char* txt, long long sample_index, bool leader, unsigned int* soh_slot)
{
    // acars.c:350-369: queue the block.  lvl = 10*log10(MskLvlSum/MskBitCount) is taken on the host
    // from the two operands (same libm call as the reference).
    // the queue is a ring with a monotonic counter: the host consumes behind it (acg_collect_frames)
    if (leader) {
        const unsigned int slot = atomicAdd(a.frame_count, 1u);
        AcgFrameRec* f = a.frames + (slot & (a.frame_cap - 1));   // frame_cap is a power of two: consistent across the counter's wrap
        f->chn = ch;
        f->len = L.blen;
        f->err = L.berr;
        f->bitcount = L.bitcount;
        f->lvlsum = L.lvlsum;
        f->end_bit = L.nbit_total;
        f->end_sample = sample_index;
        f->crc[0] = (unsigned char)L.crc0;
        f->crc[1] = crc1;
        f->status = 0;
        f->pad[0] = 0;
        // where the reference stamps the block's time: at its SOH (acars.c:290 gettimeofday(&blk->tv)), as a distance back from
        // the closing bit (a block is < 2^16 samples long: the 32-bit difference is exact across the index's wrap)
        f->soh_back = (int)((unsigned int)sample_index - *soh_slot);
        const uint4* s = (const uint4*)txt;
        uint4* d = (uint4*)f->txt;
        const int nv = (L.blen + 15) >> 4;
        for (int i = 0; i < nv; ++i) d[i] = s[i];
    }
    L.astate = END;
    L.nbits = 8;
}

__device__ __forceinline__ void decode_acars(Lane& L, const MskArgs& a, int ch, unsigned char* txt,
                                             long long sample_index, bool leader, unsigned int* soh_slot)
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
            if (leader) *soh_slot = (unsigned int)sample_index;    // acars.c:290: the block's time stamp is taken here
            return;
        }
        reset_acars(L);
        return;
    case TXT:                                                  // acars.c:303-341
        if (leader) txt[L.blen] = (unsigned char)r;
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
            put_frame(L, a, ch, c1, txt, sample_index, leader, soh_slot);
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
        put_frame(L, a, ch, (unsigned char)r, txt, sample_index, leader, soh_slot);
        return;
    default:                                                   // END, acars.c:370-373
        reset_acars(L);
        L.nbits = 8;
        return;
    }
}

// sin/cos of x in [0, 2*pi) (any moderate |x| works): Cody-Waite reduction by pi/2 with a two-term
// constant, then the classic minimax kernels on |r| <= pi/4 (coefficients: fdlibm k_sin.c/k_cos.c,
// Sun Microsystems 1993, freely distributable; < 1 ulp).  The reference calls glibc's cexp (also
// < 1 ulp, different algorithm): results agree to the last bit except in rare last-place cases,
// and only the float-rounded product in*cos / in*sin is kept (msk.c:90).
// acc*z + C with the constant as an SGPR operand: one v_fma_f64.  (Left to the compiler, every Horner
// step becomes v_mov_b64 + v_fmac_f64 because the constant has to be copied into the accumulator.)
__device__ __forceinline__ double fma_zc(double acc, double z, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(acc), "v"(z), "s"(c));
    return r;
}

__device__ __forceinline__ void sincos_2pi(double x, double* sn, double* cs)
{
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01);           // x * 2/pi
    const int q = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);               // pi/2, high 33 bits
    r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);                      // pi/2, tail
    const double z = r * r;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = fma_zc(S6, z, S5);
    ps = fma_zc(ps, z, S4);
    ps = fma_zc(ps, z, S3);
    ps = fma_zc(ps, z, S2);
    const double v = z * r;
    const double s = __builtin_fma(v, fma_zc(ps, z, S1), r);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = fma_zc(C6, z, C5);
    pc = fma_zc(pc, z, C4);
    pc = fma_zc(pc, z, C3);
    pc = fma_zc(pc, z, C2);
    pc = fma_zc(pc, z, C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * (z * pc));
    // quadrant
    const double ss = (q & 1) ? c : s;
    const double cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// The mixer's sin/cos as the product ships it: a 128-entry table of (cos, sin)(j * 2 pi / 128) in LDS (correctly rounded
// doubles, acg_host_sincos_table) and a rotation by the small remainder |r| <= pi / 128,
//     cos p = cj + (cj (cos r - 1) - sj sin r),    sin p = sj + (sj (cos r - 1) + cj sin r),
// with sin r - r and cos r - 1 from three-term series (next terms < 4e-19 relative).  23 vector instructions and one LDS
// read instead of ~40 for sincos_2pi above (no quadrant logic, shorter polynomials) on a loop that is bound by the number
// of instructions it issues.  Error <= 2.1 ulp (the table entry's and the last addition's rounding dominate)
// where sincos_2pi has < 1; what the demodulator keeps is (float)(in * cos), (float)(in * -sin) (msk.c:90), and on 4e7 random
// phases those are identical to glibc cexp's for both (tests/sincos_model.c: the same operations on the CPU, in the CPU suite).
__device__ __forceinline__ void sincos_tab(double x, const double* __restrict__ tab /* LDS, [128][2] */, double* sn, double* cs)
{
    const double kd = __builtin_rint(x * (6.36619772367581382433e-01 * 32.0));   // x * 128 / (2 pi)
    const int q = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00 / 32.0, x);        // 2 pi / 128, high 33 bits (exact products: kd <= 128)
    r = __builtin_fma(-kd, 6.07710050650619224932e-11 / 32.0, r);               // tail
    const double2 e = *(const double2*)(tab + 2 * (q & (ACG_SINCOS_N - 1)));
    const double cj = e.x, sj = e.y;
    const double z = r * r;
#ifdef ACG_MSK_AB_FMA3
    // A/B build: the Horner steps as three-address v_fma_f64 with the constants as scalar / resident vector operands (left to the
    // compiler each becomes v_mov_b64 + v_fmac_f64: the accumulator form overwrites the constant it starts from)
    double ps, pc;
    {
        const double c2 = 8.33333333333333333333e-03, c5 = 4.16666666666666666667e-02;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ps) : "v"(z), "s"(-1.98412698412698412698e-04), "v"(c2));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ps) : "v"(z), "v"(ps), "s"(-1.66666666666666666667e-01));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(pc) : "v"(z), "s"(-1.38888888888888888889e-03), "v"(c5));
    }
    const double sm = (r * z) * ps;                                              // sin r - r
#else
    double ps = __builtin_fma(z, -1.98412698412698412698e-04, 8.33333333333333333333e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666666667e-01);
    const double sm = (r * z) * ps;                                              // sin r - r
    double pc = __builtin_fma(z, -1.38888888888888888889e-03, 4.16666666666666666667e-02);
#endif
    pc = __builtin_fma(z, pc, -0.5);
    const double cm1 = z * pc;                                                   // cos r - 1
    const double sr = r + sm;
    const double u = __builtin_fma(cj, cm1, -(sj * sr));
    const double w = __builtin_fma(sj, cm1, cj * sr);
    *cs = cj + u;
    *sn = sj + w;
}

// n0 / d and n1 / d, both correctly rounded, for d in [1e-8, ~1e2] and |n| < ~1e2 or n == +0 (the matched filter's
// sum starts from +0 and so is never -0, the one numerator whose quotient would come out as +0 here): see the call site.
// acg_selftest_div2 compares it with the compiler's IEEE division on the device.
__device__ __forceinline__ void div2_shared_rcp(double n0, double n1, double d, double* q0, double* q1)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    const double a = n0 * r, b = n1 * r;
    *q0 = __builtin_fma(__builtin_fma(-d, a, n0), r, a);
    *q1 = __builtin_fma(__builtin_fma(-d, b, n1), r, b);
}

// one quotient, same construction (the tap phase clk / s, msk.c:103: s ~ 0.9, |clk| < 1)
__device__ __forceinline__ double div1_rcp(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    const double a = n * r;
    return __builtin_fma(__builtin_fma(-d, a, n), r, a);
}

// sqrt(x), correctly rounded, for x = 0 or x in [1e-100, 1e100]: the compiler's own expansion (rsq, one coupled
// Newton step on (g, h) = (sqrt, 1 / (2 sqrt)), two remainder steps) without the exponent scaling it wraps around
// it for arguments below 2^-767.  |v|^2 (msk.c:110 through cabsf) is a sum of two squares of floats.
__device__ __forceinline__ double sqrt_rn_positive(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return g;
}

__device__ __forceinline__ double sqrt_rn_midrange(double x)
{
    return x == 0.0 ? x : sqrt_rn_positive(x);
}

// (float)sqrt(x) for x = a*a + b*b of two floats: such an x is 0 or >= 2^-298, and everything below 2^-300 has the float
// square root 0 -- so the zero case is one v_max_f64 (sqrt(1e-300) = 1e-150 -> 0.0f) instead of compare + two selects,
// or worse a branch around the Newton steps: any branch inside the bit decision splits its basic block, and each split
// measured ~3 % per bit (the scheduler fills latency slots only within a block)
__device__ __forceinline__ float sqrtf_of_sum_of_squares(double x)
{
    return (float)sqrt_rn_positive(__builtin_fmax(x, 1e-300));
}

// msk.c:83 `if (p >= 2*M_PI) p -= 2*M_PI;` as compare + one select + one fma: fma(-1, 2pi, p) is p - 2pi
// with its single rounding, fma(-0.0, 2pi, p) is p itself (p + -0.0), and -1.0 / -0.0 differ in the high
// word only.  Same results, one instruction less than subtract + two-word select on the serial chain
// (measured: the three-level form "difference beside compare, then a two-word select" is 4 % slower per bit --
// the loop is bound by the number of instructions it issues, not by the depth of this chain).
__device__ __forceinline__ double wrap_2pi(double p)
{
#ifdef ACG_MSK_AB_WRAP3
    // A/B build (round 6, on msk_lean.hip's shorter period): the difference beside the compare, then a two-word select -- one
    // instruction more, one level less on the phase chain
    const double w = p - K_TWOPI;
    return p >= K_TWOPI ? w : p;
#else
    const double k = __hiloint2double(p >= K_TWOPI ? (int)0xBFF00000 : (int)0x80000000, 0);
    return __builtin_fma(k, K_TWOPI, p);
#endif
}

