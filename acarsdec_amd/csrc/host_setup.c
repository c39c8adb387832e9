/*
 * host_setup.c -- the host-side set-up arithmetic of the hot path, in plain C99 so that the
 * float/double/complex promotions are exactly the ones the reference's C implies.
 * Compile with -O2 -ffp-contract=off (no fast-math): these values feed the kernels and must be
 * bit-identical to what the reference computes on the host.
 *
 *   acg_rtl_choose_fc   <- rtl.c:131-168  chooseFc()
 *   acg_rtl_taps        <- rtl.c:283-286  the per-channel NCO*boxcar taps wf[]
 *   acg_host_msk_h      <- msk.c:44-48    the matched-filter prototype h[]
 *   acg_host_sincos_table  (msk.c:86-91)  table behind the device's cexp(-p*I)
 *   acg_host_level_db   <- acars.c:351    blk->lvl
 */
#include <math.h>
#include <complex.h>
#include <stdlib.h>
#include "acarsdec_amd.h"

#define FLENO (ACG_FLEN * 12 + 1)
#ifndef ACG_SINCOS_N
#define ACG_SINCOS_N 128
#endif

unsigned int acg_rtl_choose_fc(unsigned int *Fd, unsigned int nbch, int decim)
{
	const int rate = ACG_INTRATE * decim;          /* rtl.c:214 rtlInRate */
	const int guard = 2 * ACG_INTRATE;
	unsigned int i, j;
	int Fc;

	if (!Fd || nbch == 0)
		return 0;
	/* rtl.c:136-147 sorts ascending (any stable sort gives the same array) */
	for (i = 1; i < nbch; i++) {
		unsigned int v = Fd[i];
		for (j = i; j > 0 && Fd[j - 1] > v; j--)
			Fd[j] = Fd[j - 1];
		Fd[j] = v;
	}
	if (Fd[nbch - 1] - Fd[0] > (unsigned int)(rate - 2 * guard))       /* rtl.c:149 */
		return 0;
	/* rtl.c:154-165: walk down from just above the highest channel until every channel is
	 * inside the usable band, clear of DC, and not the mirror image of its lower neighbour */
	for (Fc = (int)(Fd[nbch - 1] + guard); (unsigned int)Fc > Fd[0] - guard; Fc--) {
		int ok = 1;
		for (i = 0; i < nbch && ok; i++) {
			const int off = abs(Fc - (int)Fd[i]);
			if (off > rate / 2 - guard || off < guard)
				ok = 0;
			else if (i > 0 && (unsigned int)Fc - Fd[i - 1] == Fd[i] - (unsigned int)Fc)
				ok = 0;
		}
		if (ok)
			break;
	}
	return (unsigned int)Fc;
}

int acg_rtl_taps(int Fr_hz, unsigned int Fc_hz, int decim, float *taps_out)
{
	int k;
	float AMFreq;

	if (!taps_out || decim < 1 || decim > ACG_MAXDECIM)
		return ACG_EINVAL;
	/* rtl.c:283: (int - float)/float in float, then *2.0*M_PI in double, stored as float */
	AMFreq = (Fr_hz - (float)Fc_hz) / (float)(ACG_INTRATE * decim) * 2.0 * M_PI;
	for (k = 0; k < decim; k++) {
		/* rtl.c:285: float complex / int, then / double, narrowed to float complex */
		const float complex w = cexpf(AMFreq * k * -I) / decim / 127.5;
		taps_out[2 * k] = crealf(w);
		taps_out[2 * k + 1] = cimagf(w);
	}
	return ACG_OK;
}

void acg_host_msk_h(float *h)
{
	int i;
	for (i = 0; i < FLENO; i++) {
		/* msk.c:46-47 */
		const float c = cosf(2.0 * M_PI * 600.0 / ACG_INTRATE / 12 * (i - (FLENO - 1) / 2));
		h[i] = c < 0 ? 0 : c;
	}
}

/* The demodulator's mixer (msk.c:86-91, cexp(-p*I)) on the device: (cos, sin) of j * 2 pi / 128, j = 0..127 -- the table
 * behind msk.hip sincos_tab().  The first octant (j = 0..16) is embedded as correctly rounded doubles (generated offline with
 * 80-digit decimal arithmetic: no dependence on the host's long double); the other seven octants follow by symmetry, so the
 * axis entries are exactly 0 and +-1 and cos/sin of mirrored angles are the same doubles. */
static const double acg_octant[17][2] = {
	{0x1.0000000000000p+0, 0x0.0p+0},
	{0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5},
	{0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4},
	{0x1.fa7557f08a517p-1, 0x1.2c8106e8e613ap-3},
	{0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3},
	{0x1.f0a7efb9230d7p-1, 0x1.f19f97b215f1bp-3},
	{0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2},
	{0x1.e212104f686e5p-1, 0x1.58f9a75ab1fddp-2},
	{0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},
	{0x1.ced7af43cc773p-1, 0x1.b5d1009e15cc0p-2},
	{0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2},
	{0x1.b728345196e3ep-1, 0x1.073879922ffeep-1},
	{0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1},
	{0x1.9b3e047f38741p-1, 0x1.30ff7fce17035p-1},
	{0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1},
	{0x1.7b5df226aafafp-1, 0x1.57d69348ceca0p-1},
	{0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1},
};

void acg_host_sincos_table(double *tab)
{
	int j;
	for (j = 0; j < ACG_SINCOS_N; j++) {
		int k = j & 63;                      /* angle within a half turn, in units of pi / 64 */
		double c, s;
		if (k > 32)
			k = 64 - k;                  /* cos(pi - t) = -cos t, sin(pi - t) = sin t */
		if (k <= 16) {
			c = acg_octant[k][0];
			s = acg_octant[k][1];
		} else {                             /* cos(pi/2 - t) = sin t */
			c = acg_octant[32 - k][1];
			s = acg_octant[32 - k][0];
		}
		if ((j & 63) > 32)
			c = -c;
		if (j >= 64) {                       /* + pi: both change sign */
			c = -c;
			s = -s;
		}
		tab[2 * j] = c + 0.0;                /* (-0.0 -> +0.0 on the axes) */
		tab[2 * j + 1] = s + 0.0;
	}
}

float acg_host_level_db(double lvlsum, int bitcount)
{
	return 10 * log10(lvlsum / bitcount);          /* acars.c:351, double narrowed to float */
}

/* syndrom.h:15-49 (reflected CRC-CCITT byte table, poly 0x8408) and syndrom.h:52-295 (entry i + 8k =
 * remainder of a single wrong bit i in the byte followed by k more bytes), generated from their
 * definitions.  crc[256], synd[8*242]. */
/* nk rows of 8 syndromes.  The reference's table has 242 rows (syndrom.h:52-295) but its block repair indexes
 * row len - pr + 1 = 242 for a 241-byte text with a flagged first byte (acars.c:48,77): an out-of-bounds read
 * there.  The device table carries that row too (same definition), so the result is defined and in bounds. */
void acg_host_crc_tables_n(unsigned short *crc, unsigned short *synd, int nk)
{
	int i, k, j;
	for (i = 0; i < 256; i++) {
		unsigned short c = (unsigned short)i;
		for (k = 0; k < 8; k++)
			c = (c & 1) ? (unsigned short)((c >> 1) ^ 0x8408) : (unsigned short)(c >> 1);
		crc[i] = c;
	}
	for (k = 0; k < nk; k++)
		for (i = 0; i < 8; i++) {
			unsigned short s = crc[1 << i];
			for (j = 0; j < k; j++)
				s = (unsigned short)((s >> 8) ^ crc[s & 0xff]);
			synd[8 * k + i] = s;
		}
}

void acg_host_crc_tables(unsigned short *crc, unsigned short *synd)
{
	acg_host_crc_tables_n(crc, synd, 242);
}

/* soapy.c:163-166: oscillator[ind] = cexpf(-j*AMFreq*ind)/rateMult with ch->Fr a float (acarsdec.h:70) */
int acg_soapy_taps(float Fr_hz, int freq_hz, int decim, float *taps_out)
{
	int k;
	float AMFreq;
	if (!taps_out || decim < 1 || decim > ACG_MAXDECIM_SAMPLES)
		return ACG_EINVAL;
	AMFreq = (Fr_hz - (float)freq_hz) / (float)(ACG_INTRATE * decim) * 2.0 * M_PI;
	for (k = 0; k < decim; k++) {
		const float complex w = cexpf(AMFreq * k * -I) / decim;
		taps_out[2 * k] = crealf(w);
		taps_out[2 * k + 1] = cimagf(w);
	}
	return ACG_OK;
}

/* sdrplay.c:160-164 (fixed SDRPLAY_MULT = 160; the phase is a double there, unlike soapy.c) */
int acg_sdrplay_taps(float Fr_hz, unsigned int Fc_hz, float *taps_out)
{
	const int decim = 160;
	int k;
	double phase;
	if (!taps_out)
		return ACG_EINVAL;
	phase = (Fr_hz - (float)Fc_hz) / (float)(ACG_INTRATE * decim) * 2.0 * M_PI;
	for (k = 0; k < decim; k++) {
		const float complex w = cexpf(phase * k * -I) / decim;
		taps_out[2 * k] = crealf(w);
		taps_out[2 * k + 1] = cimagf(w);
	}
	return ACG_OK;
}

/* air.c:62: centre frequency without the R820T IF-filter branch (taken only at exactly 5 Msps) */
unsigned int acg_airspy_choose_fc(unsigned int minF_hz, unsigned int maxF_hz)
{
	return ((maxF_hz + minF_hz) / 2 + ACG_INTRATE / 2) / ACG_INTRATE * ACG_INTRATE;
}

/* air.c:278-285: channels are mixed down from around Fs/4 of the real spectrum; the phase
 * accumulates in double with wraps; Fc - Fr + Fs/4 is evaluated in unsigned arithmetic */
int acg_airspy_taps(int Fr_hz, int Fc_hz, unsigned int inrate, float *taps_out)
{
	const unsigned int decim = inrate / ACG_INTRATE;
	unsigned int i;
	double AMFreq, Ph;
	if (!taps_out || decim < 1 || decim > ACG_MAXDECIM_SAMPLES || decim * ACG_INTRATE != inrate)
		return ACG_EINVAL;
	AMFreq = 2.0 * M_PI * (double)(Fc_hz - Fr_hz + inrate / 4) / (double)(inrate);
	for (i = 0, Ph = 0; i < decim; i++) {
		const float complex w = cexpf(Ph * -I) / decim;
		taps_out[2 * i] = crealf(w);
		taps_out[2 * i + 1] = cimagf(w);
		Ph += AMFreq;
		if (Ph > 2.0 * M_PI) Ph -= 2.0 * M_PI;
		if (Ph < -2.0 * M_PI) Ph += 2.0 * M_PI;
	}
	return ACG_OK;
}
