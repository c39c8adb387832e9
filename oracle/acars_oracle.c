/*
 * acars_oracle.c -- CPU restatement of acarsdec's per-channel DSP hot path
 * (rtl.c down-converter + msk.c demodulator + the acars.c framing FSM that
 * feeds back into the MSK loop).
 *
 * TEST INFRASTRUCTURE ONLY (see acars_oracle.h).  Build with
 *   gcc -O2 -ffp-contract=off   (IEEE, no FMA contraction, no -ffast-math)
 * so that every rounding below is the one the C expression in the reference
 * implies.  libm calls (cexp, cexpf, cosf, cabsf, log10) are the same libm
 * calls the reference makes.
 *
 * Type promotions are the whole point of this file; each statement notes the
 * reference line it restates and the promotion C applies there.
 */
#define _GNU_SOURCE
#include <math.h>
#include <complex.h>
#include <string.h>
#include <stdlib.h>
#include "acars_oracle.h"

/* ------------------------------------------------------------------ */
/* tables                                                             */
/* ------------------------------------------------------------------ */

static float g_h[ORC_FLENO];
static int g_h_ready;
static unsigned short g_crc_tab[256];
static int g_crc_ready;

/* msk.c:44-48: h[i] = cosf(2*pi*600/INTRATE/MFLTOVER*(i-(FLENO-1)/2)), negative lobes clipped.
 * The argument is a double expression (2.0*M_PI*600.0/INTRATE/MFLTOVER*(int)) narrowed by cosf's
 * float parameter. */
void orc_msk_h(float *h)
{
	int i;
	for (i = 0; i < ORC_FLENO; i++) {
		h[i] = cosf(2.0 * M_PI * 600.0 / ORC_INTRATE / ORC_MFLTOVER * (i - (ORC_FLENO - 1) / 2));
		if (h[i] < 0)
			h[i] = 0;
	}
}

static void crc_init(void)
{
	/* syndrom.h:15-48 is the standard reflected CRC-CCITT (poly 0x8408) byte table */
	int i, k;
	for (i = 0; i < 256; i++) {
		unsigned short c = (unsigned short)i;
		for (k = 0; k < 8; k++)
			c = (c & 1) ? (unsigned short)((c >> 1) ^ 0x8408) : (unsigned short)(c >> 1);
		g_crc_tab[i] = c;
	}
	g_crc_ready = 1;
}

/* syndrom.h:49 update_crc */
unsigned short orc_crc_update(unsigned short crc, unsigned char c)
{
	if (!g_crc_ready)
		crc_init();
	return (unsigned short)((crc >> 8) ^ g_crc_tab[(crc ^ c) & 0xff]);
}

static int popcount8(unsigned char r)
{
	int n = 0;
	while (r) {
		n += r & 1;
		r >>= 1;
	}
	return n;      /* syndrom.h:4-13 numbits[] */
}

/* ------------------------------------------------------------------ */
/* rtl.c front end                                                    */
/* ------------------------------------------------------------------ */

/* rtl.c:283-286.  ch->Fr is an int (acarsdec.h:63), Fc unsigned, rtlInRate int. */
void orc_rtl_taps(int Fr, int Fc, int M, float *wf)
{
	int rtlInRate = ORC_INTRATE * M;                 /* rtl.c:214 */
	int ind;
	float AMFreq;

	/* int - float -> float ; / float -> float ; * 2.0 * M_PI -> double ; narrowed to float */
	AMFreq = (Fr - (float)(unsigned int)Fc) / (float)(rtlInRate) * 2.0 * M_PI;
	for (ind = 0; ind < M; ind++) {
		/* cexpf(float complex) / int -> float complex ; / 127.5 -> double complex -> float complex */
		float complex w = cexpf(AMFreq * ind * -I) / M / 127.5;
		wf[2 * ind] = crealf(w);
		wf[2 * ind + 1] = cimagf(w);
	}
}

/* rtl.c:131-168 */
int orc_choose_fc(unsigned int *Fd, unsigned int nbch, int rtlInRate)
{
	unsigned int n;
	int ne;
	int Fc;

	do {                                             /* rtl.c:136-147 bubble sort */
		ne = 0;
		for (n = 0; n + 1 < nbch; n++) {
			if (Fd[n] > Fd[n + 1]) {
				unsigned int t = Fd[n + 1];
				Fd[n + 1] = Fd[n];
				Fd[n] = t;
				ne = 1;
			}
		}
	} while (ne);

	if ((Fd[nbch - 1] - Fd[0]) > (unsigned int)(rtlInRate - 4 * ORC_INTRATE))   /* rtl.c:149 */
		return 0;

	/* rtl.c:154: the loop variable is a signed int compared against unsigned expressions */
	for (Fc = Fd[nbch - 1] + 2 * ORC_INTRATE; (unsigned int)Fc > Fd[0] - 2 * ORC_INTRATE; Fc--) {
		for (n = 0; n < nbch; n++) {
			if (abs(Fc - (int)Fd[n]) > rtlInRate / 2 - 2 * ORC_INTRATE)
				break;
			if (abs(Fc - (int)Fd[n]) < 2 * ORC_INTRATE)
				break;
			if (n > 0 && Fc - Fd[n - 1] == Fd[n] - Fc)
				break;
		}
		if (n == nbch)
			break;
	}
	return Fc;
}

/* rtl.c:332-354, one channel.  D accumulates sequentially in float; the complex
 * product is (ac-bd) + (ad+bc)i with each operation rounded to float. */
void orc_fir_u8(const uint8_t *iq, size_t nout, int M, int ntaps, const float *wf, float *dm)
{
	size_t m;
	for (m = 0; m < nout; m++) {
		const uint8_t *p = iq + 2 * (size_t)M * m;
		float Dr = 0, Di = 0;
		int ind;
		for (ind = 0; ind < ntaps; ind++) {
			float r = (float)p[2 * ind] - 127.37f;           /* rtl.c:338 */
			float g = (float)p[2 * ind + 1] - 127.37f;       /* rtl.c:339 */
			float wr = wf[2 * ind], wi = wf[2 * ind + 1];
			float pr = r * wr - g * wi;                      /* rtl.c:351 vb*wf */
			float pi = r * wi + g * wr;
			Dr = Dr + pr;                                    /* D += */
			Di = Di + pi;
		}
		dm[m] = cabsf(Dr + Di * I);                              /* rtl.c:353 */
	}
}

/* ------------------------------------------------------------------ */
/* soapy.c front end (CS16 samples, windows may straddle read buffers)  */
/* ------------------------------------------------------------------ */

/* soapy.c:163-166.  ch->Fr is a float here (acarsdec.h:70), freq an int, soapyInRate an int. */
void orc_soapy_taps(float Fr, int freq, int M, float *osc)
{
	int soapyInRate = ORC_INTRATE * M;               /* soapy.c:89 */
	int ind;
	float AMFreq;

	AMFreq = (Fr - (float)freq) / (float)(soapyInRate) * 2.0 * M_PI;
	for (ind = 0; ind < M; ind++) {
		float complex w = cexpf(AMFreq * ind * -I) / M;          /* float complex / int */
		osc[2 * ind] = crealf(w);
		osc[2 * ind + 1] = cimagf(w);
	}
}

/* soapy.c:232-254 for one channel over a whole stream.  D is carried across read buffers
 * (ch->D, current_index), so the result does not depend on how the stream was cut into reads:
 * every window is the same sequential sum.  Per term: float complex product, /32768.0 in
 * double, added to D in double, narrowed to float complex. */
void orc_fir_cs16(const int16_t *iq, size_t nout, int M, const float *osc, float *dm)
{
	size_t m;
	for (m = 0; m < nout; m++) {
		const int16_t *p = iq + 2 * (size_t)M * m;
		float Dr = 0, Di = 0;
		int ind;
		for (ind = 0; ind < M; ind++) {
			float r = (float)p[2 * ind];                     /* soapy.c:238 */
			float g = (float)p[2 * ind + 1];                 /* soapy.c:239 */
			float wr = osc[2 * ind], wi = osc[2 * ind + 1];
			float pr = r * wr - g * wi;                      /* soapy.c:241 v * oscillator */
			float pi = r * wi + g * wr;
			Dr = (float)((double)Dr + (double)pr / 32768.0);
			Di = (float)((double)Di + (double)pi / 32768.0);
		}
		dm[m] = cabsf(Dr + Di * I);                              /* soapy.c:243 */
	}
}

/* ------------------------------------------------------------------ */
/* sdrplay.c front end (split int16 planes, fixed 2.0 Msps)             */
/* ------------------------------------------------------------------ */

/* sdrplay.c:160-164.  correctionPhase is a DOUBLE here (a float in soapy.c): the phase of tap
 * `ind` is the double product narrowed by cexpf's float complex parameter. */
void orc_sdrplay_taps(float Fr, unsigned int Fc, float *osc)
{
	const int M = 160;                               /* sdrplay.c:35 SDRPLAY_MULT */
	int ind;
	double correctionPhase = (Fr - (float)Fc) / (float)(ORC_INTRATE * M) * 2.0 * M_PI;
	for (ind = 0; ind < M; ind++) {
		float complex w = cexpf(correctionPhase * ind * -I) / M;
		osc[2 * ind] = crealf(w);
		osc[2 * ind + 1] = cimagf(w);
	}
}

/* sdrplay.c:215-236 for one channel over a whole stream (D and the index are carried across
 * callbacks, so the cut into callbacks does not matter): float complex MAC, dm = cabsf(D)/4. */
void orc_fir_split16(const int16_t *xi, const int16_t *xq, size_t nout, int M, const float *osc, float *dm)
{
	size_t m;
	for (m = 0; m < nout; m++) {
		const int16_t *pi_ = xi + (size_t)M * m, *pq = xq + (size_t)M * m;
		float Dr = 0, Di = 0;
		int ind;
		for (ind = 0; ind < M; ind++) {
			float r = (float)pi_[ind];                       /* sdrplay.c:219 */
			float g = (float)pq[ind];                        /* sdrplay.c:220 */
			float wr = osc[2 * ind], wi = osc[2 * ind + 1];
			float pr = r * wr - g * wi;                      /* sdrplay.c:223 */
			float pim = r * wi + g * wr;
			Dr = Dr + pr;
			Di = Di + pim;
		}
		dm[m] = cabsf(Dr + Di * I) / 4;                          /* sdrplay.c:225 */
	}
}

/* ------------------------------------------------------------------ */
/* air.c front end (real float32 samples at Fs/4 offset)                */
/* ------------------------------------------------------------------ */

/* air.c:40-63 chooseFc without the R820T filter branch (only taken at exactly 5 Msps) */
unsigned int orc_air_choose_fc(unsigned int minF, unsigned int maxF)
{
	return ((maxF + minF) / 2 + 0 + ORC_INTRATE / 2) / ORC_INTRATE * ORC_INTRATE;
}

/* air.c:278-285.  Fc, ch->Fr are ints, AIRINRATE unsigned: Fc-Fr+AIRINRATE/4 is evaluated in
 * unsigned arithmetic and converted to double; the phase accumulates in double with wraps. */
void orc_air_taps(int Fr, int Fc, unsigned int inrate, float *wf)
{
	unsigned int M = inrate / ORC_INTRATE;           /* air.c:213 AIRMULT */
	unsigned int i;
	double AMFreq, Ph;
	AMFreq = 2.0 * M_PI * (double)(Fc - Fr + inrate / 4) / (double)(inrate);
	for (i = 0, Ph = 0; i < M; i++) {
		float complex w = cexpf(Ph * -I) / M;
		wf[2 * i] = crealf(w);
		wf[2 * i + 1] = cimagf(w);
		Ph += AMFreq;
		if (Ph > 2.0 * M_PI) Ph -= 2.0 * M_PI;
		if (Ph < -2.0 * M_PI) Ph += 2.0 * M_PI;
	}
}

/* air.c:299-338 for one channel over a whole stream (ch->D and `ind` carry partial windows
 * across transfers): D += wf[i]*S, complex tap times real sample, float. */
void orc_fir_f32r(const float *x, size_t nout, int M, const float *wf, float *dm)
{
	size_t m;
	for (m = 0; m < nout; m++) {
		const float *p = x + (size_t)M * m;
		float Dr = 0, Di = 0;
		int i;
		for (i = 0; i < M; i++) {
			float S = p[i];
			Dr = Dr + wf[2 * i] * S;                         /* air.c:315-316 */
			Di = Di + wf[2 * i + 1] * S;
		}
		dm[m] = cabsf(Dr + Di * I);                              /* air.c:318 */
	}
}

/* ------------------------------------------------------------------ */
/* acars.c framing FSM (only what is reachable from putbit)            */
/* ------------------------------------------------------------------ */

#define SYN 0x16
#define SOH 0x01
#define ETX 0x83
#define ETB 0x97
#define DLE 0x7f
#define MAXPERR 3

static void reset_acars(orc_chan *ch)               /* acars.c:239-244 */
{
	ch->Acarsstate = ORC_WSYN;
	ch->MskDf = 0;
	ch->nbits = 1;
}

static void put_frame(orc_chan *ch)                 /* acars.c:350-369 */
{
	float lvl = 10 * log10(ch->MskLvlSum / ch->MskBitCount);   /* double -> float */
	if (ch->frames && ch->frames_n < ch->frames_cap) {
		orc_frame *f = &ch->frames[ch->frames_n];
		f->chn = ch->chn;
		f->len = ch->blk_len;
		f->err = ch->blk_err;
		f->lvl = lvl;
		f->crc[0] = ch->blk_crc[0];
		f->crc[1] = ch->blk_crc[1];
		memset(f->txt, 0, sizeof(f->txt));
		if (ch->blk_len > 0)
			memcpy(f->txt, ch->blk_txt, (size_t)ch->blk_len);
		f->end_bit = ch->nbit_total;
		f->end_sample = ch->cur_sample;
		f->soh_sample = ch->soh_sample;
	}
	ch->frames_n++;
	ch->Acarsstate = ORC_END;
	ch->nbits = 8;
}

static void decode_acars(orc_chan *ch)              /* acars.c:246-375 */
{
	unsigned char r = ch->outbits;

	switch (ch->Acarsstate) {
	case ORC_WSYN:                              /* acars.c:252-265 */
		if (r == SYN) {
			ch->Acarsstate = ORC_SYN2;
			ch->nbits = 8;
			return;
		}
		if (r == (unsigned char)~SYN) {
			ch->MskS ^= 2;
			ch->Acarsstate = ORC_SYN2;
			ch->nbits = 8;
			return;
		}
		ch->nbits = 1;
		return;
	case ORC_SYN2:                              /* acars.c:267-279 */
		if (r == SYN) {
			ch->Acarsstate = ORC_SOH1;
			ch->nbits = 8;
			return;
		}
		if (r == (unsigned char)~SYN) {
			ch->MskS ^= 2;
			ch->nbits = 8;
			return;
		}
		reset_acars(ch);
		return;
	case ORC_SOH1:                              /* acars.c:281-301 */
		if (r == SOH) {
			ch->Acarsstate = ORC_TXT;
			ch->blk_len = 0;
			ch->blk_err = 0;
			ch->nbits = 8;
			ch->MskLvlSum = 0;
			ch->MskBitCount = 0;
			ch->soh_sample = ch->cur_sample;      /* acars.c:290: gettimeofday(&(ch->blk->tv)) sits here */
			return;
		}
		reset_acars(ch);
		return;
	case ORC_TXT:                               /* acars.c:303-341 */
		ch->blk_txt[ch->blk_len] = r;
		ch->blk_len++;
		if ((popcount8(r) & 1) == 0) {
			ch->blk_err++;
			if (ch->blk_err > MAXPERR + 1) {
				reset_acars(ch);
				return;
			}
		}
		if (r == ETX || r == ETB) {
			ch->Acarsstate = ORC_CRC1;
			ch->nbits = 8;
			return;
		}
		if (ch->blk_len > 20 && r == DLE) {
			ch->blk_len -= 3;
			ch->blk_crc[0] = ch->blk_txt[ch->blk_len];
			ch->blk_crc[1] = ch->blk_txt[ch->blk_len + 1];
			ch->Acarsstate = ORC_CRC2;
			put_frame(ch);
			return;
		}
		if (ch->blk_len > 240) {
			reset_acars(ch);
			return;
		}
		ch->nbits = 8;
		return;
	case ORC_CRC1:                              /* acars.c:343-347 */
		ch->blk_crc[0] = r;
		ch->Acarsstate = ORC_CRC2;
		ch->nbits = 8;
		return;
	case ORC_CRC2:                              /* acars.c:348-369 */
		ch->blk_crc[1] = r;
		put_frame(ch);
		return;
	case ORC_END:                               /* acars.c:370-373 */
		reset_acars(ch);
		ch->nbits = 8;
		return;
	}
}

static void putbit(float v, orc_chan *ch)           /* msk.c:53-63 */
{
	ch->outbits >>= 1;
	if (v > 0)
		ch->outbits |= 0x80;
	ch->nbits--;
	if (ch->nbits <= 0)
		decode_acars(ch);
}

/* ------------------------------------------------------------------ */
/* msk.c                                                              */
/* ------------------------------------------------------------------ */

void orc_chan_init(orc_chan *ch, int chn)
{
	orc_bit *bl = ch->bitlog; size_t blc = ch->bitlog_cap;
	orc_frame *fr = ch->frames; size_t frc = ch->frames_cap;
	memset(ch, 0, sizeof(*ch));
	ch->bitlog = bl; ch->bitlog_cap = blc;
	ch->frames = fr; ch->frames_cap = frc;
	ch->chn = chn;
	/* msk.c:34-41: MskPhi=MskClk=0, MskS=0, MskDf=0, idx=0, inb calloc'd */
	/* acars.c:230-234 */
	ch->outbits = 0;
	ch->nbits = 8;
	ch->Acarsstate = ORC_WSYN;
	if (!g_h_ready) {
		orc_msk_h(g_h);
		g_h_ready = 1;
	}
}

static const float PLLG = 38e-4;                    /* msk.c:65 */
static const float PLLC = 0.52;                     /* msk.c:66 */

void orc_demod_msk(orc_chan *ch, const float *dm, int len)
{
	int n;
	int idx = ch->idx;
	double p = ch->MskPhi;

	if (!g_h_ready) {
		orc_msk_h(g_h);
		g_h_ready = 1;
	}

	for (n = 0; n < len; n++) {
		float in;
		double s;
		int j, o;

		/* VCO, msk.c:81-83 (all double) */
		s = 1800.0 / ORC_INTRATE * 2.0 * M_PI + ch->MskDf;
		p += s;
		if (p >= 2.0 * M_PI)
			p -= 2.0 * M_PI;

		/* mixer, msk.c:86-91: float * double complex, narrowed to float complex on store */
		in = dm[n];
		{
			float complex x = in * cexp(-p * I);
			ch->inb[2 * idx] = crealf(x);
			ch->inb[2 * idx + 1] = cimagf(x);
		}
		idx = (idx + 1) % ORC_FLEN;

		/* bit clock, msk.c:95-96: float += double ; double compare */
		ch->MskClk += s;
		if (ch->MskClk >= 3 * M_PI / 2.0 - s / 2) {
			double dphi;
			float vo, lvl;
			float vr = 0, vi = 0;

			ch->MskClk -= 3 * M_PI / 2.0;                 /* msk.c:100 */

			/* matched filter, msk.c:103-107: o from a double expression truncated to int;
			 * h (float) * inb (float complex) accumulated in float complex */
			o = ORC_MFLTOVER * (ch->MskClk / s + 0.5);
			if (o > ORC_MFLTOVER)
				o = ORC_MFLTOVER;
			for (j = 0; j < ORC_FLEN; j++, o += ORC_MFLTOVER) {
				int k = (j + idx) % ORC_FLEN;
				vr = vr + g_h[o] * ch->inb[2 * k];
				vi = vi + g_h[o] * ch->inb[2 * k + 1];
			}

			/* normalise, msk.c:110-113 */
			lvl = cabsf(vr + vi * I);
			{
				/* float complex / double -> double complex (component-wise), narrowed */
				double d = lvl + 1e-8;
				vr = (float)((double)vr / d);
				vi = (float)((double)vi / d);
			}
			ch->MskLvlSum += lvl * lvl / 4;               /* float expr added to double */
			ch->MskBitCount++;

			if (ch->MskS & 1) {                           /* msk.c:115-121 */
				vo = vi;
				if (vo >= 0) dphi = -vr; else dphi = vr;
			} else {
				vo = vr;
				if (vo >= 0) dphi = vi; else dphi = -vi;
			}
			{
				float sv = (ch->MskS & 2) ? -vo : vo;     /* msk.c:122-126 */
				if (ch->bitlog && ch->bitlog_n < ch->bitlog_cap) {
					ch->bitlog[ch->bitlog_n].vo = sv;
					ch->bitlog[ch->bitlog_n].lvl = lvl;
				}
				ch->bitlog_n++;
				ch->cur_sample = ch->nsamp_total + n;
				putbit(sv, ch);
				ch->nbit_total++;
			}
			ch->MskS++;

			/* PLL filter, msk.c:130: float constants promoted to double */
			ch->MskDf = PLLC * ch->MskDf + (1.0 - PLLC) * PLLG * dphi;
		}
	}

	ch->idx = idx;
	ch->MskPhi = p;
	ch->nsamp_total += len;
}

/* rtl.c:314-361: all channels share the stream; then demodMSK per channel */
void orc_in_callback(orc_chan *chs, int nch, const uint8_t *iq, int nout, int M,
		     const float *wf, float *dm_scratch)
{
	int n;
	for (n = 0; n < nch; n++)
		orc_fir_u8(iq, (size_t)nout, M, M, wf + (size_t)n * 2 * M, dm_scratch + (size_t)n * nout);
	for (n = 0; n < nch; n++)
		orc_demod_msk(&chs[n], dm_scratch + (size_t)n * nout, nout);
}

/* acars.c:123-207 without the repair attempts */
int orc_frame_check(const orc_frame *f)
{
	int i, pn = 0;
	unsigned short crc = 0;
	unsigned char t12;

	if (f->len < 13)
		return -1;
	for (i = 0; i < f->len; i++) {
		unsigned char c = f->txt[i];
		if (i == 12) {                              /* acars.c:132-133 force STX/ETX */
			t12 = c;
			t12 &= (ETX | 0x02);
			t12 |= (ETX & 0x02);
			c = t12;
		}
		if ((popcount8(c) & 1) == 0)
			pn++;
		crc = orc_crc_update(crc, c);
	}
	crc = orc_crc_update(crc, f->crc[0]);
	crc = orc_crc_update(crc, f->crc[1]);
	if (pn)
		return pn;
	return crc ? -2 : 0;
}

/* ------------------------------------------------------------------ */
/* acars.c:39-215: the block thread's parity / CRC check and repair     */
/* ------------------------------------------------------------------ */

#define NSYND (8 * 242)
static unsigned short g_synd[NSYND];
static int g_synd_ready;

/* syndrom.h:52-295 is not a free-standing table: entry i + 8k is the CRC remainder produced by
 * flipping bit i of the byte that is followed by k more bytes (text or CRC) -- one pass of that
 * single bit through the CRC register, then k zero bytes.  Generated here instead of copied. */
void orc_syndrome_table(unsigned short *out, int n)
{
	int i, k, j;
	if (!g_crc_ready)
		crc_init();
	for (k = 0; k * 8 < n; k++)
		for (i = 0; i < 8 && k * 8 + i < n; i++) {
			unsigned short s = g_crc_tab[1 << i];
			for (j = 0; j < k; j++)
				s = (unsigned short)((s >> 8) ^ g_crc_tab[s & 0xff]);
			out[k * 8 + i] = s;
		}
}

static void synd_init(void)
{
	orc_syndrome_table(g_synd, NSYND);
	g_synd_ready = 1;
}

static int fixprerr(unsigned char *txt, int len, unsigned short crc, const int *pr, int pn)   /* acars.c:39-64 */
{
	int i;
	if (pn > 0) {
		for (i = 0; i < 8; i++) {
			if (fixprerr(txt, len, crc ^ g_synd[i + 8 * (len - *pr + 1)], pr + 1, pn - 1)) {
				txt[*pr] ^= (1 << i);
				return 1;
			}
		}
		return 0;
	}
	if (crc == 0)
		return 1;
	for (i = 0; i < 2 * 8; i++)
		if (g_synd[i] == crc)
			return 1;
	return 0;
}

static int fixdberr(unsigned char *txt, int len, unsigned short crc)                           /* acars.c:66-90 */
{
	int i, j, k;
	for (i = 0; i < 2 * 8; i++)
		if (g_synd[i] == crc)
			return 1;
	for (k = 0; k < len; k++) {
		int bo = 8 * (len - k + 1);
		for (i = 0; i < 8; i++)
			for (j = 0; j < 8; j++) {
				if (i == j)
					continue;
				if ((crc ^ g_synd[i + bo] ^ g_synd[j + bo]) == 0) {
					txt[k] ^= (1 << i);
					txt[k] ^= (1 << j);
					return 1;
				}
			}
	}
	return 0;
}

/* acars.c:123-207: returns 1 and fills *out with what outputmsg() receives (repaired, parity
 * stripped, err = number of parity errors found), or 0 if the block is dropped. */
int orc_blk_process(const orc_frame *in, orc_frame *out)
{
	int i, pn;
	unsigned short crc;
	int pr[MAXPERR];

	if (!g_synd_ready)
		synd_init();
	*out = *in;
	if (out->len < 13)                                      /* acars.c:124 */
		return 0;
	out->txt[12] &= (ETX | 0x02);                           /* acars.c:132-133 force STX/ETX */
	out->txt[12] |= (ETX & 0x02);
	pn = 0;                                                 /* acars.c:136-144 */
	for (i = 0; i < out->len; i++) {
		if ((popcount8(out->txt[i]) & 1) == 0) {
			if (pn < MAXPERR)
				pr[pn] = i;
			pn++;
		}
	}
	if (pn > MAXPERR)                                       /* acars.c:145 */
		return 0;
	out->err = pn;
	crc = 0;                                                /* acars.c:159-165 */
	for (i = 0; i < out->len; i++)
		crc = orc_crc_update(crc, out->txt[i]);
	crc = orc_crc_update(crc, out->crc[0]);
	crc = orc_crc_update(crc, out->crc[1]);
	if (pn) {                                               /* acars.c:170-192 */
		if (fixprerr(out->txt, out->len, crc, pr, pn) == 0)
			return 0;
	} else if (crc) {
		if (fixdberr(out->txt, out->len, crc) == 0)
			return 0;
	}
	pn = 0;                                                 /* acars.c:195-207 */
	for (i = 0; i < out->len; i++) {
		if ((popcount8(out->txt[i]) & 1) == 0)
			pn++;
		out->txt[i] &= 0x7f;
	}
	if (pn)
		return 0;
	return 1;
}


/* output.c:486-560: mode, address without dots, ACK (NAK printed as '!'), label (DEL printed as 'd'), block id,
 * start / end of text; for downlinks (block id '0'..'9', output.c:31) message number and flight id; then the text
 * (output.c:566-568,623-631).  The CLI's filters (airflt, label_filter) are not part of the split. */
void orc_msg_split(const orc_frame *blk, orc_msg *out)
{
	int i, j, k;
	memset(out, 0, sizeof(*out));
	out->chn = blk->chn;
	out->lvl = blk->lvl;
	out->err = blk->err;
	k = 0;
	out->mode = (char)blk->txt[k];
	k++;
	for (i = 0, j = 0; i < 7; i++, k++) {
		if (blk->txt[k] != '.') {
			out->addr[j] = (char)blk->txt[k];
			j++;
		}
	}
	out->addr[j] = '\0';
	out->ack = (char)blk->txt[k];
	if (out->ack == 0x15)
		out->ack = '!';
	k++;
	out->label[0] = (char)blk->txt[k];
	k++;
	out->label[1] = (char)blk->txt[k];
	if (out->label[1] == 0x7f)
		out->label[1] = 'd';
	k++;
	out->label[2] = '\0';
	out->bid = (char)blk->txt[k];
	k++;
	out->down = (out->bid >= '0' && out->bid <= '9');
	out->bs = (char)blk->txt[k];
	k++;
	out->be = (char)blk->txt[blk->len - 1];
	if (out->bs != 0x03) {
		int txt_len;
		if (out->down) {
			for (i = 0; i < 4 && k < blk->len - 1; i++, k++)
				out->no[i] = (char)blk->txt[k];
			out->no[i] = '\0';
			for (i = 0; i < 6 && k < blk->len - 1; i++, k++)
				out->fid[i] = (char)blk->txt[k];
			out->fid[i] = '\0';
		}
		txt_len = blk->len - k - 1;
		if (txt_len > 0) {
			memcpy(out->txt, blk->txt + k, (size_t)txt_len);
			out->txt_len = txt_len;
		}
	}
}
