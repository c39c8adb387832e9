/*
 * compat_msk.c -- the reference's own call surface on top of libacarsdec_amd.so.
 *
 * Build this file INSIDE the reference tree (it includes the reference's acarsdec.h and must be
 * compiled with the same WITH_* macros as the rest of acarsdec, because channel_t's layout
 * depends on them, acarsdec.h:62-74) and link it INSTEAD of msk.c.  acars.c, output.c,
 * acarsdec.c, soundfile.c ... stay byte-for-byte unchanged.  It exports
 *
 *     int  initMsk(channel_t *ch);            acarsdec.h:190, replaces msk.c:30-51
 *     void demodMSK(channel_t *ch, int len);  acarsdec.h:191, replaces msk.c:67-137
 *     void acarsdec_amd_in_callback(unsigned char *buf, uint32_t nread, void *ctx);
 *                                             replaces the static in_callback, rtl.c:314-361
 *                                             (pass it to rtlsdr_read_async at rtl.c:364)
 *     void acarsdec_amd_soapy_samples(const int16_t *iq, int nsamples);       (-DWITH_SOAPY)
 *                                             replaces the per-channel loop of soapy.c:228-254
 *
 * Division of labour: the GPU runs the down-converter and the MSK loop (with a device mirror of
 * the framing FSM, needed because decodeAcars() writes MskDf/MskS back into the loop,
 * acars.c:242,259,274); every decided bit comes back as {soft symbol, level} and is replayed here
 * through the reference's putbit() arithmetic (msk.c:53-63,112-113) into the UNCHANGED
 * decodeAcars().  channel_t stays the state carrier exactly as in the reference: after every call
 * the MSK fields of ALL channels come back in one transfer; before a call they go up (one
 * transfer) only if the caller's channel_t no longer holds what the device has -- a host that
 * re-initialised or edited a channel is honoured, the steady state pays nothing for it.
 * dm_buffer (rtl.c:353) is copied back only on request (acarsdec_amd_compat_keep_dm): nothing in
 * the reference reads it after demodMSK().
 *
 * There is no CPU fallback: if the GPU library fails the process exits like the reference does
 * on "Unable to init internal decoders" (acarsdec.c:456-459).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include "acarsdec.h"
#include "acarsdec_amd.h"
#include "acarsdec_amd_compat.h"

#define FLEN ((INTRATE / 1200) + 1)

static int g_keep_dm;
static double g_secs, g_first;       /* g_first: the first call (round 5: context creation + first launches; now an ordinary call) */
static double g_prepare;             /* context creation + one throw-away call, paid inside initMsk() of the last channel */
static unsigned long g_calls;

void acarsdec_amd_compat_keep_dm(int on) { g_keep_dm = on; }
void acarsdec_amd_compat_stats(double *seconds, unsigned long *calls)
{
	if (seconds) *seconds = g_secs;
	if (calls) *calls = g_calls;
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ACARSDEC_AMD_STATS=1 in the environment: the time spent inside the legacy entry points is printed at exit (bench.py's
 * rtl8 case reads it: ms per callback against the 81.92 ms a callback's signal lasts, rtl.c:49,213) */
void acarsdec_amd_compat_print_stats(void)
{
	if (g_calls)
		fprintf(stderr, "acarsdec_amd compat: %lu calls, %.6f s inside the legacy entry points, %.4f ms per call "
			"(first call %.3f ms; the others %.4f ms per call; context made at initMsk() time in %.1f ms)\n",
			g_calls, g_secs, 1e3 * g_secs / (double)g_calls, 1e3 * g_first,
			g_calls > 1 ? 1e3 * (g_secs - g_first) / (double)(g_calls - 1) : 0.0, 1e3 * g_prepare);
}
static void account(double t0)
{
	static int hooked;
	if (!hooked) {
		const char *e = getenv("ACARSDEC_AMD_STATS");
		hooked = 1;
		if (e && *e && *e != '0')
			atexit(acarsdec_amd_compat_print_stats);
	}
	{
		const double dt = now_s() - t0;
		if (g_calls == 0)
			g_first = dt;
		g_secs += dt;
		g_calls++;
	}
}

static acg_ctx *g_msk;          /* 1-channel context behind demodMSK() */
static int g_msk_blocks;
static acg_ctx *g_rtl;          /* nbch-channel context behind acarsdec_amd_in_callback() */
static channel_t *g_msk_last;   /* the channel whose state the demodMSK() context holds */

static void die(const char *what, acg_ctx *c, int rc)
{
	fprintf(stderr, "acarsdec_amd: %s: %s (%s)\n", what, acg_strerror(rc), c ? acg_last_error(c) : "");
	exit(1);
}

static int compat_prepare(void);

int initMsk(channel_t *ch)
{
	/* msk.c:34-42: same observable effect on channel_t */
	ch->MskPhi = ch->MskClk = 0;
	ch->MskS = 0;
	ch->MskDf = 0;
	ch->idx = 0;
	ch->inb = calloc(FLEN, sizeof(float complex));
	if (ch->inb == NULL)
		return -1;
	/* The last channel (acarsdec.c:445-454 runs this loop after the front end's init has filled wf / oscillator): make the GPU
	 * context NOW, not inside the first callback -- a live radio's first transfer period (81.92 ms on rtl.c, shorter on
	 * airspy / SDRplay) does not hold a context creation (0.14-0.27 s).  A failure is the reference's own error
	 * path: main() prints "Unable to init internal decoders" and exits with this value (acarsdec.c:456-459). */
	if (nbch > 0 && ch == &channel[nbch - 1])
		return compat_prepare();
	return 0;
}

/* what the device holds for the channels of one context, as of the last download: the upload is skipped while the caller's
 * channel_t still says the same */
typedef struct {
	acg_chan_state st[MAXNBCHANNELS];
	int valid;
} shadow_t;

static void pack(acg_chan_state *st, const channel_t *ch)
{
	int i;
	memset(st, 0, sizeof(*st));
	st->MskPhi = ch->MskPhi; st->MskDf = ch->MskDf; st->MskLvlSum = ch->MskLvlSum;
	st->MskClk = ch->MskClk; st->MskBitCount = ch->MskBitCount;
	st->MskS = ch->MskS; st->idx = ch->idx;
	for (i = 0; i < FLEN; i++) {
		st->inb[2 * i] = crealf(ch->inb[i]);
		st->inb[2 * i + 1] = cimagf(ch->inb[i]);
	}
	st->outbits = ch->outbits; st->nbits = ch->nbits; st->Acarsstate = ch->Acarsstate;
	st->blk_len = ch->blk ? ch->blk->len : 0;
	st->blk_err = ch->blk ? ch->blk->err : 0;
}

static shadow_t g_msk_sh, g_rtl_sh, g_front_sh;

/* channels chs[0..n) -> device slots 0..n-1, one transfer, and only if anything differs from what the device has */
static void upload_all(acg_ctx *c, shadow_t *sh, channel_t *const *chs, int n)
{
	acg_chan_state st[MAXNBCHANNELS];
	int i, rc;
	for (i = 0; i < n; i++)
		pack(&st[i], chs[i]);
	if (sh->valid && memcmp(st, sh->st, sizeof(st[0]) * (size_t)n) == 0)
		return;
	if ((rc = acg_set_state_n(c, 0, n, st)) != ACG_OK)
		die("set_state", c, rc);
}

/* the loop state lives on the device; the framing state was advanced by the replay through the real decodeAcars() and is
 * already in *ch -- and it is what the device's mirror of the state machine has arrived at as well, which is why the shadow
 * (the device's view) and the caller's channel_t agree before the next call */
static void download_all(acg_ctx *c, shadow_t *sh, channel_t *const *chs, int n)
{
	int i, k, rc;
	if ((rc = acg_get_state_n(c, 0, n, sh->st)) != ACG_OK)
		die("get_state", c, rc);
	sh->valid = 1;
	for (k = 0; k < n; k++) {
		channel_t *ch = chs[k];
		acg_chan_state *st = &sh->st[k];
		st->soh_back = 0;                 /* (not a channel_t field: pack() cannot know it, and this view stamps blk->tv itself) */
		ch->MskPhi = st->MskPhi; ch->MskDf = st->MskDf; ch->MskClk = st->MskClk;
		ch->MskS = st->MskS; ch->idx = st->idx;
		for (i = 0; i < FLEN; i++)
			ch->inb[i] = st->inb[2 * i] + st->inb[2 * i + 1] * I;
	}
}

/* msk.c:112-113 + putbit() msk.c:53-63, on the caller's channel_t */
static void bit_sink(void *user, int slot, float vo, float lvl)
{
	channel_t *ch = ((channel_t **)user)[slot];
	ch->MskLvlSum += lvl * lvl / 4;
	ch->MskBitCount++;
	ch->outbits >>= 1;
	if (vo > 0)
		ch->outbits |= 0x80;
	ch->nbits--;
	if (ch->nbits <= 0)
		decodeAcars(ch);
}

/* The device assembles blocks as well; this view replays the bits through the caller's own decodeAcars()
 * instead, so the device's copies are dropped to keep its queue empty (ACG_EAGAIN: more are waiting than the
 * scratch holds, come again; ACG_EOVERFLOW cannot happen to a host that drains after every call). */
static void discard_device_blocks(acg_ctx *g)
{
	acg_frame f[8];
	int n = 0, rc;
	do {
		rc = acg_drain_frames(g, f, 8, &n);
	} while (rc == ACG_EAGAIN);
	if (rc != ACG_OK)
		die("drain_frames", g, rc);
}

/* nbch channels of one stream, their tap tables from the front end's own init */
static acg_ctx *make_front_ctx(int mult, int max_blocks, float complex *const *tables)
{
	acg_config cfg;
	acg_ctx *c = NULL;
	float *taps = malloc(sizeof(float) * 2 * (size_t)mult * nbch);
	unsigned int n;
	int k, rc;
	if (taps == NULL)
		return NULL;
	memset(&cfg, 0, sizeof(cfg));
	cfg.nch = (int)nbch; cfg.nstreams = 1; cfg.decim = mult; cfg.ntaps = mult;
	cfg.max_blocks = max_blocks; cfg.flags = ACG_F_BITLOG; cfg.max_lag = 1;        /* drained after every call */
	if ((rc = acg_create(&c, &cfg)) != ACG_OK) {
		fprintf(stderr, "acarsdec_amd: acg_create: %s\n", acg_strerror(rc));
		free(taps);
		return NULL;
	}
	for (n = 0; n < nbch; n++)
		for (k = 0; k < mult; k++) {
			taps[2 * ((size_t)n * mult + k)] = crealf(tables[n][k]);
			taps[2 * ((size_t)n * mult + k) + 1] = cimagf(tables[n][k]);
		}
	rc = acg_set_taps(c, 0, (int)nbch, taps);
	free(taps);
	if (rc != ACG_OK) {
		fprintf(stderr, "acarsdec_amd: set_taps: %s (%s)\n", acg_strerror(rc), acg_last_error(c));
		acg_destroy(c);
		return NULL;
	}
	return c;
}

static acg_ctx *make_msk_ctx(int blocks)
{
	acg_config cfg;
	acg_ctx *c = NULL;
	int rc;
	memset(&cfg, 0, sizeof(cfg));
	cfg.nch = 1; cfg.nstreams = 1; cfg.decim = 8; cfg.ntaps = 8;
	cfg.max_blocks = blocks;
	cfg.flags = ACG_F_BITLOG;
	cfg.max_lag = 1;                              /* drained after every call */
	if ((rc = acg_create(&c, &cfg)) != ACG_OK) {
		fprintf(stderr, "acarsdec_amd: acg_create: %s\n", acg_strerror(rc));
		return NULL;
	}
	return c;
}

static void null_sink(void *user, int slot, float vo, float lvl) { (void)user; (void)slot; (void)vo; (void)lvl; }

/* One throw-away call through every entry point the real calls use (kernels get loaded, staging buffers pinned), then the
 * context goes back to the state acg_create left it in.  fmt < 0: the 12.5 kHz path of demodMSK(). */
static int warm_up(acg_ctx *c, int fmt, int mult, int nchan)
{
	const size_t ns = (size_t)ACG_BLOCK * (size_t)(mult > 0 ? mult : 1);
	acg_chan_state *st = malloc(sizeof(acg_chan_state) * (size_t)nchan);
	void *buf = calloc(ns, fmt == 0 ? 2 : 4);
	void *buf2 = fmt == ACG_FMT_S16_SPLIT ? calloc(ns, 2) : NULL;
	acg_frame f[8];
	int rc = ACG_OK, n = 0;
	if (st == NULL || buf == NULL || (fmt == ACG_FMT_S16_SPLIT && buf2 == NULL))
		rc = ACG_ENOMEM;
	else if (fmt < 0)
		rc = acg_process_dm_host(c, (const float *)buf, ns, ACG_BLOCK);
	else if (fmt == 0) {
		memset(buf, 127, ns * 2);
		rc = acg_process_iq_u8_host(c, buf, ns * 2, 1);
	} else
		rc = acg_feed_samples_host(c, fmt, buf, buf2, 0, ns);
	if (rc == ACG_OK) rc = acg_replay_bits(c, null_sink, NULL);
	if (rc == ACG_OK) rc = acg_get_state_n(c, 0, nchan, st);
	if (rc == ACG_OK) {
		do { rc = acg_drain_frames(c, f, 8, &n); } while (rc == ACG_EAGAIN);
	}
	if (rc == ACG_OK) rc = acg_reset(c);
	if (rc != ACG_OK)
		fprintf(stderr, "acarsdec_amd: warm-up call: %s (%s)\n", acg_strerror(rc), acg_last_error(c));
	free(st); free(buf); free(buf2);
	return rc;
}

void demodMSK(channel_t *ch, int len)
{
	int rc;
	channel_t *one[1];
	const double t0 = now_s();

	if (len <= 0)
		return;
	if (g_msk == NULL || (len + ACG_BLOCK - 1) / ACG_BLOCK > g_msk_blocks) {       /* (made by initMsk(); here: a longer call than foreseen) */
		if (g_msk)
			acg_destroy(g_msk);
		g_msk_blocks = (len + ACG_BLOCK - 1) / ACG_BLOCK;
		if (g_msk_blocks < 4)
			g_msk_blocks = 4;                     /* soundfile.c:27 MAXNBFRAMES 4096 */
		if ((g_msk = make_msk_ctx(g_msk_blocks)) == NULL)
			exit(1);
		g_msk_sh.valid = 0;
	}
	one[0] = ch;
	if (ch != g_msk_last)                                 /* the reference calls demodMSK() for one channel after the other */
		g_msk_sh.valid = 0;
	g_msk_last = ch;
	upload_all(g_msk, &g_msk_sh, one, 1);
	if ((rc = acg_process_dm_host(g_msk, ch->dm_buffer, (size_t)len, len)) != ACG_OK)
		die("process_dm", g_msk, rc);
	if ((rc = acg_replay_bits(g_msk, bit_sink, one)) != ACG_OK)
		die("replay", g_msk, rc);
	download_all(g_msk, &g_msk_sh, one, 1);
	discard_device_blocks(g_msk);
	account(t0);
}

#ifdef WITH_RTL
void acarsdec_amd_in_callback(unsigned char *rtlinbuff, uint32_t nread, void *ctx)
{
	unsigned int n;
	int rc;
	channel_t *chs[MAXNBCHANNELS];
	const double t0 = now_s();
	(void)ctx;

	if (nread != (uint32_t)(ACG_BLOCK * rtlMult * 2)) {          /* rtl.c:322-326 */
		fprintf(stderr, "warning: partial read\n");
		return;
	}
	if (g_rtl == NULL) {                                         /* (normally made by initMsk(): a host that skipped it lands here) */
		float complex *tab[MAXNBCHANNELS];
		for (n = 0; n < nbch; n++) tab[n] = channel[n].wf;       /* wf from initRtl, rtl.c:283-286 */
		if ((g_rtl = make_front_ctx(rtlMult, 1, tab)) == NULL)
			exit(1);
	}
	for (n = 0; n < nbch; n++)
		chs[n] = &channel[n];
	upload_all(g_rtl, &g_rtl_sh, chs, (int)nbch);
	if ((rc = acg_process_iq_u8_host(g_rtl, rtlinbuff, (size_t)nread, 1)) != ACG_OK)
		die("process_iq", g_rtl, rc);
	if (g_keep_dm)                                               /* rtl.c:353: dm_buffer observable on request */
		for (n = 0; n < nbch; n++)
			if ((rc = acg_read_dm(g_rtl, (int)n, channel[n].dm_buffer, ACG_BLOCK)) != ACG_OK)
				die("read_dm", g_rtl, rc);
	if ((rc = acg_replay_bits(g_rtl, bit_sink, chs)) != ACG_OK)       /* rtl.c:357-360 order */
		die("replay", g_rtl, rc);
	download_all(g_rtl, &g_rtl_sh, chs, (int)nbch);
	discard_device_blocks(g_rtl);
	account(t0);
}
#endif

#ifdef WITH_SOAPY
/*
 * The per-channel loop of soapy.c:228-254 for all channels: `nsamples` CS16 samples of a read of any size.  The reference
 * carries the partial window across reads in ch->D / current_index; here the incomplete window's samples wait on the device
 * (acg_feed_samples_host), the sums are the same sequential windows.  Every complete window is demodulated at once (the
 * reference waits until SOAPYOUTBUFSZ = 1024 of them have gathered, soapy.c:245-248: the same per-channel bit sequence,
 * decodeAcars() called a little earlier), the bits are replayed into the caller's channel_t.  The binding a maintainer adds is
 * one hunk in soapy.c's reader:
 *     -		for (n = 0; n < nbch; n++) { ... }            soapy.c:231-253
 *     -		current_index = (current_index + res) % rateMult;
 *     +		acarsdec_amd_soapy_samples(soapyInBuf, res);
 */
static acg_ctx *g_soapy;
static size_t g_soapy_carry;          /* samples of the incomplete window the device holds */

void acarsdec_amd_soapy_samples(const int16_t *iq, int nsamples)
{
	unsigned int n;
	int rc;
	channel_t *chs[MAXNBCHANNELS];
	const double t0 = now_s();

	if (nsamples <= 0)
		return;
	if (g_soapy == NULL) {                                       /* (normally made by initMsk()) */
		float complex *tab[MAXNBCHANNELS];
		for (n = 0; n < nbch; n++) tab[n] = channel[n].oscillator;   /* oscillator[] from initSoapy, soapy.c:131-137 */
		/* a read is at most SOAPYOUTBUFSZ windows (soapy.c:59) + the carry: two blocks */
		if ((g_soapy = make_front_ctx(rateMult, 2, tab)) == NULL)
			exit(1);
	}
	if ((g_soapy_carry + (size_t)nsamples) / (size_t)rateMult == 0) {        /* no window completes: nothing to demodulate yet */
		if ((rc = acg_feed_samples_host(g_soapy, ACG_FMT_CS16, iq, NULL, 0, (size_t)nsamples)) != ACG_OK)
			die("feed_samples", g_soapy, rc);
		g_soapy_carry += (size_t)nsamples;
		account(t0);
		return;
	}
	for (n = 0; n < nbch; n++)
		chs[n] = &channel[n];
	upload_all(g_soapy, &g_front_sh, chs, (int)nbch);
	if ((rc = acg_feed_samples_host(g_soapy, ACG_FMT_CS16, iq, NULL, 0, (size_t)nsamples)) != ACG_OK)
		die("feed_samples", g_soapy, rc);
	g_soapy_carry = (g_soapy_carry + (size_t)nsamples) % (size_t)rateMult;
	if ((rc = acg_replay_bits(g_soapy, bit_sink, chs)) != ACG_OK)           /* channel order, as soapy.c:231 iterates */
		die("replay", g_soapy, rc);
	download_all(g_soapy, &g_front_sh, chs, (int)nbch);
	discard_device_blocks(g_soapy);
	account(t0);
}
#endif

#if defined(WITH_AIR) || defined(WITH_SDRPLAY)
/* the front ends whose DSP sits in a vendor callback: samples of any count per call, the partial window carried on the device */
static acg_ctx *g_fe;
static int g_fe_mult;
static size_t g_fe_carry;

static void fe_samples(int fmt, const void *p0, const void *p1, int nsamples, int mult, float complex *const *tables)
{
	unsigned int n;
	int rc;
	channel_t *chs[MAXNBCHANNELS];
	const double t0 = now_s();

	if (nsamples <= 0)
		return;
	if (g_fe != NULL && g_fe_mult != mult) {                     /* initMsk() guessed another window length: start over */
		acg_destroy(g_fe);
		g_fe = NULL;
		g_front_sh.valid = 0;
	}
	if (g_fe == NULL) {                                          /* (normally made by initMsk()) */
		if ((g_fe = make_front_ctx(mult, 2, tables)) == NULL)
			exit(1);
		g_fe_mult = mult;
	}
	if ((g_fe_carry + (size_t)nsamples) / (size_t)mult == 0) {               /* no window completes: nothing to demodulate yet */
		if ((rc = acg_feed_samples_host(g_fe, fmt, p0, p1, 0, (size_t)nsamples)) != ACG_OK)
			die("feed_samples", g_fe, rc);
		g_fe_carry += (size_t)nsamples;
		account(t0);
		return;
	}
	for (n = 0; n < nbch; n++)
		chs[n] = &channel[n];
	upload_all(g_fe, &g_front_sh, chs, (int)nbch);
	if ((rc = acg_feed_samples_host(g_fe, fmt, p0, p1, 0, (size_t)nsamples)) != ACG_OK)
		die("feed_samples", g_fe, rc);
	g_fe_carry = (g_fe_carry + (size_t)nsamples) % (size_t)mult;
	if ((rc = acg_replay_bits(g_fe, bit_sink, chs)) != ACG_OK)
		die("replay", g_fe, rc);
	download_all(g_fe, &g_front_sh, chs, (int)nbch);
	discard_device_blocks(g_fe);
	account(t0);
}
#endif

#ifdef WITH_AIR
/*
 * The body of rx_callback() (air.c:291-341) for all channels: `count` real float32 samples of a transfer of any size; airmult
 * is air.c's AIRMULT (static there; the call sits inside air.c, where it is in scope):
 *     static int rx_callback(airspy_transfer_t *transfer)
 *     {
 *     +	acarsdec_amd_air_samples((float *)transfer->samples, transfer->sample_count, AIRMULT);
 *     +	return 0;
 *     -	... air.c:293-340
 */
void acarsdec_amd_air_samples(const float *samples, int count, int airmult)
{
	float complex *tab[MAXNBCHANNELS];
	unsigned int n;
	for (n = 0; n < nbch; n++) tab[n] = channel[n].wf;                       /* air.c:278-285 */
	fe_samples(ACG_FMT_F32_REAL, samples, NULL, count, airmult, tab);
}
#endif

#ifdef WITH_SDRPLAY
/* The body of myStreamCallback() (sdrplay.c:215-236) for all channels: numSamples int16 I and Q samples; SDRPLAY_MULT = 160. */
void acarsdec_amd_sdrplay_samples(const int16_t *xi, const int16_t *xq, int nsamples)
{
	float complex *tab[MAXNBCHANNELS];
	unsigned int n;
	for (n = 0; n < nbch; n++) tab[n] = channel[n].oscillator;               /* sdrplay.c:160-164 */
	fe_samples(ACG_FMT_S16_SPLIT, xi, xq, nsamples, 160, tab);
}
#endif


/* The context of whichever front end this build serves, made when initMsk() sees the last channel (above).  The front end's
 * own init has run (acarsdec.c:420-437 before the loop at :445): its tap tables exist, or -- sound-file / ALSA input -- they
 * do not, and demodMSK()'s 1-channel context is what will be used. */
static int compat_prepare(void)
{
	const double t0 = now_s();
	float complex *tab[MAXNBCHANNELS];
	unsigned int n;
	int rc = ACG_OK, made = 0;
	(void)tab; (void)n;
#if defined(WITH_RTL)
	if (g_rtl == NULL && rtlMult > 0 && channel[0].wf != NULL) {
		for (n = 0; n < nbch; n++) tab[n] = channel[n].wf;
		if ((g_rtl = make_front_ctx(rtlMult, 1, tab)) == NULL)
			return -1;
		rc = warm_up(g_rtl, 0, rtlMult, (int)nbch);
		made = 1;
	}
#elif defined(WITH_SOAPY)
	if (g_soapy == NULL && rateMult > 0 && channel[0].oscillator != NULL) {
		for (n = 0; n < nbch; n++) tab[n] = channel[n].oscillator;
		if ((g_soapy = make_front_ctx(rateMult, 2, tab)) == NULL)
			return -1;
		rc = warm_up(g_soapy, ACG_FMT_CS16, rateMult, (int)nbch);
		made = 1;
	}
#elif defined(WITH_SDRPLAY)
	if (g_fe == NULL && channel[0].oscillator != NULL) {
		for (n = 0; n < nbch; n++) tab[n] = channel[n].oscillator;
		if ((g_fe = make_front_ctx(160, 2, tab)) == NULL)              /* SDRPLAY_MULT, sdrplay.c:40 */
			return -1;
		g_fe_mult = 160;
		rc = warm_up(g_fe, ACG_FMT_S16_SPLIT, 160, (int)nbch);
		made = 1;
	}
#elif defined(WITH_AIR)
	if (g_fe == NULL && channel[0].wf != NULL && crealf(channel[0].wf[0]) > 0) {
		/* AIRMULT is static in air.c; wf[0] = cexpf(0) / AIRMULT (air.c:279-280) says what it is */
		const int mult = (int)(1.0f / crealf(channel[0].wf[0]) + 0.5f);
		if (mult >= 8 && mult <= 1024 && mult % 4 == 0) {
			for (n = 0; n < nbch; n++) tab[n] = channel[n].wf;
			if ((g_fe = make_front_ctx(mult, 2, tab)) == NULL)
				return -1;
			g_fe_mult = mult;
			rc = warm_up(g_fe, ACG_FMT_F32_REAL, mult, (int)nbch);
			made = 1;
		}
	}
#endif
	if (!made && g_msk == NULL) {                                    /* soundfile.c / alsa.c feed demodMSK() directly */
		g_msk_blocks = 4;                                            /* soundfile.c:27 MAXNBFRAMES 4096 */
		if ((g_msk = make_msk_ctx(g_msk_blocks)) == NULL)
			return -1;
		rc = warm_up(g_msk, -1, 0, 1);
	}
	g_prepare = now_s() - t0;
	return rc == ACG_OK ? 0 : -1;
}
