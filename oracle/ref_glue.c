/*
 * ref_glue.c -- harness around the UNMODIFIED reference sources (msk.c, acars.c, rtl.c),
 * which oracle/Makefile compiles in place from /root/reference into oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/acars_oracle.c and, optionally, as the
 * "reference" CPU baseline of bench.py.  No reference source is copied into this repo;
 * this file only supplies what the reference expects from its environment:
 *   - the globals of acarsdec.c (acarsdec.c:34-58) that the three files read,
 *   - no-op librtlsdr entry points (see stub/rtl-sdr.h),
 *   - outputmsg() (output.c:486), replaced by a capture buffer,
 *   - pthread_create / pthread_cond_wait hooks (injected with -D on acars.c) that turn
 *     the asynchronous blk_thread (acars.c:93) into an on-demand synchronous drain,
 *   - linker --wrap hooks on decodeAcars()/demodMSK() that snapshot blocks as they are
 *     queued (acars.c:356-364), and a cabsf hook (-D on msk.c, -O2 build only) that logs
 *     the matched-filter output of every bit (msk.c:110).
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <setjmp.h>
#include <math.h>
#include <complex.h>
#ifdef WITH_RTL
#include <rtl-sdr.h>
#endif
#ifdef WITH_SOAPY
#include <SoapySDR/Device.h>
#endif
#ifdef WITH_SDRPLAY
#include <mirsdrapi-rsp.h>
#endif
#ifdef WITH_AIR
#include <libairspy/airspy.h>
#endif
#include "acarsdec.h"

/* ---- globals the reference files expect (acarsdec.c:34-75) ---- */
channel_t channel[MAXNBCHANNELS];
unsigned int nbch;
int verbose = 0;
int signalExit = 0;
#ifdef WITH_RTL
int gain = -100;
int ppm = 0;
int rtlMult = 160;
#endif
#ifdef WITH_SOAPY
char *antenna = NULL;
double gain = -10.0;
int ppm = 0;
int rateMult = 160;
int freq = 0;
#endif
#ifdef WITH_SDRPLAY
int lnaState = 2;
int GRdB = 20;
int ppm = 0;
int gain = 0;
#endif
#ifdef WITH_AIR
int gain = 18;
#endif

#ifdef WITH_RTL
/* ---- librtlsdr stand-ins ---- */
static uint32_t g_center_freq;
uint32_t rtlsdr_get_device_count(void) { return 1; }
const char *rtlsdr_get_device_name(uint32_t index) { (void)index; return "oracle-stub"; }
int rtlsdr_get_device_usb_strings(uint32_t index, char *m, char *p, char *s)
{
	(void)index;
	if (m) strcpy(m, "stub");
	if (p) strcpy(p, "stub");
	if (s) strcpy(s, "00000000");
	return 0;
}
int rtlsdr_open(rtlsdr_dev_t **dev, uint32_t index) { (void)index; *dev = (rtlsdr_dev_t *)&g_center_freq; return 0; }
int rtlsdr_close(rtlsdr_dev_t *dev) { (void)dev; return 0; }
int rtlsdr_set_center_freq(rtlsdr_dev_t *dev, uint32_t freq) { (void)dev; g_center_freq = freq; return 0; }
int rtlsdr_set_freq_correction(rtlsdr_dev_t *dev, int p) { (void)dev; (void)p; return 0; }
int rtlsdr_get_tuner_gains(rtlsdr_dev_t *dev, int *gains) { (void)dev; if (gains) gains[0] = 0; return 1; }
int rtlsdr_set_tuner_gain(rtlsdr_dev_t *dev, int g) { (void)dev; (void)g; return 0; }
int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *dev, int m) { (void)dev; (void)m; return 0; }
int rtlsdr_set_sample_rate(rtlsdr_dev_t *dev, uint32_t r) { (void)dev; (void)r; return 0; }
int rtlsdr_reset_buffer(rtlsdr_dev_t *dev) { (void)dev; return 0; }
int rtlsdr_read_async(rtlsdr_dev_t *dev, rtlsdr_read_async_cb_t cb, void *ctx, uint32_t n, uint32_t l)
{ (void)dev; (void)cb; (void)ctx; (void)n; (void)l; return 0; }
int rtlsdr_cancel_async(rtlsdr_dev_t *dev) { (void)dev; return 0; }
#endif /* WITH_RTL */

/* ---- capture buffers ---- */
typedef struct {
	int chn, len, err;
	float lvl;
	unsigned char crc[2];
	unsigned char txt[250];
} ref_frame;

#define REF_MAXFRAMES 65536
static ref_frame *g_raw;       /* blocks as queued by decodeAcars (before blk_thread) */
static int g_nraw;
static ref_frame *g_out;       /* blocks as they reach outputmsg (after parity/CRC repair + strip) */
static int g_nout;

static void grab(ref_frame *dst, const msgblk_t *blk)
{
	dst->chn = blk->chn;
	dst->len = blk->len;
	dst->err = blk->err;
	dst->lvl = blk->lvl;
	dst->crc[0] = blk->crc[0];
	dst->crc[1] = blk->crc[1];
	memset(dst->txt, 0, sizeof(dst->txt));
	if (blk->len > 0)
		memcpy(dst->txt, blk->txt, (size_t)(blk->len < 250 ? blk->len : 250));
}

void outputmsg(const msgblk_t *blk)           /* stands in for output.c:486 */
{
	if (!g_out)
		g_out = calloc(REF_MAXFRAMES, sizeof(ref_frame));
	if (g_nout < REF_MAXFRAMES)
		grab(&g_out[g_nout], blk);
	g_nout++;
}

/* ---- blk_thread made synchronous ---- */
static void *(*g_blk_fn)(void *);
static jmp_buf g_jb;
static int g_draining;

int ref_hook_pthread_create(pthread_t *t, const pthread_attr_t *a, void *(*fn)(void *), void *arg)
{
	(void)a; (void)arg;
	if (t) memset(t, 0, sizeof(*t));
	g_blk_fn = fn;                     /* acars.c:225: remember blk_thread, do not start it */
	return 0;
}

int ref_hook_cond_wait(pthread_cond_t *c, pthread_mutex_t *m)
{
	(void)c;
	/* acars.c:106-107: queue empty -> blk_thread would sleep: leave it instead */
	if (g_draining) {
		pthread_mutex_unlock(m);
		longjmp(g_jb, 1);
	}
	return 0;
}

void ref_drain(void)
{
	if (!g_blk_fn)
		return;
	g_draining = 1;
	if (setjmp(g_jb) == 0)
		g_blk_fn(NULL);
	g_draining = 0;
}

/* ---- --wrap hooks ---- */
void __real_decodeAcars(channel_t *ch);
void __wrap_decodeAcars(channel_t *ch)
{
	msgblk_t *before = ch->blk;
	int st = ch->Acarsstate;
	__real_decodeAcars(ch);
	if (before && ch->blk == NULL && ch->Acarsstate == END && (st == CRC2 || st == TXT)) {
		/* acars.c:356-366: `before` has just been queued; blk_thread has not touched it */
		if (!g_raw)
			g_raw = calloc(REF_MAXFRAMES, sizeof(ref_frame));
		if (g_nraw < REF_MAXFRAMES)
			grab(&g_raw[g_nraw], before);
		g_nraw++;
	}
}

static channel_t *g_cur;
/* optional log of everything handed to demodMSK (the front end's dm output), per channel */
static float *g_dmlog[MAXNBCHANNELS];
static size_t g_dmlog_n[MAXNBCHANNELS], g_dmlog_cap;
void ref_dmlog_enable(size_t cap_per_channel)
{
	int i;
	for (i = 0; i < MAXNBCHANNELS; i++) {
		free(g_dmlog[i]);
		g_dmlog[i] = cap_per_channel ? calloc(cap_per_channel, sizeof(float)) : NULL;
		g_dmlog_n[i] = 0;
	}
	g_dmlog_cap = cap_per_channel;
}
size_t ref_dmlog_count(int n) { return g_dmlog_n[n]; }
const float *ref_dmlog(int n) { return g_dmlog[n]; }
void __real_demodMSK(channel_t *ch, int len);
void __wrap_demodMSK(channel_t *ch, int len)
{
	if (g_dmlog_cap && ch->chn >= 0 && ch->chn < MAXNBCHANNELS && g_dmlog[ch->chn]) {
		size_t room = g_dmlog_cap - g_dmlog_n[ch->chn];
		size_t k = (size_t)len < room ? (size_t)len : room;
		memcpy(g_dmlog[ch->chn] + g_dmlog_n[ch->chn], ch->dm_buffer, k * sizeof(float));
		g_dmlog_n[ch->chn] += k;
	}
	g_cur = ch;
	__real_demodMSK(ch, len);
	g_cur = NULL;
}

/* per-bit matched-filter log (msk.c:110 `lvl=cabsf(v)`), filled by the cabsf hook */
typedef struct { float vr, vi; unsigned int MskS; int chn; } ref_bit;
static ref_bit *g_bits;
static size_t g_nbits, g_bits_cap;

float ref_hook_cabsf(float complex v)
{
	if (g_cur && g_bits) {
		if (g_nbits < g_bits_cap) {
			g_bits[g_nbits].vr = crealf(v);
			g_bits[g_nbits].vi = cimagf(v);
			g_bits[g_nbits].MskS = g_cur->MskS;
			g_bits[g_nbits].chn = g_cur->chn;
		}
		g_nbits++;
	}
	return cabsf(v);
}

void ref_bitlog_enable(size_t cap)
{
	free(g_bits);
	g_bits = cap ? calloc(cap, sizeof(ref_bit)) : NULL;
	g_bits_cap = cap;
	g_nbits = 0;
}
size_t ref_bitlog_count(void) { return g_nbits; }
const ref_bit *ref_bitlog(void) { return g_bits; }

/* ---- driver API (ctypes) ---- */
#ifdef WITH_RTL
void ref_rtl_in_callback(unsigned char *buf, uint32_t nread);   /* ref_rtl_unit.c */
#endif

static void common_init(void)
{
	unsigned int n;
	g_nraw = g_nout = 0;
	for (n = 0; n < nbch; n++) {           /* acarsdec.c:445-454 */
		channel[n].chn = n;
		channel[n].MskLvlSum = 0;          /* static storage in the reference: zero at start */
		channel[n].MskBitCount = 0;
		channel[n].blk = NULL;
		initMsk(&channel[n]);
		initAcars(&channel[n]);
	}
}

#ifdef WITH_RTL
/* RTL path: freqs are decimal MHz strings exactly as on the reference command line
 * (acarsdec -m mult -r 0 f1 f2 ...).  Returns the centre frequency chosen (rtl.c:268), <0 on error. */
long ref_init_rtl(int nfreq, const char **freqs, int mult)
{
	char *argv[MAXNBCHANNELS + 3];
	int i, r;
	if (nfreq > MAXNBCHANNELS)
		return -1;
	rtlMult = mult;
	argv[0] = (char *)"0";
	for (i = 0; i < nfreq; i++)
		argv[1 + i] = (char *)freqs[i];
	argv[1 + nfreq] = NULL;
	g_center_freq = 0;
	r = initRtl(argv, 0);
	if (r)
		return -2;
	common_init();
	return (long)g_center_freq;
}
void ref_in_callback(unsigned char *buf, unsigned int nread) { ref_rtl_in_callback(buf, nread); }
int ref_get_wf(int n, float *out, int M)
{
	int i;
	for (i = 0; i < M; i++) {
		out[2 * i] = crealf(channel[n].wf[i]);
		out[2 * i + 1] = cimagf(channel[n].wf[i]);
	}
	return channel[n].Fr;
}
/* Overwrites the tap table initRtl built for channel n (rtl.c:283-286) with the caller's: lets a test run the
 * reference's in_callback on the very taps another implementation was given (bench.py's reference -O2 vs -Ofast leg).
 * Taps beyond ntaps are zero.  Harness only: nothing of the reference is modified. */
int ref_set_wf(int n, const float *taps, int ntaps)
{
	int i;
	if (n < 0 || (unsigned int)n >= nbch || ntaps < 0 || ntaps > rtlMult)
		return -1;
	for (i = 0; i < rtlMult; i++)
		channel[n].wf[i] = i < ntaps ? taps[2 * i] + taps[2 * i + 1] * I : 0;
	return 0;
}
#endif /* WITH_RTL */

#ifdef WITH_SOAPY
/* ---- SoapySDR stand-ins: set-up calls do nothing, readStream serves the test's samples ---- */
static const int16_t *g_feed;          /* interleaved I,Q */
static size_t g_feed_left;             /* complex samples left */
static size_t g_feed_chunk;            /* complex samples per readStream (0 = whatever is asked for) */
const char *SoapySDRDevice_lastError(void) { return "stub"; }
SoapySDRDevice *SoapySDRDevice_makeStrArgs(const char *a) { (void)a; return (SoapySDRDevice *)&g_feed; }
int SoapySDRDevice_unmake(SoapySDRDevice *d) { (void)d; return 0; }
int SoapySDRDevice_setGainMode(SoapySDRDevice *d, int dir, size_t c, bool a) { (void)d; (void)dir; (void)c; (void)a; return 0; }
int SoapySDRDevice_setGain(SoapySDRDevice *d, int dir, size_t c, double v) { (void)d; (void)dir; (void)c; (void)v; return 0; }
int SoapySDRDevice_setFrequencyCorrection(SoapySDRDevice *d, int dir, size_t c, double v) { (void)d; (void)dir; (void)c; (void)v; return 0; }
int SoapySDRDevice_setFrequency(SoapySDRDevice *d, int dir, size_t c, double f, const SoapySDRKwargs *a) { (void)d; (void)dir; (void)c; (void)f; (void)a; return 0; }
int SoapySDRDevice_setSampleRate(SoapySDRDevice *d, int dir, size_t c, double r) { (void)d; (void)dir; (void)c; (void)r; return 0; }
int SoapySDRDevice_setAntenna(SoapySDRDevice *d, int dir, size_t c, const char *n) { (void)d; (void)dir; (void)c; (void)n; return 0; }
SoapySDRStream *SoapySDRDevice_setupStream(SoapySDRDevice *d, int dir, const char *f, const size_t *ch, size_t n, const SoapySDRKwargs *a)
{ (void)d; (void)dir; (void)f; (void)ch; (void)n; (void)a; return (SoapySDRStream *)&g_feed_left; }
int SoapySDRDevice_closeStream(SoapySDRDevice *d, SoapySDRStream *s) { (void)d; (void)s; return 0; }
int SoapySDRDevice_activateStream(SoapySDRDevice *d, SoapySDRStream *s, int f, long long t, size_t n) { (void)d; (void)s; (void)f; (void)t; (void)n; return 0; }
int SoapySDRDevice_deactivateStream(SoapySDRDevice *d, SoapySDRStream *s, int f, long long t) { (void)d; (void)s; (void)f; (void)t; return 0; }
int SoapySDRDevice_readStream(SoapySDRDevice *d, SoapySDRStream *s, void *const *buffs, size_t numElems, int *flags, long long *timeNs, long timeoutUs)
{
	size_t n = numElems;
	(void)d; (void)s; (void)flags; (void)timeNs; (void)timeoutUs;
	if (g_feed_chunk && g_feed_chunk < n)
		n = g_feed_chunk;
	if (n > g_feed_left)
		n = g_feed_left;
	if (n == 0)
		return 0;                       /* soapy.c:221-227: the reader loop ends */
	memcpy(buffs[0], g_feed, n * 2 * sizeof(int16_t));
	g_feed += 2 * n;
	g_feed_left -= n;
	return (int)n;
}

void ref_soapy_run_reader(void);               /* ref_soapy_unit.c */

/* acarsdec -m mult -d <dev> f1 f2 ...  Returns the centre frequency chosen (soapy.c:139-140). */
long ref_init_soapy(int nfreq, const char **freqs, int mult)
{
	char *argv[MAXNBCHANNELS + 3];
	int i, r;
	if (nfreq > MAXNBCHANNELS)
		return -1;
	rateMult = mult;
	freq = 0;
	argv[0] = (char *)"driver=stub";
	for (i = 0; i < nfreq; i++)
		argv[1 + i] = (char *)freqs[i];
	argv[1 + nfreq] = NULL;
	r = initSoapy(argv, 0);
	if (r)
		return -2;
	common_init();
	return (long)freq;
}

/* feed n complex CS16 samples through the reference's reader loop, `chunk` per readStream call */
void ref_soapy_feed(const int16_t *iq, size_t n, size_t chunk)
{
	g_feed = iq;
	g_feed_left = n;
	g_feed_chunk = chunk;
	signalExit = 0;
	ref_soapy_run_reader();
	signalExit = 0;
}

int ref_get_oscillator(int n, float *out, int M)
{
	int i;
	for (i = 0; i < M; i++) {
		out[2 * i] = crealf(channel[n].oscillator[i]);
		out[2 * i + 1] = cimagf(channel[n].oscillator[i]);
	}
	return (int)channel[n].Fr;
}
/* Overwrites the oscillator table initSoapy built for channel n (soapy.c:163-166) with the caller's, like ref_set_wf on the
 * rtl.c path: the reference's reader loop on the very table another implementation was given.  Harness only. */
int ref_set_oscillator(int n, const float *taps, int ntaps)
{
	int i;
	if (n < 0 || (unsigned int)n >= nbch || ntaps < 0 || ntaps > rateMult)
		return -1;
	for (i = 0; i < rateMult; i++)
		channel[n].oscillator[i] = i < ntaps ? taps[2 * i] + taps[2 * i + 1] * I : 0;
	return 0;
}
#endif /* WITH_SOAPY */

#ifdef WITH_SDRPLAY
/* ---- SDRplay API stand-ins ---- */
mir_sdr_ErrT mir_sdr_ApiVersion(float *v) { *v = MIR_SDR_API_VERSION; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_GetDevices(mir_sdr_DeviceT *d, unsigned int *n, unsigned int m)
{ (void)m; d[0].SerNo = (char *)"0"; d[0].DevNm = (char *)"stub"; d[0].hwVer = 1; d[0].devAvail = 1; *n = 1; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetDeviceIdx(unsigned int i) { (void)i; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_ReleaseDeviceIdx(void) { return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_StreamInit(int *g, double fs, double rf, mir_sdr_Bw_MHzT bw, mir_sdr_If_kHzT i, int l, int *gs, mir_sdr_SetGrModeT m,
				int *spp, mir_sdr_StreamCallback_t s, mir_sdr_GainChangeCallback_t c, void *x)
{ (void)g; (void)fs; (void)rf; (void)bw; (void)i; (void)l; (void)gs; (void)m; (void)spp; (void)s; (void)c; (void)x; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_AgcControl(mir_sdr_AgcControlT e, int a, int b, unsigned int c, unsigned int d, int f, int g)
{ (void)e; (void)a; (void)b; (void)c; (void)d; (void)f; (void)g; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetPpm(double p) { (void)p; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetDcMode(int a, int b) { (void)a; (void)b; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetDcTrackTime(int t) { (void)t; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_DCoffsetIQimbalanceControl(unsigned int a, unsigned int b) { (void)a; (void)b; return mir_sdr_Success; }

int initSdrplay(char **argv, int optind);
void ref_sdrplay_callback(int16_t *xi, int16_t *xq, uint32_t n);     /* ref_sdrplay_unit.c */
unsigned int ref_sdrplay_fc(void);

long ref_init_sdrplay(int nfreq, const char **freqs)
{
	char *argv[MAXNBCHANNELS + 2];
	int i, r;
	if (nfreq > MAXNBCHANNELS)
		return -1;
	for (i = 0; i < nfreq; i++)
		argv[i] = (char *)freqs[i];
	argv[nfreq] = NULL;
	r = initSdrplay(argv, 0);
	if (r)
		return -2;
	common_init();
	return (long)ref_sdrplay_fc();
}

/* n samples per plane, delivered in callbacks of `chunk` samples (sdrplay.c:202-237 carries D) */
void ref_sdrplay_feed(const int16_t *xi, const int16_t *xq, size_t n, size_t chunk)
{
	size_t pos = 0;
	if (!chunk)
		chunk = n;
	while (pos < n) {
		size_t k = n - pos < chunk ? n - pos : chunk;
		ref_sdrplay_callback((int16_t *)xi + pos, (int16_t *)xq + pos, (uint32_t)k);
		pos += k;
	}
}

int ref_get_oscillator(int n, float *out, int M)
{
	int i;
	for (i = 0; i < M; i++) {
		out[2 * i] = crealf(channel[n].oscillator[i]);
		out[2 * i + 1] = cimagf(channel[n].oscillator[i]);
	}
	return (int)channel[n].Fr;
}
#endif /* WITH_SDRPLAY */

#ifdef WITH_AIR
/* ---- libairspy stand-ins: one device, one sample rate (set by the test before init) ---- */
static uint32_t g_air_rate = 2500000;
static uint32_t g_air_freq;
int airspy_list_devices(uint64_t *serials, int count) { if (serials && count > 0) serials[0] = 0x1234; return 1; }
int airspy_open_sn(struct airspy_device **d, uint64_t sn) { (void)sn; *d = (struct airspy_device *)&g_air_rate; return AIRSPY_SUCCESS; }
int airspy_open(struct airspy_device **d) { *d = (struct airspy_device *)&g_air_rate; return AIRSPY_SUCCESS; }
int airspy_close(struct airspy_device *d) { (void)d; return AIRSPY_SUCCESS; }
int airspy_exit(void) { return AIRSPY_SUCCESS; }
const char *airspy_error_name(int e) { (void)e; return "stub"; }
int airspy_set_sample_type(struct airspy_device *d, enum airspy_sample_type t) { (void)d; (void)t; return AIRSPY_SUCCESS; }
int airspy_get_samplerates(struct airspy_device *d, uint32_t *b, const uint32_t len)
{ (void)d; if (len == 0) *b = 1; else b[0] = g_air_rate; return AIRSPY_SUCCESS; }
int airspy_set_samplerate(struct airspy_device *d, uint32_t s) { (void)d; (void)s; return AIRSPY_SUCCESS; }
int airspy_set_packing(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_linearity_gain(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_vga_gain(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_freq(struct airspy_device *d, const uint32_t f) { (void)d; g_air_freq = f; return AIRSPY_SUCCESS; }
int airspy_r820t_write(struct airspy_device *d, uint8_t r, uint8_t v) { (void)d; (void)r; (void)v; return AIRSPY_SUCCESS; }
int airspy_start_rx(struct airspy_device *d, airspy_sample_block_cb_fn cb, void *c) { (void)d; (void)cb; (void)c; return AIRSPY_SUCCESS; }
int airspy_is_streaming(struct airspy_device *d) { (void)d; return 0; }

int ref_air_callback(float *samples, int count);                   /* ref_air_unit.c */
unsigned int ref_air_mult(void);

/* acarsdec -s <dev> f1 f2 ... with a stub Airspy offering `rate` samples/s.  Returns Fc (air.c:246). */
long ref_init_air(int nfreq, const char **freqs, unsigned int rate)
{
	char *argv[MAXNBCHANNELS + 3];
	int i, r;
	if (nfreq > MAXNBCHANNELS)
		return -1;
	g_air_rate = rate;
	argv[0] = (char *)"0";
	for (i = 0; i < nfreq; i++)
		argv[1 + i] = (char *)freqs[i];
	argv[1 + nfreq] = NULL;
	nbch = 0;
	r = initAirspy(argv, 0);
	if (r)
		return -2;
	common_init();
	return (long)g_air_freq;
}

void ref_air_feed(const float *x, size_t n, size_t chunk)
{
	size_t pos = 0;
	if (!chunk)
		chunk = n;
	while (pos < n) {
		size_t k = n - pos < chunk ? n - pos : chunk;
		ref_air_callback((float *)x + pos, (int)k);
		pos += k;
	}
}

int ref_get_wf(int n, float *out, int M)
{
	int i;
	for (i = 0; i < M; i++) {
		out[2 * i] = crealf(channel[n].wf[i]);
		out[2 * i + 1] = cimagf(channel[n].wf[i]);
	}
	return channel[n].Fr;
}
unsigned int ref_air_get_mult(void) { return ref_air_mult(); }
/* Overwrites the tap table initAirspy built for channel n (air.c:278-285) with the caller's.  Harness only. */
int ref_set_wf(int n, const float *taps, int ntaps)
{
	int i, M = (int)ref_air_mult();
	if (n < 0 || (unsigned int)n >= nbch || ntaps < 0 || ntaps > M)
		return -1;
	for (i = 0; i < M; i++)
		channel[n].wf[i] = i < ntaps ? taps[2 * i] + taps[2 * i + 1] * I : 0;
	return 0;
}
#endif /* WITH_AIR */

/* sound-file path (soundfile.c:30-56): nch channels of 12.5 kHz real samples */
int ref_init_file(int nch)
{
	int n;
	if (nch > MAXNBCHANNELS)
		return -1;
	nbch = nch;
	for (n = 0; n < nch; n++)
		channel[n].dm_buffer = malloc(sizeof(float) * 4096);
	common_init();
	return 0;
}

/* soundfile.c:71-77 for one channel: len <= 4096 */
void ref_demod(int n, const float *dm, int len)
{
	memcpy(channel[n].dm_buffer, dm, sizeof(float) * (size_t)len);
	demodMSK(&channel[n], len);
}

const float *ref_get_dm(int n) { return channel[n].dm_buffer; }

typedef struct {
	double MskPhi, MskDf, MskLvlSum;
	float MskClk;
	int MskBitCount;
	unsigned int MskS, idx;
	float inb[22];
	int outbits, nbits, Acarsstate;
} ref_state;

void ref_get_state(int n, ref_state *s)
{
	channel_t *ch = &channel[n];
	int i;
	s->MskPhi = ch->MskPhi; s->MskDf = ch->MskDf; s->MskLvlSum = ch->MskLvlSum;
	s->MskClk = ch->MskClk; s->MskBitCount = ch->MskBitCount;
	s->MskS = ch->MskS; s->idx = ch->idx;
	for (i = 0; i < 11; i++) {
		s->inb[2 * i] = crealf(ch->inb[i]);
		s->inb[2 * i + 1] = cimagf(ch->inb[i]);
	}
	s->outbits = ch->outbits; s->nbits = ch->nbits; s->Acarsstate = ch->Acarsstate;
}

int ref_nraw(void) { return g_nraw; }
int ref_nout(void) { return g_nout; }
const ref_frame *ref_raw(void) { return g_raw; }
const ref_frame *ref_out(void) { return g_out; }
unsigned int ref_nbch(void) { return nbch; }
