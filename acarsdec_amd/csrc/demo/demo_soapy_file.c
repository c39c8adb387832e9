/*
 * demo_soapy_file.c -- a "SoapySDR device" that plays a file of interleaved int16 I/Q (CS16, env ACARSDEC_IQ_FILE)
 * through the reference's soapy.c, for the end-to-end drop-in demo of the SoapySDR path (the stub headers are
 * oracle/stub/SoapySDR/).  Set-up calls do nothing; readStream hands out the file in reads of RAGGED size (never a whole
 * number of decimation windows), the way a real driver does -- so the window carry across reads (soapy.c:232-254:
 * `D`, `current_index`) is exercised; at end of file it reports a stream error, which is how soapy.c's reader ends.
 *
 * Built twice (acarsdec_amd/_build.py, oracle/Makefile):
 *   _ref/acarsdec_cpu_soapy   the reference's soapy.c UNCHANGED: its own CPU loop
 *   lib/acarsdec_gpu_soapy    soapy.c with the ONE hunk of INTEGRATION.md applied at build time (the per-channel loop of
 *                             soapy.c:228-254 replaced by a call of acarsdec_amd_soapy_samples(), compat_msk.c)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <SoapySDR/Device.h>
#include <SoapySDR/Formats.h>
#include <SoapySDR/Types.h>

static int g_dev, g_stream;
static FILE *g_file;
static unsigned int g_reads;

const char *SoapySDRDevice_lastError(void) { return "end of file"; }
SoapySDRDevice *SoapySDRDevice_makeStrArgs(const char *args) { (void)args; return (SoapySDRDevice *)&g_dev; }
int SoapySDRDevice_unmake(SoapySDRDevice *d) { (void)d; return 0; }
int SoapySDRDevice_setGainMode(SoapySDRDevice *d, int dir, size_t ch, bool a) { (void)d; (void)dir; (void)ch; (void)a; return 0; }
int SoapySDRDevice_setGain(SoapySDRDevice *d, int dir, size_t ch, double v) { (void)d; (void)dir; (void)ch; (void)v; return 0; }
int SoapySDRDevice_setFrequencyCorrection(SoapySDRDevice *d, int dir, size_t ch, double v) { (void)d; (void)dir; (void)ch; (void)v; return 0; }
int SoapySDRDevice_setFrequency(SoapySDRDevice *d, int dir, size_t ch, double f, const SoapySDRKwargs *a) { (void)d; (void)dir; (void)ch; (void)f; (void)a; return 0; }
int SoapySDRDevice_setSampleRate(SoapySDRDevice *d, int dir, size_t ch, double r) { (void)d; (void)dir; (void)ch; (void)r; return 0; }
int SoapySDRDevice_setAntenna(SoapySDRDevice *d, int dir, size_t ch, const char *n) { (void)d; (void)dir; (void)ch; (void)n; return 0; }
SoapySDRStream *SoapySDRDevice_setupStream(SoapySDRDevice *d, int dir, const char *fmt, const size_t *chans, size_t n, const SoapySDRKwargs *a)
{
	(void)d; (void)dir; (void)fmt; (void)chans; (void)n; (void)a;
	return (SoapySDRStream *)&g_stream;
}
int SoapySDRDevice_closeStream(SoapySDRDevice *d, SoapySDRStream *s) { (void)d; (void)s; return 0; }
int SoapySDRDevice_activateStream(SoapySDRDevice *d, SoapySDRStream *s, int fl, long long t, size_t n)
{
	const char *path = getenv("ACARSDEC_IQ_FILE");
	(void)d; (void)s; (void)fl; (void)t; (void)n;
	g_file = path ? fopen(path, "rb") : NULL;
	if (!g_file)
		fprintf(stderr, "demo SoapySDR device: set ACARSDEC_IQ_FILE to a CS16 file\n");
	return g_file ? 0 : -1;
}
int SoapySDRDevice_deactivateStream(SoapySDRDevice *d, SoapySDRStream *s, int fl, long long t) { (void)d; (void)s; (void)fl; (void)t; return 0; }

int SoapySDRDevice_readStream(SoapySDRDevice *d, SoapySDRStream *s, void *const *buffs, size_t numElems, int *flags, long long *timeNs,
			      long timeoutUs)
{
	size_t want, got;
	(void)d; (void)s; (void)flags; (void)timeNs; (void)timeoutUs;
	if (!g_file)
		return -1;
	/* between about a third of the buffer and all of it, never twice the same, hardly ever a multiple of the window */
	want = numElems / 3 + (size_t)((g_reads * 7919u + 13u) % (unsigned int)(numElems - numElems / 3));
	g_reads++;
	if (want < 1) want = 1;
	if (want > numElems) want = numElems;
	got = fread(buffs[0], 2 * sizeof(int16_t), want, g_file);
	if (got == 0) {
		fclose(g_file);
		g_file = NULL;
		return -1;                      /* soapy.c:222-227: a failed read ends the reader */
	}
	return (int)got;
}
