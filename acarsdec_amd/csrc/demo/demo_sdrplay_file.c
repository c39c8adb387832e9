/*
 * demo_sdrplay_file.c -- an "SDRplay RSP" that plays a file of int16 I/Q PAIRS (interleaved on disk, env ACARSDEC_IQ_FILE;
 * handed to the callback as separate xi[] / xq[] arrays like the vendor API does) through the reference's UNCHANGED sdrplay.c
 * (stub header: oracle/stub/mirsdrapi-rsp.h).  Packets have ragged sizes (the carry of sdrplay.c:215-236 is exercised).
 * The reference's runSdrplaySample() never returns (`while (1) sleep(2)`, sdrplay.c:284-285): the player ends the process at end of
 * file, after giving the block thread a moment to print what is queued.
 *
 * It knows nothing about the GPU: every packet goes to the callback sdrplay.c registers -- the reference's own myStreamCallback in
 * the CPU twin (oracle/_ref/acarsdec_cpu_sdrplay), the bound one in lib/acarsdec_gpu_sdrplay, whose sdrplay.c carries the one hunk
 * of INTEGRATION.md (applied to the reference's text at build time, acarsdec_amd/_build.py patched_source).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <pthread.h>
#include <mirsdrapi-rsp.h>

static mir_sdr_StreamCallback_t g_cb;
static pthread_t g_thread;

mir_sdr_ErrT mir_sdr_ApiVersion(float *v) { if (v) *v = MIR_SDR_API_VERSION; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_GetDevices(mir_sdr_DeviceT *devs, unsigned int *n, unsigned int max)
{
	static char ser[] = "FILE0001", nm[] = "file";
	if (n) *n = 1;
	if (devs && max > 0) { devs[0].SerNo = ser; devs[0].DevNm = nm; devs[0].hwVer = 2; devs[0].devAvail = 1; }
	return mir_sdr_Success;
}
mir_sdr_ErrT mir_sdr_SetDeviceIdx(unsigned int i) { (void)i; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_ReleaseDeviceIdx(void) { return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_AgcControl(mir_sdr_AgcControlT e, int a, int b, unsigned int c, unsigned int d, int f, int g)
{ (void)e; (void)a; (void)b; (void)c; (void)d; (void)f; (void)g; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetPpm(double p) { (void)p; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetDcMode(int a, int b) { (void)a; (void)b; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_SetDcTrackTime(int t) { (void)t; return mir_sdr_Success; }
mir_sdr_ErrT mir_sdr_DCoffsetIQimbalanceControl(unsigned int a, unsigned int b) { (void)a; (void)b; return mir_sdr_Success; }

void acarsdec_amd_compat_print_stats(void) __attribute__((weak));

static void *player(void *arg)
{
	const char *path = getenv("ACARSDEC_IQ_FILE");
	FILE *f = path ? fopen(path, "rb") : NULL;
	const size_t cap = 40000;                     /* < 512 windows of 160 per packet (sdrplay.c:227: demodMSK every 512 outputs) */
	int16_t *pair = malloc(cap * 2 * sizeof(int16_t)), *xi = malloc(cap * sizeof(int16_t)), *xq = malloc(cap * sizeof(int16_t));
	unsigned int k = 0, first = 0;
	(void)arg;
	if (!f || !pair || !xi || !xq)
		fprintf(stderr, "demo RSP: set ACARSDEC_IQ_FILE to an interleaved int16 I/Q file\n");
	while (f && pair && xi && xq) {
		const size_t want = cap / 3 + (size_t)((k * 7919u + 13u) % (unsigned int)(cap - cap / 3));
		const size_t got = fread(pair, 2 * sizeof(int16_t), want, f);
		size_t i;
		k++;
		if (got == 0)
			break;
		for (i = 0; i < got; i++) { xi[i] = pair[2 * i]; xq[i] = pair[2 * i + 1]; }
		g_cb(xi, xq, first, 0, 0, 0, (unsigned int)got, 0, 0, NULL);
		first += (unsigned int)got;
	}
	usleep(500 * 1000);                           /* the block thread prints what is queued (acars.c:93-215) */
	fflush(stdout);
	if (acarsdec_amd_compat_print_stats && getenv("ACARSDEC_AMD_STATS"))      /* (only the GPU twin links compat_msk.c) */
		acarsdec_amd_compat_print_stats();
	_exit(0);                                     /* sdrplay.c:284-285 has no way out of its run loop */
	return NULL;
}

mir_sdr_ErrT mir_sdr_StreamInit(int *gRdB, double fsMHz, double rfMHz, mir_sdr_Bw_MHzT bw, mir_sdr_If_kHzT ift, int lna, int *gsys,
				mir_sdr_SetGrModeT mode, int *spp, mir_sdr_StreamCallback_t cb, mir_sdr_GainChangeCallback_t gcb, void *ctx)
{
	(void)gRdB; (void)fsMHz; (void)rfMHz; (void)bw; (void)ift; (void)lna; (void)mode; (void)gcb; (void)ctx;
	if (gsys) *gsys = 40;
	if (spp) *spp = 504;
	g_cb = cb;
	return pthread_create(&g_thread, NULL, player, NULL) == 0 ? mir_sdr_Success : mir_sdr_Fail;
}
