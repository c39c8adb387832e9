/*
 * demo_airspy_file.c -- an "Airspy" that plays a file of real float32 samples (env ACARSDEC_IQ_FILE; one sample rate, env
 * ACARSDEC_AIR_RATE, default 2 500 000) through the reference's UNCHANGED air.c, for the end-to-end drop-in demo of the Airspy
 * path (stub header: oracle/stub/libairspy/airspy.h).  Transfers have RAGGED sizes (never a whole number of decimation windows), so
 * the carry of air.c:299-338 (`ind`, ch->D) is exercised at every callback.
 *
 * It knows nothing about the GPU: every transfer goes to the callback air.c registers -- the reference's own rx_callback in the
 * CPU twin (oracle/_ref/acarsdec_cpu_air), the bound one in lib/acarsdec_gpu_air, whose air.c carries the one hunk of
 * INTEGRATION.md (applied to the reference's text at build time, acarsdec_amd/_build.py patched_source).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <libairspy/airspy.h>

static uint32_t g_rate;
static volatile int g_streaming;
static pthread_t g_thread;
static airspy_sample_block_cb_fn g_cb;

static uint32_t rate(void)
{
	if (!g_rate) {
		const char *r = getenv("ACARSDEC_AIR_RATE");
		g_rate = r ? (uint32_t)atoi(r) : 2500000u;
	}
	return g_rate;
}

int airspy_list_devices(uint64_t *serials, int count) { if (serials && count > 0) serials[0] = 0x1234; return 1; }
int airspy_open_sn(struct airspy_device **d, uint64_t sn) { (void)sn; *d = (struct airspy_device *)&g_rate; return AIRSPY_SUCCESS; }
int airspy_open(struct airspy_device **d) { *d = (struct airspy_device *)&g_rate; return AIRSPY_SUCCESS; }
int airspy_close(struct airspy_device *d) { (void)d; return AIRSPY_SUCCESS; }
int airspy_exit(void) { return AIRSPY_SUCCESS; }
const char *airspy_error_name(int e) { (void)e; return "file device"; }
int airspy_set_sample_type(struct airspy_device *d, enum airspy_sample_type t) { (void)d; (void)t; return AIRSPY_SUCCESS; }
int airspy_get_samplerates(struct airspy_device *d, uint32_t *b, const uint32_t len)
{
	(void)d;
	if (len == 0) *b = 1; else b[0] = rate();
	return AIRSPY_SUCCESS;
}
int airspy_set_samplerate(struct airspy_device *d, uint32_t s) { (void)d; (void)s; return AIRSPY_SUCCESS; }
int airspy_set_packing(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_linearity_gain(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_vga_gain(struct airspy_device *d, uint8_t v) { (void)d; (void)v; return AIRSPY_SUCCESS; }
int airspy_set_freq(struct airspy_device *d, const uint32_t f) { (void)d; (void)f; return AIRSPY_SUCCESS; }
int airspy_r820t_write(struct airspy_device *d, uint8_t r, uint8_t v) { (void)d; (void)r; (void)v; return AIRSPY_SUCCESS; }

static void *player(void *arg)
{
	const char *path = getenv("ACARSDEC_IQ_FILE");
	FILE *f = path ? fopen(path, "rb") : NULL;
	const size_t cap = 60000;                     /* <= 1024 windows per transfer at any rate >= 1 Msps (air.c:268: dm_buffer) */
	float *buf = malloc(cap * sizeof(float));
	unsigned int k = 0;
	(void)arg;
	if (!f || !buf)
		fprintf(stderr, "demo Airspy: set ACARSDEC_IQ_FILE to a real-float32 sample file\n");
	while (f && buf) {
		const size_t want = cap / 3 + (size_t)((k * 7919u + 13u) % (unsigned int)(cap - cap / 3));
		const size_t got = fread(buf, sizeof(float), want, f);
		airspy_transfer_t t;
		k++;
		if (got == 0)
			break;
		memset(&t, 0, sizeof(t));
		t.samples = buf;
		t.sample_count = (int)got;
		t.sample_type = AIRSPY_SAMPLE_FLOAT32_REAL;
		g_cb(&t);
	}
	if (f) fclose(f);
	free(buf);
	g_streaming = 0;
	return NULL;
}

int airspy_start_rx(struct airspy_device *d, airspy_sample_block_cb_fn cb, void *c)
{
	(void)d; (void)c;
	g_cb = cb;
	g_streaming = 1;
	return pthread_create(&g_thread, NULL, player, NULL) == 0 ? AIRSPY_SUCCESS : -1;
}
int airspy_is_streaming(struct airspy_device *d)
{
	(void)d;
	if (!g_streaming && g_thread) { pthread_join(g_thread, NULL); g_thread = 0; }
	return g_streaming ? AIRSPY_TRUE : 0;
}
