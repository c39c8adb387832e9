/*
 * demo_rtlsdr_file.c -- a "dongle" that plays a file of interleaved u8 I/Q (env ACARSDEC_IQ_FILE)
 * through the reference's UNCHANGED rtl.c, for the end-to-end drop-in demo of the RTL path.
 *
 * It knows nothing about the GPU: every buffer goes to the callback rtl.c hands to
 * rtlsdr_read_async() -- the reference's own in_callback in the CPU twin (oracle/_ref/acarsdec_cpu_rtl),
 * the bound callback in lib/acarsdec_gpu_rtl, whose rtl.c carries the one hunk of INTEGRATION.md
 * (applied to the reference's text at build time, acarsdec_amd/_build.py patched_source).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rtl-sdr.h"

static int g_dev;
static volatile int g_cancel;

uint32_t rtlsdr_get_device_count(void) { return 1; }
const char *rtlsdr_get_device_name(uint32_t i) { (void)i; return "iq-file"; }
int rtlsdr_get_device_usb_strings(uint32_t i, char *m, char *p, char *s)
{
	(void)i;
	if (m) strcpy(m, "file");
	if (p) strcpy(p, "file");
	if (s) strcpy(s, "00000000");
	return 0;
}
int rtlsdr_open(rtlsdr_dev_t **dev, uint32_t i) { (void)i; *dev = (rtlsdr_dev_t *)&g_dev; return 0; }
int rtlsdr_close(rtlsdr_dev_t *dev) { (void)dev; return 0; }
int rtlsdr_set_center_freq(rtlsdr_dev_t *dev, uint32_t f) { (void)dev; (void)f; return 0; }
int rtlsdr_set_freq_correction(rtlsdr_dev_t *dev, int p) { (void)dev; (void)p; return 0; }
int rtlsdr_get_tuner_gains(rtlsdr_dev_t *dev, int *g) { (void)dev; if (g) g[0] = 0; return 1; }
int rtlsdr_set_tuner_gain(rtlsdr_dev_t *dev, int g) { (void)dev; (void)g; return 0; }
int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *dev, int m) { (void)dev; (void)m; return 0; }
int rtlsdr_set_sample_rate(rtlsdr_dev_t *dev, uint32_t r) { (void)dev; (void)r; return 0; }
int rtlsdr_reset_buffer(rtlsdr_dev_t *dev) { (void)dev; return 0; }
int rtlsdr_cancel_async(rtlsdr_dev_t *dev) { (void)dev; g_cancel = 1; return 0; }

int rtlsdr_read_async(rtlsdr_dev_t *dev, rtlsdr_read_async_cb_t cb, void *ctx, uint32_t buf_num, uint32_t buf_len)
{
	const char *path = getenv("ACARSDEC_IQ_FILE");
	FILE *f = path ? fopen(path, "rb") : NULL;
	unsigned char *buf = malloc(buf_len);
	(void)dev; (void)buf_num;
	if (!f || !buf) {
		fprintf(stderr, "demo dongle: set ACARSDEC_IQ_FILE to a u8 I/Q file\n");
		free(buf);
		return -1;
	}
	while (!g_cancel && fread(buf, 1, buf_len, f) == buf_len) {
		cb(buf, buf_len, ctx);
	}
	fclose(f);
	free(buf);
	return 0;
}
