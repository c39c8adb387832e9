/*
 * wav_frontend.c -- a libsndfile-free stand-in for the reference's sound-file front end, used
 * only to build the end-to-end demo binary (reference acarsdec.c/acars.c/output.c unchanged +
 * compat_msk.c) in an image without libsndfile.  Same two entry points and the same behaviour
 * as soundfile.c:30-81: PCM16 frames are scaled by 1/32768 (libsndfile's sf_read_float
 * normalisation), read in chunks of 4096 frames, de-interleaved into channel[n].dm_buffer and
 * handed to demodMSK() channel by channel; returns -1 at end of file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "acarsdec.h"

#define MAXNBFRAMES 4096

static FILE *g_f;
static long g_frames_left;
static int g_nch;

static uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int initSoundfile(char **argv, int optind)
{
	unsigned char hdr[12], ck[8], fmt[40];
	int have_fmt = 0, rate = 0, bits = 0, tag = 0;
	unsigned int n;

	g_f = fopen(argv[optind], "rb");
	if (g_f == NULL || fread(hdr, 1, 12, g_f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) {
		fprintf(stderr, "could not open %s\n", argv[optind]);
		return 1;
	}
	while (fread(ck, 1, 8, g_f) == 8) {
		uint32_t sz = rd32(ck + 4);
		if (!memcmp(ck, "fmt ", 4)) {
			uint32_t take = sz < sizeof(fmt) ? sz : (uint32_t)sizeof(fmt);
			memset(fmt, 0, sizeof(fmt));
			if (fread(fmt, 1, take, g_f) != take)
				return 1;
			fseek(g_f, (long)(sz - take) + (sz & 1), SEEK_CUR);
			tag = rd16(fmt); g_nch = rd16(fmt + 2); rate = (int)rd32(fmt + 4); bits = rd16(fmt + 14);
			if (tag == 0xFFFE)
				tag = rd16(fmt + 24);              /* WAVE_FORMAT_EXTENSIBLE sub-format */
			have_fmt = 1;
		} else if (!memcmp(ck, "data", 4)) {
			if (!have_fmt || tag != 1 || bits != 16) {
				fprintf(stderr, "unsupported wav format\n");
				return 1;
			}
			g_frames_left = (long)(sz / (2u * (unsigned)g_nch));
			break;
		} else {
			fseek(g_f, (long)sz + (sz & 1), SEEK_CUR);
		}
	}
	nbch = (unsigned int)g_nch;
	if (nbch > MAXNBCHANNELS) {
		fprintf(stderr, "Too much input channels : %d\n", nbch);
		return 1;
	}
	if (rate != INTRATE) {
		fprintf(stderr, "unsupported sample rate : %d (must be %d)\n", rate, INTRATE);
		return 1;
	}
	for (n = 0; n < nbch; n++)
		channel[n].dm_buffer = malloc(sizeof(float) * MAXNBFRAMES);
	return 0;
}

int runSoundfileSample(void)
{
	static int16_t pcm[MAXNBFRAMES * MAXNBCHANNELS];
	unsigned int n;

	for (;;) {
		long want = g_frames_left < MAXNBFRAMES ? g_frames_left : MAXNBFRAMES;
		long got = want > 0 ? (long)fread(pcm, 2 * (size_t)g_nch, (size_t)want, g_f) : 0;
		int i, len = (int)got;
		if (got <= 0)
			return -1;
		g_frames_left -= got;
		for (n = 0; n < nbch; n++) {
			for (i = 0; i < len; i++)
				channel[n].dm_buffer[i] = (float)pcm[n + (size_t)i * nbch] / 32768.0f;
			demodMSK(&channel[n], len);
		}
	}
}
