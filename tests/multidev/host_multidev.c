/*
 * host_multidev.c -- a C host that drives SEVERAL GPUs through the C ABI, the way the reference's one DSP thread
 * iterates its channels (rtl.c:344-360: `for (n = 0; n < nbch; n++)` over channel[n]): one acg_ctx per device, channel c
 * belongs to context c mod N (BASELINE configs[3]'s round-robin shard), ONE host thread issues the call of every context
 * and then collects every context's results one call behind (acg_collect_*(lag = 1)), so that all devices stay busy while
 * the host merges.  No collective, no second process: the channels are independent.
 *
 * On a box with fewer GPUs than contexts (the 1-GPU rehearsal of tests/test_gpu_multidev.py) context k runs on device
 * k mod acg_device_count(): N contexts share device 0 -- the same code path, the same shard arithmetic.
 *
 *   host_multidev <iq.u8> <taps.f32> <nch> <decim> <nblk> <cb> <N> [--host] [--msgs] [--time R]
 *
 *   iq.u8     [nch][nblk*1024*decim*2] bytes: one u8 I/Q stream per channel (rtl.c:330 layout), whole callbacks;
 *             or the word `random`: every shard is filled on its device with seeded bytes (acg_fill_random_u8_dev) --
 *             for timing runs at widths no file should hold
 *   taps.f32  [nch][decim][2] floats (wf[] of rtl.c:283-286 per channel); or the word `rtl`: acg_rtl_taps() at a
 *             channel-dependent offset
 *   cb        callbacks per acg_process call, N = contexts
 *   --host    feed from host memory with acg_process_iq_u8_host (buffer reusable on return, rtl.c:314-330 semantics)
 *             instead of uploading every shard once and calling acg_process_iq_u8_dev
 *   --msgs    ACG_F_REPAIR + acg_collect_msgs (the delivered records) instead of raw blocks
 *   --time R  afterwards R more passes over the input: all contexts together, then every context alone in turn
 *   --shared-input  (with `random`, measurement aid) all contexts of one device read ONE input buffer instead of a buffer each:
 *             separates "the context" from "the (input buffer, output buffer) pair" when equal contexts run at unequal rates
 *
 * stdout: one line per block / message, merged over all contexts and ordered by (global channel, end_bit):
 *   B <chn> <end_bit> <len> <err> <crc hex> <txt hex>          or          M <chn> <end_bit> <err> <mode> <addr> <label> <bid> <txt hex>
 * stderr: the shard table and the timing.  Exit code 0, or 2 with the library's message on any error (3: no GPU).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "acarsdec_amd.h"

typedef struct {
	int chn;                 /* GLOBAL channel */
	long long end_bit;
	char *line;
} rec_t;

static rec_t *g_rec;
static size_t g_nrec, g_caprec;

static void die(const char *what, acg_ctx *c, int rc)
{
	fprintf(stderr, "host_multidev: %s failed: %s (%s)\n", what, acg_strerror(rc), c ? acg_last_error(c) : "");
	exit(rc == ACG_ENODEV ? 3 : 2);
}

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void push(int chn, long long end_bit, char *line)
{
	if (g_nrec == g_caprec) {
		g_caprec = g_caprec ? 2 * g_caprec : 4096;
		g_rec = realloc(g_rec, g_caprec * sizeof(*g_rec));
		if (!g_rec) { fprintf(stderr, "out of memory\n"); exit(2); }
	}
	g_rec[g_nrec].chn = chn;
	g_rec[g_nrec].end_bit = end_bit;
	g_rec[g_nrec].line = line;
	g_nrec++;
}

static void hex(char *dst, const unsigned char *src, int n)
{
	static const char d[] = "0123456789abcdef";
	int i;
	for (i = 0; i < n; i++) { dst[2 * i] = d[src[i] >> 4]; dst[2 * i + 1] = d[src[i] & 15]; }
	dst[2 * n] = 0;
}

static int by_chn_bit(const void *a, const void *b)
{
	const rec_t *x = a, *y = b;
	if (x->chn != y->chn) return x->chn < y->chn ? -1 : 1;
	return x->end_bit < y->end_bit ? -1 : x->end_bit > y->end_bit;
}

typedef struct {
	acg_ctx *ctx;
	int k, dev, nk;          /* context index, device, channels owned: k, k + N, k + 2N, ... */
	uint8_t *d_iq;           /* device copy of the shard's rows (dev mode) */
} shard_t;

static int N, use_host, use_msgs, shared_input;
static acg_frame *fbuf;
static acg_msg *mbuf;
static int cap;

/* everything context s has ready `lag` calls behind (lag < 0: drain all), relabelled with the global channel id */
static long collect(shard_t *s, int lag)
{
	long total = 0;
	int n, rc, i;
	do {
		if (use_msgs) {
			rc = lag < 0 ? acg_drain_msgs(s->ctx, mbuf, cap, &n) : acg_collect_msgs(s->ctx, lag, mbuf, cap, &n);
			if (rc != ACG_OK && rc != ACG_EAGAIN) die("collect_msgs", s->ctx, rc);
			for (i = 0; i < n; i++) {
				const acg_msg *m = &mbuf[i];
				char *l = malloc(96 + 2 * (size_t)m->txt_len);
				int g = m->chn * N + s->k, o;
				o = sprintf(l, "M %d %lld %d %c %s %s %c ", g, m->end_bit, m->err, m->mode ? m->mode : '-', m->addr, m->label,
					    m->bid ? m->bid : '-');
				hex(l + o, (const unsigned char *)m->txt, m->txt_len);
				push(g, m->end_bit, l);
			}
		} else {
			rc = lag < 0 ? acg_drain_frames(s->ctx, fbuf, cap, &n) : acg_collect_frames(s->ctx, lag, fbuf, cap, &n);
			if (rc != ACG_OK && rc != ACG_EAGAIN) die("collect_frames", s->ctx, rc);
			for (i = 0; i < n; i++) {
				const acg_frame *f = &fbuf[i];
				int len = f->len < 0 ? 0 : f->len > ACG_TXTMAX ? ACG_TXTMAX : f->len;
				char *l = malloc(64 + 2 * (size_t)len);
				int g = f->chn * N + s->k, o;
				o = sprintf(l, "B %d %lld %d %d %02x%02x ", g, f->end_bit, f->len, f->err, f->crc[0], f->crc[1]);
				hex(l + o, f->txt, len);
				push(g, f->end_bit, l);
			}
		}
		total += n;
	} while (rc == ACG_EAGAIN);
	return total;
}

int main(int argc, char **argv)
{
	const char *iq_path, *taps_path;
	int nch, decim, nblk, cb, i, k, j, ndev, ncall, rc, reps = 0, random_iq;
	size_t row, cb_bytes;
	uint8_t *iq;
	float *taps;
	shard_t *sh;
	FILE *f;
	double t0, t1;

	if (argc < 8) {
		fprintf(stderr, "usage: host_multidev <iq.u8> <taps.f32> <nch> <decim> <nblk> <cb> <N> [--host] [--msgs] [--time R]\n");
		return 2;
	}
	iq_path = argv[1]; taps_path = argv[2];
	nch = atoi(argv[3]); decim = atoi(argv[4]); nblk = atoi(argv[5]); cb = atoi(argv[6]); N = atoi(argv[7]);
	for (i = 8; i < argc; i++) {
		if (!strcmp(argv[i], "--host")) use_host = 1;
		else if (!strcmp(argv[i], "--msgs")) use_msgs = 1;
		else if (!strcmp(argv[i], "--shared-input")) shared_input = 1;
		else if (!strcmp(argv[i], "--time") && i + 1 < argc) reps = atoi(argv[++i]);
	}
	if (nch < 1 || decim < 8 || nblk < 1 || cb < 1 || nblk % cb || N < 1 || N > nch) {
		fprintf(stderr, "host_multidev: bad arguments (nblk must be a multiple of cb, 1 <= N <= nch)\n");
		return 2;
	}
	/* Several contexts on ONE device (the rehearsal): the HIP runtime maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues per
	 * device and priority, round robin, and streams that share a queue serialise against each other -- with 4 contexts (12 pooled
	 * streams) the stages of different contexts stop overlapping: all together 1.89 M channel*Msps with 4 queues, 2.48 M with 8
	 * (profiles/r04_multidev_hw_queues.txt).  The variable is read when the runtime starts, so the HOST sets it, before its first
	 * HIP call.  One context per device (the deployment) stays inside the default. */
	if (N > 1)
		setenv("GPU_MAX_HW_QUEUES", "16", 0);
	ndev = acg_device_count();
	if (ndev < 1) die("acg_device_count", NULL, ACG_ENODEV);          /* no CPU fallback anywhere */
	row = (size_t)nblk * ACG_BLOCK * decim * 2;
	cb_bytes = (size_t)cb * ACG_BLOCK * decim * 2;
	ncall = nblk / cb;
	random_iq = !strcmp(iq_path, "random");
	if (random_iq && use_host) { fprintf(stderr, "host_multidev: `random` input is generated on the devices (no --host)\n"); return 2; }
	iq = random_iq ? NULL : malloc((size_t)nch * row);
	taps = malloc((size_t)nch * decim * 2 * sizeof(float));
	if ((!iq && !random_iq) || !taps) { fprintf(stderr, "out of memory\n"); return 2; }
	if (!random_iq) {
		if (!(f = fopen(iq_path, "rb")) || fread(iq, row, (size_t)nch, f) != (size_t)nch) { fprintf(stderr, "cannot read %s\n", iq_path); return 2; }
		fclose(f);
	}
	if (!strcmp(taps_path, "rtl")) {
		for (i = 0; i < nch; i++)                   /* offsets on the 12.5 kHz raster, >= 25 kHz from the centre (rtl.c:131-168) */
			if ((rc = acg_rtl_taps(131000000 + 25000 * (2 + i % 40) * ((i & 1) ? -1 : 1), 131000000u, decim, taps + (size_t)i * decim * 2)) != ACG_OK)
				die("acg_rtl_taps", NULL, rc);
	} else {
		if (!(f = fopen(taps_path, "rb")) || fread(taps, sizeof(float) * 2 * decim, (size_t)nch, f) != (size_t)nch) { fprintf(stderr, "cannot read %s\n", taps_path); return 2; }
		fclose(f);
	}

	cap = 4 * nch + 1024;
	fbuf = malloc((size_t)cap * sizeof(*fbuf));
	mbuf = malloc((size_t)cap * sizeof(*mbuf));
	sh = calloc((size_t)N, sizeof(*sh));
	if (!fbuf || !mbuf || !sh) { fprintf(stderr, "out of memory\n"); return 2; }

	/* ---- one context per device (rehearsal: per device slot), channel c -> context c mod N, local index c / N */
	for (k = 0; k < N; k++) {
		shard_t *s = &sh[k];
		acg_config cfg;
		memset(&cfg, 0, sizeof(cfg));
		s->k = k;
		s->dev = k % ndev;
		s->nk = (nch - k + N - 1) / N;
		cfg.device = s->dev; cfg.nch = s->nk; cfg.nstreams = s->nk; cfg.decim = decim; cfg.ntaps = decim;
		cfg.max_blocks = cb; cfg.flags = use_msgs ? ACG_F_REPAIR : 0; cfg.max_lag = 1;
		if ((rc = acg_create(&s->ctx, &cfg)) != ACG_OK) die("acg_create", NULL, rc);
		for (j = 0; j < s->nk; j++)
			if ((rc = acg_set_taps(s->ctx, j, 1, taps + (size_t)(j * N + k) * decim * 2)) != ACG_OK) die("acg_set_taps", s->ctx, rc);
		if (!use_host && shared_input && random_iq && k >= ndev) {
			s->d_iq = sh[k % ndev].d_iq;              /* the buffer of the first context on this device (it is the largest shard) */
		} else if (!use_host) {
			/* the shard's rows (k, k + N, ...) uploaded once: source pitch N rows, destination dense */
			if (hipSetDevice(s->dev) != hipSuccess || hipMalloc((void **)&s->d_iq, (size_t)s->nk * row) != hipSuccess ||
			    (random_iq ? (acg_fill_random_u8_dev(s->d_iq, row, s->nk, row, 0xACA25u + (unsigned)k, NULL) != ACG_OK || hipDeviceSynchronize() != hipSuccess)
				       : hipMemcpy2D(s->d_iq, row, iq + (size_t)k * row, (size_t)N * row, row, (size_t)s->nk, hipMemcpyHostToDevice) != hipSuccess)) {
				fprintf(stderr, "host_multidev: upload of shard %d failed: %s\n", k, hipGetErrorString(hipGetLastError()));
				return 2;
			}
		}
		fprintf(stderr, "context %d: device %d of %d, %d channels (global %d, %d, ...), max_lag %d\n", k, s->dev, ndev, s->nk, k, k + N,
			acg_max_lag(s->ctx));
	}

	/* ---- the decode pass: issue on every context, then collect every context one call behind */
	t0 = now_ms();
	for (j = 0; j < ncall; j++) {
		for (k = 0; k < N; k++) {
			shard_t *s = &sh[k];
			if (use_host)
				rc = acg_process_iq_u8_host(s->ctx, iq + (size_t)k * row + (size_t)j * cb_bytes, (size_t)N * row, cb);
			else
				rc = acg_process_iq_u8_dev(s->ctx, s->d_iq + (size_t)j * cb_bytes, row, cb, NULL);
			if (rc != ACG_OK) die("acg_process_iq_u8", s->ctx, rc);
		}
		for (k = 0; k < N; k++) collect(&sh[k], 1);
	}
	for (k = 0; k < N; k++) collect(&sh[k], -1);
	t1 = now_ms();
	qsort(g_rec, g_nrec, sizeof(*g_rec), by_chn_bit);
	for (i = 0; i < (int)g_nrec; i++) puts(g_rec[i].line);
	fprintf(stderr, "decoded %zu %s from %d channels on %d context(s) in %.2f ms (first pass, includes first-touch)\n", g_nrec,
		use_msgs ? "messages" : "blocks", nch, N, t1 - t0);

	/* ---- timing: R passes with all contexts in flight, then every context alone in turn */
	if (reps > 0) {
		const double samples = (double)nch * nblk * ACG_BLOCK * decim;
		int r;
		size_t keep = g_nrec;
		t0 = now_ms();
		for (r = 0; r < reps; r++)
			for (j = 0; j < ncall; j++) {
				for (k = 0; k < N; k++) {
					shard_t *s = &sh[k];
					rc = use_host ? acg_process_iq_u8_host(s->ctx, iq + (size_t)k * row + (size_t)j * cb_bytes, (size_t)N * row, cb)
						      : acg_process_iq_u8_dev(s->ctx, s->d_iq + (size_t)j * cb_bytes, row, cb, NULL);
					if (rc != ACG_OK) die("acg_process_iq_u8", s->ctx, rc);
				}
				for (k = 0; k < N; k++) collect(&sh[k], 1);
			}
		for (k = 0; k < N; k++) collect(&sh[k], -1);
		t1 = now_ms();
		fprintf(stderr, "all %d contexts together: %.1f channel*Msamples/s (%.3f ms per pass)\n", N, samples * reps / (t1 - t0) / 1e3,
			(t1 - t0) / reps);
		for (k = 0; k < N; k++) {
			shard_t *s = &sh[k];
			for (j = 0; j < 2 * ncall; j++) {                     /* a turn for nothing: the context's first calls after the others' */
				rc = use_host ? acg_process_iq_u8_host(s->ctx, iq + (size_t)k * row + (size_t)(j % ncall) * cb_bytes, (size_t)N * row, cb)
					      : acg_process_iq_u8_dev(s->ctx, s->d_iq + (size_t)(j % ncall) * cb_bytes, row, cb, NULL);
				if (rc != ACG_OK) die("acg_process_iq_u8", s->ctx, rc);
				collect(s, 1);
			}
			collect(s, -1);
			t0 = now_ms();
			for (r = 0; r < reps; r++)
				for (j = 0; j < ncall; j++) {
					rc = use_host ? acg_process_iq_u8_host(s->ctx, iq + (size_t)k * row + (size_t)j * cb_bytes, (size_t)N * row, cb)
						      : acg_process_iq_u8_dev(s->ctx, s->d_iq + (size_t)j * cb_bytes, row, cb, NULL);
					if (rc != ACG_OK) die("acg_process_iq_u8", s->ctx, rc);
					collect(s, 1);
				}
			collect(s, -1);
			t1 = now_ms();
			fprintf(stderr, "context %d alone: %.1f channel*Msamples/s (%.4f ms per call)\n", k,
				(double)s->nk * nblk * ACG_BLOCK * decim * reps / (t1 - t0) / 1e3, (t1 - t0) / reps / ncall);
		}
		for (i = (int)keep; i < (int)g_nrec; i++) free(g_rec[i].line);
		g_nrec = keep;
	}
	for (k = 0; k < N; k++) {
		acg_destroy(sh[k].ctx);
		if (sh[k].d_iq && !(shared_input && k >= ndev)) { hipSetDevice(sh[k].dev); hipFree(sh[k].d_iq); }
	}
	return 0;
}
