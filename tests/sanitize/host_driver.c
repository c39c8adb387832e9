/* host_driver.c -- exercises the host side of libacarsdec_amd (host_setup.c: tap builders, chooseFc, CRC / syndrome
 * tables, level; acg_api.cpp: argument validation and error paths of every entry point) in a build with
 * -fsanitize=address,undefined (the reference has no sanitizer coverage; tests/test_host_logic.py builds
 * and runs this on the CPU box).  No GPU is needed: without one acg_create() must fail with ACG_ENODEV, and every
 * entry point must reject a NULL context.  The kernel launchers are stubbed -- nothing here may reach them. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acarsdec_amd.h"
#include "acarsdec_amd_lab.h"

void acg_host_msk_h(float *h);
float acg_host_level_db(double lvlsum, int bitcount);
void acg_host_crc_tables(unsigned short *crc, unsigned short *synd);
void acg_host_crc_tables_n(unsigned short *crc, unsigned short *synd, int nk);

#define STUB(name) int name() { fprintf(stderr, "launcher " #name " reached without a GPU\n"); abort(); return -1; }
STUB(acg_launch_fir) STUB(acg_launch_fir_generic) STUB(acg_launch_fir_shared) STUB(acg_launch_regroup_taps)
STUB(acg_launch_fir_fmt) STUB(acg_launch_msk) STUB(acg_launch_msk2) STUB(acg_launch_blk_repair) STUB(acg_launch_sincos_selftest)
STUB(acg_launch_div2_selftest) STUB(acg_launch_msg_split) STUB(acg_launch_synth_iq) STUB(acg_launch_fill_random) STUB(acg_launch_read_probe)
STUB(acg_fir_mm_takes) STUB(acg_launch_fir_mm_prep) STUB(acg_launch_fir_mm)
STUB(acg_fir_mm1_takes) STUB(acg_launch_fir_mm1_prep) STUB(acg_launch_fir_mm1)
size_t acg_fir_mm1_image_bytes() { return 0; }
size_t acg_fir_lds_bytes() { return 0; }
size_t acg_fir_mm_image_bytes() { return 0; }

static int fails;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

static unsigned int lcg(unsigned int *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
	unsigned int seed = 12345;
	int M, i, k;
	/* tap builders at the documented rates and the size limits */
	static const int mults[] = {8, 160, 164, 192, 200, 320};
	for (k = 0; k < 6; ++k) {
		float *t;
		M = mults[k];
		t = malloc(sizeof(float) * 2 * (size_t)M);
		CHECK(acg_rtl_taps(131725000, 131850000u, M, t) == ACG_OK);
		CHECK(fabsf(hypotf(t[0], t[1]) - 1.0f / M / 127.5f) < 1e-9f);
		CHECK(acg_soapy_taps(131725000.0f, 131850000, M, t) == ACG_OK);
		free(t);
	}
	{
		float t[2 * 1024];
		CHECK(acg_rtl_taps(1, 1, 0, t) == ACG_EINVAL && acg_rtl_taps(1, 1, 100000, t) == ACG_EINVAL && acg_rtl_taps(1, 1, 8, NULL) == ACG_EINVAL);
		CHECK(acg_sdrplay_taps(131725000.0f, 131850000u, t) == ACG_OK);
		CHECK(acg_airspy_taps(131725000, 131850000, 2500000u, t) == ACG_OK && acg_airspy_taps(131725000, 131850000, 10000000u, t) == ACG_OK);
		CHECK(acg_airspy_choose_fc(131525000u, 131825000u) != 0);
	}
	/* chooseFc on random and hard sets: 1 .. 16 channels, too far apart, duplicates */
	for (i = 0; i < 2000; ++i) {
		unsigned int fd[16], n = 1 + lcg(&seed) % 16, j, fc;
		int mult = (lcg(&seed) & 1) ? 160 : 200;
		for (j = 0; j < n; ++j)
			fd[j] = 131000000u + 12500u * (lcg(&seed) % ((i & 3) ? 80 : 400));
		fc = acg_rtl_choose_fc(fd, n, mult);
		for (j = 1; j < n; ++j)
			CHECK(fd[j - 1] <= fd[j]);                      /* sorted in place, rtl.c:135-145 */
		if (fc)
			for (j = 0; j < n; ++j)
				CHECK(abs((int)fc - (int)fd[j]) <= 12500 * mult / 2);
	}
	/* CRC table, syndromes (the reference's 242 rows and the row beyond), matched-filter prototype, level */
	{
		unsigned short crc[256], synd[8 * 243], synd2[8 * 242];
		float h[136];
		acg_host_crc_tables_n(crc, synd, 243);
		acg_host_crc_tables(crc, synd2);
		CHECK(memcmp(synd, synd2, sizeof(synd2)) == 0 && crc[1] == 0x1189);
		acg_host_msk_h(h);
		CHECK(h[66] == 1.0f && h[0] == 0.0f && h[132] == 0.0f);
		CHECK(isfinite(acg_host_level_db(3.5, 120)) && acg_host_level_db(0.0, 0) != 12345.0f);
	}
	/* the ABI's error behaviour: bad configurations, no GPU, NULL contexts */
	{
		acg_ctx *c = (acg_ctx *)1;
		acg_config cfg;
		acg_frame fr[2];
		acg_chan_state st;
		int n = -1, rc;
		double ms = 0;
		float buf[16] = {0};
		memset(&cfg, 0, sizeof(cfg));
		CHECK(acg_create(&c, &cfg) == ACG_EINVAL && c == NULL);
		cfg.nch = 4; cfg.nstreams = 5; cfg.decim = 200; cfg.ntaps = 200; cfg.max_blocks = 1;
		CHECK(acg_create(&c, &cfg) == ACG_EINVAL);
		cfg.nstreams = 1; cfg.ntaps = 201;
		CHECK(acg_create(&c, &cfg) == ACG_EINVAL);
		cfg.ntaps = 200;
		rc = acg_create(&c, &cfg);
		CHECK(rc == ACG_ENODEV || rc == ACG_OK);                /* OK only on a box with a GPU */
		if (rc == ACG_OK)
			acg_destroy(c);
		CHECK(acg_create(NULL, &cfg) == ACG_EINVAL && acg_create(&c, NULL) == ACG_EINVAL);
		for (rc = 1; rc >= -8; --rc)
			CHECK(acg_strerror(rc) != NULL && strlen(acg_strerror(rc)) > 0);
		CHECK(strlen(acg_version()) > 0 && acg_last_error(NULL)[0] == 0);
		acg_destroy(NULL);
		CHECK(acg_reset(NULL) == ACG_EINVAL && acg_sync(NULL) == ACG_EINVAL);
		CHECK(acg_set_taps(NULL, 0, 1, buf) == ACG_EINVAL && acg_set_channel_streams(NULL, &n) == ACG_EINVAL);
		CHECK(acg_process_iq_u8_dev(NULL, (const uint8_t *)buf, 64, 1, NULL) == ACG_EINVAL);
		CHECK(acg_process_iq_u8_host(NULL, (const uint8_t *)buf, 64, 1) == ACG_EINVAL);
		CHECK(acg_process_dm_dev(NULL, buf, 16, 16, NULL) == ACG_EINVAL && acg_process_dm_host(NULL, buf, 16, 16) == ACG_EINVAL);
		CHECK(acg_fir_only_dev(NULL, (const uint8_t *)buf, 64, 1, NULL) == ACG_EINVAL);
		{
			double ms = 0;
			CHECK(acg_placement_trial(NULL, (const uint8_t *)buf, 64, 1, 1, NULL, &ms) == ACG_EINVAL);
		}
		CHECK(acg_process_samples_dev(NULL, ACG_FMT_CS16, buf, 64, 0, 1, NULL) == ACG_EINVAL);
		CHECK(acg_feed_samples_host(NULL, ACG_FMT_CS16, buf, NULL, 4, 4) == ACG_EINVAL);
		CHECK(acg_drain_frames(NULL, fr, 2, &n) == ACG_EINVAL && acg_collect_frames(NULL, 1, fr, 2, &n) == ACG_EINVAL);
		CHECK(acg_drain_msgs(NULL, NULL, 0, &n) == ACG_EINVAL && acg_collect_msgs(NULL, 1, NULL, 0, &n) == ACG_EINVAL);
		CHECK(acg_max_lag(NULL) == 0);
		{ double ms; CHECK(acg_placement_trial_samples(NULL, ACG_FMT_CS16, NULL, 0, 0, 1, 1, NULL, &ms) == ACG_EINVAL); }
		CHECK(acg_tune("PATH", "x") == ACG_EINVAL && acg_tune(NULL, "1") == ACG_EINVAL);
		CHECK(acg_tune("ACG_FIR_VARIANT", "55") == ACG_OK && acg_tune("ACG_FIR_VARIANT", NULL) == ACG_OK && acg_tune("ACG_NOT_SET", NULL) == ACG_OK);
		CHECK(acg_read_bits(NULL, 0, buf, buf, 4, &n) == ACG_EINVAL && acg_read_bits_all(NULL, &n, buf, buf) == ACG_EINVAL);
		CHECK(acg_bit_capacity(NULL) == 0 && acg_read_dm(NULL, 0, buf, 4) == ACG_EINVAL);
		CHECK(acg_get_state(NULL, 0, &st) == ACG_EINVAL && acg_set_state(NULL, 0, &st) == ACG_EINVAL);
		CHECK(acg_get_state_n(NULL, 0, 1, &st) == ACG_EINVAL && acg_set_state_n(NULL, 0, 1, &st) == ACG_EINVAL);
		CHECK(acg_read_dm_n(NULL, 0, 1, buf, 4, 4) == ACG_EINVAL);
		CHECK(acg_lab_set_block_counter(NULL, 1u) == ACG_EINVAL && acg_lab_block_ring_size(NULL) == 0u);
		CHECK(acg_replay_bits(NULL, NULL, NULL) == ACG_EINVAL);
		CHECK(acg_get_timing(NULL, &ms, &n, &ms, &n) == ACG_EINVAL && acg_set_timing(NULL, 1) == ACG_EINVAL);
		CHECK(acg_fill_random_u8_dev(NULL, 16, 1, 16, 1, NULL) == ACG_EINVAL);
		CHECK(acg_synth_iq_u8_dev(NULL, 0, 0, 0, 0, NULL, 0, NULL, NULL, NULL, 0.f, 0.f, 0, NULL) == ACG_EINVAL);
		CHECK(acg_probe_read_dev(NULL, 0, 0, NULL) == ACG_EINVAL);
		CHECK(acg_selftest_sincos(NULL, NULL, NULL, 0) == ACG_EINVAL && acg_selftest_div2(NULL, NULL, NULL, NULL, 0) == ACG_EINVAL);
	}
	if (fails) {
		fprintf(stderr, "%d check(s) failed\n", fails);
		return 1;
	}
	puts("sanitized host driver: ok");
	return 0;
}
