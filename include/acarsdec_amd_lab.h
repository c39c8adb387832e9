/*
 * acarsdec_amd_lab.h -- measurement and diagnostic entry points of libacarsdec_amd.so.  NOT product API: nothing here has a
 * counterpart in the reference, and a host that decodes ACARS needs none of it.  bench.py, the GPU tests and the probes under
 * profiles/probe/ use it: tuning switches for same-process A/B, timing trials, bandwidth probes, self tests of the device
 * arithmetic, device-side generators of synthetic input, and the test hook that moves the block counters next to their wrap.
 * Everything is exported by the product library as well (bench.py times the product), except where noted.
 */
#ifndef ACARSDEC_AMD_LAB_H
#define ACARSDEC_AMD_LAB_H

#include "acarsdec_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement / layout switches (ACG_PIPE_BLOCKS, ACG_MSK_LPC, ACG_FIR_WAVES_PER_WG, ...: profiles/LEDGER.md).  A production
 * process has none and pays one atomic load per look-up.  This call sets one (value NULL removes it).  The environment is
 * read ONCE per process, at the library's first look-up, and only if ACG_ALLOW_TUNING=1 is set too (otherwise stray ACG_*
 * variables are named on stderr and ignored).  The measurement-only kernel variants and debug shapes exist only in the lab
 * build of the library (libacarsdec_amd_lab.so, which tests and probes load; it always takes the environment).
 * Not product configuration. */
int  acg_tune(const char *name, const char *value);
int  acg_is_lab_build(void);
/* Diagnostic: times a context on a sample of the caller's input.  Runs one untimed and `repeats` timed acg_process_iq_u8_dev
 * calls back to back (host clock around a device sync), reports the mean in *ms_per_call, and returns the context to its reset
 * state (acg_reset: channel state, queues).  Rounds 2-3 used it to pick the fastest of several contexts; round 4 found that
 * contexts timed for >= 0.25 s of streaming calls agree to 2 % and that what a short trial sees between them is mostly the
 * trial (profiles/LEDGER.md): a host creates ONE context per device and uses it.  The reference has no counterpart. */
int  acg_placement_trial(acg_ctx *ctx, const uint8_t *iq_dev, size_t pitch_bytes, int nblocks, int repeats,
			 void *hip_stream, double *ms_per_call);

/* the same for the acg_process_samples_dev formats (declared below) */
int  acg_placement_trial_samples(acg_ctx *ctx, int fmt, const void *dev, size_t pitch_bytes, size_t plane_bytes, int nblocks,
				 int repeats, void *hip_stream, double *ms_per_call);

/* device-side generator for large synthetic workloads: fills [nstreams] rows with seeded
 * uniform bytes */
int  acg_fill_random_u8_dev(uint8_t *dev, size_t pitch_bytes, int nrows, size_t row_bytes,
			    uint64_t seed, void *hip_stream);

/* measurement aid: a pure streaming reader (non-temporal 16-byte loads, nothing else) over `bytes` of
 * device memory, `repeats` back-to-back launches timed with HIP events on the default stream: the read
 * bandwidth this GPU delivers to a kernel that only reads.  Synchronises. */
int  acg_probe_read_dev(const void *dev, size_t bytes, int repeats, double *gb_per_s);

/* measurement aid: `repeats` synchronous host-to-device copies of `bytes` (after one for nothing), host clock: the rate
 * the *_host entry points can at best be fed at from that host memory (pinned or pageable) */
int  acg_probe_h2d(void *dev, const void *host, size_t bytes, int repeats, double *gb_per_s);

/* device-side AM up-converter: row r = scale*env[env_index[r]][n/decim] *
 * exp(j(2*pi*off_hz[r]*n/(12500*decim) + phase[r])) + N(0, noise_sigma^2), quantised like an RTL
 * dongle (u8 = clip(rint(127.37 + 127.5 x))).  All pointers are device pointers. */
int  acg_synth_iq_u8_dev(uint8_t *iq_dev, size_t pitch_bytes, int nrows, int nout, int decim,
			 const float *env_dev, size_t env_pitch_floats, const int *env_index_dev,
			 const float *off_hz_dev, const float *phase_dev, float scale, float noise_sigma,
			 uint64_t seed, void *hip_stream);
/* diagnostics: the device sin/cos used by the mixer (msk.c:90 calls cexp), evaluated on the GPU
 * for n host arguments in [0, 2*pi) -- lets a test bound its error against libm */
int  acg_selftest_sincos(const double *x_host, double *sin_host, double *cos_host, int n);
/* diagnostics: the loop's f64 divisions and square root as the device computes them (msk.c:103,110-111: shared
 * reciprocal, no exponent scaling) next to the compiler's IEEE forms, for n host triples:
 * out[8i..8i+7] = {n0/d, n1/d (shared reciprocal), n0/d, n1/d (IEEE), sqrt(x), sqrt(x) (IEEE), n0/d (single), x}
 * with x = n0^2 + n1^2 -- a test checks the pairs are bit-identical over the operand range the loop produces */
int  acg_selftest_div2(const double *n0_host, const double *n1_host, const double *d_host, double *out_host, int n);


/* Test hook (the block ring across the wrap of its 32-bit counters): right after acg_reset, with nothing queued,
 * sets the device's block counter, the repair pass's mark and the host's consumer mark to `value`, as if that many blocks had
 * been produced and consumed already.  ACG_ESTATE if calls have been issued since the reset. */
int  acg_lab_set_block_counter(acg_ctx *ctx, unsigned int value);
/* length of the block ring (a power of two) */
unsigned int acg_lab_block_ring_size(const acg_ctx *ctx);

/* only in the stamp build (libacarsdec_amd_stamp.so, -DACG_MSK_STAMP): phase cycle sums of the LAST demodulator launch,
 * [waves][10], and the lanes per channel of the context's demodulator */
int  acg_msk_stamp_read(acg_ctx *ctx, unsigned long long *out, int nwaves);
int  acg_msk_lanes_per_channel(const acg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ACARSDEC_AMD_LAB_H */
