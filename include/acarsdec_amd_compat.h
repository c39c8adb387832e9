/*
 * acarsdec_amd_compat.h -- entry points of the legacy view (acarsdec_amd/csrc/compat_msk.c), the counterpart of the
 * front-end declarations in the reference's acarsdec.h:152-191.  compat_msk.c is compiled INSIDE the reference tree (it needs
 * channel_t) and linked instead of msk.c; it defines initMsk() / demodMSK() with the reference's own prototypes
 * (acarsdec.h:190-191, declared there) and the functions below, one per front end whose DSP it replaces.  The one-hunk
 * bindings are in INTEGRATION.md; _build.py applies each of them to the reference text at build time and the GPU tests run the
 * resulting programs against their CPU twins.
 *
 * All of them work on the reference's globals (channel[], nbch, and the front end's rate multiplier) exactly like the code
 * they replace, run the GPU path for all channels of the dongle, and replay every decided bit through the UNCHANGED
 * decodeAcars().  There is no CPU fallback: on any library error they print the reason and exit(1), like the reference does on
 * "Unable to init internal decoders" (acarsdec.c:456-459).
 */
#ifndef ACARSDEC_AMD_COMPAT_H
#define ACARSDEC_AMD_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rtl.c:314-361 in_callback(): pass it to rtlsdr_read_async() at rtl.c:364 (same rtlsdr_read_async_cb_t signature) */
void acarsdec_amd_in_callback(unsigned char *rtlinbuff, uint32_t nread, void *ctx);
/* soapy.c:228-254: the per-channel loop of the reader thread, for one SoapySDRDevice_readStream() result of `nsamples` CS16
 * samples (any count; the partial window is carried on the device) */
void acarsdec_amd_soapy_samples(const int16_t *iq, int nsamples);
/* air.c:291-341 rx_callback(): the transfer's real float32 samples; airmult = air.c's AIRMULT (static there) */
void acarsdec_amd_air_samples(const float *samples, int count, int airmult);
/* sdrplay.c:215-236 myStreamCallback(): numSamples int16 I and Q samples */
void acarsdec_amd_sdrplay_samples(const int16_t *xi, const int16_t *xq, int nsamples);

/* rtl.c:353 leaves every channel's 12.5 kHz magnitudes in channel[n].dm_buffer; nothing in the reference reads them after
 * demodMSK() has run, so the legacy view does not copy them back from the device unless asked to (on != 0) */
void acarsdec_amd_compat_keep_dm(int on);
/* host time spent inside the legacy entry points so far and the number of calls (a harness reports ms per callback
 * against the 81.92 ms a callback's signal lasts, rtl.c:49,213) */
void acarsdec_amd_compat_stats(double *seconds, unsigned long *calls);
/* the line ACARSDEC_AMD_STATS=1 prints at exit, on demand (a front end whose run loop never returns, sdrplay.c:284-285, ends the
 * process with _exit(): no atexit handlers) */
void acarsdec_amd_compat_print_stats(void);

#ifdef __cplusplus
}
#endif
#endif /* ACARSDEC_AMD_COMPAT_H */
